// Spatial attention with the qkv projection inside (reference rpe.py:139-169, spatial instance: no RPE terms, no mask):
//
//   qkv = x_norm W^T + b          x_norm [P tokens][C] of ONE frame, W = qkv.weight rows {q, k, v} x head h
//   o   = softmax(q k^T / sqrt(F)) v
//
// Workgroup = (frame n, head h), 4 waves.  The separate qkv GEMM wrote 3C floats per token to memory and the attention
// kernel read them back block by block through registers (four load -> ds_write -> barrier rounds per 64 queries, a
// fourth of the MFMA rate); here the frame's normalised tokens and the head's 3F filter rows are staged ONCE by LDS-DMA
// (buffer_load ... lds, XOR-swizzled 16-byte slots: conflict-free b128 fragment reads), q / k / v of the head are
// produced on fp32 MFMA 16x16x4 straight into LDS tiles, and the flash loop then runs over resident keys with no
// barrier and no staging.  One launch less per attention block, no qkv round trip (7.9 MB at 16x16 / 64 channels).
//
// MFMA 16x16x4 operand mapping (as in attention.hip): A lane (i = l & 15, k = l >> 4), B lane (k = l >> 4, j = l & 15),
// D lane: column j = l & 15, rows 4 * (l >> 4) + r.  Inside a 16-wide group g of the reduction index, step e of a lane
// group kk = l >> 4 uses index 16 g + 4 kk + e on both operands (any fixed permutation of the k order is a dot product).
#include <cstdlib>

#include "common_hip.h"

namespace {

constexpr unsigned kOOBs = 0x40000000u;

__device__ __forceinline__ bool sf_wg(int N, int heads, int& n, int& h) {
    // frames round-robin over the 8 XCDs, all heads of a frame on one XCD (they read the same token rows)
    const int L = blockIdx.x;
    const int xcd = L & 7, r = L >> 3;
    const int slot = r / heads;
    h = r - slot * heads;
    n = slot * 8 + xcd;
    return n < N;
}

// FP = head dim (multiple of 16); LDS (floats): Xs [PP][C] | Ws [3 FP][C] | Qs [PP][FP + 8] | Ks [PP][FP + 8] | Vs [PP][FP + 4]
template <int FP>
__global__ __launch_bounds__(1024) void attn_spatial_fused_kernel(const float* __restrict__ xn, const float* __restrict__ W,
                                                                 const float* __restrict__ bias, float* __restrict__ o,
                                                                 int N, int P, int C, int heads) {
    constexpr int FG = FP / 16;
    constexpr int QLD = FP + 8, VLD = FP + 4;
    extern __shared__ __attribute__((aligned(16))) float sf_smem[];
    int n, h;
    if (!sf_wg(N, heads, n, h)) return;                    // workgroup-uniform (grid padding)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = (int)(blockDim.x >> 6);                 // 4 ... 16 waves share the staged images: up to four waves per SIMD
                                                           // cover each other's LDS and dependent-MFMA latency
    const int lq = lane & 15, kk = lane >> 4;
    const int PP = (P + 15) & ~15;
    const int SPR = C >> 2;                                // 16-byte slots per row (>= 16: C >= 64)
    float* Xs = sf_smem;
    float* Ws = Xs + (size_t)PP * C;
    float* Qs = Ws + (size_t)3 * FP * C;
    float* Ks = Qs + (size_t)PP * QLD;
    float* Vs = Ks + (size_t)PP * QLD;

    // ---- stage the frame's tokens and the head's filter rows: pieces of 64 slots (1 KiB), round-robin over the waves.
    // Physical slot s of row r holds logical slot s ^ (r & 15); token rows >= P and slots past the image read zeros
    // (offset beyond the descriptor's num_records).
    {
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(xn + (size_t)n * P * C), 0, P * C * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 3 * C * C * 4, 0x00020000);
        const int sh = 31 - __builtin_clz(SPR);            // SPR is a power of two times ...: rows per piece below
        const int nX = (PP * SPR + 63) >> 6, nW = (3 * FP * SPR + 63) >> 6;
        const bool pow2 = (SPR & (SPR - 1)) == 0;
        for (int pc = wave; pc < nX + nW; pc += NW) {
            const bool isX = pc < nX;
            const int ps = (isX ? pc : pc - nX) * 64 + lane;
            int r, s;
            if (pow2) { r = ps >> sh; s = ps & (SPR - 1); }
            else { r = (int)(((float)ps + 0.5f) * __builtin_amdgcn_rcpf((float)SPR)); s = ps - r * SPR; }
            const int ls = (s & ~15) | ((s ^ r) & 15);     // logical slot stored at physical slot s
            unsigned off;
            if (isX) {
                off = r < P ? (unsigned)(r * C + ls * 4) * 4u : kOOBs;
            } else {
                const int which = r / FP, f = r - which * FP;               // FP is a compile-time constant
                off = r < 3 * FP ? (unsigned)((which * C + h * FP + f) * C + ls * 4) * 4u : kOOBs;
            }
            float* dst = (isX ? Xs : Ws) + (size_t)(isX ? pc : pc - nX) * 256;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(isX ? rsX : rsW, (__attribute__((address_space(3))) void*)dst, 16, (int)off, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- projection: [PP][3 FP] = Xs [PP][C] . Ws^T.  Work item = (16-token row tile, 16-wide feature group fg): the q, k
    // and v tiles of that group share the token fragment and run as three independent MFMA chains; bias added, q pre-scaled
    {
        // q carries 1/sqrt(F) AND log2(e): the softmax below exponentiates in base 2 (one v_exp_f32 per logit, no multiply)
        const float scale = rsqrtf((float)FP) * 1.4426950408889634f;
        const int NRT = PP >> 4;
        const int CG = C >> 4;                              // 16-wide reduction groups
        for (int item = wave; item < NRT * FG; item += NW) {
            const int rt = item / FG, fg = item - rt * FG;
            const int xr = 16 * rt + lq;
            const float* xrow = Xs + (size_t)xr * C;
            int wr[3];
            const float* wrow[3];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                wr[w] = w * FP + 16 * fg + lq;
                wrow[w] = Ws + (size_t)wr[w] * C;
            }
            f32x4 acc[3];
#pragma unroll
            for (int w = 0; w < 3; ++w) acc[w] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int g = 0; g < CG; ++g) {
                const int s = 4 * g + kk;
                const f32x4 a4 = ld4(xrow + (((s & ~15) | ((s ^ xr) & 15)) << 2));
                f32x4 b4[3];
#pragma unroll
                for (int w = 0; w < 3; ++w) b4[w] = ld4(wrow[w] + (((s & ~15) | ((s ^ wr[w]) & 15)) << 2));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int w = 0; w < 3; ++w) acc[w] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], b4[w][e], acc[w], 0, 0, 0);
            }
            // D: column j = lq (feature 16 fg + lq), rows 4 kk + r (tokens 16 rt + 4 kk + r)
            const int f = 16 * fg + lq;
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const float bv = bias[w * C + h * FP + f];
                float* dst = w == 0 ? Qs : (w == 1 ? Ks : Vs);
                const int ld = w == 2 ? VLD : QLD;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[w][r] + bv;
                    if (w == 0) v *= scale;
                    dst[(size_t)(16 * rt + 4 * kk + r) * ld + f] = v;
                }
            }
        }
    }
    __syncthreads();

    // ---- flash loop over the resident keys: one 16-query tile per wave at a time, key blocks of 64
    const int NQT = PP >> 4;
    for (int qt = wave; qt < NQT; qt += NW) {
        const int q = 16 * qt + lq;
        f32x4 qf[FG], oacc[FG], oacc2[FG];          // two output chains (even / odd key tiles) per feature group
#pragma unroll
        for (int g = 0; g < FG; ++g) {
            qf[g] = ld4(Qs + (size_t)q * QLD + 16 * g + 4 * kk);
            oacc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            oacc2[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        float m_run = -INFINITY, l_run = 0.f;
        for (int kb = 0; kb < P; kb += 64) {
            f32x4 s[4];
            int keyr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                keyr[j] = min(kb + 16 * j + lq, PP - 1);
            }
#pragma unroll
            for (int g = 0; g < FG; ++g) {                  // four key tiles = four independent MFMA chains
                f32x4 k4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) k4[j] = ld4(Ks + (size_t)keyr[j] * QLD + 16 * g + 4 * kk);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[j][e], qf[g][e], s[j], 0, 0, 0);
            }
            float mx = -INFINITY;
            if (kb + 64 > P) {                              // ragged last block only (wave-uniform)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kb + 16 * j + 4 * kk + r >= P) s[j][r] = -INFINITY;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[j][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);      // m_run = -inf on the first block -> 0
            float psum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(s[j][r] - m_new);
                    s[j][r] = pv;
                    psum += pv;
                }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int g = 0; g < FG; ++g) { oacc[g] *= alpha; oacc2[g] *= alpha; }
            // O^T[f][q] += V^T[f][key] * P^T[key][q]; MFMA (j, r): k index kk <-> key kb + 16 j + 4 kk + r
#pragma unroll
            for (int j = 0; j < 4; j += 2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key0 = min(kb + 16 * j + 4 * kk + r, PP - 1);      // probabilities of keys >= P are exactly 0
                    const int key1 = min(kb + 16 * (j + 1) + 4 * kk + r, PP - 1);
                    const float* v0 = Vs + (size_t)key0 * VLD + lq;
                    const float* v1 = Vs + (size_t)key1 * VLD + lq;
#pragma unroll
                    for (int g = 0; g < FG; ++g) {
                        oacc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[16 * g], s[j][r], oacc[g], 0, 0, 0);
                        oacc2[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1[16 * g], s[j + 1][r], oacc2[g], 0, 0, 0);
                    }
                }
        }
        l_run += __shfl_xor(l_run, 16, 64);
        l_run += __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_run;
        if (q < P) {
#pragma unroll
            for (int g = 0; g < FG; ++g) st4(o + ((size_t)n * P + q) * C + h * FP + 16 * g + 4 * kk, (oacc[g] + oacc2[g]) * inv);
        }
    }
}

inline size_t sf_lds_bytes(int P, int C, int FP) {
    const size_t PP = (size_t)((P + 15) & ~15);
    // images are filled in whole 1 KiB pieces: round each of the two DMA images up to 256 floats
    const size_t xs = (PP * C + 255) & ~(size_t)255, ws = ((size_t)3 * FP * C + 255) & ~(size_t)255;
    return (xs + ws + 2 * PP * (FP + 8) + PP * (FP + 4)) * sizeof(float);
}

}  // namespace

// 0 = the fused launch covers this shape; LFVDM_E_UNSUPPORTED = use lfvdm_conv_igemm (qkv) + lfvdm_attn_spatial
extern "C" int lfvdm_attn_spatial_fused_ok(int N, int P, int C, int heads) {
    if (N <= 0 || P <= 0 || heads <= 0 || C % heads) return LFVDM_E_SHAPE;
    const int F = C / heads;
    if ((F != 16 && F != 32 && F != 64) || C % 64) return LFVDM_E_UNSUPPORTED;      // rows of whole 16-slot swizzle groups
    if ((long)P * C * 4 >= (1L << 30) || sf_lds_bytes(P, C, F) > 160 * 1024) return LFVDM_E_UNSUPPORTED;
    return LFVDM_OK;
}

extern "C" int lfvdm_attn_spatial_fused(const float* xn, const float* Wqkv, const float* bqkv, float* o, int N, int P, int C,
                                        int heads, void* stream) {
    if (!xn || !Wqkv || !bqkv || !o) return LFVDM_E_SHAPE;
    if (int rc = lfvdm_attn_spatial_fused_ok(N, P, C, heads)) return rc;
    const int F = C / heads;
    const size_t lds = sf_lds_bytes(P, C, F);
    const unsigned grid = 8u * (unsigned)((N + 7) / 8) * (unsigned)heads;
    // waves per workgroup: one per 16-query tile (and per projection item), 4 ... 16
    const int PP = (P + 15) & ~15, items = (PP / 16) * (F / 16);
    int nw = PP / 16 > items ? PP / 16 : items;
    nw = nw < 4 ? 4 : nw > 16 ? 16 : nw;
    static const int force_nw = getenv("LFVDM_SF_WAVES") ? atoi(getenv("LFVDM_SF_WAVES")) : 0;     // A/B aid
    if (force_nw >= 1 && force_nw <= 16) nw = force_nw;
    hipStream_t s = (hipStream_t)stream;
#define LFVDM_SF(FV)                                                                                                         \
    {                                                                                                                        \
        static DynLdsLimit limit;                                                                                            \
        if (int rc = limit.ensure(reinterpret_cast<const void*>(&attn_spatial_fused_kernel<FV>), lds)) return rc;            \
        hipLaunchKernelGGL((attn_spatial_fused_kernel<FV>), dim3(grid), dim3(64 * nw), lds, s, xn, Wqkv, bqkv, o, N, P, C, heads); \
    }
    if (F == 16) LFVDM_SF(16)
    else if (F == 32) LFVDM_SF(32)
    else LFVDM_SF(64)
#undef LFVDM_SF
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}
