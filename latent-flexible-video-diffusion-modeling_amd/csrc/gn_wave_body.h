// One-wave GroupNorm32(+FiLM)(+activation) of a (sample, 16 channels) slice (see norm_embed.hip, "gn_wave"), shared by
// the per-launch kernel and the persistent level chain (level_chain.hip).
#pragma once
#include "common_hip.h"

namespace {

__device__ __forceinline__ f32x4 ld_cat(const float* s0, const float* s1, int C0, int C1, size_t pos, int c) {
    return c < C0 ? ld4(s0 + pos * C0 + c) : ld4(s1 + pos * C1 + (c - C0));
}
// the same element through `sc1` loads (level chain: the sources were written by other workgroups of this launch)
__device__ __forceinline__ f32x4 ld_cat_sc1(const float* s0, const float* s1, int C0, int C1, unsigned pos, int c) {
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)s0, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(s1 ? s1 : s0), 0, 0x7fffffff, 0x00020000);
    const unsigned o0 = (pos * (unsigned)C0 + (unsigned)c) * 4u, o1 = (pos * (unsigned)C1 + (unsigned)(c - C0)) * 4u;
    return c < C0 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r0, (int)o0, 0, 16))
                  : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, (int)o1, 0, 16));
}

// CG = channels per group: 2, 4, 8, 16.  One wave: sample n, channels [16 cb, 16 cb + 16).  CHAIN: sources read and the
// output written with sc1 (write-through) accesses - see ChainCtx in conv_igemm_body.h.
// NW = 4 (round 6, maps of >= 128 positions): the FOUR waves of a workgroup share the unit, wave `wsub` takes a quarter of
// the positions (4 float4 per lane instead of 16: a quarter of the 64 SiLU evaluations and 16 stores a lone wave issues one
// instruction at a time) and the two group sums are completed through 256 B of LDS - two barriers, partials added in wave
// order (deterministic).
template <int CG, bool CHAIN, int NW = 1>
__device__ __forceinline__ void gn_wave_body(
    int n, int cb, int lane, const float* __restrict__ s0, const float* __restrict__ s1, int C0, int C1, int P,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ film, int film_div,
    int film_ld, float eps, float* __restrict__ coefA, float* __restrict__ coefB, float* __restrict__ stats,
    float* __restrict__ act_out, int act_mode, int ldo = 0, int wsub = 0) {
    constexpr int KEEP = 16 / NW;
    const int C = C0 + C1;
    if (ldo == 0) ldo = C;              // row stride of act_out (a part of a wider tensor: lfvdm_gn_apply_part)
    const int q = lane & 3, pl = lane >> 2;
    const int c = cb * 16 + q * 4;
    const int Pw = NW > 1 ? P / NW : P;                 // positions of this wave (NW > 1: P is a multiple of 16 * NW)
    const size_t pos0 = (size_t)n * P + (size_t)wsub * Pw;
    P = Pw;                                             // (below: this wave's share; the divisor is taken from Pall)
    const int Pall = Pw * NW;
    // coefficient operands first: their latency hides behind the statistics
    const f32x4 gam = ld4(gamma + c), bet = ld4(beta + c);
    f32x4 fsc = {0.f, 0.f, 0.f, 0.f}, fsh = fsc;
    if (film) {
        const float* f = film + (size_t)(n / film_div) * film_ld;
        fsc = ld4(f + c);
        fsh = ld4(f + C + c);
    }
    f32x4 keep[KEEP];
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
        const int p = pl + 16 * i;
        if constexpr (CHAIN) keep[i] = p < P ? ld_cat_sc1(s0, s1, C0, C1, (unsigned)(pos0 + p), c) : (f32x4){0.f, 0.f, 0.f, 0.f};
        else keep[i] = p < P ? ld_cat(s0, s1, C0, C1, pos0 + p, c) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // sum over a group: the lane's own channels of the group, the pixel lanes (xor 4 ... 32), the quad lanes of the group
    auto group_sum = [&](f32x4 v) -> f32x4 {
        if constexpr (CG == 2) { const float a = v.x + v.y, b = v.z + v.w; v = (f32x4){a, a, b, b}; }
        else { const float a = (v.x + v.y) + (v.z + v.w); v = (f32x4){a, a, a, a}; }
#pragma unroll
        for (int off = 4; off < 64; off <<= 1) {
            v.x += __shfl_xor(v.x, off, 64);
            if constexpr (CG == 2) v.z += __shfl_xor(v.z, off, 64);
        }
        if constexpr (CG >= 8) v.x += __shfl_xor(v.x, 1, 64);
        if constexpr (CG >= 16) v.x += __shfl_xor(v.x, 2, 64);
        if constexpr (CG == 2) return (f32x4){v.x, v.x, v.z, v.z};
        else return (f32x4){v.x, v.x, v.x, v.x};
    };
    const float inv = 1.0f / (float)(CG * Pall);
    // NW > 1: the waves' partial group sums -> LDS -> every lane adds the NW partials of its channel quad in wave order
    auto unit_sum = [&](f32x4 v, int pass) -> f32x4 {
        v = group_sum(v);
        if constexpr (NW > 1) {
            __shared__ f32x4 s_xch[2][NW][4];         // (only in the NW > 1 instances: the chain kernel's LDS budget is tight)
            if (pl == 0) s_xch[pass][wsub][q] = v;
            __syncthreads();
            v = s_xch[pass][0][q];
#pragma unroll
            for (int w = 1; w < NW; ++w) v += s_xch[pass][w][q];
        }
        return v;
    };
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KEEP; ++i) sum += keep[i];
    const f32x4 mean = unit_sum(sum, 0) * inv;
    f32x4 sq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
        if (pl + 16 * i < P) {
            const f32x4 d = keep[i] - mean;
            sq += d * d;
        }
    }
    const f32x4 var = unit_sum(sq, 1) * inv;
    f32x4 rstd;
    rstd.x = 1.0f / sqrtf(var.x + eps); rstd.y = 1.0f / sqrtf(var.y + eps);
    rstd.z = 1.0f / sqrtf(var.z + eps); rstd.w = 1.0f / sqrtf(var.w + eps);
    f32x4 A = rstd * gam;
    f32x4 B = bet - mean * A;
    if (film) {
        const f32x4 sc = fsc + (f32x4){1.f, 1.f, 1.f, 1.f};
        A = A * sc;
        B = B * sc + fsh;
    }
    if (pl == 0 && wsub == 0) {
        if (coefA) {
            st4(coefA + (size_t)n * C + c, A);
            st4(coefB + (size_t)n * C + c, B);
        }
        if (stats) {       // (mean, rstd) per (sample, group): the first channel of a group reports
            float* st = stats + (size_t)n * 64;
            if constexpr (CG == 2) {
                st[2 * (c / 2)] = mean.x; st[2 * (c / 2) + 1] = rstd.x;
                st[2 * (c / 2 + 1)] = mean.z; st[2 * (c / 2 + 1) + 1] = rstd.z;
            } else if (c % CG == 0) {
                st[2 * (c / CG)] = mean.x; st[2 * (c / CG) + 1] = rstd.x;
            }
        }
    }
    if (act_out == nullptr) return;
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
        const int p = pl + 16 * i;
        if (p < P) {
            f32x4 v = keep[i] * A + B;
            if (act_mode == LFVDM_ACT_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
            if constexpr (CHAIN)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v),
                                                       __builtin_amdgcn_make_buffer_rsrc((void*)act_out, 0, 0x7fffffff, 0x00020000),
                                                       (int)(((unsigned)(pos0 + p) * (unsigned)ldo + (unsigned)c) * 4u), 0, 16 /* sc1 */);
            else st4(act_out + (pos0 + p) * ldo + c, v);
        }
    }
}


}  // namespace
