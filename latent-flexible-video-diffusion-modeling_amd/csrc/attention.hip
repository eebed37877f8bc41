// Attention cores and the RPE networks for gfx950.
//
//  * rpe_nets_kernel   : every RPENet output projection of one forward (rpe.py:20-31) as one
//                        grouped launch; hidden rows are generated on the fly into LDS and
//                        multiplied with Wout on fp32 MFMA 32x32x2.
//  * attn_spatial      : flash-style MHA over the H*W tokens of a frame on fp32 MFMA 16x16x4.
//                        S^T = K.Q^T is computed with keys on the accumulator rows, so the
//                        softmax probabilities are already the B operand of O^T = V^T.P^T
//                        (no LDS round trip for P); softmax reductions are wave shuffles.
//  * attn_temporal     : one wave per (batch, pixel, head); T <= 32 frames attend with the three
//                        RPE terms and the two-clique mask (rpe.py:143-169) on the VALU.
#include "common_hip.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// ======================================================================================
// RPE nets
// ======================================================================================
constexpr int RPE_LDR = 36;

__global__ __launch_bounds__(256) void rpe_nets_kernel(const lfvdm_rpe_job* __restrict__ jobs, int njobs,
                                                       const int64_t* __restrict__ fi, int B, int T) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int j = 0;
    while (j + 1 < njobs && jobs[j + 1].tile0 <= (int)blockIdx.x) ++j;
    const lfvdm_rpe_job J = jobs[j];
    const int C = J.C;
    const int M = B * T * T;
    const int m0 = ((int)blockIdx.x - J.tile0) * 32;
    const int ALD = C + 4;                       // A tile row stride (floats), 16B aligned, conflict-free b128
    float* As = smem;                            // [32][ALD] silu(hidden)
    float* Wst = smem + 32 * ALD + wave * 32 * RPE_LDR;  // wave-private W chunk [32][36]

    // ---- hidden rows: silu(tproj[b] + Wd feats + bd) ----
    __shared__ float rowf[32][4];  // f0, f1, f2, batch index (or -1 past the end)
    if (threadIdx.x < 32) {
        const int m = m0 + threadIdx.x;
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, fb = -1.f;
        if (m < M) {
            const int b = m / (T * T);
            const int rem = m - b * T * T;
            const int t = rem / T, s = rem - t * T;
            const float d = (float)(fi[b * T + t] - fi[b * T + s]);
            f0 = log1pf(fmaxf(d, 0.f));
            f1 = log1pf(fmaxf(-d, 0.f));
            f2 = d == 0.f ? 1.f : 0.f;
            fb = (float)b;
        }
        rowf[threadIdx.x][0] = f0; rowf[threadIdx.x][1] = f1; rowf[threadIdx.x][2] = f2; rowf[threadIdx.x][3] = fb;
    }
    __syncthreads();
    // thread = (row wave + 4 i, column lane + 64 j): no per-element index division, the column's three feature weights
    // and bias loaded once for its 8 rows (same arithmetic per element as before)
    for (int k = lane; k < C; k += 64) {
        const float w0 = J.Wd[k * 3 + 0], w1 = J.Wd[k * 3 + 1], w2 = J.Wd[k * 3 + 2], bk = J.bd[k];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave + 4 * i;
            float v = 0.f;
            if (rowf[r][3] >= 0.f) {
                const int b = (int)rowf[r][3];
                const float hd = (rowf[r][0] * w0 + rowf[r][1] * w1 + rowf[r][2] * w2) + bk;
                v = silu_f(J.tproj[b * J.tproj_ld + k] + hd);
                if (J.act != nullptr) J.act[(size_t)(m0 + r) * C + k] = v;
            }
            As[r * ALD + k] = v;
        }
    }
    __syncthreads();

    // ---- each wave: n-tiles wave, wave+4, ... of 32 output channels ----
    const int st_off = (lane >> 3) * RPE_LDR + (lane & 7) * 4;
    const int frw = (lane & 31) * RPE_LDR + (lane >> 5) * 4;
    const int fra = (lane & 31) * ALD + (lane >> 5) * 4;
    for (int nt = wave; nt * 32 < C; nt += 4) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        // the filter chunk of step kc + 32 is in flight while chunk kc is multiplied (one load per trip made the loop a chain
        // of C / 32 dependent round trips per filter tile: 48 us for the 21 networks of a training step)
        f32x4 w[4];
        const float* wsrc = J.Wout + (size_t)(nt * 32 + (lane >> 3)) * C + (lane & 7) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = ld4(wsrc + (size_t)r * 8 * C);
        for (int kc = 0; kc < C; kc += 32) {
#pragma unroll
            for (int r = 0; r < 4; ++r) st4(Wst + r * 8 * RPE_LDR + st_off, w[r]);
            wave_lds_fence();
            if (kc + 32 < C) {
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = ld4(wsrc + (size_t)r * 8 * C + kc + 32);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a4 = ld4(As + fra + kc + g * 8);
                const f32x4 b4 = ld4(Wst + frw + g * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc, 0, 0, 0);
            }
            wave_lds_fence();
        }
        const int co = nt * 32 + (lane & 31);
        const float bo = J.bout[co];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < M) J.R[(size_t)m * C + co] = acc[r] + bo;
        }
    }
}

// ======================================================================================
// Spatial attention (flash style, fp32 MFMA 16x16x4)
// XCD-aware workgroup mapping of the spatial kernels.  Consecutive workgroup ids go round-robin to the 8 XCDs, each
// with its own L2.  All heads and query blocks of one frame read the same qkv rows (the heads interleave inside a
// row, two heads per 128-byte line), so the frame's workgroups are placed on ONE XCD: frame n lives on XCD n % 8.
// (With the natural (q block, head, frame) grid order the K/V lines of a frame-head were fetched by four XCDs:
// 50 MB of L2 fills per launch at 16x16 instead of the 10.5 MB the launch touches.)
// Flat grid of 8 * ceil(N / 8) * (heads * qblocks) workgroups; returns false for the padding ones.
__device__ __forceinline__ bool spatial_wg(int N, int heads, int qblocks, int& n, int& h, int& qb) {
    const int L = blockIdx.x;
    const int xcd = L & 7, r = L >> 3;
    const int per = heads * qblocks;
    const int slot = r / per, inner = r - slot * per;
    n = slot * 8 + xcd;
    h = inner / qblocks;
    qb = inner - h * qblocks;
    return n < N;
}
__host__ inline unsigned spatial_grid(int N, int heads, int qblocks) { return 8u * ((N + 7) / 8) * heads * qblocks; }

// ======================================================================================
// Workgroup = 4 waves = 64 queries of one (frame n, head h); key blocks of 64 are staged in LDS
// (K rows padded to F+8, V rows to F+4 floats: conflict-free b128 / b32 fragment reads).
// k-index mapping inside a 16-wide f group g: MFMA step e uses f = 16g + 4*kk + e (kk = lane>>4)
// on both K (A operand) and Q (B operand).
// FP = head dim padded to a multiple of 16 (compile time); F = real head dim (multiple of 4): the
// pad columns are zero in LDS / in the Q fragments and are never stored.
template <int FP, int KB>
__global__ __launch_bounds__(256) void attn_spatial_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                           float* __restrict__ lse, int N, int P, int C, int heads, int F) {
    constexpr int FG = FP / 16;   // 16-wide f groups
    constexpr int KT = KB / 16;   // 16-key tiles per staged key block (KB = 32 keeps FP = 128 under 64 KB of LDS)
    constexpr int KLD = FP + 8, VLD = FP + 4;
    __shared__ __attribute__((aligned(16))) float Ks[KB * KLD];
    __shared__ __attribute__((aligned(16))) float Vs[KB * VLD];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int n, h, qb;
    if (!spatial_wg(N, heads, (P + 63) / 64, n, h, qb)) return;      // workgroup-uniform
    const int q0 = qb * 64 + wave * 16;
    const int lq = lane & 15, kk = lane >> 4;
    const float scale = rsqrtf((float)F);
    const size_t ld = (size_t)3 * C;
    const float* base = qkv + (size_t)n * P * ld + h * F;

    // Q fragments (B operand), pre-scaled: lane (q = lq, kk) holds Q[q][16g + 4kk + e]
    f32x4 qf[FG];
    {
        const int q = q0 + lq;
#pragma unroll
        for (int g = 0; g < FG; ++g) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q < P && 16 * g + 4 * kk < F) v = ld4(base + (size_t)q * ld + 16 * g + 4 * kk);
            qf[g] = v * scale;
        }
    }
    f32x4 oacc[FG];
#pragma unroll
    for (int g = 0; g < FG; ++g) oacc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    for (int kb = 0; kb < P; kb += KB) {
        __syncthreads();  // previous block fully consumed
        {   // all loads of the block in flight before the first LDS write (the trip count is a compile-time constant)
            constexpr int NIT = (KB * (FP / 4) + 255) / 256;
            f32x4 kv[NIT], vv[NIT];
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int e = threadIdx.x + 256 * u;
                const int key = e / (FP / 4), fq = e - key * (FP / 4);
                kv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                vv[u] = kv[u];
                if (e < KB * (FP / 4) && kb + key < P && fq * 4 < F) {
                    const float* row = base + (size_t)(kb + key) * ld + fq * 4;
                    kv[u] = ld4(row + C);
                    vv[u] = ld4(row + 2 * C);
                }
            }
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int e = threadIdx.x + 256 * u;
                const int key = e / (FP / 4), fq = e - key * (FP / 4);
                if (e < KB * (FP / 4)) {
                    st4(Ks + key * KLD + fq * 4, kv[u]);
                    st4(Vs + key * VLD + fq * 4, vv[u]);
                }
            }
        }
        __syncthreads();

        // S^T tiles: rows = keys 16j + 4kk + r, column = query lq
        f32x4 s[KT];
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < FG; ++g) {
                const f32x4 k4 = ld4(Ks + (16 * j + lq) * KLD + 16 * g + 4 * kk);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[e], qf[g][e], acc, 0, 0, 0);
            }
            s[j] = acc;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (kb + 16 * j + 4 * kk + r >= P) s[j][r] = -INFINITY;
                mx = fmaxf(mx, s[j][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);  // m_run = -inf on the first block -> 0
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(s[j][r] - m_new);
                s[j][r] = pv;
                psum += pv;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int g = 0; g < FG; ++g) oacc[g] *= alpha;
        // O^T[f][q] += V^T[f][key] * P^T[key][q]; MFMA (j, r): k index kk <-> key 16j + 4kk + r
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vrow = Vs + (16 * j + 4 * kk + r) * VLD + lq;
#pragma unroll
                for (int g = 0; g < FG; ++g)
                    oacc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[16 * g], s[j][r], oacc[g], 0, 0, 0);
            }
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_run;
    const int q = q0 + lq;
    if (lse && q < P && kk == 0) lse[((size_t)n * heads + h) * P + q] = m_run + __logf(l_run);   // saved for the backward
    if (q < P) {
#pragma unroll
        for (int g = 0; g < FG; ++g)
            if (16 * g + 4 * kk < F) st4(o + ((size_t)n * P + q) * C + h * F + 16 * g + 4 * kk, oacc[g] * inv);
    }
}

// --------------------------------------------------------------------------------------
// Spatial attention backward (flash style, recomputes S from q, k and the saved log-sum-exp).
// One kernel body, two roles:
//   DKV = false: workgroup = 64 QUERIES held as MFMA B operands (q*scale, dO), key blocks streamed through LDS
//                -> dq = scale * sum_keys dS * K
//   DKV = true : workgroup = 64 KEYS held as B operands (k, v), query blocks (q*scale, dO, lse, delta) streamed
//                -> dk = sum_queries dS^T * (q*scale),  dv = sum_queries P^T * dO
// with P = exp(S - lse[query]), dP = dO . V^T, dS = P * (dP - delta[query]), delta = rowdot(O, dO).
// Tiles are T[y][x]: y = streamed token (row 16j + 4kk + r), x = stationary token (column lq), so the same
// LDS rows serve the row-wise b128 reads of the two tile products and the transposed scalar reads of the
// accumulations (row stride FP + 4: both patterns at most 2-way conflicted).
template <int FP, int KB, bool DKV>
__global__ __launch_bounds__(256) void attn_spatial_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dO,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               float* __restrict__ dqkv, int N, int P, int C, int heads, int F) {
    constexpr int FG = FP / 16, KT = KB / 16, LD = FP + 4;
    __shared__ __attribute__((aligned(16))) float A1s[KB * LD];
    __shared__ __attribute__((aligned(16))) float A2s[KB * LD];
    __shared__ float Ls[KB], Ds[KB];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int n, h, qb;
    if (!spatial_wg(N, heads, (P + 63) / 64, n, h, qb)) return;      // workgroup-uniform
    const int x = qb * 64 + wave * 16 + (lane & 15);
    const int lq = lane & 15, kk = lane >> 4;
    const float scale = rsqrtf((float)F);
    const size_t ld = (size_t)3 * C;
    const float* base = qkv + (size_t)n * P * ld + h * F;
    const float* dobase = dO + (size_t)n * P * C + h * F;
    const size_t sbase = ((size_t)n * heads + h) * P;
    const bool xin = x < P;

    f32x4 B1[FG], B2[FG], acc1[FG], acc2[FG];
#pragma unroll
    for (int g = 0; g < FG; ++g) {
        f32x4 v1 = {0.f, 0.f, 0.f, 0.f}, v2 = {0.f, 0.f, 0.f, 0.f};
        if (xin && 16 * g + 4 * kk < F) {
            if (DKV) {
                v1 = ld4(base + (size_t)x * ld + C + 16 * g + 4 * kk);
                v2 = ld4(base + (size_t)x * ld + 2 * C + 16 * g + 4 * kk);
            } else {
                v1 = ld4(base + (size_t)x * ld + 16 * g + 4 * kk) * scale;
                v2 = ld4(dobase + (size_t)x * C + 16 * g + 4 * kk);
            }
        }
        B1[g] = v1;
        B2[g] = v2;
        acc1[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc2[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float Lx = 0.f, Dx = 0.f;
    if (!DKV && xin) {
        Lx = lse[sbase + x];
        Dx = delta[sbase + x];
    }

    for (int yb = 0; yb < P; yb += KB) {
        __syncthreads();   // previous block fully consumed
        {   // all loads of the block in flight before the first LDS write
            constexpr int NIT = (KB * (FP / 4) + 255) / 256;
            f32x4 v1[NIT], v2[NIT];
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int e = threadIdx.x + 256 * u;
                const int y = e / (FP / 4), fq = e - y * (FP / 4);
                v1[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                v2[u] = v1[u];
                if (e < KB * (FP / 4) && yb + y < P && fq * 4 < F) {
                    const float* row = base + (size_t)(yb + y) * ld + fq * 4;
                    if (DKV) {
                        v1[u] = ld4(row) * scale;
                        v2[u] = ld4(dobase + (size_t)(yb + y) * C + fq * 4);
                    } else {
                        v1[u] = ld4(row + C);
                        v2[u] = ld4(row + 2 * C);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int e = threadIdx.x + 256 * u;
                const int y = e / (FP / 4), fq = e - y * (FP / 4);
                if (e < KB * (FP / 4)) {
                    st4(A1s + y * LD + fq * 4, v1[u]);
                    st4(A2s + y * LD + fq * 4, v2[u]);
                }
            }
        }
        if (DKV && threadIdx.x < KB) {
            const bool in = yb + (int)threadIdx.x < P;
            Ls[threadIdx.x] = in ? lse[sbase + yb + threadIdx.x] : 0.f;
            Ds[threadIdx.x] = in ? delta[sbase + yb + threadIdx.x] : 0.f;
        }
        __syncthreads();

        f32x4 pt[KT], dst[KT];
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < FG; ++g) {
                const f32x4 a1 = ld4(A1s + (16 * j + lq) * LD + 16 * g + 4 * kk);
                const f32x4 a2 = ld4(A2s + (16 * j + lq) * LD + 16 * g + 4 * kk);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], B1[g][e], t1, 0, 0, 0);
                    t2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[e], B2[g][e], t2, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int yl = 16 * j + 4 * kk + r;
                const float L = DKV ? Ls[yl] : Lx;
                const float D = DKV ? Ds[yl] : Dx;
                const float pv = (yb + yl < P) ? __expf(t1[r] - L) : 0.f;
                pt[j][r] = pv;
                dst[j][r] = pv * (t2[r] - D);
            }
        }
        // acc1^T[f][x] += A1^T[f][y] * dS[y][x];  (DKV) acc2^T[f][x] += A2^T[f][y] * P[y][x]
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (16 * j + 4 * kk + r) * LD + lq;
#pragma unroll
                for (int g = 0; g < FG; ++g) {
                    acc1[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1s[row + 16 * g], dst[j][r], acc1[g], 0, 0, 0);
                    if (DKV) acc2[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2s[row + 16 * g], pt[j][r], acc2[g], 0, 0, 0);
                }
            }
    }
    if (xin) {
        float* out = dqkv + ((size_t)n * P + x) * ld + h * F;
#pragma unroll
        for (int g = 0; g < FG; ++g)
            if (16 * g + 4 * kk < F) {
                if (DKV) {
                    st4(out + C + 16 * g + 4 * kk, acc1[g]);
                    st4(out + 2 * C + 16 * g + 4 * kk, acc2[g]);
                } else {
                    st4(out + 16 * g + 4 * kk, acc1[g] * scale);
                }
            }
    }
}

// delta[n][h][p] = sum_f O[n,p,h,f] * dO[n,p,h,f]
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ dO,
                                                         float* __restrict__ delta, long total, int P, int C, int heads, int F) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // ((n*P + p)*heads + h)
    if (i >= total) return;
    const int h = (int)(i % heads);
    const long tok = i / heads;
    const int p = (int)(tok % P);
    const long n = tok / P;
    const float* a = o + tok * C + h * F;
    const float* b = dO + tok * C + h * F;
    float acc = 0.f;
    for (int f = 0; f < F; f += 4) {
        const f32x4 u = ld4(a + f), v = ld4(b + f);
        acc += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
    }
    delta[((size_t)n * heads + h) * P + p] = acc;
}

// probabilities for logging (return_attn_weights): plain two-pass softmax, one wave per query row
__global__ __launch_bounds__(256) void attn_spatial_probs_kernel(const float* __restrict__ qkv, float* __restrict__ attn,
                                                                 int P, int C, int heads, int F) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // (h, q) of frame n
    const int n = blockIdx.y;
    if (row >= (long)heads * P) return;
    const int h = (int)(row / P), q = (int)(row % P);
    const float scale = rsqrtf((float)F);
    const size_t ld = (size_t)3 * C;
    const float* base = qkv + (size_t)n * P * ld + h * F;
    const float* qr = base + (size_t)q * ld;
    float* out = attn + (((size_t)n * heads + h) * P + q) * P;
    float mx = -INFINITY;
    for (int k = lane; k < P; k += 64) {
        const float* kr = base + (size_t)k * ld + C;
        float d = 0.f;
        for (int f = 0; f < F; ++f) d += qr[f] * scale * kr[f];
        out[k] = d;
        mx = fmaxf(mx, d);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < P; k += 64) {
        const float e = __expf(out[k] - mx);
        out[k] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int k = lane; k < P; k += 64) out[k] *= inv;
}

// ======================================================================================
// Temporal attention with RPE
// ======================================================================================
// Workgroup = one (b, head) and a strip of 4*PPW pixels; wave = PPW pixels x T query frames, lane = j*T + t.
// A lane owns ALL T logits of its (pixel, query frame) in registers, so the softmax needs no cross-lane
// step.  The head dim is walked in chunks of FC channels: per chunk the R_k / R_q slices of this (b, head)
// - [T][T][FC], shared by every pixel - are staged ONCE per workgroup in LDS (R_q transposed to [t][s]), the
// k (then v) rows of the wave's pixels wave-privately; q stays in registers.  Rows are padded by 4 floats:
// lanes of consecutive t read b128 words 4 banks apart (conflict-free), lanes of one pixel share k/v words
// (LDS broadcast).  TMAX bounds the unrolled key loop (T <= TMAX).
constexpr int TA_MAXT = 32;

// The kernel is a short pipeline of phases: NC logit chunks, then NC PV chunks.  The global loads of phase
// i+1 are issued into registers BEFORE phase i computes (the loop over chunks is latency-bound otherwise:
// every staging pass costs a full L2 round trip), and committed to LDS after it.
template <int TMAX, int FC>
struct TAStage {
    static constexpr int NQ = FC / 4;
    static constexpr int RB = (TMAX * TMAX * NQ + 255) / 256;   // R float4 per thread per slice
    f32x4 ra[RB], rb[RB], kv[NQ], q[NQ];
};

template <int TMAX, int FC>
__global__ __launch_bounds__(256)
void attn_temporal_kernel(const float* __restrict__ qkv, const float* __restrict__ Rq, const float* __restrict__ Rk,
                          const float* __restrict__ Rv, const float* __restrict__ mask, float* __restrict__ o,
                          float* __restrict__ attn_out, int T, int P, int C, int heads, int PPW,
                          RSel rsel) {
    using ST = TAStage<TMAX, FC>;
    constexpr int NQ = ST::NQ, RB = ST::RB;
    extern __shared__ __attribute__((aligned(16))) float ta_smem[];
    const int RST = T * FC + 4;                 // row stride of the R slices ([t] rows) and of a pixel's k/v rows
    float* Rk_s = ta_smem;                      // [T][RST]   (PV phases: R_v)
    float* Rq_s = Rk_s + T * RST;               // [T][RST]   R_q transposed: row t holds R_q[s][t] for all s
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* kv_s = Rq_s + T * RST + wave * PPW * RST;   // [PPW][RST] wave-private
    const int b = blockIdx.z, h = blockIdx.y;
    const int F = C / heads;
    const int NC = F / FC;
    const float invT = 1.0f / (float)T;
    const int j = (int)(((float)lane + 0.5f) * invT), t = lane - j * T;
    const int p0 = (blockIdx.x * 4 + wave) * PPW;      // first pixel of this wave
    const int p = p0 + j;
    const bool active = j < PPW && p < P;
    const float scale = rsqrtf((float)F);
    const size_t ld = (size_t)3 * C;
    const float* qrow = qkv + ((size_t)(b * T + t) * P + p) * ld + h * F;   // dereferenced only if active
    // rsel: the R tensors are tables over the sampler's timesteps ([n_t][B][T][T][C]); rsel[b] picks the slice
    const size_t rb = rsel.slice(b, (int)gridDim.z);
    const float* Rbase[3] = {Rk + rb * T * T * C + h * F, Rq + rb * T * T * C + h * F, Rv + rb * T * T * C + h * F};

    // per-thread staging slots: global offsets (without the chunk offset) and LDS offsets, computed once.
    // Slots past the end load from offset 0 (always valid) and are simply not committed: issue() is
    // branch-free, all loads of a phase go out back to back.
    int r_g[RB], r_la[RB], r_lb[RB];
    unsigned r_ok = 0, k_ok = 0, k_keep = 0;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int e = tid + 256 * i;
        const int u = e % NQ, ts = e / NQ;
        const int a = (int)(((float)ts + 0.5f) * invT), c = ts - a * T;
        const bool ok = e < T * T * NQ;
        r_ok |= ok ? (1u << i) : 0u;
        r_g[i] = ok ? ts * C + 4 * u : 0;
        r_la[i] = a * RST + c * FC + 4 * u;
        r_lb[i] = c * RST + a * FC + 4 * u;
    }
    int k_g[NQ], k_l[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int e = lane + 64 * i;
        const int u = e % NQ, js = e / NQ;
        const int jj = (int)(((float)js + 0.5f) * invT), ss = js - jj * T;
        const bool ok = e < PPW * T * NQ, inside = ok && p0 + jj < P;
        k_ok |= ok ? (1u << i) : 0u;
        k_keep |= inside ? (1u << i) : 0u;       // pixels past the end are staged as zeros
        k_g[i] = inside ? (int)(((size_t)(b * T + ss) * P + p0 + jj) * ld) + h * F + 4 * u : 0;
        k_l[i] = jj * RST + ss * FC + 4 * u;
    }
    const float* qsafe = active ? qrow : qkv;

    ST st;
    auto issue = [&](int ph) {   // ph is wave-uniform
        const bool lg = ph < NC;
        const int f0 = (lg ? ph : ph - NC) * FC;
        const float* A = (lg ? Rbase[0] : Rbase[2]) + f0;
        const float* Bq = Rbase[1] + (lg ? f0 : 0);
        const float* KV = qkv + (lg ? C : 2 * C) + f0;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            st.ra[i] = ld4(A + r_g[i]);
            st.rb[i] = ld4(Bq + r_g[i]);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) st.kv[i] = ld4(KV + k_g[i]);
#pragma unroll
        for (int u = 0; u < NQ; ++u) st.q[u] = ld4(qsafe + (lg ? f0 : 0) + 4 * u);
    };
    auto commit = [&](int ph) {
        const bool lg = ph < NC;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (r_ok & (1u << i)) {
                st4(Rk_s + r_la[i], st.ra[i]);
                if (lg) st4(Rq_s + r_lb[i], st.rb[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            if (k_ok & (1u << i)) st4(kv_s + k_l[i], (k_keep & (1u << i)) ? st.kv[i] : (f32x4){0.f, 0.f, 0.f, 0.f});
    };

    float logit[TMAX];
#pragma unroll
    for (int s = 0; s < TMAX; ++s) logit[s] = 0.f;

    issue(0);
    for (int ph = 0; ph < NC; ++ph) {
        __syncthreads();           // previous phase fully consumed
        commit(ph);
        f32x4 q4[NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) q4[u] = st.q[u] * scale;
        __syncthreads();
        issue(ph + 1);             // next logit chunk, or the first PV chunk
#ifdef TA_DEBUG
        if (active && !(TA_DEBUG & 1)) {
#else
        if (active) {
#endif
            const float* kr = kv_s + j * RST;
            const float* rkr = Rk_s + t * RST;
            const float* rqr = Rq_s + t * RST;
#pragma unroll
            for (int s = 0; s < TMAX; ++s) {
                if (s < T) {
                    float a0 = 0.f, a1 = 0.f;
#pragma unroll
                    for (int u = 0; u < NQ; ++u) {
                        const f32x4 k4 = ld4(kr + s * FC + 4 * u);
                        const f32x4 rk4 = ld4(rkr + s * FC + 4 * u);
                        const f32x4 rq4 = ld4(rqr + s * FC + 4 * u);
                        a0 += q4[u].x * (k4.x + rk4.x) + q4[u].y * (k4.y + rk4.y) + q4[u].z * (k4.z + rk4.z) + q4[u].w * (k4.w + rk4.w);
                        a1 += k4.x * rq4.x + k4.y * rq4.y + k4.z * rq4.z + k4.w * rq4.w;
                    }
                    logit[s] += a0 + a1 * scale;
                }
            }
        }
    }

    // two-clique mask + softmax, all in this lane's registers
    if (active) {
        const float mt = mask ? mask[b * T + t] : 1.f;
        float mx = -INFINITY;
#pragma unroll
        for (int s = 0; s < TMAX; ++s) {
            float v = -INFINITY;
            if (s < T) {
                v = logit[s];
                if (mask) {
                    const float ms = mask[b * T + s];
                    const float pen = 1.f - (mt * ms + (1.f - mt) * (1.f - ms));
                    v -= (pen == 1.f) ? INFINITY : pen;
                }
            }
            logit[s] = v;
            mx = fmaxf(mx, v);
        }
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < TMAX; ++s) {
            const float e = (logit[s] == -INFINITY) ? 0.f : __expf(logit[s] - mx);
            logit[s] = e;
            sum += e;
        }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int s = 0; s < TMAX; ++s) logit[s] *= inv;
        if (attn_out) {
            float* ar = attn_out + ((((size_t)b * P + p) * heads + h) * T + t) * T;
#pragma unroll
            for (int s = 0; s < TMAX; ++s)
                if (s < T) ar[s] = logit[s];
        }
    }

    // o[t][f] = sum_s p[t][s] * (v[s][f] + R_v[t][s][f]), chunk by chunk
    for (int ph = NC; ph < 2 * NC; ++ph) {
        __syncthreads();
        commit(ph);
        __syncthreads();
        if (ph + 1 < 2 * NC) issue(ph + 1);
#ifdef TA_DEBUG
        if (active && !(TA_DEBUG & 2)) {
#else
        if (active) {
#endif
            const float* vr = kv_s + j * RST;
            const float* rvr = Rk_s + t * RST;
            f32x4 acc[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < TMAX; ++s) {
                if (s < T) {
                    // opaque copy: stops the compiler from hoisting 24 broadcast register pairs (for packed
                    // FMAs) out of the chunk loop and spilling them to scratch
                    float pr = logit[s];
                    asm volatile("" : "+v"(pr));
#pragma unroll
                    for (int u = 0; u < NQ; ++u) acc[u] += pr * (ld4(vr + s * FC + 4 * u) + ld4(rvr + s * FC + 4 * u));
                }
            }
            float* orow = o + ((size_t)(b * T + t) * P + p) * C + h * F + (ph - NC) * FC;
#pragma unroll
            for (int u = 0; u < NQ; ++u) st4(orow + 4 * u, acc[u]);
        }
    }
}

template <int TMAX, int FC>
int launch_temporal(const float* qkv, const float* Rq, const float* Rk, const float* Rv, const float* mask, float* o,
                    float* attn_out, int B, int T, int P, int C, int heads, RSel rsel, hipStream_t s) {
    const int PPW = 64 / T;                                  // pixels per wave
    const int RST = T * FC + 4;
    const size_t lds = (size_t)(2 * T + 4 * PPW) * RST * sizeof(float);
    if (lds > 160 * 1024) return LFVDM_E_UNSUPPORTED;
    static DynLdsLimit limit;       // raised (per device) when a larger T needs it
    if (int rc = limit.ensure(reinterpret_cast<const void*>(&attn_temporal_kernel<TMAX, FC>), lds)) return rc;
    const dim3 grid((unsigned)((P + 4 * PPW - 1) / (4 * PPW)), (unsigned)heads, (unsigned)B);
    hipLaunchKernelGGL((attn_temporal_kernel<TMAX, FC>), grid, dim3(256), lds, s, qkv, Rq, Rk, Rv, mask, o, attn_out, T, P, C,
                       heads, PPW, rsel);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

template <int FC>
int launch_temporal_t(const float* qkv, const float* Rq, const float* Rk, const float* Rv, const float* mask, float* o,
                      float* attn_out, int B, int T, int P, int C, int heads, RSel rsel, hipStream_t s) {
    if (T <= 8) return launch_temporal<8, FC>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    if (T <= 16) return launch_temporal<16, FC>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    if (T <= 24) return launch_temporal<24, FC>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    return launch_temporal<32, FC>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
}

}  // namespace

extern "C" int lfvdm_rpe_nets_maxc(const lfvdm_rpe_job* jobs_dev, int njobs, int total_tiles, const int64_t* fi, int B, int T,
                                   int max_channels, void* stream) {
    if (njobs <= 0 || total_tiles <= 0 || B <= 0 || T <= 0 || max_channels <= 0 || max_channels > 512) return LFVDM_E_SHAPE;
    // LDS for the widest network of the launch: A tile 32*(C+4) + 4 wave-private W chunks.  (Sized for 512 channels the
    // 84 KB allow one workgroup per CU; at the 64..128 channels of the latent models four fit, which is what hides the
    // latency of the row generation: 19 -> 6 ms for the 21 networks x 1000 timesteps of a sampler's R tables.)
    const int maxC = (max_channels + 31) / 32 * 32;
    const size_t lds = (size_t)(32 * (maxC + 4) + 4 * 32 * RPE_LDR) * sizeof(float);
    static DynLdsLimit limit;
    if (int rc = limit.ensure(reinterpret_cast<const void*>(&rpe_nets_kernel), (size_t)(32 * (512 + 4) + 4 * 32 * RPE_LDR) * sizeof(float)))
        return rc;
    hipLaunchKernelGGL(rpe_nets_kernel, dim3(total_tiles), dim3(256), lds, (hipStream_t)stream, jobs_dev, njobs, fi, B, T);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_rpe_nets(const lfvdm_rpe_job* jobs_dev, int njobs, int total_tiles, const int64_t* fi, int B, int T,
                              void* stream) {
    return lfvdm_rpe_nets_maxc(jobs_dev, njobs, total_tiles, fi, B, T, 512, stream);
}

extern "C" int lfvdm_attn_spatial(const float* qkv, float* o, float* attn_out, float* lse_out, int N, int P, int C, int heads,
                                  void* stream) {
    if (N <= 0 || P <= 0 || heads <= 0 || C % heads) return LFVDM_E_SHAPE;
    const int F = C / heads;
    const dim3 grid(spatial_grid(N, heads, (P + 63) / 64));
    hipStream_t s = (hipStream_t)stream;
    if (F % 4 || F > 128) return LFVDM_E_UNSUPPORTED;
    const int FP = (F + 15) / 16 * 16;
#define LFVDM_SPATIAL_FWD(FPV, KBV) \
    hipLaunchKernelGGL((attn_spatial_kernel<FPV, KBV>), grid, dim3(256), 0, s, qkv, o, lse_out, N, P, C, heads, F)
    switch (FP) {
        case 16: LFVDM_SPATIAL_FWD(16, 64); break;
        case 32: LFVDM_SPATIAL_FWD(32, 64); break;
        case 48: LFVDM_SPATIAL_FWD(48, 64); break;
        case 64: LFVDM_SPATIAL_FWD(64, 64); break;
        case 80: LFVDM_SPATIAL_FWD(80, 64); break;
        case 96: LFVDM_SPATIAL_FWD(96, 64); break;
        case 112: LFVDM_SPATIAL_FWD(112, 32); break;
        case 128: LFVDM_SPATIAL_FWD(128, 32); break;
        default: return LFVDM_E_UNSUPPORTED;
    }
#undef LFVDM_SPATIAL_FWD
    LFVDM_CHECK_LAUNCH();
    if (attn_out) {
        hipLaunchKernelGGL(attn_spatial_probs_kernel, dim3((heads * P + 3) / 4, N), dim3(256), 0, s, qkv, attn_out, P, C, heads, F);
        LFVDM_CHECK_LAUNCH();
    }
    return LFVDM_OK;
}

template <int FP, int KB>
static void launch_spatial_bwd(const float* qkv, const float* dO, const float* lse, const float* delta, float* dqkv, int N, int P,
                               int C, int heads, int F, hipStream_t s) {
    const dim3 grid(spatial_grid(N, heads, (P + 63) / 64));
    hipLaunchKernelGGL((attn_spatial_bwd_kernel<FP, KB, false>), grid, dim3(256), 0, s, qkv, dO, lse, delta, dqkv, N, P, C, heads, F);
    hipLaunchKernelGGL((attn_spatial_bwd_kernel<FP, KB, true>), grid, dim3(256), 0, s, qkv, dO, lse, delta, dqkv, N, P, C, heads, F);
}

extern "C" int lfvdm_attn_spatial_bwd(const float* qkv, const float* o, const float* d_o, const float* lse, float* delta_ws,
                                      float* dqkv, int N, int P, int C, int heads, void* stream) {
    if (N <= 0 || P <= 0 || heads <= 0 || C % heads) return LFVDM_E_SHAPE;
    if (!qkv || !o || !d_o || !lse || !delta_ws || !dqkv) return LFVDM_E_SHAPE;
    const int F = C / heads;
    if (F % 4 || F > 128) return LFVDM_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)N * P * heads;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, o, d_o, delta_ws, total, P, C,
                       heads, F);
    LFVDM_CHECK_LAUNCH();
    const int FP = (F + 15) / 16 * 16;
    switch (FP) {
        case 16: launch_spatial_bwd<16, 64>(qkv, d_o, lse, delta_ws, dqkv, N, P, C, heads, F, s); break;
        case 32: launch_spatial_bwd<32, 64>(qkv, d_o, lse, delta_ws, dqkv, N, P, C, heads, F, s); break;
        case 48: launch_spatial_bwd<48, 64>(qkv, d_o, lse, delta_ws, dqkv, N, P, C, heads, F, s); break;
        case 64: launch_spatial_bwd<64, 64>(qkv, d_o, lse, delta_ws, dqkv, N, P, C, heads, F, s); break;
        case 80: launch_spatial_bwd<80, 32>(qkv, d_o, lse, delta_ws, dqkv, N, P, C, heads, F, s); break;
        case 96: launch_spatial_bwd<96, 32>(qkv, d_o, lse, delta_ws, dqkv, N, P, C, heads, F, s); break;
        case 112: launch_spatial_bwd<112, 32>(qkv, d_o, lse, delta_ws, dqkv, N, P, C, heads, F, s); break;
        case 128: launch_spatial_bwd<128, 32>(qkv, d_o, lse, delta_ws, dqkv, N, P, C, heads, F, s); break;
        default: return LFVDM_E_UNSUPPORTED;
    }
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

int lfvdm_attn_temporal2_try(const float* qkv, const float* Rq, const float* Rk, const float* Rv, const float* mask, float* o,
                             float* attn_out, int B, int T, int P, int C, int heads, RSel rsel, hipStream_t s);

extern "C" int lfvdm_attn_temporal_ring(const float* qkv, const float* Rq, const float* Rk, const float* Rv, const float* mask,
                                        float* o, float* attn_out, int B, int T, int P, int C, int heads, const int64_t* rsel_p,
                                        int ring, void* stream) {
    if (B <= 0 || T <= 0 || T > TA_MAXT || P <= 0 || heads <= 0 || C % heads) return LFVDM_E_SHAPE;
    if (!Rq || !Rk || !Rv || ring < 0 || (ring > 0 && !rsel_p)) return LFVDM_E_SHAPE;
    const RSel rsel = {rsel_p, ring};
    const int F = C / heads;
    hipStream_t s = (hipStream_t)stream;
    // second-generation kernel (attention_temporal2.hip) for head dims 16 / 32 / 64 and launches that do not fill the chip
    {
        const int rc = lfvdm_attn_temporal2_try(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
        if (rc != LFVDM_E_UNSUPPORTED) return rc;
    }
    // head dim 32 (the 128-channel levels): the whole head in ONE chunk - two staging phases instead of four.  The R
    // slices of a chunk are then 2 x T x (32T + 4) floats: fits the 160 KiB of LDS up to T = 24 (launch_temporal checks)
    static const bool no_fc32 = getenv("LFVDM_ATTN_NO_FC32") != nullptr;      // A/B aid
    if (F == 32 && T <= 24 && !no_fc32) {
        const int rc = launch_temporal_t<32>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
        if (rc != LFVDM_E_UNSUPPORTED) return rc;
    }
    if (F % 16 == 0 && T <= 24) return launch_temporal_t<16>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    if (F % 8 == 0) return launch_temporal_t<8>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    return LFVDM_E_UNSUPPORTED;
}

extern "C" int lfvdm_attn_temporal_sel(const float* qkv, const float* Rq, const float* Rk, const float* Rv, const float* mask,
                                       float* o, float* attn_out, int B, int T, int P, int C, int heads, const int64_t* rsel,
                                       void* stream) {
    return lfvdm_attn_temporal_ring(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, 0, stream);
}

extern "C" int lfvdm_attn_temporal(const float* qkv, const float* Rq, const float* Rk, const float* Rv, const float* mask,
                                   float* o, float* attn_out, int B, int T, int P, int C, int heads, void* stream) {
    return lfvdm_attn_temporal_sel(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, nullptr, stream);
}
