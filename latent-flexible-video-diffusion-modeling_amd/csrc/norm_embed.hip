// GroupNorm statistics, the fused input compositing + first conv, and the small-M grouped
// linear ("row-dot") used for every embedding projection.  All HBM/L2-bound; wave = 64.
#include "common_hip.h"
#include "gn_wave_body.h"

namespace {

// -------------------------------------------------------------------------------------
// gn_coef: one workgroup per (sample n, slice of 8 groups).  Two exact passes (mean, then
// centred variance) like ATen's CPU GroupNorm; small slices (<= 8 float4 per thread, i.e. every level of the
// 16x16 latent configs) stay in registers between the passes, larger ones are re-read from L2 for pass 2.
// Thread (pl, q): pixel lane pl strides over the P positions, q = float4 channel quad.
// -------------------------------------------------------------------------------------
constexpr int GN_GPW = 8;        // groups per workgroup
constexpr int GN_THREADS = 256;
constexpr int GN_MAXCW = 256;    // channels per workgroup: supports C <= 1024


template <int KEEP>      // float4 per thread held in registers between the passes: 8, or 16 for slices of up to 16 pixel lanes' worth
__global__ __launch_bounds__(GN_THREADS) void gn_coef_kernel(
    const float* __restrict__ s0, const float* __restrict__ s1, int C0, int C1, int P,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ film, int film_div,
    int film_ld, float eps, float* __restrict__ coefA, float* __restrict__ coefB, float* __restrict__ stats,
    float* __restrict__ act_out, int act_mode) {
    const int C = C0 + C1;
    const int cg = C / 32;
    const float rcg = __builtin_amdgcn_rcpf((float)cg);
    const int CW = GN_GPW * cg;       // channels handled here
    const int Q = CW / 4;             // float4 quads
    const int n = blockIdx.x;
    const int cbase = blockIdx.y * CW;
    const int PL = GN_THREADS / Q;    // pixel lanes
    const int tid = threadIdx.x;
    const bool active = tid < PL * Q;
    const int q = active ? tid % Q : 0;
    const int pl = active ? tid / Q : 0;
    const int c = cbase + q * 4;

    __shared__ float part[GN_THREADS * 4];  // [pl][CW] partial channel sums
    __shared__ float chs[GN_MAXCW];
    __shared__ float gmean[GN_GPW], grstd[GN_GPW];
    __shared__ __attribute__((aligned(16))) float shA[GN_MAXCW], shB[GN_MAXCW];

    const size_t pos0 = (size_t)n * P;
    const bool pow2q = (Q & (Q - 1)) == 0 && Q <= 64;      // workgroup-uniform
    const bool cached = P <= KEEP * PL;       // workgroup-uniform
    f32x4 keep[KEEP];
    for (int pass = 0; pass < 2; ++pass) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (active) {
            f32x4 mu = {0.f, 0.f, 0.f, 0.f};
            if (pass) {
                mu.x = gmean[fdiv_small(q * 4 + 0, cg, rcg)]; mu.y = gmean[fdiv_small(q * 4 + 1, cg, rcg)];
                mu.z = gmean[fdiv_small(q * 4 + 2, cg, rcg)]; mu.w = gmean[fdiv_small(q * 4 + 3, cg, rcg)];
            }
            if (cached) {
#pragma unroll
                for (int i = 0; i < KEEP; ++i) {
                    const int p = pl + i * PL;
                    if (pass == 0) {
                        keep[i] = (p < P) ? ld_cat(s0, s1, C0, C1, pos0 + p, c) : (f32x4){0.f, 0.f, 0.f, 0.f};
                        s += keep[i];
                    } else if (p < P) {
                        const f32x4 v = keep[i] - mu;
                        s += v * v;
                    }
                }
            } else {
                for (int p = pl; p < P; p += PL) {
                    f32x4 v = ld_cat(s0, s1, C0, C1, pos0 + p, c);
                    if (pass) { v = v - mu; s += v * v; } else { s += v; }
                }
            }
            if (pow2q) {
                // quads of a wave repeat every Q lanes: butterfly over the pixel lanes of the wave, then one partial
                // row per wave (a serial sum over all PL pixel lanes - 64 dependent LDS reads at C = 64 - was half of
                // this kernel's time)
                for (int off = Q; off < 64; off <<= 1) {
                    s.x += __shfl_xor(s.x, off, 64); s.y += __shfl_xor(s.y, off, 64);
                    s.z += __shfl_xor(s.z, off, 64); s.w += __shfl_xor(s.w, off, 64);
                }
                if ((tid & 63) < Q) st4(part + ((tid >> 6) * Q + q) * 4, s);
            } else {
                st4(part + (pl * Q + q) * 4, s);
            }
        }
        __syncthreads();
        const int rows = pow2q ? GN_THREADS / 64 : PL;
        for (int cc = tid; cc < CW; cc += GN_THREADS) {
            float t = 0.f;
            for (int i = 0; i < rows; ++i) t += part[i * CW + cc];
            chs[cc] = t;
        }
        __syncthreads();
        if (tid < GN_GPW) {
            float t = 0.f;
            for (int i = 0; i < cg; ++i) t += chs[tid * cg + i];
            const float inv = 1.0f / (float)(cg * P);
            if (pass == 0) gmean[tid] = t * inv;
            else grstd[tid] = 1.0f / sqrtf(t * inv + eps);
        }
        __syncthreads();
    }
    if (stats && tid < GN_GPW) {   // (mean, rstd) per (sample, group) for the backward pass
        stats[((size_t)n * 32 + blockIdx.y * GN_GPW + tid) * 2 + 0] = gmean[tid];
        stats[((size_t)n * 32 + blockIdx.y * GN_GPW + tid) * 2 + 1] = grstd[tid];
    }
    for (int cc = tid; cc < CW; cc += GN_THREADS) {
        const int ch = cbase + cc;
        const int g = fdiv_small(cc, cg, rcg);
        float A = grstd[g] * gamma[ch];
        float B = beta[ch] - gmean[g] * A;
        if (film) {
            const float* f = film + (size_t)(n / film_div) * film_ld;
            const float sc = 1.0f + f[ch];
            A *= sc;
            B = B * sc + f[C + ch];
        }
        if (coefA) {
            coefA[(size_t)n * C + ch] = A;
            coefB[(size_t)n * C + ch] = B;
        }
        shA[cc] = A;
        shB[cc] = B;
    }
    if (act_out == nullptr) return;
    // materialise act(x*A + B) as ONE contiguous [N*P][C] tensor (the virtual concat becomes a real one): on gfx950
    // the fp32 MFMAs run on the vector ALUs, so a prologue fused into the consuming implicit GEMM - re-evaluated for
    // every tap and every filter tile - sits on the GEMM's critical path; evaluating it once here is cheaper.
    __syncthreads();
    if (active) {
        const f32x4 a4 = ld4(shA + q * 4), b4 = ld4(shB + q * 4);
        if (cached) {
#pragma unroll
            for (int i = 0; i < KEEP; ++i) {
                const int p = pl + i * PL;
                if (p < P) {
                    f32x4 v = keep[i] * a4 + b4;
                    if (act_mode == LFVDM_ACT_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
                    st4(act_out + (pos0 + p) * C + c, v);
                }
            }
        } else {
            for (int p = pl; p < P; p += PL) {
                f32x4 v = ld_cat(s0, s1, C0, C1, pos0 + p, c) * a4 + b4;
                if (act_mode == LFVDM_ACT_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
                st4(act_out + (pos0 + p) * C + c, v);
            }
        }
    }
}

// -------------------------------------------------------------------------------------
// gn_wave: the same GroupNorm for maps of <= 256 positions with 2 / 4 / 8 / 16 channels per group (C = 64 ... 512), ONE
// WAVE per (sample, 16 channels): lane = (pixel lane pl = lane >> 2, channel quad q = lane & 3), the slice
// [P][16 channels] sits in <= 16 float4 per lane, and both reductions are wave butterflies (over the 16 pixel lanes,
// then over the 1 / 2 / 4 quad lanes of a group) - no LDS, no barrier.  The workgroup kernel above spends seven
// barriers and four LDS round trips on 16 KB of data: 5.8 us per launch against ~4 here, 15 launches per denoising step.
// -------------------------------------------------------------------------------------
template <int CG>      // channels per group: 2, 4, 8, 16
__global__ __launch_bounds__(64) void gn_wave_kernel(
    const float* __restrict__ s0, const float* __restrict__ s1, int C0, int C1, int P,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ film, int film_div,
    int film_ld, float eps, float* __restrict__ coefA, float* __restrict__ coefB, float* __restrict__ stats,
    float* __restrict__ act_out, int act_mode, int ldo) {
    gn_wave_body<CG, false>(blockIdx.x, blockIdx.y, threadIdx.x, s0, s1, C0, C1, P, gamma, beta, film, film_div, film_ld, eps, coefA,
                            coefB, stats, act_out, act_mode, ldo);
}

// the same unit shared by the four waves of a workgroup (gn_wave_body, NW = 4): maps of >= 128 positions
template <int CG>
__global__ __launch_bounds__(256) void gn_wave4_kernel(
    const float* __restrict__ s0, const float* __restrict__ s1, int C0, int C1, int P,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ film, int film_div,
    int film_ld, float eps, float* __restrict__ coefA, float* __restrict__ coefB, float* __restrict__ stats,
    float* __restrict__ act_out, int act_mode, int ldo) {
    gn_wave_body<CG, false, 4>(blockIdx.x, blockIdx.y, threadIdx.x & 63, s0, s1, C0, C1, P, gamma, beta, film, film_div, film_ld, eps,
                               coefA, coefB, stats, act_out, act_mode, ldo, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
}

// one wave per (sample, 16 channels) when the map and the group width allow it; false = use the workgroup kernels
inline bool gn_wave_launch(const float* src0, const float* src1, int C0, int C1, int N, int P, const float* gamma,
                           const float* beta, const float* film, int film_div, int film_ld, float eps, float* coefA,
                           float* coefB, float* stats, float* out, int act, hipStream_t s) {
    static const bool off = getenv("LFVDM_GN_NO_WAVE") != nullptr;          // A/B aid
    const int C = C0 + C1, cg = C / 32;
    if (off || P > 256 || C0 % 16 || (cg != 2 && cg != 4 && cg != 8 && cg != 16) || N > 65535) return false;
    const dim3 grid(N, C / 16);
    // four waves per unit on the larger maps (LFVDM_GN_WAVE4_MINP: smallest map that takes it, default 128; 0 = never).
    // Measured at cfg B (12 launches per denoising step, 16x16 maps): 70-74 -> 60-63 us per step, 1157-1160 -> 1170-1175
    // steps/s; eight waves per unit (512 threads, 2 float4 per lane): equal to four, not kept
    static const int minp4 = getenv("LFVDM_GN_WAVE4_MINP") ? atoi(getenv("LFVDM_GN_WAVE4_MINP")) : 128;
    if (minp4 > 0 && P >= minp4 && P % 64 == 0) {
#define LFVDM_GNW4(G)                                                                                                   \
    hipLaunchKernelGGL(gn_wave4_kernel<G>, grid, dim3(256), 0, s, src0, src1, C0, C1, P, gamma, beta, film, film_div, film_ld, \
                       eps, coefA, coefB, stats, out, act, 0)
        if (cg == 2) LFVDM_GNW4(2);
        else if (cg == 4) LFVDM_GNW4(4);
        else if (cg == 8) LFVDM_GNW4(8);
        else LFVDM_GNW4(16);
#undef LFVDM_GNW4
        return true;
    }
#define LFVDM_GNW(G)                                                                                                   \
    hipLaunchKernelGGL(gn_wave_kernel<G>, grid, dim3(64), 0, s, src0, src1, C0, C1, P, gamma, beta, film, film_div, film_ld, \
                       eps, coefA, coefB, stats, out, act, 0)
    if (cg == 2) LFVDM_GNW(2);
    else if (cg == 4) LFVDM_GNW(4);
    else if (cg == 8) LFVDM_GNW(8);
    else LFVDM_GNW(16);
#undef LFVDM_GNW
    return true;
}

// -------------------------------------------------------------------------------------
// Large maps (P > 8 pixel lanes' worth: pixel space, 32x32 latents): the (sample, 8 groups) decomposition above gives
// N*4 workgroups that each walk their whole slice three times, one load at a time - 80 workgroups streaming 2 MB each at
// 128x128 (1-2 ms per GroupNorm, a third of the pixel-space step).  Here a workgroup owns a CHUNK of 8*PL positions of
// the slice, held in registers:
//   pass 1 (gn_chunk_stats_kernel): exact two-pass (mean, M2) of the chunk per group -> part[n][g][s] = (mean_s, M2_s)
//   pass 2 (gn_chunk_apply_kernel): every workgroup combines the S partials of its 8 groups in a fixed order
//     mean = sum n_s mean_s / n,  M2 = sum (M2_s + n_s (mean_s - mean)^2)      (Chan et al.; as exact as the two-pass)
//   and applies the affine (+FiLM)(+activation) to its chunk, whose loads were issued before the combination.
// Deterministic; x is read twice and written once by thousands of workgroups.
// -------------------------------------------------------------------------------------
constexpr int GN_KEEP = 8;

struct GnChunkGeom { int C0, C1, P, S, PL; };

__device__ __forceinline__ void gn_chunk_load(const float* s0, const float* s1, const GnChunkGeom& g, size_t pos0, int p_first,
                                              int pl, int c, f32x4 (&keep)[GN_KEEP]) {
#pragma unroll
    for (int i = 0; i < GN_KEEP; ++i) {
        const int p = p_first + pl + i * g.PL;
        keep[i] = ld_cat(s0, s1, g.C0, g.C1, pos0 + min(p, g.P - 1), c);
    }
}

__global__ __launch_bounds__(GN_THREADS) void gn_chunk_stats_kernel(const float* __restrict__ s0, const float* __restrict__ s1,
                                                                    GnChunkGeom g, float* __restrict__ part_out) {
    const int C = g.C0 + g.C1, cg = C / 32, CW = GN_GPW * cg, Q = CW / 4, PL = g.PL;
    const float rcg = __builtin_amdgcn_rcpf((float)cg);
    const int sidx = blockIdx.x, n = blockIdx.z, cbase = blockIdx.y * CW;
    const int tid = threadIdx.x;
    const bool active = tid < PL * Q;
    const int q = active ? tid % Q : 0, pl = active ? tid / Q : 0;
    const int c = cbase + q * 4;
    const int p_first = sidx * (GN_KEEP * PL);
    const int npos = min(GN_KEEP * PL, g.P - p_first);
    __shared__ float part[GN_THREADS * 4];
    __shared__ float chs[GN_MAXCW];
    __shared__ float gmean[GN_GPW];
    f32x4 keep[GN_KEEP];
    gn_chunk_load(s0, s1, g, (size_t)n * g.P, p_first, pl, c, keep);
    for (int pass = 0; pass < 2; ++pass) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (active) {
            f32x4 mu = {0.f, 0.f, 0.f, 0.f};
            if (pass) {
                mu.x = gmean[fdiv_small(q * 4 + 0, cg, rcg)]; mu.y = gmean[fdiv_small(q * 4 + 1, cg, rcg)];
                mu.z = gmean[fdiv_small(q * 4 + 2, cg, rcg)]; mu.w = gmean[fdiv_small(q * 4 + 3, cg, rcg)];
            }
#pragma unroll
            for (int i = 0; i < GN_KEEP; ++i) {
                if (pl + i * PL < npos) {
                    if (pass == 0) { s += keep[i]; } else { const f32x4 v = keep[i] - mu; s += v * v; }
                }
            }
            st4(part + (pl * Q + q) * 4, s);
        }
        __syncthreads();
        for (int cc = tid; cc < CW; cc += GN_THREADS) {
            float t = 0.f;
            for (int i = 0; i < PL; ++i) t += part[i * CW + cc];
            chs[cc] = t;
        }
        __syncthreads();
        if (tid < GN_GPW) {
            float t = 0.f;
            for (int i = 0; i < cg; ++i) t += chs[tid * cg + i];
            float* o = part_out + (((size_t)n * 32 + blockIdx.y * GN_GPW + tid) * g.S + sidx) * 2;
            if (pass == 0) { gmean[tid] = t / (float)(cg * npos); o[0] = gmean[tid]; }
            else o[1] = t;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(GN_THREADS) void gn_chunk_apply_kernel(
    const float* __restrict__ s0, const float* __restrict__ s1, GnChunkGeom g, const float* __restrict__ part_in,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ film, int film_div,
    int film_ld, float eps, float* __restrict__ coefA, float* __restrict__ coefB, float* __restrict__ stats,
    float* __restrict__ act_out, int act_mode) {
    const int C = g.C0 + g.C1, cg = C / 32, CW = GN_GPW * cg, Q = CW / 4, PL = g.PL;
    const float rcg = __builtin_amdgcn_rcpf((float)cg);
    const int sidx = blockIdx.x, n = blockIdx.z, cbase = blockIdx.y * CW;
    const int tid = threadIdx.x;
    const bool active = tid < PL * Q;
    const int q = active ? tid % Q : 0, pl = active ? tid / Q : 0;
    const int c = cbase + q * 4;
    const int PC = GN_KEEP * PL;
    const int p_first = sidx * PC;
    const int npos = min(PC, g.P - p_first);
    __shared__ float gmean[GN_GPW], grstd[GN_GPW];
    __shared__ __attribute__((aligned(16))) float shA[GN_MAXCW], shB[GN_MAXCW];
    f32x4 keep[GN_KEEP];
    gn_chunk_load(s0, s1, g, (size_t)n * g.P, p_first, pl, c, keep);       // in flight during the combination
    {   // 32 lanes per group, fixed summation order
        const int grp = tid >> 5, j = tid & 31;
        const float* pp = part_in + ((size_t)n * 32 + blockIdx.y * GN_GPW + grp) * g.S * 2;
        float sm = 0.f;
        for (int s = j; s < g.S; s += 32) sm += (float)(cg * min(PC, g.P - s * PC)) * pp[2 * s];
        for (int o = 16; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
        const float ntot = (float)cg * (float)g.P;
        const float mean = sm / ntot;
        float m2 = 0.f;
        for (int s = j; s < g.S; s += 32) {
            const float d = pp[2 * s] - mean;
            m2 += pp[2 * s + 1] + (float)(cg * min(PC, g.P - s * PC)) * d * d;
        }
        for (int o = 16; o > 0; o >>= 1) m2 += __shfl_xor(m2, o, 64);
        if (j == 0) {
            gmean[grp] = mean;
            grstd[grp] = 1.0f / sqrtf(m2 / ntot + eps);
        }
    }
    __syncthreads();
    if (stats && sidx == 0 && tid < GN_GPW) {
        stats[((size_t)n * 32 + blockIdx.y * GN_GPW + tid) * 2 + 0] = gmean[tid];
        stats[((size_t)n * 32 + blockIdx.y * GN_GPW + tid) * 2 + 1] = grstd[tid];
    }
    for (int cc = tid; cc < CW; cc += GN_THREADS) {
        const int ch = cbase + cc;
        const int gi = fdiv_small(cc, cg, rcg);
        float A = grstd[gi] * gamma[ch];
        float B = beta[ch] - gmean[gi] * A;
        if (film) {
            const float* f = film + (size_t)(n / film_div) * film_ld;
            const float sc = 1.0f + f[ch];
            A *= sc;
            B = B * sc + f[C + ch];
        }
        if (coefA && sidx == 0) {
            coefA[(size_t)n * C + ch] = A;
            coefB[(size_t)n * C + ch] = B;
        }
        shA[cc] = A;
        shB[cc] = B;
    }
    if (act_out == nullptr) return;
    __syncthreads();
    if (active) {
        const f32x4 a4 = ld4(shA + q * 4), b4 = ld4(shB + q * 4);
        const size_t pos0 = (size_t)n * g.P + p_first;
#pragma unroll
        for (int i = 0; i < GN_KEEP; ++i) {
            const int p = pl + i * PL;
            if (p < npos) {
                f32x4 v = keep[i] * a4 + b4;
                if (act_mode == LFVDM_ACT_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
                st4(act_out + (pos0 + p) * C + c, v);
            }
        }
    }
}

// does the slice need (and fit) the 16-float4 register budget of the single-launch kernel?
inline bool gn_keep16(int C, int P) {
    const int PL = GN_THREADS / (GN_GPW * (C / 32) / 4);
    return P > 8 * PL && P <= 16 * PL;
}

// chunked decomposition for (C, P): chunks per slice (0 = the single-launch kernel holds the slice in registers)
inline int gn_chunks(int C, int P, int* pl_out) {
    const int Q = GN_GPW * (C / 32) / 4;
    const int PL = GN_THREADS / Q;
    if (pl_out) *pl_out = PL;
    if (P <= 2 * GN_KEEP * PL) return 0;       // up to 16 float4 per thread: the single-launch kernel keeps the slice
    return (P + GN_KEEP * PL - 1) / (GN_KEEP * PL);
}

// -------------------------------------------------------------------------------------
// gn_temporal: one wave per (b, pixel); the sample is [T][C] with row stride P*C.
// Writes the normalised rows (they are also the residual of the attention block).
// -------------------------------------------------------------------------------------
constexpr int GT_MAXC = 512;

__global__ __launch_bounds__(256) void gn_temporal_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          float* __restrict__ y, int B, int T, int P, int C) {
    __shared__ float chs_all[4][GT_MAXC];
    __shared__ float gstat_all[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long sample = (long)blockIdx.x * 4 + wave;
    if (sample >= (long)B * P) return;  // whole wave exits together
    float* chs = chs_all[wave];
    float* gstat = gstat_all[wave];
    const int b = (int)(sample / P), p = (int)(sample % P);
    const int cg = C / 32;
    const float rcg = __builtin_amdgcn_rcpf((float)cg);
    const int Q = C / 4;
    const size_t base = ((size_t)b * T * P + p) * C;
    const size_t tstride = (size_t)P * C;
    const int E = T * Q;

    // lane layout: QL lanes along the channel quads, TL lanes along the frames (TL > 1 only when
    // the quads of one frame fill less than a wave); per-channel sums are combined with shuffles,
    // so the result is deterministic (no LDS atomics).
    const int TL = (Q <= 64 && 64 % Q == 0) ? 64 / Q : 1;
    const int QL = 64 / TL;
    const int ql = lane % QL, tl = lane / QL;
    for (int pass = 0; pass < 2; ++pass) {
        for (int q = ql; q < Q; q += QL) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            f32x4 mu = {0.f, 0.f, 0.f, 0.f};
            if (pass) {
                mu.x = gstat[fdiv_small(q * 4 + 0, cg, rcg)]; mu.y = gstat[fdiv_small(q * 4 + 1, cg, rcg)];
                mu.z = gstat[fdiv_small(q * 4 + 2, cg, rcg)]; mu.w = gstat[fdiv_small(q * 4 + 3, cg, rcg)];
            }
            for (int t = tl; t < T; t += TL) {
                f32x4 v = ld4(x + base + t * tstride + q * 4);
                if (pass) { v = v - mu; s += v * v; } else { s += v; }
            }
            for (int o = QL; o < 64; o <<= 1) {
                s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64);
                s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
            }
            if (tl == 0) st4(chs + q * 4, s);
        }
        wave_lds_fence();
        if (lane < 32) {
            float t = 0.f;
            for (int i = 0; i < cg; ++i) t += chs[lane * cg + i];
            const float inv = 1.0f / (float)(cg * T);
            if (pass == 0) gstat[lane] = t * inv;
            else gstat[32 + lane] = 1.0f / sqrtf(t * inv + eps);
        }
        wave_lds_fence();
    }
    for (int e = lane; e < E; e += 64) {
        const int t = e / Q, q = e - t * Q;
        f32x4 v = ld4(x + base + t * tstride + q * 4);
        const f32x4 ga = ld4(gamma + q * 4), be = ld4(beta + q * 4);
        f32x4 o;
        o.x = (v.x - gstat[fdiv_small(q * 4 + 0, cg, rcg)]) * gstat[32 + fdiv_small(q * 4 + 0, cg, rcg)] * ga.x + be.x;
        o.y = (v.y - gstat[fdiv_small(q * 4 + 1, cg, rcg)]) * gstat[32 + fdiv_small(q * 4 + 1, cg, rcg)] * ga.y + be.y;
        o.z = (v.z - gstat[fdiv_small(q * 4 + 2, cg, rcg)]) * gstat[32 + fdiv_small(q * 4 + 2, cg, rcg)] * ga.z + be.z;
        o.w = (v.w - gstat[fdiv_small(q * 4 + 3, cg, rcg)]) * gstat[32 + fdiv_small(q * 4 + 3, cg, rcg)] * ga.w + be.w;
        st4(y + base + t * tstride + q * 4, o);
    }
}

// Register-resident variant for C <= 256 (the channel quads of a frame fit one wave): the (b, pixel) sample - T x C
// floats, <= TIT float4 per lane - is read ONCE and both statistic passes and the normalisation run on registers
// (the kernel above reads it three times from L2, each pass a dependent round trip).  Same lane layout and same
// summation order as above: bitwise the same result.
template <int TIT>
__global__ __launch_bounds__(256) void gn_temporal_reg_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps,
                                                              float* __restrict__ y, int B, int T, int P, int C) {
    __shared__ float chs_all[4][256];
    __shared__ float gstat_all[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long sample = (long)blockIdx.x * 4 + wave;
    if (sample >= (long)B * P) return;  // whole wave exits together
    float* chs = chs_all[wave];
    float* gstat = gstat_all[wave];
    const int b = (int)(sample / P), p = (int)(sample % P);
    const int cg = C / 32;
    const float rcg = __builtin_amdgcn_rcpf((float)cg);
    const int Q = C / 4;                 // <= 64, divides 64
    const size_t base = ((size_t)b * T * P + p) * C;
    const size_t tstride = (size_t)P * C;
    const int TL = 64 / Q;
    const int q = lane % Q, tl = lane / Q;
    f32x4 keep[TIT];
#pragma unroll
    for (int i = 0; i < TIT; ++i) {
        const int t = tl + i * TL;
        keep[i] = t < T ? ld4(x + base + t * tstride + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TIT; ++i) {
            if (tl + i * TL < T) {
                if (pass) { const f32x4 v = keep[i] - mu; s += v * v; } else { s += keep[i]; }
            }
        }
        for (int o = Q; o < 64; o <<= 1) {
            s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64);
            s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
        }
        if (tl == 0) st4(chs + q * 4, s);
        wave_lds_fence();
        if (lane < 32) {
            float t = 0.f;
            for (int i = 0; i < cg; ++i) t += chs[lane * cg + i];
            const float inv = 1.0f / (float)(cg * T);
            if (pass == 0) gstat[lane] = t * inv;
            else gstat[32 + lane] = 1.0f / sqrtf(t * inv + eps);
        }
        wave_lds_fence();
        if (pass == 0) {
            mu.x = gstat[fdiv_small(q * 4 + 0, cg, rcg)]; mu.y = gstat[fdiv_small(q * 4 + 1, cg, rcg)];
            mu.z = gstat[fdiv_small(q * 4 + 2, cg, rcg)]; mu.w = gstat[fdiv_small(q * 4 + 3, cg, rcg)];
        } else {
            rs.x = gstat[32 + fdiv_small(q * 4 + 0, cg, rcg)]; rs.y = gstat[32 + fdiv_small(q * 4 + 1, cg, rcg)];
            rs.z = gstat[32 + fdiv_small(q * 4 + 2, cg, rcg)]; rs.w = gstat[32 + fdiv_small(q * 4 + 3, cg, rcg)];
        }
    }
    const f32x4 ga = ld4(gamma + q * 4), be = ld4(beta + q * 4);
#pragma unroll
    for (int i = 0; i < TIT; ++i) {
        const int t = tl + i * TL;
        if (t < T) {
            f32x4 o;
            o.x = (keep[i].x - mu.x) * rs.x * ga.x + be.x;
            o.y = (keep[i].y - mu.y) * rs.y * ga.y + be.y;
            o.z = (keep[i].z - mu.z) * rs.z * ga.z + be.z;
            o.w = (keep[i].w - mu.w) * rs.w * ga.w + be.w;
            st4(y + base + t * tstride + q * 4, o);
        }
    }
}

// -------------------------------------------------------------------------------------
// conv_in: x*(1-obs)+x0*obs, indicator channel = obs, 3x3 pad-1 conv, channels-last output.
// Workgroup = 64 consecutive pixels x all filters: lane = pixel (the NCHW input planes are read as contiguous runs of
// the wave), wave w = filters [w*Cout/4, (w+1)*Cout/4) in quads; weights [(C+1)*9][Cout] in LDS, read as wave-uniform
// (broadcast) float4.  (The first version gave every thread one pixel and ONE quad: 16 threads re-loaded each input
// value and four times as many workgroups staged the weight table - 13 us for a 30 MFLOP layer.)
// -------------------------------------------------------------------------------------
// The sampler's clock rides in this launch (the first of a denoising step, and nothing in it reads the timestep): one
// extra workgroup does what lfvdm_sampler_tick_fetch does - t <- max(t - 1, 0), model timestep <- table[t], FiLM rows of
// the new t fetched from the chain's table - one launch less per step.
struct ConvInTick {
    int64_t* t;                // [B] or NULL: no clock in this launch
    const float* table;
    float* model_t;
    const float* rows_all;
    float* rows;
    int B, rows_ld, row_floats;
};

// The clock's workgroup.  Its FiLM-row fetch reads a cold corner of the chain's table: the rows are copied eight float4
// per thread at a time with all loads issued before the first store (the one-load-per-trip loop was six dependent round
// trips - longer than the convolution's workgroups once those run on the matrix cores).
__device__ __forceinline__ void conv_in_clock(const ConvInTick& tk, int64_t* tnew) {
    if ((int)threadIdx.x < tk.B) {
        int64_t v = tk.t[threadIdx.x] - 1;
        v = v < 0 ? 0 : v;
        tk.t[threadIdx.x] = v;
        tnew[threadIdx.x] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < tk.B) tk.model_t[threadIdx.x] = tk.table[tnew[threadIdx.x]];     // (off the rows' critical path)
    const int q4 = tk.row_floats >> 2;
    for (int b = 0; b < tk.B; ++b) {
        const float* src = tk.rows_all + ((size_t)tnew[b] * tk.B + b) * tk.rows_ld;
        float* dst = tk.rows + (size_t)b * tk.rows_ld;
        for (int i0 = threadIdx.x; i0 < q4; i0 += 256 * 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 256 * u;
                v[u] = i < q4 ? ld4(src + 4 * i) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 256 * u;
                if (i < q4) st4(dst + 4 * i, v[u]);
            }
        }
    }
}

template <int QW, int CIT>      // QW float4 quads of output channels per thread (Cout = 16 * QW); CIT = C + 1 when it is
                                // known at compile time (4 / 5: the input gather is then fully unrolled), else 0
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ x0,
                                                      const float* __restrict__ obs, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out, int N,
                                                      int C, int H, int W, ConvInTick tk) {
    constexpr int Cout = 16 * QW;
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [k = ci*9+tap][Cout]
    if (tk.t != nullptr && blockIdx.x == gridDim.x - 1) {        // the clock's workgroup (workgroup-uniform branch)
        conv_in_clock(tk, reinterpret_cast<int64_t*>(wl));
        return;
    }
    const int Ci = CIT ? CIT : C + 1;
    const int K = Ci * 9;
    constexpr int CP = Cout + 4;                  // padded LDS row: the transposing writes below spread over the banks
    // w is OIHW = [co][k]: read it as it lies (consecutive threads -> consecutive floats: coalesced; the first version
    // walked co fastest, 256 cache lines per wave load, in every one of the 160 workgroups) and transpose in LDS
    {
        // (four 16-byte loads in flight per thread: the one-load-per-trip form was eleven dependent L2 round trips)
        const float rK = __builtin_amdgcn_rcpf((float)K);
        const int total4 = (K * Cout) >> 2;                   // Cout is a multiple of 16
        for (int i0 = threadIdx.x; i0 < total4; i0 += 256 * 4) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i4 = i0 + 256 * u;
                v[u] = i4 < total4 ? ld4(w + 4 * (size_t)i4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i4 = i0 + 256 * u;
                if (i4 < total4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * i4 + e;
                        int co = (int)((float)i * rK);
                        int k = i - co * K;
                        if (k < 0) { k += K; co -= 1; }
                        if (k >= K) { k -= K; co += 1; }
                        wl[k * CP + co] = v[u][e];
                    }
                }
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int HW = H * W;
    const long total = (long)N * HW;
    const long pix = (long)blockIdx.x * 64 + lane;
    const bool live = pix < total;
    const long pp = live ? pix : total - 1;
    const int n = (int)(pp / HW);
    const int p = (int)(pp - (long)n * HW);
    const int oy = p / W, ox = p - oy * W;
    const float ob = obs[n];
    f32x4 acc[QW];
#pragma unroll
    for (int q = 0; q < QW; ++q) acc[q] = ld4(bias + (grp * QW + q) * 4);
    const float* xn = x + (size_t)n * C * HW;
    const float* x0n = x0 + (size_t)n * C * HW;
    const float* wg = wl + grp * QW * 4;
    if constexpr (CIT > 0) {
        // all 9 * (C + 1) composited inputs first (independent loads: one exposed latency), then the FMAs
        float v[9][CIT];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
            const bool inb = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int ip = inb ? iy * W + ix : p;
#pragma unroll
            for (int ci = 0; ci < CIT; ++ci) {
                float t = ob;
                if (ci < CIT - 1) t = xn[ci * HW + ip] * (1.0f - ob) + x0n[ci * HW + ip] * ob;
                v[tap][ci] = inb ? t : 0.f;
            }
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ci = 0; ci < CIT; ++ci) {
                const float* wk = wg + (ci * 9 + tap) * CP;
#pragma unroll
                for (int q = 0; q < QW; ++q) acc[q] += v[tap][ci] * ld4(wk + q * 4);
            }
    } else {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
            const bool inb = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int ip = inb ? iy * W + ix : p;
            for (int ci = 0; ci < Ci; ++ci) {
                float v = ob;
                if (ci < C) v = xn[ci * HW + ip] * (1.0f - ob) + x0n[ci * HW + ip] * ob;
                v = inb ? v : 0.f;
                const float* wk = wg + (ci * 9 + tap) * CP;
#pragma unroll
                for (int q = 0; q < QW; ++q) acc[q] += v * ld4(wk + q * 4);
            }
        }
    }
    if (live) {
        float* o = out + (size_t)pix * Cout + grp * QW * 4;
#pragma unroll
        for (int q = 0; q < QW; ++q) st4(o + q * 4, acc[q]);
    }
}

// The same convolution on the matrix cores, for the latent / pixel models (C + 1 = 5 or 4 input channels: K = 45 / 36).
// Per workgroup of 64 pixels the composited 3x3 patches are laid out as an im2col tile A[64][KP] in LDS (KP = K padded
// to a multiple of 4; a thread's pixel is fixed and its k values are wave-uniform: coalesced reads of the (B,T,C,H,W)
// tensors, scalar tap decode), the filters - OIHW rows ARE [Cout][K] - as W[Cout][KP]; each wave multiplies its 16 pixels
// by all filters with v_mfma_f32_16x16x4_f32.  Every global load of a phase is issued before the first use (fully
// unrolled: K is a template parameter): the scalar kernel above measures 11.5 us whatever its arithmetic costs - a first
// matrix-core version with a rolled gather loop (twelve dependent round trips) measured the same.
template <int NCT, int CI>      // Cout = 16 NCT; CI = C + 1
__global__ __launch_bounds__(256) void conv_in_mfma_kernel(const float* __restrict__ x, const float* __restrict__ x0,
                                                           const float* __restrict__ obs, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, int N,
                                                           int H, int W, ConvInTick tk) {
    constexpr int Cout = 16 * NCT, C = CI - 1, K = 9 * CI, KP = (K + 3) & ~3, KS = KP + 1;   // odd LDS row stride
    constexpr int NKI = KP / 4;          // k values per wave (k = wave + 4 i)
    extern __shared__ __attribute__((aligned(16))) float wl[];
    if (tk.t != nullptr && blockIdx.x == gridDim.x - 1) {        // the clock's workgroup (see conv_in_kernel)
        conv_in_clock(tk, reinterpret_cast<int64_t*>(wl));
        return;
    }
    float* As = wl;                  // [64][KS]
    float* Ws = wl + 64 * KS;        // [Cout][KS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- all global loads first: the filters (wave w stages rows w, w + 4, ...; lane = k) and the im2col values
    float wv[Cout / 4];
#pragma unroll
    for (int i = 0; i < Cout / 4; ++i) wv[i] = lane < K ? w[(size_t)(wave + 4 * i) * K + lane] : 0.f;
    const int HW = H * W;
    const long total = (long)N * HW;
    const long pix = (long)blockIdx.x * 64 + lane;
    const bool live = pix < total;
    const long pp = live ? pix : total - 1;
    const int n = (int)(pp / HW);
    const int pq = (int)(pp - (long)n * HW);
    const int oy = pq / W, ox = pq - oy * W;
    const float ob = obs[n];
    const float* xn = x + (size_t)n * C * HW;
    const float* x0n = x0 + (size_t)n * C * HW;
    float xa[NKI], xb[NKI];
    bool inb[NKI];
#pragma unroll
    for (int i = 0; i < NKI; ++i) {
        const int k = wave + 4 * i;              // wave-uniform
        const int ci = k / 9, tap = k - ci * 9;
        const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
        inb[i] = live && k < K && iy >= 0 && iy < H && ix >= 0 && ix < W;
        const int ip = inb[i] ? iy * W + ix : pq;
        const int cc = ci < C ? ci : 0;
        xa[i] = xn[cc * HW + ip];
        xb[i] = x0n[cc * HW + ip];
    }
#pragma unroll
    for (int i = 0; i < Cout / 4; ++i)
        if (lane < KP) Ws[(wave + 4 * i) * KS + lane] = wv[i];
#pragma unroll
    for (int i = 0; i < NKI; ++i) {
        const int k = wave + 4 * i;
        const int ci = k / 9;
        float v = ci < C ? xa[i] * (1.0f - ob) + xb[i] * ob : ob;
        As[lane * KS + k] = inb[i] ? v : 0.f;
    }
    __syncthreads();
    const int r16 = lane & 15, kq = lane >> 4;
    f32x4 acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        const float b = bias[16 * c + r16];
        acc[c] = (f32x4){b, b, b, b};
    }
    const float* ar = As + (16 * wave + r16) * KS + kq;
    const float* wr = Ws + r16 * KS + kq;
#pragma unroll
    for (int s4 = 0; s4 < KP; s4 += 4) {
        const float a = ar[s4];
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wr[16 * c * KS + s4], acc[c], 0, 0, 0);
    }
    // D: lane holds rows 4 (lane >> 4) + r, column lane & 15 of each 16x16 tile
    const long p0 = (long)blockIdx.x * 64 + 16 * wave + 4 * kq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (p0 + r < total) {
            float* o = out + (size_t)(p0 + r) * Cout + r16;
#pragma unroll
            for (int c = 0; c < NCT; ++c) o[16 * c] = acc[c][r];
        }
    }
}

// -------------------------------------------------------------------------------------
// rowdot: one wave per output row o of a job; up to 8 input rows share each weight load.
// -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowdot_kernel(const lfvdm_rowdot_job* __restrict__ jobs, int njobs, int total_rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= total_rows) return;
    int j = 0;
    while (j + 1 < njobs && jobs[j + 1].row0 <= row) ++j;
    const lfvdm_rowdot_job J = jobs[j];
    const int o = row - J.row0;
    const float* wrow = J.W + (size_t)o * J.K;
    const float bo = J.b ? J.b[o] : 0.f;
    for (int m0 = 0; m0 < J.M; m0 += 8) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int k = lane * 4; k < J.K; k += 256) {
            const f32x4 wv = ld4(wrow + k);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (m0 + i >= J.M) break;
                f32x4 v;
                if (J.in_mode == 2) {
                    // sinusoidal embedding [cos | sin]; frequency table appended after the timesteps
                    const float t = J.in[m0 + i];
                    const float* fr = J.in + J.ldin;  // freqs[K/2]
                    const int half = J.K >> 1;
                    float e[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int kk = k + u;
                        e[u] = kk < half ? cosf(t * fr[kk]) : sinf(t * fr[kk - half]);
                    }
                    v.x = e[0]; v.y = e[1]; v.z = e[2]; v.w = e[3];
                } else {
                    v = ld4(J.in + (size_t)(m0 + i) * J.ldin + k);
                    if (J.in_mode == 1) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
                }
                acc[i] += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (m0 + i >= J.M) break;
            const float s = wave_sum(acc[i]);
            if (lane == 0) J.out[(size_t)(m0 + i) * J.ldout + o] = s + bo;
        }
    }
}

// Backward of the grouped small-M linears: one wave per block of RDB_ROWS output rows of one job.
//   dW[o][k] += sum_m dout[m][o] * actin(in[m][k]);  db[o] += sum_m dout[m][o];
//   din[m][k] += sum_o dout[m][o] * W[o][k]        (gradient w.r.t. the ACTIVATED input; float atomics, may be NULL)
// Every W row is read once.  din is tiny (M x K) and EVERY wave of a job adds to it: done per wave that is ~1700 waves
// hammering the same 1024 addresses (measured 63 us per launch, most of it atomic contention).  So the four waves of a
// workgroup sum their din partials in LDS and issue ONE set of atomics: 31.8 us.  (More tasks per wave would cut the
// atomics further but leaves too few workgroups for the dW read-modify-write stream: 2 tasks 44 us, 4 tasks 79 us.
// Workgroups whose tasks span two jobs fall back to per-task atomics.)
constexpr int RDB_ROWS = 8;    // rows per wave and task: the dW read-modify-write chain of a wave is serial, so keep it short
constexpr int RDB_TPW = 4;     // tasks per wave: upper bound of the LFVDM_ROWDOT_TPW tuning aid; the launcher uses 1
constexpr int RDB_KIT = 4;     // K <= 1024 on the grouped path (256 floats per lane sweep)
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const lfvdm_rowdot_bwd_job* __restrict__ jobs, int njobs, int total_tasks,
                                                         int tpw, const DetSlab ds) {
    __shared__ f32x4 red[3][4][RDB_KIT][64];         // waves 1..3 -> wave 0: [wave - 1][m][k sweep][lane]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int first = blockIdx.x * (4 * tpw);
    const int last = min(first + 4 * tpw, total_tasks) - 1;
    int jf = 0, jl = 0;
    while (jf + 1 < njobs && jobs[jf + 1].task0 <= first) ++jf;
    while (jl + 1 < njobs && jobs[jl + 1].task0 <= last) ++jl;
    const lfvdm_rowdot_bwd_job J0 = jobs[jf];
    // (deterministic mode: every wave task stores its din partial to ITS row of the zero-filled slab - no grouping)
    const bool grouped = jf == jl && J0.din != nullptr && J0.M <= 4 && J0.K <= 256 * RDB_KIT && !ds.slab;   // workgroup-uniform
    f32x4 accG[4][RDB_KIT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int it = 0; it < RDB_KIT; ++it) accG[i][it] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < tpw; ++r) {
        const int task = first + wave + 4 * r;
        if (task >= total_tasks) break;
        int j = jf;
        while (j + 1 < njobs && jobs[j + 1].task0 <= task) ++j;
        const lfvdm_rowdot_bwd_job J = jobs[j];
        const int o0 = (task - J.task0) * RDB_ROWS;
        const int o1 = min(o0 + RDB_ROWS, J.O);
        for (int m0 = 0; m0 < J.M; m0 += 4) {
            const int mc = min(4, J.M - m0);
            int it = 0;
            for (int k = lane * 4; k < J.K; k += 256, ++it) {
                f32x4 inv[4], accD[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    accD[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    inv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (i < mc) {
                        if (J.in_mode == 2) {
                            const float t = J.in[m0 + i];
                            const float* fr = J.in + J.ldin;
                            const int half = J.K >> 1;
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int kk = k + u;
                                inv[i][u] = kk < half ? cosf(t * fr[kk]) : sinf(t * fr[kk - half]);
                            }
                        } else {
                            f32x4 v = ld4(J.in + (size_t)(m0 + i) * J.ldin + k);
                            if (J.in_mode == 1) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
                            inv[i] = v;
                        }
                    }
                }
                // the task's RDB_ROWS output rows: weight rows, gradient rows and dout values are all requested before the
                // first use (one row per trip was a chain of dependent round trips: 32 us per launch)
                f32x4 wv[RDB_ROWS], gv[RDB_ROWS];
                float dv[RDB_ROWS][4];
#pragma unroll
                for (int u = 0; u < RDB_ROWS; ++u) {
                    const int o = min(o0 + u, o1 - 1);
                    wv[u] = ld4(J.W + (size_t)o * J.K + k);
                    gv[u] = ld4(J.dW + (size_t)o * J.K + k);
#pragma unroll
                    for (int i = 0; i < 4; ++i) dv[u][i] = i < mc ? J.dout[(size_t)(m0 + i) * J.lddout + o] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < RDB_ROWS; ++u) {
                    if (o0 + u < o1) {
                        f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (i < mc) {
                                g += inv[i] * dv[u][i];
                                accD[i] += wv[u] * dv[u][i];
                            }
                        }
                        st4(J.dW + (size_t)(o0 + u) * J.K + k, gv[u] + g);
                    }
                }
                if (grouped) {                       // (M <= 4: a single m0 pass; `it` < RDB_KIT)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int q = 0; q < RDB_KIT; ++q)
                            if (q == it) accG[i][q] += accD[i];
                } else if (J.din) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (i < mc) {
                            float* dst = J.din + (size_t)(m0 + i) * J.lddin + k;
                            det_add(ds, task, dst + 0, accD[i].x); det_add(ds, task, dst + 1, accD[i].y);
                            det_add(ds, task, dst + 2, accD[i].z); det_add(ds, task, dst + 3, accD[i].w);
                        }
                    }
                }
            }
        }
        if (J.db && lane < o1 - o0) {
            float t = 0.f;
            for (int m = 0; m < J.M; ++m) t += J.dout[(size_t)m * J.lddout + o0 + lane];
            J.db[o0 + lane] += t;
        }
    }
    if (!grouped) return;                            // workgroup-uniform: no barrier is skipped by part of the workgroup
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < RDB_KIT; ++q) red[wave - 1][i][q][lane] = accG[i][q];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < RDB_KIT; ++q) {
            const int k = lane * 4 + 256 * q;
            if (k < J0.K) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < J0.M) {
                        const f32x4 t = accG[i][q] + red[0][i][q][lane] + red[1][i][q][lane] + red[2][i][q][lane];
                        float* dst = J0.din + (size_t)i * J0.lddin + k;
                        atomicAdd(dst + 0, t.x); atomicAdd(dst + 1, t.y);
                        atomicAdd(dst + 2, t.z); atomicAdd(dst + 3, t.w);
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" int lfvdm_rowdot_bwd(const lfvdm_rowdot_bwd_job* jobs_dev, int njobs, int total_tasks, void* stream) {
    if (!jobs_dev || njobs <= 0 || total_tasks <= 0) return LFVDM_E_SHAPE;
    static const int tpw_env = getenv("LFVDM_ROWDOT_TPW") ? atoi(getenv("LFVDM_ROWDOT_TPW")) : 0;     // tuning aid
    const int tpw = tpw_env >= 1 && tpw_env <= RDB_TPW ? tpw_env : 1;
    const DetSlab none = {nullptr, nullptr, 0};
    hipLaunchKernelGGL(rowdot_bwd_kernel, dim3((total_tasks + 4 * tpw - 1) / (4 * tpw)), dim3(256), 0, (hipStream_t)stream,
                       jobs_dev, njobs, total_tasks, tpw, none);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

// Deterministic form: din_base / din_n = the array that holds every job's `din` rows (all jobs of the launch accumulate
// into it); det_ws holds total_tasks rows of din_n floats.  Zero-fill, partial stores, ordered sum.
extern "C" int lfvdm_rowdot_bwd_det(const lfvdm_rowdot_bwd_job* jobs_dev, int njobs, int total_tasks, float* din_base,
                                    int64_t din_n, float* det_ws, int64_t det_ws_floats, void* stream) {
    if (!jobs_dev || njobs <= 0 || total_tasks <= 0) return LFVDM_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    DetSlab ds = {nullptr, nullptr, 0};
    if (din_base) {
        if (!det_ws || din_n <= 0 || (int64_t)total_tasks * din_n > det_ws_floats) return LFVDM_E_SHAPE;
        if (hipMemsetAsync(det_ws, 0, (size_t)total_tasks * din_n * sizeof(float), s) != hipSuccess) return LFVDM_E_LAUNCH;
        ds = {det_ws, din_base, (long)din_n};
    }
    hipLaunchKernelGGL(rowdot_bwd_kernel, dim3((total_tasks + 3) / 4), dim3(256), 0, s, jobs_dev, njobs, total_tasks, 1, ds);
    LFVDM_CHECK_LAUNCH();
    if (din_base) return lfvdm_det_reduce_launch(din_base, det_ws, (long)din_n, total_tasks, s);
    return LFVDM_OK;
}

extern "C" int lfvdm_gn_coef_stats(const float* src0, const float* src1, int C0, int C1, int N, int P, const float* gamma,
                                   const float* beta, const float* film, int film_div, int film_ld, float eps,
                                   float* coefA, float* coefB, float* stats, void* stream) {
    const int C = C0 + C1;
    if (N <= 0 || P <= 0 || C <= 0 || C % 32 || C0 % 4 || C > 1024) return LFVDM_E_SHAPE;
    if (C1 > 0 && !src1) return LFVDM_E_SHAPE;
    if (film && film_div <= 0) return LFVDM_E_SHAPE;
    if (gn_wave_launch(src0, src1, C0, C1, N, P, gamma, beta, film, film_div, film_ld, eps, coefA, coefB, stats, nullptr, 0,
                       (hipStream_t)stream)) {
    } else if (gn_keep16(C, P))
        hipLaunchKernelGGL(gn_coef_kernel<16>, dim3(N, 32 / GN_GPW), dim3(GN_THREADS), 0, (hipStream_t)stream, src0, src1, C0,
                           C1, P, gamma, beta, film, film_div, film_ld, eps, coefA, coefB, stats, (float*)nullptr, 0);
    else
        hipLaunchKernelGGL(gn_coef_kernel<8>, dim3(N, 32 / GN_GPW), dim3(GN_THREADS), 0, (hipStream_t)stream, src0, src1, C0,
                           C1, P, gamma, beta, film, film_div, film_ld, eps, coefA, coefB, stats, (float*)nullptr, 0);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_gn_apply(const float* src0, const float* src1, int C0, int C1, int N, int P, const float* gamma,
                              const float* beta, const float* film, int film_div, int film_ld, float eps, int act,
                              float* out, float* coefA, float* coefB, float* stats, void* stream) {
    const int C = C0 + C1;
    if (N <= 0 || P <= 0 || C <= 0 || C % 32 || C0 % 4 || C > 1024 || !out) return LFVDM_E_SHAPE;
    if (C1 > 0 && !src1) return LFVDM_E_SHAPE;
    if (film && film_div <= 0) return LFVDM_E_SHAPE;
    if ((coefA == nullptr) != (coefB == nullptr)) return LFVDM_E_SHAPE;
    if (gn_wave_launch(src0, src1, C0, C1, N, P, gamma, beta, film, film_div, film_ld, eps, coefA, coefB, stats, out, act,
                       (hipStream_t)stream)) {
    } else if (gn_keep16(C, P))
        hipLaunchKernelGGL(gn_coef_kernel<16>, dim3(N, 32 / GN_GPW), dim3(GN_THREADS), 0, (hipStream_t)stream, src0, src1, C0,
                           C1, P, gamma, beta, film, film_div, film_ld, eps, coefA, coefB, stats, out, act);
    else
        hipLaunchKernelGGL(gn_coef_kernel<8>, dim3(N, 32 / GN_GPW), dim3(GN_THREADS), 0, (hipStream_t)stream, src0, src1, C0,
                           C1, P, gamma, beta, film, film_div, film_ld, eps, coefA, coefB, stats, out, act);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_gn_apply_part(const float* src, int C, int N, int P, int cg, const float* gamma, const float* beta, float eps,
                                   int act, float* out, int ldo, void* stream) {
    if (!src || !gamma || !beta || !out || N <= 0 || N > 65535 || P <= 0 || P > 256 || C <= 0 || C % 16 || ldo < C || ldo % 4)
        return LFVDM_E_SHAPE;
    if ((cg != 2 && cg != 4 && cg != 8 && cg != 16) || C % cg) return LFVDM_E_UNSUPPORTED;
    const dim3 grid(N, C / 16);
    hipStream_t s = (hipStream_t)stream;
#define LFVDM_GNP(G)                                                                                                          \
    hipLaunchKernelGGL(gn_wave_kernel<G>, grid, dim3(64), 0, s, src, (const float*)nullptr, C, 0, P, gamma, beta, (const float*)nullptr, \
                       1, 0, eps, (float*)nullptr, (float*)nullptr, (float*)nullptr, out, act, ldo)
    if (cg == 2) LFVDM_GNP(2);
    else if (cg == 4) LFVDM_GNP(4);
    else if (cg == 8) LFVDM_GNP(8);
    else LFVDM_GNP(16);
#undef LFVDM_GNP
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" long lfvdm_gn_apply_ws_floats(int C, int N, int P) {
    if (C <= 0 || C % 32 || C > 1024 || N <= 0 || P <= 0) return 0;
    const int S = gn_chunks(C, P, nullptr);
    return (long)N * 32 * S * 2;
}

extern "C" int lfvdm_gn_apply_ws(const float* src0, const float* src1, int C0, int C1, int N, int P, const float* gamma,
                                 const float* beta, const float* film, int film_div, int film_ld, float eps, int act,
                                 float* out, float* coefA, float* coefB, float* stats, float* ws, long ws_floats,
                                 void* stream) {
    const int C = C0 + C1;
    if (N <= 0 || P <= 0 || C <= 0 || C % 32 || C0 % 4 || C > 1024 || !out) return LFVDM_E_SHAPE;
    if (C1 > 0 && !src1) return LFVDM_E_SHAPE;
    if (film && film_div <= 0) return LFVDM_E_SHAPE;
    if ((coefA == nullptr) != (coefB == nullptr)) return LFVDM_E_SHAPE;
    int PL = 0;
    const int S = gn_chunks(C, P, &PL);
    if (S == 0)
        return lfvdm_gn_apply(src0, src1, C0, C1, N, P, gamma, beta, film, film_div, film_ld, eps, act, out, coefA, coefB, stats,
                              stream);
    if (!ws || ws_floats < (long)N * 32 * S * 2 || N > 65535) return LFVDM_E_SHAPE;
    const GnChunkGeom g = {C0, C1, P, S, PL};
    const dim3 grid(S, 32 / GN_GPW, N);
    hipLaunchKernelGGL(gn_chunk_stats_kernel, grid, dim3(GN_THREADS), 0, (hipStream_t)stream, src0, src1, g, ws);
    LFVDM_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_chunk_apply_kernel, grid, dim3(GN_THREADS), 0, (hipStream_t)stream, src0, src1, g, (const float*)ws,
                       gamma, beta, film, film_div, film_ld, eps, coefA, coefB, stats, out, act);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_gn_coef(const float* src0, const float* src1, int C0, int C1, int N, int P, const float* gamma,
                             const float* beta, const float* film, int film_div, int film_ld, float eps, float* coefA,
                             float* coefB, void* stream) {
    return lfvdm_gn_coef_stats(src0, src1, C0, C1, N, P, gamma, beta, film, film_div, film_ld, eps, coefA, coefB, nullptr,
                               stream);
}

extern "C" int lfvdm_gn_temporal(const float* x, const float* gamma, const float* beta, float eps, float* y, int B, int T,
                                 int P, int C, void* stream) {
    if (B <= 0 || T <= 0 || P <= 0 || C % 32 || C > GT_MAXC) return LFVDM_E_SHAPE;
    const long samples = (long)B * P;
    const dim3 grid((unsigned)((samples + 3) / 4));
    const int Q = C / 4;
    if (Q <= 64 && 64 % Q == 0) {           // register-resident sample: frames per lane = ceil(T / (64 / Q))
        const int per_lane = (T + 64 / Q - 1) / (64 / Q);
        hipStream_t s = (hipStream_t)stream;
        if (per_lane <= 8) {
            hipLaunchKernelGGL(gn_temporal_reg_kernel<8>, grid, dim3(256), 0, s, x, gamma, beta, eps, y, B, T, P, C);
            LFVDM_CHECK_LAUNCH();
            return LFVDM_OK;
        }
        if (per_lane <= 16) {
            hipLaunchKernelGGL(gn_temporal_reg_kernel<16>, grid, dim3(256), 0, s, x, gamma, beta, eps, y, B, T, P, C);
            LFVDM_CHECK_LAUNCH();
            return LFVDM_OK;
        }
        if (per_lane <= 32) {
            hipLaunchKernelGGL(gn_temporal_reg_kernel<32>, grid, dim3(256), 0, s, x, gamma, beta, eps, y, B, T, P, C);
            LFVDM_CHECK_LAUNCH();
            return LFVDM_OK;
        }
    }
    hipLaunchKernelGGL(gn_temporal_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps, y, B, T, P, C);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

static int conv_in_launch(const float* x, const float* x0, const float* obs, const float* w, const float* bias, float* out,
                          int N, int C, int H, int W, int Cout, const ConvInTick& tk, hipStream_t s) {
    if (N <= 0 || C <= 0 || Cout % 16 || Cout > 256 || H <= 0 || W <= 0) return LFVDM_E_SHAPE;
    size_t lds = (size_t)(C + 1) * 9 * (Cout + 4) * sizeof(float);      // padded rows (see the kernel)
    if (lds > 64 * 1024) return LFVDM_E_UNSUPPORTED;
    if (lds < 64 * sizeof(int64_t)) lds = 64 * sizeof(int64_t);
    const long total = (long)N * H * W;
    const dim3 grid((unsigned)((total + 63) / 64) + (tk.t ? 1u : 0u));
    {   // matrix-core form (see conv_in_mfma_kernel) for the latent / pixel inputs and 64 / 128 filters
        static const bool off = getenv("LFVDM_CONV_IN_SCALAR") != nullptr;       // A/B aid
        if (!off && (C == 4 || C == 3) && (Cout == 64 || Cout == 128)) {
            const int KP = (9 * (C + 1) + 3) & ~3;
            size_t l2 = (size_t)(64 + Cout) * (KP + 1) * sizeof(float);
            if (l2 < 64 * sizeof(int64_t)) l2 = 64 * sizeof(int64_t);
#define LFVDM_CIM(NCT, CI) hipLaunchKernelGGL((conv_in_mfma_kernel<NCT, CI>), grid, dim3(256), l2, s, x, x0, obs, w, bias, out, N, H, W, tk)
            if (C == 4) { if (Cout == 64) LFVDM_CIM(4, 5); else LFVDM_CIM(8, 5); }
            else { if (Cout == 64) LFVDM_CIM(4, 4); else LFVDM_CIM(8, 4); }
#undef LFVDM_CIM
            LFVDM_CHECK_LAUNCH();
            return LFVDM_OK;
        }
    }
#define LFVDM_CONV_IN(Q)                                                                                              \
    case Q:                                                                                                           \
        if (C == 4) hipLaunchKernelGGL((conv_in_kernel<Q, 5>), grid, dim3(256), lds, s, x, x0, obs, w, bias, out, N, C, H, W, tk);      \
        else if (C == 3) hipLaunchKernelGGL((conv_in_kernel<Q, 4>), grid, dim3(256), lds, s, x, x0, obs, w, bias, out, N, C, H, W, tk); \
        else hipLaunchKernelGGL((conv_in_kernel<Q, 0>), grid, dim3(256), lds, s, x, x0, obs, w, bias, out, N, C, H, W, tk);             \
        break;
    switch (Cout / 16) {
        LFVDM_CONV_IN(1) LFVDM_CONV_IN(2) LFVDM_CONV_IN(3) LFVDM_CONV_IN(4) LFVDM_CONV_IN(5) LFVDM_CONV_IN(6) LFVDM_CONV_IN(7)
        LFVDM_CONV_IN(8) LFVDM_CONV_IN(9) LFVDM_CONV_IN(10) LFVDM_CONV_IN(11) LFVDM_CONV_IN(12) LFVDM_CONV_IN(13)
        LFVDM_CONV_IN(14) LFVDM_CONV_IN(15) LFVDM_CONV_IN(16)
        default: return LFVDM_E_SHAPE;
    }
#undef LFVDM_CONV_IN
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_conv_in(const float* x, const float* x0, const float* obs, const float* w, const float* bias,
                             float* out, int N, int C, int H, int W, int Cout, void* stream) {
    const ConvInTick none = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    return conv_in_launch(x, x0, obs, w, bias, out, N, C, H, W, Cout, none, (hipStream_t)stream);
}

extern "C" int lfvdm_conv_in_tick(const float* x, const float* x0, const float* obs, const float* w, const float* bias,
                                  float* out, int N, int C, int H, int W, int Cout, int64_t* t,
                                  const float* model_timestep_table, float* model_t, int B, const float* rows_all, int rows_ld,
                                  float* rows, int row_floats, void* stream) {
    if (B <= 0 || B > 64 || !t || !model_timestep_table || !model_t || !rows_all || !rows) return LFVDM_E_SHAPE;
    if (row_floats <= 0 || (row_floats & 3) || (rows_ld & 3) || rows_ld < row_floats) return LFVDM_E_SHAPE;
    const ConvInTick tk = {t, model_timestep_table, model_t, rows_all, rows, B, rows_ld, row_floats};
    return conv_in_launch(x, x0, obs, w, bias, out, N, C, H, W, Cout, tk, (hipStream_t)stream);
}

namespace {
__global__ __launch_bounds__(256) void silu_kernel(const float* __restrict__ in, float* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 v = ld4(in + 4 * i);
        v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w);
        st4(out + 4 * i, v);
    }
}
}  // namespace

// out = x * sigmoid(x), the very silu_f the row-dot launches apply to their inputs in in_mode 1: materialising it once
// lets a launch with MANY batch rows (the sampler's per-chain tables) run in in_mode 0 instead of re-evaluating it per
// output row.  n must be a multiple of 4.
extern "C" int lfvdm_silu(const float* in, float* out, int64_t n, void* stream) {
    if (!in || !out || n <= 0 || (n & 3)) return LFVDM_E_SHAPE;
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(silu_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, (long)(n / 4));
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_rowdot(const lfvdm_rowdot_job* jobs_dev, int njobs, int total_rows, void* stream) {
    if (njobs <= 0 || total_rows <= 0) return LFVDM_E_SHAPE;
    hipLaunchKernelGGL(rowdot_kernel, dim3((total_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs,
                       total_rows);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}
