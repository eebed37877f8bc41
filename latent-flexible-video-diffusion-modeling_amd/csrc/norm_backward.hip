// GroupNorm(+FiLM)(+SiLU) backward for the fused "statistics -> affine in the conv operand" scheme, and the
// temporal GroupNorm backward.  HBM/L2-bound reductions; wave = 64.
//
// Forward (norm_embed.hip):  z = x*A + B,  a = act(z),  A = rstd*g',  B = b' - mean*A,  g' = gamma*(1+scale).
// Given da (gradient w.r.t. a, from the conv data gradient):
//   dz      = da * act'(z)
//   s1[n,c] = sum_p dz,          s2[n,c] = sum_p dz * xhat          (xhat = (x-mean)*rstd)
//   S1[n,g] = sum_{c in g} g'_c s1[n,c],   S2[n,g] = sum_{c in g} g'_c s2[n,c]
//   dx      = rstd * (g'_c dz - (S1 + xhat*S2) / (cg*P))
// and the parameter / FiLM gradients are tiny contractions of s1, s2 done by the caller.
#include "common_hip.h"

namespace {

constexpr int GB_GPW = 8;
constexpr int GB_THREADS = 256;

__device__ __forceinline__ f32x4 ldcat(const float* s0, const float* s1, int C0, int C1, size_t pos, int c) {
    return c < C0 ? ld4(s0 + pos * C0 + c) : ld4(s1 + pos * C1 + (c - C0));
}

__device__ __forceinline__ float dsilu(float z) {
    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
    return sg * (1.0f + z * (1.0f - sg));
}

// pass 1: per-(n, c) sums s1, s2 -> sums[n][c][2]
__global__ __launch_bounds__(GB_THREADS) void gn_bwd_stats_kernel(
    const float* __restrict__ da, const float* __restrict__ s0, const float* __restrict__ s1p, int C0, int C1, int P,
    const float* __restrict__ coefA, const float* __restrict__ coefB, const float* __restrict__ stats, int act,
    float* __restrict__ sums) {
    const int C = C0 + C1;
    const int cg = C / 32;
    const float rcg = __builtin_amdgcn_rcpf((float)cg);
    const int CW = GB_GPW * cg;
    const int Q = CW / 4;
    const int n = blockIdx.x;
    const int cbase = blockIdx.y * CW;
    const int PL = GB_THREADS / Q;
    const int tid = threadIdx.x;
    const bool active = tid < PL * Q;
    const int q = active ? tid % Q : 0;
    const int pl = active ? tid / Q : 0;
    const int c = cbase + q * 4;
    __shared__ float part[2][GB_THREADS * 4];

    f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const f32x4 A = ld4(coefA + (size_t)n * C + c), B = ld4(coefB + (size_t)n * C + c);
        f32x4 mu, rs;
        const float* st = stats + (size_t)n * 64;
        mu.x = st[2 * (fdiv_small(c + 0, cg, rcg))]; rs.x = st[2 * (fdiv_small(c + 0, cg, rcg)) + 1];
        mu.y = st[2 * (fdiv_small(c + 1, cg, rcg))]; rs.y = st[2 * (fdiv_small(c + 1, cg, rcg)) + 1];
        mu.z = st[2 * (fdiv_small(c + 2, cg, rcg))]; rs.z = st[2 * (fdiv_small(c + 2, cg, rcg)) + 1];
        mu.w = st[2 * (fdiv_small(c + 3, cg, rcg))]; rs.w = st[2 * (fdiv_small(c + 3, cg, rcg)) + 1];
        const size_t pos0 = (size_t)n * P;
        for (int p = pl; p < P; p += PL) {
            const f32x4 x = ldcat(s0, s1p, C0, C1, pos0 + p, c);
            f32x4 dz = ld4(da + (pos0 + p) * C + c);
            if (act == LFVDM_ACT_SILU) {
                const f32x4 z = x * A + B;
                dz.x *= dsilu(z.x); dz.y *= dsilu(z.y); dz.z *= dsilu(z.z); dz.w *= dsilu(z.w);
            }
            a1 += dz;
            a2 += dz * ((x - mu) * rs);
        }
        st4(part[0] + (pl * Q + q) * 4, a1);
        st4(part[1] + (pl * Q + q) * 4, a2);
    }
    __syncthreads();
    for (int cc = tid; cc < CW; cc += GB_THREADS) {
        float t1 = 0.f, t2 = 0.f;
        for (int i = 0; i < PL; ++i) { t1 += part[0][i * CW + cc]; t2 += part[1][i * CW + cc]; }
        sums[((size_t)n * C + cbase + cc) * 2 + 0] = t1;
        sums[((size_t)n * C + cbase + cc) * 2 + 1] = t2;
    }
}

// pass 2: dx (split over the two concat destinations, overwrite or accumulate)
struct GnParamGradArgs {
    const float *gamma, *beta, *film;
    float *dgamma, *dbeta, *dfilm;
    int film_ld, dfilm_ld, T;
    const float* add;     // optional [N*P][add_ld] rows added to dx (the skip path's gradient of a ResBlock input)
    int add_ld;
    const float* add2;    // a second optional addend (the gradient the decoder's skip connection sends to the same tensor)
    int add2_ld;
};

__global__ __launch_bounds__(GB_THREADS) void gn_bwd_apply_kernel(
    const float* __restrict__ da, const float* __restrict__ s0, const float* __restrict__ s1p, int C0, int C1, int P,
    const float* __restrict__ coefA, const float* __restrict__ coefB, const float* __restrict__ stats,
    const float* __restrict__ sums, int act, float* __restrict__ out0, float* __restrict__ out1, int acc0, int acc1,
    GnParamGradArgs pg) {
    const int C = C0 + C1;
    const int cg = C / 32;
    const float rcg = __builtin_amdgcn_rcpf((float)cg);
    const int CW = GB_GPW * cg;
    const int Q = CW / 4;
    const int n = blockIdx.x;
    const int cbase = blockIdx.y * CW;
    const int PL = GB_THREADS / Q;
    const int tid = threadIdx.x;
    __shared__ float gS1[GB_GPW], gS2[GB_GPW];
    if (pg.dgamma != nullptr) {
        // parameter gradients of this (sample, channel slice), folded in with float atomics (same sums as
        // gn_param_grads_kernel, one launch less per GroupNorm):
        //   dgamma[c] += s2 * (1 + scale);  dbeta[c] += s1 * (1 + scale);  dfilm[b][c] += s2*gamma + s1*beta;  dfilm[b][C+c] += s1
        for (int cc = tid; cc < CW; cc += GB_THREADS) {
            const int ch = cbase + cc;
            const float s1 = sums[((size_t)n * C + ch) * 2 + 0], s2 = sums[((size_t)n * C + ch) * 2 + 1];
            float sc1 = 1.0f;
            if (pg.film != nullptr) {
                const int b = n / pg.T;
                sc1 += pg.film[(size_t)b * pg.film_ld + ch];
                atomicAdd(pg.dfilm + (size_t)b * pg.dfilm_ld + ch, s2 * pg.gamma[ch] + s1 * pg.beta[ch]);
                atomicAdd(pg.dfilm + (size_t)b * pg.dfilm_ld + C + ch, s1);
            }
            atomicAdd(pg.dgamma + ch, s2 * sc1);
            atomicAdd(pg.dbeta + ch, s1 * sc1);
        }
    }
    if (tid < GB_GPW) {
        const int g = blockIdx.y * GB_GPW + tid;
        const float rstd = stats[((size_t)n * 32 + g) * 2 + 1];
        float S1 = 0.f, S2 = 0.f;
        for (int i = 0; i < cg; ++i) {
            const int ch = g * cg + i;
            const float gp = coefA[(size_t)n * C + ch] / rstd;   // g' = A / rstd (rstd > 0)
            S1 += gp * sums[((size_t)n * C + ch) * 2 + 0];
            S2 += gp * sums[((size_t)n * C + ch) * 2 + 1];
        }
        const float inv = 1.0f / (float)(cg * P);
        gS1[tid] = S1 * inv;
        gS2[tid] = S2 * inv;
    }
    __syncthreads();
    if (tid >= PL * Q) return;
    const int q = tid % Q, pl = tid / Q;
    const int c = cbase + q * 4;
    const f32x4 A = ld4(coefA + (size_t)n * C + c), B = ld4(coefB + (size_t)n * C + c);
    f32x4 mu, rs, S1, S2;
    const float* st = stats + (size_t)n * 64;
#define LFVDM_G(k, f)                                                             \
    { const int g = fdiv_small(c + k, cg, rcg); mu.f = st[2 * g]; rs.f = st[2 * g + 1];         \
      S1.f = gS1[g - blockIdx.y * GB_GPW]; S2.f = gS2[g - blockIdx.y * GB_GPW]; }
    LFVDM_G(0, x) LFVDM_G(1, y) LFVDM_G(2, z) LFVDM_G(3, w)
#undef LFVDM_G
    const size_t pos0 = (size_t)n * P;
    const bool first = c < C0;
    float* out = first ? out0 : out1;
    const int Cd = first ? C0 : C1;
    const int cd = first ? c : c - C0;
    const int acc = first ? acc0 : acc1;
    for (int p = pl; p < P; p += PL) {
        const f32x4 x = ldcat(s0, s1p, C0, C1, pos0 + p, c);
        f32x4 dz = ld4(da + (pos0 + p) * C + c);
        if (act == LFVDM_ACT_SILU) {
            const f32x4 z = x * A + B;
            dz.x *= dsilu(z.x); dz.y *= dsilu(z.y); dz.z *= dsilu(z.z); dz.w *= dsilu(z.w);
        }
        const f32x4 xh = (x - mu) * rs;
        f32x4 dx = A * dz - rs * (S1 + xh * S2);     // rstd*g'*dz == A*dz
        float* o = out + (pos0 + p) * Cd + cd;
        if (acc) dx += ld4(o);
        if (pg.add != nullptr) dx += ld4(pg.add + (pos0 + p) * pg.add_ld + c);
        if (pg.add2 != nullptr) dx += ld4(pg.add2 + (pos0 + p) * pg.add2_ld + c);
        st4(o, dx);
    }
}

// Both passes in ONE launch for the training path (same grid: a workgroup owns a (sample, 8 groups) slice, so nothing
// crosses workgroups): pass 1 leaves the per-channel sums in LDS, the parameter / FiLM gradients go out as float atomics,
// pass 2 re-reads the slice (L2-hot: this workgroup has just streamed it).  Saves a launch and the second ramp-up per
// GroupNorm (36 per training micro-step); loads are issued four positions deep in both passes.
__global__ __launch_bounds__(GB_THREADS) void gn_bwd_fused_kernel(
    const float* __restrict__ da, const float* __restrict__ s0, const float* __restrict__ s1p, int C0, int C1, int P,
    const float* __restrict__ coefA, const float* __restrict__ coefB, const float* __restrict__ stats, int act,
    float* __restrict__ out0, float* __restrict__ out1, GnParamGradArgs pg, float* __restrict__ sums_out) {
    const int C = C0 + C1;
    const int cg = C / 32;
    const float rcg = __builtin_amdgcn_rcpf((float)cg);
    const int CW = GB_GPW * cg;
    const int Q = CW / 4;
    const int n = blockIdx.x;
    const int cbase = blockIdx.y * CW;
    const int PL = GB_THREADS / Q;
    const int tid = threadIdx.x;
    const bool active = tid < PL * Q;
    const int q = active ? tid % Q : 0;
    const int pl = active ? tid / Q : 0;
    const int c = cbase + q * 4;
    __shared__ float part[2][GB_THREADS * 4];
    __shared__ float chS[2][GB_GPW * 32];            // per-channel s1, s2 of the slice (cg <= 32)
    __shared__ float gS1[GB_GPW], gS2[GB_GPW];

    const f32x4 A = ld4(coefA + (size_t)n * C + c), B = ld4(coefB + (size_t)n * C + c);
    f32x4 mu, rs;
    const float* st = stats + (size_t)n * 64;
    mu.x = st[2 * (fdiv_small(c + 0, cg, rcg))]; rs.x = st[2 * (fdiv_small(c + 0, cg, rcg)) + 1];
    mu.y = st[2 * (fdiv_small(c + 1, cg, rcg))]; rs.y = st[2 * (fdiv_small(c + 1, cg, rcg)) + 1];
    mu.z = st[2 * (fdiv_small(c + 2, cg, rcg))]; rs.z = st[2 * (fdiv_small(c + 2, cg, rcg)) + 1];
    mu.w = st[2 * (fdiv_small(c + 3, cg, rcg))]; rs.w = st[2 * (fdiv_small(c + 3, cg, rcg)) + 1];
    const size_t pos0 = (size_t)n * P;
    const bool silu = act == LFVDM_ACT_SILU;
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        for (int p = pl; p < P; p += 4 * PL) {
            f32x4 x[4], dz[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pp = min(p + u * PL, P - 1);
                x[u] = ldcat(s0, s1p, C0, C1, pos0 + pp, c);
                dz[u] = ld4(da + (pos0 + pp) * C + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (p + u * PL < P) {
                    if (silu) {
                        const f32x4 z = x[u] * A + B;
                        dz[u].x *= dsilu(z.x); dz[u].y *= dsilu(z.y); dz[u].z *= dsilu(z.z); dz[u].w *= dsilu(z.w);
                    }
                    a1 += dz[u];
                    a2 += dz[u] * ((x[u] - mu) * rs);
                }
            }
        }
        st4(part[0] + (pl * Q + q) * 4, a1);
        st4(part[1] + (pl * Q + q) * 4, a2);
    }
    __syncthreads();
    for (int cc = tid; cc < CW; cc += GB_THREADS) {
        float t1 = 0.f, t2 = 0.f;
        for (int i = 0; i < PL; ++i) { t1 += part[0][i * CW + cc]; t2 += part[1][i * CW + cc]; }
        chS[0][cc] = t1;
        chS[1][cc] = t2;
        const int ch = cbase + cc;
        if (sums_out != nullptr) {      // deterministic mode: every (sample, channel) pair is written by exactly one
            sums_out[((size_t)n * C + ch) * 2 + 0] = t1;      // workgroup; lfvdm_gn_param_grads adds them in sample order
            sums_out[((size_t)n * C + ch) * 2 + 1] = t2;
            continue;
        }
        float sc1 = 1.0f;
        if (pg.film != nullptr) {
            const int b = n / pg.T;
            sc1 += pg.film[(size_t)b * pg.film_ld + ch];
            atomicAdd(pg.dfilm + (size_t)b * pg.dfilm_ld + ch, t2 * pg.gamma[ch] + t1 * pg.beta[ch]);
            atomicAdd(pg.dfilm + (size_t)b * pg.dfilm_ld + C + ch, t1);
        }
        atomicAdd(pg.dgamma + ch, t2 * sc1);
        atomicAdd(pg.dbeta + ch, t1 * sc1);
    }
    __syncthreads();
    if (tid < GB_GPW) {
        const int g = blockIdx.y * GB_GPW + tid;
        const float rstd = stats[((size_t)n * 32 + g) * 2 + 1];
        float S1 = 0.f, S2 = 0.f;
        for (int i = 0; i < cg; ++i) {
            const float gp = coefA[(size_t)n * C + g * cg + i] / rstd;   // g' = A / rstd (rstd > 0)
            S1 += gp * chS[0][tid * cg + i];
            S2 += gp * chS[1][tid * cg + i];
        }
        const float inv = 1.0f / (float)(cg * P);
        gS1[tid] = S1 * inv;
        gS2[tid] = S2 * inv;
    }
    __syncthreads();
    if (!active) return;
    f32x4 S1, S2;
    S1.x = gS1[fdiv_small(q * 4 + 0, cg, rcg)]; S2.x = gS2[fdiv_small(q * 4 + 0, cg, rcg)];
    S1.y = gS1[fdiv_small(q * 4 + 1, cg, rcg)]; S2.y = gS2[fdiv_small(q * 4 + 1, cg, rcg)];
    S1.z = gS1[fdiv_small(q * 4 + 2, cg, rcg)]; S2.z = gS2[fdiv_small(q * 4 + 2, cg, rcg)];
    S1.w = gS1[fdiv_small(q * 4 + 3, cg, rcg)]; S2.w = gS2[fdiv_small(q * 4 + 3, cg, rcg)];
    const bool first = c < C0;
    float* out = first ? out0 : out1;
    const int Cd = first ? C0 : C1;
    const int cd = first ? c : c - C0;
    for (int p = pl; p < P; p += 4 * PL) {
        f32x4 x[4], dz[4], ad[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pp = min(p + u * PL, P - 1);
            x[u] = ldcat(s0, s1p, C0, C1, pos0 + pp, c);
            dz[u] = ld4(da + (pos0 + pp) * C + c);
            ad[u] = pg.add != nullptr ? ld4(pg.add + (pos0 + pp) * pg.add_ld + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (pg.add2 != nullptr) ad[u] += ld4(pg.add2 + (pos0 + pp) * pg.add2_ld + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (p + u * PL < P) {
                if (silu) {
                    const f32x4 z = x[u] * A + B;
                    dz[u].x *= dsilu(z.x); dz[u].y *= dsilu(z.y); dz[u].z *= dsilu(z.z); dz[u].w *= dsilu(z.w);
                }
                const f32x4 xh = (x[u] - mu) * rs;
                st4(out + (pos0 + p + u * PL) * Cd + cd, A * dz[u] - rs * (S1 + xh * S2) + ad[u]);
            }
        }
    }
}

// -------------------------------------------------------------------------------------
// Large maps (pixel space, 32x32 latents, wide concats): the (sample, 8 groups) decomposition above gives N*4 workgroups
// whatever P is - 80 workgroups streaming ~1 GB per GroupNorm at 20 x 128 x 128 x 128.  Chunked form, the decomposition
// of the forward's gn_chunk_* kernels (norm_embed.hip): a workgroup owns a CHUNK of 8*PL positions of a (sample, 8 groups)
// slice and keeps x and dz of the chunk in registers.
//   pass 1 (gn_bwd_chunk_stats_kernel): per-channel partial sums s1, s2 of the chunk -> part[n][chunk][C][2]
//   pass 2 (gn_bwd_chunk_apply_kernel): every workgroup sums the S partials of its channels IN CHUNK ORDER (its chunk's
//     loads are already in flight), forms the group sums and writes dx of its chunk; the chunk-0 workgroup of a slice
//     also delivers the per-(n, c) sums: float atomics into dgamma / dbeta / dfilm (training path) and / or plain stores
//     to sums_out[n][C][2] (deterministic mode and the autograd delivery mode, which reduce them in a fixed order).
// Thousands of workgroups, x and da read twice, dx written once; everything except the optional atomics is deterministic.
// -------------------------------------------------------------------------------------
constexpr int GBC_KEEP = 8;

struct GnBwdChunkGeom { int C0, C1, P, S, PL; };

struct GnBwdChunkRegs {
    f32x4 x[GBC_KEEP], dz[GBC_KEEP];
};

// loads of the chunk (clamped positions: lanes past the end re-read the last row and are masked by the caller), then
// dz = da * act'(z) in place
__device__ __forceinline__ void gn_bwd_chunk_load(const float* da, const float* s0, const float* s1p, const GnBwdChunkGeom& g,
                                                  size_t pos0, int npos, int pl, int c, GnBwdChunkRegs& r) {
    const int C = g.C0 + g.C1;
#pragma unroll
    for (int i = 0; i < GBC_KEEP; ++i) {
        const int p = min(pl + i * g.PL, npos - 1);
        r.x[i] = ldcat(s0, s1p, g.C0, g.C1, pos0 + p, c);
        r.dz[i] = ld4(da + (pos0 + p) * C + c);
    }
}

__global__ __launch_bounds__(GB_THREADS) void gn_bwd_chunk_stats_kernel(
    const float* __restrict__ da, const float* __restrict__ s0, const float* __restrict__ s1p, GnBwdChunkGeom g,
    const float* __restrict__ coefA, const float* __restrict__ coefB, const float* __restrict__ stats, int act,
    float* __restrict__ part_out) {
    const int C = g.C0 + g.C1, cg = C / 32, CW = GB_GPW * cg, Q = CW / 4, PL = g.PL;
    const float rcg = __builtin_amdgcn_rcpf((float)cg);
    const int sidx = blockIdx.x, n = blockIdx.z, cbase = blockIdx.y * CW;
    const int tid = threadIdx.x;
    const bool active = tid < PL * Q;
    const int q = active ? tid % Q : 0, pl = active ? tid / Q : 0;
    const int c = cbase + q * 4;
    const int p_first = sidx * (GBC_KEEP * PL);
    const int npos = min(GBC_KEEP * PL, g.P - p_first);
    __shared__ float part[2][GB_THREADS * 4];
    GnBwdChunkRegs r;
    gn_bwd_chunk_load(da, s0, s1p, g, (size_t)n * g.P + p_first, npos, pl, c, r);
    const f32x4 A = ld4(coefA + (size_t)n * C + c), B = ld4(coefB + (size_t)n * C + c);
    f32x4 mu, rs;
    const float* st = stats + (size_t)n * 64;
    mu.x = st[2 * (fdiv_small(c + 0, cg, rcg))]; rs.x = st[2 * (fdiv_small(c + 0, cg, rcg)) + 1];
    mu.y = st[2 * (fdiv_small(c + 1, cg, rcg))]; rs.y = st[2 * (fdiv_small(c + 1, cg, rcg)) + 1];
    mu.z = st[2 * (fdiv_small(c + 2, cg, rcg))]; rs.z = st[2 * (fdiv_small(c + 2, cg, rcg)) + 1];
    mu.w = st[2 * (fdiv_small(c + 3, cg, rcg))]; rs.w = st[2 * (fdiv_small(c + 3, cg, rcg)) + 1];
    const bool silu = act == LFVDM_ACT_SILU;
    if (active) {
        f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < GBC_KEEP; ++i) {
            if (pl + i * PL < npos) {
                f32x4 dz = r.dz[i];
                if (silu) {
                    const f32x4 z = r.x[i] * A + B;
                    dz.x *= dsilu(z.x); dz.y *= dsilu(z.y); dz.z *= dsilu(z.z); dz.w *= dsilu(z.w);
                }
                a1 += dz;
                a2 += dz * ((r.x[i] - mu) * rs);
            }
        }
        st4(part[0] + (pl * Q + q) * 4, a1);
        st4(part[1] + (pl * Q + q) * 4, a2);
    }
    __syncthreads();
    for (int cc = tid; cc < CW; cc += GB_THREADS) {
        float t1 = 0.f, t2 = 0.f;
        for (int i = 0; i < PL; ++i) { t1 += part[0][i * CW + cc]; t2 += part[1][i * CW + cc]; }
        float* o = part_out + (((size_t)n * g.S + sidx) * C + cbase + cc) * 2;
        o[0] = t1;
        o[1] = t2;
    }
}

__global__ __launch_bounds__(GB_THREADS) void gn_bwd_chunk_apply_kernel(
    const float* __restrict__ da, const float* __restrict__ s0, const float* __restrict__ s1p, GnBwdChunkGeom g,
    const float* __restrict__ coefA, const float* __restrict__ coefB, const float* __restrict__ stats, int act,
    const float* __restrict__ part_in, float* __restrict__ out0, float* __restrict__ out1, GnParamGradArgs pg,
    float* __restrict__ sums_out) {
    const int C = g.C0 + g.C1, cg = C / 32, CW = GB_GPW * cg, Q = CW / 4, PL = g.PL;
    const float rcg = __builtin_amdgcn_rcpf((float)cg);
    const int sidx = blockIdx.x, n = blockIdx.z, cbase = blockIdx.y * CW;
    const int tid = threadIdx.x;
    const bool active = tid < PL * Q;
    const int q = active ? tid % Q : 0, pl = active ? tid / Q : 0;
    const int c = cbase + q * 4;
    const int p_first = sidx * (GBC_KEEP * PL);
    const int npos = min(GBC_KEEP * PL, g.P - p_first);
    const size_t pos0 = (size_t)n * g.P + p_first;
    __shared__ float comb[2][GB_THREADS];
    __shared__ float chS[2][GB_GPW * 32];
    __shared__ float gS1[GB_GPW], gS2[GB_GPW];
    GnBwdChunkRegs r;
    gn_bwd_chunk_load(da, s0, s1p, g, pos0, npos, pl, c, r);         // in flight during the combination
    {   // partial sums of this slice's channels, chunk order: thread (channel cc, lane sl) takes chunks sl, sl + SL, ...
        const int SL = GB_THREADS / CW;
        const int cc = tid % CW, sl = tid / CW;
        float t1 = 0.f, t2 = 0.f;
        if (sl < SL) {
            const float* pp = part_in + ((size_t)n * g.S * C + cbase + cc) * 2;
            for (int s = sl; s < g.S; s += SL) {
                const float2 v = *reinterpret_cast<const float2*>(pp + (size_t)s * C * 2);
                t1 += v.x;
                t2 += v.y;
            }
            comb[0][sl * CW + cc] = t1;
            comb[1][sl * CW + cc] = t2;
        }
        __syncthreads();
        if (tid < CW) {
            t1 = 0.f; t2 = 0.f;
            for (int i = 0; i < SL; ++i) { t1 += comb[0][i * CW + tid]; t2 += comb[1][i * CW + tid]; }
            chS[0][tid] = t1;
            chS[1][tid] = t2;
            if (sidx == 0) {        // one workgroup per (sample, slice) delivers the per-channel sums
                const int ch = cbase + tid;
                if (sums_out != nullptr) {
                    sums_out[((size_t)n * C + ch) * 2 + 0] = t1;
                    sums_out[((size_t)n * C + ch) * 2 + 1] = t2;
                }
                if (pg.dgamma != nullptr) {
                    float sc1 = 1.0f;
                    if (pg.film != nullptr) {
                        const int b = n / pg.T;
                        sc1 += pg.film[(size_t)b * pg.film_ld + ch];
                        atomicAdd(pg.dfilm + (size_t)b * pg.dfilm_ld + ch, t2 * pg.gamma[ch] + t1 * pg.beta[ch]);
                        atomicAdd(pg.dfilm + (size_t)b * pg.dfilm_ld + C + ch, t1);
                    }
                    atomicAdd(pg.dgamma + ch, t2 * sc1);
                    atomicAdd(pg.dbeta + ch, t1 * sc1);
                }
            }
        }
        __syncthreads();
    }
    if (tid < GB_GPW) {
        const int gi = blockIdx.y * GB_GPW + tid;
        const float rstd = stats[((size_t)n * 32 + gi) * 2 + 1];
        float S1 = 0.f, S2 = 0.f;
        for (int i = 0; i < cg; ++i) {
            const float gp = coefA[(size_t)n * C + gi * cg + i] / rstd;   // g' = A / rstd (rstd > 0)
            S1 += gp * chS[0][tid * cg + i];
            S2 += gp * chS[1][tid * cg + i];
        }
        const float inv = 1.0f / ((float)cg * (float)g.P);
        gS1[tid] = S1 * inv;
        gS2[tid] = S2 * inv;
    }
    __syncthreads();
    if (!active) return;
    const f32x4 A = ld4(coefA + (size_t)n * C + c), B = ld4(coefB + (size_t)n * C + c);
    f32x4 mu, rs, S1, S2;
    const float* st = stats + (size_t)n * 64;
#define LFVDM_G(k, f)                                                                                \
    { const int gq = fdiv_small(q * 4 + k, cg, rcg); const int gg = blockIdx.y * GB_GPW + gq;        \
      mu.f = st[2 * gg]; rs.f = st[2 * gg + 1]; S1.f = gS1[gq]; S2.f = gS2[gq]; }
    LFVDM_G(0, x) LFVDM_G(1, y) LFVDM_G(2, z) LFVDM_G(3, w)
#undef LFVDM_G
    const bool silu = act == LFVDM_ACT_SILU;
    const bool first = c < g.C0;
    float* out = first ? out0 : out1;
    const int Cd = first ? g.C0 : g.C1;
    const int cd = first ? c : c - g.C0;
#pragma unroll
    for (int i = 0; i < GBC_KEEP; ++i) {
        const int p = pl + i * PL;
        if (p < npos) {
            f32x4 dz = r.dz[i];
            if (silu) {
                const f32x4 z = r.x[i] * A + B;
                dz.x *= dsilu(z.x); dz.y *= dsilu(z.y); dz.z *= dsilu(z.z); dz.w *= dsilu(z.w);
            }
            const f32x4 xh = (r.x[i] - mu) * rs;
            f32x4 dx = A * dz - rs * (S1 + xh * S2);
            if (pg.add != nullptr) dx += ld4(pg.add + (pos0 + p) * pg.add_ld + c);
            if (pg.add2 != nullptr) dx += ld4(pg.add2 + (pos0 + p) * pg.add2_ld + c);
            st4(out + (pos0 + p) * Cd + cd, dx);
        }
    }
}

// chunks per (sample, 8 groups) slice; 0 = the slice is small enough for the single-workgroup kernels
inline int gn_bwd_chunks(int C, int P, int* pl_out) {
    const int Q = GB_GPW * (C / 32) / 4;
    const int PL = GB_THREADS / Q;
    if (pl_out) *pl_out = PL;
    if (P <= 2 * GBC_KEEP * PL) return 0;
    return (P + GBC_KEEP * PL - 1) / (GBC_KEEP * PL);
}

// temporal GroupNorm backward (rpe.py:135-137): sample = (b, pixel), elements [T][C/32-group]; one wave per sample.
// y = (x-mean)*rstd*gamma + beta.  Writes dx and accumulates dgamma/dbeta with atomics.
constexpr int GTB_MAXC = 512;
__global__ __launch_bounds__(256) void gn_temporal_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              const float* __restrict__ gamma, float eps,
                                                              float* __restrict__ dx, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int B, int T, int P, int C,
                                                              int accumulate, float* __restrict__ det_slab) {
    __shared__ float ch_all[4][3][GTB_MAXC];
    __shared__ float gst_all[4][4][32];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long sample_raw = (long)blockIdx.x * 4 + wave;
    const bool live = sample_raw < (long)B * P;        // (no early return: the workgroup meets at a barrier below)
    const long sample = live ? sample_raw : 0;
    float* chx = ch_all[wave][0];   // per-channel sums (reused per pass)
    float* chd = ch_all[wave][1];
    float* chdx = ch_all[wave][2];
    float* gmean = gst_all[wave][0];
    float* grstd = gst_all[wave][1];
    float* gS1 = gst_all[wave][2];
    float* gS2 = gst_all[wave][3];
    const int b = (int)(sample / P), p = (int)(sample % P);
    const int cg = C / 32;
    const float rcg = __builtin_amdgcn_rcpf((float)cg);
    const int Q = C / 4;
    const size_t base = ((size_t)b * T * P + p) * C;
    const size_t tstride = (size_t)P * C;
    const int TL = (Q <= 64 && 64 % Q == 0) ? 64 / Q : 1;
    const int QL = 64 / TL;
    const int ql = lane % QL, tl = lane / QL;

    // statistics (two passes) exactly as in the forward
    for (int pass = 0; pass < 2; ++pass) {
        for (int q = ql; q < Q; q += QL) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, mu = {0.f, 0.f, 0.f, 0.f};
            if (pass) { mu.x = gmean[fdiv_small(q * 4, cg, rcg)]; mu.y = gmean[fdiv_small(q * 4 + 1, cg, rcg)]; mu.z = gmean[fdiv_small(q * 4 + 2, cg, rcg)]; mu.w = gmean[fdiv_small(q * 4 + 3, cg, rcg)]; }
            for (int t = tl; t < T; t += TL) {
                f32x4 v = ld4(x + base + t * tstride + q * 4);
                if (pass) { v = v - mu; s += v * v; } else { s += v; }
            }
            for (int o = QL; o < 64; o <<= 1) {
                s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64); s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
            }
            if (tl == 0) st4(chx + q * 4, s);
        }
        wave_lds_fence();
        if (lane < 32) {
            float t = 0.f;
            for (int i = 0; i < cg; ++i) t += chx[lane * cg + i];
            const float inv = 1.0f / (float)(cg * T);
            if (pass == 0) gmean[lane] = t * inv; else grstd[lane] = 1.0f / sqrtf(t * inv + eps);
        }
        wave_lds_fence();
    }
    // per-channel sums of dy and dy*xhat
    for (int q = ql; q < Q; q += QL) {
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, mu, rs;
        mu.x = gmean[fdiv_small(q * 4, cg, rcg)]; mu.y = gmean[fdiv_small(q * 4 + 1, cg, rcg)]; mu.z = gmean[fdiv_small(q * 4 + 2, cg, rcg)]; mu.w = gmean[fdiv_small(q * 4 + 3, cg, rcg)];
        rs.x = grstd[fdiv_small(q * 4, cg, rcg)]; rs.y = grstd[fdiv_small(q * 4 + 1, cg, rcg)]; rs.z = grstd[fdiv_small(q * 4 + 2, cg, rcg)]; rs.w = grstd[fdiv_small(q * 4 + 3, cg, rcg)];
        for (int t = tl; t < T; t += TL) {
            const f32x4 xv = ld4(x + base + t * tstride + q * 4);
            const f32x4 d = ld4(dy + base + t * tstride + q * 4);
            s1 += d;
            s2 += d * ((xv - mu) * rs);
        }
        for (int o = QL; o < 64; o <<= 1) {
            s1.x += __shfl_xor(s1.x, o, 64); s1.y += __shfl_xor(s1.y, o, 64); s1.z += __shfl_xor(s1.z, o, 64); s1.w += __shfl_xor(s1.w, o, 64);
            s2.x += __shfl_xor(s2.x, o, 64); s2.y += __shfl_xor(s2.y, o, 64); s2.z += __shfl_xor(s2.z, o, 64); s2.w += __shfl_xor(s2.w, o, 64);
        }
        if (tl == 0) { st4(chd + q * 4, s1); st4(chdx + q * 4, s2); }
    }
    wave_lds_fence();
    if (lane < 32) {
        float S1 = 0.f, S2 = 0.f;
        for (int i = 0; i < cg; ++i) { const int ch = lane * cg + i; S1 += gamma[ch] * chd[ch]; S2 += gamma[ch] * chdx[ch]; }
        const float inv = 1.0f / (float)(cg * T);
        gS1[lane] = S1 * inv; gS2[lane] = S2 * inv;
    }
    // parameter gradients: the four samples of the workgroup are summed first, then one atomic per channel
    // (all B*P samples hit the same C addresses: fewer, fatter atomics)
    __syncthreads();
    for (int ch = threadIdx.x; ch < C; ch += 256) {
        float g = 0.f, bsum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if ((long)blockIdx.x * 4 + w < (long)B * P) {
                g += ch_all[w][2][ch];
                bsum += ch_all[w][1][ch];
            }
        }
        // (deterministic mode: this workgroup's row of the slab instead of atomics; every workgroup writes all C channels)
        const DetSlab dsG = {det_slab, dgamma, (long)C};
        const DetSlab dsB = {det_slab ? det_slab + (size_t)gridDim.x * C : nullptr, dbeta, (long)C};
        det_add(dsG, blockIdx.x, dgamma + ch, g);
        det_add(dsB, blockIdx.x, dbeta + ch, bsum);
    }
    if (!live) return;
    const int E = T * Q;
    for (int e = lane; e < E; e += 64) {
        const int t = e / Q, q = e - t * Q;
        const f32x4 xv = ld4(x + base + t * tstride + q * 4);
        const f32x4 d = ld4(dy + base + t * tstride + q * 4);
        const f32x4 ga = ld4(gamma + q * 4);
        f32x4 o;
#define LFVDM_T(k, f)                                                                                   \
    { const int g = fdiv_small(q * 4 + k, cg, rcg); const float xh = (xv.f - gmean[g]) * grstd[g];                    \
      o.f = grstd[g] * (ga.f * d.f - (gS1[g] + xh * gS2[g])); }
        LFVDM_T(0, x) LFVDM_T(1, y) LFVDM_T(2, z) LFVDM_T(3, w)
#undef LFVDM_T
        float* dst = dx + base + t * tstride + q * 4;
        if (accumulate) o += ld4(dst);
        st4(dst, o);
    }
}

// Register-resident variant for C <= 256 (like gn_temporal_reg_kernel of the forward): x and dy of the (b, pixel) sample -
// 2 x T x C floats, <= 2 x TIT float4 per lane - are read ONCE; the kernel above reads x four times and dy twice, each
// pass a dependent round trip through L2 (24 us per launch at the cfg-C shapes, 7 launches per training step).
template <int TIT>
__global__ __launch_bounds__(256) void gn_temporal_bwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  const float* __restrict__ gamma, float eps,
                                                                  float* __restrict__ dx, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, int B, int T, int P, int C,
                                                                  int accumulate, float* __restrict__ det_slab) {
    __shared__ float ch_all[4][3][256];
    __shared__ float gst_all[4][4][32];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long sample_raw = (long)blockIdx.x * 4 + wave;
    const bool live = sample_raw < (long)B * P;        // (no early return: the workgroup meets at a barrier below)
    const long sample = live ? sample_raw : 0;
    float* chx = ch_all[wave][0];
    float* chd = ch_all[wave][1];
    float* chdx = ch_all[wave][2];
    float* gmean = gst_all[wave][0];
    float* grstd = gst_all[wave][1];
    float* gS1 = gst_all[wave][2];
    float* gS2 = gst_all[wave][3];
    const int b = (int)(sample / P), p = (int)(sample % P);
    const int cg = C / 32;
    const float rcg = __builtin_amdgcn_rcpf((float)cg);
    const int Q = C / 4;                 // <= 64, divides 64
    const size_t base = ((size_t)b * T * P + p) * C;
    const size_t tstride = (size_t)P * C;
    const int TL = 64 / Q;
    const int q = lane % Q, tl = lane / Q;
    f32x4 kx[TIT], kd[TIT];
#pragma unroll
    for (int i = 0; i < TIT; ++i) {
        const int t = tl + i * TL;
        const bool ok = t < T;
        kx[i] = ok ? ld4(x + base + t * tstride + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        kd[i] = ok ? ld4(dy + base + t * tstride + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const f32x4 ga = ld4(gamma + q * 4);
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {         // statistics exactly as in the forward
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TIT; ++i) {
            if (tl + i * TL < T) {
                if (pass) { const f32x4 v = kx[i] - mu; s += v * v; } else { s += kx[i]; }
            }
        }
        for (int o = Q; o < 64; o <<= 1) {
            s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64);
            s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
        }
        if (tl == 0) st4(chx + q * 4, s);
        wave_lds_fence();
        if (lane < 32) {
            float t = 0.f;
            for (int i = 0; i < cg; ++i) t += chx[lane * cg + i];
            const float inv = 1.0f / (float)(cg * T);
            if (pass == 0) gmean[lane] = t * inv; else grstd[lane] = 1.0f / sqrtf(t * inv + eps);
        }
        wave_lds_fence();
        if (pass == 0) {
            mu.x = gmean[fdiv_small(q * 4 + 0, cg, rcg)]; mu.y = gmean[fdiv_small(q * 4 + 1, cg, rcg)];
            mu.z = gmean[fdiv_small(q * 4 + 2, cg, rcg)]; mu.w = gmean[fdiv_small(q * 4 + 3, cg, rcg)];
        } else {
            rs.x = grstd[fdiv_small(q * 4 + 0, cg, rcg)]; rs.y = grstd[fdiv_small(q * 4 + 1, cg, rcg)];
            rs.z = grstd[fdiv_small(q * 4 + 2, cg, rcg)]; rs.w = grstd[fdiv_small(q * 4 + 3, cg, rcg)];
        }
    }
    // per-channel sums of dy and dy * xhat (xhat replaces x in the registers)
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < TIT; ++i) {
        kx[i] = (kx[i] - mu) * rs;
        if (tl + i * TL < T) {
            s1 += kd[i];
            s2 += kd[i] * kx[i];
        }
    }
    for (int o = Q; o < 64; o <<= 1) {
        s1.x += __shfl_xor(s1.x, o, 64); s1.y += __shfl_xor(s1.y, o, 64); s1.z += __shfl_xor(s1.z, o, 64); s1.w += __shfl_xor(s1.w, o, 64);
        s2.x += __shfl_xor(s2.x, o, 64); s2.y += __shfl_xor(s2.y, o, 64); s2.z += __shfl_xor(s2.z, o, 64); s2.w += __shfl_xor(s2.w, o, 64);
    }
    if (tl == 0) { st4(chd + q * 4, s1); st4(chdx + q * 4, s2); }
    wave_lds_fence();
    if (lane < 32) {
        float S1 = 0.f, S2 = 0.f;
        for (int i = 0; i < cg; ++i) { const int ch = lane * cg + i; S1 += gamma[ch] * chd[ch]; S2 += gamma[ch] * chdx[ch]; }
        const float inv = 1.0f / (float)(cg * T);
        gS1[lane] = S1 * inv; gS2[lane] = S2 * inv;
    }
    // parameter gradients: the four samples of the workgroup are summed first, then one atomic per channel
    __syncthreads();
    for (int ch = threadIdx.x; ch < C; ch += 256) {
        float g = 0.f, bsum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if ((long)blockIdx.x * 4 + w < (long)B * P) {
                g += ch_all[w][2][ch];
                bsum += ch_all[w][1][ch];
            }
        }
        // (deterministic mode: this workgroup's row of the slab instead of atomics; every workgroup writes all C channels)
        const DetSlab dsG = {det_slab, dgamma, (long)C};
        const DetSlab dsB = {det_slab ? det_slab + (size_t)gridDim.x * C : nullptr, dbeta, (long)C};
        det_add(dsG, blockIdx.x, dgamma + ch, g);
        det_add(dsB, blockIdx.x, dbeta + ch, bsum);
    }
    if (!live) return;
    f32x4 S1, S2;
    S1.x = gS1[fdiv_small(q * 4 + 0, cg, rcg)]; S1.y = gS1[fdiv_small(q * 4 + 1, cg, rcg)]; S1.z = gS1[fdiv_small(q * 4 + 2, cg, rcg)]; S1.w = gS1[fdiv_small(q * 4 + 3, cg, rcg)];
    S2.x = gS2[fdiv_small(q * 4 + 0, cg, rcg)]; S2.y = gS2[fdiv_small(q * 4 + 1, cg, rcg)]; S2.z = gS2[fdiv_small(q * 4 + 2, cg, rcg)]; S2.w = gS2[fdiv_small(q * 4 + 3, cg, rcg)];
#pragma unroll
    for (int i = 0; i < TIT; ++i) {
        const int t = tl + i * TL;
        if (t < T) {
            f32x4 o = rs * (ga * kd[i] - (S1 + kx[i] * S2));
            float* dst = dx + base + t * tstride + q * 4;
            if (accumulate) o += ld4(dst);
            st4(dst, o);
        }
    }
}

// Parameter gradients of a (FiLM-modulated) GroupNorm from the per-(sample, channel) sums of gn_bwd_stats
// (sums[n][c] = {sum dz, sum dz*xhat}).  Block = 64 channels x 4 sample lanes, fixed summation order (deterministic):
//   dgamma[c] += sum_n s2 * (1 + scale[n/T][c]);  dbeta[c] += sum_n s1 * (1 + scale);
//   dfilm[b][c] = sum_t (s2 * gamma[c] + s1 * beta[c]);  dfilm[b][C + c] = sum_t s1        (FiLM only)
__global__ __launch_bounds__(256) void gn_param_grads_kernel(const float* __restrict__ sums, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ film,
                                                             int film_ld, int T, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ dfilm, int dfilm_ld,
                                                             int N, int C) {
    __shared__ float red[4][64][4];
    const int cl = threadIdx.x & 63, nl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool ok = c < C;
    float dg = 0.f, db = 0.f;
    if (film) {
        const float g = ok ? gamma[c] : 0.f, be = ok ? beta[c] : 0.f;
        const int B = N / T;
        for (int b = 0; b < B; ++b) {
            const float sc1 = ok ? 1.0f + film[(size_t)b * film_ld + c] : 0.f;
            float dsc = 0.f, dsh = 0.f, g1 = 0.f, b1 = 0.f;
            if (ok)
                for (int t = nl; t < T; t += 4) {
                    const float* s = sums + ((size_t)(b * T + t) * C + c) * 2;
                    const float s1 = s[0], s2 = s[1];
                    g1 += s2;
                    b1 += s1;
                    dsc += s2 * g + s1 * be;
                    dsh += s1;
                }
            red[nl][cl][0] = dsc; red[nl][cl][1] = dsh; red[nl][cl][2] = g1; red[nl][cl][3] = b1;
            __syncthreads();
            if (nl == 0 && ok) {
                float t4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) t4[k] = red[0][cl][k] + red[1][cl][k] + red[2][cl][k] + red[3][cl][k];
                dfilm[(size_t)b * dfilm_ld + c] = t4[0];
                dfilm[(size_t)b * dfilm_ld + C + c] = t4[1];
                dg += t4[2] * sc1;
                db += t4[3] * sc1;
            }
            __syncthreads();
        }
    } else {
        float g1 = 0.f, b1 = 0.f;
        if (ok)
            for (int n = nl; n < N; n += 4) {
                const float* s = sums + ((size_t)n * C + c) * 2;
                b1 += s[0];
                g1 += s[1];
            }
        red[nl][cl][0] = g1; red[nl][cl][1] = b1;
        __syncthreads();
        if (nl == 0) {
            dg = red[0][cl][0] + red[1][cl][0] + red[2][cl][0] + red[3][cl][0];
            db = red[0][cl][1] + red[1][cl][1] + red[2][cl][1] + red[3][cl][1];
        }
    }
    if (nl == 0 && ok) {
        dgamma[c] += dg;
        dbeta[c] += db;
    }
}

}  // namespace

extern "C" int lfvdm_gn_bwd_stats(const float* da, const float* src0, const float* src1, int C0, int C1, int N, int P,
                                  const float* coefA, const float* coefB, const float* stats, int act, float* sums,
                                  void* stream) {
    const int C = C0 + C1;
    if (N <= 0 || P <= 0 || C <= 0 || C % 32 || C0 % 4 || C > 1024) return LFVDM_E_SHAPE;
    hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(N, 32 / GB_GPW), dim3(GB_THREADS), 0, (hipStream_t)stream, da, src0, src1,
                       C0, C1, P, coefA, coefB, stats, act, sums);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_gn_bwd_apply(const float* da, const float* src0, const float* src1, int C0, int C1, int N, int P,
                                  const float* coefA, const float* coefB, const float* stats, const float* sums, int act,
                                  float* out0, float* out1, int acc0, int acc1, void* stream) {
    const int C = C0 + C1;
    if (N <= 0 || P <= 0 || C <= 0 || C % 32 || C0 % 4 || C > 1024) return LFVDM_E_SHAPE;
    if (C1 > 0 && (!src1 || !out1)) return LFVDM_E_SHAPE;
    const GnParamGradArgs none = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 1, nullptr, 0, nullptr, 0};
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(N, 32 / GB_GPW), dim3(GB_THREADS), 0, (hipStream_t)stream, da, src0, src1,
                       C0, C1, P, coefA, coefB, stats, sums, act, out0, out1, acc0, acc1, none);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_gn_bwd_apply_params(const float* da, const float* src0, const float* src1, int C0, int C1, int N, int P,
                                         const float* coefA, const float* coefB, const float* stats, const float* sums,
                                         int act, float* out0, float* out1, int acc0, int acc1, const float* gamma,
                                         const float* beta, const float* film, int film_ld, int T, float* dgamma,
                                         float* dbeta, float* dfilm, int dfilm_ld, const float* add, int add_ld,
                                         void* stream) {
    const int C = C0 + C1;
    if (N <= 0 || P <= 0 || C <= 0 || C % 32 || C0 % 4 || C > 1024) return LFVDM_E_SHAPE;
    if (C1 > 0 && (!src1 || !out1)) return LFVDM_E_SHAPE;
    if (!dgamma || !dbeta) return LFVDM_E_SHAPE;
    if (add && (add_ld < C || add_ld % 4)) return LFVDM_E_SHAPE;
    if (film && (!gamma || !beta || !dfilm || T <= 0 || N % T || film_ld < 2 * C || dfilm_ld < 2 * C)) return LFVDM_E_SHAPE;
    const GnParamGradArgs pg = {gamma, beta, film, dgamma, dbeta, dfilm, film_ld, dfilm_ld, T > 0 ? T : 1, add, add_ld, nullptr, 0};
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(N, 32 / GB_GPW), dim3(GB_THREADS), 0, (hipStream_t)stream, da, src0, src1,
                       C0, C1, P, coefA, coefB, stats, sums, act, out0, out1, acc0, acc1, pg);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_gn_bwd_fused(const float* da, const float* src0, const float* src1, int C0, int C1, int N, int P,
                                  const float* coefA, const float* coefB, const float* stats, int act, float* out0,
                                  float* out1, const float* gamma, const float* beta, const float* film, int film_ld, int T,
                                  float* dgamma, float* dbeta, float* dfilm, int dfilm_ld, const float* add, int add_ld,
                                  const float* add2, int add2_ld, void* stream) {
    const int C = C0 + C1;
    if (N <= 0 || P <= 0 || C <= 0 || C % 32 || C0 % 4 || C > 1024) return LFVDM_E_SHAPE;
    if (C1 > 0 && (!src1 || !out1)) return LFVDM_E_SHAPE;
    if (!dgamma || !dbeta) return LFVDM_E_SHAPE;
    if (add && (add_ld < C || add_ld % 4)) return LFVDM_E_SHAPE;
    if (add2 && (add2_ld < C || add2_ld % 4)) return LFVDM_E_SHAPE;
    if (film && (!gamma || !beta || !dfilm || T <= 0 || N % T || film_ld < 2 * C || dfilm_ld < 2 * C)) return LFVDM_E_SHAPE;
    const GnParamGradArgs pg = {gamma, beta, film, dgamma, dbeta, dfilm, film_ld, dfilm_ld, T > 0 ? T : 1, add, add_ld, add2, add2_ld};
    hipLaunchKernelGGL(gn_bwd_fused_kernel, dim3(N, 32 / GB_GPW), dim3(GB_THREADS), 0, (hipStream_t)stream, da, src0, src1,
                       C0, C1, P, coefA, coefB, stats, act, out0, out1, pg, (float*)nullptr);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_gn_bwd_fused_sums(const float* da, const float* src0, const float* src1, int C0, int C1, int N, int P,
                                       const float* coefA, const float* coefB, const float* stats, int act, float* out0,
                                       float* out1, const float* add, int add_ld, const float* add2, int add2_ld, float* sums,
                                       void* stream) {
    const int C = C0 + C1;
    if (N <= 0 || P <= 0 || C <= 0 || C % 32 || C0 % 4 || C > 1024 || !sums) return LFVDM_E_SHAPE;
    if (C1 > 0 && (!src1 || !out1)) return LFVDM_E_SHAPE;
    if (add && (add_ld < C || add_ld % 4)) return LFVDM_E_SHAPE;
    if (add2 && (add2_ld < C || add2_ld % 4)) return LFVDM_E_SHAPE;
    const GnParamGradArgs pg = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 1, add, add_ld, add2, add2_ld};
    hipLaunchKernelGGL(gn_bwd_fused_kernel, dim3(N, 32 / GB_GPW), dim3(GB_THREADS), 0, (hipStream_t)stream, da, src0, src1,
                       C0, C1, P, coefA, coefB, stats, act, out0, out1, pg, sums);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" long lfvdm_gn_bwd_ws_floats(int C, int N, int P) {
    if (C <= 0 || C % 32 || C > 1024 || N <= 0 || P <= 0) return 0;
    static const bool off = getenv("LFVDM_GN_BWD_NO_CHUNKS") != nullptr;       // A/B aid
    if (off) return 0;
    const int S = gn_bwd_chunks(C, P, nullptr);
    return (long)N * S * C * 2;
}

extern "C" int lfvdm_gn_bwd_ws(const float* da, const float* src0, const float* src1, int C0, int C1, int N, int P,
                               const float* coefA, const float* coefB, const float* stats, int act, float* out0,
                               float* out1, const float* gamma, const float* beta, const float* film, int film_ld, int T,
                               float* dgamma, float* dbeta, float* dfilm, int dfilm_ld, const float* add, int add_ld,
                               const float* add2, int add2_ld, float* sums_out, float* ws, long ws_floats, void* stream) {
    const int C = C0 + C1;
    if (N <= 0 || P <= 0 || C <= 0 || C % 32 || C0 % 4 || C > 1024 || N > 65535) return LFVDM_E_SHAPE;
    if (C1 > 0 && (!src1 || !out1)) return LFVDM_E_SHAPE;
    if (!da || !src0 || !out0 || !coefA || !coefB || !stats) return LFVDM_E_SHAPE;
    if ((dgamma == nullptr) != (dbeta == nullptr)) return LFVDM_E_SHAPE;
    if (!dgamma && !sums_out) return LFVDM_E_SHAPE;          // the per-channel sums must go somewhere
    if (add && (add_ld < C || add_ld % 4)) return LFVDM_E_SHAPE;
    if (add2 && (add2_ld < C || add2_ld % 4)) return LFVDM_E_SHAPE;
    if (dgamma && film && (!gamma || !beta || !dfilm || T <= 0 || N % T || film_ld < 2 * C || dfilm_ld < 2 * C)) return LFVDM_E_SHAPE;
    int PL = 0;
    const int S = gn_bwd_chunks(C, P, &PL);
    if (S == 0 || !ws || ws_floats < (long)N * S * C * 2) return LFVDM_E_SHAPE;
    const GnBwdChunkGeom g = {C0, C1, P, S, PL};
    const GnParamGradArgs pg = {gamma, beta, dgamma ? film : nullptr, dgamma, dbeta, dfilm, film_ld, dfilm_ld, T > 0 ? T : 1,
                                add, add_ld, add2, add2_ld};
    const dim3 grid(S, 32 / GB_GPW, N);
    hipLaunchKernelGGL(gn_bwd_chunk_stats_kernel, grid, dim3(GB_THREADS), 0, (hipStream_t)stream, da, src0, src1, g, coefA, coefB,
                       stats, act, ws);
    LFVDM_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_bwd_chunk_apply_kernel, grid, dim3(GB_THREADS), 0, (hipStream_t)stream, da, src0, src1, g, coefA, coefB,
                       stats, act, (const float*)ws, out0, out1, pg, sums_out);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

static int gn_temporal_bwd_impl(const float* x, const float* dy, const float* gamma, float eps, float* dx, float* dgamma,
                               float* dbeta, int B, int T, int P, int C, int accumulate, float* det_ws, long det_ws_floats,
                               hipStream_t s) {
    if (B <= 0 || T <= 0 || P <= 0 || C % 32 || C > GTB_MAXC) return LFVDM_E_SHAPE;
    const long samples = (long)B * P;
    const int Q = C / 4;
    const long nwg = (samples + 3) / 4;
    if (det_ws && det_ws_floats < 2 * nwg * C) return LFVDM_E_SHAPE;
    const dim3 grid((unsigned)nwg);
    static const bool no_reg = getenv("LFVDM_GNT_BWD_NO_REG") != nullptr;        // A/B aid
    bool launched = false;
    if (!no_reg && Q <= 64 && 64 % Q == 0) {       // register-resident sample: frames per lane = ceil(T / (64 / Q))
        const int per_lane = (T + 64 / Q - 1) / (64 / Q);
#define LFVDM_GNTB(N)                                                                                                  \
        if (!launched && per_lane <= N) {                                                                              \
            hipLaunchKernelGGL(gn_temporal_bwd_reg_kernel<N>, grid, dim3(256), 0, s, x, dy, gamma, eps, dx, dgamma, dbeta, B, T, \
                               P, C, accumulate, det_ws);                                                              \
            launched = true;                                                                                           \
        }
        LFVDM_GNTB(8) LFVDM_GNTB(16) LFVDM_GNTB(32)
#undef LFVDM_GNTB
    }
    if (!launched)
        hipLaunchKernelGGL(gn_temporal_bwd_kernel, grid, dim3(256), 0, s, x, dy, gamma, eps, dx, dgamma, dbeta, B, T, P, C,
                           accumulate, det_ws);
    LFVDM_CHECK_LAUNCH();
    if (det_ws) {       // ordered sum of the workgroups' partial parameter gradients
        return lfvdm_det_reduce2_launch(dgamma, det_ws, C, dbeta, det_ws + (size_t)nwg * C, C, nwg, s);
    }
    return LFVDM_OK;
}

extern "C" int lfvdm_gn_temporal_bwd(const float* x, const float* dy, const float* gamma, float eps, float* dx,
                                     float* dgamma, float* dbeta, int B, int T, int P, int C, int accumulate, void* stream) {
    return gn_temporal_bwd_impl(x, dy, gamma, eps, dx, dgamma, dbeta, B, T, P, C, accumulate, nullptr, 0, (hipStream_t)stream);
}

extern "C" int lfvdm_gn_temporal_bwd_det(const float* x, const float* dy, const float* gamma, float eps, float* dx,
                                         float* dgamma, float* dbeta, int B, int T, int P, int C, int accumulate, float* det_ws,
                                         int64_t det_ws_floats, void* stream) {
    if (!det_ws) return LFVDM_E_SHAPE;
    return gn_temporal_bwd_impl(x, dy, gamma, eps, dx, dgamma, dbeta, B, T, P, C, accumulate, det_ws, (long)det_ws_floats,
                                (hipStream_t)stream);
}

extern "C" int lfvdm_gn_param_grads(const float* sums, const float* gamma, const float* beta, const float* film, int film_ld,
                                    int T, float* dgamma, float* dbeta, float* dfilm, int dfilm_ld, int N, int C, void* stream) {
    if (!sums || !dgamma || !dbeta || N <= 0 || C <= 0) return LFVDM_E_SHAPE;
    if (film && (!gamma || !beta || !dfilm || T <= 0 || N % T || film_ld < 2 * C || dfilm_ld < 2 * C)) return LFVDM_E_SHAPE;
    hipLaunchKernelGGL(gn_param_grads_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, sums, gamma, beta, film,
                       film_ld, T, dgamma, dbeta, dfilm, dfilm_ld, N, C);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}
