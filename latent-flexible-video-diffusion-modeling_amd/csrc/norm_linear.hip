// GroupNorm32 + Linear in one launch: the head of both attention halves of a FactorizedAttentionBlock
// (reference rpe.py:133-141: x = norm(x); qkv = self.qkv(x)), for the sampler's forward plan.
//
//   xn = GN32(x)            (written: it is also the residual of the block, rpe.py:172)
//   y  = xn W^T + bias      (W [Nout][C]: the nn.Linear qkv weight)
//
// Normalisation units (the reference reshapes so that GroupNorm sees (C, tokens) per unit):
//   mode 0, temporal instance (unet.py:218-221): unit = (b, pixel), statistics over the T frames x C/32 channels of a group;
//   mode 1, spatial instance  (unet.py:214-217): unit = frame,      statistics over the P pixels x C/32 channels.
// A workgroup owns whole units: a strip of S pixels with all T frames (temporal) or one frame (spatial), i.e. R <= 256
// rows of C channels resident in LDS: two exact passes for the statistics, normalisation in place, then the rows are the
// A operand of a 32x32x2 fp32 MFMA GEMM against a 64-column slice of W (blockIdx.y).  Every column slice recomputes the
// (cheap) normalisation; slice 0 writes xn.  Replaces two dependent launches (lfvdm_gn_temporal / lfvdm_gn_apply, then
// the 1x1 implicit GEMM) and the round trip of xn between them.
#include "common_hip.h"

namespace {

constexpr int NL_NS = 64;          // output columns per workgroup

struct NLGeom {
    int mode, T, P, C, Nout;
    int S, logS, strips;           // temporal: pixels per strip (power of two), strips per batch element
    int R, Rp;                     // rows of the block, padded to 32
    float eps;
};

template <int C>
__global__ __launch_bounds__(256) void norm_linear_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ xn,
                                                          float* __restrict__ y, NLGeom g) {
    constexpr int LD = C + 4;                      // LDS row stride (floats): b128 fragment reads 4 banks apart per row
    constexpr int Q = C / 4, CG = C / 32;
    extern __shared__ __attribute__((aligned(16))) float nl_smem[];
    float* Xs = nl_smem;                           // [Rp][LD]
    float* Ws = Xs + (size_t)g.Rp * LD;            // [64][LD]
    float* st_mu = Ws + (size_t)NL_NS * LD;        // [units * 32]
    float* st_rs = st_mu + 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * NL_NS;
    // ---- rows of this block
    int b = 0, p0 = 0, Seff = 0;
    size_t row_base = 0;                           // spatial: first global row of the frame
    if (g.mode == 0) {
        b = blockIdx.x / g.strips;
        p0 = (blockIdx.x - b * g.strips) * g.S;
        Seff = min(g.S, g.P - p0);
    } else {
        row_base = (size_t)blockIdx.x * g.P;
    }
    auto global_row = [&](int r) -> long {         // -1: padding row
        if (g.mode == 0) {
            const int t = r >> g.logS, s = r & (g.S - 1);
            return (t < g.T && s < Seff) ? (long)((size_t)(b * g.T + t) * g.P + p0 + s) : -1;
        }
        return r < g.P ? (long)(row_base + r) : -1;
    };
    // ---- stage x rows and the W slice
    for (int idx = tid; idx < g.Rp * Q; idx += 256) {
        const int r = idx / Q, q = idx - r * Q;
        const long m = global_row(r);
        const f32x4 v = m >= 0 ? ld4(x + (size_t)m * C + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
        st4(Xs + r * LD + 4 * q, v);
    }
    for (int idx = tid; idx < NL_NS * Q; idx += 256) {
        const int n = idx / Q, q = idx - n * Q;
        const f32x4 v = n0 + n < g.Nout ? ld4(W + (size_t)(n0 + n) * C + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
        st4(Ws + n * LD + 4 * q, v);
    }
    __syncthreads();
    // ---- statistics: mean, then centred variance (exact two-pass, like nn.GroupNorm)
    if (g.mode == 0) {
        const int s = tid >> 5, grp = tid & 31;                 // unit (pixel s, group grp): up to 8 x 32 threads
        if (s < Seff) {
            const float inv = 1.0f / (float)(g.T * CG);
            float sum = 0.f;
            for (int t = 0; t < g.T; ++t)
#pragma unroll
                for (int i = 0; i < CG; ++i) sum += Xs[((t << g.logS) + s) * LD + grp * CG + i];
            const float mu = sum * inv;
            float var = 0.f;
            for (int t = 0; t < g.T; ++t)
#pragma unroll
                for (int i = 0; i < CG; ++i) {
                    const float d = Xs[((t << g.logS) + s) * LD + grp * CG + i] - mu;
                    var += d * d;
                }
            st_mu[tid] = mu;
            st_rs[tid] = 1.0f / sqrtf(var * inv + g.eps);
        }
    } else {
        const int grp = tid >> 3, part = tid & 7;               // 8 threads per group, rows part, part + 8, ...
        const float inv = 1.0f / (float)(g.P * CG);
        float sum = 0.f;
        for (int r = part; r < g.P; r += 8)
#pragma unroll
            for (int i = 0; i < CG; ++i) sum += Xs[r * LD + grp * CG + i];
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        sum += __shfl_xor(sum, 4, 64);
        const float mu = sum * inv;
        float var = 0.f;
        for (int r = part; r < g.P; r += 8)
#pragma unroll
            for (int i = 0; i < CG; ++i) {
                const float d = Xs[r * LD + grp * CG + i] - mu;
                var += d * d;
            }
        var += __shfl_xor(var, 1, 64);
        var += __shfl_xor(var, 2, 64);
        var += __shfl_xor(var, 4, 64);
        if (part == 0) {
            st_mu[grp] = mu;
            st_rs[grp] = 1.0f / sqrtf(var * inv + g.eps);
        }
    }
    __syncthreads();
    // ---- normalise in place (+ write xn once)
    for (int idx = tid; idx < g.R * Q; idx += 256) {
        const int r = idx / Q, q = idx - r * Q;
        const long m = global_row(r);
        if (m < 0) continue;
        const int ubase = g.mode == 0 ? (r & (g.S - 1)) * 32 : 0;
        f32x4 v = ld4(Xs + r * LD + 4 * q);
        const f32x4 ga = ld4(gamma + 4 * q), be = ld4(beta + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int grp = (4 * q + e) / CG;
            v[e] = (v[e] - st_mu[ubase + grp]) * st_rs[ubase + grp] * ga[e] + be[e];
        }
        st4(Xs + r * LD + 4 * q, v);
        if (blockIdx.y == 0 && xn) st4(xn + (size_t)m * C + 4 * q, v);
    }
    __syncthreads();
    // ---- y tile by tile: 32x32 output tiles, K = C in steps of 8 (one b128 per operand, four MFMAs)
    const int TM = g.Rp >> 5;
    const int frag = (lane & 31) * LD + 4 * (lane >> 5);
    for (int id = wave; id < TM * 2; id += 4) {
        const int tm = id >> 1, tn = id & 1;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float* ap = Xs + (size_t)tm * 32 * LD + frag;
        const float* bp = Ws + (size_t)tn * 32 * LD + frag;
#pragma unroll
        for (int k8 = 0; k8 < C / 8; ++k8) {
            const f32x4 a4 = ld4(ap + 8 * k8), b4 = ld4(bp + 8 * k8);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc, 0, 0, 0);
        }
        const int col = n0 + tn * 32 + (lane & 31);
        if (col < g.Nout) {
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                const long m = global_row(r);
                if (m >= 0) y[(size_t)m * g.Nout + col] = acc[i] + bv;
            }
        }
    }
}

template <int C>
int launch_nl(const float* x, const float* gamma, const float* beta, float eps, const float* W, const float* bias, float* xn,
              float* y, int mode, int BN, int T, int P, int Nout, hipStream_t s) {
    NLGeom g{};
    g.mode = mode; g.T = T; g.P = P; g.C = C; g.Nout = Nout; g.eps = eps;
    const int slices = (Nout + NL_NS - 1) / NL_NS;
    const size_t max_rows = ((size_t)160 * 1024 / 4 - 512 - (size_t)NL_NS * (C + 4)) / (C + 4);     // rows that fit the LDS
    unsigned blocks;
    if (mode == 0) {
        // strip of S pixels (power of two, <= 8: 256 statistic threads): the largest that fits and still gives the
        // grid >= 128 workgroups, else the smallest
        int S = 8;
        while (S > 1 && ((size_t)((T * S + 31) & ~31) > max_rows || S > P)) S >>= 1;
        while (S > 1 && (long)BN * ((P + S - 1) / S) * slices < 128) S >>= 1;
        if ((size_t)((T * S + 31) & ~31) > max_rows) return LFVDM_E_UNSUPPORTED;
        g.S = S;
        g.logS = S == 8 ? 3 : S == 4 ? 2 : S == 2 ? 1 : 0;
        g.strips = (P + S - 1) / S;
        g.R = T * S;
        blocks = (unsigned)(BN * g.strips);
    } else {
        if (P > 256 || (size_t)((P + 31) & ~31) > max_rows) return LFVDM_E_UNSUPPORTED;
        g.S = 1; g.logS = 0; g.strips = 1;
        g.R = P;
        blocks = (unsigned)BN;
    }
    g.Rp = (g.R + 31) & ~31;
    const size_t lds = ((size_t)(g.Rp + NL_NS) * (C + 4) + 512) * sizeof(float);
    static DynLdsLimit limit;
    if (int rc = limit.ensure(reinterpret_cast<const void*>(&norm_linear_kernel<C>), lds)) return rc;
    hipLaunchKernelGGL((norm_linear_kernel<C>), dim3(blocks, (unsigned)slices), dim3(256), lds, s, x, gamma, beta, W, bias, xn, y, g);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

}  // namespace

// 1 if lfvdm_norm_linear covers the shape (plans are built without launching anything)
extern "C" int lfvdm_norm_linear_supported(int mode, int T, int P, int C) {
    if (C != 64 && C != 128) return 0;
    const size_t max_rows = ((size_t)160 * 1024 / 4 - 512 - (size_t)NL_NS * (C + 4)) / (C + 4);
    if (mode == 0) return T > 0 && (size_t)((T + 31) & ~31) <= max_rows;
    if (mode == 1) return P > 0 && P <= 256 && (size_t)((P + 31) & ~31) <= max_rows;
    return 0;
}

extern "C" int lfvdm_norm_linear(const float* x, const float* gamma, const float* beta, float eps, const float* W,
                                 const float* bias, float* xn, float* y, int mode, int BN, int T, int P, int C, int Nout,
                                 void* stream) {
    if (!x || !gamma || !beta || !W || !y || BN <= 0 || T <= 0 || P <= 0 || Nout <= 0 || (mode != 0 && mode != 1))
        return LFVDM_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (C == 64) return launch_nl<64>(x, gamma, beta, eps, W, bias, xn, y, mode, BN, T, P, Nout, s);
    if (C == 128) return launch_nl<128>(x, gamma, beta, eps, W, bias, xn, y, mode, BN, T, P, Nout, s);
    return LFVDM_E_UNSUPPORTED;
}
