// Temporal GroupNorm + the qkv projection of the temporal attention in ONE launch (reference rpe.py:136 `x = self.norm(x)`
// and :139 `qkv = self.qkv(x)` of the temporal RPEAttention instance; GroupNorm32(32, C) over (C/32 channels x T frames)
// per (batch, pixel), nn.py:93-101; qkv = nn.Linear(C, 3C)).
//
// Both steps are local to a (batch, pixel) COLUMN of T rows: lfvdm_gn_temporal read the column, normalised it and wrote
// it; the 1x1 GEMM launch read it back (7-11 TFLOP/s: a K = 64 ... 256 product is all ramp-up and tail) - two launches of
// ~5.5 and ~7 us in every one of the seven attention blocks of a cfg-B denoising step.  Here a workgroup owns PX pixels
// (PX * T <= 48 rows) and a 1/NSPLIT share of the 3C output columns:
//   * statistics in registers: a lane holds channel quad `quad` of frames tp, tp + 8, ... of one pixel; the sum over the
//     frames of the other seven lanes and over the channels of a group are DPP steps inside 16 lanes (no LDS, no barrier);
//     two passes (mean, squared deviations) and (x - mean) * rstd * gamma + beta: the arithmetic of lfvdm_gn_temporal;
//   * the normalised rows go to an XOR-swizzled LDS image (the A operand) and - from the share-0 workgroup - to `xn_out`
//     (the residual of the block's output projection, rpe.py:172);
//   * fp32 MFMA 16x16x4: wave w takes column tiles w, w + 4, ...; the filter rows of a tile are read from global memory
//     straight into B fragments (a lane reads 16 bytes per 16-wide K group: W is [3C][C] row-major = nn.Linear's weight),
//     the NEXT tile's fragments are in flight while the current tile is multiplied, the first tile's while the statistics run;
//   * the splits of a column strip sit on one XCD (blockIdx % 8) and share its x rows in that L2.
// MFMA 16x16x4 computes the TRANSPOSED tile, D^T = W . xn^T: A lane (filter row i = l & 15, k = l >> 4), B lane (k = l >> 4,
// token j = l & 15), D lane: token j = l & 15, filters 4 (l >> 4) + r - four consecutive output columns of one row per lane;
// inside a 16-wide K group, step e of lane group kk uses k = 16 g + 4 kk + e on both operands.
#include <cstdlib>
#include <type_traits>

#include "common_hip.h"

namespace {

constexpr int TQ_TF = 3;          // frames per lane: T <= 24

struct TqGeom {
    int B, T, P;
    int NSPLIT, NCT;              // shares of the 3C columns, 16-wide column tiles per share (a multiple of 4)
    int ngroups;                  // column strips: B * P / PX
};

template <int C_>
struct TqCfg {
    static constexpr int SPR = C_ / 4;                    // 16-byte slots per row
    static constexpr int NSL = SPR > 32 ? SPR / 32 : 1;   // (pixel, quad) slots per thread: 32 slots per pass
    static constexpr int PX = 32 * NSL / SPR;             // pixels per workgroup: 2 (C = 64), 1 (C = 128, 256)
    static constexpr int RT = (PX * 24 + 15) / 16;        // 16-row tiles
    static constexpr int KG = C_ / 16;                    // 16-wide K groups
    static constexpr int CG = C_ / 32;                    // channels per group: 2, 4, 8
};

__device__ __forceinline__ float tq_dpp_b1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float tq_dpp_4e(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float tq_dpp_hm(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); }
__device__ __forceinline__ float tq_dpp_r8(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true)); }

// sum over the 8 frame lanes of a slot (lane bits 0-2); CG = 8: also over the neighbouring channel quad (lane bit 3)
template <int CG>
__device__ __forceinline__ float tq_reduce(float v) {
    v += tq_dpp_b1(v);            // quad_perm [1,0,3,2]
    v += tq_dpp_4e(v);            // quad_perm [2,3,0,1]
    v += tq_dpp_hm(v);            // row_half_mirror: lane i <-> 7 - i of the 8 lanes (the other quad)
    if constexpr (CG == 8) v += tq_dpp_r8(v);     // row_ror:8
    return v;
}

// group sums of a lane's channel quad: CG = 2: (x + y) and (z + w); CG = 4, 8: one group
template <int CG>
__device__ __forceinline__ f32x4 tq_group(f32x4 s) {
    if constexpr (CG == 2) {
        const float a = tq_reduce<CG>(s.x + s.y), b = tq_reduce<CG>(s.z + s.w);
        return (f32x4){a, a, b, b};
    } else {
        const float a = tq_reduce<CG>((s.x + s.y) + (s.z + s.w));
        return (f32x4){a, a, a, a};
    }
}

template <int C_>
__global__ __launch_bounds__(256) void gn_temporal_qkv_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps,
                                                              float* __restrict__ xn_out, const float* __restrict__ W,
                                                              const float* __restrict__ bias, float* __restrict__ qkv,
                                                              TqGeom g) {
    using CF = TqCfg<C_>;
    constexpr int SPR = CF::SPR, NSL = CF::NSL, PX = CF::PX, RT = CF::RT, KG = CF::KG, CG = CF::CG;
    __shared__ __attribute__((aligned(16))) float As[RT * 16 * C_];
    // (column strip, share): the shares of a strip on one XCD
    const int id = blockIdx.x;
    const int xcd = id & 7, r8 = id >> 3;
    const int grp = (r8 / g.NSPLIT) * 8 + xcd, sp = r8 % g.NSPLIT;
    if (grp >= g.ngroups) return;                          // grid padding (workgroup-uniform)
    const int T = g.T, P = g.P;
    const int col0 = grp * PX;
    const int b = col0 / P, p0 = col0 - b * P;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, kk = lane >> 4;

    // ---- the column's rows: lane = (frame part tp, slot): frames tp, tp + 8, tp + 16
    const int tp = tid & 7, rest = tid >> 3;
    f32x4 xk[NSL][TQ_TF];
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
        const int s = rest + 32 * sl;
        const int quad = s % SPR, pix = s / SPR;
#pragma unroll
        for (int i = 0; i < TQ_TF; ++i) {
            const int t = tp + 8 * i;
            xk[sl][i] = t < T ? ld4(x + ((size_t)(b * T + t) * P + p0 + pix) * C_ + 4 * quad) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    // ---- filter fragments of this wave's first column tile (in flight during the statistics)
    const int NCT = g.NCT;
    auto wrow = [&](int ct) -> const float* { return W + (size_t)((sp * NCT + ct) * 16 + lq) * C_ + 4 * kk; };
    f32x4 wf[KG];
    {
        const float* wr = wrow(wave);
#pragma unroll
        for (int k = 0; k < KG; ++k) wf[k] = ld4(wr + 16 * k);
    }
    f32x4 gam[NSL], bet[NSL];
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
        const int quad = (rest + 32 * sl) % SPR;
        gam[sl] = ld4(gamma + 4 * quad);
        bet[sl] = ld4(beta + 4 * quad);
    }
    // ---- statistics, normalisation, LDS image (+ the normalised rows to memory)
    const float inv = 1.0f / (float)(CG * T);
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
        const int s = rest + 32 * sl;
        const int quad = s % SPR, pix = s / SPR;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TQ_TF; ++i) a += xk[sl][i];                 // (frames past T are zeros)
        const f32x4 mu = tq_group<CG>(a) * inv;
        f32x4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TQ_TF; ++i) {
            if (tp + 8 * i < T) {
                const f32x4 d = xk[sl][i] - mu;
                q += d * d;
            }
        }
        const f32x4 var = tq_group<CG>(q) * inv;
        f32x4 rs;
        rs.x = 1.0f / sqrtf(var.x + eps);
        rs.z = CG == 2 ? 1.0f / sqrtf(var.z + eps) : rs.x;
        rs.y = rs.x;
        rs.w = rs.z;
#pragma unroll
        for (int i = 0; i < TQ_TF; ++i) {
            const int t = tp + 8 * i;
            if (t < T) {
                const f32x4 v = (xk[sl][i] - mu) * rs * gam[sl] + bet[sl];
                const int row = pix * T + t;
                st4(As + ((size_t)row * SPR + ((quad & ~15) | ((quad ^ row) & 15))) * 4, v);
                if (sp == 0 && xn_out != nullptr) st4(xn_out + ((size_t)(b * T + t) * P + p0 + pix) * C_ + 4 * quad, v);
            }
        }
    }
    __syncthreads();

    // ---- [PX T rows][16 NCT columns] = As . W^T + bias
    const int rows = PX * T;
    for (int ct = wave; ct < NCT; ct += 4) {
        f32x4 wn[KG];
        const bool more = ct + 4 < NCT;
        if (more) {
            const float* wr = wrow(ct + 4);
#pragma unroll
            for (int k = 0; k < KG; ++k) wn[k] = ld4(wr + 16 * k);
        }
        // D^T = W . xn^T: the filter fragment is the A operand, so a lane ends up with FOUR CONSECUTIVE output columns of one
        // row (D lane: column = row l & 15 of the tile, rows = features 4 (l >> 4) + r) - one 16-byte store per lane and
        // row tile (the first version's scalar stores, column l & 15 per lane, were 4.4 of its 11.2 us)
        const int colq = (sp * NCT + ct) * 16 + 4 * kk;
        const f32x4 bv = ld4(bias + colq);
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KG; ++k) {
            f32x4 a4[RT];
            const int s = 4 * k + kk;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int row = 16 * rt + lq;
                a4[rt] = ld4(As + ((size_t)row * SPR + ((s & ~15) | ((s ^ row) & 15))) * 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[k][e], a4[rt][e], acc[rt], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row = 16 * rt + lq;
            if (row < rows) {
                const int pix = (PX > 1 && row >= T) ? 1 : 0;
                const int t = row - pix * T;
                st4(qkv + ((size_t)(b * T + t) * P + p0 + pix) * (3 * C_) + colq, acc[rt] + bv);
            }
        }
        if (more) {
#pragma unroll
            for (int k = 0; k < KG; ++k) wf[k] = wn[k];
        }
    }
}

// shares of the 3C columns: the fewest that give the chip ~a workgroup per CU, each a whole number of 4-tile rounds
int tq_split(int C, long ngroups, int& nct) {
    const int tiles = 3 * C / 16;
    int best = 1;
    static const int force = getenv("LFVDM_TQ_SPLIT") ? atoi(getenv("LFVDM_TQ_SPLIT")) : 0;      // tuning aid
    for (int ns : {1, 2, 3, 6, 12}) {
        if (tiles % ns || (tiles / ns) % 4) continue;
        best = ns;
        if (force ? ns >= force : ngroups * ns >= 256) break;
    }
    nct = tiles / best;
    return best;
}

}  // namespace

// LFVDM_OK if the fused launch covers the shape AND pays: C = 64 / 128 / 256 (two, one, one pixel per workgroup), T <= 24
// frames, the pixel count a multiple of the strip; at 128 / 256 channels only while the projection is small (<= 0.5 GFLOP:
// a workgroup multiplies T <= 24 rows in two 16-row tiles and re-reads its filter rows from L2 - measured, us per launch
// fused vs two launches (tools/gn_tqkv_bench.py): 8.2 / 11.5 at (B 2, 16x16, C 64), 23.2 / 26.5 at (B 8, 16x16, C 64),
// 8.9 / 11.4 at (8x8, C 128), 4.9 / 9.0 at (4x4 | 2x2, C 128), but 25.7 / 21.3 at (16x16, C 128) and 23.8 / 21.1 at
// (8x8, C 256), 1 GFLOP each).  Else LFVDM_E_UNSUPPORTED (lfvdm_gn_temporal + lfvdm_conv_igemm).
// LFVDM_TQ_ALWAYS=1 lifts the work limit (tests, tuning).
extern "C" int lfvdm_gn_temporal_qkv_ok(int B, int T, int P, int C) {
    if (B <= 0 || T <= 0 || P <= 0 || C <= 0) return LFVDM_E_SHAPE;
    if (C != 64 && C != 128 && C != 256) return LFVDM_E_UNSUPPORTED;
    const int PX = C == 64 ? 2 : 1;
    if (T > 24 || P % PX) return LFVDM_E_UNSUPPORTED;
    if ((long)B * T * P * 3 * C * 4 >= (1L << 31)) return LFVDM_E_UNSUPPORTED;
    const bool always = [] { const char* e = getenv("LFVDM_TQ_ALWAYS"); return e && e[0] == '1'; }();
    if (C != 64 && !always && 2.0 * B * T * P * 3.0 * C * C > 0.5e9) return LFVDM_E_UNSUPPORTED;
    return LFVDM_OK;
}

extern "C" int lfvdm_gn_temporal_qkv(const float* x, const float* gamma, const float* beta, float eps, float* xn_out,
                                     const float* Wqkv, const float* bqkv, float* qkv, int B, int T, int P, int C, void* stream) {
    if (!x || !gamma || !beta || !Wqkv || !bqkv || !qkv || x == xn_out) return LFVDM_E_SHAPE;
    if (B <= 0 || T <= 0 || P <= 0) return LFVDM_E_SHAPE;
    if (C != 64 && C != 128 && C != 256) return LFVDM_E_UNSUPPORTED;
    const int PX = C == 64 ? 2 : 1;
    if (T > 24 || P % PX || (long)B * T * P * 3 * C * 4 >= (1L << 31)) return LFVDM_E_UNSUPPORTED;      // (no work limit here)
    TqGeom g;
    g.B = B; g.T = T; g.P = P;
    g.ngroups = B * P / PX;
    g.NSPLIT = tq_split(C, g.ngroups, g.NCT);
    const unsigned grid = 8u * (unsigned)((g.ngroups + 7) / 8) * (unsigned)g.NSPLIT;
    hipStream_t s = (hipStream_t)stream;
    if (C == 64) hipLaunchKernelGGL(gn_temporal_qkv_kernel<64>, dim3(grid), dim3(256), 0, s, x, gamma, beta, eps, xn_out, Wqkv, bqkv, qkv, g);
    else if (C == 128) hipLaunchKernelGGL(gn_temporal_qkv_kernel<128>, dim3(grid), dim3(256), 0, s, x, gamma, beta, eps, xn_out, Wqkv, bqkv, qkv, g);
    else hipLaunchKernelGGL(gn_temporal_qkv_kernel<256>, dim3(grid), dim3(256), 0, s, x, gamma, beta, eps, xn_out, Wqkv, bqkv, qkv, g);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}
