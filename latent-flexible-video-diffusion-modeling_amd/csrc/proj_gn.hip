// 1x1 projection + residual + the NEXT GroupNorm in one launch, for maps of 256 positions: the temporal attention's output
// projection (reference rpe.py:171-172 `x + self.proj_out(out)`, nn.Linear(C, C), residual = the temporally normalised
// tokens) followed by the spatial attention's `self.norm` (rpe.py:136, GroupNorm32(32, C) per frame, nn.py:93-101).
//
// At the 8x8 level and below the GroupNorm already rides in the projection's epilogue (a GEMM tile holds whole frames); at
// 16x16 a frame is 256 rows and the plan ran lfvdm_conv_igemm (1x1, 84 MFLOP: ~5.3 us, all latency) and then lfvdm_gn_apply
// (~5.1 us).  GroupNorm units are independent, so the decomposition of gn_wave4 carries over: a workgroup owns (frame n,
// 16 output channels), wave w the positions [64 w, 64 w + 64) (8x8 frames, where the GEMM's GroupNorm epilogue costs 8.5 us
// for 84 MFLOP: [16 w, 16 w + 16)):
//   * fp32 MFMA 16x16x4 on the TRANSPOSED tile, D^T = W . o^T: the 16 filter rows of the unit are the A operand, the
//     attention output rows of the wave the B operand - both read from global memory straight into fragments (16 bytes per
//     lane and 16-wide K group); a lane ends up with channels 4 kk ... 4 kk + 3 of token l & 15 of each of its four row
//     tiles: the (channel quad, position) layout of the GroupNorm kernels;
//   * + bias + residual, then two-pass statistics: over the lane's row tiles in registers, over the 16 tokens of a DPP row
//     (quad_perm, row_ror), over the four waves through 256 B of LDS in wave order (deterministic), x * A + B as gn_wave_body;
//   * the four workgroups of a frame sit on one XCD (blockIdx % 8) and share its rows of `o` in that L2.
// The un-normalised sum is written only on request (`raw_out`): inside an attention block nothing else reads it (the residual
// of the spatial projection is the NORMALISED tensor, rpe.py:136 + :172).  The same launch serves the block's LAST projection
// when the consumer normalises its output first: the U-Net head `out = [GroupNorm32, SiLU, conv]` (unet.py:418-422) with
// act = SiLU.
#include <cstdlib>

#include "common_hip.h"

namespace {

#define LFVDM_PG_DPP(name, ctrl)                                                                                          \
    __device__ __forceinline__ float name(float v) {                                                                       \
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true));      \
    }
LFVDM_PG_DPP(pg_b1, 0xB1)       // quad_perm [1,0,3,2]
LFVDM_PG_DPP(pg_4e, 0x4E)       // quad_perm [2,3,0,1]
LFVDM_PG_DPP(pg_r4, 0x124)      // row_ror:4
LFVDM_PG_DPP(pg_r8, 0x128)      // row_ror:8
#undef LFVDM_PG_DPP

// sum over the 16 lanes of a DPP row (the 16 tokens of a row tile; every lane of the row ends up with the total)
__device__ __forceinline__ float pg_row_sum(float v) {
    v += pg_b1(v);
    v += pg_4e(v);
    v += pg_r4(v);
    v += pg_r8(v);
    return v;
}

template <int C_, int P>
__global__ __launch_bounds__(256) void proj_gn_kernel(const float* __restrict__ o, const float* __restrict__ W,
                                                      const float* __restrict__ bias, const float* __restrict__ res,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                      float* __restrict__ out, int N, int act, float* __restrict__ raw_out) {
    constexpr int KG = C_ / 16, CG = C_ / 32, NCB = C_ / 16, RT = P / 64, PW = P / 4;      // PW: positions per wave
    __shared__ f32x4 xch[2][4][4];                        // [pass][wave][kk]: the waves' partial group sums
    const int id = blockIdx.x;
    const int xcd = id & 7, r8 = id >> 3;
    const int n = (r8 / NCB) * 8 + xcd, cb = r8 % NCB;
    if (n >= N) return;                                   // grid padding (workgroup-uniform)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, kk = lane >> 4;
    const int c = 16 * cb + 4 * kk;                       // this lane's channel quad
    const size_t row0 = (size_t)n * P + PW * wave + lq;   // + 16 rt

    // ---- operands: the unit's filter rows (A), the wave's attention-output rows (B), bias / residual / affine
    f32x4 wf[KG];
    {
        const float* wr = W + (size_t)(16 * cb + lq) * C_ + 4 * kk;
#pragma unroll
        for (int k = 0; k < KG; ++k) wf[k] = ld4(wr + 16 * k);
    }
    f32x4 acc[RT];
    f32x4 rs[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) rs[rt] = ld4(res + (row0 + 16 * rt) * C_ + c);
    const f32x4 bv = ld4(bias + c), gam = ld4(gamma + c), bet = ld4(beta + c);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const float* orow = o + (row0 + 16 * rt) * C_ + 4 * kk;
        f32x4 a4[KG];
#pragma unroll
        for (int k = 0; k < KG; ++k) a4[k] = ld4(orow + 16 * k);
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KG; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[k][e], a4[k][e], d, 0, 0, 0);
        acc[rt] = d + bv + rs[rt];                        // token PW wave + 16 rt + lq, channels c ... c + 3
        if (raw_out != nullptr) st4(raw_out + (row0 + 16 * rt) * C_ + c, acc[rt]);
    }
    // ---- GroupNorm of the unit: CG = 2: groups (x, y) and (z, w) of the quad; CG = 4: the quad is one group
    const float inv = 1.0f / (float)(CG * P);
    auto unit_sum = [&](f32x4 v, int pass) -> f32x4 {     // v: this lane's sum over its row tiles
        float a, b;
        if constexpr (CG == 2) { a = pg_row_sum(v.x + v.y); b = pg_row_sum(v.z + v.w); }
        else { a = pg_row_sum((v.x + v.y) + (v.z + v.w)); b = a; }
        if (lq == 0) xch[pass][wave][kk] = (f32x4){a, b, 0.f, 0.f};
        __syncthreads();
        f32x4 t = xch[pass][0][kk];
#pragma unroll
        for (int w = 1; w < 4; ++w) t += xch[pass][w][kk];
        return (f32x4){t.x, t.x, t.y, t.y};
    };
    f32x4 s = acc[0];
#pragma unroll
    for (int rt = 1; rt < RT; ++rt) s += acc[rt];
    const f32x4 mean = unit_sum(s, 0) * inv;
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const f32x4 d = acc[rt] - mean;
        q += d * d;
    }
    const f32x4 var = unit_sum(q, 1) * inv;
    f32x4 rstd;
    rstd.x = 1.0f / sqrtf(var.x + eps);
    rstd.z = CG == 2 ? 1.0f / sqrtf(var.z + eps) : rstd.x;
    rstd.y = rstd.x;
    rstd.w = rstd.z;
    const f32x4 A = rstd * gam;
    const f32x4 B = bet - mean * A;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        f32x4 v = acc[rt] * A + B;
        if (act == LFVDM_ACT_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
        st4(out + (row0 + 16 * rt) * C_ + c, v);
    }
}

}  // namespace

// LFVDM_OK: frames of 256 or 64 positions (four or one 16-row tile per wave), 64 or 128 channels (a GroupNorm unit of 16
// channels = 8 or 4 whole groups)
extern "C" int lfvdm_proj_gn_ok(int N, int P, int C) {
    if (N <= 0 || P <= 0 || C <= 0) return LFVDM_E_SHAPE;
    if ((P != 256 && P != 64) || (C != 64 && C != 128)) return LFVDM_E_UNSUPPORTED;
    if ((long)N * P * C * 4 >= (1L << 31)) return LFVDM_E_UNSUPPORTED;
    return LFVDM_OK;
}

extern "C" int lfvdm_proj_gn(const float* o, const float* W, const float* bias, const float* res, const float* gamma,
                             const float* beta, float eps, int act, float* out, float* raw_out, int N, int P, int C, void* stream) {
    if (!o || !W || !bias || !res || !gamma || !beta || !out || out == o || out == res) return LFVDM_E_SHAPE;
    if (raw_out == o || raw_out == res || raw_out == out) return LFVDM_E_SHAPE;     // (NULL: the raw sum is not written)
    if (act != LFVDM_ACT_NONE && act != LFVDM_ACT_SILU) return LFVDM_E_SHAPE;
    if (int rc = lfvdm_proj_gn_ok(N, P, C)) return rc;
    const unsigned grid = 8u * (unsigned)((N + 7) / 8) * (unsigned)(C / 16);
    hipStream_t s = (hipStream_t)stream;
#define LFVDM_PG(CV, PV)                                                                                                      \
    hipLaunchKernelGGL((proj_gn_kernel<CV, PV>), dim3(grid), dim3(256), 0, s, o, W, bias, res, gamma, beta, eps, out, N, act, raw_out)
    if (C == 64 && P == 256) LFVDM_PG(64, 256);
    else if (C == 64) LFVDM_PG(64, 64);
    else if (P == 256) LFVDM_PG(128, 256);
    else LFVDM_PG(128, 64);
#undef LFVDM_PG
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}
