// Implicit-GEMM convolution / linear for gfx950 on fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Workgroup = WK "k-groups" of WM x WN waves.  A k-group owns a contiguous slice of the K loop
// (K = taps*Cin in 32- or 64-channel chunks, plus the optional fused 1x1 skip segment) and computes the
// whole (32*WM) x (32*NT*WN) block tile for that slice; the k-groups' partial tiles are summed
// through LDS at the end.  Inside a k-group the A tile (output pixels x KC channels of one filter
// tap, gathered from the channels-last RAW activation - GroupNorm / FiLM / SiLU are materialised by the
// producer's epilogue or by lfvdm_gn_apply, never applied here) and the W tile are staged by LDS-DMA
// (buffer_load ... lds) into 2 or 3 unpadded, XOR-swizzled stages: no VGPR staging, no ds_write, zero padding by
// out-of-range offsets; one s_barrier per chunk.  The K loop runs channel-chunk-major / tap-minor.
// (The register-staged loop with the GroupNorm prologue folded into the operand load - round 1, LFVDM_FUSED_GN in
// round 2 - was measured 15-30 % slower and is gone: DESIGN.md section 5.)
//
// MFMA operand mapping (cdna guide section 3): A lane l holds A[i=l&31][k=l>>5], B lane l holds
// B[k=l>>5][j=l&31]; within a group of 8 k the e-th MFMA uses k = 8g + 4h + e on both operands.
// D: lane l holds column j=l&31, rows (r&3) + 8*(r>>2) + 4*(l>>5), r=0..15.
#include <stdlib.h>

#include "common_hip.h"

#ifdef LFVDM_STAMP
// diagnostic build only (never compiled into the product): per-workgroup phase stamps of thread 0 on the 100 MHz
// s_memrealtime clock, which is common to all CUs - launch skew, arrival order and the split-K seam become visible
constexpr int kStampWGs = 2048, kStampN = 16;
__device__ unsigned long long g_stamps[kStampWGs * kStampN];
#define STAMP(i) do { const unsigned sb_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                  \
                      if (threadIdx.x == 0 && sb_ < kStampWGs) g_stamps[sb_ * kStampN + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int lfvdm_debug_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) == hipSuccess ? 0 : 2;
}
extern "C" int lfvdm_debug_stamps_clear(void) {
    static unsigned long long zeros[kStampWGs * kStampN];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zeros, sizeof(zeros)) == hipSuccess ? 0 : 2;
}
#else
#define STAMP(i) do {} while (0)
#endif

#include "conv_igemm_body.h"

namespace {

template <int WM, int WN, int WK, int NT, int KCH, bool SIMPLE, int GL>
__global__ __launch_bounds__(64 * WM * WN * WK) void conv_igemm_kernel(const lfvdm_conv_args p_in, int hyb_nfull, int hyb_kz,
                                                                        int par_mt) {
    const lfvdm_conv_args p = p_in;   // private SSA copy: helpers take it by reference (keeps it out of scratch)
    conv_igemm_body<WM, WN, WK, NT, KCH, SIMPLE, GL, false>(p, hyb_nfull, hyb_kz, par_mt, ChainCtx{});
}

__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ o, int Cout, int Cin, int taps) {
    // o[co][tap][ci] = w[co][ci][tap]
    const size_t total = (size_t)Cout * Cin * taps;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const size_t t2 = i / Cin;
        const int tap = (int)(t2 % taps);
        const int co = (int)(t2 / taps);
        o[i] = w[((size_t)co * Cin + ci) * taps + tap];
    }
}

template <int WM, int WN, int WK, int NT, int KCH, bool SIMPLE, int GL>
int launch_pro(const lfvdm_conv_args* a, hipStream_t s, long M, int kz) {
    using CF = Cfg<WM, WN, WK, NT, KCH, GL>;
    if (CF::LDS_BYTES > 160 * 1024) return LFVDM_E_UNSUPPORTED;
    static DynLdsLimit limit;
    if (int rc = limit.ensure(reinterpret_cast<const void*>(&conv_igemm_kernel<WM, WN, WK, NT, KCH, SIMPLE, GL>), CF::LDS_BYTES))
        return rc;
    const long MT = (M + CF::BM - 1) / CF::BM, NT2 = (a->Cout + CF::BN - 1) / CF::BN;
    if constexpr (!SIMPLE) {
        if (parity_classes(a)) {
            const long MTc = (M / 4 + CF::BM - 1) / CF::BM;
            if (kz == kHybridKz || 4 * MTc >= (1L << 20)) return LFVDM_E_UNSUPPORTED;
            hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, WK, NT, KCH, SIMPLE, GL>), dim3((unsigned)(4 * MTc), (unsigned)NT2, (unsigned)kz),
                               dim3(CF::NTHREADS), CF::LDS_BYTES, s, *a, 0, 0, (int)MTc);
            LFVDM_CHECK_LAUNCH();
            return LFVDM_OK;
        }
    }
    if (kz == kHybridKz) {   // tail split (see the kernel): flat grid
        const HybridPlan h = hybrid_plan(MT * NT2);
        hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, WK, NT, KCH, SIMPLE, GL>), dim3((unsigned)(h.nfull + h.tail * h.kz)),
                           dim3(CF::NTHREADS), CF::LDS_BYTES, s, *a, (int)h.nfull, h.kz, 0);
        LFVDM_CHECK_LAUNCH();
        return LFVDM_OK;
    }
    static const bool no_xmap = getenv("LFVDM_CONV_NO_XCD_MAP") != nullptr;       // A/B aid
    // XCD-aware map (see the kernel) for split-K launches: flat grid of 8 equal per-XCD ranges.  (Padding the PAIR count to
    // a multiple of 8 instead was measured: 1063 -> 983 steps/s - the XCDs that own a padding pair idle while others run two.)
    const long total = MT * NT2 * kz, per = (total + 7) / 8;
    if (!no_xmap && kz > 1 && total >= 16 && 8 * per < (1L << 20)) {
        hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, WK, NT, KCH, SIMPLE, GL>), dim3((unsigned)(8 * per)), dim3(CF::NTHREADS),
                           CF::LDS_BYTES, s, *a, (int)NT2, -kz, 0);
        LFVDM_CHECK_LAUNCH();
        return LFVDM_OK;
    }
    const dim3 grid((unsigned)MT, (unsigned)NT2, (unsigned)kz);
    hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, WK, NT, KCH, SIMPLE, GL>), grid, dim3(CF::NTHREADS), CF::LDS_BYTES, s, *a, 0, 0, 0);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

template <int WM, int WN, int WK, int NT, int KCH>
int launch_kc(const lfvdm_conv_args* a, hipStream_t s, long M, int kz, int gl) {
    const bool simple = a->C1 == 0 && a->s2C0 + a->s2C1 == 0 && a->up == 0;
    if constexpr (glds_lds_bytes(WM, WN, WK, NT, KCH, 3) <= 160 * 1024) {
        if (gl == 3) return simple ? launch_pro<WM, WN, WK, NT, KCH, true, 3>(a, s, M, kz)
                                   : launch_pro<WM, WN, WK, NT, KCH, false, 3>(a, s, M, kz);
    }
    if constexpr (glds_lds_bytes(WM, WN, WK, NT, KCH, 2) <= 160 * 1024) {
        return simple ? launch_pro<WM, WN, WK, NT, KCH, true, 2>(a, s, M, kz)
                      : launch_pro<WM, WN, WK, NT, KCH, false, 2>(a, s, M, kz);
    }
    return LFVDM_E_UNSUPPORTED;
}

template <int WM, int WN, int WK, int NT>
int launch_cfg(const lfvdm_conv_args* a, hipStream_t s, long M, int kch, int kz, int gl) {
    if constexpr (NT == 1) {   // 64-channel chunks: single-filter-tile configurations
        if (kch == 64) return launch_kc<WM, WN, WK, NT, 64>(a, s, M, kz, gl);
    }
    return launch_kc<WM, WN, WK, NT, 32>(a, s, M, kz, gl);
}

// Modelled makespan (cycles) of one launch: 256 CUs x 4 SIMDs, 64 cycles per 32x32x2 MFMA, one barrier
// per chunk.  Only the starting point: every launch shape is timed over its legal variants on first sight.
double model_cycles(const TileCfg& c, int Cout, long M, int NK, int kch, int kz) {
    const int BM = 32 * c.WM, BN = 32 * c.NT * c.WN;
    const int waves = c.WM * c.WN * c.WK;
    const double lds = (double)c.WK * 2.0 * (BM + BN) * kch * 4.0;
    int resident = (int)(160.0 * 1024.0 / lds);
    const int by_vgpr = (c.vgpr_waves * 4) / waves;
    if (by_vgpr < resident) resident = by_vgpr;
    if (resident < 1) resident = 1;
    const long wgs = ((M + BM - 1) / BM) * ((Cout + BN - 1) / BN) * kz;
    const long slots = 256L * resident;
    const long rounds = (wgs + slots - 1) / slots;
    long per_cu = (wgs + 255) / 256;
    if (per_cu > resident) per_cu = resident;
    const double simd_waves = (double)per_cu * ((waves + 3) / 4);
    const double chunks = (double)(((NK + kz - 1) / kz + c.WK - 1) / c.WK);
    const double mfma = 16.0 * c.NT * 64.0 * (kch / 32);
    double per_iter = mfma * simd_waves;
    const double floor_iter = 1000.0 + 0.3 * mfma;   // barrier + staging + exposed latency of a lone wave
    if (per_iter < floor_iter) per_iter = floor_iter;
    return (double)rounds * (chunks * per_iter + 3500.0 + 900.0 + 250.0 * c.WK) + (kz > 1 ? 2500.0 : 0.0);
}

Pick pick_cfg(const lfvdm_conv_args* a, long M) {
    const int Cin = a->C0 + a->C1, C2 = a->s2C0 + a->s2C1;
    if (a->tune > 0) {   // explicit choice (autotuner); fall through to the model if it is not legal here
        const int t = a->tune - 1;
        const int id = t & 15, kch = (t & 16) ? 64 : 32, kz = kKzTable[(t >> 5) & 7], gl = ((t >> 8) & 3) + 1;
        if (cfg_valid(a, id, kch, kz, gl)) return {id, kch, a->ksize * a->ksize * (Cin / kch) + C2 / kch, kz, gl};
    }
    static const int forced = getenv("LFVDM_CONV_CFG") ? atoi(getenv("LFVDM_CONV_CFG")) : -1;  // tuning aid
    Pick best = {-1, 32, a->ksize * a->ksize * (Cin / 32) + C2 / 32, 1, 2};
    double best_t = 1e30;
    for (int i = 0; i < kNumCfgs; ++i) {
        if (forced >= 0 && i != forced) continue;
        for (int kch = 32; kch <= 64; kch += 32) {
            if (!cfg_valid(a, i, kch, 1, 2)) continue;          // the built-in model never splits K over workgroups
            const int NK = a->ksize * a->ksize * (Cin / kch) + C2 / kch;
            const double est = model_cycles(kCfgs[i], a->Cout, M, NK, kch, 1);
            if (est < best_t) { best_t = est; best = {i, kch, NK, 1, 2}; }
        }
    }
    if (best.id < 0 && !a->gn_out) best.id = 0;   // nothing passed the filters (tiny K with a narrow output): the 64x64 tile always
                                                  // works; with a fused GroupNorm no tile holds whole samples and groups: refused
    return best;
}

}  // namespace

static int conv_igemm_one(const lfvdm_conv_args* a, hipStream_t s) {
    const long M = (long)a->N * a->Ho * a->Wo;
    const Pick pk = pick_cfg(a, M);
    const int kch = pk.kch, kz = pk.kz, gl = pk.gl;
    switch (pk.id) {
        case 0: return launch_cfg<2, 2, 1, 1>(a, s, M, kch, kz, gl);
        case 1: return launch_cfg<2, 2, 1, 2>(a, s, M, kch, kz, gl);
        case 2: return launch_cfg<1, 2, 2, 1>(a, s, M, kch, kz, gl);
        case 3: return launch_cfg<1, 2, 4, 1>(a, s, M, kch, kz, gl);
        case 4: return launch_cfg<1, 1, 8, 1>(a, s, M, kch, kz, gl);
        case 5: return launch_cfg<2, 2, 2, 1>(a, s, M, kch, kz, gl);
        case 6: return launch_cfg<1, 1, 4, 1>(a, s, M, kch, kz, gl);
    }
    return LFVDM_E_UNSUPPORTED;
}

extern "C" int lfvdm_conv_igemm(const lfvdm_conv_args* a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const int Cin = a->C0 + a->C1;
    const int C2 = a->s2C0 + a->s2C1;
    if (a->N <= 0 || a->Cout <= 0 || Cin <= 0) return LFVDM_E_SHAPE;
    if (Cin % 32 || a->C0 % 32 || C2 % 32 || a->s2C0 % 32) return LFVDM_E_SHAPE;
    if (a->ksize != 1 && a->ksize != 3) return LFVDM_E_SHAPE;
    if (a->stride != 1 && a->stride != 2) return LFVDM_E_SHAPE;
    if (a->C1 > 0 && !a->src1) return LFVDM_E_SHAPE;
    if (C2 > 0 && (!a->W2 || !a->s2src0 || (a->s2C1 > 0 && !a->s2src1))) return LFVDM_E_SHAPE;
    // the operand prologue (GroupNorm coefficients applied while staging) is gone: normalise with lfvdm_gn_apply or a
    // producer's gn_* epilogue and pass the raw tensor
    if (a->coefA || a->coefB) return LFVDM_E_UNSUPPORTED;
    if (a->out_mode == LFVDM_OUT_ROWS && (a->Cout % 4 || a->ldo % 4 || (a->res && a->ldr % 4))) return LFVDM_E_SHAPE;
    {   // output size must agree with the conv arithmetic the kernel assumes
        const int Hin = a->up ? 2 * a->Hs : a->Hs, Win = a->up ? 2 * a->Ws : a->Ws;
        const int pad = a->ksize == 3 ? 1 : 0;
        if ((Hin + 2 * pad - a->ksize) / a->stride + 1 != a->Ho) return LFVDM_E_SHAPE;
        if ((Win + 2 * pad - a->ksize) / a->stride + 1 != a->Wo) return LFVDM_E_SHAPE;
    }
    if ((long)a->Cout * a->ksize * a->ksize * Cin >= (1L << 28) || (long)a->Cout * C2 >= (1L << 28)) return LFVDM_E_UNSUPPORTED;
    if (9L * Cin + C2 >= (1L << 20)) return LFVDM_E_UNSUPPORTED;     // K-slice arithmetic of the kernel: NK * KZ < 2^21
    if (a->gn_out && (!a->gn_gamma || !a->gn_beta || a->gn_film_div <= 0 || (a->gn_film && a->gn_film_ld < 2 * a->Cout)))
        return LFVDM_E_SHAPE;
    if (a->gn_out && ((a->gn_gw && (a->gn_gw < 0 || a->Cout % a->gn_gw)) || (a->gn_ld && (a->gn_ld < a->Cout || a->gn_ld % 4))))
        return LFVDM_E_SHAPE;
    if (glds_ok(a)) return conv_igemm_one(a, s);
    // Per-lane byte offsets are 32-bit and end below 2^30: a batch whose tensors exceed that (pixel space at large
    // batch) is processed in sample ranges - samples are independent in every operand and in the fused GroupNorm; the
    // FiLM rows of the epilogue are indexed by sample / gn_film_div, so ranges start at multiples of it.
    const long Cmax = a->C0 > a->C1 ? a->C0 : a->C1, C2max = a->s2C0 > a->s2C1 ? a->s2C0 : a->s2C1;
    long per = (long)a->Hs * a->Ws * Cmax * 4;
    if ((long)a->Ho * a->Wo * C2max * 4 > per) per = (long)a->Ho * a->Wo * C2max * 4;
    long nmax = ((1L << 30) - 1) / per;
    const int step = (a->gn_out && a->gn_film) ? a->gn_film_div : 1;
    nmax = nmax / step * step;
    if (nmax < 1) return LFVDM_E_UNSUPPORTED;              // one sample alone exceeds the offset range
    const long P = (long)a->Ho * a->Wo, Ps = (long)a->Hs * a->Ws;
    for (long n0 = 0; n0 < a->N; n0 += nmax) {
        lfvdm_conv_args b = *a;
        b.N = (int)(a->N - n0 < nmax ? a->N - n0 : nmax);
        b.src0 = a->src0 + n0 * Ps * a->C0;
        if (a->src1) b.src1 = a->src1 + n0 * Ps * a->C1;
        if (a->s2src0) b.s2src0 = a->s2src0 + n0 * P * a->s2C0;
        if (a->s2src1) b.s2src1 = a->s2src1 + n0 * P * a->s2C1;
        if (a->res) b.res = a->res + n0 * P * a->ldr;
        if (a->resA) { b.resA = a->resA + n0 * a->Cout; b.resB = a->resB + n0 * a->Cout; }
        b.out = a->out + (a->out_mode == LFVDM_OUT_NCHW ? n0 * a->Cout * P : n0 * P * a->ldo);
        if (a->gn_out) {
            b.gn_out = a->gn_out + n0 * P * (a->gn_ld ? a->gn_ld : a->Cout);
            if (a->gn_film) b.gn_film = a->gn_film + (n0 / a->gn_film_div) * a->gn_film_ld;
        }
        if (!glds_ok(&b)) return LFVDM_E_UNSUPPORTED;
        if (int rc = conv_igemm_one(&b, s)) return rc;
    }
    return LFVDM_OK;
}

extern "C" int lfvdm_pack_conv_weight(const float* w, float* o, int Cout, int Cin, int ksize, void* stream) {
    if (Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3)) return LFVDM_E_SHAPE;
    const size_t total = (size_t)Cout * Cin * ksize * ksize;
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, o, Cout, Cin, ksize * ksize);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

// Which template instance lfvdm_conv_igemm would launch for these arguments (profiling aid: lets
// bench.py attribute per-launch HIP-event timings to the kernel symbol rocprofv3 reports).
// All legal tune codes for these arguments (for an autotuner); returns how many were written.
extern "C" int lfvdm_conv_igemm_candidates(const lfvdm_conv_args* a, int* codes, int max_codes) {
    int n = 0;
    for (int gl = 2; gl <= 3; ++gl)
        for (int id = 0; id < kNumCfgs; ++id)
            for (int kch = 32; kch <= 64; kch += 32)
                for (int l = 0; l < 8; ++l)
                    if (cfg_valid(a, id, kch, kKzTable[l], gl) && n < max_codes) codes[n++] = encode_tune(id, kch, kKzTable[l], gl);
    return n;
}

extern "C" int lfvdm_conv_igemm_config(const lfvdm_conv_args* a, int* nt, int* nwaves) {
    const int Cin = a->C0 + a->C1, C2 = a->s2C0 + a->s2C1;
    if (Cin <= 0 || Cin % 32 || C2 % 32) return LFVDM_E_SHAPE;
    const long M = (long)a->N * a->Ho * a->Wo;
    const Pick pk = pick_cfg(a, M);
    const int id = pk.id, kch = pk.kch;
    if (id < 0) return LFVDM_E_UNSUPPORTED;
    // encoded as (WM*1000 + WN*100 + WK*10 + NT, waves) so that callers can print the template instance
    *nt = kCfgs[id].WM * 1000 + kCfgs[id].WN * 100 + kCfgs[id].WK * 10 + kCfgs[id].NT;
    *nwaves = kCfgs[id].WM * kCfgs[id].WN * kCfgs[id].WK + 1000 * kch;
    return LFVDM_OK;
}

extern "C" int lfvdm_abi_version(void) { return 9; }
