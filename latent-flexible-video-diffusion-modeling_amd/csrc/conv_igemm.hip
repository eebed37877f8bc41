// Implicit-GEMM convolution / linear for gfx950 on fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// One workgroup = one 32(M) x 32*NT(N) output tile; its NWAVES waves split the K dimension
// (K = taps*Cin in 32-channel chunks, plus the optional fused 1x1 skip segment) and are summed
// through LDS at the end.  Each wave stages its own A chunk (32 output pixels x 32 channels of
// one filter tap, gathered from the channels-last activation with the GroupNorm/FiLM affine and
// SiLU applied on the fly, zero outside the image) and W chunk (32*NT filters x 32 k) in a
// wave-private LDS region, so the main loop has no s_barrier; the next chunk's global loads are
// issued before the current chunk's 16*NT MFMAs.  LDS rows are padded to 36 floats so that the
// ds_read_b128 fragment reads (lane (i, h) reads k = 8g+4h..+3 of row i) are conflict-free.
//
// MFMA operand mapping (cdna guide §3): A lane l holds A[i=l&31][k=l>>5], B lane l holds
// B[k=l>>5][j=l&31]; within a group of 8 k the e-th MFMA uses k = 8g + 4h + e on both operands.
// D: lane l holds column j=l&31, rows (r&3) + 8*(r>>2) + 4*(l>>5), r=0..15.
#include "common.cuh"

namespace {

constexpr int KC = 32;    // channels per chunk
constexpr int LDR = 36;   // padded LDS row (floats)

struct RowInfo {
    int n, oy, ox;
    bool valid;
};

// Raw operands of one K chunk as they come back from memory.  Loads are issued unconditionally
// (out-of-image taps and rows past M read a clamped, valid address and are masked afterwards), so
// no s_waitcnt sits between them: the whole chunk is in flight while the previous chunk's MFMAs run.
template <int NT>
struct ChunkRegs {
    f32x4 a[4];
    f32x4 ca[4], cb[4];
    f32x4 w[4 * NT];
    unsigned amask;   // bit r: row r of this lane is inside the image
    unsigned wmask;   // bit r: filter row r exists (co < Cout)
    bool main_seg;
};

template <int NT>
__device__ __forceinline__ void issue_chunk(const lfvdm_conv_args& p, int kc, int NK1, int cpt, int Cin, int n0,
                                            const RowInfo (&ri)[4], int m0, int M, int lane, ChunkRegs<NT>& R) {
    const int col = (lane & 7) * 4;
    const int rsub = lane >> 3;
    R.amask = 0;
    R.wmask = 0;
    R.main_seg = kc < NK1;
    if (kc < NK1) {
        const int tap = kc / cpt;
        const int cc = (kc - tap * cpt) * KC;
        int dy = 0, dx = 0;
        if (p.ksize == 3) {
            dy = tap / 3 - 1;
            dx = tap - (tap / 3) * 3 - 1;
        }
        const float* src;
        int Csrc, cl;
        if (cc < p.C0) {
            src = p.src0; Csrc = p.C0; cl = cc;
        } else {
            src = p.src1; Csrc = p.C1; cl = cc - p.C0;
        }
        const int Hin = p.up ? 2 * p.Hs : p.Hs;
        const int Win = p.up ? 2 * p.Ws : p.Ws;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const RowInfo& q = ri[r];
            const int iy = q.oy * p.stride + dy;
            const int ix = q.ox * p.stride + dx;
            const bool inb = q.valid && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
            R.amask |= (inb ? 1u : 0u) << r;
            const int cy = min(max(iy, 0), Hin - 1), cx = min(max(ix, 0), Win - 1);
            const int sy = p.up ? (cy >> 1) : cy;
            const int sx = p.up ? (cx >> 1) : cx;
            R.a[r] = ld4(src + ((size_t)(q.n * p.Hs + sy) * p.Ws + sx) * Csrc + cl + col);
            if (p.coefA) {
                R.ca[r] = ld4(p.coefA + (size_t)q.n * Cin + cc + col);
                R.cb[r] = ld4(p.coefB + (size_t)q.n * Cin + cc + col);
            }
        }
        const int Ktot = NK1 * KC;
#pragma unroll
        for (int r = 0; r < 4 * NT; ++r) {
            const int co = n0 + r * 8 + rsub;
            R.wmask |= (co < p.Cout ? 1u : 0u) << r;
            R.w[r] = ld4(p.W + (size_t)min(co, p.Cout - 1) * Ktot + kc * KC + col);
        }
    } else {
        const int cc = (kc - NK1) * KC;
        const int C2 = p.s2C0 + p.s2C1;
        const float* src;
        int Csrc, cl;
        if (cc < p.s2C0) {
            src = p.s2src0; Csrc = p.s2C0; cl = cc;
        } else {
            src = p.s2src1; Csrc = p.s2C1; cl = cc - p.s2C0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            R.amask |= (ri[r].valid ? 1u : 0u) << r;
            R.a[r] = ld4(src + (size_t)min(m0 + r * 8 + rsub, M - 1) * Csrc + cl + col);
        }
#pragma unroll
        for (int r = 0; r < 4 * NT; ++r) {
            const int co = n0 + r * 8 + rsub;
            R.wmask |= (co < p.Cout ? 1u : 0u) << r;
            R.w[r] = ld4(p.W2 + (size_t)min(co, p.Cout - 1) * C2 + cc + col);
        }
    }
}

// affine + activation + masking in registers, then the wave-private LDS stores
template <int NT>
__device__ __forceinline__ void finish_chunk(const lfvdm_conv_args& p, ChunkRegs<NT>& R, float* As, float* Ws, int st_off) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        f32x4 v = R.a[r];
        if (R.main_seg) {
            if (p.coefA) v = v * R.ca[r] + R.cb[r];
            if (p.act == LFVDM_ACT_SILU) {
                v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w);
            }
        }
        v = ((R.amask >> r) & 1u) ? v : zero;
        st4(As + r * 8 * LDR + st_off, v);
    }
#pragma unroll
    for (int r = 0; r < 4 * NT; ++r) st4(Ws + r * 8 * LDR + st_off, ((R.wmask >> r) & 1u) ? R.w[r] : zero);
}

template <int NT, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void conv_igemm_kernel(const lfvdm_conv_args p) {
    constexpr int BN = 32 * NT;
    constexpr int WAVE_LDS = (32 + BN) * LDR;  // floats per wave
    constexpr int RED_LD = BN + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 32;
    const int n0 = blockIdx.y * BN;
    const int HoWo = p.Ho * p.Wo;
    const int M = p.N * HoWo;
    const int Cin = p.C0 + p.C1;
    const int cpt = Cin / KC;
    const int NK1 = p.ksize * p.ksize * cpt;
    const int NK = NK1 + (p.s2C0 + p.s2C1) / KC;

    float* As = smem + wave * WAVE_LDS;
    float* Ws = As + 32 * LDR;

    // balanced K split across the waves of this workgroup
    const int kbeg = (int)(((long)NK * wave) / NWAVES);
    const int kend = (int)(((long)NK * (wave + 1)) / NWAVES);

    RowInfo ri[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + r * 8 + (lane >> 3);
        ri[r].valid = m < M;
        const int mm = ri[r].valid ? m : 0;
        ri[r].n = mm / HoWo;
        const int rem = mm - ri[r].n * HoWo;
        ri[r].oy = rem / p.Wo;
        ri[r].ox = rem - ri[r].oy * p.Wo;
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    ChunkRegs<NT> R;
    if (kbeg < kend) issue_chunk<NT>(p, kbeg, NK1, cpt, Cin, n0, ri, m0, M, lane, R);

    const int st_off = (lane >> 3) * LDR + (lane & 7) * 4;          // staging store offset
    const int fr_off = (lane & 31) * LDR + (lane >> 5) * 4;         // fragment read offset

    for (int kc = kbeg; kc < kend; ++kc) {
        finish_chunk<NT>(p, R, As, Ws, st_off);
        wave_lds_fence();
        if (kc + 1 < kend) issue_chunk<NT>(p, kc + 1, NK1, cpt, Cin, n0, ri, m0, M, lane, R);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a4 = ld4(As + fr_off + g * 8);
            f32x4 b4[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) b4[t] = ld4(Ws + t * 32 * LDR + fr_off + g * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[t][e], acc[t], 0, 0, 0);
        }
        wave_lds_fence();
    }

    // ---- cross-wave K reduction through LDS (each wave reuses its own staging region) ----
    float* red = smem + wave * WAVE_LDS;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            red[row * RED_LD + t * 32 + (lane & 31)] = acc[t][r];
        }
    __syncthreads();

    const bool nchw = p.out_mode == LFVDM_OUT_NCHW;
    for (int e = threadIdx.x; e < 32 * BN; e += NWAVES * 64) {
        int row, col;
        if (nchw) { col = e >> 5; row = e & 31; } else { row = e / BN; col = e - row * BN; }
        const int m = m0 + row;
        const int co = n0 + col;
        if (m >= M || co >= p.Cout) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) v += smem[w * WAVE_LDS + row * RED_LD + col];
        if (p.bias) v += p.bias[co];
        if (p.bias2) v += p.bias2[co];
        const int n = m / HoWo;
        if (p.res) {
            float rv = p.res[(size_t)m * p.ldr + co];
            if (p.resA) rv = rv * p.resA[(size_t)n * p.Cout + co] + p.resB[(size_t)n * p.Cout + co];
            v += rv;
        }
        if (nchw) p.out[((size_t)n * p.Cout + co) * HoWo + (m - n * HoWo)] = v;
        else p.out[(size_t)m * p.ldo + co] = v;
    }
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

template <int NT, int NWAVES>
int launch_cfg(const lfvdm_conv_args* a, hipStream_t s, int mt, int ntiles) {
    constexpr size_t lds = (size_t)NWAVES * (32 + 32 * NT) * LDR * sizeof(float);
    static bool attr_set = false;  // raising the dynamic-LDS limit is idempotent
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<NT, NWAVES>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return LFVDM_E_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<NT, NWAVES>), dim3(mt, ntiles), dim3(NWAVES * 64), lds, s, *a);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

// (NT, NWAVES) by a small makespan model: 256 CUs, 4 SIMDs, 64 cycles per 32x32x2 MFMA.
void pick_cfg(int Cout, int mt, int NK, int* oNT, int* oNW) {
    int bestNT = 1, bestNW = 1;
    double best = 1e30;
    for (int NT = 1; NT <= 2; ++NT) {
        if (NT == 2 && Cout < 64) continue;
        const int ntiles = (Cout + 32 * NT - 1) / (32 * NT);
        for (int NW = 1; NW <= 16; NW *= 2) {
            if (NW > NK) continue;
            if (NW == 16 && NT == 2) continue;  // LDS budget
            const double chunk = 16.0 * NT * 64.0 + 350.0;
            const double per_wave = (double)((NK + NW - 1) / NW) * chunk;
            const long wgs = (long)mt * ntiles;
            const double waves_per_simd = (double)((wgs + 255) / 256) * ((NW + 3) / 4);
            const double est = waves_per_simd * per_wave + 600.0 + 40.0 * NW + (NT == 1 ? 0.0 : -1.0);
            if (est < best) { best = est; bestNT = NT; bestNW = NW; }
        }
    }
    *oNT = bestNT;
    *oNW = bestNW;
}

}  // namespace

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
    if ((a->coefA == nullptr) != (a->coefB == nullptr)) return LFVDM_E_SHAPE;
    {   // output size must agree with the conv arithmetic the kernel assumes
        const int Hin = a->up ? 2 * a->Hs : a->Hs, Win = a->up ? 2 * a->Ws : a->Ws;
        const int pad = a->ksize == 3 ? 1 : 0;
        if ((Hin + 2 * pad - a->ksize) / a->stride + 1 != a->Ho) return LFVDM_E_SHAPE;
        if ((Win + 2 * pad - a->ksize) / a->stride + 1 != a->Wo) return LFVDM_E_SHAPE;
    }
    const long M = (long)a->N * a->Ho * a->Wo;
    const int mt = (int)((M + 31) / 32);
    const int NK = a->ksize * a->ksize * (Cin / 32) + C2 / 32;

    int bestNT = 1, bestNW = 1;
    pick_cfg(a->Cout, mt, NK, &bestNT, &bestNW);
    const int ntiles = (a->Cout + 32 * bestNT - 1) / (32 * bestNT);
#define LFVDM_CASE(NT_, NW_) if (bestNT == NT_ && bestNW == NW_) return launch_cfg<NT_, NW_>(a, s, mt, ntiles)
    LFVDM_CASE(1, 1); LFVDM_CASE(1, 2); LFVDM_CASE(1, 4); LFVDM_CASE(1, 8); LFVDM_CASE(1, 16);
    LFVDM_CASE(2, 1); LFVDM_CASE(2, 2); LFVDM_CASE(2, 4); LFVDM_CASE(2, 8);
#undef LFVDM_CASE
    return LFVDM_E_UNSUPPORTED;
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
extern "C" int lfvdm_conv_igemm_config(const lfvdm_conv_args* a, int* nt, int* nwaves) {
    const int Cin = a->C0 + a->C1, C2 = a->s2C0 + a->s2C1;
    if (Cin <= 0 || Cin % 32 || C2 % 32) return LFVDM_E_SHAPE;
    const long M = (long)a->N * a->Ho * a->Wo;
    pick_cfg(a->Cout, (int)((M + 31) / 32), a->ksize * a->ksize * (Cin / 32) + C2 / 32, nt, nwaves);
    return LFVDM_OK;
}

extern "C" int lfvdm_abi_version(void) { return 1; }
