// Weight / bias gradients of the implicit-GEMM convolution (and of nn.Linear) on fp32 MFMA, plus the
// transposed-flipped weight packing that turns the data gradient into a forward conv_igemm launch.
//
//   dWp[co][tap*Cin + ci] += sum_m dout[m][co] * f(src)[m, tap, ci]        (packed [Cout][tap][Cin])
//   db[co]                += sum_m dout[m][co]
// f is the same fused prologue as in the forward (GroupNorm/FiLM affine + SiLU, zero padding), so the
// normalised activation is never materialised for the backward either.
//
// Decomposition: the reduction dimension is M (output pixels).  One WAVE owns a 32(co) x 32(k) tile of
// dWp for a slice of M and walks it in 32-row chunks with wave-private LDS staging (no barriers); partial
// tiles of different M slices are combined with float atomics into the (zero-initialised) packed gradient
// (atomic traffic: Cout*K*msplit*4 B per launch, far below the ~1.3 TB/s atomic rate).
// MFMA mapping: D[co][k] += A[co][m] * B[m][k]; A lane (i=co, h) reads dout[m = 8g+4h+e][co] and B lane
// (j=k, h) reads a[m = 8g+4h+e][k] as conflict-free column reads of the row-major LDS tiles.
#include <cstdlib>

#include "common_hip.h"

namespace {

constexpr int WLD = 36;  // padded LDS row (floats)

struct WRow {
    int n, oy, ox, m;
    bool valid;
};

__device__ __forceinline__ int fdiv(int a, int d, float rd) {
    int q = (int)((float)a * rd);
    const int r = a - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

// exact a / d for 0 <= a, d >= 1 with a quotient below 2^21 (v_rcp_f32 is accurate to 1 ulp: the truncated product is off by
// at most one, fdiv's fix-up repairs it).  The kernel prologues are instruction-bound: a generic 32-bit division is ~25
// instructions, a 64-bit one ~150.
__device__ __forceinline__ int qdiv(int a, int d) { return fdiv(a, d, __builtin_amdgcn_rcpf((float)d)); }
// even partition of n chunks over `parts` slices without wide products: slice i = [beg, end)
__device__ __forceinline__ void slice_of(int n, int parts, int i, int& beg, int& end) {
    const int q = qdiv(n, parts), r = n - q * parts;
    beg = i * q + min(i, r);
    end = beg + q + (i < r ? 1 : 0);
}

template <class T>
__device__ __forceinline__ T selv(bool c, T a, T b) {
    return c ? a : b;
}

// args reuse lfvdm_conv_args: src*/C*/N/Hs/Ws/up/stride/ksize/Ho/Wo/coefA/coefB/act describe the forward
// operand; `res` = dout rows [M][ldr] (ldr >= Cout); `out` = packed dW [Cout][taps*Cin] (accumulated);
// `bias` (non-const use) = db [Cout] or NULL.
// one wave task (k tile, co tile, m slice) of a weight-gradient launch; Ds = this wave's private LDS (2 * 32 * WLD floats)
__device__ __forceinline__ void wgrad_wave_task(const lfvdm_conv_args& p, int msplit, long task, float* Ds, int lane) {
    float* As = Ds + 32 * WLD;         // a tile [32 m][32 k]; Ds: dout tile [32 m][32 co]

    const int Cin = p.C0 + p.C1;
    const int taps = p.ksize * p.ksize;
    const int cpt = Cin / 32;
    const int NKT = taps * cpt;                        // k tiles
    const int NCT = (p.Cout + 31) / 32;                // co tiles
    const int HoWo = p.Ho * p.Wo;
    const int M = p.N * HoWo;
    const int nchunks = (M + 31) / 32;

    // wave task = (k tile, co tile, m slice)
    const long ntasks = (long)NKT * NCT * msplit;
    if (task >= ntasks) return;
    const int t1 = qdiv((int)task, msplit);
    const int ms = (int)task - t1 * msplit;
    const int kt = qdiv(t1, NCT);
    const int ct = t1 - kt * NCT;
    const int tap = qdiv(kt, cpt);
    const int cc = (kt - tap * cpt) * 32;
    const int dy = p.ksize == 3 ? tap / 3 - 1 : 0;
    const int dx = p.ksize == 3 ? tap - (tap / 3) * 3 - 1 : 0;
    const bool second = cc >= p.C0;
    const float* src = selv(second, p.src1, p.src0);
    const int Csrc = selv(second, p.C1, p.C0);
    const int cl = second ? cc - p.C0 : cc;
    const int Hin = p.up ? 2 * p.Hs : p.Hs, Win = p.up ? 2 * p.Ws : p.Ws;
    int c_beg, c_end;
    slice_of(nchunks, msplit, ms, c_beg, c_end);
    const int co0 = ct * 32;

    const int col = (lane & 7) * 4;
    const int rsub = lane >> 3;
    const float rHoWo = __builtin_amdgcn_rcpf((float)HoWo), rWo = __builtin_amdgcn_rcpf((float)p.Wo);     // quotients < 2^21
    const float* dout = p.res;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const int st_off = rsub * WLD + col;

    for (int c = c_beg; c < c_end; ++c) {
        const int m0 = c * 32;
        f32x4 dv[4], av[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + r * 8 + rsub;
            const bool valid = m < M;
            const int mm = valid ? m : 0;
            const int n = fdiv(mm, HoWo, rHoWo);
            const int rem = mm - n * HoWo;
            const int oy = fdiv(rem, p.Wo, rWo);
            const int ox = rem - oy * p.Wo;
            // dout tile (zero for rows past M and filters past Cout)
            f32x4 d = zero;
            if (valid && co0 + col < p.Cout) d = ld4(dout + (size_t)mm * p.ldr + co0 + col);
            dv[r] = d;
            // forward operand tile
            const int iy = oy * p.stride + dy, ix = ox * p.stride + dx;
            const bool inb = valid && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
            f32x4 v = zero;
            if (inb) {
                const int sy = p.up ? (iy >> 1) : iy, sx = p.up ? (ix >> 1) : ix;
                v = ld4(src + ((size_t)(n * p.Hs + sy) * p.Ws + sx) * Csrc + cl + col);
                if (p.coefA) v = v * ld4(p.coefA + (size_t)n * Cin + cc + col) + ld4(p.coefB + (size_t)n * Cin + cc + col);
                if (p.act == LFVDM_ACT_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
            }
            av[r] = v;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bsum += dv[r];
            st4(Ds + r * 8 * WLD + st_off, dv[r]);
            st4(As + r * 8 * WLD + st_off, av[r]);
        }
        wave_lds_fence();
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int mrow = 8 * g + 4 * (lane >> 5) + e;
                const float a = Ds[mrow * WLD + (lane & 31)];
                const float b = As[mrow * WLD + (lane & 31)];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        wave_lds_fence();
    }
    // ---- accumulate the partial tile: D lane l holds column k = l&31, rows co = (r&3)+8*(r>>2)+4*(l>>5)
    // (float atomics, or - deterministic mode, splitk_ws = slab - plain stores to this M slice's slab row)
    const int Ktot = taps * Cin;
    float* dW = p.out;
    const bool oihw = p.out_mode == 1;   // accumulate straight into the OIHW parameter gradient
    const DetSlab dsW = {p.splitk_ws, p.out, (long)p.Cout * Ktot};
    const DetSlab dsB = {p.splitk_ws ? p.splitk_ws + (size_t)msplit * dsW.n : nullptr, p.bias, (long)p.Cout};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < p.Cout) {
            const int ci = cc + (lane & 31);
            float* dst = oihw ? dW + ((size_t)co * Cin + ci) * taps + tap : dW + (size_t)co * Ktot + (size_t)tap * Cin + ci;
            det_add(dsW, ms, dst, acc[r]);
        }
    }
    if (kt == 0 && p.bias != nullptr) {
        // bias gradient: column sums of this wave's dout rows (lanes with equal col differ in bits 3..5)
        float* db = const_cast<float*>(p.bias);
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
            bsum.x += __shfl_xor(bsum.x, o, 64); bsum.y += __shfl_xor(bsum.y, o, 64);
            bsum.z += __shfl_xor(bsum.z, o, 64); bsum.w += __shfl_xor(bsum.w, o, 64);
        }
        if (lane < 8) {     // every column guarded on its own: Cout need not be a multiple of 4 (pixel-space head: 3)
            const int c = co0 + col;
            if (c + 0 < p.Cout) det_add(dsB, ms, db + c + 0, bsum.x);
            if (c + 1 < p.Cout) det_add(dsB, ms, db + c + 1, bsum.y);
            if (c + 2 < p.Cout) det_add(dsB, ms, db + c + 2, bsum.z);
            if (c + 3 < p.Cout) det_add(dsB, ms, db + c + 3, bsum.w);
        }
    }
}

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const lfvdm_conv_args p_in, int msplit) {
    const lfvdm_conv_args p = p_in;
    __shared__ __attribute__((aligned(16))) float smem[4][2 * 32 * WLD];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    wgrad_wave_task(p, msplit, (long)blockIdx.x * 4 + wave, smem[wave], lane);
}

// Grouped form: the waves of a workgroup may belong to different jobs (wave-private LDS, no workgroup barrier);
// the job's arguments are read from the device table with scalar loads.
__global__ __launch_bounds__(256) void conv_wgrad_grouped_kernel(const lfvdm_wgrad_job* __restrict__ jobs, int njobs,
                                                                 int total_tasks) {
    __shared__ __attribute__((aligned(16))) float smem[4][2 * 32 * WLD];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int task = (int)blockIdx.x * 4 + wave;
    if (task >= total_tasks) return;
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                       // last job with task0 <= task
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].task0 <= task) lo = mid; else hi = mid - 1;
    }
    const lfvdm_conv_args p = jobs[lo].a;
    wgrad_wave_task(p, jobs[lo].msplit, (long)(task - jobs[lo].task0), smem[wave], lane);
}

// --------------------------------------------------------------------------------------------------------
// Cooperative variant for the wide layers: a 4-wave workgroup owns COT x KT 32x32 tiles of dW (COT*32 filters
// x KT*32 input channels of one tap) for a slice of M.  Per 32-row chunk the dout tile [32][COT*32] and the
// fused-prologue operand tile [32][KT*32] are staged ONCE in LDS and shared by all waves (the wave-private
// kernel above re-reads every dout row once per k tile and every operand row once per filter tile from L2);
// wave (wc, wk) multiplies the (COT/2) x (KT/2) tiles of its quadrant, re-using each LDS fragment KT/2 resp.
// COT/2 times.  The global loads of chunk c+1 are in flight while chunk c is on the MFMA pipe.
template <int COT, int KT>
__global__ __launch_bounds__(256) void conv_wgrad_coop_kernel(const lfvdm_conv_args p_in, int msplit) {
    const lfvdm_conv_args p = p_in;
    constexpr int DLD = COT * 32 + 4, ALD = KT * 32 + 4;
    constexpr int DQ = COT * 8, AQ = KT * 8;          // float4 per tile row
    constexpr int ND = COT, NA = KT;                  // float4 per thread per chunk
    constexpr int TC = COT / 2, TK = KT / 2;          // tiles per wave along co / k
    __shared__ __attribute__((aligned(16))) float Ds[32 * DLD];
    __shared__ __attribute__((aligned(16))) float As[32 * ALD];
    __shared__ float bias_red[8][COT * 32];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 1, wk = wave >> 1;

    const int Cin = p.C0 + p.C1;
    const int taps = p.ksize * p.ksize;
    const int cpg = Cin / (32 * KT);                   // channel groups per tap
    const int NKG = taps * cpg;
    const int NCG = (p.Cout + 32 * COT - 1) / (32 * COT);
    const int HoWo = p.Ho * p.Wo;
    const int M = p.N * HoWo;
    const int nchunks = (M + 31) / 32;

    const int task = blockIdx.x;                       // (k group, co group, m slice)
    const int t1 = qdiv(task, msplit);
    const int ms = task - t1 * msplit;
    const int kg = qdiv(t1, NCG);
    const int cg = t1 - kg * NCG;
    const int tap = qdiv(kg, cpg);
    const int cc = (kg - tap * cpg) * 32 * KT;         // first input channel of the group
    const int dy = p.ksize == 3 ? tap / 3 - 1 : 0;
    const int dx = p.ksize == 3 ? tap - (tap / 3) * 3 - 1 : 0;
    const bool second = cc >= p.C0;
    const float* src = selv(second, p.src1, p.src0);
    const int Csrc = selv(second, p.C1, p.C0);
    const int cl = second ? cc - p.C0 : cc;
    const int Hin = p.up ? 2 * p.Hs : p.Hs, Win = p.up ? 2 * p.Ws : p.Ws;
    int c_beg, c_end;
    slice_of(nchunks, msplit, ms, c_beg, c_end);
    const int co0 = cg * 32 * COT;
    const float rHoWo = __builtin_amdgcn_rcpf((float)HoWo), rWo = __builtin_amdgcn_rcpf((float)p.Wo);     // quotients < 2^21
    const float* dout = p.res;
    const bool has_coef = p.coefA != nullptr;
    const bool silu = p.act == LFVDM_ACT_SILU;

    // staging slots of this thread: dout float4 (row dr, col dcol) x ND, operand float4 (row ar, col acol) x NA
    const int dcol = (tid % DQ) * 4, drow0 = tid / DQ;        // rows drow0 + i * (256 / DQ)
    const int acol = (tid % AQ) * 4, arow0 = tid / AQ;
    constexpr int DRS = 256 / DQ, ARS = 256 / AQ;
    const bool dcol_ok = co0 + dcol < p.Cout;

    f32x16 acc[TC][TK];
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    f32x4 dv[ND], av[NA], ca[NA], cb[NA];
    unsigned inb_mask = 0;
    auto issue = [&](int c) {
        const int m0 = c * 32;
        inb_mask = 0;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int m = m0 + drow0 + i * DRS;
            const bool ok = m < M && dcol_ok;
            dv[i] = ld4(dout + (ok ? (size_t)m * p.ldr + co0 + dcol : 0));
            if (!ok) dv[i] = zero;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m0 + arow0 + i * ARS;
            const bool valid = m < M;
            const int mm = valid ? m : 0;
            const int n = fdiv(mm, HoWo, rHoWo);
            const int rem = mm - n * HoWo;
            const int oy = fdiv(rem, p.Wo, rWo);
            const int ox = rem - oy * p.Wo;
            const int iy = oy * p.stride + dy, ix = ox * p.stride + dx;
            const bool inb = valid && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
            const int sy = p.up ? (iy >> 1) : iy, sx = p.up ? (ix >> 1) : ix;
            av[i] = ld4(src + (inb ? ((size_t)(n * p.Hs + sy) * p.Ws + sx) * Csrc + cl + acol : 0));
            if (has_coef) {
                ca[i] = ld4(p.coefA + (size_t)n * Cin + cc + acol);
                cb[i] = ld4(p.coefB + (size_t)n * Cin + cc + acol);
            }
            inb_mask |= inb ? (1u << i) : 0u;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            bsum += dv[i];
            st4(Ds + (drow0 + i * DRS) * DLD + dcol, dv[i]);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            f32x4 v = av[i];
            if (has_coef) v = v * ca[i] + cb[i];
            if (silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
            if (!(inb_mask & (1u << i))) v = zero;
            st4(As + (arow0 + i * ARS) * ALD + acol, v);
        }
    };

    if (c_beg < c_end) issue(c_beg);
    for (int c = c_beg; c < c_end; ++c) {
        __syncthreads();            // previous chunk consumed
        commit();
        __syncthreads();
        if (c + 1 < c_end) issue(c + 1);
        const float* dcolp = Ds + wc * TC * 32 + (lane & 31);
        const float* acolp = As + wk * TK * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int mrow = 8 * g + 4 * (lane >> 5) + e;
                float a[TC], b[TK];
#pragma unroll
                for (int x = 0; x < TC; ++x) a[x] = dcolp[mrow * DLD + 32 * x];
#pragma unroll
                for (int y = 0; y < TK; ++y) b[y] = acolp[mrow * ALD + 32 * y];
#pragma unroll
                for (int x = 0; x < TC; ++x)
#pragma unroll
                    for (int y = 0; y < TK; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x], b[y], acc[x][y], 0, 0, 0);
            }
    }
    // ---- accumulate the partial tiles (float atomics; D lane l: column k = l&31, rows (r&3)+8*(r>>2)+4*(l>>5))
    const int Ktot = taps * Cin;
    float* dW = p.out;
    const bool oihw = p.out_mode == 1;
    const DetSlab dsW = {p.splitk_ws, p.out, (long)p.Cout * Ktot};      // deterministic mode: slab row ms instead of atomics
    const DetSlab dsB = {p.splitk_ws ? p.splitk_ws + (size_t)msplit * dsW.n : nullptr, p.bias, (long)p.Cout};
#pragma unroll
    for (int x = 0; x < TC; ++x)
#pragma unroll
        for (int y = 0; y < TK; ++y) {
            const int ci = cc + (wk * TK + y) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wc * TC + x) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < p.Cout) {
                    float* dst = oihw ? dW + ((size_t)co * Cin + ci) * taps + tap : dW + (size_t)co * Ktot + (size_t)tap * Cin + ci;
                    det_add(dsW, ms, dst, acc[x][y][r]);
                }
            }
        }
    if (kg == 0 && p.bias != nullptr) {     // bias gradient: column sums of this slice's dout rows, once per co group
        __syncthreads();
        float* br = &bias_red[tid / DQ % 8][0];
        if (tid / DQ < 8) { br[dcol + 0] = bsum.x; br[dcol + 1] = bsum.y; br[dcol + 2] = bsum.z; br[dcol + 3] = bsum.w; }
        __syncthreads();
        // rows of bias_red beyond the first 8 thread rows are folded in sequentially
        for (int rr = 8; rr < DRS; rr += 8) {
            if (tid / DQ >= rr && tid / DQ < rr + 8) { br[dcol + 0] += bsum.x; br[dcol + 1] += bsum.y; br[dcol + 2] += bsum.z; br[dcol + 3] += bsum.w; }
            __syncthreads();
        }
        if (tid < COT * 32 && co0 + tid < p.Cout) {
            float t = 0.f;
            const int nr = DRS < 8 ? DRS : 8;
            for (int r = 0; r < nr; ++r) t += bias_red[r][tid];
            det_add(dsB, ms, const_cast<float*>(p.bias) + co0 + tid, t);
        }
    }
}

// --------------------------------------------------------------------------------------------------------
// LDS-DMA form of the cooperative kernel for raw operands (no GroupNorm coefficients - the default training
// plan materialises the activated tensors): the dout tile [32][COT*32] and the operand tile [32][KT*32] of a chunk
// are written straight into LDS by `buffer_load_dwordx4 ... lds` (64 lanes x 16 B = one contiguous 1 KiB piece per
// wave instruction; per-lane source offsets; rows past M, filters past Cout and taps outside the image pass an
// out-of-range offset and arrive as zeros).  The tiles are plain unpadded row-major images: the MFMA operands
// are COLUMN reads (32 consecutive floats of one row per half wave), conflict-free without padding or swizzle.
// NS stages, chunk c+NS-1 in flight while chunk c is multiplied, one barrier per chunk: counted vmcnt (this
// wave's pieces of the chunk have landed) + lgkmcnt(0) (its reads of the buffer that is restaged next have returned).
// CR = rows of M per staged chunk (32 or 64: twice the MFMAs per barrier and per round of address arithmetic - what the
// narrow tiles of the 16x16 / 8x8 latent levels are short of: 16 MFMAs per wave and chunk at COT = KT = 2, CR = 32).
template <int COT, int KT, int NS, int CR = 32>
__global__ __launch_bounds__(256) void conv_wgrad_dma_kernel(const lfvdm_conv_args p_in, int msplit) {
    const lfvdm_conv_args p = p_in;
    constexpr int DLD = COT * 32, ALD = KT * 32;
    constexpr int DQ = COT * 8, AQ = KT * 8;          // float4 per tile row
    constexpr int ND = COT * CR / 32, NA = KT * CR / 32;   // pieces per thread per chunk
    constexpr int TC = COT / 2, TK = KT / 2;          // tiles per wave along co / k
    constexpr int STAGE = CR * DLD + CR * ALD;        // floats
    constexpr unsigned kOOB = 0x40000000u;            // >= num_records of both descriptors (checked by the launcher)
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float* bias_red = wsm + NS * STAGE;               // [8][COT * 32]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 1, wk = wave >> 1;

    const int Cin = p.C0 + p.C1;
    const int taps = p.ksize * p.ksize;
    const int cpg = Cin / (32 * KT);
    const int NCG = (p.Cout + 32 * COT - 1) / (32 * COT);
    const int HoWo = p.Ho * p.Wo;
    const int M = p.N * HoWo;
    const int nchunks = (M + CR - 1) / CR;

    const int task = blockIdx.x;                       // (k group, co group, m slice)
    const int t1 = qdiv(task, msplit);
    const int ms = task - t1 * msplit;
    const int kg = qdiv(t1, NCG);
    const int cg = t1 - kg * NCG;
    const int tap = qdiv(kg, cpg);
    const int cc = (kg - tap * cpg) * 32 * KT;
    const int dy = p.ksize == 3 ? tap / 3 - 1 : 0;
    const int dx = p.ksize == 3 ? tap - (tap / 3) * 3 - 1 : 0;
    const bool second = cc >= p.C0;
    const float* src = selv(second, p.src1, p.src0);
    const int Csrc = selv(second, p.C1, p.C0);
    const int cl = second ? cc - p.C0 : cc;
    const int Hin = p.up ? 2 * p.Hs : p.Hs, Win = p.up ? 2 * p.Ws : p.Ws;
    int c_beg, c_end;
    slice_of(nchunks, msplit, ms, c_beg, c_end);
    const int co0 = cg * 32 * COT;
    const float rHoWo = __builtin_amdgcn_rcpf((float)HoWo), rWo = __builtin_amdgcn_rcpf((float)p.Wo);     // quotients < 2^21
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, (int)((unsigned)M * p.ldr * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)src, 0, (int)((unsigned)p.N * p.Hs * p.Ws * Csrc * 4u), 0x00020000);

    const int dcol = (tid % DQ) * 4, drow0 = tid / DQ;
    const int acol = (tid % AQ) * 4, arow0 = tid / AQ;
    constexpr int DRS = 256 / DQ, ARS = 256 / AQ;
    const unsigned dbase = (co0 + dcol < p.Cout) ? (unsigned)(co0 + dcol) * 4u : kOOB;
    const unsigned abase = (unsigned)(cl + acol) * 4u;
    const int upsh = p.up ? 1 : 0;

    f32x16 acc[TC][TK];
#pragma unroll
    for (int a = 0; a < TC; ++a)
#pragma unroll
        for (int b = 0; b < TK; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const bool want_bias = kg == 0 && p.bias != nullptr;    // workgroup-uniform

    auto issue = [&](int c, int stage) {
        const int m0 = c * CR;
        const bool live = c < c_end;
        float* Ds = wsm + stage * STAGE + wave * 256;          // this wave's first piece
        float* As = wsm + stage * STAGE + CR * DLD + wave * 256;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int m = m0 + drow0 + i * DRS;
            const unsigned off = (live && m < M) ? dbase + (unsigned)m * p.ldr * 4u : kOOB;   // dbase >= kOOB: filter past Cout
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (__attribute__((address_space(3))) void*)(Ds + i * 1024), 16, (int)off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m0 + arow0 + i * ARS;
            const bool valid = live && m < M;
            const int mm = valid ? m : 0;
            const int n = fdiv(mm, HoWo, rHoWo);
            const int rem = mm - n * HoWo;
            const int oy = fdiv(rem, p.Wo, rWo);
            const int ox = rem - oy * p.Wo;
            const int iy = oy * p.stride + dy, ix = ox * p.stride + dx;
            const bool inb = valid && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
            const int sy = iy >> upsh, sx = ix >> upsh;
            const unsigned off = inb ? abase + (unsigned)((n * p.Hs + sy) * p.Ws + sx) * Csrc * 4u : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(As + i * 1024), 16, (int)off, 0, 0, 0);
        }
    };

#pragma unroll
    for (int d = 0; d < NS - 1; ++d) issue(c_beg + d, d);
    int stage = 0;
    for (int c = c_beg; c < c_end; ++c) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * (ND + NA)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int nxt = stage + NS - 1;
        nxt = nxt >= NS ? nxt - NS : nxt;
        issue(c + NS - 1, nxt);
        const float* Dst = wsm + stage * STAGE;
        const float* Ast = Dst + CR * DLD;
        if (want_bias) {        // column sums of this slice's dout rows (each thread re-reads the slots of its own pieces)
#pragma unroll
            for (int i = 0; i < ND; ++i) bsum += ld4(Dst + (drow0 + i * DRS) * DLD + dcol);
        }
        const float* dcolp = Dst + wc * TC * 32 + (lane & 31);
        const float* acolp = Ast + wk * TK * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < CR / 8; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int mrow = 8 * g + 4 * (lane >> 5) + e;
                float a[TC], b[TK];
#pragma unroll
                for (int x = 0; x < TC; ++x) a[x] = dcolp[mrow * DLD + 32 * x];
#pragma unroll
                for (int y = 0; y < TK; ++y) b[y] = acolp[mrow * ALD + 32 * y];
#pragma unroll
                for (int x = 0; x < TC; ++x)
#pragma unroll
                    for (int y = 0; y < TK; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x], b[y], acc[x][y], 0, 0, 0);
            }
        stage = stage + 1 == NS ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // zero-filled look-ahead pieces
    // ---- accumulate the partial tiles (float atomics; D lane l: column k = l&31, rows (r&3)+8*(r>>2)+4*(l>>5))
    const int Ktot = taps * Cin;
    float* dW = p.out;
    const bool oihw = p.out_mode == 1;
    const DetSlab dsW = {p.splitk_ws, p.out, (long)p.Cout * Ktot};      // deterministic mode: slab row ms instead of atomics
    const DetSlab dsB = {p.splitk_ws ? p.splitk_ws + (size_t)msplit * dsW.n : nullptr, p.bias, (long)p.Cout};
#pragma unroll
    for (int x = 0; x < TC; ++x)
#pragma unroll
        for (int y = 0; y < TK; ++y) {
            const int ci = cc + (wk * TK + y) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + (wc * TC + x) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < p.Cout) {
                    float* dst = oihw ? dW + ((size_t)co * Cin + ci) * taps + tap : dW + (size_t)co * Ktot + (size_t)tap * Cin + ci;
                    det_add(dsW, ms, dst, acc[x][y][r]);
                }
            }
        }
    if (want_bias) {
        __syncthreads();
        constexpr int BRL = COT * 32;
        float* br = bias_red + (tid / DQ % 8) * BRL;
        if (tid / DQ < 8) { br[dcol + 0] = bsum.x; br[dcol + 1] = bsum.y; br[dcol + 2] = bsum.z; br[dcol + 3] = bsum.w; }
        __syncthreads();
        for (int rr = 8; rr < DRS; rr += 8) {
            if (tid / DQ >= rr && tid / DQ < rr + 8) { br[dcol + 0] += bsum.x; br[dcol + 1] += bsum.y; br[dcol + 2] += bsum.z; br[dcol + 3] += bsum.w; }
            __syncthreads();
        }
        if (tid < COT * 32 && co0 + tid < p.Cout) {
            float t = 0.f;
            const int nr = DRS < 8 ? DRS : 8;
            for (int r = 0; r < nr; ++r) t += bias_red[r * BRL + tid];
            det_add(dsB, ms, const_cast<float*>(p.bias) + co0 + tid, t);
        }
    }
}

// --------------------------------------------------------------------------------------------------------
// Tap-fused form for 3x3 / stride 1 layers on raw operands: a workgroup owns a 64-filter x 64-channel tile of dW for
// THREE taps (one filter row dy, NTY = 1) or all NINE (NTY = 3).  The kernels above stage, for every tap separately, the
// dout tile and the operand tile shifted by that tap: 16 FLOP per staged byte, and on the cfg-C layers (M = 10240) the
// 36 (tap, channel group, filter group) tiles of a 128 -> 128 layer pull 189 MB through L2 for 10 MB of tensors - L2 ->
// LDS bandwidth, not MFMA issue, bounds them (52 TFLOP/s).  Here a chunk of 32 consecutive output pixels (R = 32 / WSEG
// image rows of WSEG pixels; WSEG = 32 for maps at least 32 wide) is staged ONCE with its halo - dout [32][64] plus the
// operand window [(R | R + 2) x (WSEG + 2) pixels][64] - and every tap reads its operand fragment from the window at a
// compile-time offset: 2.8x (NTY = 1) / 5.5x (NTY = 3) fewer staged bytes per FLOP, 48 / 144 MFMAs per wave between
// barriers instead of 16.  Price: 3 / 9 accumulator tiles per wave (48 / 144 VGPRs) and 3x / 9x the atomics per M slice,
// which is why the tuner picks NTY per layer shape (NTY = 3: pixel-space maps, NTY = 1: the 16x16 / 8x8 latent levels).
// LDS-DMA staging (zero fill outside the image / past M by out-of-range offsets), two stages, one barrier per chunk.
template <int NTY, int WSEG>
__global__ __launch_bounds__(256) void conv_wgrad_taps_kernel(const lfvdm_conv_args p_in, int msplit) {
    const lfvdm_conv_args p = p_in;
    constexpr int R = 32 / WSEG;                       // image rows per chunk
    constexpr int WC = WSEG + 2;                       // window columns (halo of one pixel either side)
    constexpr int WR = NTY == 3 ? R + 2 : R;           // window rows
    constexpr int WP = WR * WC;                        // window pixels
    constexpr int NA = (WP + 15) / 16;                 // rounds of 16 pixels (4 waves x 4 pixels of 64 channels)
    constexpr int ND = 2;                              // dout [32][64]: 16 rows per round
    constexpr int NTAP = 3 * NTY;
    constexpr int STAGE = 32 * 64 + NA * 16 * 64;      // floats
    constexpr unsigned kOOB = 0x40000000u;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float* bias_red = wsm + 2 * STAGE;                 // [8][64]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 1, wk = wave >> 1;

    const int Cin = p.C0 + p.C1;
    const int cpg = Cin / 64;
    const int NCG = (p.Cout + 63) / 64;
    const int HoWo = p.Ho * p.Wo;
    const int M = p.N * HoWo;
    const int nchunks = M / 32;                        // (HoWo % 32 == 0: checked by the launcher)
    const int segs = p.Wo / WSEG;                      // chunks per image row (1 when the map is narrower than 32)

    const int task = blockIdx.x;                       // ((dy row, channel group), filter group, m slice)
    const int t1 = qdiv(task, msplit);
    const int ms = task - t1 * msplit;
    const int kg = qdiv(t1, NCG);
    const int cg = t1 - kg * NCG;
    const int ty0 = NTY == 3 ? 0 : qdiv(kg, cpg);      // first filter row of this workgroup
    const int cc = (kg - (NTY == 3 ? 0 : ty0 * cpg)) * 64;
    const bool second = cc >= p.C0;
    const float* src = selv(second, p.src1, p.src0);
    const int Csrc = selv(second, p.C1, p.C0);
    const int cl = second ? cc - p.C0 : cc;
    int c_beg, c_end;
    slice_of(nchunks, msplit, ms, c_beg, c_end);
    const int co0 = cg * 64;
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, (int)((unsigned)M * p.ldr * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)src, 0, (int)((unsigned)p.N * p.Hs * p.Ws * Csrc * 4u), 0x00020000);

    const int q16 = (tid & 15) * 4;                    // channel / filter quad of this thread's pieces
    const int prow = tid >> 4;                         // 0..15: dout row / window pixel inside a round
    const unsigned dbase = (co0 + q16 < p.Cout) ? (unsigned)(co0 + q16) * 4u : kOOB;
    const unsigned abase = (unsigned)(cl + q16) * 4u;
    // window pixel -> (wy, wx) of this thread's NA pieces: compile-time divisor
    int wy[NA], wx[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int w = prow + 16 * i;
        wy[i] = w / WC;
        wx[i] = w - wy[i] * WC;
    }
    const int dy_first = NTY == 3 ? -1 : ty0 - 1;      // image-row offset of window row 0
    const float rHoWo = __builtin_amdgcn_rcpf((float)HoWo);

    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const bool want_bias = kg == 0 && p.bias != nullptr;    // workgroup-uniform

    auto issue = [&](int c, int stage) {
        const bool live = c < c_end;
        const int m0 = c * 32;
        // chunk -> (sample, first image row, first column): scalar arithmetic (c is wave-uniform)
        const int n = fdiv(m0, HoWo, rHoWo);
        const int rem = m0 - n * HoWo;
        int oy0, ox0;
        if (WSEG == 32) { const int rowi = qdiv(rem, p.Wo); oy0 = rowi; ox0 = rem - rowi * p.Wo; (void)segs; }
        else { oy0 = rem / WSEG; ox0 = 0; }
        float* Ds = wsm + stage * STAGE + wave * 256;
        float* Ws = wsm + stage * STAGE + 32 * 64 + wave * 256;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int m = m0 + prow + 16 * i;
            const unsigned off = live ? dbase + (unsigned)m * p.ldr * 4u : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (__attribute__((address_space(3))) void*)(Ds + i * 1024), 16, (int)off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int iy = oy0 + wy[i] + dy_first, ix = ox0 + wx[i] - 1;
            const bool inb = live && (prow + 16 * i < WP) && iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws;
            const unsigned off = inb ? abase + (unsigned)((n * p.Hs + iy) * p.Ws + ix) * Csrc * 4u : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(Ws + i * 1024), 16, (int)off, 0, 0, 0);
        }
    };

    issue(c_beg, 0);
    int stage = 0;
    const int h4 = 4 * (lane >> 5);
    for (int c = c_beg; c < c_end; ++c) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(c + 1, stage ^ 1);
        const float* Dst = wsm + stage * STAGE;
        const float* Wst = Dst + 32 * 64;
        if (want_bias) {
#pragma unroll
            for (int i = 0; i < ND; ++i) bsum += ld4(Dst + (prow + 16 * i) * 64 + q16);
        }
        // lane base: output pixel 8g + 4h + e of the chunk, h = lane >> 5.  WSEG is a multiple of 8, so (8g + 4h + e) and
        // (8g + e) lie in the same image row: the h term is +4 window pixels; everything else is a compile-time offset
        const float* dcolp = Dst + h4 * 64 + wc * 32 + (lane & 31);
        const float* wcolp = Wst + h4 * 64 + wk * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 8 * g + e;                       // (compile-time after unrolling)
                const int jr = j / WSEG, jc = j - jr * WSEG;
                const float a = dcolp[j * 64];
                float b[NTAP];
#pragma unroll
                for (int ty = 0; ty < NTY; ++ty)
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) b[ty * 3 + tx] = wcolp[((jr + ty) * WC + jc + tx) * 64];
#pragma unroll
                for (int t = 0; t < NTAP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
            }
        stage ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the zero-filled look-ahead pieces
    // ---- accumulate the partial tiles (float atomics, or slab row ms in deterministic mode)
    const int Ktot = 9 * Cin;
    float* dW = p.out;
    const DetSlab dsW = {p.splitk_ws, p.out, (long)p.Cout * Ktot};
    const DetSlab dsB = {p.splitk_ws ? p.splitk_ws + (size_t)msplit * dsW.n : nullptr, p.bias, (long)p.Cout};
    const int ci = cc + wk * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        const int tap = (NTY == 3 ? 0 : ty0 * 3) + t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (co < p.Cout) det_add(dsW, ms, dW + (size_t)co * Ktot + (size_t)tap * Cin + ci, acc[t][r]);
        }
    }
    if (want_bias) {
        __syncthreads();
        float* br = bias_red + (prow & 7) * 64;
        if (prow < 8) { br[q16 + 0] = bsum.x; br[q16 + 1] = bsum.y; br[q16 + 2] = bsum.z; br[q16 + 3] = bsum.w; }
        __syncthreads();
        if (prow >= 8) { br[q16 + 0] += bsum.x; br[q16 + 1] += bsum.y; br[q16 + 2] += bsum.z; br[q16 + 3] += bsum.w; }
        __syncthreads();
        if (tid < 64 && co0 + tid < p.Cout) {
            float t = 0.f;
            for (int r = 0; r < 8; ++r) t += bias_red[r * 64 + tid];
            det_add(dsB, ms, const_cast<float*>(p.bias) + co0 + tid, t);
        }
    }
}

// raw operands whose tensors are addressable with 32-bit byte offsets below 2^30: LDS-DMA kernel
static bool wgrad_dma_ok(const lfvdm_conv_args* a, long M) {
    const bool off = getenv("LFVDM_WGRAD_NO_DMA") != nullptr;             // A/B aid (read per launch: tests toggle it)
    const long Cmax = a->C0 > a->C1 ? a->C0 : a->C1, lim = 1L << 30;
    return !off && !a->coefA && M * a->ldr * 4 < lim && (long)a->N * a->Hs * a->Ws * Cmax * 4 < lim;
}

// Deterministic mode (lfvdm_conv_args.splitk_ws = slab of splitk_ws_floats floats): every M slice stores its partial
// dW (and db) to its own slab row - each (slice, element) exactly once: all (tile, slice) workgroups exist - and
// det_finish adds the rows in slice order.  det_fit shrinks the number of slices to what the slab holds.
static int det_fit(const lfvdm_conv_args* a, long& msplit) {
    if (!a->splitk_ws) return LFVDM_OK;
    const long per = (long)a->Cout * a->ksize * a->ksize * (a->C0 + a->C1) + a->Cout;
    if (a->splitk_ws_floats < per) return LFVDM_E_SHAPE;
    if (msplit * per > a->splitk_ws_floats) msplit = a->splitk_ws_floats / per;
    return LFVDM_OK;
}
static int det_finish(const lfvdm_conv_args* a, long msplit, hipStream_t s) {
    if (!a->splitk_ws) return LFVDM_OK;
    if (hipGetLastError() != hipSuccess) return LFVDM_E_LAUNCH;
    const long n = (long)a->Cout * a->ksize * a->ksize * (a->C0 + a->C1);
    if (a->bias)        // weight and bias gradient in one launch
        return lfvdm_det_reduce2_launch(a->out, a->splitk_ws, n, const_cast<float*>(a->bias), a->splitk_ws + (size_t)msplit * n,
                                        a->Cout, msplit, s);
    return lfvdm_det_reduce_launch(a->out, a->splitk_ws, n, msplit, s);
}

template <int COT, int KT, int NS, int CR = 32>
static int launch_wgrad_dma(const lfvdm_conv_args* a, hipStream_t s, int nchunks32, long msplit_req) {
    const int nchunks = (nchunks32 * 32 + CR - 1) / CR;
    const int Cin = a->C0 + a->C1;
    const int NKG = a->ksize * a->ksize * (Cin / (32 * KT));
    const int NCG = (a->Cout + 32 * COT - 1) / (32 * COT);
    const long tiles = (long)NKG * NCG;
    static const long target = getenv("LFVDM_WGRAD_WGS") ? atol(getenv("LFVDM_WGRAD_WGS")) : 384;
    long msplit = msplit_req > 0 ? msplit_req : (target + tiles - 1) / tiles;      // (tuned per layer shape, or 1.5 per CU)
    if (msplit > nchunks / 2) msplit = nchunks / 2;
    if (msplit < 1) msplit = 1;
    constexpr size_t lds = (size_t)(NS * (CR * COT * 32 + CR * KT * 32) + 8 * COT * 32) * sizeof(float);
    static_assert(lds <= 160 * 1024, "stages must fit the CU's LDS");
    static DynLdsLimit limit;
    if (int rc = limit.ensure(reinterpret_cast<const void*>(&conv_wgrad_dma_kernel<COT, KT, NS, CR>), lds)) return rc;
    if (int rc = det_fit(a, msplit)) return rc;
    hipLaunchKernelGGL((conv_wgrad_dma_kernel<COT, KT, NS, CR>), dim3((unsigned)(tiles * msplit)), dim3(256), lds, s, *a, (int)msplit);
    return det_finish(a, msplit, s);
}

// which map widths the tap-fused kernel is instantiated for (0: not eligible)
static int wgrad_taps_wseg(const lfvdm_conv_args* a, long M) {
    static const bool off = getenv("LFVDM_WGRAD_NO_TAPS") != nullptr;          // A/B aid
    const int Cin = a->C0 + a->C1;
    if (off || a->ksize != 3 || a->stride != 1 || a->up != 0 || a->coefA || a->act != LFVDM_ACT_NONE) return 0;
    if (a->out_mode != 0) return 0;       // the tap-fused kernels only write the packed [Cout][tap][Cin] accumulator layout
    if (Cin % 64 || a->C0 % 64 || a->Cout < 64 || a->Hs != a->Ho || a->Ws != a->Wo || (a->Ho * a->Wo) % 32) return 0;
    if (!wgrad_dma_ok(a, M)) return 0;
    if (a->Wo % 32 == 0) return 32;
    if (a->Wo == 16 || a->Wo == 8) return a->Wo;
    return 0;
}

template <int NTY, int WSEG>
static int launch_wgrad_taps(const lfvdm_conv_args* a, hipStream_t s, long M, long msplit_req) {
    const int Cin = a->C0 + a->C1;
    const int nchunks = (int)(M / 32);
    const long tiles = (long)(NTY == 3 ? 1 : 3) * (Cin / 64) * ((a->Cout + 63) / 64);
    long msplit = msplit_req > 0 ? msplit_req : (256 + tiles - 1) / tiles;
    if (msplit > nchunks / 2) msplit = nchunks / 2;
    if (msplit < 1) msplit = 1;
    constexpr int WPv = (NTY == 3 ? 32 / WSEG + 2 : 32 / WSEG) * (WSEG + 2);
    constexpr int NAv = (WPv + 15) / 16;
    constexpr size_t lds = (size_t)(2 * (32 * 64 + NAv * 16 * 64) + 8 * 64) * sizeof(float);
    static DynLdsLimit limit;
    if (int rc = limit.ensure(reinterpret_cast<const void*>(&conv_wgrad_taps_kernel<NTY, WSEG>), lds)) return rc;
    if (int rc = det_fit(a, msplit)) return rc;
    hipLaunchKernelGGL((conv_wgrad_taps_kernel<NTY, WSEG>), dim3((unsigned)(tiles * msplit)), dim3(256), lds, s, *a, (int)msplit);
    return det_finish(a, msplit, s);
}

template <int COT, int KT>
static int launch_wgrad_coop(const lfvdm_conv_args* a, hipStream_t s, int nchunks) {
    const int Cin = a->C0 + a->C1;
    const int NKG = a->ksize * a->ksize * (Cin / (32 * KT));
    const int NCG = (a->Cout + 32 * COT - 1) / (32 * COT);
    const long tiles = (long)NKG * NCG;
    static const long target = getenv("LFVDM_WGRAD_WGS") ? atol(getenv("LFVDM_WGRAD_WGS")) : 384;
    long msplit = (target + tiles - 1) / tiles;        // 1.5 workgroups per CU measured best (tile traffic vs bytes of atomics)
    if (msplit > nchunks / 2) msplit = nchunks / 2;    // at least two chunks per slice (the prefetch needs a successor)
    if (msplit < 1) msplit = 1;
    if (int rc = det_fit(a, msplit)) return rc;
    hipLaunchKernelGGL((conv_wgrad_coop_kernel<COT, KT>), dim3((unsigned)(tiles * msplit)), dim3(256), 0, s, *a, (int)msplit);
    return det_finish(a, msplit, s);
}

// OIHW -> [Cin][k*k][Cout] with the taps flipped: Wt[ci][t][co] = W[co][ci][k*k-1-t]
__global__ void pack_conv_weight_t_kernel(const float* __restrict__ w, float* __restrict__ o, int Cout, int Cin, int taps) {
    const size_t total = (size_t)Cout * Cin * taps;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        const size_t t2 = i / Cout;
        const int t = (int)(t2 % taps);
        const int ci = (int)(t2 / taps);
        o[i] = w[((size_t)co * Cin + ci) * taps + (taps - 1 - t)];
    }
}

// packed gradient [Cout][taps][Cin] -> += into the OIHW parameter gradient [Cout][Cin][taps]
__global__ void unpack_conv_grad_kernel(const float* __restrict__ gp, float* __restrict__ g, int Cout, int Cin, int taps,
                                        int accumulate) {
    const size_t total = (size_t)Cout * Cin * taps;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % taps);
        const size_t t2 = i / taps;
        const int ci = (int)(t2 % Cin);
        const int co = (int)(t2 / Cin);
        const float v = gp[((size_t)co * taps + t) * Cin + ci];
        g[i] = accumulate ? g[i] + v : v;
    }
}

// Grouped fold of packed gradients into the OIHW parameter gradients: one workgroup per filter row,
//   g[co][ci][t] += gp[co][t][ci];  gp[co][t][ci] = 0      (coalesced on both sides through an LDS transpose)
// so that the wgrad kernels can keep accumulating into the atomic-friendly packed layout (consecutive lanes =
// consecutive addresses; the OIHW layout would scatter every wave atomic over ~18 cache lines).
__global__ __launch_bounds__(256) void unpack_conv_grads_kernel(const lfvdm_unpack_job* __restrict__ jobs, int njobs) {
    extern __shared__ float urow[];
    int j = 0;
    while (j + 1 < njobs && jobs[j + 1].row0 <= (int)blockIdx.x) ++j;
    const lfvdm_unpack_job J = jobs[j];
    const int co = blockIdx.x - J.row0;
    const int ldp = J.ldp ? J.ldp : J.Cin;       // the accumulator's channel stride (> Cin: zero-padded operand channels)
    const int n = J.taps * J.Cin, np = J.taps * ldp;
    float* gp = J.gp + (size_t)co * np;
    float* g = J.g + (size_t)co * n;
    for (int i = threadIdx.x; i < np; i += 256) {
        urow[i] = gp[i];
        gp[i] = 0.f;
    }
    __syncthreads();
    if (J.taps == 9) {          // i / 9 by multiplication (exact below 74906; a row has at most 16384 floats): the generic
        for (int i = threadIdx.x; i < n; i += 256) {      // division is ~40 instructions per element of a copy kernel
            const int ci = (int)(((unsigned)i * 58255u) >> 19), t = i - ci * 9;
            g[i] += urow[t * ldp + ci];
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 256) {
            const int ci = i / J.taps, t = i - ci * J.taps;
            g[i] += urow[t * ldp + ci];
        }
    }
}

// Grouped weight packing for a training step: every job packs one OIHW weight into the forward operand layout
// [Cout][tap][Cin] (transposed = 0) or the data-gradient layout [Cin][tap][Cout] with flipped taps (transposed = 1).
// A workgroup moves one tile of 32 filters x 32 input channels x all taps through LDS: the source is read in runs of
// 32*taps contiguous floats, both destination layouts are written in runs of 32 contiguous floats.
__global__ __launch_bounds__(256) void pack_conv_weights_kernel(const lfvdm_pack_job* __restrict__ jobs, int njobs) {
    __shared__ float tile[32 * (32 * 9 + 1)];
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                       // last job with blk0 <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const lfvdm_pack_job J = jobs[lo];
    const int tci = (J.Cin + 31) / 32;
    const int b = blockIdx.x - J.blk0;
    const int co0 = (b / tci) * 32, ci0 = (b % tci) * 32;
    const int nco = min(32, J.Cout - co0), nci = min(32, J.Cin - ci0);
    const int taps = J.taps;
    const int row = 32 * taps + 1;          // padded LDS stride between filters
    const int run = nci * taps;             // contiguous source floats per filter
    // Thread (i = lane of 32, w = one of 8 rows of lanes): every index is a loop variable - the first version decoded
    // (filter, channel, tap) from a flat element index with three integer divisions per element and was instruction-bound
    // (149 us per training step for 366 MB of traffic; reciprocal divisions: 125 us).
    const int i = threadIdx.x & 31, w = threadIdx.x >> 5;
    {   // (4 filters x <= 9 runs of 32 floats per thread: all loads in flight before the first LDS write)
        float v[4][9];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int c = w + 8 * cc;
            const float* src = J.src + ((size_t)(co0 + min(c, nco - 1)) * J.Cin + ci0) * taps;
#pragma unroll
            for (int rr = 0; rr < 9; ++rr) {
                const int r = i + 32 * rr;
                v[cc][rr] = (c < nco && r < run) ? src[r] : 0.f;
            }
        }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int c = w + 8 * cc;
#pragma unroll
            for (int rr = 0; rr < 9; ++rr) {
                const int r = i + 32 * rr;
                if (c < nco && r < run) tile[c * row + r] = v[cc][rr];
            }
        }
    }
    __syncthreads();
    // ld > natural width: the destination keeps zero padding channels (written once by the host, never here)
    const int ldd = J.ld ? J.ld : (J.transposed ? J.Cout : J.Cin);
    if (J.transposed) {                     // [ci][t][co], taps flipped: lanes along co (LDS stride `row`: odd)
        if (i < nco)
            for (int ii = w; ii < nci; ii += 8)
                for (int t = 0; t < taps; ++t)
                    J.dst[((size_t)(ci0 + ii) * taps + t) * ldd + co0 + i] = tile[i * row + ii * taps + (taps - 1 - t)];
    } else {                                // [co][t][ci]: lanes along ci (LDS stride `taps`: 9 or 1, odd)
        if (i < nci)
            for (int c = w; c < nco; c += 8)
                for (int t = 0; t < taps; ++t)
                    J.dst[((size_t)(co0 + c) * taps + t) * ldd + ci0 + i] = tile[c * row + i * taps + t];
    }
}

}  // namespace

extern "C" int lfvdm_pack_conv_weights(const lfvdm_pack_job* jobs_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || njobs <= 0 || total_blocks <= 0) return LFVDM_E_SHAPE;
    hipLaunchKernelGGL(pack_conv_weights_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_unpack_conv_grads(const lfvdm_unpack_job* jobs_dev, int njobs, int total_rows, int max_row_floats,
                                       void* stream) {
    if (!jobs_dev || njobs <= 0 || total_rows <= 0 || max_row_floats <= 0) return LFVDM_E_SHAPE;
    const size_t lds = (size_t)max_row_floats * sizeof(float);
    if (lds > 64 * 1024) return LFVDM_E_UNSUPPORTED;
    hipLaunchKernelGGL(unpack_conv_grads_kernel, dim3((unsigned)total_rows), dim3(256), lds, (hipStream_t)stream, jobs_dev, njobs);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

// dW (packed) += dout^T * f(src); db += colsum(dout).  See the kernel comment for the meaning of the fields.
extern "C" int lfvdm_conv_wgrad(const lfvdm_conv_args* a, void* stream) {
    const int Cin = a->C0 + a->C1;
    if (a->N <= 0 || a->Cout <= 0 || Cin <= 0 || Cin % 32 || a->C0 % 32) return LFVDM_E_SHAPE;
    if (a->ksize != 1 && a->ksize != 3) return LFVDM_E_SHAPE;
    // Cout that is not a multiple of 4 (the 3-channel head of a pixel-space model): narrow layers only (the wave-private
    // kernel guards every filter row and bias column on its own); the dout rows are read as float4, so they must be
    // padded to a multiple of 4 columns
    if (!a->res || !a->out || a->ldr < (a->Cout + 3) / 4 * 4 || a->ldr % 4) return LFVDM_E_SHAPE;
    if (a->Cout % 4 && a->Cout >= 64) return LFVDM_E_SHAPE;
    if (a->C1 > 0 && !a->src1) return LFVDM_E_SHAPE;
    if ((a->coefA == nullptr) != (a->coefB == nullptr)) return LFVDM_E_SHAPE;
    const long M = (long)a->N * a->Ho * a->Wo;
    const int nchunks = (int)((M + 31) / 32);
    {   // wide layers: cooperative workgroup tiles (the channel group must not straddle the two sources)
        // tile shape (measured on the cfg-C layers): 128 filters x 64 channels for the big 3x3 layers, 64 x 64 for
        // 1x1 / linear layers and the low-resolution levels (smaller tiles = fewer bytes of atomics per workgroup)
        const bool big3 = a->ksize == 3 && M >= 2560;
        int cot = (a->Cout >= 128 && big3) ? 4 : a->Cout >= 64 ? 2 : 0;
        // tune code (lfvdm_conv_args::tune, measured per layer shape by the caller): 1 + tile + 4 * stages + 16 * M slices;
        // tile 1 / 2 / 3 = 64 x 64 / 128 x 64 / 128 x 128 (filters x channels) per workgroup, stages 1 / 2 / 3 = two / three
        // LDS-DMA stages / two stages of 64-row chunks; stage field 0 with tile 1 / 2 = the tap-fused kernels (three / nine
        // taps per workgroup) where the layer is eligible (wgrad_taps_wseg); tune 0 = this heuristic
        const int tcode = a->tune > 0 ? a->tune - 1 : 0;
        const int t_cot = tcode & 3, t_ns = (tcode >> 2) & 3;
        const long t_ms = tcode >> 4;
        if (t_cot == 1 && a->Cout >= 64) cot = 2;
        if ((t_cot == 2 || t_cot == 3) && a->Cout >= 128) cot = 4;
        int kt = 0;
        for (int k : {2})
            if (kt == 0 && Cin % (32 * k) == 0 && a->C0 % (32 * k) == 0) kt = k;
        // tile code 3: 128 filters x 128 channels per workgroup (four 32x32 tiles per wave: twice the MFMAs per staged chunk
        // and per barrier, 32 instead of 21 FLOP per staged byte); LDS-DMA kernel only
        const bool wide_k = t_cot == 3 && cot == 4 && Cin % 128 == 0 && a->C0 % 128 == 0;
        if (const char* f = getenv("LFVDM_WGRAD_TILE")) {   // tuning aid: "<cot><kt>", e.g. 22
            const int v = atoi(f);
            if (cot >= v / 10) cot = v / 10;
            if (kt >= v % 10) kt = v % 10;
        }
        if (cot && kt && !getenv("LFVDM_WGRAD_WAVE")) {
            hipStream_t s = (hipStream_t)stream;
            // tune codes with stage field 0: tile 1 = three taps (one filter row) per workgroup, tile 2 = all nine
            const int wseg = (t_ns == 0 && (t_cot == 1 || t_cot == 2)) ? wgrad_taps_wseg(a, M) : 0;
            if (wseg) {
                int rc;
                if (t_cot == 1) rc = wseg == 32 ? launch_wgrad_taps<1, 32>(a, s, M, t_ms) : wseg == 16 ? launch_wgrad_taps<1, 16>(a, s, M, t_ms)
                                                                                                      : launch_wgrad_taps<1, 8>(a, s, M, t_ms);
                else rc = wseg == 32 ? launch_wgrad_taps<3, 32>(a, s, M, t_ms) : wseg == 16 ? launch_wgrad_taps<3, 16>(a, s, M, t_ms)
                                                                                            : launch_wgrad_taps<3, 8>(a, s, M, t_ms);
                if (rc != LFVDM_OK) return rc;
                LFVDM_CHECK_LAUNCH();
                return LFVDM_OK;
            }
            if (wgrad_dma_ok(a, M) && kt == 2) {
                int ns = getenv("LFVDM_WGRAD_STAGES") ? atoi(getenv("LFVDM_WGRAD_STAGES")) : 3;
                if (t_ns) ns = t_ns == 2 ? 3 : 2;
                const bool rows64 = t_ns == 3 && nchunks >= 4;         // stage code 3: two stages of 64-row chunks
                int rc;
                if (rows64 && wide_k) rc = launch_wgrad_dma<4, 4, 2, 64>(a, s, nchunks, t_ms);
                else if (rows64 && cot == 4) rc = launch_wgrad_dma<4, 2, 2, 64>(a, s, nchunks, t_ms);
                else if (rows64) rc = launch_wgrad_dma<2, 2, 2, 64>(a, s, nchunks, t_ms);
                else if (wide_k) rc = ns == 2 ? launch_wgrad_dma<4, 4, 2>(a, s, nchunks, t_ms) : launch_wgrad_dma<4, 4, 3>(a, s, nchunks, t_ms);
                else if (cot == 4) rc = ns == 2 ? launch_wgrad_dma<4, 2, 2>(a, s, nchunks, t_ms) : launch_wgrad_dma<4, 2, 3>(a, s, nchunks, t_ms);
                else rc = ns == 2 ? launch_wgrad_dma<2, 2, 2>(a, s, nchunks, t_ms) : launch_wgrad_dma<2, 2, 3>(a, s, nchunks, t_ms);
                if (rc != LFVDM_OK) return rc;
                LFVDM_CHECK_LAUNCH();
                return LFVDM_OK;
            }
            int rc;
            if (cot == 4 && kt == 4) rc = launch_wgrad_coop<4, 4>(a, s, nchunks);
            else if (cot == 4) rc = launch_wgrad_coop<4, 2>(a, s, nchunks);
            else if (kt == 4) rc = launch_wgrad_coop<2, 4>(a, s, nchunks);
            else rc = launch_wgrad_coop<2, 2>(a, s, nchunks);
            if (rc != LFVDM_OK) return rc;
            LFVDM_CHECK_LAUNCH();
            return LFVDM_OK;
        }
    }
    const long tiles = (long)a->ksize * a->ksize * (Cin / 32) * ((a->Cout + 31) / 32);
    long msplit = (4096 + tiles - 1) / tiles;          // aim at ~4k wave tasks
    if (msplit > nchunks) msplit = nchunks;
    if (msplit < 1) msplit = 1;
    if (int rc = det_fit(a, msplit)) return rc;
    const long ntasks = tiles * msplit;
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3((unsigned)((ntasks + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a,
                       (int)msplit);
    LFVDM_CHECK_LAUNCH();
    return det_finish(a, msplit, (hipStream_t)stream);
}

extern "C" int lfvdm_conv_wgrad_grouped(const lfvdm_wgrad_job* jobs_dev, int njobs, int total_tasks, void* stream) {
    if (!jobs_dev || njobs <= 0 || total_tasks <= 0) return LFVDM_E_SHAPE;
    hipLaunchKernelGGL(conv_wgrad_grouped_kernel, dim3((unsigned)((total_tasks + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       jobs_dev, njobs, total_tasks);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_pack_conv_weight_t(const float* w, float* o, int Cout, int Cin, int ksize, void* stream) {
    if (Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3)) return LFVDM_E_SHAPE;
    const size_t total = (size_t)Cout * Cin * ksize * ksize;
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_conv_weight_t_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, o, Cout, Cin, ksize * ksize);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_unpack_conv_grad(const float* gp, float* g, int Cout, int Cin, int ksize, int accumulate, void* stream) {
    if (Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3)) return LFVDM_E_SHAPE;
    const size_t total = (size_t)Cout * Cin * ksize * ksize;
    const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(unpack_conv_grad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, gp, g, Cout, Cin,
                       ksize * ksize, accumulate);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}
