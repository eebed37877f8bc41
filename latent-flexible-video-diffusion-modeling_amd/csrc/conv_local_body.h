// Sample-local implicit-GEMM stage of the persistent level chain (LFVDM_CHAIN_LOCAL, round 6; level_chain.hip).
//
// Why: on 2x2 / 4x4 maps the tile family of conv_igemm_body.h mixes 2-8 samples in a 32-row tile and cuts K over 3-6
// workgroups; a stage then pays a split-K seam (slab store, ticket, slab read: 1.7 us + up to 1.5 us of slice skew), a
// GroupNorm that spans waves (1.0-1.5 us) and cold filter chunks inside its K loop (profiles/r05_chain_stamps.txt: 8.2-9.6 us
// per stage, of which 2-3 us is the K loop).  A 3x3 convolution on such maps mixes only the pixels of ONE sample and
// GroupNorm32 is per sample and per group of Cout/32 channels (unet.py:194-207, nn.py:17-19), so a work item
//     (16*RT output rows = whole samples, 16 filters = whole groups, the FULL K range)
// needs no cross-workgroup reduction and no cross-wave normalisation, and its filter slice [16][K] (72 KB for a 128 -> 128
// 3x3 layer) depends on nothing the chain computes: it is fetched by LDS-DMA before the item waits for its producers - for
// the workgroup's NEXT item as soon as the current K loop has ended - and is resident when the flags arrive.  Only the
// item's activation rows (2-16 KB, sc1 loads) sit on the path behind the flag hop.
//
// Work item = workgroup of 4 waves.  Wave w owns a quarter of K; D^T = W * X^T on v_mfma_f32_16x16x4_f32 (A = 16 filters,
// B = 16 output rows): a lane ends up with 4 CONSECUTIVE channels of one output row, i.e. with whole GroupNorm groups
// (gw = 4) or halves / quarters of them (gw = 8 / 16: one or two xor steps over the channel-quad lanes); the rows of a
// sample are the low lane bits (P = Ho*Wo divides 16).  Partials of the 4 waves go through LDS once (fixed summation order:
// deterministic); wave rt < RT finishes row tile rt: bias, residual, raw store, exact two-pass statistics by lane
// butterflies (DPP), affine + FiLM + SiLU, store - all write-through (sc1), drained, then the tile's flag (one flag per
// filter slice and 16-row tile, set by the wave that stored it: no workgroup barrier in front of the hand-off).
//
// LDS: [partials 4*RT KB][activation image of the item's samples, one zero row, the 1x1 skip segment's rows][filter
// slice].  Both images are XOR-swizzled in 16-byte slots (slot ^= row & 15: conflict-free ds_read_b128 fragments for the
// lane groups of MI355X_MICROARCH.md, LDS); the filter image is written by global_load_lds (1 KiB pieces, per-lane source
// address carries the swizzle), the activation image through registers (sc1 loads of bytes other workgroups of this
// launch published, two-source concat, rows past the batch read as zeros).
#pragma once
#include "conv_igemm_body.h"
#include <type_traits>

#include "gn_wave_body.h"

namespace {

constexpr int kLocalFS = 16;            // filters per work item
constexpr int kLocalMaxLds = 160 * 1024 - 512;      // dynamic LDS a LOCAL stage may ask for (static words of the kernel on top)

constexpr int kLocalRedFloats = 4 * 2 * 256;        // partial tiles [4 waves][RT <= 2][64 lanes] float4: the same 8 KB for every
                                                    // stage, so that a filter slice fetched ahead for the NEXT item (whose
                                                    // front region may be smaller) never lands on partials still being read

// floats in front of the filter slice: partial tiles, main activation image (+ one zero row), skip-segment rows
inline long local_front_floats(const lfvdm_conv_args* a, int rt) {
    const long P = (long)a->Ho * a->Wo, Pin = (long)a->Hs * a->Ws, rows = 16L * rt;
    const long Cin = a->C0 + a->C1, C2 = a->s2C0 + a->s2C1;
    return kLocalRedFloats + ((rows / P) * Pin + 1) * Cin + rows * C2;      // (a multiple of 64 floats: Cin, C2 are)
}
// Filter image: 16 rows of [main taps | 1x1 skip segment], each part padded to whole 1 KiB LDS-DMA pieces (64 slots of 16 B):
// a piece then lies inside ONE filter row and ONE source tensor, and its source address is a scalar base plus the lane's
// swizzled slot - no per-lane row decode (the first version computed row = slot / (K / 4) per lane and piece: ~250 cycles
// per piece, 2.5 us to request a 72 KB slice).  -> pieces per row of the two parts
inline int local_pieces_main(int K1) { return (K1 + 255) / 256; }
inline int local_pieces_skip(int C2) { return (C2 + 255) / 256; }
inline long local_filter_floats(const lfvdm_conv_args* a) {
    const int K1 = a->ksize * a->ksize * (a->C0 + a->C1), C2 = a->s2C0 + a->s2C1;
    return (long)kLocalFS * 256 * (local_pieces_main(K1) + local_pieces_skip(C2));
}

// can this launch run as a LOCAL stage with `rt` row tiles per item?
inline bool local_stage_ok(const lfvdm_conv_args* a, int rt) {
    if (rt != 1 && rt != 2) return false;
    if (a->out_mode != LFVDM_OUT_ROWS || a->coefA || a->coefB || a->resA || a->resB || a->act) return false;
    if ((a->up != 0 && a->up != 1) || (a->ksize != 1 && a->ksize != 3) || (a->stride != 1 && a->stride != 2)) return false;
    if (a->up && a->stride != 1) return false;
    const int Cin = a->C0 + a->C1, C2 = a->s2C0 + a->s2C1;
    if (a->N <= 0 || a->Cout <= 0 || a->Cout % kLocalFS || Cin <= 0 || Cin % 64 || a->C0 % 4 || C2 % 64 || a->s2C0 % 4) return false;
    if (!a->src0 || !a->W || !a->out || (a->C1 > 0 && !a->src1) || (C2 > 0 && (!a->s2src0 || !a->W2)) || (a->s2C1 > 0 && !a->s2src1)) return false;
    const int P = a->Ho * a->Wo, Pin = a->Hs * a->Ws;
    if (P <= 0 || P > 16 || (P & (P - 1)) || Pin <= 0 || a->Wo <= 0 || a->Ws <= 0) return false;
    const int pad = a->ksize / 2, Hin = a->up ? 2 * a->Hs : a->Hs, Win = a->up ? 2 * a->Ws : a->Ws;
    if ((Hin + 2 * pad - a->ksize) / a->stride + 1 != a->Ho || (Win + 2 * pad - a->ksize) / a->stride + 1 != a->Wo) return false;
    if (a->ldo % 4 || a->ldo < a->Cout || (a->res && (a->ldr % 4 || a->ldr < a->Cout))) return false;
    if (a->gn_out) {
        const int gw = a->gn_gw ? a->gn_gw : a->Cout / 32, gld = a->gn_ld ? a->gn_ld : a->Cout;
        if (!a->gn_gw && a->Cout % 32) return false;
        if ((gw != 2 && gw != 4 && gw != 8 && gw != 16) || gld % 4 || gld < a->Cout || !a->gn_gamma || !a->gn_beta) return false;
        if (a->gn_film && (a->gn_film_div <= 0 || a->gn_film_ld % 4)) return false;
    }
    if ((local_front_floats(a, rt) + local_filter_floats(a)) * 4 > kLocalMaxLds) return false;
    const long M = (long)a->N * P, lim = 1L << 31;
    const long wid = std::max(std::max((long)a->ldo, (long)(a->gn_ld ? a->gn_ld : a->Cout)), std::max((long)a->ldr, (long)std::max(a->C0, a->C1)));
    if ((long)a->Cout * a->ksize * a->ksize * Cin * 4 >= lim || (long)a->Cout * C2 * 4 >= lim) return false;
    if (M * wid * 4 >= lim || (long)a->N * Pin * std::max(a->C0, a->C1) * 4 >= lim || M * std::max(a->s2C0, a->s2C1) * 4 >= lim) return false;
    return true;
}

// the filter slice a workgroup fetches ahead for its next LOCAL item
struct LocalNext {
    const float* W;
    const float* W2;
    int K1, C2, Cout, slice;
    int front;            // its front region (floats): where its filter slice starts
    bool valid;
};

// Filter slice (16 filters of `slice`) -> LDS image at Wl, requested by `n_issuers` waves (this one: `rank`).  Row f of the
// image = pm pieces of the main taps + ps pieces of the skip segment; logical slot s of a part sits at physical slot
// s ^ (f & 15) (the xor stays inside an aligned group of 16 slots).  Lanes past the end of a part read the next filter row
// (or, past the tensor, zeros: the descriptor's num_records) into padding slots nobody reads.
__device__ __forceinline__ void local_issue_filter(const float* W, const float* W2, int K1, int C2, int Cout, int slice, float* Wl,
                                                   int rank, int n_issuers, int lane) {
    const int pm = (K1 + 255) >> 8, ps = (C2 + 255) >> 8, pp = pm + ps;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, Cout * K1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(W2 ? W2 : W), 0, W2 ? Cout * C2 * 4 : 0, 0x00020000);
    for (int f = rank; f < kLocalFS; f += n_issuers) {
        const int voff = (lane ^ f) << 4;                        // (f < 16: the row's swizzle key)
        float* row = Wl + f * pp * 256;
        const int s1 = (slice * kLocalFS + f) * K1 * 4, s2 = (slice * kLocalFS + f) * C2 * 4;
        for (int j = 0; j < pm; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (__attribute__((address_space(3))) void*)(row + 256 * j), 16, voff, s1 + 1024 * j, 0, 0);
        for (int j = 0; j < ps; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (__attribute__((address_space(3))) void*)(row + 256 * (pm + j)), 16, voff,
                                                     s2 + 1024 * j, 0, 0);
    }
}

// LDS byte offset of a pointer into the dynamic LDS, and a 16-byte LDS load from a byte offset (ds_read_b128)
__device__ __forceinline__ unsigned lds_off(const float* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}
__device__ __forceinline__ f32x4 lds_ld4(unsigned off) {
    return *(const __attribute__((address_space(3))) f32x4*)(uintptr_t)off;
}

// KS: filter size (1 | 3); CPW: 16-value units per tap and wave = Cin / 64 (1, 2, 4; 0 = run-time)
template <int RT, int KS, int CPW>
__device__ __forceinline__ bool conv_local_body(const lfvdm_conv_args& p, int front, bool filter_resident, const LocalNext& nx,
                                                const ChainCtx& cx, const int* res_deps, int res_n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * RT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = p.Ho * p.Wo, Pin = p.Hs * p.Ws;
    const int Cin = p.C0 + p.C1, taps = p.ksize * p.ksize, K1 = taps * Cin, C2 = p.s2C0 + p.s2C1, K = K1 + C2;
    const int M = p.N * P;
    const int NS = p.Cout >> 4;
    const int rg = div_small(cx.item, NS), slice = cx.item - rg * NS;
    const int MT16 = (int)((unsigned)(M + 15) >> 4);              // 16-row tiles of the output: one flag each per filter slice
    const int m0 = rg * ROWS, f0 = slice * kLocalFS;
    const int lgP = 31 - __builtin_clz(P);                       // P is a power of two <= 16
    const int spi = ROWS >> lgP, n0 = rg * spi, rows_in = spi * Pin;
    float* red = smem;                                           // [4 waves][RT][64 lanes] float4
    float* act = smem + kLocalRedFloats;                         // [rows_in + 1][Cin], row rows_in = zeros
    float* s2a = act + (rows_in + 1) * Cin;                      // [ROWS][C2]
    float* Wl = smem + front;                                    // [16][row of whole pieces]
    STAMP(0);
    if (!filter_resident) local_issue_filter(p.W, p.W2, K1, C2, p.Cout, slice, Wl, wave, 4, lane);

    // ---- epilogue operands of the finishing waves (wave rt < RT: row tile rt): they depend on nothing the chain computes
    const int er = lane & 15, cq = lane >> 4;                    // output row of the tile, channel quad
    const int co = f0 + 4 * cq;
    const int em = m0 + 16 * wave + er;                          // (meaningful for wave < RT)
    const bool evalid = wave < RT && em < M;
    const bool gn = p.gn_out != nullptr;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f}, gam = bsum, bet = bsum, fsc = bsum, fsh = bsum, rv = bsum;
    if (wave < RT) {
        if (p.bias) bsum += ld4(p.bias + co);
        if (p.bias2) bsum += ld4(p.bias2 + co);
        if (gn) {
            gam = ld4(p.gn_gamma + co);
            bet = ld4(p.gn_beta + co);
            if (p.gn_film && evalid) {
                const int n = em >> lgP;
                const float* fl = p.gn_film + (size_t)div_small(n, p.gn_film_div) * p.gn_film_ld + co;
                fsc = ld4(fl);
                fsh = ld4(fl + p.Cout);
            }
        }
    }

    // ---- K-loop addresses of this lane (KAddr): LDS byte offset of the source pixel's row for every tap and row tile - the
    // zero row where the tap leaves the image - and its swizzle key; the filter row.  Geometry only: computed while the item
    // has nothing else to do.
    const unsigned wkey = (unsigned)(lane & 15) << 4;
    const int wpm = (K1 + 255) >> 8, wpitch = (wpm + ((C2 + 255) >> 8)) * 1024;      // filter image: pieces of the main part, row bytes
    const unsigned wbase = lds_off(Wl) + (unsigned)((lane & 15) * wpitch);
    unsigned xrow[KS * KS][RT], xkey[KS * KS][RT];
    {
        const int r = lane & 15;
        const int Hin = p.up ? 2 * p.Hs : p.Hs, Win = p.up ? 2 * p.Ws : p.Ws;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int ml = 16 * t + r, pix = ml & (P - 1), nl = ml >> lgP;
            const int oy = div_small(pix, p.Wo), ox = pix - oy * p.Wo;
#pragma unroll
            for (int tap = 0; tap < KS * KS; ++tap) {
                const int dy = KS == 3 ? tap / 3 - 1 : 0, dx = KS == 3 ? tap % 3 - 1 : 0;
                const int iy = oy * p.stride + dy, ix = ox * p.stride + dx;
                const bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
                const int sy = p.up ? iy >> 1 : iy, sx = p.up ? ix >> 1 : ix;
                const int q = ok ? nl * Pin + sy * p.Ws + sx : rows_in;
                xrow[tap][t] = lds_off(act) + (unsigned)(q * Cin * 4);
                xkey[tap][t] = (unsigned)(q & 15) << 4;
            }
        }
    }

    // ---- every wave waits for ITS producers and stages ITS quarter of the channels: wave w multiplies channels
    // [w Cin / 4, (w + 1) Cin / 4) of every tap (and the same quarter of the skip segment), so it needs the producer tiles of
    // those columns only (cx.deps is this wave's list: lfvdm_chain_plan writes four per item) and reads back only what it
    // wrote itself.  Staging overlaps the wait for the item's slowest producer, and no workgroup barrier sits between the
    // flag hop and the loads.  Residual rows (wave rt < RT, its list carries their producers) are requested first.
    __shared__ int s_ok_l[4];
    STAMP(16);
    {
        const bool ok = chain_poll(cx, lane);
        if (lane == 0) s_ok_l[wave] = ok ? 1 : 0;
    }
    STAMP(17);
    // Residual rows (wave rt < RT finishes row tile rt; their producers are a list of their own, cx_res): requested now if
    // the producers are done - the usual case, a ResBlock's input is two stages old - and they land under the K loop; if
    // not (the partial sums of a split concat convolution, computed beside the main path, can be late) the K loop does not
    // wait for them: the wave polls again in front of its epilogue.
    bool res_pending = false;
    if (p.res && wave < RT) {
        const int* f = cx.flags + (lane < res_n ? res_deps[lane] : 0);
        const int v = lane < res_n ? __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : cx.gen;
        res_pending = __builtin_amdgcn_ballot_w64(v != cx.gen) != 0;
        if (!res_pending && evalid) rv = ld4_sc1(whole_rsrc(p.res), (unsigned)(em * p.ldr + co) * 4u);
    }
    {
        // This wave's quarter of an image: rows x (Ct / 16) float4, logical slot w Ct / 16 + sl of row r at physical slot
        // (.. ^ (r & 15)).  A lone wave issues an instruction every 4-5 cycles, so the address arithmetic IS the cost of this
        // phase (a version with a per-element division, two descriptors and three guards per load took 1.0-1.9 us for two
        // float4 per lane): the common case - a power-of-two quarter that lies inside ONE of the two concatenated sources -
        // is shifts, one multiply-add and a descriptor whose num_records turns rows past the tensor (ragged last item) and
        // lanes past the image into zeros.
        auto stage_image = [&](float* img, const float* s0, const float* s1, int Ca, int Cb, int rows, unsigned grow0, unsigned grows, int Ct) {
            const int QW = Ct >> 4, c_lo = wave * (Ct >> 2), tot = rows * QW;
            const bool second = c_lo >= Ca, straddles = !second && c_lo + (Ct >> 2) > Ca;
            constexpr int UNR = 4;
            if (!straddles && (QW & (QW - 1)) == 0) {
                const int lg = 31 - __builtin_clz(QW);
                const int Cs = second ? Cb : Ca;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(second ? s1 : s0), 0, (int)(grows * (unsigned)Cs * 4u), 0x00020000);
                const unsigned cb = (unsigned)(second ? c_lo - Ca : c_lo) * 4u, rowb = (unsigned)Cs * 4u;
                for (int e0 = lane; e0 < tot; e0 += 64 * UNR) {
                    f32x4 v[UNR];
#pragma unroll
                    for (int j = 0; j < UNR; ++j) {
                        const int e = e0 + 64 * j, r = e >> lg, sl = e & (QW - 1);
                        const unsigned off = e < tot ? (grow0 + (unsigned)r) * rowb + cb + ((unsigned)sl << 4) : 0x7fffffffu;
                        v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16 /* sc1 */));
                    }
#pragma unroll
                    for (int j = 0; j < UNR; ++j) {
                        const int e = e0 + 64 * j, r = e >> lg, sl = e & (QW - 1);
                        if (e < tot) st4(img + r * Ct + (((wave * QW + sl) ^ (r & 15)) << 2), v[j]);
                    }
                }
            } else {
                const float rQW = __builtin_amdgcn_rcpf((float)QW);
                for (int e0 = lane; e0 < tot; e0 += 64 * UNR) {
                    f32x4 v[UNR];
                    int dst[UNR];
#pragma unroll
                    for (int j = 0; j < UNR; ++j) {
                        const int e = e0 + 64 * j;
                        const int r = fast_div(e < tot ? e : 0, QW, rQW), sg = wave * QW + (e < tot ? e : 0) - r * QW;
                        dst[j] = e < tot ? r * Ct + ((sg ^ (r & 15)) << 2) : -1;
                        v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (e < tot && grow0 + (unsigned)r < grows) v[j] = ld_cat_sc1(s0, s1, Ca, Cb, grow0 + (unsigned)r, sg << 2);
                    }
#pragma unroll
                    for (int j = 0; j < UNR; ++j)
                        if (dst[j] >= 0) st4(img + dst[j], v[j]);
                }
            }
        };
        stage_image(act, p.src0, p.src1, p.C0, p.C1, rows_in, (unsigned)n0 * (unsigned)Pin, (unsigned)p.N * (unsigned)Pin, Cin);
        if (C2 > 0) stage_image(s2a, p.s2src0, p.s2src1, p.s2C0, p.s2C1, ROWS, (unsigned)m0, (unsigned)M, C2);
        const int QW = Cin >> 4;
        if (lane < QW) st4(act + rows_in * Cin + (((wave * QW + lane) ^ (rows_in & 15)) << 2), (f32x4){0.f, 0.f, 0.f, 0.f});
    }
    STAMP(4);
    // the filter slice (LDS-DMA by other waves: no register dependency tells the compiler) has landed everywhere
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    STAMP(1);
    if (!(s_ok_l[0] & s_ok_l[1] & s_ok_l[2] & s_ok_l[3])) return false;     // a wait timed out / the chain was aborted

    // ---- K loop.  Wave w takes a quarter of the CHANNELS of every tap (and of the skip segment): Cin / 64 units of 16 K
    // values per tap, 4 MFMAs of K = 4 per unit and row tile.  The lane's LDS addresses - one per tap and row tile - were
    // computed in front of the poll (KAddr); with the tap count and the units per tap known at compile time (KS, CPW) the
    // loop is straight-line code: fragment reads are issued far ahead of the MFMAs that consume them and waited for with
    // counted lgkmcnt (a first version that decoded the tap per block and branched on block lengths ran at 88 cycles per
    // MFMA: one wave per SIMD issues an instruction every 4-5 cycles and hides no latency behind another wave).
    // Two accumulator chains per row tile (an MFMA 16x16x4 issues every 32 cycles, a dependent one every 40) - for RT = 2 as
    // well, so that an output element is summed in the same order whatever the row tiles per item: results do not depend
    // on that tuning choice.
    constexpr int NA = 2;
    f32x4 acc[RT][NA];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < NA; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        const int cpw = CPW > 0 ? CPW : (Cin >> 6);                  // units per tap of this wave
        const unsigned v0 = (unsigned)(((lane >> 4) << 4) + wave * cpw * 64);   // byte offset inside a tap's channels: this wave's
                                                                     // first unit + the lane's k quad
        struct Frag { f32x4 a, b[RT]; };
        auto load = [&](Frag& f, unsigned wa, const unsigned (&xa)[RT], const unsigned (&xx)[RT], unsigned vb) {
            f.a = lds_ld4(wa + (vb ^ wkey));
#pragma unroll
            for (int t = 0; t < RT; ++t) f.b[t] = lds_ld4(xa[t] + (vb ^ xx[t]));
        };
        auto mfma = [&](const Frag& f) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < RT; ++t) acc[t][e % NA] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[e], f.b[t][e], acc[t][e % NA], 0, 0, 0);
        };
        if constexpr (CPW > 0) {
            // software pipeline over the NU units of the main segment: the fragments of unit i + LA are requested before the
            // MFMAs of unit i (a ring of LA + 1 register sets, all indices static); sched_barrier pins that order - left to
            // itself hipcc sinks every read to just in front of its first use and waits lgkmcnt(0) once per unit
            constexpr int NU = KS * KS * CPW, LA = NU < 3 ? NU : 3, R = LA + 1;
            Frag f[R];
#pragma unroll
            for (int i = 0; i < LA; ++i) load(f[i], wbase + (unsigned)((i / CPW) * Cin * 4), xrow[i / CPW], xkey[i / CPW], v0 + 64u * (i % CPW));
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                if (i + LA < NU) {
                    const int n = i + LA;
                    load(f[n % R], wbase + (unsigned)((n / CPW) * Cin * 4), xrow[n / CPW], xkey[n / CPW], v0 + 64u * (n % CPW));
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma(f[i % R]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int tap = 0; tap < KS * KS; ++tap) {
                const unsigned wa = wbase + (unsigned)(tap * Cin * 4);
                for (int j = 0; j < cpw; ++j) {
                    Frag f;
                    load(f, wa, xrow[tap], xkey[tap], v0 + 64u * j);
                    mfma(f);
                }
            }
        }
        if (C2 > 0) {
            const int c2w = C2 >> 6;
            unsigned sa[RT], sx[RT];
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                sa[t] = lds_off(s2a) + (unsigned)((16 * t + (lane & 15)) * C2 * 4);
                sx[t] = (unsigned)(lane & 15) << 4;
            }
            const unsigned wa = wbase + (unsigned)(wpm * 1024);
            const unsigned vs = (unsigned)(((lane >> 4) << 4) + wave * c2w * 64);
            // (two units in flight: the segment is C2 / 64 = 2 ... 4 units per wave)
            Frag g0, g1;
            load(g0, wa, sa, sx, vs);
            for (int j = 0; j < c2w; j += 2) {
                if (j + 1 < c2w) load(g1, wa, sa, sx, vs + 64u * (j + 1));
                __builtin_amdgcn_sched_barrier(0);
                mfma(g0);
                __builtin_amdgcn_sched_barrier(0);
                if (j + 1 < c2w) {
                    if (j + 2 < c2w) load(g0, wa, sa, sx, vs + 64u * (j + 2));
                    __builtin_amdgcn_sched_barrier(0);
                    mfma(g1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    STAMP(2);
    // ---- partial tiles of the 4 waves -> LDS (lane-contiguous float4: conflict-free both ways)
#pragma unroll
    for (int t = 0; t < RT; ++t) st4(red + ((wave * RT + t) * 64 + lane) * 4, NA == 2 ? acc[t][0] + acc[t][NA - 1] : acc[t][0]);
    lds_barrier();                      // every K loop has ended: activation image and filter slice are free
    STAMP(3);
    if (wave >= RT) {
        // the workgroup's next LOCAL item: its filter slice is requested now, by the waves that store nothing below (their
        // vmcnt queue is not drained before the flag), and lands while this item is finished and the next one waits
        if (nx.valid) local_issue_filter(nx.W, nx.W2, nx.K1, nx.C2, nx.Cout, nx.slice, smem + nx.front, wave - RT, 4 - RT, lane);
    } else {
        f32x4 t = ld4(red + ((0 * RT + wave) * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) t += ld4(red + ((w * RT + wave) * 64 + lane) * 4);
        t += bsum;
        if (p.res) {
            if (res_pending) {        // (wave-uniform) the producers were not done in front of the K loop: wait for them now
                ChainCtx cr = cx;
                cr.deps = res_deps;
                cr.ndeps = res_n;
                // on a timeout the abort word is up (every other wait of the chain ends); this tile is then garbage like
                // everything behind it, and the workgroup leaves the kernel at its next wait
                if (chain_poll(cr, lane) && evalid) rv = ld4_sc1(whole_rsrc(p.res), (unsigned)(em * p.ldr + co) * 4u);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            t += rv;
        }
        const bool store_raw = !gn || p.gn_skip_raw == 0;
        if (store_raw && evalid) st4_sc1(whole_rsrc(p.out), (unsigned)(em * p.ldo + co) * 4u, t);
        STAMP(7);
        if (gn) {
            // exact two-pass statistics of (sample, group): the rows of a sample are the low lane bits (xor 1 ... P/2), the
            // channel quads of a group the lane bits 4, 5 (gw = 8, 16); gw = 2: two groups per float4
            const int gw = p.gn_gw ? p.gn_gw : p.Cout >> 5, gld = p.gn_ld ? p.gn_ld : p.Cout;
            const bool two = gw == 2;
            const float inv = 1.0f / (float)(P * gw);
            // (the 16 rows of a tile are one DPP row: quad_perm xor 1 / xor 2, then row rotations by 4 and 8 - every lane
            // ends with its sample's total; a ds_bpermute butterfly costs an LDS round trip per step)
            auto dpp = [](float v, auto ctrl) {
                return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
            };
            auto rows_sum = [&](float v) {
                if (P >= 2) v += dpp(v, std::integral_constant<int, 0xB1>{});         // quad_perm [1,0,3,2]
                if (P >= 4) v += dpp(v, std::integral_constant<int, 0x4E>{});         // quad_perm [2,3,0,1]
                if (P >= 8) v += dpp(v, std::integral_constant<int, 0x124>{});        // row_ror:4
                if (P >= 16) v += dpp(v, std::integral_constant<int, 0x128>{});       // row_ror:8
                return v;
            };
            auto unit_sum = [&](float& a, float& b) {
                a = rows_sum(a);
                if (two) b = rows_sum(b);
                // channel quads of a wider group: the other DPP rows (ds_bpermute; v_permlane16_swap / v_permlane32_swap with both
                // operands the same value did NOT give the xor-16 / xor-32 all-reduce here - four op-level cases failed - and
                // the step is worth ~1 us per denoising step: left as a shuffle)
                if (gw >= 8) a += __shfl_xor(a, 16, 64);
                if (gw >= 16) a += __shfl_xor(a, 32, 64);
            };
            float s0 = two ? t.x + t.y : (t.x + t.y) + (t.z + t.w), s1 = two ? t.z + t.w : 0.f;
            unit_sum(s0, s1);
            const float mu0 = s0 * inv, mu1 = two ? s1 * inv : mu0;
            const f32x4 mean4 = {mu0, mu0, mu1, mu1};
            const f32x4 d = t - mean4;
            float v0 = two ? d.x * d.x + d.y * d.y : (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w), v1 = two ? d.z * d.z + d.w * d.w : 0.f;
            unit_sum(v0, v1);
            // (v_rsq_f32: 1 ulp; the IEEE sqrt + division of the tile kernels is ~60 instructions of a lone wave here)
            const float r0 = __builtin_amdgcn_rsqf(v0 * inv + p.gn_eps), r1 = two ? __builtin_amdgcn_rsqf(v1 * inv + p.gn_eps) : r0;
            f32x4 A = (f32x4){r0, r0, r1, r1} * gam;
            f32x4 B = bet - mean4 * A;
            if (p.gn_film) {
                const f32x4 sc = fsc + (f32x4){1.f, 1.f, 1.f, 1.f};
                A = A * sc;
                B = B * sc + fsh;
            }
            f32x4 y = t * A + B;
            if (p.gn_act == LFVDM_ACT_SILU) { y.x = silu_f(y.x); y.y = silu_f(y.y); y.z = silu_f(y.z); y.w = silu_f(y.w); }
            if (evalid) st4_sc1(whole_rsrc(p.gn_out), (unsigned)(em * gld + co) * 4u, y);
        }
        STAMP(8);
        // publish: a flag per (filter slice, 16-row tile), set by the wave that stored the tile once ITS write-through stores
        // have drained - no workgroup barrier: the waves that fetch the next filter slice are not waited for
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && m0 + 16 * wave < M)
            __hip_atomic_store(cx.flags + cx.flag_base + slice * MT16 + rg * RT + wave, cx.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        STAMP(18);
    }
    return true;
}

}  // namespace
