// Body of the implicit-GEMM convolution (see conv_igemm.hip for the design notes), shared by the per-launch kernel
// (conv_igemm.hip) and the persistent level chain (level_chain.hip), plus the host-side tile tables both need.
// Everything here has internal linkage (anonymous namespace): include it once per translation unit, after defining STAMP.
#pragma once
#include <stdlib.h>

#include "common_hip.h"

namespace {

// KCH = channels per K chunk (32 or 64; 64 halves the barriers per MFMA and the chunks a short K slice needs);
// GL = number of LDS-DMA stages (2 or 3): tiles are written by buffer_load ... lds into GL unpadded, XOR-swizzled stages
template <int WM, int WN, int WK, int NT, int KCH, int GL>
struct Cfg {
    static constexpr int KC = KCH;
    static constexpr int LDR = KCH;                      // LDS row (floats): unpadded, XOR-swizzled 16-byte slots
    static constexpr int QPR = KCH / 4;                  // float4 per row
    static constexpr int RSH = KCH == 64 ? 4 : 3;        // log2(QPR)
    static constexpr int BM = 32 * WM;
    static constexpr int BN = 32 * NT * WN;
    static constexpr int GT = 64 * WM * WN;              // threads per k-group
    static constexpr int NTHREADS = GT * WK;
    static constexpr int AE = (BM * QPR) / GT;           // 1 KiB pieces of the A tile per wave and chunk
    static constexpr int WE = (BN * QPR) / GT;           // ... of the W tile
    static constexpr int STAGE = (BM + BN) * LDR;        // floats per LDS stage
    static constexpr int GROUP_LDS = GL * STAGE;
    static constexpr size_t LDS_BYTES = (size_t)WK * GROUP_LDS * sizeof(float);
    static_assert((BM * QPR) % GT == 0 && (BN * QPR) % GT == 0, "tile must divide over the group");
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt (workgroup-scope release of
// global memory): in the epilogue that makes every wave wait for the acknowledgement of its output stores and for
// the cold loads of the GroupNorm parameters at each of the three barriers of the fused normalisation.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <class T>
__device__ __forceinline__ T sel(bool c, T a, T b) {
    return c ? a : b;
}

struct RowInfo {
    int n, oy, ox;
    bool valid;
    int pix;        // (n*Hs + oy*stride)*Ws + ox*stride : source pixel of the centre tap (non-upsampled sources)
    unsigned taps;  // bit t: filter tap t of this output pixel lies inside the image
    int m;          // clamped output row index (second-segment / residual addressing)
};

// exact floor(a / d) for 0 <= a < 2^24 with rd = 1.0f / d, or with rd = v_rcp_f32(d) (1 ulp) while the QUOTIENT stays
// below 2^21: the truncated product is then off by at most one and the fix-up step repairs it (avoids the
// ~40-instruction integer division sequence)
__device__ __forceinline__ int fast_div(int a, int d, float rd) {
    int q = (int)((float)a * rd);
    const int r = a - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

// exact a / d for 0 <= a < 2^21, d >= 1: v_rcp_f32 is accurate to 1 ulp, so the truncated product is off by at most
// one and fast_div's fix-up step repairs it (a generic 32-bit division is ~25 instructions, a 64-bit one ~150; the
// kernel prologue is instruction-bound: a lone wave issues one instruction per 4-5 cycles)
__device__ __forceinline__ int div_small(int a, int d) {
    return fast_div(a, d, __builtin_amdgcn_rcpf((float)d));
}

// Context of a launch that runs as ONE STAGE of the persistent level chain (level_chain.hip): the work item replaces
// blockIdx, activations produced by earlier stages of the same launch are read with `sc1` loads (served below this CU's
// L1 and the XCD's non-coherent L2 lines) AFTER the producers' tile flags have been seen, outputs are published
// write-through (`sc1` stores, drained) and the tile's flag is set to the launch's generation.  Filter pieces - which
// depend on nothing the chain produces - are in flight before the wait.
struct ChainCtx {
    int item;                 // work item of the stage: index into the XCD-aware flat grid (what blockIdx.x is per launch)
    int n_items;              // 8 * per (the flat grid's size)
    const int* deps;          // this item's producer flags (indices into `flags`), ndeps <= 64
    int ndeps;
    int* flags;               // tile flags of the whole chain
    int flag_base;            // this stage's first flag: flag_base + tile id
    int gen;                  // generation of this launch: a flag equal to it means "complete in this launch"
    int* abort_word;          // raised by any poller whose wall-clock timeout expired
    long long timeout_ticks;  // of the 100 MHz s_memrealtime clock
    int stage;                // stage index (diagnostic stamps only)
};

#ifndef LFVDM_POLL_SLEEP
#define LFVDM_POLL_SLEEP 1      // s_sleep units (64 cycles) between two polls of a wave (A/B builds: -DLFVDM_POLL_SLEEP=n)
#endif
// Poll (lanes < ndeps of the calling wave, one line each) until every dependency carries this launch's generation.
// -> false: gave up (own timeout, or the abort word was raised elsewhere).  Bounded: the loop always ends.
__device__ __forceinline__ bool chain_poll(const ChainCtx& c, int lane) {
    const int* f = c.flags + (lane < c.ndeps ? c.deps[lane] : 0);
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    int spins = 0;
    for (;;) {
        const int v = lane < c.ndeps ? __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : c.gen;
        if (__builtin_amdgcn_ballot_w64(v != c.gen) == 0) return true;
        __builtin_amdgcn_s_sleep(LFVDM_POLL_SLEEP);
        if ((++spins & 31) == 0) {
            if (__hip_atomic_load(c.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > c.timeout_ticks) {
                __hip_atomic_store(c.abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}

// 16-byte global accesses with the sc1 bit through a buffer descriptor (offsets in bytes, < 2^31)
__device__ __forceinline__ f32x4 ld4_sc1(const __amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16 /* sc1 */));
}
__device__ __forceinline__ void st4_sc1(const __amdgpu_buffer_rsrc_t rs, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (int)off, 0, 16 /* sc1 */);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t whole_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
}

// CHAIN = false: the body of conv_igemm_kernel (one workgroup = blockIdx).  CHAIN = true: one work item of a chain stage;
// -> false if the chain was aborted while this item waited (the caller leaves the kernel).
template <int WM, int WN, int WK, int NT, int KCH, bool SIMPLE, int GL, bool CHAIN>
__device__ __forceinline__ bool conv_igemm_body(const lfvdm_conv_args& p, int hyb_nfull, int hyb_kz, int par_mt, const ChainCtx& cx) {
    using CF = Cfg<WM, WN, WK, NT, KCH, GL>;
    static_assert(GL == 2 || GL == 3, "two or three LDS-DMA stages");
    constexpr int BM = CF::BM, BN = CF::BN, KC = CF::KC, LDR = CF::LDR;
    constexpr int RED_LD = BN + 1;
    static_assert(BM * RED_LD <= CF::GROUP_LDS, "reduction tile must fit the group's stages");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // wave index as a SCALAR: everything derived from it (k-group, K slice, chunk parameters, segment
    // branches) then lives in SGPRs / scalar branches instead of per-lane VGPR arithmetic under exec masks
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave / (WM * WN);
    const int wmn = wave - wk * (WM * WN);
    const int wm = wmn / WN, wn = wmn - wm * WN;
    const int gt = tid - wk * CF::GT;
    const int HoWo = p.Ho * p.Wo;
    // Zero-inserted source (up == 2: the data gradient of a stride-2 convolution) BY OUTPUT PARITY (par_mt > 0, plain
    // grid, gridDim.x = 4 * par_mt): an output pixel (oy, ox) only sees the taps with oy + dy and ox + dx even, i.e. 1, 2,
    // 2 or 4 of the 9 depending on (oy & 1, ox & 1).  A workgroup takes rows of ONE parity class - enumerated on the
    // source grid, M = N * Hs * Ws per class - and loops over that class's taps only: 9/4 taps per output pixel on
    // average instead of 9 with three quarters of the staged pieces zero.  Heavy classes (4 taps) get the low block ids.
    const bool par = !SIMPLE && !CHAIN && par_mt > 0;
    int pcls = 0;
    if (par) pcls = 3 - div_small((int)blockIdx.x, par_mt);      // (never in a chain)
    const int ppy = pcls >> 1, ppx = pcls & 1;
    const int M = par ? p.N * p.Hs * p.Ws : p.N * HoWo;
    // Workgroup -> (output tile, K slice).  Plain launches: grid (m tiles, n tiles, KZ slices).  "Tail split"
    // launches (hyb_kz > 0, flat grid): the first hyb_nfull tiles - a multiple of the CU count - are computed whole,
    // each remaining tile by hyb_kz workgroups, so that the last, partial wave of tiles does not cost a full tile time.
    int bx = CHAIN ? 0 : blockIdx.x, by = CHAIN ? 0 : blockIdx.y, KZ = CHAIN ? 1 : gridDim.z, kz = CHAIN ? 0 : blockIdx.z;
    size_t tile_id = CHAIN ? 0 : (size_t)by * gridDim.x + bx;          // ticket / slab index of this tile
    if (par) bx -= (3 - pcls) * par_mt;
    if (!CHAIN && hyb_kz > 0) {
        const int MT = (M + BM - 1) / BM;
        const int b = blockIdx.x;
        int tile = b;
        KZ = 1;
        kz = 0;
        if (b >= hyb_nfull) {
            const int r = b - hyb_nfull;
            const int q = div_small(r, hyb_kz);
            tile = hyb_nfull + q;
            kz = r - q * hyb_kz;
            KZ = hyb_kz;
        }
        by = div_small(tile, MT);
        bx = tile - by * MT;
        tile_id = (size_t)(tile - hyb_nfull);
    } else if (CHAIN || hyb_kz < 0) {
        // XCD-aware flat grid (hyb_nfull = filter tiles, -hyb_kz = K slices, gridDim.x = 8 * per): consecutive workgroup
        // ids go round-robin to the 8 XCDs, each with its own L2.  The (filter tile, K slice, output-row tile) triples are
        // laid out pair-major and XCD x takes the contiguous range [x * per, (x + 1) * per): the row tiles of one (filter
        // tile, K slice) pair run on ONE XCD (two where a range boundary cuts the pair), so every filter byte is fetched
        // into one or two L2s instead of up to eight - at the low-resolution levels the filters ARE the traffic (590 KB of
        // filters against 82 KB of activations for a 128 -> 128 3x3 layer on 2x2 maps) - and every XCD gets the same
        // number of workgroups (+-1) whatever the pair count.  Traffic only: these launches wait on latency.
        const int MT = (M + BM - 1) / BM;
        KZ = -hyb_kz;
        const int id = CHAIN ? cx.item : (int)blockIdx.x, per = CHAIN ? (cx.n_items >> 3) : (int)(gridDim.x >> 3);
        const int L = (id & 7) * per + (id >> 3);
        if (L >= MT * hyb_nfull * KZ) return true;     // padding of the last range (whole workgroup, before any barrier)
        const int pair = div_small(L, MT);
        bx = L - pair * MT;
        by = div_small(pair, KZ);
        kz = pair - by * KZ;
        tile_id = (size_t)by * MT + bx;
    }
    const int m0 = bx * BM;
    const int n0 = by * BN;
    const int Cin = p.C0 + p.C1;
    const int taps = p.ksize * p.ksize;
    const int ltaps = par ? (1 + ppy) * (1 + ppx) : taps;      // taps the K loop walks (parity classes: the live ones)
    const int NK1 = ltaps * (int)((unsigned)Cin / (unsigned)KC);
    const int NK = NK1 + (int)((unsigned)(p.s2C0 + p.s2C1) / (unsigned)KC);

    STAMP(0);
    float* gbase = smem + wk * CF::GROUP_LDS;

    // K is first split over KZ workgroups (small-M layers: more workgroups than output tiles; each writes its
    // partial tile to a slab of the split-K workspace and the last arriver of a tile - ticket with agent-scope
    // release/acquire - sums the slabs in a fixed order: deterministic, no float atomics), then over the
    // k-groups of the workgroup.  Every group runs `iters` iterations (same barrier count); a group that
    // owns fewer chunks replays its last chunk with everything masked to zero.
    // (all quantities are small - NK * KZ < 2^21 is checked by the launcher - and the divisors WK are powers of two)
    int zbeg = 0, zend = NK, zmax = NK;                // zmax: chunks of the largest K slice (workgroup-uniform)
    if (KZ > 1) {
        zbeg = div_small(NK * kz, KZ);
        zend = div_small(NK * (kz + 1), KZ);
        zmax = div_small(NK + KZ - 1, KZ);
    }
    const int NKz = zend - zbeg;
    const int kbeg = zbeg + (int)((unsigned)(NKz * wk) / (unsigned)WK);
    const int kend = zbeg + (int)((unsigned)(NKz * (wk + 1)) / (unsigned)WK);
    const int iters_g = (int)((unsigned)(zmax + WK - 1) / (unsigned)WK);

    RowInfo ri[CF::AE];
    auto decode_rows = [&]() {
        // quotients here are sample / image-row indices (< 2^21): the 1-ulp reciprocal is exact after fast_div's fix-up
        const int dHW = par ? p.Hs * p.Ws : HoWo, dW = par ? p.Ws : p.Wo;     // grid the rows are enumerated on
        const float rHoWo = __builtin_amdgcn_rcpf((float)dHW), rWo = __builtin_amdgcn_rcpf((float)dW);
        const int Hin = p.up ? 2 * p.Hs : p.Hs, Win = p.up ? 2 * p.Ws : p.Ws;
#pragma unroll
        for (int j = 0; j < CF::AE; ++j) {
            const int m = m0 + ((gt + j * CF::GT) >> CF::RSH);
            ri[j].valid = m < M;
            const int mm = ri[j].valid ? m : 0;
            ri[j].m = mm;
            ri[j].n = fast_div(mm, dHW, rHoWo);
            const int rem = mm - ri[j].n * dHW;
            ri[j].oy = fast_div(rem, dW, rWo);
            ri[j].ox = rem - ri[j].oy * dW;
            if (par) {
                ri[j].oy = 2 * ri[j].oy + ppy;
                ri[j].ox = 2 * ri[j].ox + ppx;
            }
            const int cy = ri[j].oy * p.stride, cx = ri[j].ox * p.stride;
            ri[j].pix = (ri[j].n * p.Hs + cy) * p.Ws + cx;
            // bit t = 3 * (dy + 1) + (dx + 1): tap inside the image.  The centre row / column always is (conv arithmetic
            // checked by the launcher), so the mask is an outer product of a row and a column triple
            unsigned tm = 1u;
            if (p.ksize == 3) {
                const unsigned xb = ((unsigned)(cx - 1) < (unsigned)Win ? 1u : 0u) | 2u | ((unsigned)(cx + 1) < (unsigned)Win ? 4u : 0u);
                tm = ((unsigned)(cy - 1) < (unsigned)Hin ? xb : 0u) | (xb << 3) | ((unsigned)(cy + 1) < (unsigned)Hin ? (xb << 6) : 0u);
            }
            ri[j].taps = ri[j].valid ? tm : 0u;
        }
    };

    int wrow[CF::WE];       // clamped filter row of this thread's W elements
    unsigned wmask = 0;     // bit j: that filter row exists
#pragma unroll
    for (int j = 0; j < CF::WE; ++j) {
        const int co = n0 + ((gt + j * CF::GT) >> CF::RSH);
        wmask |= (co < p.Cout ? 1u : 0u) << j;
        wrow[j] = min(co, p.Cout - 1);
    }
    const int col = (gt & (CF::QPR - 1)) * 4;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // ---- epilogue operands: bias, residual rows and the coefficients of the fused GroupNorm depend on nothing this
    // launch computes (in a chain: on tiles whose flags this item has already seen), and their ~0.5-0.8 us of cold-load
    // latency used to sit behind the split-K seam, on the path of the tile's last arriver (round-5 chain stamps: epi
    // 0.45-0.7 us, gn 1.1-1.5 us once the statistics had gone to registers; requested just in front of the seam they only
    // moved into its vmcnt drain; requested inside the K loop, into the counted vmcnt of its next chunk: vmcnt counts in
    // order).  Small tiles (EPV <= 2 float4 per thread: the latency-bound low-resolution launches) request them FIRST, in
    // front of the filter pieces; large tiles - where the registers are worth more than the latency - in front of the
    // seam.  A thread owns 4 consecutive output columns of EPV rows.
    constexpr int QN = BN / 4;                              // float4 per tile row
    constexpr int EPV = (BM * QN + CF::NTHREADS - 1) / CF::NTHREADS;   // float4 per thread
    constexpr int RSTEP = CF::NTHREADS / QN;
    static_assert(CF::NTHREADS % QN == 0, "column ownership");
    const bool nchw = p.out_mode == LFVDM_OUT_NCHW;
    const bool gn = p.gn_out != nullptr;
    const int c4 = (tid % QN) * 4;
    const int row0 = tid / QN;
    const int co = n0 + c4;
    const bool cok = co < p.Cout && row0 < BM;   // Cout % 4 == 0 is checked by the launcher for this layout
    const int cc = cok ? co : 0;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    f32x4 rv[EPV], ra[EPV], rb[EPV];
    f32x4 gam = {0.f, 0.f, 0.f, 0.f}, bet = gam, fsc[EPV], fsh[EPV];
    int mo[EPV];
    constexpr bool EPI_EARLY = EPV <= 2;
    auto load_epilogue_operands = [&](bool params, bool residual, bool gn_coefs) {      // each part exactly once per tile
      if (!nchw) {
        if (params) {
            if (p.bias) bsum += ld4(p.bias + cc);
            if (p.bias2) bsum += ld4(p.bias2 + cc);
        }
        const float rHoWo = 1.0f / (float)HoWo;
        // parity classes: class row (n, i, j) of the source grid -> output row (n, 2i + py, 2j + px)
        const int sHW = p.Hs * p.Ws;
        const float rsHW = __builtin_amdgcn_rcpf((float)sHW), rsW = __builtin_amdgcn_rcpf((float)p.Ws);
        auto out_row = [&](int m) {
            if (!par) return m;
            const int n = fast_div(m, sHW, rsHW), rem = m - n * sHW;
            const int i = fast_div(rem, p.Ws, rsW), j = rem - i * p.Ws;
            return (n * p.Ho + 2 * i + ppy) * p.Wo + 2 * j + ppx;
        };
#pragma unroll
        for (int i = 0; i < EPV; ++i) {
            const int m = min(m0 + min(row0 + i * RSTEP, BM - 1), M - 1);
            if (residual) {
                mo[i] = out_row(m);
                rv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (p.res) {
                    if constexpr (CHAIN) rv[i] = ld4_sc1(whole_rsrc(p.res), (unsigned)(mo[i] * p.ldr + cc) * 4u);
                    else rv[i] = ld4(p.res + (unsigned)(mo[i] * p.ldr + cc));
                }
            }
            if (params) {
                ra[i] = (f32x4){1.f, 1.f, 1.f, 1.f};
                rb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (p.resA) {
                    const int n = fast_div(m, HoWo, rHoWo);
                    ra[i] = ld4(p.resA + (unsigned)(n * p.Cout + cc));
                    rb[i] = ld4(p.resB + (unsigned)(n * p.Cout + cc));
                }
            }
        }
        if (gn && gn_coefs) {
            const float rPF = __builtin_amdgcn_rcpf((float)(HoWo * p.gn_film_div));
            if (cok) {
                gam = ld4(p.gn_gamma + cc);
                bet = ld4(p.gn_beta + cc);
            }
#pragma unroll
            for (int i = 0; i < EPV; ++i) {
                fsc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                fsh[i] = fsc[i];
                const int m = min(m0 + min(row0 + i * RSTEP, BM - 1), M - 1);
                if (p.gn_film && cok) {
                    const float* fl = p.gn_film + (size_t)fast_div(m, HoWo * p.gn_film_div, rPF) * p.gn_film_ld + cc;
                    fsc[i] = ld4(fl);
                    fsh[i] = ld4(fl + p.Cout);
                }
            }
        }
    }
    };
    // small tiles: parameters at once (in flight while the filter pieces are issued and - chain stage - while the item
    // waits for its producers); the residual rows with them, or - chain stage - behind the poll (K loop, first chunk)
    if constexpr (EPI_EARLY) load_epilogue_operands(true, !CHAIN, true);

  {
    // ---- LDS-DMA main loop.  `buffer_load_dwordx4 ... lds` writes 64 lanes x 16 B to ONE contiguous 1 KiB piece
    // of LDS (wave-uniform base + lane * 16) while every lane supplies its own source offset.  The stage image is
    // therefore the plain row-major [rows][KC] tile (a piece = 64 / QPR rows), no VGPR staging, no ds_write, no
    // masking selects: a lane whose tap lies outside the image (or past the last filter / K chunk) passes an offset
    // beyond the buffer descriptor's num_records and the hardware writes zeros.  Bank conflicts of the b128 fragment
    // reads are avoided by an XOR swizzle of the 16-byte slot inside a row, applied to the per-lane SOURCE column
    // and to the read (same involution on both sides, CDNA guide rule 21): slot ^= (row >> 1) & 7 for 128-byte rows,
    // slot ^= row & 15 for 256-byte rows (distinct slots for the rows of every 16-lane read group).
    // GL stages: chunk k+GL-1 is in flight while chunk k is multiplied; ONE barrier per chunk; the DMA queue is
    // drained with counted vmcnt waits (never to zero inside the loop when GL = 3).
    constexpr int QPR = CF::QPR, RSH = CF::RSH, AE = CF::AE, WE = CF::WE;
    constexpr unsigned kOOB = 0x40000000u;       // >= num_records of every descriptor (checked by the launcher);
                                                 // sums of two such terms stay below 2^32 (no wrap back into range)
    // SIMPLE: one raw source.  Otherwise also the virtual concat (src0 | src1) and the fused 1x1 skip segment
    // (s2src0 | s2src1 at output resolution, weights W2); the buffer descriptor of a chunk is built from scalar
    // selects of base pointer and size (holding six descriptors at once spills SGPRs to scratch).
    const int wld = taps * Cin, w2ld = p.s2C0 + p.s2C1;
    const unsigned pixA = (unsigned)p.N * p.Hs * p.Ws * 4u;
    auto desc = [](const float* base, unsigned bytes) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
    };
    // per-lane byte offsets without the chunk's (tap, channel) shift; rows / filters that do not exist start at kOOB
    constexpr int XE = SIMPLE ? 1 : AE, XW = SIMPLE ? 1 : WE;
    unsigned aoff[AE], woff[WE], aoff1[XE], soff0[XE], soff1[XE], w2off[XW], amsk[AE];
    // nearest-2x upsampled / zero-inserted main source (general variant only): the tap is no uniform pixel shift there
    // - source pixel ((oy + dy) >> 1, (ox + dx) >> 1) - so the per-lane offset is rebuilt per chunk from (image base,
    // oy, ox); a handful of VALU instructions per piece on the three such launches of a forward pass
    int upy[XE], upx[XE];
    unsigned upb[XE], upq[XE];
#pragma unroll
    for (int j = 0; j < WE; ++j) {
        const int e = gt + j * CF::GT, r = e >> RSH, sl = e & (QPR - 1);
        const unsigned q16 = (unsigned)(sl ^ (KC == 32 ? ((r >> 1) & 7) : (r & 15))) * 16u;
        const bool ok = (wmask >> j) & 1u;
        woff[j] = ok ? (unsigned)wrow[j] * wld * 4u + q16 : kOOB;
        if constexpr (!SIMPLE) w2off[j] = ok ? (unsigned)wrow[j] * w2ld * 4u + q16 : kOOB;
    }
    // The output-row decode (decode_rows: ~60 instructions per staged row) runs AFTER the filter pieces of the
    // look-ahead chunks have been issued: filters are cold (last read one denoising step ago), their pieces are the
    // long pole of the first chunk, and they depend on nothing but the tile position.
    auto prepare_rows = [&]() {
        decode_rows();
#pragma unroll
        for (int j = 0; j < AE; ++j) {
            const int e = gt + j * CF::GT, r = e >> RSH, sl = e & (QPR - 1);
            const unsigned q16 = (unsigned)(sl ^ (KC == 32 ? ((r >> 1) & 7) : (r & 15))) * 16u;
            aoff[j] = (unsigned)ri[j].pix * p.C0 * 4u + q16;
            amsk[j] = ri[j].taps | (ri[j].valid ? 0x80000000u : 0u);     // bit 31: the output row exists (skip segment)
            if constexpr (!SIMPLE) {
                aoff1[j] = (unsigned)ri[j].pix * p.C1 * 4u + q16;
                soff0[j] = (unsigned)ri[j].m * p.s2C0 * 4u + q16;
                soff1[j] = (unsigned)ri[j].m * p.s2C1 * 4u + q16;
                upy[j] = ri[j].oy * p.stride;
                upx[j] = ri[j].ox * p.stride;
                upb[j] = (unsigned)ri[j].n * p.Hs * p.Ws;
                upq[j] = q16;
            }
        }
    };
    // fragment read offsets (floats, relative to the stage): A row 32*wm + (lane & 31), W row 32*NT*wn + (lane & 31)
    // (+ 32 t: same swizzle), logical slot 2g + (lane >> 5)
    int offA[KC / 8], offW[KC / 8];
    {
        const int ra = 32 * wm + (lane & 31), rw = 32 * NT * wn + (lane & 31), h = lane >> 5;
        const int fa = KC == 32 ? ((ra >> 1) & 7) : (ra & 15), fw = KC == 32 ? ((rw >> 1) & 7) : (rw & 15);
#pragma unroll
        for (int g = 0; g < KC / 8; ++g) {
            offA[g] = ra * KC + (((2 * g + h) ^ fa) << 2);
            offW[g] = (BM + rw) * KC + (((2 * g + h) ^ fw) << 2);
        }
    }
    const int k3 = p.ksize == 3 ? 1 : 0;
    const int pC0 = p.C0, pC1 = p.C1, pS0 = p.s2C0, pS1 = p.s2C1, pWs = p.Ws;   // by-value copies for the selects below
    // (channel chunk, tap) of the next chunk: the chunks of a k-group are issued in order, so the pair is carried as
    // scalar state and advanced after every issue - one division per kernel instead of one per chunk.  Chunks past the
    // group's slice (kc_raw >= kend: padding iterations of a group that owns fewer chunks) pass out-of-range offsets
    // on every lane, whatever the pair says.
    int nx_ci = div_small(kbeg, ltaps), nx_tap = kbeg - nx_ci * ltaps;
    auto issue = [&](int kc_raw, int stage, bool doA = true, bool doW = true) {   // all but the per-lane offsets is wave-uniform
        const bool live = kc_raw < kend;
        const bool main_seg = SIMPLE ? true : kc_raw < NK1;
        const int ci = main_seg ? nx_ci : kc_raw - NK1, cc = ci * KC;
        int tap = main_seg ? nx_tap : 0;
        if (par) {      // l-th live tap of the class: dy = 0 (even rows) or -1, +1 (odd rows), the same for dx
            const int ty = ppx ? (tap >> 1) : tap, tx = ppx ? (tap & 1) : 0;
            tap = 3 * (ppy ? 2 * ty : 1) + (ppx ? 2 * tx : 1);
        }
        nx_tap += 1;
        if (nx_tap == ltaps) { nx_tap = 0; nx_ci += 1; }
        const int c0 = sel(main_seg, pC0, pS0);
        const bool second = SIMPLE ? false : cc >= c0;
        const int cl = second ? cc - c0 : cc;
        const int Csrc = sel(main_seg, sel(second, pC1, pC0), sel(second, pS1, pS0));
        const int t3 = tap / 3, km = main_seg ? k3 : 0;                // branch-free: no tap offset for 1x1 / skip chunks
        const int dy = (t3 - 1) * km, dx = (tap - t3 * 3 - 1) * km;
        const int ashift = ((dy * pWs + dx) * Csrc + cl) * 4;          // bytes, may be negative
        const int upm = SIMPLE ? 0 : sel(main_seg, p.up, 0);            // 0 none, 1 nearest x2, 2 zero insertion (wave-uniform)
        const unsigned wshift = live ? (unsigned)(main_seg ? tap * Cin + cc : cc) * 4u : kOOB;
        const unsigned tapbit = live ? (main_seg ? (1u << tap) : 0x80000000u) : 0u;
        const float* abase = SIMPLE ? p.src0 : sel(main_seg, sel(second, p.src1, p.src0), sel(second, p.s2src1, p.s2src0));
        const float* bbase = SIMPLE ? p.W : sel(main_seg, p.W, p.W2);
        const __amdgpu_buffer_rsrc_t rsA = desc(abase, sel(main_seg, pixA, (unsigned)M * 4u) * (unsigned)Csrc);
        const __amdgpu_buffer_rsrc_t rsB = desc(bbase, (unsigned)p.Cout * (unsigned)sel(main_seg, wld, w2ld) * 4u);
        float* As = gbase + stage * CF::STAGE + wmn * 256;             // this wave's first piece
        float* Wst = As + BM * KC;
#pragma unroll
        for (int j = 0; j < AE; ++j) {
            if (!doA) break;
            unsigned base = aoff[j];
            if constexpr (!SIMPLE) {   // by-value selects (a ternary on array elements selects an ADDRESS: arrays go to scratch)
                const unsigned o0 = aoff[j], o1 = aoff1[j], o2 = soff0[j], o3 = soff1[j];
                base = sel(main_seg, sel(second, o1, o0), sel(second, o3, o2));
            }
            unsigned off = (amsk[j] & tapbit) ? base + (unsigned)ashift : kOOB;
            if constexpr (!SIMPLE) {
                if (upm) {                  // scalar branch
                    const int iy = upy[j] + dy, ix = upx[j] + dx;       // inside the upsampled image iff the tap bit is set
                    const bool ok = (amsk[j] & tapbit) && (upm == 1 || ((iy | ix) & 1) == 0);
                    const unsigned px = upb[j] + (unsigned)((iy >> 1) * pWs + (ix >> 1));
                    off = ok ? px * (unsigned)Csrc * 4u + (unsigned)cl * 4u + upq[j] : kOOB;
                }
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(As + j * CF::GT * 4), 16,
                                                     (int)off, 0, 0, CHAIN ? 16 /* sc1 */ : 0);
        }
#pragma unroll
        for (int j = 0; j < WE; ++j) {
            if (!doW) break;
            unsigned base = woff[j];
            if constexpr (!SIMPLE) {
                const unsigned o0 = woff[j], o1 = w2off[j];
                base = sel(main_seg, o0, o1);
            }
            const unsigned off = base + wshift;                        // missing filter row / padding chunk: >= kOOB
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(Wst + j * CF::GT * 4), 16,
                                                     (int)off, 0, 0, 0);
        }
    };
#define LFVDM_GSTEP(S_, IT_)                                                                                    \
    do {                                                                                                       \
        /* RAW: this wave's pieces of the chunk have landed (counted vmcnt) before it arrives at the barrier.  \
           WAR: its fragment reads of the previous chunk have RETURNED (lgkmcnt) before it arrives - the other  \
           waves restage that buffer right after the barrier, and a zero-filled (out-of-range) piece lands      \
           within a few cycles */                                                                              \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((GL - 2) * (AE + WE)) : "memory");                  \
        __builtin_amdgcn_s_barrier();                                                                          \
        asm volatile("" ::: "memory");                                                                         \
        if ((IT_) == 0) STAMP(15);                                                                             \
        /* look-ahead pieces first (maximum time to land), then per MFMA group: fragment reads + 4*NT MFMAs.    \
           Measured against hoisting all fragment reads / pinning the order with sched_barriers: this plain     \
           form, which lets the scheduler slide the wait + barrier of the next step above the last MFMA group, \
           was the fastest on every shape (tools/ab_libs.sh, tools/ab_shapes.sh) */                            \
        issue(kbeg + (IT_) + GL - 1, ((S_) + GL - 1) % GL);                                                    \
        /* chain stage, small tile: the residual rows (another stage's tile: legal only behind the poll) are     \
           requested behind the first chunk's wait and land under the MFMAs */                                   \
        if constexpr (EPI_EARLY && CHAIN) { if ((IT_) == 0) load_epilogue_operands(false, true, false); }        \
        const float* st_ = gbase + (S_) * CF::STAGE;                                                           \
        _Pragma("unroll") for (int g = 0; g < KC / 8; ++g) {                                                   \
            const f32x4 a4 = ld4(st_ + offA[g]);                                                               \
            f32x4 b4[NT];                                                                                      \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) b4[t] = ld4(st_ + offW[g] + t * 32 * KC);           \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                      \
                _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                 \
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[t][e], acc[t], 0, 0, 0);           \
        }                                                                                                      \
    } while (0)
    {   // look-ahead chunks: filter pieces, row decode, activation pieces
        const int ci0 = nx_ci, tap0 = nx_tap;
#pragma unroll
        for (int d = 0; d < GL - 1; ++d) issue(kbeg + d, d, false, true);
        nx_ci = ci0;
        nx_tap = tap0;
        prepare_rows();
        if constexpr (CHAIN) {
            // the producers' tiles: polled by wave 0 (its filter pieces are in flight; the first poll result queues behind
            // them), the other waves park at the barrier - an LDS-only barrier, the DMA stays in flight
            __shared__ int s_go;
            STAMP(16);
            if (wave == 0) {
                const bool ok = chain_poll(cx, lane);
                if (lane == 0) s_go = ok ? 1 : 0;
            }
            lds_barrier();
            STAMP(17);
            if (!s_go) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                return false;
            }
        }
#pragma unroll
        for (int d = 0; d < GL - 1; ++d) issue(kbeg + d, d, true, false);
        // queue order is W(0) [W(1)] A(0) [A(1)]: chunk 0 is complete once only A(1) is outstanding (the counted wait of
        // the first step assumes the steady-state order A(k) W(k))
        if constexpr (GL == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AE) : "memory");
    }
    STAMP(1);
    int it = 0;
    if constexpr (GL == 2) {
        for (; it + 2 <= iters_g; it += 2) { LFVDM_GSTEP(0, it); LFVDM_GSTEP(1, it + 1); }
        if (it < iters_g) { LFVDM_GSTEP(0, it); }
    } else {
        for (; it + 3 <= iters_g; it += 3) { LFVDM_GSTEP(0, it); LFVDM_GSTEP(1, it + 1); LFVDM_GSTEP(2, it + 2); }
        if (it < iters_g) {
            LFVDM_GSTEP(0, it);
            if (it + 1 < iters_g) { LFVDM_GSTEP(1, it + 1); }
        }
    }
#undef LFVDM_GSTEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the zero-filled look-ahead pieces must land before the stages are reused
  }
    STAMP(2);
    lds_barrier();   // all fragment reads done before the stages are reused for the reduction

    // ---- cross-k-group reduction through LDS: group wk writes its partial block tile ----
    float* red = gbase;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            red[row * RED_LD + 32 * NT * wn + t * 32 + (lane & 31)] = acc[t][r];
        }
    lds_barrier();
    STAMP(3);

    // ---- split-K over workgroups (KZ > 1): every slice stores its partial tile to a slab of the
    // workspace; the slice that arrives LAST at the tile's counter sums all slabs in FIXED order (slice
    // 0..KZ-1), so the result is bitwise reproducible whatever the arrival order.  Cross-workgroup
    // visibility follows the agent-scope release/acquire recipe of the CDNA guide (G16): stores ->
    // every wave s_waitcnt vmcnt(0) -> barrier -> lane 0: release fence + drained ticket atomic;
    // last arriver: acquire fence -> barrier -> plain loads.
    bool do_epilogue = true;
    if (KZ > 1) {
        // Slabs are published WRITE-THROUGH (16-byte `sc1` stores: the bytes leave the XCD's L2 at once, no dirty
        // lines for a release fence to write back) and read back with `sc1` loads (served below the reading CU's L1),
        // which takes the `buffer_wbl2` / `buffer_inv` pair - 1.3 us of the 3.0 us seam in the in-kernel phase table,
        // DESIGN.md §5 - off the critical path (MI355X guide, "Valid forms": every store of the handed-off bytes `sc1`,
        // every storing wave drains vmcnt, workgroup barrier, ONE lane's agent-scope counter add; the workgroup whose
        // add returned last reads after a barrier that lane joins, every load `sc1`).  LFVDM_SEAM_FENCES builds keep
        // the fenced form (plain stores, release fence; acquire fence, plain loads) for A/B runs.
        constexpr int QNs = BN / 4;
        float* slab = p.splitk_ws + (tile_id * KZ + kz) * (size_t)(BM * BN);
#ifndef LFVDM_SEAM_FENCES
        const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, BM * BN * 4, 0x00020000);
#endif
        for (int e = tid; e < BM * QNs; e += CF::NTHREADS) {
            const int row = e / QNs, c4s = (e - row * QNs) * 4;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < WK; ++w) {
                const float* r = smem + w * CF::GROUP_LDS + row * RED_LD + c4s;
                t.x += r[0]; t.y += r[1]; t.z += r[2]; t.w += r[3];
            }
#ifdef LFVDM_SEAM_FENCES
            st4(slab + row * BN + c4s, t);
#else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), rs_slab, (row * BN + c4s) * 4, 0, 16 /* sc1 */);
#endif
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        STAMP(4);
        __shared__ int s_last;
        if (tid == 0) {
#ifdef LFVDM_SEAM_FENCES
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            STAMP(9);
            const int ticket = __hip_atomic_fetch_add(p.splitk_cnt + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (ticket == KZ - 1) ? 1 : 0;
            STAMP(10);
            if (s_last) {
                // self-cleaning ticket: nobody else touches it once all KZ slices have arrived, and the next
                // launch is stream-ordered behind this one - no memset node per launch
                __hip_atomic_store(p.splitk_cnt + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef LFVDM_SEAM_FENCES
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                STAMP(11);
            }
        }
        __syncthreads();
        STAMP(5);
        do_epilogue = s_last != 0;
        if (do_epilogue) {
            // ordered sum of the KZ slabs back into the LDS tile of group 0 (the epilogue below reads it)
            const float* base = p.splitk_ws + tile_id * KZ * (size_t)(BM * BN);
#ifndef LFVDM_SEAM_FENCES
            const __amdgpu_buffer_rsrc_t rs_all = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, KZ * BM * BN * 4, 0x00020000);
#endif
            for (int e = tid; e < BM * QNs; e += CF::NTHREADS) {
                const int row = e / QNs, c4s = (e - row * QNs) * 4;
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
#ifdef LFVDM_SEAM_FENCES
                for (int z = 0; z < KZ; ++z) t += ld4(base + (size_t)z * (BM * BN) + row * BN + c4s);
#else
                // all slab loads of the element in flight at once (KZ <= 8: cfg_valid), summed in slice order
                u32x4 sv[8];
#pragma unroll
                for (int z = 0; z < 8; ++z)
                    sv[z] = __builtin_amdgcn_raw_buffer_load_b128(rs_all, z < KZ ? (z * (BM * BN) + row * BN + c4s) * 4 : -1, 0, 16 /* sc1 */);
#pragma unroll
                for (int z = 0; z < 8; ++z)
                    if (z < KZ) t += __builtin_bit_cast(f32x4, sv[z]);
#endif
                float* r = smem + row * RED_LD + c4s;
                r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
            }
            lds_barrier();
            STAMP(6);
        }
    }
    if (!do_epilogue) return true;
    const int WKE = (KZ > 1) ? 1 : WK;    // after a split-K combine the full sum sits in group 0's tile

    // ---- epilogue: a thread owns 4 consecutive output columns of EPV rows: bias is read once as a
    // float4, all residual loads are issued back to back, stores are 16-byte.
    if (!nchw) {
        // large tiles: bias / residual rows here, GroupNorm coefficients inside the normalisation (their live ranges stay
        // apart: hoisting all of them in front of the seam cost the 64 x 128 tile half its occupancy, 143 -> 230 VGPRs)
        if constexpr (!EPI_EARLY) load_epilogue_operands(true, true, false);
        const bool store_raw = !gn || p.gn_skip_raw == 0;
        f32x4 tv[EPV];
#pragma unroll
        for (int i = 0; i < EPV; ++i) {
            const int row = min(row0 + i * RSTEP, BM - 1);
            f32x4 t = bsum;
            for (int w = 0; w < WKE; ++w) {
                const float* r = smem + w * CF::GROUP_LDS + row * RED_LD + c4;
                t.x += r[0]; t.y += r[1]; t.z += r[2]; t.w += r[3];
            }
            if (p.res) t += rv[i] * ra[i] + rb[i];
            tv[i] = t;
            const int m = m0 + row;
            if (store_raw && m < M && cok && row0 + i * RSTEP < BM) {
                if constexpr (CHAIN) st4_sc1(whole_rsrc(p.out), (unsigned)(mo[i] * p.ldo + co) * 4u, t);
                else st4(p.out + ((size_t)mo[i] * p.ldo + co), t);
            }
        }
        STAMP(7);
        if (gn) {
            // ---- fused GroupNorm(+FiLM)(+activation) of the output tile.  The launcher guarantees whole samples
            // (P = Ho*Wo divides BM) and whole groups (gw = Cout/32 divides BN) per tile.  The finished values go
            // back into group 0's LDS tile (each element is read and rewritten by its one owner thread), then one
            // exact two-pass mean / variance per (sample, group) unit, then the affine on the registers.
            // the per-channel / per-sample coefficients are fetched first: their latency hides behind the statistics
            const int P = HoWo, gw = p.gn_gw ? p.gn_gw : p.Cout >> 5, gld = p.gn_ld ? p.gn_ld : p.Cout;
            const float rP = __builtin_amdgcn_rcpf((float)P), rgw = __builtin_amdgcn_rcpf((float)gw);   // fast_div operands here are < 2^21
            if constexpr (!EPI_EARLY) load_epilogue_operands(false, false, true);
            // ---- register form (round 5; small tiles: EPV <= 2 float4 per thread - larger tiles keep the general form, whose
            // registers do not pile up on top of the K loop's): P and gw powers of two (every 64 ... 512-channel layer on
            // 2x2 ... 8x8 maps).  The
            // finished values stay in their owners' registers; a unit's sums are lane butterflies over the quads of the
            // group and the rows of the sample inside a wave, plus - where a sample spans several waves - ONE exchange
            // through LDS per pass.  Two barriers instead of four, no tile rewrite, no per-unit loops: the general form
            // below spends ~2 us on a 32 x 32 tile in LDS round trips, divisions and serial row loops
            // (profiles/r03_conv_phase_stamps.txt, r05 chain stamps).  Exact two-pass statistics as before.
            constexpr int RPW = 64 / QN;                       // tile rows per wave (QN = float4 per tile row: 8, 16 or 32)
            constexpr int QSH = QN == 8 ? 3 : QN == 16 ? 4 : 5;
            static_assert(QN == 8 || QN == 16 || QN == 32, "epilogue lane map");
            const bool pow2 = EPI_EARLY && (P & (P - 1)) == 0 && (gw & (gw - 1)) == 0 && gw >= 2 && gw <= 16 && !p.gn_general;
            if (pow2) {
                const int q = tid & (QN - 1);
                const bool two = gw == 2;                      // two groups per float4
                const int gq = gw >= 4 ? gw >> 2 : 1;          // float4 quads per group
                const int rw = P < RPW ? P : RPW;              // rows of a sample inside one wave
                const int ipg = P > RSTEP ? P / RSTEP : 1;     // consecutive i (row0 + i * RSTEP) of one sample
                const int wps = P > RPW ? (P < RSTEP ? P : RSTEP) / RPW : 1;      // waves a sample spans
                const float inv = 1.0f / (float)(P * gw);
                float* part = smem + BM * RED_LD;              // [pass][i][wave][QN][2], behind group 0's reduction tile
                f32x2 sv[EPV];
                auto combine = [&](int pass) {
                    // sv[j] (j = first i of an i-group) holds this thread's partial of its unit(s): sum it over the unit
#pragma unroll
                    for (int j = 0; j < EPV; ++j) {
                        f32x2 v = sv[j];
                        for (int o = 1; o < gq; o <<= 1) v.x += __shfl_xor(v.x, o, 64);
                        for (int o = QN; o < QN * rw; o <<= 1) {
                            v.x += __shfl_xor(v.x, o, 64);
                            if (two) v.y += __shfl_xor(v.y, o, 64);
                        }
                        sv[j] = v;
                    }
                    if (wps > 1) {          // workgroup-uniform
                        float* pp = part + pass * (EPV * (CF::NTHREADS / 64) * QN * 2);
                        if ((lane >> QSH) == 0) {
#pragma unroll
                            for (int j = 0; j < EPV; ++j) {
                                float* w2 = pp + ((j * (CF::NTHREADS / 64) + wave) * QN + q) * 2;
                                w2[0] = sv[j].x;
                                w2[1] = sv[j].y;
                            }
                        }
                        lds_barrier();
                        const int w0 = (wave / wps) * wps;
#pragma unroll
                        for (int j = 0; j < EPV; ++j) {
                            f32x2 t = {0.f, 0.f};
                            for (int k = 0; k < wps; ++k) {
                                const float* r2 = pp + ((j * (CF::NTHREADS / 64) + w0 + k) * QN + q) * 2;
                                t.x += r2[0];
                                t.y += r2[1];
                            }
                            sv[j] = t;
                        }
                    }
                };
                // pass 1: sums
#pragma unroll
                for (int j = 0; j < EPV; ++j) sv[j] = (f32x2){0.f, 0.f};
#pragma unroll
                for (int i = 0; i < EPV; ++i) {
                    const int j = (i / ipg) * ipg;             // (ipg is a power of two <= EPV)
                    const f32x4 t = tv[i];
                    const f32x2 a = two ? (f32x2){t.x + t.y, t.z + t.w} : (f32x2){(t.x + t.y) + (t.z + t.w), 0.f};
#pragma unroll
                    for (int jj = 0; jj < EPV; ++jj)
                        if (jj == j) sv[jj] += a;
                }
                combine(0);
                f32x4 mean4[EPV];
#pragma unroll
                for (int i = 0; i < EPV; ++i) {
                    const int j = (i / ipg) * ipg;
                    f32x2 mu = {0.f, 0.f};
#pragma unroll
                    for (int jj = 0; jj < EPV; ++jj)
                        if (jj == j) mu = sv[jj] * inv;
                    mean4[i] = two ? (f32x4){mu.x, mu.x, mu.y, mu.y} : (f32x4){mu.x, mu.x, mu.x, mu.x};
                }
                // pass 2: centred squares
#pragma unroll
                for (int j = 0; j < EPV; ++j) sv[j] = (f32x2){0.f, 0.f};
#pragma unroll
                for (int i = 0; i < EPV; ++i) {
                    const int j = (i / ipg) * ipg;
                    const f32x4 d = tv[i] - mean4[i];
                    const f32x2 a = two ? (f32x2){d.x * d.x + d.y * d.y, d.z * d.z + d.w * d.w}
                                        : (f32x2){(d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w), 0.f};
#pragma unroll
                    for (int jj = 0; jj < EPV; ++jj)
                        if (jj == j) sv[jj] += a;
                }
                STAMP(12);
                combine(1);
                STAMP(13);
                STAMP(14);
                if (cok) {
#pragma unroll
                    for (int i = 0; i < EPV; ++i) {
                        const int row = row0 + i * RSTEP;
                        const int m = m0 + row;
                        if (row < BM && m < M) {
                            const int j = (i / ipg) * ipg;
                            f32x2 var = {0.f, 0.f};
#pragma unroll
                            for (int jj = 0; jj < EPV; ++jj)
                                if (jj == j) var = sv[jj] * inv;
                            const float r0 = 1.0f / sqrtf(var.x + p.gn_eps), r1 = two ? 1.0f / sqrtf(var.y + p.gn_eps) : r0;
                            f32x4 A = (f32x4){r0, r0, r1, r1} * gam;
                            f32x4 B = bet - mean4[i] * A;
                            if (p.gn_film) {
                                const f32x4 sc = fsc[i] + (f32x4){1.f, 1.f, 1.f, 1.f};
                                A = A * sc;
                                B = B * sc + fsh[i];
                            }
                            f32x4 y = tv[i] * A + B;
                            if (p.gn_act == LFVDM_ACT_SILU) { y.x = silu_f(y.x); y.y = silu_f(y.y); y.z = silu_f(y.z); y.w = silu_f(y.w); }
                            if constexpr (CHAIN) st4_sc1(whole_rsrc(p.gn_out), (unsigned)(m * gld + co) * 4u, y);
                            else st4(p.gn_out + ((size_t)m * gld + co), y);
                        }
                    }
                }
            } else {
            lds_barrier();                 // every partial-tile read above is done before group 0's tile is rewritten
#pragma unroll
            for (int i = 0; i < EPV; ++i) {
                if (row0 + i * RSTEP < BM) {
                    float* r = smem + (row0 + i * RSTEP) * RED_LD + c4;
                    r[0] = tv[i].x; r[1] = tv[i].y; r[2] = tv[i].z; r[3] = tv[i].w;
                }
            }
            const int SPT = div_small(BM, P), GPT = div_small(BN, gw), U = SPT * GPT;   // samples / groups / units per tile
            float* ustat = smem + BM * RED_LD;                               // [U][2] (mean, rstd)
            lds_barrier();
            STAMP(12);
            // tpu lanes per unit (a power of two <= 64, no more than the unit has rows); lane li takes whole rows
            // li, li + tpu, ... of the unit's [P][gw] slice - one address per row, the gw columns at constant offsets
            int tpu = div_small(CF::NTHREADS, U);
            tpu = tpu < 1 ? 1 : tpu > 64 ? 64 : tpu;
            tpu = tpu > P ? P : tpu;
            const int ushift = 31 - __builtin_clz(tpu);
            tpu = 1 << ushift;
            const int li = tid & (tpu - 1);
            const float inv = 1.0f / (float)(P * gw);
            for (int u = tid >> ushift; u < U; u += CF::NTHREADS >> ushift) {    // uniform trip count within a unit's lanes
                const int sI = div_small(u, GPT), g = u - sI * GPT;
                const float* base = smem + (sI * P + li) * RED_LD + g * gw;
                float s1 = 0.f;
                for (int r = li; r < P; r += tpu) {
                    const float* row = base + (r - li) * RED_LD;
                    for (int c = 0; c < gw; ++c) s1 += row[c];
                }
                for (int o = tpu >> 1; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
                const float mean = s1 * inv;
                float s2 = 0.f;
                for (int r = li; r < P; r += tpu) {
                    const float* row = base + (r - li) * RED_LD;
                    for (int c = 0; c < gw; ++c) {
                        const float d = row[c] - mean;
                        s2 += d * d;
                    }
                }
                for (int o = tpu >> 1; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
                if (li == 0) {
                    ustat[2 * u] = mean;
                    ustat[2 * u + 1] = 1.0f / sqrtf(s2 * inv + p.gn_eps);
                }
            }
            STAMP(13);
            lds_barrier();
            STAMP(14);
            if (cok) {
#pragma unroll
                for (int i = 0; i < EPV; ++i) {
                    const int row = row0 + i * RSTEP;
                    const int m = m0 + row;
                    if (row < BM && m < M) {
                        const int sI = fast_div(row, P, rP);
                        const float* us = ustat + 2 * (sI * GPT);
                        f32x4 A, B;
#define LFVDM_GNC(k, f)                                                                        \
                        { const float* q = us + 2 * fast_div(c4 + k, gw, rgw); A.f = q[1] * gam.f; B.f = bet.f - q[0] * A.f; }
                        LFVDM_GNC(0, x) LFVDM_GNC(1, y) LFVDM_GNC(2, z) LFVDM_GNC(3, w)
#undef LFVDM_GNC
                        if (p.gn_film) {
                            const f32x4 sc = fsc[i] + (f32x4){1.f, 1.f, 1.f, 1.f};
                            A = A * sc;
                            B = B * sc + fsh[i];
                        }
                        f32x4 y = tv[i] * A + B;
                        if (p.gn_act == LFVDM_ACT_SILU) { y.x = silu_f(y.x); y.y = silu_f(y.y); y.z = silu_f(y.z); y.w = silu_f(y.w); }
                        if constexpr (CHAIN) st4_sc1(whole_rsrc(p.gn_out), (unsigned)(m * gld + co) * 4u, y);
                        else st4(p.gn_out + ((size_t)m * gld + co), y);
                    }
                }
            }
            }       // general form
        }
        STAMP(8);
    } else {
        // frame layout out[(n*Cout + co)*HoWo + pix]: consecutive threads take consecutive pixels
        for (int e = tid; e < BM * BN; e += CF::NTHREADS) {
            const int col = e / BM, row = e - col * BM;
            const int m = m0 + row, co = n0 + col;
            if (m >= M || co >= p.Cout) continue;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WK; ++w) t += smem[w * CF::GROUP_LDS + row * RED_LD + col];
            if (p.bias) t += p.bias[co];
            if (p.bias2) t += p.bias2[co];
            const int n = m / HoWo;
            if (p.res) {
                float r = p.res[(size_t)m * p.ldr + co];
                if (p.resA) r = r * p.resA[(size_t)n * p.Cout + co] + p.resB[(size_t)n * p.Cout + co];
                t += r;
            }
            p.out[((size_t)n * p.Cout + co) * HoWo + (m - n * HoWo)] = t;
        }
    }
    if constexpr (CHAIN) {
        // publish the tile: every storing wave has drained its write-through stores, then ONE lane sets the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid == 0) __hip_atomic_store(cx.flags + cx.flag_base + (int)tile_id, cx.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        STAMP(18);
    }
    return true;
}


// "tail split": tiles beyond the last full wave of workgroups (a multiple of the CU count) are K-split
constexpr int kHybridKz = 16;      // pseudo kz value selecting the tail-split launch (tune code l = 4)
constexpr long kNumCUs = 256;      // MI355X
struct HybridPlan { long nfull, tail; int kz; };
inline HybridPlan hybrid_plan(long tiles) {
    HybridPlan h;
    h.nfull = (tiles / kNumCUs) * kNumCUs;
    h.tail = tiles - h.nfull;
    long k = h.tail > 0 ? kNumCUs / h.tail : 1;
    h.kz = (int)(k < 2 ? 2 : k > 8 ? 8 : k);
    return h;
}

// zero-inserted source handled by output parity classes (see the kernel): the data gradient of a stride-2 convolution
inline bool parity_classes(const lfvdm_conv_args* a) {
    static const bool off = getenv("LFVDM_CONV_NO_PARITY") != nullptr;        // A/B aid
    return !off && a->up == 2 && a->ksize == 3 && a->stride == 1 && a->C1 == 0 && a->s2C0 + a->s2C1 == 0 && !a->gn_out &&
           !a->resA && a->out_mode == LFVDM_OUT_ROWS;
}

// LDS bytes of a configuration with `gl` stages
constexpr long glds_lds_bytes(int WM, int WN, int WK, int NT, int kch, int gl) {
    return (long)WK * gl * (32 * WM + 32 * NT * WN) * kch * 4;
}

// every tensor a launch stages must be addressable with 32-bit byte offsets below kOOB (2^30); larger batches are
// cut into sample ranges by lfvdm_conv_igemm
inline bool glds_ok(const lfvdm_conv_args* a) {
    const long Cin = a->C0 + a->C1, C2 = a->s2C0 + a->s2C1, lim = 1L << 30;
    const long Cmax = a->C0 > a->C1 ? a->C0 : a->C1, C2max = a->s2C0 > a->s2C1 ? a->s2C0 : a->s2C1;
    return (long)a->N * a->Hs * a->Ws * Cmax * 4 < lim && (long)a->N * a->Ho * a->Wo * C2max * 4 < lim &&
           (long)a->Cout * a->ksize * a->ksize * Cin * 4 < lim && (long)a->Cout * C2 * 4 < lim;
}


// Tile configurations: {WM, WN, WK, NT, waves/SIMD allowed by the VGPR allocation}.
struct TileCfg { int WM, WN, WK, NT, vgpr_waves; };
constexpr TileCfg kCfgs[] = {
    {2, 2, 1, 1, 4},  // 0: 64x64,  4 waves
    {2, 2, 1, 2, 3},  // 1: 64x128, 4 waves
    {1, 2, 2, 1, 3},  // 2: 32x64,  2 k-groups (4 waves)
    {1, 2, 4, 1, 3},  // 3: 32x64,  4 k-groups (8 waves)
    {1, 1, 8, 1, 2},  // 4: 32x32,  8 k-groups (8 waves)   (tiny M: low-resolution levels)
    {2, 2, 2, 1, 3},  // 5: 64x64,  2 k-groups (8 waves)
    {1, 1, 4, 1, 2},  // 6: 32x32,  4 k-groups (4 waves)   (narrow outputs, e.g. Cout = 4)
    {2, 2, 2, 2, 2},  // 7: 64x128, 2 k-groups (8 waves)
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

struct Pick { int id, kch, NK, kz, gl; };

// tune code (lfvdm_conv_args::tune): 0 = heuristic, else 1 + id + 16*(kch == 64) + 32*l + 256*(gl - 1), gl = 2 / 3
// LDS-DMA stages, l = index of the split-K factor in kKzTable: powers of two, the tail split, and 3 / 6 / 5 - a layer
// with 80 output tiles (64x64 tiles of a 128-filter conv on 8x8 maps) fills 160 of the 256 CUs at kz = 2 and takes
// 1.25 rounds at kz = 4; kz = 3 makes it 240 workgroups in one round
constexpr int kKzTable[8] = {1, 2, 4, 8, 16 /* = kHybridKz */, 3, 6, 5};
inline int encode_tune(int id, int kch, int kz, int gl) {
    int l = 0;
    while (l < 7 && kKzTable[l] != kz) ++l;
    return 1 + id + 16 * (kch == 64 ? 1 : 0) + 32 * l + 256 * (gl - 1);
}
// fused output GroupNorm: the tile must hold whole samples and whole groups, and the unit statistics must fit
// behind the reduction tile in the first k-group's LDS (checked against the smallest stage layout: 2 stages of 32 channels)
inline bool gn_tile_ok(const lfvdm_conv_args* a, int BM, int BN) {
    const int P = a->Ho * a->Wo, gw = a->gn_gw ? a->gn_gw : a->Cout / 32;
    if (a->Cout % 32 || a->out_mode != LFVDM_OUT_ROWS || P <= 0 || BM % P || gw <= 0 || BN % gw || a->Cout % 4) return false;
    const int U = (BM / P) * (BN / gw);
    return BM * (BN + 1) + 2 * U <= 2 * (BM + BN) * 32;
}
// is (tile id, chunk width, split-K, stages) a legal configuration for these arguments?
bool cfg_valid(const lfvdm_conv_args* a, int id, int kch, int kz, int gl) {
    const int Cin = a->C0 + a->C1, C2 = a->s2C0 + a->s2C1;
    if (id < 0 || id >= kNumCfgs || id == 7 || (kch != 32 && kch != 64) || kz < 1 || (kz > 8 && kz != kHybridKz)) return false;
    if (gl != 2 && gl != 3) return false;
    const TileCfg c = kCfgs[id];
    const int BM = 32 * c.WM, BN = 32 * c.NT * c.WN;
    if (a->Cout <= 32 && BN > 32) return false;
    if (a->gn_out && !gn_tile_ok(a, BM, BN)) return false;
    const bool can64 = Cin % 64 == 0 && a->C0 % 64 == 0 && C2 % 64 == 0 && a->s2C0 % 64 == 0;
    if (kch == 64 && (!can64 || c.NT > 1)) return false;
    const bool pc = parity_classes(a);
    // (parity classes: the lightest class walks ONE tap)
    const int NK = pc ? Cin / kch : a->ksize * a->ksize * (Cin / kch) + C2 / kch;
    if (c.WK > NK) return false;
    if (pc && kz == kHybridKz) return false;
    if (glds_lds_bytes(c.WM, c.WN, c.WK, c.NT, kch, gl) > 160 * 1024) return false;
    // split-K over workgroups needs the caller's workspace (slabs + tile tickets) and the rows layout
    if (kz > 1) {
        if (!a->splitk_ws || !a->splitk_cnt || a->out_mode != LFVDM_OUT_ROWS) return false;
        const long M = (long)a->N * a->Ho * a->Wo;
        const long tiles = (pc ? 4 * ((M / 4 + BM - 1) / BM) : (M + BM - 1) / BM) * ((a->Cout + BN - 1) / BN);
        if (kz == kHybridKz) {
            const HybridPlan h = hybrid_plan(tiles);
            if (h.nfull == 0 || h.tail == 0 || 4 * h.tail > 3 * kNumCUs || NK < h.kz * c.WK) return false;
            if (h.tail * h.kz * (long)(BM * BN) > a->splitk_ws_floats || h.tail > a->splitk_cnt_ints) return false;
        } else {
            if (NK < kz * c.WK) return false;
            if (tiles * kz * (long)(BM * BN) > a->splitk_ws_floats || tiles > a->splitk_cnt_ints) return false;
        }
    }
    return true;
}

}  // namespace
