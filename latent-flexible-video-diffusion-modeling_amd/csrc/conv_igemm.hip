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

template <int WM, int WN, int WK, int NT, int KCH, bool SIMPLE, int GL>
__global__ __launch_bounds__(64 * WM * WN * WK) void conv_igemm_kernel(const lfvdm_conv_args p_in, int hyb_nfull, int hyb_kz,
                                                                        int par_mt) {
    const lfvdm_conv_args p = p_in;   // private SSA copy: helpers take it by reference (keeps it out of scratch)
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
    const bool par = !SIMPLE && par_mt > 0;
    int pcls = 0;
    if (par) pcls = 3 - div_small((int)blockIdx.x, par_mt);
    const int ppy = pcls >> 1, ppx = pcls & 1;
    const int M = par ? p.N * p.Hs * p.Ws : p.N * HoWo;
    // Workgroup -> (output tile, K slice).  Plain launches: grid (m tiles, n tiles, KZ slices).  "Tail split"
    // launches (hyb_kz > 0, flat grid): the first hyb_nfull tiles - a multiple of the CU count - are computed whole,
    // each remaining tile by hyb_kz workgroups, so that the last, partial wave of tiles does not cost a full tile time.
    int bx = blockIdx.x, by = blockIdx.y, KZ = gridDim.z, kz = blockIdx.z;
    size_t tile_id = (size_t)by * gridDim.x + bx;          // ticket / slab index of this tile
    if (par) bx -= (3 - pcls) * par_mt;
    if (hyb_kz > 0) {
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
    } else if (hyb_kz < 0) {
        // XCD-aware flat grid (hyb_nfull = filter tiles, -hyb_kz = K slices, gridDim.x = 8 * per): consecutive workgroup
        // ids go round-robin to the 8 XCDs, each with its own L2.  The (filter tile, K slice, output-row tile) triples are
        // laid out pair-major and XCD x takes the contiguous range [x * per, (x + 1) * per): the row tiles of one (filter
        // tile, K slice) pair run on ONE XCD (two where a range boundary cuts the pair), so every filter byte is fetched
        // into one or two L2s instead of up to eight - at the low-resolution levels the filters ARE the traffic (590 KB of
        // filters against 82 KB of activations for a 128 -> 128 3x3 layer on 2x2 maps) - and every XCD gets the same
        // number of workgroups (+-1) whatever the pair count.  Traffic only: these launches wait on latency.
        const int MT = (M + BM - 1) / BM;
        KZ = -hyb_kz;
        const int id = blockIdx.x, per = (int)(gridDim.x >> 3);
        const int L = (id & 7) * per + (id >> 3);
        if (L >= MT * hyb_nfull * KZ) return;          // padding of the last range (whole workgroup, before any barrier)
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
                                                     (int)off, 0, 0, 0);
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
            const int row = e / QNs, c4 = (e - row * QNs) * 4;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < WK; ++w) {
                const float* r = smem + w * CF::GROUP_LDS + row * RED_LD + c4;
                t.x += r[0]; t.y += r[1]; t.z += r[2]; t.w += r[3];
            }
#ifdef LFVDM_SEAM_FENCES
            st4(slab + row * BN + c4, t);
#else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), rs_slab, (row * BN + c4) * 4, 0, 16 /* sc1 */);
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
                const int row = e / QNs, c4 = (e - row * QNs) * 4;
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
#ifdef LFVDM_SEAM_FENCES
                for (int z = 0; z < KZ; ++z) t += ld4(base + (size_t)z * (BM * BN) + row * BN + c4);
#else
                // all slab loads of the element in flight at once (KZ <= 8: cfg_valid), summed in slice order
                u32x4 sv[8];
#pragma unroll
                for (int z = 0; z < 8; ++z)
                    sv[z] = __builtin_amdgcn_raw_buffer_load_b128(rs_all, z < KZ ? (z * (BM * BN) + row * BN + c4) * 4 : -1, 0, 16 /* sc1 */);
#pragma unroll
                for (int z = 0; z < 8; ++z)
                    if (z < KZ) t += __builtin_bit_cast(f32x4, sv[z]);
#endif
                float* r = smem + row * RED_LD + c4;
                r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
            }
            lds_barrier();
            STAMP(6);
        }
    }
    if (!do_epilogue) return;
    const int WKE = (KZ > 1) ? 1 : WK;    // after a split-K combine the full sum sits in group 0's tile

    // ---- epilogue: a thread owns 4 consecutive output columns of EPV rows: bias is read once as a
    // float4, all residual loads are issued back to back, stores are 16-byte.
    constexpr int QN = BN / 4;                              // float4 per tile row
    constexpr int EPV = (BM * QN + CF::NTHREADS - 1) / CF::NTHREADS;   // float4 per thread
    const bool nchw = p.out_mode == LFVDM_OUT_NCHW;
    if (!nchw) {
        const int c4 = (tid % QN) * 4;
        const int row0 = tid / QN;
        constexpr int RSTEP = CF::NTHREADS / QN;
        static_assert(CF::NTHREADS % QN == 0, "column ownership");
        const int co = n0 + c4;
        const bool cok = co < p.Cout && row0 < BM;   // Cout % 4 == 0 is checked by the launcher for this layout
        const int cc = cok ? co : 0;
        f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bsum += ld4(p.bias + cc);
        if (p.bias2) bsum += ld4(p.bias2 + cc);
        f32x4 rv[EPV], ra[EPV], rb[EPV];
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
        int mo[EPV];
#pragma unroll
        for (int i = 0; i < EPV; ++i) {
            const int m = min(m0 + min(row0 + i * RSTEP, BM - 1), M - 1);
            mo[i] = out_row(m);
            rv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            ra[i] = (f32x4){1.f, 1.f, 1.f, 1.f};
            rb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p.res) rv[i] = ld4(p.res + (unsigned)(mo[i] * p.ldr + cc));
            if (p.resA) {
                const int n = fast_div(m, HoWo, rHoWo);
                ra[i] = ld4(p.resA + (unsigned)(n * p.Cout + cc));
                rb[i] = ld4(p.resB + (unsigned)(n * p.Cout + cc));
            }
        }
        const bool gn = p.gn_out != nullptr;
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
            if (store_raw && m < M && cok && row0 + i * RSTEP < BM) st4(p.out + ((size_t)mo[i] * p.ldo + co), t);
        }
        STAMP(7);
        if (gn) {
            // ---- fused GroupNorm(+FiLM)(+activation) of the output tile.  The launcher guarantees whole samples
            // (P = Ho*Wo divides BM) and whole groups (gw = Cout/32 divides BN) per tile.  The finished values go
            // back into group 0's LDS tile (each element is read and rewritten by its one owner thread), then one
            // exact two-pass mean / variance per (sample, group) unit, then the affine on the registers.
            // the per-channel / per-sample coefficients are fetched first: their latency hides behind the statistics
            const int P = HoWo, gw = p.Cout >> 5;
            const float rPF = __builtin_amdgcn_rcpf((float)(P * p.gn_film_div)), rP = __builtin_amdgcn_rcpf((float)P),
                        rgw = __builtin_amdgcn_rcpf((float)gw);       // fast_div operands here are < 2^21
            f32x4 gam = {0.f, 0.f, 0.f, 0.f}, bet = gam, fsc[EPV], fsh[EPV];
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
                    const float* fl = p.gn_film + (size_t)fast_div(m, P * p.gn_film_div, rPF) * p.gn_film_ld + cc;
                    fsc[i] = ld4(fl);
                    fsh[i] = ld4(fl + p.Cout);
                }
            }
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
                        st4(p.gn_out + ((size_t)m * p.Cout + co), y);
                    }
                }
            }
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
    const int P = a->Ho * a->Wo, gw = a->Cout / 32;
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
            b.gn_out = a->gn_out + n0 * P * a->Cout;
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

extern "C" int lfvdm_abi_version(void) { return 8; }
