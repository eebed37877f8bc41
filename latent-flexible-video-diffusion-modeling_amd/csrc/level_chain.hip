// Persistent level chain (include/lfvdm_hip.h, "Persistent level chain"): the implicit GEMMs and small-map GroupNorms of
// the low-resolution levels of one forward pass as ONE launch.  A stage is the stand-alone kernel's body (conv_igemm_body.h,
// gn_wave_body.h: same tiles, same K-slice order, same epilogue - bitwise the same values) run on work items instead of
// workgroups; launch boundaries become point-to-point tile flags (ChainCtx).  gfx950 only.
//
// Why flags and not a grid barrier (tools/grid_barrier_bench, profiles/r05_grid_barrier.json): a relaxed agent-scope add +
// sc1 poll on ONE word costs 13 ns per arrival or poll - 1.4 / 2.1 / 3.9 us per barrier among 64 / 128 / 256 workgroups -
// while a consumer that polls the flags of the 1-16 producer tiles it actually reads (one lane per flag, one load
// instruction) sees them 0.7-1.15 us after they were set, whatever the grid.  The producers' bytes travel without fences:
// sc1 (write-through) stores, drained, then the flag; sc1 loads (registers and LDS-DMA) on the consumer side.
#include <algorithm>
#include <array>
#include <map>
#include <set>
#include <vector>

#include "common_hip.h"

#ifdef LFVDM_CHAIN_STAMP
// diagnostic build only (tools/chain_stamps.py; never compiled into the product): thread 0 of every workgroup stamps the
// 100 MHz s_memrealtime clock, common to all CUs, at the phase boundaries of every stage it works on
constexpr int kCsStages = 64, kCsWGs = 256, kCsN = 20;
__device__ unsigned long long g_cstamps[kCsStages * kCsWGs * kCsN];
#define CSTAMP(stage, i) do { if (threadIdx.x == 0 && (stage) < kCsStages)                                                   \
        g_cstamps[((stage) * kCsWGs + blockIdx.x) * kCsN + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define STAMP(i) CSTAMP(cx.stage, i)
extern "C" int lfvdm_debug_chain_stamps(unsigned long long* host_out, int clear) {
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_cstamps), sizeof(g_cstamps)) != hipSuccess) return 2;
    if (clear) {
        static unsigned long long zeros[kCsStages * kCsWGs * kCsN];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_cstamps), zeros, sizeof(zeros)) != hipSuccess) return 2;
    }
    return 0;
}
#else
#define CSTAMP(stage, i) do {} while (0)
#define STAMP(i) do {} while (0)
#endif
#include "conv_igemm_body.h"
#include "gn_wave_body.h"
#include "conv_local_body.h"

namespace {

typedef const lfvdm_chain_stage __attribute__((address_space(4))) * StagePtr;     // constant address space: scalar loads
typedef const int __attribute__((address_space(4))) * DepPtr;

// kernel-body instances a stage can ask for: tile id 6 = <1,1,4,1> / id 2 = <1,2,2,1> (both 4 waves), 32-channel chunks,
// 2 / 3 LDS-DMA stages, one-source ("simple") or general operand form
constexpr int kChainThreads = 256;
inline int chain_cfg_index(int id, int gl, bool simple) { return (id == 2 ? 4 : 0) + (gl == 3 ? 2 : 0) + (simple ? 0 : 1); }
inline bool chain_simple(const lfvdm_conv_args* a) { return a->C1 == 0 && a->s2C0 + a->s2C1 == 0 && a->up == 0; }

template <int WM, int WN, int WK, int NT, bool SIMPLE, int GL>
__device__ __forceinline__ bool run_conv(StagePtr st, const ChainCtx& cx) {
    const lfvdm_conv_args p = *(const lfvdm_conv_args*)&st->conv;   // constant address space, uniform index: scalar loads
    return conv_igemm_body<WM, WN, WK, NT, 32, SIMPLE, GL, true>(p, st->nt2, -st->kz, 0, cx);
}

__global__ __launch_bounds__(kChainThreads) void level_chain_kernel(const lfvdm_chain_stage* stages_g, int n_stages,
                                                                    const int* deps_g, int* flags, int* ctl,
                                                                    long long timeout_ticks) {
    const StagePtr stages = (StagePtr)(uintptr_t)stages_g;
    const DepPtr deps = (DepPtr)(uintptr_t)deps_g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gen = __builtin_amdgcn_readfirstlane(
                        __hip_atomic_load(ctl + LFVDM_CHAIN_CTL_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + 1;
    __shared__ int s_go2;
    bool alive = true;
    // this workgroup's work items in order: stage s, items first(s), first(s) + wg_count, ... where the stage's workgroups are
    // [wg_lo, wg_lo + wg_count) and first(s) = (blockIdx - wg_lo - wg_off) mod wg_count
    auto first_of = [&](int s_) {
        const int b = (int)blockIdx.x - stages[s_].wg_lo, cnt = stages[s_].wg_count;
        if (b < 0 || b >= cnt) return 0x3fffffff;            // not one of the stage's workgroups: no item
        const int f = b - stages[s_].wg_off;                  // (0 <= wg_off < wg_count: lfvdm_chain_plan)
        return f < 0 ? f + cnt : f;
    };
    auto settle = [&](int& s_, int& item_) {
        while (s_ < n_stages && item_ >= stages[s_].n_items) {
            ++s_;
            if (s_ < n_stages) item_ = first_of(s_);
        }
    };
    int s = 0, item = first_of(0);
    settle(s, item);
    // (res_stage, res_item): the LOCAL item whose filter slice this workgroup requested while it finished an earlier one
    int res_stage = -1, res_item = 0, pf_stage = -1, pf_item = 0;
    while (s < n_stages && alive) {
        int s2 = s, item2 = item + stages[s].wg_count;
        settle(s2, item2);
        const StagePtr st = stages + s;
        const int kind = st->kind, n_items = st->n_items, dstride = st->dep_stride, dbase = st->dep_base;
        {
            // the previous item's LDS tiles are no longer read (LDS traffic only where a prefetched filter slice is in flight:
            // __syncthreads() would drain the DMA queue)
            if (res_stage >= 0) lds_barrier();
            else __syncthreads();
            const DepPtr dl = deps + dbase + (size_t)item * dstride;
            ChainCtx cx;
            cx.item = item;
            cx.n_items = n_items;
            cx.ndeps = dl[0];
            cx.deps = (const int*)(deps_g + dbase + (size_t)item * dstride + 1);
            cx.flags = flags;
            cx.flag_base = st->flag_base;
            cx.gen = gen;
            cx.abort_word = ctl + LFVDM_CHAIN_CTL_ABORT;
            cx.timeout_ticks = timeout_ticks;
            cx.stage = s;
            if (kind == LFVDM_CHAIN_LOCAL) {
                const lfvdm_conv_args p = *(const lfvdm_conv_args*)&st->conv;
                // a LOCAL item's row: [n0, n1, n2, n3, r0, r1 | operand producers of wave 0, 1, 2, 3 | residual producers of row
                // tile 0, 1]: a wave waits for the tiles of the channel quarter it stages; the wave that finishes a row tile
                // looks at that tile's residual producers separately (conv_local_body)
                const int c0 = dl[0], c1 = dl[1], c2 = dl[2], c3 = dl[3], r0 = dl[4], r1 = dl[5];
                const int* lists = (const int*)(deps_g + dbase + (size_t)item * dstride + 6);
                cx.ndeps = wave == 0 ? c0 : wave == 1 ? c1 : wave == 2 ? c2 : c3;
                cx.deps = lists + (wave == 0 ? 0 : wave == 1 ? c0 : wave == 2 ? c0 + c1 : c0 + c1 + c2);
                const int* res_deps = lists + c0 + c1 + c2 + c3 + (wave == 1 ? r0 : 0);
                const int res_n = wave == 0 ? r0 : wave == 1 ? r1 : 0;
                // the next LOCAL item of this workgroup: its filter slice is requested while this one finishes.  GroupNorm
                // items in between use no LDS and are looked past; a tile-kernel item would overwrite the slice
                int sp = s2, ip = item2;
                while (sp < n_stages && stages[sp].kind == LFVDM_CHAIN_GN) {
                    ip += stages[sp].wg_count;
                    settle(sp, ip);
                }
                LocalNext nx;
                nx.valid = sp < n_stages && stages[sp].kind == LFVDM_CHAIN_LOCAL;
                nx.W = nx.W2 = nullptr;
                nx.K1 = nx.C2 = nx.Cout = nx.slice = nx.front = 0;
                if (nx.valid) {
                    const StagePtr sn = stages + sp;
                    const int NSn = sn->conv.Cout >> 4;
                    nx.W = sn->conv.W;
                    nx.W2 = sn->conv.W2;
                    nx.K1 = sn->conv.ksize * sn->conv.ksize * (sn->conv.C0 + sn->conv.C1);
                    nx.C2 = sn->conv.s2C0 + sn->conv.s2C1;
                    nx.Cout = sn->conv.Cout;
                    nx.slice = ip - div_small(ip, NSn) * NSn;
                    nx.front = sn->kz;
                    pf_stage = sp;
                    pf_item = ip;
                }
                const bool resident = res_stage == s && res_item == item;
                // instances: row tiles x filter size x units per tap and wave (Cin / 64: the 64 / 128 / 256-channel layers
                // get straight-line K loops, other widths the run-time loop)
                const int cpw = (p.C0 + p.C1) >> 6;
                const int inst = (st->cfg == 2 ? 8 : 0) + (p.ksize == 3 ? 4 : 0) + (cpw == 1 ? 1 : cpw == 2 ? 2 : cpw == 4 ? 3 : 0);
                switch (inst) {
#define LFVDM_LOCAL_CASE(I, RT_, KS_, CPW_) case I: alive = conv_local_body<RT_, KS_, CPW_>(p, st->kz, resident, nx, cx, res_deps, res_n); break;
                    LFVDM_LOCAL_CASE(0, 1, 1, 0) LFVDM_LOCAL_CASE(1, 1, 1, 1) LFVDM_LOCAL_CASE(2, 1, 1, 2) LFVDM_LOCAL_CASE(3, 1, 1, 4)
                    LFVDM_LOCAL_CASE(4, 1, 3, 0) LFVDM_LOCAL_CASE(5, 1, 3, 1) LFVDM_LOCAL_CASE(6, 1, 3, 2) LFVDM_LOCAL_CASE(7, 1, 3, 4)
                    LFVDM_LOCAL_CASE(8, 2, 1, 0) LFVDM_LOCAL_CASE(9, 2, 1, 1) LFVDM_LOCAL_CASE(10, 2, 1, 2) LFVDM_LOCAL_CASE(11, 2, 1, 4)
                    LFVDM_LOCAL_CASE(12, 2, 3, 0) LFVDM_LOCAL_CASE(13, 2, 3, 1) LFVDM_LOCAL_CASE(14, 2, 3, 2)
                    default: alive = conv_local_body<2, 3, 4>(p, st->kz, resident, nx, cx, res_deps, res_n); break;
#undef LFVDM_LOCAL_CASE
                }
                res_stage = nx.valid ? pf_stage : -1;
                res_item = pf_item;
            } else if (kind == LFVDM_CHAIN_CONV) {
                res_stage = -1;               // (the tile body's stages overwrite a slice fetched ahead: never requested across one)
                switch (st->cfg) {
                    case 0: alive = run_conv<1, 1, 4, 1, true, 2>(st, cx); break;
                    case 1: alive = run_conv<1, 1, 4, 1, false, 2>(st, cx); break;
                    case 2: alive = run_conv<1, 1, 4, 1, true, 3>(st, cx); break;
                    case 3: alive = run_conv<1, 1, 4, 1, false, 3>(st, cx); break;
                    case 4: alive = run_conv<1, 2, 2, 1, true, 2>(st, cx); break;
                    case 5: alive = run_conv<1, 2, 2, 1, false, 2>(st, cx); break;
                    case 6: alive = run_conv<1, 2, 2, 1, true, 3>(st, cx); break;
                    default: alive = run_conv<1, 2, 2, 1, false, 3>(st, cx); break;
                }
            } else {
                // GroupNorm item = 4 waves = 4 consecutive (sample, 16-channel) units of one sample
                const lfvdm_gn_args g = *(const lfvdm_gn_args*)&st->gn;
                CSTAMP(s, 0);
                CSTAMP(s, 16);
                if (wave == 0) {
                    const bool ok = chain_poll(cx, lane);
                    if (lane == 0) s_go2 = ok ? 1 : 0;
                }
                __syncthreads();
                if (!s_go2) {
                    alive = false;
                    break;
                }
                CSTAMP(s, 17);
                const int C = g.C0 + g.C1, cbs = C >> 4, cg = g.cg ? g.cg : C >> 5, ldo = g.ldo ? g.ldo : C;
                const int u = item * 4 + wave;
                if (u < g.N * cbs) {
                    const int n = u / cbs, cb = u - n * cbs;
#define LFVDM_GNB(G)                                                                                                          \
    gn_wave_body<G, true>(n, cb, lane, g.src0, g.src1, g.C0, g.C1, g.P, g.gamma, g.beta, g.film, g.film_div, g.film_ld, g.eps, \
                          nullptr, nullptr, nullptr, g.out, g.act, ldo)
                    if (cg == 2) LFVDM_GNB(2);
                    else if (cg == 4) LFVDM_GNB(4);
                    else if (cg == 8) LFVDM_GNB(8);
                    else LFVDM_GNB(16);
#undef LFVDM_GNB
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0)
                    __hip_atomic_store(flags + cx.flag_base + item, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                CSTAMP(s, 18);
            }
        }
        s = s2;
        item = item2;
    }
    // a filter slice requested for an item this workgroup never reached (abort) must land before the LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // last workgroup out advances the generation (the next launch is stream-ordered behind this one)
    __syncthreads();
    if (tid == 0) {
        const int t = __hip_atomic_fetch_add(ctl + LFVDM_CHAIN_CTL_EXIT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (int)gridDim.x - 1) {
            __hip_atomic_store(ctl + LFVDM_CHAIN_CTL_EXIT, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctl + LFVDM_CHAIN_CTL_EPOCH, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------ host-side planning
struct Pick2 { int id, kch, kz, gl; };
inline bool decode_chain_code(const lfvdm_conv_args* a, Pick2* pk) {
    if (a->tune <= 0) return false;
    const int t = a->tune - 1;
    pk->id = t & 15;
    pk->kch = (t & 16) ? 64 : 32;
    pk->kz = kKzTable[(t >> 5) & 7];
    pk->gl = ((t >> 8) & 3) + 1;
    return true;
}

bool conv_stage_ok(const lfvdm_conv_args* a, Pick2* pk) {
    if (!decode_chain_code(a, pk)) return false;
    if ((pk->id != 6 && pk->id != 2) || pk->kch != 32 || pk->kz < 1 || pk->kz > 8 || (pk->gl != 2 && pk->gl != 3)) return false;
    if (a->out_mode != LFVDM_OUT_ROWS || a->coefA || a->coefB || a->resA || parity_classes(a) || a->up == 2) return false;
    const int Cin = a->C0 + a->C1, C2 = a->s2C0 + a->s2C1;
    if (a->N <= 0 || a->Cout <= 0 || Cin <= 0 || Cin % 32 || a->C0 % 32 || C2 % 32 || a->s2C0 % 32 || a->Cout % 4 || a->ldo % 4) return false;
    if (a->res && a->ldr % 4) return false;
    if (!glds_ok(a)) return false;
    // 32-bit byte offsets of the sc1 epilogue accesses
    const long M = (long)a->N * a->Ho * a->Wo;
    if (M * std::max(std::max(a->ldo, a->gn_ld), std::max(a->ldr, a->Cout)) * 4 >= (1L << 31)) return false;
    lfvdm_conv_args b = *a;        // capacity checks of cfg_valid against a notional workspace: the plan sizes the real one
    b.splitk_ws = (float*)(uintptr_t)16;
    b.splitk_cnt = (int32_t*)(uintptr_t)16;
    b.splitk_ws_floats = 1L << 40;
    b.splitk_cnt_ints = 1L << 40;
    return cfg_valid(&b, pk->id, pk->kch, pk->kz, pk->gl);
}

bool gn_stage_ok(int C0, int C1, int N, int P, int cg_given = 0) {
    const int C = C0 + C1, cg = cg_given ? cg_given : C / 32;
    return N > 0 && P > 0 && P <= 256 && C % 64 == 0 && C0 % 16 == 0 && (cg == 2 || cg == 4 || cg == 8 || cg == 16) &&
           (long)N * P * C * 4 < (1L << 31);
}

// who wrote a buffer inside the chain, and in which units
struct Writer {
    int stage;
    bool conv;
    int BM, BN, MT;       // conv: tile rows / columns, row tiles;  flag = flag_base + by * MT + bx
    int P, cbs4;          // gn: rows per sample, 64-column blocks per sample;  flag = flag_base + n * cbs4 + j
    int flag_base;
    long rows, cols;
    long col0;            // first column of the buffer this writer covers (a concat operand written half by half)
};

}  // namespace

extern "C" int lfvdm_chain_conv_ok(const lfvdm_conv_args* a) {
    Pick2 pk;
    return conv_stage_ok(a, &pk) ? LFVDM_OK : LFVDM_E_UNSUPPORTED;
}

extern "C" int lfvdm_chain_gn_ok(int C0, int C1, int N, int P) { return gn_stage_ok(C0, C1, N, P) ? LFVDM_OK : LFVDM_E_UNSUPPORTED; }

extern "C" int lfvdm_chain_local_ok(const lfvdm_conv_args* a, int row_tiles) {
    return a && local_stage_ok(a, row_tiles) ? LFVDM_OK : LFVDM_E_UNSUPPORTED;
}

// Workgroups of the chain kernel that can be resident at once on the current device with `lds_bytes` of dynamic LDS: the
// persistent kernel's waits only end if every workgroup of its grid is running (a partitioned or CU-masked device, or a part
// with fewer CUs, has less room than the 256 of a whole MI355X).  < 0: the query failed.
extern "C" int lfvdm_chain_capacity(int lds_bytes) {
    if (lds_bytes < 0 || lds_bytes > 160 * 1024) return -LFVDM_E_SHAPE;
    static DynLdsLimit limit;
    if (limit.ensure(reinterpret_cast<const void*>(&level_chain_kernel), (size_t)lds_bytes)) return -LFVDM_E_LAUNCH;
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return -LFVDM_E_LAUNCH;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, level_chain_kernel, kChainThreads, (size_t)lds_bytes) != hipSuccess)
        return -LFVDM_E_LAUNCH;
    // the occupancy API can come out one block per CU high for some SGPR counts (MI355X_MICROARCH.md, Correctness
    // boundaries); a chain never needs more than its LDS lets a CU hold
    const int by_lds = lds_bytes > 0 ? (160 * 1024) / lds_bytes : 8;
    per_cu = std::min(per_cu, std::max(by_lds, 1));
    return cus * std::max(per_cu, 0);
}

extern "C" int lfvdm_chain_plan(lfvdm_chain_stage* stages, int n_stages, int32_t* deps, int64_t deps_cap, int64_t* deps_used,
                                int32_t* n_flags_out, int64_t* ws_floats, int64_t* cnt_ints, int32_t* grid_out, int32_t* lds_out) {
    if (!stages || n_stages <= 0 || !deps || !deps_used || !n_flags_out || !ws_floats || !cnt_ints || !grid_out || !lds_out)
        return LFVDM_E_SHAPE;
    std::multimap<const float*, Writer> writers;      // (a buffer may have several writers, each with its own columns)
    std::set<const float*> was_read;         // a later write to a buffer that was read (or to columns that were written)
                                             // has no launch boundary to order it: refused
    long nflags = 0, ndeps = 0, ws = 0, cnt = 0;
    int grid = 1, lds = 0;
    // *grid_out on entry: the most workgroups the caller can keep resident (lfvdm_chain_capacity; <= 0: a whole MI355X)
    const int grid_cap = *grid_out > 0 ? std::min(*grid_out, 256) : 256;
    bool any_local = false;

    // flags of the units of `w` that overlap rows [r0, r1) x columns [c0, c1)
    auto add_region = [&](std::vector<int>& out, const float* base, long r0, long r1, long c0, long c1) {
        if (!base || r1 <= r0 || c1 <= c0) return;
        auto range = writers.equal_range(base);     // (none: produced by an earlier launch, ordered by the launch boundary)
        for (auto it = range.first; it != range.second; ++it) {
            const Writer& w = it->second;
            const long rr1 = std::min(r1, w.rows);
            const long a0 = std::max(c0, w.col0) - w.col0, a1 = std::min(c1, w.col0 + w.cols) - w.col0;   // in the writer's columns
            if (rr1 <= r0 || a1 <= a0) continue;
            if (w.conv) {
                for (long by = a0 / w.BN; by <= (a1 - 1) / w.BN; ++by)
                    for (long bx = r0 / w.BM; bx <= (rr1 - 1) / w.BM; ++bx) out.push_back(w.flag_base + (int)(by * w.MT + bx));
            } else {
                for (long n = r0 / w.P; n <= (rr1 - 1) / w.P; ++n)
                    for (long j = a0 / 64; j <= (a1 - 1) / 64; ++j) out.push_back(w.flag_base + (int)(n * w.cbs4 + j));
            }
        }
    };
    // columns [c0, c1) of a virtual concat (a | b) -> regions of a and b
    auto add_cat = [&](std::vector<int>& out, const float* a, int Ca, const float* b, long r0, long r1, long c0, long c1) {
        if (c0 < Ca) add_region(out, a, r0, r1, c0, std::min<long>(c1, Ca));
        if (c1 > Ca) add_region(out, b, r0, r1, std::max<long>(c0, Ca) - Ca, c1 - Ca);
    };
    auto note_write = [&](const float* base, long col0, long cols) {
        if (!base) return true;
        if (was_read.count(base)) return false;
        auto range = writers.equal_range(base);
        for (auto it = range.first; it != range.second; ++it)
            if (col0 < it->second.col0 + it->second.cols && it->second.col0 < col0 + cols) return false;
        return true;
    };

    for (int s = 0; s < n_stages; ++s) {
        lfvdm_chain_stage& st = stages[s];
        st.flag_base = (int)nflags;
        st.dep_base = (int)ndeps;
        st.ws_off = st.cnt_off = 0;
        std::vector<std::vector<int>> item_deps;
        std::vector<std::array<int, 6>> local_counts;       // LOCAL stages: lengths of an item's four per-wave lists + two residual lists
        if (st.kind == LFVDM_CHAIN_CONV) {
            const lfvdm_conv_args& a = st.conv;
            Pick2 pk;
            if (!conv_stage_ok(&a, &pk)) return LFVDM_E_UNSUPPORTED;
            const TileCfg tc = kCfgs[pk.id];
            const int BM = 32 * tc.WM, BN = 32 * tc.NT * tc.WN, KC = 32;
            const long M = (long)a.N * a.Ho * a.Wo;
            const int MT = (int)((M + BM - 1) / BM), NT2 = (a.Cout + BN - 1) / BN, KZ = pk.kz;
            const long total = (long)MT * NT2 * KZ, per = (total + 7) / 8;
            if (8 * per >= (1L << 20)) return LFVDM_E_UNSUPPORTED;
            st.n_items = (int)(8 * per);
            st.n_flags = MT * NT2;
            st.cfg = chain_cfg_index(pk.id, pk.gl, chain_simple(&a));
            st.kz = KZ;
            st.nt2 = NT2;
            if (KZ > 1) {
                st.ws_off = ws;
                st.cnt_off = cnt;
                ws += (long)MT * NT2 * KZ * BM * BN;
                cnt += (long)MT * NT2;
            }
            lds = std::max(lds, (int)glds_lds_bytes(tc.WM, tc.WN, tc.WK, tc.NT, KC, pk.gl));
            const int Cin = a.C0 + a.C1, taps = a.ksize * a.ksize;
            const int NK1 = taps * (Cin / KC), NK = NK1 + (a.s2C0 + a.s2C1) / KC;
            const long HoWo = (long)a.Ho * a.Wo, Ps = (long)a.Hs * a.Ws;
            item_deps.resize(st.n_items);
            for (int id = 0; id < st.n_items; ++id) {
                const long L = (long)(id & 7) * per + (id >> 3);       // the kernel's XCD-aware map
                if (L >= total) continue;                              // padding item
                const int pair = (int)(L / MT), bx = (int)(L - (long)pair * MT), by = pair / KZ, kz = pair - by * KZ;
                const long m0 = (long)bx * BM, m1 = std::min<long>(m0 + BM, M);
                const long n_lo = m0 / HoWo, n_hi = (m1 - 1) / HoWo;
                std::vector<int>& d = item_deps[id];
                const int zbeg = (int)((long)NK * kz / KZ), zend = (int)((long)NK * (kz + 1) / KZ);
                for (int k = zbeg; k < zend; ++k) {
                    if (k < NK1) {           // main segment: whole source samples of the tile's rows (halo included)
                        const int ci = k / taps;
                        add_cat(d, a.src0, a.C0, a.src1, n_lo * Ps, (n_hi + 1) * Ps, (long)ci * KC, (long)ci * KC + KC);
                    } else {                 // 1x1 skip segment at output resolution
                        const int ci = k - NK1;
                        add_cat(d, a.s2src0, a.s2C0, a.s2src1, m0, m1, (long)ci * KC, (long)ci * KC + KC);
                    }
                }
                // the epilogue runs in whichever slice arrives last: every slice of the tile carries its reads
                add_region(d, a.res, m0, m1, (long)by * BN, std::min<long>((long)by * BN + BN, a.Cout));
                std::sort(d.begin(), d.end());
                d.erase(std::unique(d.begin(), d.end()), d.end());
                if ((int)d.size() > LFVDM_CHAIN_MAX_DEPS) return LFVDM_E_UNSUPPORTED;
            }
            for (const float* r : {a.src0, a.src1, a.s2src0, a.s2src1, a.res})
                if (r) was_read.insert(r);
            const bool raw = !(a.gn_out && a.gn_skip_raw);
            const Writer w{s, true, BM, BN, MT, 0, 0, st.flag_base, M, a.Cout, 0};
            if (raw) {
                if (!note_write(a.out, 0, a.Cout)) return LFVDM_E_UNSUPPORTED;
                writers.insert({a.out, w});
            }
            if (a.gn_out) {         // (gn_ld > Cout: the left part of a wider operand; columns [0, Cout) of it)
                if (!note_write(a.gn_out, 0, a.Cout)) return LFVDM_E_UNSUPPORTED;
                writers.insert({a.gn_out, w});
            }
        } else if (st.kind == LFVDM_CHAIN_LOCAL) {
            // sample-local stage (conv_local_body.h): item = (16 * rt rows = whole samples, 16 filters, all of K)
            const lfvdm_conv_args& a = st.conv;
            const int rt = st.cfg;
            if (!local_stage_ok(&a, rt)) return LFVDM_E_UNSUPPORTED;
            const int rows = 16 * rt, P = a.Ho * a.Wo, NS = a.Cout / kLocalFS;
            const long M = (long)a.N * P, Ps = (long)a.Hs * a.Ws;
            const int MT = (int)((M + rows - 1) / rows);
            if ((long)MT * NS >= (1L << 20)) return LFVDM_E_UNSUPPORTED;
            const int MT16 = (int)((M + 15) / 16);           // flags: one per (filter slice, 16-row tile)
            st.n_items = MT * NS;
            st.n_flags = MT16 * NS;
            st.kz = (int)local_front_floats(&a, rt);
            st.nt2 = NS;
            lds = std::max(lds, (int)((local_front_floats(&a, rt) + local_filter_floats(&a)) * 4));
            any_local = true;
            const int Cin = a.C0 + a.C1, C2 = a.s2C0 + a.s2C1;
            item_deps.resize(st.n_items);
            local_counts.assign(st.n_items, {0, 0, 0, 0, 0, 0});
            for (int id = 0; id < st.n_items; ++id) {
                const int rg = id / NS, slice = id - rg * NS;
                const long m0 = (long)rg * rows, m1 = std::min<long>(m0 + rows, M);
                const long n_lo = m0 / P, n_hi = (m1 - 1) / P;
                // one list per wave: wave w stages and multiplies channel quarter w of the main source and of the skip segment
                for (int w = 0; w < 4; ++w) {
                    std::vector<int> d;
                    add_cat(d, a.src0, a.C0, a.src1, n_lo * Ps, (n_hi + 1) * Ps, (long)w * Cin / 4, (long)(w + 1) * Cin / 4);
                    if (C2 > 0) add_cat(d, a.s2src0, a.s2C0, a.s2src1, m0, m1, (long)w * C2 / 4, (long)(w + 1) * C2 / 4);
                    std::sort(d.begin(), d.end());
                    d.erase(std::unique(d.begin(), d.end()), d.end());
                    if ((int)d.size() > LFVDM_CHAIN_MAX_DEPS) return LFVDM_E_UNSUPPORTED;
                    local_counts[id][w] = (int)d.size();
                    item_deps[id].insert(item_deps[id].end(), d.begin(), d.end());
                }
                for (int t = 0; t < 2; ++t) {       // residual tiles: one list per row tile
                    std::vector<int> d;
                    if (t < rt && m0 + 16L * t < m1)
                        add_region(d, a.res, m0 + 16L * t, std::min<long>(m0 + 16L * t + 16, m1), (long)slice * kLocalFS, (long)(slice + 1) * kLocalFS);
                    std::sort(d.begin(), d.end());
                    d.erase(std::unique(d.begin(), d.end()), d.end());
                    if ((int)d.size() > LFVDM_CHAIN_MAX_DEPS) return LFVDM_E_UNSUPPORTED;
                    local_counts[id][4 + t] = (int)d.size();
                    item_deps[id].insert(item_deps[id].end(), d.begin(), d.end());
                }
            }
            for (const float* r : {a.src0, a.src1, a.s2src0, a.s2src1, a.res})
                if (r) was_read.insert(r);
            const bool raw = !(a.gn_out && a.gn_skip_raw);
            const Writer w{s, true, 16, kLocalFS, MT16, 0, 0, st.flag_base, M, a.Cout, 0};     // flag = base + slice * MT16 + 16-row tile
            if (raw) {
                if (!note_write(a.out, 0, a.Cout)) return LFVDM_E_UNSUPPORTED;
                writers.insert({a.out, w});
            }
            if (a.gn_out) {
                if (!note_write(a.gn_out, 0, a.Cout)) return LFVDM_E_UNSUPPORTED;
                writers.insert({a.gn_out, w});
            }
        } else if (st.kind == LFVDM_CHAIN_GN) {
            const lfvdm_gn_args& g = st.gn;
            if (!gn_stage_ok(g.C0, g.C1, g.N, g.P, g.cg) || !g.src0 || !g.out || (g.C1 > 0 && !g.src1) || (g.film && g.film_div <= 0))
                return LFVDM_E_UNSUPPORTED;
            if ((g.ldo && (g.ldo < g.C0 + g.C1 || g.ldo % 4)) || (long)g.N * g.P * std::max(g.ldo, g.C0 + g.C1) * 4 >= (1L << 31) ||
                (g.out_base && (g.out_col < 0 || g.out != g.out_base + g.out_col)))
                return LFVDM_E_SHAPE;
            const int C = g.C0 + g.C1, cbs4 = C / 64;
            st.n_items = g.N * cbs4;
            st.n_flags = st.n_items;
            st.cfg = st.kz = st.nt2 = 0;
            item_deps.resize(st.n_items);
            for (int id = 0; id < st.n_items; ++id) {
                const int n = id / cbs4, j = id - n * cbs4;
                std::vector<int>& d = item_deps[id];
                add_cat(d, g.src0, g.C0, g.src1, (long)n * g.P, (long)(n + 1) * g.P, 64L * j, 64L * j + 64);
                std::sort(d.begin(), d.end());
                d.erase(std::unique(d.begin(), d.end()), d.end());
                if ((int)d.size() > LFVDM_CHAIN_MAX_DEPS) return LFVDM_E_UNSUPPORTED;
            }
            was_read.insert(g.src0);
            if (g.src1) was_read.insert(g.src1);
            const float* base = g.out_base ? g.out_base : g.out;
            if (!note_write(base, g.out_col, C)) return LFVDM_E_UNSUPPORTED;
            writers.insert({base, Writer{s, false, 0, 0, 0, g.P, cbs4, st.flag_base, (long)g.N * g.P, C, g.out_col}});
        } else {
            return LFVDM_E_SHAPE;
        }
        size_t maxd = 0;
        for (const auto& d : item_deps) maxd = std::max(maxd, d.size());
        const int head = st.kind == LFVDM_CHAIN_LOCAL ? 6 : 1;      // count(s) in front of the flag indices
        st.dep_stride = (int)maxd + head;
        if (ndeps + (long)st.n_items * st.dep_stride > deps_cap) return LFVDM_E_SHAPE;
        for (int id = 0; id < st.n_items; ++id) {
            int32_t* row = deps + ndeps + (long)id * st.dep_stride;
            if (head == 6) for (int w = 0; w < 6; ++w) row[w] = local_counts[id][w];
            else row[0] = (int32_t)item_deps[id].size();
            for (size_t k = 0; k < maxd; ++k) row[head + k] = k < item_deps[id].size() ? item_deps[id][k] : 0;
        }
        ndeps += (long)st.n_items * st.dep_stride;
        nflags += st.n_flags;
        grid = std::max(grid, std::min(st.n_items, grid_cap));
    }
    // LOCAL stages rotate over the whole device: stage k + 1 starts on the workgroup behind stage k's last item, so a
    // workgroup gets an item every grid / n_items stages and its next filter slice has that long to arrive
    if (any_local) grid = grid_cap;
    // stages none of whose items waits for anything inside the chain (the skip half of a concat GroupNorm: its source
    // comes from an earlier launch) go to the workgroups at the top of the grid, which the GEMM stages - 24 ... 240 work
    // items, assigned from workgroup 0 upwards - mostly leave idle: they run at once, beside the chain's critical path
    // Workgroups.  Side stages (caller-marked: off the critical path) get the top 5/16 of the grid to themselves, the
    // main path the rest: a workgroup walks its items in stage order, so side work in front of a main-path item would hold
    // that item back.  Inside each range consecutive sample-local stages rotate.
    bool any_side = false;
    for (int s = 0; s < n_stages; ++s) any_side = any_side || stages[s].side != 0;
    int side_n = 0;
    if (any_side && grid >= 64) side_n = std::max(8, (grid * 5 / 16) / 8 * 8);       // 80 of 256: one round of an 80-item stage
    const int main_n = grid - side_n;
    int rot_main = 0, rot_side = 0;
    for (int s = 0; s < n_stages; ++s) {
        lfvdm_chain_stage& st = stages[s];
        const bool side = st.side != 0 && side_n > 0;
        st.wg_lo = side ? main_n : 0;
        st.wg_count = side ? side_n : main_n;
        st.wg_off = 0;
        int& rot = side ? rot_side : rot_main;
        if (st.kind == LFVDM_CHAIN_LOCAL || side) {
            st.wg_off = rot;
            rot = (rot + st.n_items) % st.wg_count;
            continue;
        }
        if (st.kind != LFVDM_CHAIN_GN || st.n_items > st.wg_count) continue;
        bool free_standing = true;
        for (int id = 0; id < st.n_items && free_standing; ++id) free_standing = deps[st.dep_base + (long)id * st.dep_stride] == 0;
        if (free_standing) st.wg_off = (st.wg_count - st.n_items) % st.wg_count;
    }
    *deps_used = ndeps;
    *n_flags_out = (int32_t)nflags;
    *ws_floats = ws;
    *cnt_ints = cnt;
    *grid_out = grid;
    *lds_out = lds;
    return LFVDM_OK;
}

extern "C" int lfvdm_level_chain(const lfvdm_chain_stage* stages_dev, int n_stages, const int32_t* deps_dev, int32_t* flags,
                                 int32_t* ctl, int grid, int lds_bytes, double timeout_s, void* stream) {
    if (!stages_dev || n_stages <= 0 || !deps_dev || !flags || !ctl || grid <= 0 || grid > 256 || lds_bytes < 0 ||
        lds_bytes > 160 * 1024 || !(timeout_s > 0))
        return LFVDM_E_SHAPE;
    static DynLdsLimit limit;
    if (int rc = limit.ensure(reinterpret_cast<const void*>(&level_chain_kernel), (size_t)lds_bytes)) return rc;
    {
        // co-residency is what ends the kernel's waits: refuse a grid the device cannot hold at once (queried once per
        // (device, LDS size); the planner was given the same number as its grid cap)
        static std::mutex mu;
        static std::map<std::pair<int, int>, int> cap;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return LFVDM_E_LAUNCH;
        std::lock_guard<std::mutex> lock(mu);
        auto it = cap.find({dev, lds_bytes});
        if (it == cap.end()) it = cap.insert({{dev, lds_bytes}, lfvdm_chain_capacity(lds_bytes)}).first;
        if (it->second < grid) return LFVDM_E_UNSUPPORTED;
    }
    const long long ticks = (long long)(timeout_s * 1.0e8);          // s_memrealtime: 100 MHz
    hipLaunchKernelGGL(level_chain_kernel, dim3((unsigned)grid), dim3(kChainThreads), (size_t)lds_bytes, (hipStream_t)stream,
                       stages_dev, n_stages, (const int*)deps_dev, (int*)flags, (int*)ctl, ticks);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}
