// Temporal attention with relative position terms (reference rpe.py:143-169), second generation.
//
//   logits[t][s] = q_t . (k_s + R_k[t][s]) + scale * k_s . R_q[s][t]     (q already scaled)
//   o[t]         = sum_s softmax_s(logits + two-clique mask)[s] * (v_s + R_v[t][s])
//
// per (batch b, head h, pixel p); T <= 32 frames, head dim F in {16, 32, 64} (others: the first kernel).  Same per-lane arithmetic as the first
// kernel (attention.hip: a lane owns one (pixel, query frame) and all T logits, softmax without cross-lane traffic),
// restructured around what that kernel measured as its limits on MI355X (tools/attn_bench.py):
//   * the workgroup's operands - the R_k / R_q / R_v slices of its query frames and the k / v rows of its pixels - are
//     staged ONCE, by LDS-DMA (buffer_load ... lds: no VGPR staging, no ds_write pass, no second phase), as images of
//     16-byte slots XOR-swizzled by row so that the b128 reads of a 16-lane group hit 16 different bank quads; R_q is
//     transposed by the DMA itself (every lane has its own source address);
//   * the query frames are split into TG groups: a workgroup needs 1/TG of every R slice, so two workgroups fit a CU
//     and even the 8x8 / 2x2 levels have >= 100 workgroups;
//   * workgroups that share 128-byte lines of R and k / v (adjacent heads when F < 32) and the frame groups of the
//     same pixels are placed on the same XCD (blockIdx % 8), so every line is fetched by one L2 only.
#include "common_hip.h"

namespace {

struct T2Geom {
    int T, P, C, heads, B;
    int TG, TGN, PPW, NW;      // frame groups, frames per group, pixels per wave, waves per workgroup
    int strips;                // pixel strips of NW*PPW pixels
    int NL, HL, XPG;           // line groups (b, head / HL), heads per 128-byte line, XCDs per line group (0: plain map)
};

template <int F, int TCAP>
struct T2Cfg {
    static constexpr int NQ = F / 4;
    static constexpr int RS = (TCAP * NQ + 15) / 16 * 16;     // 16-byte slots per image row (a multiple of 16)
};

// which (b, head, frame group, strip) a workgroup owns: XCD-aware decode of the flat block index
__device__ __forceinline__ void t2_decode(const T2Geom& g, int& b, int& h, int& tg, int& strip) {
    {
        const int id = blockIdx.x;
        if (g.XPG > 0) {
            const int x = id & 7, r = id >> 3;
            const int grp = x % g.NL, sub = x / g.NL;
            const int inner = r % (g.HL * g.TG), sblk = r / (g.HL * g.TG);
            strip = sblk * g.XPG + sub;
            const int hl = inner % g.HL;
            tg = inner / g.HL;
            const int hpb = g.heads / g.HL;                    // line groups per batch element
            b = grp / hpb;
            h = (grp % hpb) * g.HL + hl;
        } else {
            const int NP = g.B * g.heads;
            const int pair = id % NP, rest = id / NP;
            b = pair / g.heads;
            h = pair % g.heads;
            tg = rest % g.TG;
            strip = rest / g.TG;
        }
    }
}

// LDS images of a workgroup, in slots of 4 floats: R[tq][3][RS] (R_k | R_q transposed | R_v) at Rimg, KV[pixel][2][RS] at KVimg
template <int F, int TCAP>
__device__ __forceinline__ void t2_stage(const float* __restrict__ qkv, const float* __restrict__ Rq, const float* __restrict__ Rk,
                                         const float* __restrict__ Rv, RSel rsel, const T2Geom& g, int b, int h,
                                         int tq0, int p0, int wave, int lane, float* Rimg, float* KVimg) {
    using CF = T2Cfg<F, TCAP>;
    constexpr int NQ = CF::NQ, RS = CF::RS;
    const int T = g.T, P = g.P, C = g.C;
    const int NPX = g.NW * g.PPW;
    const float invRS = 1.0f / (float)RS;
    // ---- LDS-DMA staging (global_load_lds_dwordx4: 16 bytes per lane from a per-lane address to the wave's next 1 KiB
    // of LDS): pieces of 64 slots, dealt round-robin to the waves.  Slots nobody reads (row padding, frames past T,
    // pixels past P) are filled from a valid dummy address.
    {
        const size_t rb = rsel.slice(b, g.B);
        const float* Rsrc[3] = {Rk + rb * T * T * C + h * F, Rq + rb * T * T * C + h * F, Rv + rb * T * T * C + h * F};
        const float* kvsrc = qkv + (size_t)b * T * P * 3 * C + C + h * F;     // k of (b, frame 0, pixel 0); v is C further
        const int vpx = min(NPX, P - p0);                              // pixel rows that exist
        const int nR = (min(g.TGN, T - tq0) * 3 * RS + 63) >> 6, nKV = (vpx * 2 * RS + 63) >> 6;
        const int TNQ = T * NQ;
        auto dma = [](const float* src, float* dst) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        };
        for (int pc = wave; pc < nR; pc += g.NW) {
            const int ps = pc * 64 + lane;
            const int row3 = (int)(((float)ps + 0.5f) * invRS);        // (query frame, array) row
            const int w = ps - row3 * RS;
            const int tq = (int)(((float)row3 + 0.5f) * (1.0f / 3.0f)), arr = row3 - 3 * tq;
            const int c = w ^ (tq & 15);
            const int s = c / NQ, u = c - s * NQ;
            const int t = tq0 + tq;
            const bool ok = tq < g.TGN && t < T && c < TNQ;
            // R_q is needed as R_q[s][t] by query frame t: the DMA transposes (per-lane source address)
            const int row = arr == 1 ? s * T + t : t * T + s;
            const float* base = arr == 0 ? Rsrc[0] : (arr == 1 ? Rsrc[1] : Rsrc[2]);
            dma(ok ? base + (size_t)row * C + 4 * u : qkv, Rimg + (size_t)pc * 256);
        }
        for (int pc = wave; pc < nKV; pc += g.NW) {
            const int ps = pc * 64 + lane;
            const int row2 = (int)(((float)ps + 0.5f) * invRS);        // (pixel, k | v) row
            const int w = ps - row2 * RS;
            const int jw = row2 >> 1, arr = row2 & 1;
            const int c = w ^ (jw & 15);
            const int s = c / NQ, u = c - s * NQ;
            const int p = p0 + jw;
            const bool ok = jw < NPX && p < P && c < TNQ;
            dma(ok ? kvsrc + ((size_t)s * P + p) * 3 * C + arr * C + 4 * u : qkv, KVimg + (size_t)pc * 256);
        }
    }

}

template <int F, int TCAP>
__global__ __launch_bounds__(256, 3)          // three waves per SIMD (<= 168 VGPRs): the waves hide each other's LDS latency
void attn_temporal2_kernel(const float* __restrict__ qkv, const float* __restrict__ Rq, const float* __restrict__ Rk,
                           const float* __restrict__ Rv, const float* __restrict__ mask, float* __restrict__ o,
                           float* __restrict__ attn_out, RSel rsel, T2Geom g) {
    using CF = T2Cfg<F, TCAP>;
    constexpr int NQ = CF::NQ, RS = CF::RS;
    extern __shared__ __attribute__((aligned(16))) float t2_smem[];
    const int T = g.T, P = g.P, C = g.C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NPX = g.NW * g.PPW;
    int b, h, tg, strip;
    t2_decode(g, b, h, tg, strip);
    if (strip >= g.strips) return;                             // (grid padding of the XCD-aware map; workgroup-uniform)
    const int tq0 = tg * g.TGN;
    const int p0 = strip * NPX;
    // LDS images, in slots of 4 floats: R[tq][3][RS] (R_k | R_q transposed | R_v), then KV[pixel][2][RS]
    // (each image is padded to whole 64-slot DMA pieces: a piece always writes 1 KiB)
    float* Rimg = t2_smem;
    float* KVimg = t2_smem + (size_t)((g.TGN * 3 * RS + 63) & ~63) * 4;
    const float invRS = 1.0f / (float)RS;

    t2_stage<F, TCAP>(qkv, Rq, Rk, Rv, rsel, g, b, h, tq0, p0, wave, lane, Rimg, KVimg);

    // ---- FOUR lanes per (pixel, query frame): lane = (pair << 2) | sq, pair = (pixel j, frame tq), and lane sq owns
    // the key frames s = sq, sq + 4, ...  The whole problem is only ~700 (pixel, frame) waves for 1024 SIMDs, and a lone
    // wave cannot hide the ~100-cycle LDS latency of its ~400 dependent-looking reads (measured: the math phase took as
    // long on an empty chip as on a full one); a quarter of the keys per lane gives four times the waves with chains a
    // quarter as long.  Softmax and the output sum cross the four lanes with quad DPP moves.
    const float invN = 1.0f / (float)g.TGN;
    const int sq = lane & 3, pair = lane >> 2;
    const int j = (int)(((float)pair + 0.5f) * invN), tq = pair - j * g.TGN;
    const int jw = wave * g.PPW + j;
    const int t = tq0 + tq, p = p0 + jw;
    const bool active = j < g.PPW && p < P && t < T;
    const float scale = rsqrtf((float)F);
    const size_t ld = (size_t)3 * C;
    const float* qrow = active ? qkv + ((size_t)(b * T + t) * P + p) * ld + h * F : qkv;
    f32x4 q4[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) q4[u] = ld4(qrow + 4 * u) * scale;
    // slot of (key s = sq + 4 i, quad u) in a row: c = NQ s + u = 4 NQ i + (NQ sq + u), NQ a power of two >= 4: the low
    // four bits of c are those of NQ sq + u for every i and the bits above them add without carry.  So the swizzled address
    // is a per-lane base per quad (computed once) plus an immediate per i.
    constexpr int KQ = (TCAP + 3) / 4;                      // keys per lane
    constexpr int ISTRIDE = 4 * NQ * 4;                     // floats between consecutive i
    int kb[NQ], rbs[NQ];
    {
        const int jr = active ? jw : 0, tr = active ? tq : 0;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int low = (NQ * sq + u) & 15, high = (NQ * sq + u) & ~15;
            kb[u] = (jr * 2 * RS + high + (low ^ (jr & 15))) * 4;
            rbs[u] = (tr * 3 * RS + high + (low ^ (tr & 15))) * 4;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces have landed (q too)
    __syncthreads();                                       // ... and everybody else's

    // quad exchanges (DPP quad_perm): partner lanes sq ^ 1 and sq ^ 2
    auto quad_x1 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); };
    auto quad_x2 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); };

    {                            // (inactive lanes run along on row 0: the quad exchanges need all four lanes)
        float logit[KQ];
        // packed fp32 math on naturally adjacent register pairs (q, k, R rows arrive as b128 = two aligned pairs)
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const f32x4 k4 = ld4(KVimg + kb[u] + i * ISTRIDE);
                const f32x4 r4 = ld4(Rimg + rbs[u] + i * ISTRIDE);
                const f32x4 g4 = ld4(Rimg + rbs[u] + RS * 4 + i * ISTRIDE);
                a0 = __builtin_elementwise_fma(q4[u].xy, k4.xy + r4.xy, a0);
                a0 = __builtin_elementwise_fma(q4[u].zw, k4.zw + r4.zw, a0);
                a1 = __builtin_elementwise_fma(k4.xy, g4.xy, a1);
                a1 = __builtin_elementwise_fma(k4.zw, g4.zw, a1);
            }
            logit[i] = (a0.x + a0.y) + (a1.x + a1.y) * scale;
            // Left alone, the scheduler hoists all ~100 reads of the unrolled loops to the top and the register allocator
            // spills them straight to scratch (ds_read -> lgkmcnt(0) -> scratch_store): keep every key's reads with its math;
            // the other waves of the SIMD cover the latency.
            // ... the fence has to CONSUME the result, or instruction selection sinks the (side-effect free) math below it
            asm volatile("" : "+v"(logit[i]) : : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        // two-clique mask + softmax over the keys of the four lanes (rpe.py:156-163)
        const int tm = active ? t : 0;
        const float mt = mask ? mask[b * T + tm] : 1.f;
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            const int sk = sq + 4 * i;
            float v = -INFINITY;
            if (sk < T) {
                v = logit[i];
                if (mask) {
                    const float ms = mask[b * T + sk];
                    const float pen = 1.f - (mt * ms + (1.f - mt) * (1.f - ms));
                    v -= (pen == 1.f) ? INFINITY : pen;
                }
            }
            logit[i] = v;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, quad_x1(mx));
        mx = fmaxf(mx, quad_x2(mx));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            const float e = (logit[i] == -INFINITY) ? 0.f : __expf(logit[i] - mx);
            logit[i] = e;
            sum += e;
        }
        sum += quad_x1(sum);
        sum += quad_x2(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int i = 0; i < KQ; ++i) logit[i] *= inv;
        if (attn_out && active) {
            float* ar = attn_out + ((((size_t)b * P + p) * g.heads + h) * T + t) * T;
#pragma unroll
            for (int i = 0; i < KQ; ++i)
                if (sq + 4 * i < T) ar[sq + 4 * i] = logit[i];
        }
        // o[t][f] = sum_s p[s] * (v[s][f] + R_v[t][s][f]): partial sums over this lane's keys, then over the quad
        f32x4 acc[NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            const f32x2 pr = {logit[i], logit[i]};       // exactly 0 for key frames >= T (their slots hold finite filler)
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const f32x4 v4 = ld4(KVimg + kb[u] + RS * 4 + i * ISTRIDE);
                const f32x4 r4 = ld4(Rimg + rbs[u] + 2 * RS * 4 + i * ISTRIDE);
                f32x2 lo = acc[u].xy, hi = acc[u].zw;
                lo = __builtin_elementwise_fma(pr, v4.xy + r4.xy, lo);
                hi = __builtin_elementwise_fma(pr, v4.zw + r4.zw, hi);
                acc[u] = (f32x4){lo.x, lo.y, hi.x, hi.y};
                asm volatile("" : "+v"(acc[u]) : : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[u][e];
                v += quad_x1(v);
                v += quad_x2(v);
                acc[u][e] = v;
            }
        }
        if (active) {            // lane sq stores quads sq, sq + 4, ...: the four lanes write 64 contiguous bytes
            float* orow = o + ((size_t)(b * T + t) * P + p) * C + h * F;
#pragma unroll
            for (int u = 0; u < NQ; ++u)
                if ((u & 3) == sq) st4(orow + 4 * u, acc[u]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward, "rows" part (per (pixel, query frame t): attention_bwd.hip has the definitions) in the same decomposition:
//   P = softmax(logits)  (recomputed exactly like the forward),  dP[s] = dO_t . (v_s + R_v[t][s]),
//   dS = P * (dP - sum_s P dP)  -> rows of P and dS to the workspaces (the cols / rpe kernels read them),
//   dq_t = scale * sum_s dS[s] * (k_s + R_k[t][s]).
// Everything it needs is what the forward stages: the R_k / R_q^T / R_v slices of the query group and the k / v rows of the
// strip.  Four lanes share a (pixel, query frame), lane sq owns keys sq, sq + 4, ...; the row sums and dq cross the quad
// with DPP moves.  (The first-generation rows kernel staged whole [T][T][16] slices in twelve barrier-separated phases:
// 58 us per launch at the cfg-C shapes, 17 % of the attention time of a training step.)
template <int F, int TCAP>
__global__ __launch_bounds__(256, 2)
void attn_temporal2_bwd_rows_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, const float* __restrict__ Rq,
                                    const float* __restrict__ Rk, const float* __restrict__ Rv, const float* __restrict__ mask,
                                    float* __restrict__ dqkv, float* __restrict__ Pg, float* __restrict__ dSg, T2Geom g) {
    using CF = T2Cfg<F, TCAP>;
    constexpr int NQ = CF::NQ, RS = CF::RS;
    extern __shared__ __attribute__((aligned(16))) float t2_smem[];
    const int T = g.T, P = g.P, C = g.C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NPX = g.NW * g.PPW;
    int b, h, tg, strip;
    t2_decode(g, b, h, tg, strip);
    if (strip >= g.strips) return;
    const int tq0 = tg * g.TGN;
    const int p0 = strip * NPX;
    float* Rimg = t2_smem;
    float* KVimg = t2_smem + (size_t)((g.TGN * 3 * RS + 63) & ~63) * 4;
    t2_stage<F, TCAP>(qkv, Rq, Rk, Rv, RSel{nullptr, 0}, g, b, h, tq0, p0, wave, lane, Rimg, KVimg);

    const float invN = 1.0f / (float)g.TGN;
    const int sq = lane & 3, pair = lane >> 2;
    const int j = (int)(((float)pair + 0.5f) * invN), tq = pair - j * g.TGN;
    const int jw = wave * g.PPW + j;
    const int t = tq0 + tq, p = p0 + jw;
    const bool active = j < g.PPW && p < P && t < T;
    const float scale = rsqrtf((float)F);
    const size_t ld = (size_t)3 * C;
    const size_t tok = active ? (size_t)(b * T + t) * P + p : 0;
    const float* qrow = qkv + tok * ld + h * F;
    const float* dorow = dO + tok * C + h * F;
    f32x4 q4[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) q4[u] = ld4(qrow + 4 * u) * scale;
    constexpr int KQ = (TCAP + 3) / 4;                      // keys per lane
    constexpr int ISTRIDE = 4 * NQ * 4;                     // floats between consecutive i
    int kb[NQ], rbs[NQ];
    {
        const int jr = active ? jw : 0, tr = active ? tq : 0;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int low = (NQ * sq + u) & 15, high = (NQ * sq + u) & ~15;
            kb[u] = (jr * 2 * RS + high + (low ^ (jr & 15))) * 4;
            rbs[u] = (tr * 3 * RS + high + (low ^ (tr & 15))) * 4;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces have landed (q too)
    __syncthreads();                                       // ... and everybody else's

    auto quad_x1 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); };
    auto quad_x2 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); };

    // ---- logits and softmax: the forward's arithmetic, operation for operation
    float pr[KQ];
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const f32x4 k4 = ld4(KVimg + kb[u] + i * ISTRIDE);
            const f32x4 r4 = ld4(Rimg + rbs[u] + i * ISTRIDE);
            const f32x4 g4 = ld4(Rimg + rbs[u] + RS * 4 + i * ISTRIDE);
            a0 = __builtin_elementwise_fma(q4[u].xy, k4.xy + r4.xy, a0);
            a0 = __builtin_elementwise_fma(q4[u].zw, k4.zw + r4.zw, a0);
            a1 = __builtin_elementwise_fma(k4.xy, g4.xy, a1);
            a1 = __builtin_elementwise_fma(k4.zw, g4.zw, a1);
        }
        pr[i] = (a0.x + a0.y) + (a1.x + a1.y) * scale;
        asm volatile("" : "+v"(pr[i]) : : "memory");       // keep every key's reads with its math (see the forward kernel)
        __builtin_amdgcn_sched_barrier(0);
    }
    {
        const int tm = active ? t : 0;
        const float mt = mask ? mask[b * T + tm] : 1.f;
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            const int sk = sq + 4 * i;
            float v = -INFINITY;
            if (sk < T) {
                v = pr[i];
                if (mask) {
                    const float ms = mask[b * T + sk];
                    const float pen = 1.f - (mt * ms + (1.f - mt) * (1.f - ms));
                    v -= (pen == 1.f) ? INFINITY : pen;
                }
            }
            pr[i] = v;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, quad_x1(mx));
        mx = fmaxf(mx, quad_x2(mx));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            const float e = (pr[i] == -INFINITY) ? 0.f : __expf(pr[i] - mx);
            pr[i] = e;
            sum += e;
        }
        sum += quad_x1(sum);
        sum += quad_x2(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int i = 0; i < KQ; ++i) pr[i] *= inv;
    }
    // ---- dP[s] = dO_t . (v_s + R_v[t][s]) for this lane's keys
    f32x4 d4[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) d4[u] = ld4(dorow + 4 * u);
    float dp[KQ];
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
        f32x2 a0 = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const f32x4 v4 = ld4(KVimg + kb[u] + RS * 4 + i * ISTRIDE);
            const f32x4 r4 = ld4(Rimg + rbs[u] + 2 * RS * 4 + i * ISTRIDE);
            a0 = __builtin_elementwise_fma(d4[u].xy, v4.xy + r4.xy, a0);
            a0 = __builtin_elementwise_fma(d4[u].zw, v4.zw + r4.zw, a0);
        }
        dp[i] = a0.x + a0.y;
        asm volatile("" : "+v"(dp[i]) : : "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- dS = P * (dP - sum_s P dP): the sum over the keys of the four lanes; rows of P and dS to the workspaces
    float dsum = 0.f;
#pragma unroll
    for (int i = 0; i < KQ; ++i) dsum += pr[i] * dp[i];    // (P is exactly 0 for key frames >= T)
    dsum += quad_x1(dsum);
    dsum += quad_x2(dsum);
    {
        const size_t wid = ((size_t)b * P + (active ? p : 0)) * g.heads + h;
        float* prow = Pg + (wid * T + (active ? t : 0)) * T;
        float* srow = dSg + (wid * T + (active ? t : 0)) * T;
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            dp[i] = pr[i] * (dp[i] - dsum);
            if (active && sq + 4 * i < T) {
                prow[sq + 4 * i] = pr[i];
                srow[sq + 4 * i] = dp[i];
            }
        }
    }
    // ---- dq_t = scale * sum_s dS[s] * (k_s + R_k[t][s]): partial sums over this lane's keys, then over the quad
    f32x4 acc[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KQ; ++i) {
        const f32x2 w = {dp[i], dp[i]};                     // exactly 0 for key frames >= T (their slots hold finite filler)
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const f32x4 k4 = ld4(KVimg + kb[u] + i * ISTRIDE);
            const f32x4 r4 = ld4(Rimg + rbs[u] + i * ISTRIDE);
            f32x2 lo = acc[u].xy, hi = acc[u].zw;
            lo = __builtin_elementwise_fma(w, k4.xy + r4.xy, lo);
            hi = __builtin_elementwise_fma(w, k4.zw + r4.zw, hi);
            acc[u] = (f32x4){lo.x, lo.y, hi.x, hi.y};
            asm volatile("" : "+v"(acc[u]) : : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[u][e];
            v += quad_x1(v);
            v += quad_x2(v);
            acc[u][e] = v * scale;
        }
    }
    if (active) {            // lane sq stores quads sq, sq + 4, ...: the four lanes write 64 contiguous bytes
        float* orow = dqkv + tok * ld + h * F;
#pragma unroll
        for (int u = 0; u < NQ; ++u)
            if ((u & 3) == sq) st4(orow + 4 * u, acc[u]);
    }
}

// LDS of a workgroup: the R image and the k / v image, each padded to whole DMA pieces of 64 slots (1 KiB)
inline size_t t2_lds_bytes(int TGN, int NPX, int RS) {
    return ((size_t)((TGN * 3 * RS + 63) & ~63) + (size_t)((NPX * 2 * RS + 63) & ~63)) * 16;
}

// Frame groups TG and waves per workgroup NW.  Measured on MI355X (T = 14 / 20, maps of 4 to 256 pixels, head dims 16 / 32):
// groups of <= 5 query frames (three pixels per wave, 15 of 16 lane quads busy, a small R image) with FOUR waves per
// workgroup were the fastest or within 5 % of it everywhere - four waves issue the image's DMA pieces in parallel and share
// one copy of the R slices, also where the map has fewer pixels than the workgroup has slots.  Fewer waves only when the
// images of four do not fit the LDS.  -> false: no decomposition fits.
inline bool t2_geometry(int B, int T, int P, int C, int heads, int F, int RS, T2Geom& g, size_t& lds, unsigned& grid) {
    g = T2Geom{};
    g.T = T; g.P = P; g.C = C; g.heads = heads; g.B = B;
    int best_tg = (T + 4) / 5, best_nw = 0;
    {
        const int TGN = (T + best_tg - 1) / best_tg, PPW = 16 / TGN;
        for (int NW = 4; NW >= 1 && best_nw == 0; NW /= 2)
            if (t2_lds_bytes(TGN, NW * PPW, RS) <= 160 * 1024) best_nw = NW;
        if (best_nw == 0) best_tg = 0;
    }
    if (best_tg == 0) return false;
    g.TG = best_tg; g.NW = best_nw;
    g.TGN = (T + g.TG - 1) / g.TG;
    g.PPW = 16 / g.TGN;
    const int NPX = g.NW * g.PPW;
    g.strips = (P + NPX - 1) / NPX;
    lds = t2_lds_bytes(g.TGN, NPX, RS);
    // XCD-aware placement (speed only): heads that share 128-byte lines and all frame groups of a strip on one XCD
    g.HL = F >= 32 ? 1 : 32 / F;
    if (g.HL > heads || heads % g.HL) g.HL = 1;
    g.NL = B * (heads / g.HL);
    g.XPG = (g.NL <= 8 && 8 % g.NL == 0) ? 8 / g.NL : 0;
    if (g.XPG > 0) {
        const int sblks = (g.strips + g.XPG - 1) / g.XPG;
        grid = 8u * (unsigned)(g.HL * g.TG) * (unsigned)sblks;
    } else {
        grid = (unsigned)(B * heads) * (unsigned)g.TG * (unsigned)g.strips;
    }
    return true;
}

template <int F, int TCAP>
int launch_t2(const float* qkv, const float* Rq, const float* Rk, const float* Rv, const float* mask, float* o, float* attn_out,
              int B, int T, int P, int C, int heads, RSel rsel, hipStream_t s) {
    T2Geom g;
    size_t lds;
    unsigned grid;
    if (!t2_geometry(B, T, P, C, heads, F, T2Cfg<F, TCAP>::RS, g, lds, grid)) return LFVDM_E_UNSUPPORTED;
    static DynLdsLimit limit;
    if (int rc = limit.ensure(reinterpret_cast<const void*>(&attn_temporal2_kernel<F, TCAP>), lds)) return rc;
    hipLaunchKernelGGL((attn_temporal2_kernel<F, TCAP>), dim3(grid), dim3(64 * g.NW), lds, s, qkv, Rq, Rk, Rv, mask, o, attn_out,
                       rsel, g);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

template <int F, int TCAP>
int launch_t2_bwd_rows(const float* qkv, const float* dO, const float* Rq, const float* Rk, const float* Rv, const float* mask,
                       float* dqkv, float* Pg, float* dSg, int B, int T, int P, int C, int heads, hipStream_t s) {
    T2Geom g;
    size_t lds;
    unsigned grid;
    if (!t2_geometry(B, T, P, C, heads, F, T2Cfg<F, TCAP>::RS, g, lds, grid)) return LFVDM_E_UNSUPPORTED;
    static DynLdsLimit limit;
    if (int rc = limit.ensure(reinterpret_cast<const void*>(&attn_temporal2_bwd_rows_kernel<F, TCAP>), lds)) return rc;
    hipLaunchKernelGGL((attn_temporal2_bwd_rows_kernel<F, TCAP>), dim3(grid), dim3(64 * g.NW), lds, s, qkv, dO, Rq, Rk, Rv, mask, dqkv,
                       Pg, dSg, g);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

template <int F>
int launch_t2_bwd_rows_f(const float* qkv, const float* dO, const float* Rq, const float* Rk, const float* Rv, const float* mask,
                         float* dqkv, float* Pg, float* dSg, int B, int T, int P, int C, int heads, hipStream_t s) {
    if (T <= 8) return launch_t2_bwd_rows<F, 8>(qkv, dO, Rq, Rk, Rv, mask, dqkv, Pg, dSg, B, T, P, C, heads, s);
    if (T <= 16) return launch_t2_bwd_rows<F, 16>(qkv, dO, Rq, Rk, Rv, mask, dqkv, Pg, dSg, B, T, P, C, heads, s);
    if (T <= 20) return launch_t2_bwd_rows<F, 20>(qkv, dO, Rq, Rk, Rv, mask, dqkv, Pg, dSg, B, T, P, C, heads, s);
    if (T <= 24) return launch_t2_bwd_rows<F, 24>(qkv, dO, Rq, Rk, Rv, mask, dqkv, Pg, dSg, B, T, P, C, heads, s);
    return launch_t2_bwd_rows<F, 32>(qkv, dO, Rq, Rk, Rv, mask, dqkv, Pg, dSg, B, T, P, C, heads, s);
}

template <int F>
int launch_t2_f(const float* qkv, const float* Rq, const float* Rk, const float* Rv, const float* mask, float* o,
                float* attn_out, int B, int T, int P, int C, int heads, RSel rsel, hipStream_t s) {
    if (T <= 8) return launch_t2<F, 8>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    if (T <= 16) return launch_t2<F, 16>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    if (T <= 20) return launch_t2<F, 20>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    if (T <= 24) return launch_t2<F, 24>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    return launch_t2<F, 32>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
}

}  // namespace

// Internal entry (attention.hip dispatches here first): LFVDM_E_UNSUPPORTED = shape not covered, use the first kernel.
int lfvdm_attn_temporal2_try(const float* qkv, const float* Rq, const float* Rk, const float* Rv, const float* mask, float* o,
                             float* attn_out, int B, int T, int P, int C, int heads, RSel rsel, hipStream_t s) {
    const int F = C / heads;
    // Large launches (e.g. 16x16 maps at 128 channels, batch 2: 25.9 us vs 30) keep every CU busy in the first kernel
    // too, which stages each R slice once per 12 pixels instead of once per 12 pixels AND frame group: use it there.
    if ((long)B * P * F >= 16384) return LFVDM_E_UNSUPPORTED;
    if (F == 16) return launch_t2_f<16>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    if (F == 32) return launch_t2_f<32>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    if (F == 64) return launch_t2_f<64>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    return LFVDM_E_UNSUPPORTED;
}

// Internal entry (attention_bwd.hip tries it first): the "rows" kernel of the temporal-attention backward in the
// second-generation decomposition; LFVDM_E_UNSUPPORTED = shape not covered, use the first-generation rows kernel.
int lfvdm_attn_temporal2_bwd_rows_try(const float* qkv, const float* dO, const float* Rq, const float* Rk, const float* Rv,
                                      const float* mask, float* dqkv, float* Pg, float* dSg, int B, int T, int P, int C, int heads,
                                      hipStream_t s) {
    const int F = C / heads;
    if (T > 32) return LFVDM_E_UNSUPPORTED;
    if (F == 16) return launch_t2_bwd_rows_f<16>(qkv, dO, Rq, Rk, Rv, mask, dqkv, Pg, dSg, B, T, P, C, heads, s);
    if (F == 32) return launch_t2_bwd_rows_f<32>(qkv, dO, Rq, Rk, Rv, mask, dqkv, Pg, dSg, B, T, P, C, heads, s);
    if (F == 64) return launch_t2_bwd_rows_f<64>(qkv, dO, Rq, Rk, Rv, mask, dqkv, Pg, dSg, B, T, P, C, heads, s);
    return LFVDM_E_UNSUPPORTED;
}
