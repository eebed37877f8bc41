// Temporal attention with relative position terms (reference rpe.py:143-169), second generation.
//
//   logits[t][s] = q_t . (k_s + R_k[t][s]) + scale * k_s . R_q[s][t]     (q already scaled)
//   o[t]         = sum_s softmax_s(logits + two-clique mask)[s] * (v_s + R_v[t][s])
//
// per (batch b, head h, pixel p); T <= 32 frames, head dim F in {8, 16, 32}.  Same per-lane arithmetic as the first
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

template <int F, int TCAP>
__global__ __launch_bounds__(256)
void attn_temporal2_kernel(const float* __restrict__ qkv, const float* __restrict__ Rq, const float* __restrict__ Rk,
                           const float* __restrict__ Rv, const float* __restrict__ mask, float* __restrict__ o,
                           float* __restrict__ attn_out, const int64_t* __restrict__ rsel, T2Geom g) {
    using CF = T2Cfg<F, TCAP>;
    constexpr int NQ = CF::NQ, RS = CF::RS;
    extern __shared__ __attribute__((aligned(16))) float t2_smem[];
    const int T = g.T, P = g.P, C = g.C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NPX = g.NW * g.PPW;
    // ---- which (b, head, frame group, strip): XCD-aware decode of the flat block index
    int b, h, tg, strip;
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
    if (strip >= g.strips) return;                             // (grid padding of the XCD-aware map; workgroup-uniform)
    const int tq0 = tg * g.TGN;
    const int p0 = strip * NPX;
    // LDS images, in slots of 4 floats: R[tq][3][RS] (R_k | R_q transposed | R_v), then KV[pixel][2][RS]
    float* Rimg = t2_smem;
    float* KVimg = t2_smem + (size_t)g.TGN * 3 * RS * 4;
    const float invRS = 1.0f / (float)RS;

    // ---- LDS-DMA staging (global_load_lds_dwordx4: 16 bytes per lane from a per-lane address to the wave's next 1 KiB
    // of LDS): pieces of 64 slots, dealt round-robin to the waves.  Slots nobody reads (row padding, frames past T,
    // pixels past P) are filled from a valid dummy address.
    {
        const size_t rb = rsel ? (size_t)rsel[b] * g.B + b : (size_t)b;
        const float* Rsrc[3] = {Rk + rb * T * T * C + h * F, Rq + rb * T * T * C + h * F, Rv + rb * T * T * C + h * F};
        const float* kvsrc = qkv + (size_t)b * T * P * 3 * C + C + h * F;     // k of (b, frame 0, pixel 0); v is C further
        const int nR = (g.TGN * 3 * RS + 63) >> 6, nKV = (NPX * 2 * RS + 63) >> 6;
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

    // ---- this lane's (pixel, query frame); q straight to registers while the DMA is in flight
    const float invN = 1.0f / (float)g.TGN;
    const int j = (int)(((float)lane + 0.5f) * invN), tq = lane - j * g.TGN;
    const int jw = wave * g.PPW + j;
    const int t = tq0 + tq, p = p0 + jw;
    const bool active = j < g.PPW && p < P && t < T;
    const float scale = rsqrtf((float)F);
    const size_t ld = (size_t)3 * C;
    const float* qrow = active ? qkv + ((size_t)(b * T + t) * P + p) * ld + h * F : qkv;
    f32x4 q4[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) q4[u] = ld4(qrow + 4 * u) * scale;
    // swizzled slot addresses (floats) of the 16 low slot patterns: reads are base[cl] + immediate
    int kb[16], rbs[16];
    {
        const int jr = active ? jw : 0, tr = active ? tq : 0;
#pragma unroll
        for (int cl = 0; cl < 16; ++cl) {
            kb[cl] = (jr * 2 * RS + (cl ^ (jr & 15))) * 4;
            rbs[cl] = (tr * 3 * RS + (cl ^ (tr & 15))) * 4;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces have landed (q too)
    __syncthreads();                                       // ... and everybody else's

    float logit[TCAP];
    if (active) {
#pragma unroll
        for (int s = 0; s < TCAP; ++s) {
            float a0 = 0.f, a1 = 0.f;
            if (s < T) {
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const int c = s * NQ + u, cl = c & 15, ch = (c >> 4) * 64;
                    const f32x4 k4 = ld4(KVimg + kb[cl] + ch);
                    const f32x4 rk4 = ld4(Rimg + rbs[cl] + ch);
                    const f32x4 rq4 = ld4(Rimg + rbs[cl] + RS * 4 + ch);
                    a0 += q4[u].x * (k4.x + rk4.x) + q4[u].y * (k4.y + rk4.y) + q4[u].z * (k4.z + rk4.z) + q4[u].w * (k4.w + rk4.w);
                    a1 += k4.x * rq4.x + k4.y * rq4.y + k4.z * rq4.z + k4.w * rq4.w;
                }
            }
            logit[s] = a0 + a1 * scale;
        }
        // two-clique mask + softmax, all in this lane's registers (rpe.py:156-163)
        const float mt = mask ? mask[b * T + t] : 1.f;
        float mx = -INFINITY;
#pragma unroll
        for (int s = 0; s < TCAP; ++s) {
            float v = -INFINITY;
            if (s < T) {
                v = logit[s];
                if (mask) {
                    const float ms = mask[b * T + s];
                    const float pen = 1.f - (mt * ms + (1.f - mt) * (1.f - ms));
                    v -= (pen == 1.f) ? INFINITY : pen;
                }
            }
            logit[s] = v;
            mx = fmaxf(mx, v);
        }
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < TCAP; ++s) {
            const float e = (logit[s] == -INFINITY) ? 0.f : __expf(logit[s] - mx);
            logit[s] = e;
            sum += e;
        }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int s = 0; s < TCAP; ++s) logit[s] *= inv;
        if (attn_out) {
            float* ar = attn_out + ((((size_t)b * P + p) * g.heads + h) * T + t) * T;
#pragma unroll
            for (int s = 0; s < TCAP; ++s)
                if (s < T) ar[s] = logit[s];
        }
        // o[t][f] = sum_s p[s] * (v[s][f] + R_v[t][s][f])
        f32x4 acc[NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < TCAP; ++s) {
            if (s < T) {
                float pr = logit[s];
                asm volatile("" : "+v"(pr));       // keeps the broadcast operand out of hoisted packed-FMA pairs (spills)
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const int c = s * NQ + u, cl = c & 15, ch = (c >> 4) * 64;
                    acc[u] += pr * (ld4(KVimg + kb[cl] + RS * 4 + ch) + ld4(Rimg + rbs[cl] + 2 * RS * 4 + ch));
                }
            }
        }
        float* orow = o + ((size_t)(b * T + t) * P + p) * C + h * F;
#pragma unroll
        for (int u = 0; u < NQ; ++u) st4(orow + 4 * u, acc[u]);
    }
}

template <int F, int TCAP>
int launch_t2(const float* qkv, const float* Rq, const float* Rk, const float* Rv, const float* mask, float* o, float* attn_out,
              int B, int T, int P, int C, int heads, const int64_t* rsel, hipStream_t s) {
    using CF = T2Cfg<F, TCAP>;
    constexpr int RS = CF::RS;
    // Frame groups TG and waves per workgroup NW by a small cost model (us): the kernel is bound by the CU's LDS read
    // path (5 b128 reads per (s, 4 channels) per lane: `cw` per wave) and by the DMA fill of the workgroup's images
    // (~70 GB/s per CU from L2), so what counts is the number of waves and of staged bytes on the BUSIEST CU.
    T2Geom g{};
    g.T = T; g.P = P; g.C = C; g.heads = heads; g.B = B;
    int best_tg = 0, best_nw = 0;
    double best = 1e30;
    const double cw = (double)T * CF::NQ * 5 * 4 / 2400.0;
    for (int TG = 1; TG <= 5; ++TG) {
        const int TGN = (T + TG - 1) / TG;
        if (TG > 1 && TGN * (TG - 1) >= T) continue;              // an empty last group
        const int PPW = 64 / TGN;
        if (PPW < 1) continue;
        for (int NW = 1; NW <= 4; NW *= 2) {
            const int NPX = NW * PPW;
            if (NW > 1 && (NW / 2) * PPW >= P) continue;          // waves without pixels
            const size_t lds = ((size_t)TGN * 3 + (size_t)NPX * 2) * RS * 16;
            if (lds > 160 * 1024) continue;
            const long strips = (P + NPX - 1) / NPX;
            const long wgs = strips * TG * B * heads;
            int per_cu = (int)(160 * 1024 / lds);
            per_cu = per_cu > 8 ? 8 : per_cu;
            if (per_cu * NW > 8) per_cu = 8 / NW;
            const double fill = (double)lds / 70e3;
            double t;
            if (wgs <= 256L * per_cu) {
                const long maxr = (wgs + 255) / 256;
                t = maxr * NW * cw + fill * (1.0 + 0.5 * (maxr - 1));
            } else {
                const long rounds = (wgs + 256L * per_cu - 1) / (256L * per_cu);
                t = rounds * (per_cu * NW * cw + fill * (1.0 + 0.5 * (per_cu - 1)));
            }
            if (t < best) { best = t; best_tg = TG; best_nw = NW; }
        }
    }
    if (best_tg == 0) return LFVDM_E_UNSUPPORTED;
    g.TG = best_tg; g.NW = best_nw;
    g.TGN = (T + g.TG - 1) / g.TG;
    g.PPW = 64 / g.TGN;
    const int NPX = g.NW * g.PPW;
    g.strips = (P + NPX - 1) / NPX;
    const size_t lds = ((size_t)g.TGN * 3 + (size_t)NPX * 2) * RS * 16;
    // XCD-aware placement (speed only): heads that share 128-byte lines and all frame groups of a strip on one XCD
    g.HL = F >= 32 ? 1 : 32 / F;
    if (g.HL > heads || heads % g.HL) g.HL = 1;
    g.NL = B * (heads / g.HL);
    g.XPG = (g.NL <= 8 && 8 % g.NL == 0) ? 8 / g.NL : 0;
    unsigned grid;
    if (g.XPG > 0) {
        const int sblks = (g.strips + g.XPG - 1) / g.XPG;
        grid = 8u * (unsigned)(g.HL * g.TG) * (unsigned)sblks;
    } else {
        grid = (unsigned)(B * heads) * (unsigned)g.TG * (unsigned)g.strips;
    }
    static DynLdsLimit limit;
    if (int rc = limit.ensure(reinterpret_cast<const void*>(&attn_temporal2_kernel<F, TCAP>), lds)) return rc;
    hipLaunchKernelGGL((attn_temporal2_kernel<F, TCAP>), dim3(grid), dim3(64 * g.NW), lds, s, qkv, Rq, Rk, Rv, mask, o, attn_out,
                       rsel, g);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

template <int F>
int launch_t2_f(const float* qkv, const float* Rq, const float* Rk, const float* Rv, const float* mask, float* o,
                float* attn_out, int B, int T, int P, int C, int heads, const int64_t* rsel, hipStream_t s) {
    if (T <= 8) return launch_t2<F, 8>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    if (T <= 16) return launch_t2<F, 16>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    if (T <= 20) return launch_t2<F, 20>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    if (T <= 24) return launch_t2<F, 24>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    return launch_t2<F, 32>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
}

}  // namespace

// Internal entry (attention.hip dispatches here first): LFVDM_E_UNSUPPORTED = shape not covered, use the first kernel.
int lfvdm_attn_temporal2_try(const float* qkv, const float* Rq, const float* Rk, const float* Rv, const float* mask, float* o,
                             float* attn_out, int B, int T, int P, int C, int heads, const int64_t* rsel, hipStream_t s) {
    const int F = C / heads;
    if (F == 16) return launch_t2_f<16>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    if (F == 32) return launch_t2_f<32>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    if (F == 8) return launch_t2_f<8>(qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads, rsel, s);
    return LFVDM_E_UNSUPPORTED;
}
