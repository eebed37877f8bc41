// Backward of the temporal RPE attention core (autograd of the reference's rpe.py:143-169, temporal instance).
//
//   logits[t][s] = scale * (q[t].k[s] + q[t].R_k[t][s] + k[s].R_q[s][t])  (+ two-clique mask)
//   P = softmax_s(logits),   o[t] = sum_s P[t][s] * (v[s] + R_v[t][s])
//
// Three kernels, no atomics (bitwise reproducible):
//   rows : lane = (pixel, query frame t).  Recomputes the logits exactly like the forward kernel, then
//          dP[t][s] = dO[t].(v[s] + R_v[t][s]), dS = P * (dP - sum_s P dP), writes the rows of P and dS to a
//          workspace and dq[t] = scale * sum_s dS[t][s] * (k[s] + R_k[t][s]).
//   cols : lane = (pixel, key frame s).  dk[s] = scale * sum_t dS[t][s] * (q[t] + R_q[s][t]),
//          dv[s] = sum_t P[t][s] * dO[t]   (columns of the workspace matrices).
//   rpe  : one workgroup per (b, head, frame i): the three R gradients are small GEMMs over the PIXELS,
//          dR_k[i][s][f] = scale * sum_p dS_p[i][s] q_p[i][f],  dR_v[i][s][f] = sum_p P_p[i][s] dO_p[i][f],
//          dR_q[i][t][f] = scale * sum_p dS_p[t][i] k_p[i][f]   on fp32 MFMA 16x16x4 (K dimension = pixels).
// Layouts as in the forward: qkv / dqkv rows (b, t, p) with channels [3][heads][F]; dO rows (b, t, p) x C;
// R tensors [B][T][T][C]; workspace matrices [(b*P + p)*heads + h][T][T].
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "common_hip.h"
#include "lfvdm_hip.h"

// (attention_temporal2.hip) the rows kernel in the second-generation decomposition; LFVDM_E_UNSUPPORTED = not covered
int lfvdm_attn_temporal2_bwd_rows_try(const float* qkv, const float* dO, const float* Rq, const float* Rk, const float* Rv,
                                      const float* mask, float* dqkv, float* Pg, float* dSg, int B, int T, int P, int C, int heads,
                                      hipStream_t s);

namespace {

constexpr int TB_MAXT = 32;

template <int TMAX, int FC>
struct TBStage {
    static constexpr int NQ = FC / 4;
    static constexpr int RB = (TMAX * TMAX * NQ + 255) / 256;   // R float4 per thread per slice
    f32x4 ra[RB], rb[RB], kv[NQ], kv2[NQ], q[NQ];
};

// ------------------------------------------------------------------------------------------------ rows
template <int TMAX, int FC>
__global__ __launch_bounds__(256)
void attn_temporal_bwd_rows_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, const float* __restrict__ Rq,
                                   const float* __restrict__ Rk, const float* __restrict__ Rv, const float* __restrict__ mask,
                                   float* __restrict__ dqkv, float* __restrict__ Pg, float* __restrict__ dSg, int T, int P, int C,
                                   int heads, int PPW) {
    using ST = TBStage<TMAX, FC>;
    constexpr int NQ = ST::NQ, RB = ST::RB;
    extern __shared__ __attribute__((aligned(16))) float tb_smem[];
    const int RST = T * FC + 4;
    float* Ra_s = tb_smem;                      // [T][RST]  R_k / R_v / R_k slice, row t
    float* Rb_s = Ra_s + T * RST;               // [T][RST]  R_q transposed: row t holds R_q[s][t]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* kv_s = Rb_s + T * RST + wave * PPW * RST;   // [PPW][RST] wave-private k / v rows
    const int b = blockIdx.z, h = blockIdx.y;
    const int F = C / heads;
    const int NC = F / FC;
    const float invT = 1.0f / (float)T;
    const int j = (int)(((float)lane + 0.5f) * invT), t = lane - j * T;
    const int p0 = (blockIdx.x * 4 + wave) * PPW;
    const int p = p0 + j;
    const bool active = j < PPW && p < P;
    const float scale = rsqrtf((float)F);
    const size_t ld = (size_t)3 * C;
    const size_t tok = (size_t)(b * T + t) * P + p;        // dereferenced only if active
    const float* qrow = qkv + tok * ld + h * F;
    const float* dorow = dO + tok * C + h * F;
    const size_t rofs = (size_t)b * T * T * C + h * F;
    const float* Rbase[3] = {Rk + rofs, Rq + rofs, Rv + rofs};

    int r_g[RB], r_la[RB], r_lb[RB];
    unsigned r_ok = 0, k_ok = 0, k_keep = 0;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int e = tid + 256 * i;
        const int u = e % NQ, ts = e / NQ;
        const int a = (int)(((float)ts + 0.5f) * invT), c = ts - a * T;
        const bool ok = e < T * T * NQ;
        r_ok |= ok ? (1u << i) : 0u;
        r_g[i] = ok ? ts * C + 4 * u : 0;
        r_la[i] = a * RST + c * FC + 4 * u;
        r_lb[i] = c * RST + a * FC + 4 * u;
    }
    int k_g[NQ], k_l[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int e = lane + 64 * i;
        const int u = e % NQ, js = e / NQ;
        const int jj = (int)(((float)js + 0.5f) * invT), ss = js - jj * T;
        const bool ok = e < PPW * T * NQ, inside = ok && p0 + jj < P;
        k_ok |= ok ? (1u << i) : 0u;
        k_keep |= inside ? (1u << i) : 0u;
        k_g[i] = inside ? (int)(((size_t)(b * T + ss) * P + p0 + jj) * ld) + h * F + 4 * u : 0;
        k_l[i] = jj * RST + ss * FC + 4 * u;
    }
    const float* qsafe = active ? qrow : qkv;
    const float* dosafe = active ? dorow : dO;

    // phase ph: kind = ph / NC (0 logits, 1 dP, 2 dq), chunk = ph % NC
    ST st;
    auto issue = [&](int ph) {
        const int kind = ph / NC, f0 = (ph - kind * NC) * FC;
        const float* A = (kind == 1 ? Rbase[2] : Rbase[0]) + f0;
        const float* Bq = Rbase[1] + (kind == 0 ? f0 : 0);
        const float* KV = qkv + (kind == 1 ? 2 * C : C) + f0;
        const float* QR = (kind == 1 ? dosafe : qsafe) + (kind == 2 ? 0 : f0);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            st.ra[i] = ld4(A + r_g[i]);
            st.rb[i] = ld4(Bq + r_g[i]);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) st.kv[i] = ld4(KV + k_g[i]);
#pragma unroll
        for (int u = 0; u < NQ; ++u) st.q[u] = ld4(QR + 4 * u);
    };
    auto commit = [&](int ph) {
        const bool lg = ph < NC;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            if (r_ok & (1u << i)) {
                st4(Ra_s + r_la[i], st.ra[i]);
                if (lg) st4(Rb_s + r_lb[i], st.rb[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            if (k_ok & (1u << i)) st4(kv_s + k_l[i], (k_keep & (1u << i)) ? st.kv[i] : (f32x4){0.f, 0.f, 0.f, 0.f});
    };

    float pr[TMAX], dp[TMAX];
#pragma unroll
    for (int s = 0; s < TMAX; ++s) {
        pr[s] = 0.f;
        dp[s] = 0.f;
    }

    issue(0);
    // ---- logits (identical arithmetic to the forward kernel) and dP
    for (int ph = 0; ph < 2 * NC; ++ph) {
        __syncthreads();
        commit(ph);
        const bool lg = ph < NC;
        f32x4 q4[NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) q4[u] = lg ? st.q[u] * scale : st.q[u];
        __syncthreads();
        issue(ph + 1);
        if (active) {
            const float* kr = kv_s + j * RST;
            const float* rar = Ra_s + t * RST;
            const float* rbr = Rb_s + t * RST;
            if (lg) {
#pragma unroll
                for (int s = 0; s < TMAX; ++s) {
                    if (s < T) {
                        float a0 = 0.f, a1 = 0.f;
#pragma unroll
                        for (int u = 0; u < NQ; ++u) {
                            const f32x4 k4 = ld4(kr + s * FC + 4 * u);
                            const f32x4 rk4 = ld4(rar + s * FC + 4 * u);
                            const f32x4 rq4 = ld4(rbr + s * FC + 4 * u);
                            a0 += q4[u].x * (k4.x + rk4.x) + q4[u].y * (k4.y + rk4.y) + q4[u].z * (k4.z + rk4.z) + q4[u].w * (k4.w + rk4.w);
                            a1 += k4.x * rq4.x + k4.y * rq4.y + k4.z * rq4.z + k4.w * rq4.w;
                        }
                        pr[s] += a0 + a1 * scale;
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < TMAX; ++s) {
                    if (s < T) {
                        float a0 = 0.f;
#pragma unroll
                        for (int u = 0; u < NQ; ++u) {
                            const f32x4 v4 = ld4(kr + s * FC + 4 * u);
                            const f32x4 rv4 = ld4(rar + s * FC + 4 * u);
                            a0 += q4[u].x * (v4.x + rv4.x) + q4[u].y * (v4.y + rv4.y) + q4[u].z * (v4.z + rv4.z) + q4[u].w * (v4.w + rv4.w);
                        }
                        dp[s] += a0;
                    }
                }
            }
        }
        if (ph == NC - 1 && active) {   // two-clique mask + softmax in registers (as in the forward)
            const float mt = mask ? mask[b * T + t] : 1.f;
            float mx = -INFINITY;
#pragma unroll
            for (int s = 0; s < TMAX; ++s) {
                float v = -INFINITY;
                if (s < T) {
                    v = pr[s];
                    if (mask) {
                        const float ms = mask[b * T + s];
                        const float pen = 1.f - (mt * ms + (1.f - mt) * (1.f - ms));
                        v -= (pen == 1.f) ? INFINITY : pen;
                    }
                }
                pr[s] = v;
                mx = fmaxf(mx, v);
            }
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < TMAX; ++s) {
                const float e = (pr[s] == -INFINITY) ? 0.f : __expf(pr[s] - mx);
                pr[s] = e;
                sum += e;
            }
            const float inv = 1.0f / sum;
#pragma unroll
            for (int s = 0; s < TMAX; ++s) pr[s] *= inv;
        }
    }
    // ---- dS = P * (dP - sum_s P dP); rows of P and dS to the workspace
    if (active) {
        float dsum = 0.f;
#pragma unroll
        for (int s = 0; s < TMAX; ++s) dsum += pr[s] * dp[s];
        const size_t wid = ((size_t)b * P + p) * heads + h;
        float* prow = Pg + (wid * T + t) * T;
        float* srow = dSg + (wid * T + t) * T;
#pragma unroll
        for (int s = 0; s < TMAX; ++s) {
            dp[s] = pr[s] * (dp[s] - dsum);
            if (s < T) {
                prow[s] = pr[s];
                srow[s] = dp[s];
            }
        }
    }
    // ---- dq[t][f] = scale * sum_s dS[t][s] * (k[s][f] + R_k[t][s][f])
    for (int ph = 2 * NC; ph < 3 * NC; ++ph) {
        __syncthreads();
        commit(ph);
        __syncthreads();
        if (ph + 1 < 3 * NC) issue(ph + 1);
        if (active) {
            const float* kr = kv_s + j * RST;
            const float* rkr = Ra_s + t * RST;
            f32x4 acc[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < TMAX; ++s) {
                if (s < T) {
                    float w = dp[s];
                    asm volatile("" : "+v"(w));   // see attention.hip: no hoisted broadcast pairs
#pragma unroll
                    for (int u = 0; u < NQ; ++u) acc[u] += w * (ld4(kr + s * FC + 4 * u) + ld4(rkr + s * FC + 4 * u));
                }
            }
            float* orow = dqkv + tok * ld + h * F + (ph - 2 * NC) * FC;
#pragma unroll
            for (int u = 0; u < NQ; ++u) st4(orow + 4 * u, acc[u] * scale);
        }
    }
}

// ------------------------------------------------------------------------------------------------ cols
template <int TMAX, int FC>
__global__ __launch_bounds__(256)
void attn_temporal_bwd_cols_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, const float* __restrict__ Rq,
                                   const float* __restrict__ Pg, const float* __restrict__ dSg, float* __restrict__ dqkv, int T,
                                   int P, int C, int heads, int PPW) {
    using ST = TBStage<TMAX, FC>;
    constexpr int NQ = ST::NQ, RB = ST::RB;
    extern __shared__ __attribute__((aligned(16))) float tb_smem[];
    const int RST = T * FC + 4;
    float* Rq_s = tb_smem;                                   // [T][RST]  row s: R_q[s][t] for all t
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* q_s = Rq_s + T * RST + wave * 2 * PPW * RST;      // [PPW][RST] q rows, then [PPW][RST] dO rows
    float* do_s = q_s + PPW * RST;
    const int b = blockIdx.z, h = blockIdx.y;
    const int F = C / heads;
    const int NC = F / FC;
    const float invT = 1.0f / (float)T;
    const int j = (int)(((float)lane + 0.5f) * invT), s = lane - j * T;
    const int p0 = (blockIdx.x * 4 + wave) * PPW;
    const int p = p0 + j;
    const bool active = j < PPW && p < P;
    const float scale = rsqrtf((float)F);
    const size_t ld = (size_t)3 * C;
    const float* Rbase = Rq + (size_t)b * T * T * C + h * F;

    int r_g[RB], r_la[RB];
    unsigned r_ok = 0, k_ok = 0, k_keep = 0;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int e = tid + 256 * i;
        const int u = e % NQ, ts = e / NQ;
        const int a = (int)(((float)ts + 0.5f) * invT), c = ts - a * T;
        const bool ok = e < T * T * NQ;
        r_ok |= ok ? (1u << i) : 0u;
        r_g[i] = ok ? ts * C + 4 * u : 0;
        r_la[i] = a * RST + c * FC + 4 * u;
    }
    int k_g[NQ], d_g[NQ], k_l[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int e = lane + 64 * i;
        const int u = e % NQ, js = e / NQ;
        const int jj = (int)(((float)js + 0.5f) * invT), tt = js - jj * T;
        const bool ok = e < PPW * T * NQ, inside = ok && p0 + jj < P;
        k_ok |= ok ? (1u << i) : 0u;
        k_keep |= inside ? (1u << i) : 0u;
        const size_t tk = (size_t)(b * T + tt) * P + p0 + jj;
        k_g[i] = inside ? (int)(tk * ld) + h * F + 4 * u : 0;
        d_g[i] = inside ? (int)(tk * C) + h * F + 4 * u : 0;
        k_l[i] = jj * RST + tt * FC + 4 * u;
    }

    // columns s of this pixel's P and dS matrices
    float pc[TMAX], dc[TMAX];
    {
        const size_t wid = active ? ((size_t)b * P + p) * heads + h : 0;
        const float* pcol = Pg + wid * T * T + (active ? s : 0);
        const float* scol = dSg + wid * T * T + (active ? s : 0);
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            pc[t] = (active && t < T) ? pcol[t * T] : 0.f;
            dc[t] = (active && t < T) ? scol[t * T] : 0.f;
        }
    }

    ST st;
    auto issue = [&](int ch) {
        const int f0 = ch * FC;
#pragma unroll
        for (int i = 0; i < RB; ++i) st.ra[i] = ld4(Rbase + f0 + r_g[i]);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            st.kv[i] = ld4(qkv + f0 + k_g[i]);
            st.kv2[i] = ld4(dO + f0 + d_g[i]);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < RB; ++i)
            if (r_ok & (1u << i)) st4(Rq_s + r_la[i], st.ra[i]);
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            if (k_ok & (1u << i)) {
                const bool keep = k_keep & (1u << i);
                st4(q_s + k_l[i], keep ? st.kv[i] : (f32x4){0.f, 0.f, 0.f, 0.f});
                st4(do_s + k_l[i], keep ? st.kv2[i] : (f32x4){0.f, 0.f, 0.f, 0.f});
            }
    };

    issue(0);
    for (int ch = 0; ch < NC; ++ch) {
        __syncthreads();
        commit();
        __syncthreads();
        if (ch + 1 < NC) issue(ch + 1);
        if (active) {
            const float* qr = q_s + j * RST;
            const float* dr = do_s + j * RST;
            const float* rq = Rq_s + s * RST;
            f32x4 accK[NQ], accV[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                accK[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                accV[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int t = 0; t < TMAX; ++t) {
                if (t < T) {
                    float wk = dc[t], wv = pc[t];
                    asm volatile("" : "+v"(wk), "+v"(wv));
#pragma unroll
                    for (int u = 0; u < NQ; ++u) {
                        accK[u] += wk * (ld4(qr + t * FC + 4 * u) + ld4(rq + t * FC + 4 * u));
                        accV[u] += wv * ld4(dr + t * FC + 4 * u);
                    }
                }
            }
            float* orow = dqkv + ((size_t)(b * T + s) * P + p) * ld + h * F + ch * FC;
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                st4(orow + C + 4 * u, accK[u] * scale);
                st4(orow + 2 * C + 4 * u, accV[u]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ rpe
// grid (T, heads, B); wave w owns the output tiles (term, f-tile) = w, w + 4, ...; MFMA 16x16x4 with the
// k index running over pixels.
__global__ __launch_bounds__(256)
void attn_temporal_bwd_rpe_kernel(const float* __restrict__ qkv, const float* __restrict__ dO, const float* __restrict__ Pg,
                                  const float* __restrict__ dSg, float* __restrict__ dRq, float* __restrict__ dRk,
                                  float* __restrict__ dRv, int T, int P, int C, int heads) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int F = C / heads;
    const int FT = (F + 15) / 16;
    const int lq = lane & 15, kk = lane >> 4;
    const float scale = rsqrtf((float)F);
    const size_t ld = (size_t)3 * C;
    const size_t TT = (size_t)T * T;
    for (int tile = wave; tile < 3 * FT; tile += 4) {
        const int term = tile / FT, ft = tile - term * FT;
        const int col = 16 * ft + lq;
        const bool colok = col < F;
        // B operand source: q (term 0), dO (term 1), k (term 2) of frame i, channel col
        const float* Bsrc = (term == 1) ? dO + (size_t)(b * T + i) * P * C + h * F + col
                                        : qkv + (size_t)(b * T + i) * P * ld + (term == 2 ? C : 0) + h * F + col;
        const size_t bstride = (term == 1) ? (size_t)C : ld;
        // A operand source: dS row i (term 0), P row i (term 1), dS column i (term 2), element = row index
        const float* Asrc = (term == 1 ? Pg : dSg) + ((size_t)b * P * heads + h) * TT;
        const size_t astride = (size_t)heads * TT;     // per pixel
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const int row0 = lq, row1 = 16 + lq;
        const size_t a0 = (term == 2) ? (size_t)row0 * T + i : (size_t)i * T + row0;
        const size_t a1 = (term == 2) ? (size_t)row1 * T + i : (size_t)i * T + row1;
        const bool two = T > 16;
        // 8 k-steps (32 pixels) per iteration: all 24 loads are issued before the first MFMA needs one (the loop is a
        // chain of dependent-latency loads otherwise)
        for (int pb = 0; pb < P; pb += 32) {
            float bv[8], av0[8], av1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p = pb + 4 * u + kk;
                const bool pok = p < P;
                bv[u] = (pok && colok) ? Bsrc[(size_t)p * bstride] : 0.f;
                av0[u] = (pok && row0 < T) ? Asrc[(size_t)p * astride + a0] : 0.f;
                av1[u] = (two && pok && row1 < T) ? Asrc[(size_t)p * astride + a1] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[u], bv[u], acc[0], 0, 0, 0);
                if (two) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[u], bv[u], acc[1], 0, 0, 0);
            }
        }
        float* out = (term == 0 ? dRk : term == 1 ? dRv : dRq) + ((size_t)(b * T + i) * T) * C + h * F + col;
        const float mul = (term == 1) ? 1.f : scale;
        if (colok) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * rt + 4 * kk + r;
                    if (row < T) out[(size_t)row * C] = acc[rt][r] * mul;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------ RPE net
// Hidden layer of an RPENet in the training path (rpe.py:20-31 up to the output layer):
//   act[r][c] = silu(tproj[b][c] + Wd[c][:] . feats[r][:] + bd[c]),  r = (b, t, s), feats = 3 distance features.
__global__ __launch_bounds__(256) void rpe_front_fwd_kernel(const float* __restrict__ tproj, int tld, const float* __restrict__ feats,
                                                            const float* __restrict__ Wd, const float* __restrict__ bd,
                                                            float* __restrict__ act, long rows, int rows_per_b, int C) {
    const int Q = C / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * Q) return;
    const long r = i / Q;
    const int c = (int)(i - r * Q) * 4;
    const int b = (int)(r / rows_per_b);
    const float f0 = feats[r * 3 + 0], f1 = feats[r * 3 + 1], f2 = feats[r * 3 + 2];
    const f32x4 tp = ld4(tproj + (size_t)b * tld + c), bb = ld4(bd + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* w = Wd + (size_t)(c + k) * 3;
        o[k] = silu_f(tp[k] + bb[k] + w[0] * f0 + w[1] * f1 + w[2] * f2);
    }
    st4(act + r * C + c, o);
}

// Backward: dhid = d_act * silu'(hid) (hid recomputed); dtproj[b][c] += sum_rows dhid, dWd[c][j] += sum dhid*feats[j],
// dbd[c] += sum dhid.  grid (C/64, B, row chunks), block = 4 row lanes x 64 channels; float atomics for the sums.
__global__ __launch_bounds__(256) void rpe_front_bwd_kernel(const float* __restrict__ tproj, int tld, const float* __restrict__ feats,
                                                            const float* __restrict__ Wd, const float* __restrict__ bd,
                                                            const float* __restrict__ d_act, float* __restrict__ dtproj, int dtld,
                                                            float* __restrict__ dWd, float* __restrict__ dbd, int rows_per_b,
                                                            int C, int chunk) {
    __shared__ float red[4][64][4];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int b = blockIdx.y;
    const int r0 = blockIdx.z * chunk;
    const int r1 = min(r0 + chunk, rows_per_b);
    float s = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (c < C) {
        const float tp = tproj[(size_t)b * tld + c] + bd[c];
        const float w0 = Wd[(size_t)c * 3], w1 = Wd[(size_t)c * 3 + 1], w2 = Wd[(size_t)c * 3 + 2];
        for (int rr = r0 + rl; rr < r1; rr += 4) {
            const size_t r = (size_t)b * rows_per_b + rr;
            const float f0 = feats[r * 3], f1 = feats[r * 3 + 1], f2 = feats[r * 3 + 2];
            const float h = tp + w0 * f0 + w1 * f1 + w2 * f2;
            const float sg = 1.0f / (1.0f + __expf(-h));
            const float d = d_act[r * C + c] * sg * (1.0f + h * (1.0f - sg));
            s += d; s0 += d * f0; s1 += d * f1; s2 += d * f2;
        }
    }
    red[rl][cl][0] = s; red[rl][cl][1] = s0; red[rl][cl][2] = s1; red[rl][cl][3] = s2;
    __syncthreads();
    if (rl == 0 && c < C) {
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = red[0][cl][k] + red[1][cl][k] + red[2][cl][k] + red[3][cl][k];
        atomicAdd(dtproj + (size_t)b * dtld + c, t[0]);
        atomicAdd(dbd + c, t[0]);
        atomicAdd(dWd + (size_t)c * 3 + 0, t[1]);
        atomicAdd(dWd + (size_t)c * 3 + 1, t[2]);
        atomicAdd(dWd + (size_t)c * 3 + 2, t[3]);
    }
}

// ======================================================================================
// Backward of all RPE networks of a training step in one launch (see lfvdm_rpe_bwd_job).  Same decomposition as the
// forward lfvdm_rpe_nets: workgroup = (network, 32 rows (b, t, s)); the dR tile [32][C] sits in LDS, every wave walks
// n tiles of 32 hidden channels: d_act = dR * Wout on fp32 MFMA (Wout_t chunks staged wave-privately), then the
// hidden layer's backward on the accumulator registers - lane l holds channel c = 32 nt + (l & 31) of 16 rows,
// the two half waves (l, l + 32) are folded with one shuffle and lanes 0..31 issue the float atomics.
// ======================================================================================
constexpr int RPB_LDR = 36;

// det_slab (deterministic mode): tile t stores its five per-channel partial sums to det_slab[t][5][512] instead of adding
// them with float atomics; rpe_nets_bwd_reduce_kernel then adds the tiles of every network in tile order.
constexpr int RPB_DET_C = 512, RPB_DET_LD = 5 * RPB_DET_C;
__global__ __launch_bounds__(256) void rpe_nets_bwd_kernel(const lfvdm_rpe_bwd_job* __restrict__ jobs, int njobs,
                                                           const int64_t* __restrict__ fi, int B, int T,
                                                           float* __restrict__ det_slab) {
    extern __shared__ __attribute__((aligned(16))) float rsm[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int j = 0;
    while (j + 1 < njobs && jobs[j + 1].tile0 <= (int)blockIdx.x) ++j;
    const lfvdm_rpe_bwd_job J = jobs[j];
    const int C = J.C;
    const int TT = T * T;
    const int M = B * TT;
    const int m0 = ((int)blockIdx.x - J.tile0) * 32;
    const int ALD = C + 4;
    float* As = rsm;                                        // [32][ALD] dR rows
    float* Wst = rsm + 32 * ALD + wave * 32 * RPB_LDR;      // wave-private Wout_t chunk [32][36]
    __shared__ float rowf[32][4];                           // f0, f1, f2, batch index (or -1 past the end)
    if (threadIdx.x < 32) {
        const int m = m0 + threadIdx.x;
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, fb = -1.f;
        if (m < M) {
            const int b = m / TT;
            const int rem = m - b * TT;
            const int t = rem / T, s = rem - t * T;
            const float d = (float)(fi[b * T + t] - fi[b * T + s]);
            f0 = log1pf(fmaxf(d, 0.f));
            f1 = log1pf(fmaxf(-d, 0.f));
            f2 = d == 0.f ? 1.f : 0.f;
            fb = (float)b;
        }
        rowf[threadIdx.x][0] = f0; rowf[threadIdx.x][1] = f1; rowf[threadIdx.x][2] = f2; rowf[threadIdx.x][3] = fb;
    }
    // thread = (row wave + 4 i, float4 column lane + 64 j): the loads of a column's 8 rows in flight together
    for (int k = lane * 4; k < C; k += 256) {
        f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave + 4 * i;
            v[i] = (m0 + r < M) ? ld4(J.dR + (size_t)(m0 + r) * C + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) st4(As + (wave + 4 * i) * ALD + k, v[i]);
    }
    __syncthreads();
    const int b_lo = m0 / TT;                               // T*T >= 32: the tile spans batch elements b_lo, b_lo + 1
    const int st_off = (lane >> 3) * RPB_LDR + (lane & 7) * 4;
    const int frw = (lane & 31) * RPB_LDR + (lane >> 5) * 4;
    const int fra = (lane & 31) * ALD + (lane >> 5) * 4;
    for (int nt = wave; nt * 32 < C; nt += 4) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        // (the filter chunk of step kc + 32 is in flight while chunk kc is multiplied: see rpe_nets_kernel)
        f32x4 w[4];
        const float* wsrc = J.Wout_t + (size_t)(nt * 32 + (lane >> 3)) * C + (lane & 7) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = ld4(wsrc + (size_t)r * 8 * C);
        for (int kc = 0; kc < C; kc += 32) {                // d_act[r][c] = sum_o dR[r][o] * Wout_t[c][o]
#pragma unroll
            for (int r = 0; r < 4; ++r) st4(Wst + r * 8 * RPB_LDR + st_off, w[r]);
            wave_lds_fence();
            if (kc + 32 < C) {
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = ld4(wsrc + (size_t)r * 8 * C + kc + 32);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a4 = ld4(As + fra + kc + g * 8);
                const f32x4 b4 = ld4(Wst + frw + g * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc, 0, 0, 0);
            }
            wave_lds_fence();
        }
        const int c = nt * 32 + (lane & 31);
        const float w0 = J.Wd[(size_t)c * 3], w1 = J.Wd[(size_t)c * 3 + 1], w2 = J.Wd[(size_t)c * 3 + 2], bdc = J.bd[c];
        const float tp_lo = J.tproj[(size_t)min(b_lo, B - 1) * J.tproj_ld + c] + bdc;
        const float tp_hi = J.tproj[(size_t)min(b_lo + 1, B - 1) * J.tproj_ld + c] + bdc;
        float s_lo = 0.f, s_hi = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            const float fb = rowf[r][3];
            if (fb >= 0.f) {
                const bool hi = (int)fb != b_lo;
                const float f0 = rowf[r][0], f1 = rowf[r][1], f2 = rowf[r][2];
                const float h = (hi ? tp_hi : tp_lo) + w0 * f0 + w1 * f1 + w2 * f2;
                const float sg = 1.0f / (1.0f + __expf(-h));
                const float d = acc[i] * sg * (1.0f + h * (1.0f - sg));
                s_lo += hi ? 0.f : d;
                s_hi += hi ? d : 0.f;
                s0 += d * f0; s1 += d * f1; s2 += d * f2;
            }
        }
        s_lo += __shfl_xor(s_lo, 32, 64); s_hi += __shfl_xor(s_hi, 32, 64);
        s0 += __shfl_xor(s0, 32, 64); s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (lane < 32 && det_slab) {
            float* r = det_slab + (size_t)blockIdx.x * RPB_DET_LD + c;
            r[0] = s_lo; r[RPB_DET_C] = s_hi; r[2 * RPB_DET_C] = s0; r[3 * RPB_DET_C] = s1; r[4 * RPB_DET_C] = s2;
        } else if (lane < 32) {
            atomicAdd(J.dtproj + (size_t)b_lo * J.dtproj_ld + c, s_lo);
            if (b_lo + 1 < B && (m0 + 31) / TT != b_lo) atomicAdd(J.dtproj + (size_t)(b_lo + 1) * J.dtproj_ld + c, s_hi);
            atomicAdd(J.dbd + c, s_lo + s_hi);
            atomicAdd(J.dWd + (size_t)c * 3 + 0, s0);
            atomicAdd(J.dWd + (size_t)c * 3 + 1, s1);
            atomicAdd(J.dWd + (size_t)c * 3 + 2, s2);
        }
    }
}

// one thread per (network, channel): the network's tiles in order; every destination word has ONE writer
__global__ __launch_bounds__(64) void rpe_nets_bwd_reduce_kernel(const lfvdm_rpe_bwd_job* __restrict__ jobs, const float* __restrict__ slab,
                                                                int tiles_per_job, int B, int TT) {
    const lfvdm_rpe_bwd_job J = jobs[blockIdx.x];
    const int c = blockIdx.y * 64 + threadIdx.x;
    if (c >= J.C) return;
    float dbd = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
    for (int t = 0; t < tiles_per_job; ++t) {
        const float* r = slab + (size_t)(J.tile0 + t) * RPB_DET_LD + c;
        const int m0 = t * 32, b_lo = m0 / TT;
        const float s_lo = r[0], s_hi = r[RPB_DET_C];
        J.dtproj[(size_t)b_lo * J.dtproj_ld + c] += s_lo;
        if (b_lo + 1 < B && (m0 + 31) / TT != b_lo) J.dtproj[(size_t)(b_lo + 1) * J.dtproj_ld + c] += s_hi;
        dbd += s_lo + s_hi;
        d0 += r[2 * RPB_DET_C]; d1 += r[3 * RPB_DET_C]; d2 += r[4 * RPB_DET_C];
    }
    J.dbd[c] += dbd;
    J.dWd[(size_t)c * 3 + 0] += d0; J.dWd[(size_t)c * 3 + 1] += d1; J.dWd[(size_t)c * 3 + 2] += d2;
}

template <int TMAX, int FC>
int launch_tb(const float* qkv, const float* d_o, const float* Rq, const float* Rk, const float* Rv, const float* mask, float* Pg,
              float* dSg, float* dqkv, int B, int T, int P, int C, int heads, hipStream_t s) {
    const int PPW = 64 / T;
    const int RST = T * FC + 4;
    const size_t lds_rows = (size_t)(2 * T + 4 * PPW) * RST * sizeof(float);
    const size_t lds_cols = (size_t)(T + 8 * PPW) * RST * sizeof(float);
    if (lds_rows > 160 * 1024 || lds_cols > 160 * 1024) return LFVDM_E_UNSUPPORTED;
    static DynLdsLimit limit_rows, limit_cols;
    if (int rc = limit_rows.ensure(reinterpret_cast<const void*>(&attn_temporal_bwd_rows_kernel<TMAX, FC>), lds_rows)) return rc;
    if (int rc = limit_cols.ensure(reinterpret_cast<const void*>(&attn_temporal_bwd_cols_kernel<TMAX, FC>), lds_cols)) return rc;
    const dim3 grid((unsigned)((P + 4 * PPW - 1) / (4 * PPW)), (unsigned)heads, (unsigned)B);
    static const bool rows_v1 = getenv("LFVDM_ATTN_BWD_ROWS_V1") != nullptr;     // A/B aid
    int rc2 = rows_v1 ? LFVDM_E_UNSUPPORTED : lfvdm_attn_temporal2_bwd_rows_try(qkv, d_o, Rq, Rk, Rv, mask, dqkv, Pg, dSg, B, T, P, C, heads, s);
    if (rc2 == LFVDM_E_UNSUPPORTED) {
        hipLaunchKernelGGL((attn_temporal_bwd_rows_kernel<TMAX, FC>), grid, dim3(256), lds_rows, s, qkv, d_o, Rq, Rk, Rv, mask, dqkv,
                           Pg, dSg, T, P, C, heads, PPW);
        LFVDM_CHECK_LAUNCH();
    } else if (rc2 != LFVDM_OK) {
        return rc2;
    }
    hipLaunchKernelGGL((attn_temporal_bwd_cols_kernel<TMAX, FC>), grid, dim3(256), lds_cols, s, qkv, d_o, Rq, Pg, dSg, dqkv, T, P,
                       C, heads, PPW);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

template <int FC>
int launch_tb_t(const float* qkv, const float* d_o, const float* Rq, const float* Rk, const float* Rv, const float* mask,
                float* Pg, float* dSg, float* dqkv, int B, int T, int P, int C, int heads, hipStream_t s) {
    if (T <= 8) return launch_tb<8, FC>(qkv, d_o, Rq, Rk, Rv, mask, Pg, dSg, dqkv, B, T, P, C, heads, s);
    if (T <= 16) return launch_tb<16, FC>(qkv, d_o, Rq, Rk, Rv, mask, Pg, dSg, dqkv, B, T, P, C, heads, s);
    if (T <= 24) return launch_tb<24, FC>(qkv, d_o, Rq, Rk, Rv, mask, Pg, dSg, dqkv, B, T, P, C, heads, s);
    return launch_tb<32, FC>(qkv, d_o, Rq, Rk, Rv, mask, Pg, dSg, dqkv, B, T, P, C, heads, s);
}

}  // namespace

extern "C" int lfvdm_attn_temporal_bwd(const float* qkv, const float* d_o, const float* Rq, const float* Rk, const float* Rv,
                                       const float* mask, float* ws_p, float* ws_ds, float* dqkv, float* dRq, float* dRk,
                                       float* dRv, int B, int T, int P, int C, int heads, void* stream) {
    if (B <= 0 || T <= 0 || T > TB_MAXT || P <= 0 || heads <= 0 || C % heads) return LFVDM_E_SHAPE;
    if (!qkv || !d_o || !Rq || !Rk || !Rv || !ws_p || !ws_ds || !dqkv || !dRq || !dRk || !dRv) return LFVDM_E_SHAPE;
    const int F = C / heads;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (F % 16 == 0 && T <= 24) rc = launch_tb_t<16>(qkv, d_o, Rq, Rk, Rv, mask, ws_p, ws_ds, dqkv, B, T, P, C, heads, s);
    else if (F % 8 == 0) rc = launch_tb_t<8>(qkv, d_o, Rq, Rk, Rv, mask, ws_p, ws_ds, dqkv, B, T, P, C, heads, s);
    else return LFVDM_E_UNSUPPORTED;
    if (rc != LFVDM_OK) return rc;
    hipLaunchKernelGGL(attn_temporal_bwd_rpe_kernel, dim3((unsigned)T, (unsigned)heads, (unsigned)B), dim3(256), 0, s, qkv, d_o,
                       ws_p, ws_ds, dRq, dRk, dRv, T, P, C, heads);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_rpe_front(const float* tproj, int tproj_ld, const float* feats, const float* Wd, const float* bd, float* act,
                               int B, int rows_per_b, int C, void* stream) {
    if (!tproj || !feats || !Wd || !bd || !act || B <= 0 || rows_per_b <= 0 || C <= 0 || C % 4 || tproj_ld < C || tproj_ld % 4)
        return LFVDM_E_SHAPE;
    const long rows = (long)B * rows_per_b;
    const long n = rows * (C / 4);
    hipLaunchKernelGGL(rpe_front_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tproj, tproj_ld, feats,
                       Wd, bd, act, rows, rows_per_b, C);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_rpe_front_bwd(const float* tproj, int tproj_ld, const float* feats, const float* Wd, const float* bd,
                                   const float* d_act, float* dtproj, int dtproj_ld, float* dWd, float* dbd, int B,
                                   int rows_per_b, int C, void* stream) {
    if (!tproj || !feats || !Wd || !bd || !d_act || !dtproj || !dWd || !dbd || B <= 0 || rows_per_b <= 0 || C <= 0 ||
        tproj_ld < C || dtproj_ld < C)
        return LFVDM_E_SHAPE;
    const int chunk = 32;
    hipLaunchKernelGGL(rpe_front_bwd_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)B, (unsigned)((rows_per_b + chunk - 1) / chunk)),
                       dim3(256), 0, (hipStream_t)stream, tproj, tproj_ld, feats, Wd, bd, d_act, dtproj, dtproj_ld, dWd, dbd, rows_per_b, C,
                       chunk);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

static int rpe_nets_bwd_impl(const lfvdm_rpe_bwd_job* jobs_dev, int njobs, int total_tiles, const int64_t* fi, int B, int T,
                            float* det_ws, long det_ws_floats, hipStream_t s) {
    if (!jobs_dev || !fi || njobs <= 0 || total_tiles <= 0 || B <= 0 || T <= 0 || T * T < 32) return LFVDM_E_SHAPE;
    if (det_ws && ((long)total_tiles * RPB_DET_LD > det_ws_floats || total_tiles % njobs)) return LFVDM_E_SHAPE;
    const int maxC = 512;    // LDS sized for the largest supported C: dR tile 32*(C+4) + 4 wave-private W chunks
    const size_t lds = (size_t)(32 * (maxC + 4) + 4 * 32 * RPB_LDR) * sizeof(float);
    static DynLdsLimit limit;
    if (int rc = limit.ensure(reinterpret_cast<const void*>(&rpe_nets_bwd_kernel), lds)) return rc;
    hipLaunchKernelGGL(rpe_nets_bwd_kernel, dim3(total_tiles), dim3(256), lds, s, jobs_dev, njobs, fi, B, T, det_ws);
    LFVDM_CHECK_LAUNCH();
    if (det_ws) {
        hipLaunchKernelGGL(rpe_nets_bwd_reduce_kernel, dim3(njobs, maxC / 64), dim3(64), 0, s, jobs_dev, det_ws, total_tiles / njobs, B,
                           T * T);
        LFVDM_CHECK_LAUNCH();
    }
    return LFVDM_OK;
}

extern "C" int lfvdm_rpe_nets_bwd(const lfvdm_rpe_bwd_job* jobs_dev, int njobs, int total_tiles, const int64_t* fi, int B,
                                  int T, void* stream) {
    return rpe_nets_bwd_impl(jobs_dev, njobs, total_tiles, fi, B, T, nullptr, 0, (hipStream_t)stream);
}

extern "C" int lfvdm_rpe_nets_bwd_det(const lfvdm_rpe_bwd_job* jobs_dev, int njobs, int total_tiles, const int64_t* fi, int B,
                                      int T, float* det_ws, int64_t det_ws_floats, void* stream) {
    if (!det_ws) return LFVDM_E_SHAPE;
    return rpe_nets_bwd_impl(jobs_dev, njobs, total_tiles, fi, B, T, det_ws, (long)det_ws_floats, (hipStream_t)stream);
}
