// Attention cores and the RPE networks for gfx950.
//
//  * rpe_nets_kernel   : every RPENet output projection of one forward (rpe.py:20-31) as one
//                        grouped launch; hidden rows are generated on the fly into LDS and
//                        multiplied with Wout on fp32 MFMA 32x32x2.
//  * attn_spatial      : flash-style MHA over the H*W tokens of a frame on fp32 MFMA 16x16x4.
//                        S^T = K.Q^T is computed with keys on the accumulator rows, so the
//                        softmax probabilities are already the B operand of O^T = V^T.P^T
//                        (no LDS round trip for P); softmax reductions are wave shuffles.
//  * attn_temporal     : one wave per (batch, pixel, head); T <= 32 frames attend with the three
//                        RPE terms and the two-clique mask (rpe.py:143-169) on the VALU.
#include "common.cuh"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// ======================================================================================
// RPE nets
// ======================================================================================
constexpr int RPE_LDR = 36;

__global__ __launch_bounds__(256) void rpe_nets_kernel(const lfvdm_rpe_job* __restrict__ jobs, int njobs,
                                                       const int64_t* __restrict__ fi, int B, int T) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int j = 0;
    while (j + 1 < njobs && jobs[j + 1].tile0 <= (int)blockIdx.x) ++j;
    const lfvdm_rpe_job J = jobs[j];
    const int C = J.C;
    const int M = B * T * T;
    const int m0 = ((int)blockIdx.x - J.tile0) * 32;
    const int ALD = C + 4;                       // A tile row stride (floats), 16B aligned, conflict-free b128
    float* As = smem;                            // [32][ALD] silu(hidden)
    float* Wst = smem + 32 * ALD + wave * 32 * RPE_LDR;  // wave-private W chunk [32][36]

    // ---- hidden rows: silu(tproj[b] + Wd feats + bd) ----
    __shared__ float rowf[32][4];  // f0, f1, f2, batch index (or -1 past the end)
    if (threadIdx.x < 32) {
        const int m = m0 + threadIdx.x;
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, fb = -1.f;
        if (m < M) {
            const int b = m / (T * T);
            const int rem = m - b * T * T;
            const int t = rem / T, s = rem - t * T;
            const float d = (float)(fi[b * T + t] - fi[b * T + s]);
            f0 = log1pf(fmaxf(d, 0.f));
            f1 = log1pf(fmaxf(-d, 0.f));
            f2 = d == 0.f ? 1.f : 0.f;
            fb = (float)b;
        }
        rowf[threadIdx.x][0] = f0; rowf[threadIdx.x][1] = f1; rowf[threadIdx.x][2] = f2; rowf[threadIdx.x][3] = fb;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * C; e += 256) {
        const int r = e / C, k = e - r * C;
        float v = 0.f;
        if (rowf[r][3] >= 0.f) {
            const int b = (int)rowf[r][3];
            const float hd = (rowf[r][0] * J.Wd[k * 3 + 0] + rowf[r][1] * J.Wd[k * 3 + 1] + rowf[r][2] * J.Wd[k * 3 + 2]) + J.bd[k];
            v = silu_f(J.tproj[b * C + k] + hd);
        }
        As[r * ALD + k] = v;
    }
    __syncthreads();

    // ---- each wave: n-tiles wave, wave+4, ... of 32 output channels ----
    const int st_off = (lane >> 3) * RPE_LDR + (lane & 7) * 4;
    const int frw = (lane & 31) * RPE_LDR + (lane >> 5) * 4;
    const int fra = (lane & 31) * ALD + (lane >> 5) * 4;
    for (int nt = wave; nt * 32 < C; nt += 4) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int kc = 0; kc < C; kc += 32) {
            f32x4 w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = ld4(J.Wout + (size_t)(nt * 32 + r * 8 + (lane >> 3)) * C + kc + (lane & 7) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) st4(Wst + r * 8 * RPE_LDR + st_off, w[r]);
            wave_lds_fence();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a4 = ld4(As + fra + kc + g * 8);
                const f32x4 b4 = ld4(Wst + frw + g * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc, 0, 0, 0);
            }
            wave_lds_fence();
        }
        const int co = nt * 32 + (lane & 31);
        const float bo = J.bout[co];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < M) J.R[(size_t)m * C + co] = acc[r] + bo;
        }
    }
}

// ======================================================================================
// Spatial attention (flash style, fp32 MFMA 16x16x4)
// ======================================================================================
// Workgroup = 4 waves = 64 queries of one (frame n, head h); key blocks of 64 are staged in LDS
// (K rows padded to F+8, V rows to F+4 floats: conflict-free b128 / b32 fragment reads).
// k-index mapping inside a 16-wide f group g: MFMA step e uses f = 16g + 4*kk + e (kk = lane>>4)
// on both K (A operand) and Q (B operand).
// FP = head dim padded to a multiple of 16 (compile time); F = real head dim (multiple of 4): the
// pad columns are zero in LDS / in the Q fragments and are never stored.
template <int FP>
__global__ __launch_bounds__(256) void attn_spatial_kernel(const float* __restrict__ qkv, float* __restrict__ o, int P,
                                                           int C, int heads, int F) {
    constexpr int FG = FP / 16;   // 16-wide f groups
    constexpr int KLD = FP + 8, VLD = FP + 4;
    __shared__ __attribute__((aligned(16))) float Ks[64 * KLD];
    __shared__ __attribute__((aligned(16))) float Vs[64 * VLD];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const int lq = lane & 15, kk = lane >> 4;
    const float scale = rsqrtf((float)F);
    const size_t ld = (size_t)3 * C;
    const float* base = qkv + (size_t)n * P * ld + h * F;

    // Q fragments (B operand), pre-scaled: lane (q = lq, kk) holds Q[q][16g + 4kk + e]
    f32x4 qf[FG];
    {
        const int q = q0 + lq;
#pragma unroll
        for (int g = 0; g < FG; ++g) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (q < P && 16 * g + 4 * kk < F) v = ld4(base + (size_t)q * ld + 16 * g + 4 * kk);
            qf[g] = v * scale;
        }
    }
    f32x4 oacc[FG];
#pragma unroll
    for (int g = 0; g < FG; ++g) oacc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    for (int kb = 0; kb < P; kb += 64) {
        __syncthreads();  // previous block fully consumed
        for (int e = threadIdx.x; e < 64 * (FP / 4); e += 256) {
            const int key = e / (FP / 4), fq = e - key * (FP / 4);
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (kb + key < P && fq * 4 < F) {
                const float* row = base + (size_t)(kb + key) * ld + fq * 4;
                kv = ld4(row + C);
                vv = ld4(row + 2 * C);
            }
            st4(Ks + key * KLD + fq * 4, kv);
            st4(Vs + key * VLD + fq * 4, vv);
        }
        __syncthreads();

        // S^T tiles: rows = keys 16j + 4kk + r, column = query lq
        f32x4 s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < FG; ++g) {
                const f32x4 k4 = ld4(Ks + (16 * j + lq) * KLD + 16 * g + 4 * kk);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k4[e], qf[g][e], acc, 0, 0, 0);
            }
            s[j] = acc;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (kb + 16 * j + 4 * kk + r >= P) s[j][r] = -INFINITY;
                mx = fmaxf(mx, s[j][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);  // m_run = -inf on the first block -> 0
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(s[j][r] - m_new);
                s[j][r] = pv;
                psum += pv;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int g = 0; g < FG; ++g) oacc[g] *= alpha;
        // O^T[f][q] += V^T[f][key] * P^T[key][q]; MFMA (j, r): k index kk <-> key 16j + 4kk + r
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vrow = Vs + (16 * j + 4 * kk + r) * VLD + lq;
#pragma unroll
                for (int g = 0; g < FG; ++g)
                    oacc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[16 * g], s[j][r], oacc[g], 0, 0, 0);
            }
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_run;
    const int q = q0 + lq;
    if (q < P) {
#pragma unroll
        for (int g = 0; g < FG; ++g)
            if (16 * g + 4 * kk < F) st4(o + ((size_t)n * P + q) * C + h * F + 16 * g + 4 * kk, oacc[g] * inv);
    }
}

// probabilities for logging (return_attn_weights): plain two-pass softmax, one wave per query row
__global__ __launch_bounds__(256) void attn_spatial_probs_kernel(const float* __restrict__ qkv, float* __restrict__ attn,
                                                                 int P, int C, int heads, int F) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // (h, q) of frame n
    const int n = blockIdx.y;
    if (row >= (long)heads * P) return;
    const int h = (int)(row / P), q = (int)(row % P);
    const float scale = rsqrtf((float)F);
    const size_t ld = (size_t)3 * C;
    const float* base = qkv + (size_t)n * P * ld + h * F;
    const float* qr = base + (size_t)q * ld;
    float* out = attn + (((size_t)n * heads + h) * P + q) * P;
    float mx = -INFINITY;
    for (int k = lane; k < P; k += 64) {
        const float* kr = base + (size_t)k * ld + C;
        float d = 0.f;
        for (int f = 0; f < F; ++f) d += qr[f] * scale * kr[f];
        out[k] = d;
        mx = fmaxf(mx, d);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < P; k += 64) {
        const float e = __expf(out[k] - mx);
        out[k] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int k = lane; k < P; k += 64) out[k] *= inv;
}

// ======================================================================================
// Temporal attention with RPE
// ======================================================================================
// One wave per (b, pixel, head).  Lane (t = lane&31, half = lane>>5) owns query frame t and the key
// frames s = half, half+2, ...  q/k/v of the (pixel, head) are staged wave-privately in LDS.
constexpr int TA_MAXT = 32;

template <int F>
__global__ __launch_bounds__(256) void attn_temporal_kernel(const float* __restrict__ qkv, const float* __restrict__ Rq,
                                                            const float* __restrict__ Rk, const float* __restrict__ Rv,
                                                            const float* __restrict__ mask, float* __restrict__ o,
                                                            float* __restrict__ attn_out, int B, int T, int P, int C,
                                                            int heads) {
    constexpr int LD = F + 4;
    __shared__ __attribute__((aligned(16))) float sm[4][3 * TA_MAXT * LD];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long wid = (long)blockIdx.x * 4 + wave;  // ((b*P + p)*heads + h)
    if (wid >= (long)B * P * heads) return;
    const int h = (int)(wid % heads);
    const long bp = wid / heads;
    const int p = (int)(bp % P), b = (int)(bp / P);
    float* qs = sm[wave];
    float* ks = qs + TA_MAXT * LD;
    float* vs = ks + TA_MAXT * LD;
    const float scale = rsqrtf((float)F);
    const size_t ld = (size_t)3 * C;

    // stage q*scale, k, v rows of the T frames
    for (int e = lane; e < T * (F / 4); e += 64) {
        const int t = e / (F / 4), fq = e - t * (F / 4);
        const float* row = qkv + ((size_t)(b * T + t) * P + p) * ld + h * F + fq * 4;
        st4(qs + t * LD + fq * 4, ld4(row) * scale);
        st4(ks + t * LD + fq * 4, ld4(row + C));
        st4(vs + t * LD + fq * 4, ld4(row + 2 * C));
    }
    wave_lds_fence();

    const int t = lane & 31, half = lane >> 5;
    const bool tv = t < T;
    const int tt = tv ? t : 0;
    const float mt = mask ? mask[b * T + tt] : 1.f;
    float logit[TA_MAXT / 2];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < TA_MAXT / 2; ++i) {
        const int s = 2 * i + half;
        float acc = -INFINITY;
        if (tv && s < T) {
            const float* rk = Rk + (((size_t)(b * T + tt) * T + s) * C) + h * F;
            const float* rq = Rq + (((size_t)(b * T + s) * T + tt) * C) + h * F;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int f = 0; f < F; f += 4) {
                const f32x4 q4 = ld4(qs + tt * LD + f), k4 = ld4(ks + s * LD + f);
                const f32x4 rk4 = ld4(rk + f), rq4 = ld4(rq + f);
                a0 += q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
                a1 += q4.x * rk4.x + q4.y * rk4.y + q4.z * rk4.z + q4.w * rk4.w;
                a2 += (k4.x * scale) * rq4.x + (k4.y * scale) * rq4.y + (k4.z * scale) * rq4.z + (k4.w * scale) * rq4.w;
            }
            acc = a0 + a1 + a2;
            if (mask) {
                const float ms = mask[b * T + s];
                const float allowed = mt * ms + (1.f - mt) * (1.f - ms);
                const float pen = 1.f - allowed;
                acc -= (pen == 1.f) ? INFINITY : pen;
            }
        }
        logit[i] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TA_MAXT / 2; ++i) {
        const float e = (logit[i] == -INFINITY) ? 0.f : __expf(logit[i] - mx);
        logit[i] = e;
        sum += e;
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    if (attn_out && tv) {
        float* ar = attn_out + ((size_t)wid * T + tt) * T;
#pragma unroll
        for (int i = 0; i < TA_MAXT / 2; ++i) {
            const int s = 2 * i + half;
            if (s < T) ar[s] = logit[i] * inv;
        }
    }

    // o[t][f] = sum_s p[t][s] * (v[s][f] + Rv[t][s][f])
    constexpr int FO = F < 16 ? F : 16;   // output channels per pass
    constexpr int NU = FO / 4;
#pragma unroll
    for (int f0 = 0; f0 < F; f0 += FO) {
        f32x4 acc[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TA_MAXT / 2; ++i) {
            const int s = 2 * i + half;
            if (tv && s < T) {
                const float pr = logit[i] * inv;
                const float* rv = Rv + (((size_t)(b * T + tt) * T + s) * C) + h * F + f0;
#pragma unroll
                for (int u = 0; u < NU; ++u) acc[u] += pr * (ld4(vs + s * LD + f0 + 4 * u) + ld4(rv + 4 * u));
            }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            acc[u].x += __shfl_xor(acc[u].x, 32, 64); acc[u].y += __shfl_xor(acc[u].y, 32, 64);
            acc[u].z += __shfl_xor(acc[u].z, 32, 64); acc[u].w += __shfl_xor(acc[u].w, 32, 64);
        }
        if (tv && half == 0) {
            float* orow = o + ((size_t)(b * T + tt) * P + p) * C + h * F + f0;
#pragma unroll
            for (int u = 0; u < NU; ++u) st4(orow + 4 * u, acc[u]);
        }
    }
}

}  // namespace

extern "C" int lfvdm_rpe_nets(const lfvdm_rpe_job* jobs_dev, int njobs, int total_tiles, const int64_t* fi, int B, int T,
                              void* stream) {
    if (njobs <= 0 || total_tiles <= 0 || B <= 0 || T <= 0) return LFVDM_E_SHAPE;
    // LDS sized for the largest supported C (512): A tile 32*(C+4) + 4 wave-private W chunks
    const int maxC = 512;
    const size_t lds = (size_t)(32 * (maxC + 4) + 4 * 32 * RPE_LDR) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rpe_nets_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return LFVDM_E_LAUNCH;
        attr = true;
    }
    hipLaunchKernelGGL(rpe_nets_kernel, dim3(total_tiles), dim3(256), lds, (hipStream_t)stream, jobs_dev, njobs, fi, B, T);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_attn_spatial(const float* qkv, float* o, float* attn_out, int N, int P, int C, int heads, void* stream) {
    if (N <= 0 || P <= 0 || heads <= 0 || C % heads) return LFVDM_E_SHAPE;
    const int F = C / heads;
    const dim3 grid((P + 63) / 64, heads, N);
    hipStream_t s = (hipStream_t)stream;
    if (F % 4 || F > 96) return LFVDM_E_UNSUPPORTED;
    const int FP = (F + 15) / 16 * 16;
    switch (FP) {
        case 16: hipLaunchKernelGGL(attn_spatial_kernel<16>, grid, dim3(256), 0, s, qkv, o, P, C, heads, F); break;
        case 32: hipLaunchKernelGGL(attn_spatial_kernel<32>, grid, dim3(256), 0, s, qkv, o, P, C, heads, F); break;
        case 48: hipLaunchKernelGGL(attn_spatial_kernel<48>, grid, dim3(256), 0, s, qkv, o, P, C, heads, F); break;
        case 64: hipLaunchKernelGGL(attn_spatial_kernel<64>, grid, dim3(256), 0, s, qkv, o, P, C, heads, F); break;
        case 96: hipLaunchKernelGGL(attn_spatial_kernel<96>, grid, dim3(256), 0, s, qkv, o, P, C, heads, F); break;
        default: return LFVDM_E_UNSUPPORTED;
    }
    LFVDM_CHECK_LAUNCH();
    if (attn_out) {
        hipLaunchKernelGGL(attn_spatial_probs_kernel, dim3((heads * P + 3) / 4, N), dim3(256), 0, s, qkv, attn_out, P, C, heads, F);
        LFVDM_CHECK_LAUNCH();
    }
    return LFVDM_OK;
}

extern "C" int lfvdm_attn_temporal(const float* qkv, const float* Rq, const float* Rk, const float* Rv, const float* mask,
                                   float* o, float* attn_out, int B, int T, int P, int C, int heads, void* stream) {
    if (B <= 0 || T <= 0 || T > TA_MAXT || P <= 0 || heads <= 0 || C % heads) return LFVDM_E_SHAPE;
    if (!Rq || !Rk || !Rv) return LFVDM_E_SHAPE;
    const int F = C / heads;
    const long waves = (long)B * P * heads;
    const dim3 grid((unsigned)((waves + 3) / 4));
    hipStream_t s = (hipStream_t)stream;
    switch (F) {
        case 8: hipLaunchKernelGGL(attn_temporal_kernel<8>, grid, dim3(256), 0, s, qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads); break;
        case 16: hipLaunchKernelGGL(attn_temporal_kernel<16>, grid, dim3(256), 0, s, qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads); break;
        case 32: hipLaunchKernelGGL(attn_temporal_kernel<32>, grid, dim3(256), 0, s, qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads); break;
        case 64: hipLaunchKernelGGL(attn_temporal_kernel<64>, grid, dim3(256), 0, s, qkv, Rq, Rk, Rv, mask, o, attn_out, B, T, P, C, heads); break;
        default: return LFVDM_E_UNSUPPORTED;
    }
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}
