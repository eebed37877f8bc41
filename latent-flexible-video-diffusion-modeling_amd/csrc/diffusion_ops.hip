// Elementwise Gaussian-diffusion step math (gaussian_diffusion.py:200-218, 290-346, 369-401,
// 787-788).  HBM-bound streaming kernels: float4 accesses, one table gather per batch row.
#include "common_hip.h"

namespace {

__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                       const int64_t* __restrict__ t, const float* __restrict__ sa,
                                                       const float* __restrict__ sb, float* __restrict__ out, int inner) {
    const int b = blockIdx.y;
    const float ca = sa[t[b]], cb = sb[t[b]];
    const size_t base = (size_t)b * inner;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4; i < inner; i += gridDim.x * blockDim.x * 4) {
        if (i + 3 < inner) {
            st4(out + base + i, ca * ld4(x0 + base + i) + cb * ld4(noise + base + i));
        } else {
            for (int j = i; j < inner; ++j) out[base + j] = ca * x0[base + j] + cb * noise[base + j];
        }
    }
}

// x and sample may alias (the sampler updates its state in place): no __restrict__ on them.
__global__ __launch_bounds__(256) void p_sample_kernel(const float* x, const float* __restrict__ eps,
                                                       const float* __restrict__ noise, const int64_t* __restrict__ t,
                                                       const float* __restrict__ t_recip, const float* __restrict__ t_recipm1,
                                                       const float* __restrict__ t_c1, const float* __restrict__ t_c2,
                                                       const float* __restrict__ t_logvar, int clip,
                                                       float* sample, float* __restrict__ pred,
                                                       float* __restrict__ mean_out, int inner) {
    const int b = blockIdx.y;
    const int64_t tb = t[b];
    const float r = t_recip[tb], rm1 = t_recipm1[tb], c1 = t_c1[tb], c2 = t_c2[tb];
    // sample = mean + [t != 0] * exp(0.5 * log_variance) * noise     (:396-400)
    const float sigma = tb != 0 ? expf(0.5f * t_logvar[tb]) : 0.f;
    const size_t base = (size_t)b * inner;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < inner; i += gridDim.x * blockDim.x) {
        const float xv = x[base + i];
        float p0 = r * xv - rm1 * eps[base + i];   // _predict_xstart_from_eps (:341-346)
        if (clip) p0 = fminf(fmaxf(p0, -1.f), 1.f);
        const float mean = c1 * p0 + c2 * xv;       // q_posterior_mean_variance (:228-231)
        float sv = mean;
        if (tb != 0) sv += sigma * noise[base + i];
        sample[base + i] = sv;
        if (pred) pred[base + i] = p0;
        if (mean_out) mean_out[base + i] = mean;
    }
}

// ---- the same update with the Gaussian noise drawn in the kernel (one launch less per sampling step).
// Counter-based: Philox4x32-10 (Salmon et al., SC'11) keyed by the chain's 64-bit seed, counter = (quad index within the
// row, batch row, timestep, 0) -> four uniform words -> two Box-Muller pairs = the noise of four consecutive elements.
// A given (seed, timestep, element) always gets the same value, whatever the launch geometry.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ f32x4 normal4(unsigned quad, unsigned row, unsigned step, unsigned k0, unsigned k1) {
    unsigned r[4];
    philox4x32_10(quad, row, step, 0u, k0, k1, r);
    const float two_pi = 6.283185307179586f, s32 = 2.3283064365386963e-10f;      // 2^-32
    const float u0 = ((float)r[0] + 1.0f) * s32, u1 = (float)r[1] * s32;         // u0 in (0, 1]
    const float u2 = ((float)r[2] + 1.0f) * s32, u3 = (float)r[3] * s32;
    const float m0 = sqrtf(-2.0f * logf(u0)), m1 = sqrtf(-2.0f * logf(u2));
    float s0, c0, s1, c1;
    sincosf(two_pi * u1, &s0, &c0);
    sincosf(two_pi * u3, &s1, &c1);
    return (f32x4){m0 * c0, m0 * s0, m1 * c1, m1 * s1};
}

__global__ __launch_bounds__(256) void p_sample_rng_kernel(const float* x, const float* __restrict__ eps,
                                                           float* __restrict__ noise_out, const int64_t* __restrict__ t,
                                                           const float* __restrict__ t_recip, const float* __restrict__ t_recipm1,
                                                           const float* __restrict__ t_c1, const float* __restrict__ t_c2,
                                                           const float* __restrict__ t_logvar, int clip, float* sample,
                                                           float* __restrict__ pred, float* __restrict__ mean_out, int inner,
                                                           const int64_t* __restrict__ seed) {
    const int b = blockIdx.y;
    const int64_t tb = t[b];
    const float r = t_recip[tb], rm1 = t_recipm1[tb], c1 = t_c1[tb], c2 = t_c2[tb];
    const float sigma = tb != 0 ? expf(0.5f * t_logvar[tb]) : 0.f;
    const unsigned long long key = (unsigned long long)seed[0];
    const unsigned k0 = (unsigned)key, k1 = (unsigned)(key >> 32);
    const size_t base = (size_t)b * inner;
    const int nq = (inner + 3) >> 2;
    for (int qd = blockIdx.x * blockDim.x + threadIdx.x; qd < nq; qd += gridDim.x * blockDim.x) {
        const f32x4 z = normal4((unsigned)qd, (unsigned)b, (unsigned)tb, k0, k1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * qd + e;
            if (i < inner) {
                const float xv = x[base + i];
                float p0 = r * xv - rm1 * eps[base + i];
                if (clip) p0 = fminf(fmaxf(p0, -1.f), 1.f);
                const float mean = c1 * p0 + c2 * xv;
                sample[base + i] = mean + sigma * z[e];
                if (noise_out) noise_out[base + i] = z[e];
                if (pred) pred[base + i] = p0;
                if (mean_out) mean_out[base + i] = mean;
            }
        }
    }
}

// ---- The U-Net's output convolution and the update in ONE launch (the last two launches of a replayed sampling step:
// a 64 -> 4 convolution run as an implicit GEMM with 28 of its 32 filter columns padding, 10.4 us, and the update,
// 5.2 us).  eps = conv3x3(act) + bias with act = SiLU(GN(h)) as channels-last rows (reference unet.py:399-403,462-464),
// then exactly p_sample_rng_kernel's arithmetic and noise on it.  Direct convolution: SIXTEEN lanes share an output pixel
// and split its input channels (each lane CPL float4 of every tap: a pixel's 256-byte channel rows are read contiguously),
// partial sums combined with four butterfly steps inside the 16-lane row; a wave = four consecutive x positions = one
// noise quad of each output channel, so the lanes (pixel j, channel co < CO) finish with the update of their own element.
// The packed filters [CO][9][C] (lfvdm_pack_conv_weight) sit in LDS; the four pixels of a wave read the same addresses.
struct HeadUpdate {
    const float *act, *Wp, *bias, *x, *noise_in;
    float *eps_out, *noise_out, *sample, *pred, *mean_out;
    const int64_t *t, *seed;
    const float *t_recip, *t_recipm1, *t_c1, *t_c2, *t_logvar;
    int N, T, H, W, C, clip;
};

template <int CO, int CPL>
__global__ __launch_bounds__(256) void conv_out_psample_kernel(const HeadUpdate p) {
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [CO * 9][C]
    const int tid = threadIdx.x;
    const int C = p.C;
    {   // filters -> LDS, all loads in flight before the first store (C = 64 CPL: the trip count is a constant)
        constexpr int NW4 = CO * 9 * 16 * CPL, NIT = (NW4 + 255) / 256;
        f32x4 wv[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = tid + 256 * u;
            wv[u] = i < NW4 ? ld4(p.Wp + 4 * (size_t)i) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = tid + 256 * u;
            if (i < NW4) st4(wl + 4 * i, wv[u]);
        }
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int WQ = p.W >> 2;
    const long quad = (long)blockIdx.x * 4 + wave;
    if (quad >= (long)p.N * p.H * WQ) return;                       // whole wave, after the only barrier
    const int n = (int)(quad / (p.H * WQ));
    const int rem = (int)(quad - (long)n * p.H * WQ);
    const int y = rem / WQ, xq = rem - y * WQ;
    const int j = lane >> 4, cl = lane & 15;
    const int x = 4 * xq + j;
    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
        const bool inb = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const float* a = p.act + ((size_t)(n * p.H + (inb ? iy : y)) * p.W + (inb ? ix : x)) * C;
#pragma unroll
        for (int u = 0; u < CPL; ++u) {
            const int c = (cl + 16 * u) * 4;
            f32x4 a4 = ld4(a + c);
            if (!inb) a4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                const f32x4 w4 = ld4(wl + (co * 9 + tap) * C + c);
                acc[co] += (a4.x * w4.x + a4.y * w4.y) + (a4.z * w4.z + a4.w * w4.w);
            }
        }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) acc[co] += __shfl_xor(acc[co], o, 64);
    if (cl >= CO) return;
    float e = acc[0];
#pragma unroll
    for (int co = 1; co < CO; ++co) e = cl == co ? acc[co] : e;
    const int co = cl;
    e += p.bias[co];
    const int b = n / p.T, tt = n - b * p.T;
    const int inner = p.T * CO * p.H * p.W;
    const int i = ((tt * CO + co) * p.H + y) * p.W + x;             // element of the (T, C, H, W) frame stack
    const size_t at = (size_t)b * inner + i;
    const int64_t tb = p.t[b];
    const float r = p.t_recip[tb], rm1 = p.t_recipm1[tb], c1 = p.t_c1[tb], c2 = p.t_c2[tb];
    const float sigma = tb != 0 ? expf(0.5f * p.t_logvar[tb]) : 0.f;
    float z;
    if (p.noise_in) {
        z = p.noise_in[at];
    } else {
        const unsigned long long key = (unsigned long long)p.seed[0];
        const f32x4 z4 = normal4((unsigned)(i >> 2), (unsigned)b, (unsigned)tb, (unsigned)key, (unsigned)(key >> 32));
        z = j == 0 ? z4.x : j == 1 ? z4.y : j == 2 ? z4.z : z4.w;  // x = 4 * xq + j and W % 4 == 0: i & 3 == j
    }
    const float xv = p.x[at];
    float p0 = r * xv - rm1 * e;
    if (p.clip) p0 = fminf(fmaxf(p0, -1.f), 1.f);
    const float mean = c1 * p0 + c2 * xv;
    p.sample[at] = mean + sigma * z;
    if (p.eps_out) p.eps_out[at] = e;
    if (p.noise_out) p.noise_out[at] = z;
    if (p.pred) p.pred[at] = p0;
    if (p.mean_out) p.mean_out[at] = mean;
}

// out[b] = (1/inner_total) * sum_{t, i} (a - b)^2 * mask[b, t]; one workgroup of 1024 threads per batch row, float4 loads
// (fixed summation order: deterministic).  The first version walked the row with 256 scalar-loading threads: 24 us for
// 82 k elements, twice per training step.
__global__ __launch_bounds__(1024) void masked_mse_kernel(const float* __restrict__ a, const float* __restrict__ bb,
                                                          const float* __restrict__ mask, float* __restrict__ out, int T,
                                                          int frame_inner) {
    const int b = blockIdx.x;
    const size_t base = (size_t)b * T * frame_inner;
    float acc = 0.f;
    if ((frame_inner & 3) == 0) {
        const int q = frame_inner >> 2, total = T * q;
        for (int e = threadIdx.x; e < total; e += 1024) {
            const int t = e / q;
            const float mk = mask ? mask[b * T + t] : 1.f;
            const f32x4 d = ld4(a + base + (size_t)e * 4) - ld4(bb + base + (size_t)e * 4);
            acc += ((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * mk;
        }
    } else {
        for (int t = 0; t < T; ++t) {
            const float mk = mask ? mask[b * T + t] : 1.f;
            for (int i = threadIdx.x; i < frame_inner; i += 1024) {
                const float d = a[base + (size_t)t * frame_inner + i] - bb[base + (size_t)t * frame_inner + i];
                acc += d * d * mk;
            }
        }
    }
    __shared__ float red[16];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        out[b] = t / (float)((size_t)T * frame_inner);
    }
}

// its backward: dpred = -2 (target - pred) * mask[b, t] * g[b] / inner   (closed form, one launch)
__global__ __launch_bounds__(256) void masked_mse_bwd_kernel(const float* __restrict__ target, const float* __restrict__ pred,
                                                             const float* __restrict__ mask, const float* __restrict__ g,
                                                             float* __restrict__ dpred, int T, int frame_inner) {
    const int b = blockIdx.y;
    const size_t base = (size_t)b * T * frame_inner;
    const float sc = -2.0f * g[b] / (float)((size_t)T * frame_inner);
    const long total = (long)T * frame_inner;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float mk = mask ? mask[b * T + (int)(i / frame_inner)] : 1.f;
        dpred[base + i] = (target[base + i] - pred[base + i]) * (sc * mk);
    }
}

}  // namespace

extern "C" int lfvdm_masked_mse_bwd(const float* target, const float* pred, const float* mask, const float* g, float* dpred,
                                    int B, int T, int frame_inner, void* stream) {
    if (B <= 0 || T <= 0 || frame_inner <= 0 || !target || !pred || !g || !dpred) return LFVDM_E_SHAPE;
    long gx = ((long)T * frame_inner + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(masked_mse_bwd_kernel, dim3((unsigned)gx, B), dim3(256), 0, (hipStream_t)stream, target, pred, mask, g, dpred,
                       T, frame_inner);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_acp,
                              const float* sqrt_1macp, float* out, int B, int inner, void* stream) {
    if (B <= 0 || inner <= 0) return LFVDM_E_SHAPE;
    int gx = (inner / 4 + 255) / 256;
    if (gx > 1024) gx = 1024;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(q_sample_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x0, noise, t, sqrt_acp, sqrt_1macp, out, inner);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

namespace {
// one thread per batch element: t <- max(t - 1, 0); model_t <- table[t]
__global__ void sampler_tick_kernel(int64_t* __restrict__ t, const float* __restrict__ table, float* __restrict__ model_t, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int64_t v = t[b] - 1;
    v = v < 0 ? 0 : v;
    t[b] = v;
    model_t[b] = table[v];
}

// tick + fetch in ONE workgroup: the rows of the per-timestep table that belong to the new t replace the per-step
// embedding launches (the table is built once per chain with the very same kernels, _engine.Plan.build_time_tables)
__global__ __launch_bounds__(1024) void sampler_tick_fetch_kernel(int64_t* __restrict__ t, const float* __restrict__ table,
                                                                  float* __restrict__ model_t, int B,
                                                                  const float* __restrict__ rows_all, int rows_ld,
                                                                  float* __restrict__ rows, int row_floats) {
    __shared__ int64_t tnew[64];
    if ((int)threadIdx.x < B) {
        int64_t v = t[threadIdx.x] - 1;
        v = v < 0 ? 0 : v;
        t[threadIdx.x] = v;
        model_t[threadIdx.x] = table[v];
        tnew[threadIdx.x] = v;
    }
    __syncthreads();
    const int q4 = row_floats >> 2;
    for (int e = threadIdx.x; e < B * q4; e += 1024) {
        const int b = e / q4, i = e - b * q4;
        const float* src = rows_all + ((size_t)tnew[b] * B + b) * rows_ld;      // both buffers have rows of rows_ld floats,
        st4(rows + (size_t)b * rows_ld + 4 * i, ld4(src + 4 * i));              // of which the first row_floats are fetched
    }
}

}  // namespace

extern "C" int lfvdm_sampler_tick_fetch(int64_t* t, const float* model_timestep_table, float* model_t, int B,
                                        const float* rows_all, int rows_ld, float* rows, int row_floats, void* stream) {
    if (B <= 0 || B > 64 || !t || !model_timestep_table || !model_t || !rows_all || !rows) return LFVDM_E_SHAPE;
    if (row_floats <= 0 || (row_floats & 3) || (rows_ld & 3) || rows_ld < row_floats) return LFVDM_E_SHAPE;
    hipLaunchKernelGGL(sampler_tick_fetch_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, t, model_timestep_table, model_t, B,
                       rows_all, rows_ld, rows, row_floats);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_sampler_tick(int64_t* t, const float* model_timestep_table, float* model_t, int B, void* stream) {
    if (B <= 0 || !t || !model_timestep_table || !model_t) return LFVDM_E_SHAPE;
    hipLaunchKernelGGL(sampler_tick_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, t, model_timestep_table,
                       model_t, B);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_p_sample(const float* x, const float* eps, const float* noise, const int64_t* t,
                              const float* sqrt_recip_acp, const float* sqrt_recipm1_acp, const float* coef1,
                              const float* coef2, const float* log_var, int clip, float* sample, float* pred_xstart,
                              float* mean_out, int B, int inner, void* stream) {
    if (B <= 0 || inner <= 0) return LFVDM_E_SHAPE;
    int gx = (inner + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(p_sample_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x, eps, noise, t, sqrt_recip_acp,
                       sqrt_recipm1_acp, coef1, coef2, log_var, clip, sample, pred_xstart, mean_out, inner);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_p_sample_rng(const float* x, const float* eps, float* noise_out, const int64_t* t,
                                  const float* sqrt_recip_acp, const float* sqrt_recipm1_acp, const float* coef1,
                                  const float* coef2, const float* log_var, int clip, float* sample, float* pred_xstart,
                                  float* mean_out, int B, int inner, const int64_t* seed, void* stream) {
    if (B <= 0 || inner <= 0 || !seed) return LFVDM_E_SHAPE;
    int gx = ((inner + 3) / 4 + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(p_sample_rng_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x, eps, noise_out, t, sqrt_recip_acp,
                       sqrt_recipm1_acp, coef1, coef2, log_var, clip, sample, pred_xstart, mean_out, inner, seed);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_conv_out_psample_ok(int N, int H, int W, int C, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || W % 4 || (Cout != 3 && Cout != 4)) return LFVDM_E_UNSUPPORTED;
    if (C != 64 && C != 128 && C != 256) return LFVDM_E_UNSUPPORTED;
    if ((long)N * H * W * C >= (1L << 31) / 4) return LFVDM_E_UNSUPPORTED;
    return LFVDM_OK;
}

extern "C" int lfvdm_conv_out_psample(const float* act, const float* Wp, const float* bias, float* eps_out, const float* x,
                                      const float* noise_in, float* noise_out, const int64_t* t, const float* sqrt_recip_acp,
                                      const float* sqrt_recipm1_acp, const float* coef1, const float* coef2,
                                      const float* log_var, int clip, float* sample, float* pred_xstart, float* mean_out,
                                      int B, int T, int H, int W, int C, int Cout, const int64_t* seed, void* stream) {
    if (B <= 0 || T <= 0 || !act || !Wp || !bias || !x || !sample || !t || (!noise_in && !seed)) return LFVDM_E_SHAPE;
    if (int rc = lfvdm_conv_out_psample_ok(B * T, H, W, C, Cout)) return rc;
    const HeadUpdate p = {act, Wp, bias, x, noise_in, eps_out, noise_out, sample, pred_xstart, mean_out, t, seed, sqrt_recip_acp,
                          sqrt_recipm1_acp, coef1, coef2, log_var, B * T, T, H, W, C, clip};
    const long quads = (long)B * T * H * (W / 4);
    const dim3 grid((unsigned)((quads + 3) / 4));
    const size_t lds = (size_t)Cout * 9 * C * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define LFVDM_HEAD(CO, CPL) hipLaunchKernelGGL((conv_out_psample_kernel<CO, CPL>), grid, dim3(256), lds, s, p)
    if (Cout == 4) {
        if (C == 64) LFVDM_HEAD(4, 1); else if (C == 128) LFVDM_HEAD(4, 2); else LFVDM_HEAD(4, 4);
    } else {
        if (C == 64) LFVDM_HEAD(3, 1); else if (C == 128) LFVDM_HEAD(3, 2); else LFVDM_HEAD(3, 4);
    }
#undef LFVDM_HEAD
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_masked_mse(const float* a, const float* b, const float* mask, float* out, int B, int T,
                                int frame_inner, void* stream) {
    if (B <= 0 || T <= 0 || frame_inner <= 0) return LFVDM_E_SHAPE;
    hipLaunchKernelGGL(masked_mse_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, a, b, mask, out, T, frame_inner);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}
