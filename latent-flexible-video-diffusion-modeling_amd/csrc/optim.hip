// Fused AdamW + EMA + gradient-norm over one flat fp32 parameter arena (HBM-bound, float4 streams).
// Replaces per-tensor `opt.step()` (torch AdamW), `update_ema` (nn.py:55-65: 2 launches per tensor and
// rate) and `_log_grad_norm` (train_util.py:353-357: one `.item()` host sync per parameter tensor) of the
// reference's optimize_normal (train_util.py:346-351) with a single launch and no host synchronisation.
#include "common_hip.h"

namespace {

// one float4 of the update; returns the new parameter quad and the squared gradient norm contribution
__device__ __forceinline__ f32x4 adamw_quad(const lfvdm_adamw_args& a, float step, f32x4 g, f32x4 p, f32x4& m, f32x4& v, float& sq) {
    g = g * a.grad_scale;
    sq += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
    p = p * (1.0f - a.lr * a.weight_decay);            // decoupled weight decay (torch AdamW)
    m = m * a.beta1 + g * (1.0f - a.beta1);
    v = v * a.beta2 + (g * g) * (1.0f - a.beta2);
    f32x4 d;
    d.x = sqrtf(v.x) / a.bias_corr2_sqrt + a.eps; d.y = sqrtf(v.y) / a.bias_corr2_sqrt + a.eps;
    d.z = sqrtf(v.z) / a.bias_corr2_sqrt + a.eps; d.w = sqrtf(v.w) / a.bias_corr2_sqrt + a.eps;
    p.x -= step * (m.x / d.x); p.y -= step * (m.y / d.y); p.z -= step * (m.z / d.z); p.w -= step * (m.w / d.w);
    return p;
}
__device__ __forceinline__ f32x4 ld4_nt(const float* q) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q)); }
__device__ __forceinline__ void st4_nt(float* q, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(q)); }

// NE = number of EMA copies (compile time: their loads join the batch), U float4 per thread and iteration.  Every load of
// an iteration is issued before the first use (round-3 lesson: a rolled loop of global loads is a chain of dependent round
// trips; here 2 * (4 + NE) 16-byte loads are in flight per thread), the streams that nothing re-reads soon - gradients in,
// moments and EMA copies out - are non-temporal (they would only push the parameters, which the next forward pass reads,
// out of the caches).  9 streams of 4 bytes per parameter: HBM-bound (hbm_phases.adamw_ema).
template <int NE>
__global__ __launch_bounds__(256) void adamw_ema_kernel(const lfvdm_adamw_args a) {
    // a gradient bucket was reduced before the backward pass had finished writing it (lfvdm_flag_wait gave up): the
    // gradients of this step are garbage - touch nothing, the host raises
    if (a.skip_flag && __hip_atomic_load(a.skip_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    // ... or on ANOTHER rank: the word that rode in the last bucket's SUM all-reduce (any non-zero bit pattern)
    if (a.skip_flag2 && __hip_atomic_load(a.skip_flag2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    constexpr int U = 2;
    const int64_t n4 = a.n / 4;
    float sq = 0.f;
    const float step = a.lr / a.bias_corr1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += U * stride) {
        int64_t idx[U];
        f32x4 g[U], p[U], m[U], v[U], e[U][NE > 0 ? NE : 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            idx[u] = i < n4 ? i : -1;
            const int64_t j = (i < n4 ? i : i0) * 4;          // (a clamped duplicate load; its result is not used)
            g[u] = ld4_nt(a.g + j);
            p[u] = ld4(a.p + j);
            m[u] = ld4_nt(a.m + j);
            v[u] = ld4_nt(a.v + j);
#pragma unroll
            for (int k = 0; k < NE; ++k) e[u][k] = ld4_nt(a.ema[k] + j);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (idx[u] < 0) continue;
            const int64_t j = idx[u] * 4;
            const f32x4 pn = adamw_quad(a, step, g[u], p[u], m[u], v[u], sq);
            st4(a.p + j, pn);
            st4_nt(a.m + j, m[u]);
            st4_nt(a.v + j, v[u]);
#pragma unroll
            for (int k = 0; k < NE; ++k) st4_nt(a.ema[k] + j, e[u][k] * a.ema_rate[k] + pn * (1.0f - a.ema_rate[k]));   // targ*r + src*(1-r)
        }
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const int64_t i = n4 * 4 + threadIdx.x;
        const float g = a.g[i] * a.grad_scale;
        sq += g * g;
        float p = a.p[i] * (1.0f - a.lr * a.weight_decay);
        const float m = a.m[i] * a.beta1 + g * (1.0f - a.beta1);
        const float v = a.v[i] * a.beta2 + g * g * (1.0f - a.beta2);
        p -= step * (m / (sqrtf(v) / a.bias_corr2_sqrt + a.eps));
        a.p[i] = p; a.m[i] = m; a.v[i] = v;
        for (int e = 0; e < a.n_ema; ++e) a.ema[e][i] = a.ema[e][i] * a.ema_rate[e] + p * (1.0f - a.ema_rate[e]);
    }
    if (a.grad_sqsum) {
        sq = wave_sum(sq);
        __shared__ float red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(a.grad_sqsum, red[0] + red[1] + red[2] + red[3]);
    }
}

}  // namespace

extern "C" int lfvdm_adamw_ema(const lfvdm_adamw_args* a, void* stream) {
    if (a->n <= 0 || a->n_ema < 0 || a->n_ema > 4) return LFVDM_E_SHAPE;
    int64_t blocks = (a->n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    switch (a->n_ema) {
        case 0: hipLaunchKernelGGL(adamw_ema_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a); break;
        case 1: hipLaunchKernelGGL(adamw_ema_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a); break;
        case 2: hipLaunchKernelGGL(adamw_ema_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a); break;
        case 3: hipLaunchKernelGGL(adamw_ema_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a); break;
        default: hipLaunchKernelGGL(adamw_ema_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a); break;
    }
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}
