// Fused AdamW + EMA + gradient-norm over one flat fp32 parameter arena (HBM-bound, float4 streams).
// Replaces per-tensor `opt.step()` (torch AdamW), `update_ema` (nn.py:55-65: 2 launches per tensor and
// rate) and `_log_grad_norm` (train_util.py:353-357: one `.item()` host sync per parameter tensor) of the
// reference's optimize_normal (train_util.py:346-351) with a single launch and no host synchronisation.
#include "common_hip.h"

namespace {

__global__ __launch_bounds__(256) void adamw_ema_kernel(const lfvdm_adamw_args a) {
    // a gradient bucket was reduced before the backward pass had finished writing it (lfvdm_flag_wait gave up): the
    // gradients of this step are garbage - touch nothing, the host raises
    if (a.skip_flag && __hip_atomic_load(a.skip_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    const int64_t n4 = a.n / 4;
    float sq = 0.f;
    const float step = a.lr / a.bias_corr1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 g = ld4(a.g + i * 4) * a.grad_scale;
        f32x4 p = ld4(a.p + i * 4);
        f32x4 m = ld4(a.m + i * 4), v = ld4(a.v + i * 4);
        sq += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
        p = p * (1.0f - a.lr * a.weight_decay);            // decoupled weight decay (torch AdamW)
        m = m * a.beta1 + g * (1.0f - a.beta1);
        v = v * a.beta2 + (g * g) * (1.0f - a.beta2);
        f32x4 d;
        d.x = sqrtf(v.x) / a.bias_corr2_sqrt + a.eps; d.y = sqrtf(v.y) / a.bias_corr2_sqrt + a.eps;
        d.z = sqrtf(v.z) / a.bias_corr2_sqrt + a.eps; d.w = sqrtf(v.w) / a.bias_corr2_sqrt + a.eps;
        p.x -= step * (m.x / d.x); p.y -= step * (m.y / d.y); p.z -= step * (m.z / d.z); p.w -= step * (m.w / d.w);
        st4(a.p + i * 4, p); st4(a.m + i * 4, m); st4(a.v + i * 4, v);
        for (int e = 0; e < a.n_ema; ++e) {
            const f32x4 t = ld4(a.ema[e] + i * 4);
            st4(a.ema[e] + i * 4, t * a.ema_rate[e] + p * (1.0f - a.ema_rate[e]));   // targ*r + src*(1-r)
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
    hipLaunchKernelGGL(adamw_ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}
