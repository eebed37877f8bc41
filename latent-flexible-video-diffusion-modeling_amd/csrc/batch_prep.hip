// Training-batch assembly on the device (reference train_util.py:224-241, prepare_training_batch): the host samples a
// small index table - which frames of which video make up the <= max_frames frames of each batch element, which of
// them are observed / latent / padding - and this kernel gathers the frames and writes the mask and index tensors
// the U-Net consumes, inside the captured training step.  HBM-bound gather: one workgroup per (frame slot, video), float4.
#include "common_hip.h"

namespace {

__global__ __launch_bounds__(256) void prepare_batch_kernel(const float* __restrict__ pool, const int32_t* __restrict__ table,
                                                            float* __restrict__ batch, int64_t* __restrict__ frame_indices,
                                                            float* __restrict__ obs_mask, float* __restrict__ latent_mask,
                                                            int F, int Tp, int frame_elems) {
    const int f = blockIdx.x, b = blockIdx.y;
    const int32_t* e = table + ((size_t)b * F + f) * 4;
    int row = e[0];
    row = row < 0 ? 0 : (row >= Tp ? Tp - 1 : row);          // a corrupt table must not read outside the pool
    const float* src = pool + ((size_t)b * Tp + row) * frame_elems;
    float* dst = batch + ((size_t)b * F + f) * frame_elems;
    const int n4 = frame_elems >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) st4(dst + 4 * i, ld4(src + 4 * i));
    for (int i = (n4 << 2) + threadIdx.x; i < frame_elems; i += 256) dst[i] = src[i];
    if (threadIdx.x == 0) {
        frame_indices[(size_t)b * F + f] = (int64_t)e[1];
        obs_mask[(size_t)b * F + f] = e[2] ? 1.f : 0.f;
        latent_mask[(size_t)b * F + f] = e[3] ? 1.f : 0.f;
    }
}

// Input compositing of the training forward (reference unet.py:441-450) as channels-last rows for the first 3x3 conv:
//   rows[(n, pixel)][c] = x[n][c][pixel] * (1 - obs[n]) + x0[n][c][pixel] * obs[n]   (c < Cx)
//   rows[(n, pixel)][Cx] = obs[n]  (the indicator channel),  zero up to the row width ld (the GEMM's 32-channel chunk).
// The inference path has this inside lfvdm_conv_in; training keeps the rows (operand of the weight gradient).
__global__ __launch_bounds__(256) void compose_rows_kernel(const float* __restrict__ x, const float* __restrict__ x0,
                                                           const float* __restrict__ obs, float* __restrict__ rows, int Cx,
                                                           int HW, int ld, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;           // one thread per (n, pixel, 4 channels)
    if (i >= total) return;
    const int q = ld >> 2;
    const long pos = i / q;
    const int c0 = (int)(i - pos * q) * 4;
    const long n = pos / HW;
    const int pix = (int)(pos - n * HW);
    const float o = obs[n];
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + k;
        float t = 0.f;
        if (c < Cx) {
            const size_t at = ((size_t)n * Cx + c) * HW + pix;
            t = x[at] * (1.f - o) + x0[at] * o;
        } else if (c == Cx) {
            t = o;
        }
        v[k] = t;
    }
    st4(rows + pos * ld + c0, (f32x4){v[0], v[1], v[2], v[3]});
}

}  // namespace

extern "C" int lfvdm_compose_rows(const float* x, const float* x0, const float* obs, float* rows, int N, int Cx, int H, int W,
                                  int ld, void* stream) {
    if (!x || !x0 || !obs || !rows || N <= 0 || Cx <= 0 || H <= 0 || W <= 0 || ld < Cx + 1 || (ld & 3)) return LFVDM_E_SHAPE;
    const long total = (long)N * H * W * (ld >> 2);
    if (total >= (1L << 31) * 256) return LFVDM_E_UNSUPPORTED;
    hipLaunchKernelGGL(compose_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, x0, obs,
                       rows, Cx, H * W, ld, total);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_prepare_batch(const float* pool, const int32_t* table, float* batch, int64_t* frame_indices,
                                   float* obs_mask, float* latent_mask, int B, int F, int Tp, int frame_elems, void* stream) {
    if (!pool || !table || !batch || !frame_indices || !obs_mask || !latent_mask) return LFVDM_E_SHAPE;
    if (B <= 0 || F <= 0 || Tp <= 0 || frame_elems <= 0 || B > 65535) return LFVDM_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(pool) | reinterpret_cast<uintptr_t>(batch)) & 15) return LFVDM_E_SHAPE;
    if (frame_elems & 3) return LFVDM_E_SHAPE;             // frames are whole float4 rows (C*H*W of the latents / images)
    hipLaunchKernelGGL(prepare_batch_kernel, dim3(F, B), dim3(256), 0, (hipStream_t)stream, pool, table, batch, frame_indices,
                       obs_mask, latent_mask, F, Tp, frame_elems);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}
