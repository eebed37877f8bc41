// Ordered reduction of the deterministic-accumulation slabs (common_hip.h: DetSlab).
#include "common_hip.h"

namespace {

__global__ __launch_bounds__(256) void det_reduce_kernel(float* __restrict__ dst, const float* __restrict__ slab, long n, long parts) {
    // a thread owns 4 consecutive elements (n is padded by the tail loop below); the parts are added in index order,
    // four loads in flight
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 t = ld4(dst + 4 * i);
        long p = 0;
        for (; p + 4 <= parts; p += 4) {
            const f32x4 a = ld4(slab + (size_t)p * n + 4 * i), b = ld4(slab + (size_t)(p + 1) * n + 4 * i);
            const f32x4 c = ld4(slab + (size_t)(p + 2) * n + 4 * i), d = ld4(slab + (size_t)(p + 3) * n + 4 * i);
            t += a; t += b; t += c; t += d;
        }
        for (; p < parts; ++p) t += ld4(slab + (size_t)p * n + 4 * i);
        st4(dst + 4 * i, t);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = n4 * 4 + threadIdx.x;
        float t = dst[i];
        for (long p = 0; p < parts; ++p) t += slab[(size_t)p * n + i];
        dst[i] = t;
    }
}

__global__ __launch_bounds__(256) void det_reduce_unaligned_kernel(float* __restrict__ dst, const float* __restrict__ slab, long n, long parts) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float t = dst[i];
        for (long p = 0; p < parts; ++p) t += slab[(size_t)p * n + i];
        dst[i] = t;
    }
}

// two destinations with the same number of parts in ONE launch (a weight gradient and its bias gradient, dgamma and
// dbeta): blockIdx.y selects the pair; float4 body when that pair is 16-byte aligned, scalar otherwise
__global__ __launch_bounds__(256) void det_reduce2_kernel(float* __restrict__ d0, const float* __restrict__ s0, long n0,
                                                           float* __restrict__ d1, const float* __restrict__ s1, long n1,
                                                           long parts) {
    float* dst = blockIdx.y ? d1 : d0;
    const float* slab = blockIdx.y ? s1 : s0;
    const long n = blockIdx.y ? n1 : n0;
    const bool aligned = (n % 4 == 0) && ((uintptr_t)dst % 16 == 0) && ((uintptr_t)slab % 16 == 0);
    if (aligned) {
        const long n4 = n >> 2;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
            f32x4 t = ld4(dst + 4 * i);
            long p = 0;
            for (; p + 4 <= parts; p += 4) {
                const f32x4 a = ld4(slab + (size_t)p * n + 4 * i), b = ld4(slab + (size_t)(p + 1) * n + 4 * i);
                const f32x4 c = ld4(slab + (size_t)(p + 2) * n + 4 * i), d = ld4(slab + (size_t)(p + 3) * n + 4 * i);
                t += a; t += b; t += c; t += d;
            }
            for (; p < parts; ++p) t += ld4(slab + (size_t)p * n + 4 * i);
            st4(dst + 4 * i, t);
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
            float t = dst[i];
            for (long p = 0; p < parts; ++p) t += slab[(size_t)p * n + i];
            dst[i] = t;
        }
    }
}

}  // namespace

int lfvdm_det_reduce2_launch(float* d0, const float* s0, long n0, float* d1, const float* s1, long n1, long parts, hipStream_t s) {
    if (!d0 || !s0 || n0 <= 0 || !d1 || !s1 || n1 <= 0 || parts <= 0) return LFVDM_E_SHAPE;
    const long nmax = n0 > n1 ? n0 : n1;
    long blocks = (nmax / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(det_reduce2_kernel, dim3((unsigned)blocks, 2), dim3(256), 0, s, d0, s0, n0, d1, s1, n1, parts);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

int lfvdm_det_reduce_launch(float* dst, const float* slab, long n, long parts, hipStream_t s) {
    if (!dst || !slab || n <= 0 || parts <= 0) return LFVDM_E_SHAPE;
    const bool aligned = (n % 4 == 0) && ((uintptr_t)dst % 16 == 0) && ((uintptr_t)slab % 16 == 0);
    long blocks = ((aligned ? n / 4 : n) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (aligned) hipLaunchKernelGGL(det_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, slab, n, parts);
    else hipLaunchKernelGGL(det_reduce_unaligned_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dst, slab, n, parts);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_det_reduce(float* dst, const float* slab, long n, long parts, void* stream) {
    return lfvdm_det_reduce_launch(dst, slab, n, parts, (hipStream_t)stream);
}
