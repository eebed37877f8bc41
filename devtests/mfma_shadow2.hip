// Do MFMAs of one wave overlap with VALU of ANOTHER wave on the same SIMD (gfx950)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// mode bit0: waves 0-3 run the MFMA loop; bit1: waves 4-7 run a VALU loop; bit2: waves 4-7 run the MFMA loop too
__global__ __launch_bounds__(512) void k(float* out, float a, float b, long long* cyc, int mode) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    long long t0 = clock64();
    const bool do_mfma = (wave < 4 && (mode & 1)) || (wave >= 4 && (mode & 4));
    const bool do_valu = wave >= 4 && (mode & 2);
    if (do_mfma) {
#pragma unroll 1
        for (int it = 0; it < 64; ++it)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (do_valu) {
#pragma unroll 1
        for (int it = 0; it < 64 * 16; ++it)
#pragma unroll
            for (int n = 0; n < 16; ++n) v[n & 7] = __builtin_fmaf(v[n & 7], b, a);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;     // forces completion
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
void run(const char* name, int mode) {
    float* out; long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 1.0f, 0.5f, cyc, mode);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 1.0f, 0.5f, cyc, mode);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-46s: %lld cycles\n", name, h);
}
int main() {
    run("MFMA in waves 0-3 only (1024 MFMAs)", 1);
    run("VALU in waves 4-7 only (16384 FMAs)", 2);
    run("MFMA waves 0-3 + VALU waves 4-7", 3);
    run("MFMA in all 8 waves (2 per SIMD)", 5);
    return 0;
}
