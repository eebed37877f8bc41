// Does independent VALU work issue in the shadow of dependent MFMAs of the SAME wave on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, long long* cyc) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
    long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; ++n) v[n & 7] = __builtin_fmaf(v[n & 7], b, a);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NV> void run(const char* name) {
    float* out; long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    hipLaunchKernelGGL(k<NV>, dim3(256), dim3(256), 0, 0, out, 1.0f, 0.5f, cyc);
    hipLaunchKernelGGL(k<NV>, dim3(256), dim3(256), 0, 0, out, 1.0f, 0.5f, cyc);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s: %lld cycles for 1024 MFMAs -> %.1f cycles per MFMA slot\n", name, h, h / 1024.0);
}
int main() { run<0>("0 VALU"); run<4>("4 VALU"); run<8>("8 VALU"); run<12>("12 VALU"); run<16>("16 VALU"); run<24>("24 VALU"); return 0; }
