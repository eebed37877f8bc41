// Probe: buffer_load_dwordx4 ... lds on gfx950 - lane-linear LDS destination, out-of-range offsets write zeros.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* __restrict__ g, float* __restrict__ out, int nbytes) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* dst = smem + wave * 256;
    for (int i = threadIdx.x; i < 1024; i += 256) smem[i] = -7.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
    // odd lanes: their own float4, permuted (lane ^ 2); even lanes: out of range
    const int off = (lane & 1) ? (wave * 64 + (lane ^ 2)) * 16 : 0x7fffffff;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x4 v = *reinterpret_cast<const f32x4*>(smem + wave * 256 + lane * 4);
    *reinterpret_cast<f32x4*>(out + threadIdx.x * 4) = v;
}
int main() {
    const int n = 1024;
    std::vector<float> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i + 1.f;
    float *dg, *dout;
    hipMalloc(&dg, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dg, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, dg, dout, n * 4);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        const int lane = t & 63, wave = t >> 6;
        for (int c = 0; c < 4; ++c) {
            const float want = (lane & 1) ? h[(wave * 64 + (lane ^ 2)) * 4 + c] : 0.f;
            if (o[t * 4 + c] != want) { if (bad < 8) printf("t=%d c=%d got %f want %f\n", t, c, o[t * 4 + c], want); ++bad; }
        }
    }
    printf("glds probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
