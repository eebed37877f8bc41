// Shared device helpers for the gfx950 kernels (wave = 64 lanes, fp32 MFMA, LDS tiles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

#include "lfvdm_hip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define LFVDM_CHECK_LAUNCH()                                   \
    do {                                                       \
        if (hipGetLastError() != hipSuccess) return LFVDM_E_LAUNCH; \
    } while (0)

// Host-side record of the dynamic-LDS limit raised for ONE kernel instance, per device (hipFuncSetAttribute acts on
// the current device's copy of the function).  One object per launcher template instance; safe to call from several
// host threads: the fast path is an atomic load, the slow path (first launch per device, or a larger request)
// serialises on a mutex so that the recorded limit and the attribute cannot disagree.
#define LFVDM_MAX_DEVICES 16
struct DynLdsLimit {
    std::atomic<uint32_t> bytes[LFVDM_MAX_DEVICES];
    std::mutex mu;
    DynLdsLimit() {
        for (auto& b : bytes) b.store(0, std::memory_order_relaxed);
    }
    int ensure(const void* fn, size_t need) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LFVDM_MAX_DEVICES) return LFVDM_E_LAUNCH;
        if (need <= bytes[dev].load(std::memory_order_acquire)) return LFVDM_OK;
        std::lock_guard<std::mutex> lock(mu);
        if (need <= bytes[dev].load(std::memory_order_acquire)) return LFVDM_OK;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need) != hipSuccess) return LFVDM_E_LAUNCH;
        bytes[dev].store((uint32_t)need, std::memory_order_release);
        return LFVDM_OK;
    }
};

__device__ __forceinline__ float silu_f(float v) {
    // x * sigmoid(x); v_exp_f32 + v_rcp_f32 (each <= 1 ulp)
    return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
}

// floor(a / d) for 0 <= a < 2^21, d >= 1, rd = v_rcp_f32(d): the truncated product is off by at most one and the
// fix-up repairs it (a generic 32-bit division is ~30 instructions; the normalisation kernels did eight per thread to
// find the group of a channel)
__device__ __forceinline__ int fdiv_small(int a, int d, float rd) {
    int q = (int)((float)a * rd);
    const int r = a - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// Orders this wave's LDS stores before its later LDS loads for the COMPILER only: LDS operations
// of one wave are executed in issue order by the hardware, so a tile staged and consumed by the
// same wave needs no s_barrier (wave-private staging, see conv_igemm.hip).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- Deterministic accumulation (LFVDM_DETERMINISTIC=1, lfvdm_hip.h).  Partial sums of different workgroups are
// normally combined with float atomics, whose order - hence the rounding - changes from run to run.  With a slab the
// partials go to slab[part][n] by plain stores (each (part, element) at most once; the slab is zero-filled first
// where a part does not cover every element) and ONE launch adds the parts in index order: dst[i] += sum_p slab[p][i].
// Row selector of the temporal attention's R tables (sampler plans): slice index of batch row b =
// (ring ? p[b] % ring : p[b]); p == nullptr: no tables, slice b.  `ring` > 0: the tables hold a rolling window of `ring`
// timesteps, timestep t in slot t % ring (Plan.build_R_tables refills the window every `ring` steps).
struct RSel {
    const int64_t* p;
    int ring;
    __host__ __device__ size_t slice(int b, int B) const {
        if (!p) return (size_t)b;
        const long t = (long)p[b];
        return (size_t)(ring ? t % ring : t) * B + b;
    }
};

struct DetSlab {
    float* slab;           // nullptr: float atomics
    const float* base;     // first element of the destination array
    long n;                // elements of the destination array = row length of the slab
};
__device__ __forceinline__ void det_add(const DetSlab& d, long part, float* dst, float v) {
    if (d.slab) d.slab[(size_t)part * d.n + (size_t)(dst - d.base)] = v;
    else atomicAdd(dst, v);
}
// dst[i] += slab[0][i] + slab[1][i] + ... (fixed order), i < n  (det_reduce.hip)
int lfvdm_det_reduce_launch(float* dst, const float* slab, long n, long parts, hipStream_t s);
int lfvdm_det_reduce2_launch(float* d0, const float* s0, long n0, float* d1, const float* s1, long n1, long parts, hipStream_t s);
