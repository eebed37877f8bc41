// Micro-benchmark behind DESIGN.md section 5 "persistent level chain": what does a device-wide dependency between two
// stages of ONE launch cost on MI355X when it is built WITHOUT fences?
//
//   (A) barrier:   G co-resident workgroups (one per CU), each iteration = one relaxed agent-scope add by lane 0,
//                  an sc1-load poll with s_sleep until the counter reaches the iteration's target, a workgroup barrier.
//   (B) fan-in:    only `owners` workgroups arrive (the output-tile owners of a split-K stage), all G wait.
//   (C) hand-off:  (A) plus data: every workgroup publishes a 4 KiB tile with 16-byte sc1 (write-through) stores, drains
//                  vmcnt, arrives; after the barrier it reads ANOTHER workgroup's tile - with sc1 loads to registers or
//                  with sc1 LDS-DMA (buffer_load_dwordx4 ... lds), the staging form of the implicit GEMM - and checks
//                  EVERY word.  Buffers are reused every second iteration (consumer caches warm) and the workgroups
//                  burn a pseudo-random amount of time before publishing (uneven load): the conditions under which a
//                  stale L1 / L2 line would show (MI355X guide, "Test every hand-off under UNEVEN load").
//
// Every spin is bounded: a poller that has waited longer than `timeout_ticks` of the 100 MHz wall clock raises the
// abort word and leaves, and every poll also reads the abort word - all waves reach the end of the kernel.
//
// build: hipcc -O3 --offload-arch=gfx950 -o tools/_bin/grid_barrier_bench tools/grid_barrier_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                               \
        }                                                                                          \
    } while (0)

struct Ctl {
    int counter[32];      // [0]: barrier counter (own 128-byte line)
    int abort_[32];       // [0]: raised by a poller that timed out
    int mismatches[32];   // [0]: words that did not carry the expected value
};

__device__ __forceinline__ int ld_sc1(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// -> false: aborted (timeout here or elsewhere)
__device__ __forceinline__ bool wait_ge(const int* ctr, int target, const int* abort_w, long long timeout_ticks, int sleep) {
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    int spins = 0;
    while ((int)((unsigned)ld_sc1(ctr) - (unsigned)target) < 0) {
        if (sleep) __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63) == 0) {
            if (ld_sc1(abort_w)) return false;
            if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
                __hip_atomic_store((int*)abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    return true;
}

// mode 0: flat counter, every workgroup arrives.  mode 1: only workgroups < owners arrive, all wait.
__global__ __launch_bounds__(256) void barrier_kernel(Ctl* c, int iters, int mode, int owners, int sleep, long long timeout_ticks,
                                                      long long* t_out) {
    __shared__ int s_ok;
    const int G = gridDim.x;
    const int arr = mode == 1 ? owners : G;
    long long t0 = 0;
    if (threadIdx.x == 0) t0 = (long long)__builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (threadIdx.x == 0) {
            bool ok = true;
            {
                if (mode == 0 || (int)blockIdx.x < owners)
                    __hip_atomic_fetch_add(&c->counter[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = wait_ge(&c->counter[0], (it + 1) * arr, &c->abort_[0], timeout_ticks, sleep);
            }
            s_ok = ok ? 1 : 0;
        }
        __syncthreads();
        if (!s_ok) return;          // workgroup-uniform
        __syncthreads();
    }
    if (threadIdx.x == 0) t_out[blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime() - t0;
}

// (D) point-to-point flags: no shared counter.  Workgroup w publishes flag[w] = it + 1 (one sc1 store, own 128-byte
// line) and waits for the flags of D other workgroups - D lanes of one wave poll D lines with ONE load instruction -
// the dependency pattern of a tiled stage whose work items need a few producer tiles each, not the whole grid.
__global__ __launch_bounds__(256) void p2p_kernel(Ctl* c, int* flags, int iters, int D, long long timeout_ticks, long long* t_out) {
    __shared__ int s_ok;
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    long long t0 = 0;
    if (tid == 0) t0 = (long long)__builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (tid < 64) {
            if (tid == 0) __hip_atomic_store(flags + wg * 32, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int* f = flags + ((wg + 1 + 13 * tid) % G) * 32;
            const long long w0 = (long long)__builtin_amdgcn_s_memrealtime();
            bool ok = true;
            int spins = 0;
            for (;;) {
                const int v = tid < D ? ld_sc1(f) : it + 1;
                if (__builtin_amdgcn_ballot_w64((int)((unsigned)v - (unsigned)(it + 1)) < 0) == 0) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 63) == 0) {
                    if (ld_sc1(&c->abort_[0]) || (long long)__builtin_amdgcn_s_memrealtime() - w0 > timeout_ticks) {
                        __hip_atomic_store(&c->abort_[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = false;
                        break;
                    }
                }
            }
            if (tid == 0) s_ok = ok ? 1 : 0;
        }
        __syncthreads();
        if (!s_ok) return;
        __syncthreads();
    }
    if (tid == 0) t_out[wg] = (long long)__builtin_amdgcn_s_memrealtime() - t0;
}

// Hand-off under uneven load.  tile words = 1024 (4 KiB); 256 threads x one 16-byte store / load each.
// read_mode 0: sc1 loads to registers; 1: sc1 LDS-DMA then ds_read; 2: PLAIN loads (negative control: expected stale)
__global__ __launch_bounds__(256) void handoff_kernel(Ctl* c, unsigned* tiles, int iters, int read_mode, int uneven, long long timeout_ticks,
                                                      long long* t_out) {
    __shared__ int s_ok;
    __shared__ __attribute__((aligned(16))) unsigned lds[1024];
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    long long t0 = 0;
    if (tid == 0) t0 = (long long)__builtin_amdgcn_s_memrealtime();
    int bad = 0;
    unsigned lcg = 1234567u + 977u * wg;
    for (int it = 0; it < iters; ++it) {
        if (uneven) {            // 0 .. ~3 us of sleep, different per (workgroup, iteration)
            lcg = lcg * 1664525u + 1013904223u;
            const int n = (lcg >> 24) & 63;
            for (int k = 0; k < n; ++k) __builtin_amdgcn_s_sleep(8);
        }
        unsigned* mine = tiles + ((size_t)(it & 1) * G + wg) * 1024;
        const unsigned base = (unsigned)it * 0x9e3779b1u + (unsigned)wg * 0x85ebca77u;
        u32x4 v = {base + 4u * tid, base + 4u * tid + 1u, base + 4u * tid + 2u, base + 4u * tid + 3u};
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, 4096, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, tid * 16, 0, 16 /* sc1 */);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(&c->counter[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ok = wait_ge(&c->counter[0], (it + 1) * G, &c->abort_[0], timeout_ticks, 1) ? 1 : 0;
        }
        __syncthreads();
        if (!s_ok) return;
        const int src = (wg + 37 + 8 * (it % 5)) % G;       // another workgroup, changing XCD pairings
        const unsigned* theirs = tiles + ((size_t)(it & 1) * G + src) * 1024;
        const unsigned eb = (unsigned)it * 0x9e3779b1u + (unsigned)src * 0x85ebca77u;
        u32x4 r;
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)theirs, 0, 4096, 0x00020000);
        if (read_mode == 0) {
            r = __builtin_amdgcn_raw_buffer_load_b128(rt, tid * 16, 0, 16 /* sc1 */);
        } else if (read_mode == 1) {
            // one 1 KiB piece per wave: wave-uniform LDS base + lane * 16, per-lane source offset
            const int wave = tid >> 6;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, tid * 16, 0, 0,
                                                     16 /* sc1 */);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            r = *reinterpret_cast<const u32x4*>(lds + tid * 4);
        } else {
            r = *reinterpret_cast<const u32x4*>(theirs + tid * 4);
        }
        bad += (r.x != eb + 4u * tid) + (r.y != eb + 4u * tid + 1u) + (r.z != eb + 4u * tid + 2u) + (r.w != eb + 4u * tid + 3u);
        __syncthreads();        // lds reuse
    }
    if (bad) atomicAdd(&c->mismatches[0], bad);
    if (tid == 0) t_out[wg] = (long long)__builtin_amdgcn_s_memrealtime() - t0;
}

static double median(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const long long timeout_ticks = 20 * 1000 * 100;        // 20 ms of the 100 MHz clock per wait
    Ctl* c;
    long long* t_out;
    unsigned* tiles;
    int* flags;
    CK(hipMalloc(&c, sizeof(Ctl)));
    CK(hipMalloc(&t_out, 1024 * sizeof(long long)));
    CK(hipMalloc(&tiles, 2ull * 256 * 4096));
    CK(hipMalloc(&flags, 256 * 128));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<long long> host(1024);
    printf("{\"iters\": %d, \"results\": [\n", iters);
    bool first = true;
    auto report = [&](const char* what, int G, int extra, float ms, int n_it, bool want_mis) {
        Ctl h;
        CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
        CK(hipMemcpy(host.data(), t_out, G * sizeof(long long), hipMemcpyDeviceToHost));
        std::vector<double> per;
        for (int i = 0; i < G; ++i) per.push_back(host[i] * 10.0 / n_it);       // ticks of 10 ns
        printf("%s  {\"what\": \"%s\", \"workgroups\": %d, \"arg\": %d, \"ns_per_iter_event\": %.1f, \"ns_per_iter_in_kernel_median\": %.1f, "
               "\"aborted\": %d",
               first ? "" : ",\n", what, G, extra, ms * 1e6 / n_it, median(per), h.abort_[0]);
        if (want_mis) printf(", \"mismatched_words\": %d", h.mismatches[0]);
        printf("}");
        first = false;
        fflush(stdout);
    };
    for (int rep = 0; rep < 2; ++rep) {         // rep 0 warms up
        for (int G : {64, 128, 256}) {
            for (int sleep = 0; sleep <= 1; ++sleep) {
                CK(hipMemset(c, 0, sizeof(Ctl)));
                CK(hipMemset(t_out, 0, 1024 * sizeof(long long)));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(barrier_kernel, dim3(G), dim3(256), 0, 0, c, iters, 0, 0, sleep, timeout_ticks, t_out);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) report(sleep ? "barrier_flat_sleep" : "barrier_flat_spin", G, sleep, ms, iters, false);
            }
            for (int D : {1, 4, 16}) {
                CK(hipMemset(c, 0, sizeof(Ctl)));
                CK(hipMemset(flags, 0, 256 * 128));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(p2p_kernel, dim3(G), dim3(256), 0, 0, c, flags, iters, D, timeout_ticks, t_out);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) report("p2p_flags", G, D, ms, iters, false);
            }
            for (int owners : {20, 80}) {
                CK(hipMemset(c, 0, sizeof(Ctl)));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(barrier_kernel, dim3(G), dim3(256), 0, 0, c, iters, 1, std::min(owners, G), 1, timeout_ticks, t_out);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) report("fanin_owners_only", G, std::min(owners, G), ms, iters, false);
            }
        }
        for (int G : {128, 256}) {
            for (int read_mode = 0; read_mode <= 2; ++read_mode) {
                for (int uneven = 0; uneven <= 1; ++uneven) {
                    CK(hipMemset(c, 0, sizeof(Ctl)));
                    CK(hipMemset(tiles, 0xff, 2ull * 256 * 4096));
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(handoff_kernel, dim3(G), dim3(256), 0, 0, c, tiles, iters, read_mode, uneven, timeout_ticks, t_out);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    const char* nm[3] = {"handoff_sc1_store_sc1_load", "handoff_sc1_store_sc1_lds_dma", "handoff_sc1_store_PLAIN_load_negative_control"};
                    if (rep) report(nm[read_mode], G, uneven, ms, iters, true);
                }
            }
        }
    }
    printf("\n]}\n");
    return 0;
}
