// Device-side semaphores between a replayed hipGraph and another stream (see lfvdm_hip.h).
//
// Why kernels and not events: an event recorded inside a captured region only orders work INSIDE the graph; the
// external event-record node that would do this job (hipEventRecordExternal) is refused under capture by the HIP runtime
// that ships with PyTorch-ROCm 2.10 (and by torch itself: "External events are disallowed in rocm").  A counter in device
// memory needs nothing from the runtime: lfvdm_flag_add is an ordinary kernel node of the graph, lfvdm_flag_wait an
// ordinary kernel on the waiting stream.
#include "common_hip.h"

namespace {

__global__ void flag_add_kernel(int* flag) {
    // Everything the stream ran before this node has completed (kernel boundary = agent-scope release of its stores):
    // the increment publishes "bucket complete".
    if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void flag_wait_kernel(const int* flag, int target, long long timeout_ticks, int* timed_out, int* timed_out_f32) {
    // One lane polls with device-scope loads (served by the memory side, never by this CU's L1) and sleeps between polls:
    // a parked wave, no measurable bandwidth.  The exit condition is always reached: the counter arrives, or the
    // 100 MHz wall clock passes the deadline.  Then `timed_out` is raised: the optimizer launch of the step reads the
    // word (lfvdm_adamw_args.skip_flag) and leaves parameters, moments and EMA untouched, and the host fails the step
    // (TrainLoop polls the word every step through pinned memory, one step late, without stalling).
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        // wrap-safe: both sides are plain wrapping 32-bit counters, the difference is taken modulo 2^32 and read as signed
        while ((int)((unsigned)__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)target) < 0) {
            __builtin_amdgcn_s_sleep(64);
            if (wall_clock64() - t0 > timeout_ticks) {
                atomicExch(timed_out, 1);
                // second word (optional): 1.0f, an element of the gradient arena that rides in the last bucket's SUM
                // all-reduce - any rank's timeout reaches every rank's optimizer launch without a collective of its own
                if (timed_out_f32) atomicExch(timed_out_f32, 0x3f800000);
                break;
            }
        }
        // The int word is STICKY (cleared only by the host's reset): while it is up, EVERY wait raises the riding word again -
        // the gradient arena's copy is cleared by zero_grad each step, and a rank whose own optimizer launch keeps skipping
        // on the sticky word must make its peers skip with it (a one-off timeout would otherwise leave the replicas one
        // update apart until the host has examined the word, one step late).
        if (timed_out_f32 && __hip_atomic_load(timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
            atomicExch(timed_out_f32, 0x3f800000);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

}  // namespace

extern "C" int lfvdm_flag_add(int32_t* flag, void* stream) {
    if (!flag) return LFVDM_E_SHAPE;
    hipLaunchKernelGGL(flag_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flag);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_flag_wait2(const int32_t* flag, int32_t target, double timeout_s, int32_t* timed_out, float* timed_out_f32,
                                void* stream) {
    if (!flag || !timed_out || !(timeout_s > 0)) return LFVDM_E_SHAPE;
    const long long ticks = (long long)(timeout_s * 1.0e8);      // wall_clock64 runs at 100 MHz on gfx950
    hipLaunchKernelGGL(flag_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, target, ticks, timed_out,
                       (int*)timed_out_f32);
    LFVDM_CHECK_LAUNCH();
    return LFVDM_OK;
}

extern "C" int lfvdm_flag_wait(const int32_t* flag, int32_t target, double timeout_s, int32_t* timed_out, void* stream) {
    return lfvdm_flag_wait2(flag, target, timeout_s, timed_out, nullptr, stream);
}
