// Events usable across a captured graph's boundary (see lfvdm_hip.h).  Host code only.
#include "common_hip.h"

extern "C" int lfvdm_event_create(void** event) {
    if (!event) return LFVDM_E_SHAPE;
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return LFVDM_E_LAUNCH;
    *event = (void*)ev;
    return LFVDM_OK;
}

extern "C" int lfvdm_event_destroy(void* event) {
    if (!event) return LFVDM_OK;
    return hipEventDestroy((hipEvent_t)event) == hipSuccess ? LFVDM_OK : LFVDM_E_LAUNCH;
}

extern "C" int lfvdm_event_record(void* event, void* stream) {
    if (!event) return LFVDM_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) return LFVDM_E_LAUNCH;
    const unsigned flags = st == hipStreamCaptureStatusActive ? hipEventRecordExternal : hipEventRecordDefault;
    return hipEventRecordWithFlags((hipEvent_t)event, s, flags) == hipSuccess ? LFVDM_OK : LFVDM_E_LAUNCH;
}

extern "C" int lfvdm_stream_wait_event(void* stream, void* event) {
    if (!event) return LFVDM_E_SHAPE;
    return hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0) == hipSuccess ? LFVDM_OK : LFVDM_E_LAUNCH;
}
