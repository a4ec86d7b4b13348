// Device-local events for ordering the library's own streams (the step's stream and the stream that prepares the
// next batch).  A default HIP event performs a SYSTEM-scope fence when it is recorded — cache write-back and
// invalidation that make device memory visible to the host and to other devices — which nobody needs between two
// streams of one device and which costs the recording stream ~7 us per step (measured as a gap between the last kernel
// of one step and the first of the next).  These events are created with hipEventDisableTiming |
// hipEventDisableSystemFence: kernel boundaries already order memory at device scope.
#include <hip/hip_runtime.h>

#include "idg_common.h"

extern "C" {

int idg_event_create(void** out) {
  IDG_REQUIRE(out, "idg_event_create: NULL argument");
  hipEvent_t ev = nullptr;
  IDG_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence));
  *out = ev;
  return IDG_OK;
}

int idg_event_destroy(void* event) {
  if (event) IDG_HIP(hipEventDestroy((hipEvent_t)event));
  return IDG_OK;
}

int idg_event_record(void* event, void* stream) {
  IDG_REQUIRE(event, "idg_event_record: NULL event");
  IDG_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return IDG_OK;
}

int idg_stream_wait_event(void* stream, void* event) {
  IDG_REQUIRE(event, "idg_stream_wait_event: NULL event");
  IDG_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return IDG_OK;
}

int idg_event_synchronize(void* event) {
  IDG_REQUIRE(event, "idg_event_synchronize: NULL event");
  IDG_HIP(hipEventSynchronize((hipEvent_t)event));
  return IDG_OK;
}

int idg_event_query(void* event, int* done) {
  IDG_REQUIRE(event && done, "idg_event_query: NULL argument");
  const hipError_t e = hipEventQuery((hipEvent_t)event);
  if (e == hipSuccess) {
    *done = 1;
  } else if (e == hipErrorNotReady) {
    *done = 0;
    (void)hipGetLastError();  // not an error: clear the sticky status
  } else {
    return idg::fail(IDG_E_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
  }
  return IDG_OK;
}

}  // extern "C"
