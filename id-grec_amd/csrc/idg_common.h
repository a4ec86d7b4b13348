// Shared internals of libidgrec.so: error reporting and HIP call checking.
#pragma once
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>

#include "idgrec.h"

namespace idg {

// Thread-local "last error" string behind idg_last_error().
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
const char* get_error();

inline int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  set_error("%s", buf);
  return code;
}

// A row bitmap is about to be rewritten by a library call: live-unit lists registered for it (idg_graph_live_units)
// describe its old contents and are dropped (idg_graph.hip).  bytes: extent of the write when the caller knows it (lists
// registered for a sub-range starting inside it go too); 0 = lists registered at `bitmap` itself.
void rows_changed(const void* bitmap, size_t bytes = 0);

// Library-internal forms of two entry points, for idg_step.cpp's one-call step (fewer launches per step on the host):
// idg_bpr_plan_rows_f32 whose first kernel also zeroes two words (`zero2`, nullable: the header of the unit list about to
// be built) and which may sort up to 4096 pairs with ONE workgroup (`one_launch_sort`: a launch less where the side stream
// has slack); idg_graph_live_units without its own clearing of that header.
int bpr_plan_rows(const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t B, int64_t num_users, int64_t n, void* ws,
                  uint32_t* bitmap, uint32_t* zero2, bool one_launch_sort, void* stream);
int live_units_prezeroed(const idg_graph* g, const uint32_t* bitmap, void* units_ws, int64_t max_rows, void* stream);

}  // namespace idg

#define IDG_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return idg::fail(IDG_E_INVALID, __VA_ARGS__);   \
  } while (0)

#define IDG_HIP(call)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return idg::fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? IDG_E_NODEVICE \
                                                                             : IDG_E_HIP,  \
                       "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,    \
                       __LINE__);                                                          \
  } while (0)
