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
