// common.hpp — status/error plumbing shared by the host side of libvi_amd.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/vi_amd.h"

namespace vi {

// Thread-local last-error message (vi_last_error()).
std::string &last_error_ref();

inline vi_status fail(vi_status st, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
  return st;
}

#define VI_HIP(expr)                                                                        \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      return ::vi::fail(VI_ERR_DEVICE, "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), \
                        __FILE__, __LINE__, #expr);                                         \
  } while (0)

#define VI_TRY(expr)                  \
  do {                                \
    vi_status _s = (expr);            \
    if (_s != VI_OK) return _s;       \
  } while (0)

// RAII device buffer (hipMalloc/hipFree).  288 GB of HBM per GPU: no pooling games needed,
// buffers are allocated once per index / grown-only per workspace.
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  // grow-only
  vi_status reserve(size_t count) {
    if (count <= n && p) return VI_OK;
    release();
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
    if (e != hipSuccess) {
      p = nullptr;
      return fail(VI_ERR_DEVICE, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    }
    n = count;
    return VI_OK;
  }
};

inline uint32_t ceil_div_u32(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

}  // namespace vi
