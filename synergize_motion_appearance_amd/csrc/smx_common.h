// shared host-side helpers for the libsmx translation units
#pragma once
#include <hip/hip_runtime.h>
#include "smx.h"

#define SMX_HIP(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) return SMX_ELAUNCH;          \
  } while (0)

static inline int smx_launch_status() {
  return hipGetLastError() == hipSuccess ? SMX_OK : SMX_ELAUNCH;
}

static inline int smx_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
