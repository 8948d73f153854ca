// shared host-side helpers for the libsmx translation units
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "smx.h"

#define SMX_HIP(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) return SMX_ELAUNCH;          \
  } while (0)

// hipGetLastError() is per-thread and sticky across ALL runtime calls, including the host
// framework's own (a benign hipErrorNotReady / invalid-pointer probe left behind by the caller
// must not be reported as our launch failing): clear it right before the launch, read it after.
static inline void smx_clear_stale_error() {
  const hipError_t e = hipGetLastError();
#ifdef SMX_TOOLS
  if (e != hipSuccess && getenv("SMX_DEBUG_STALE")) fprintf(stderr, "[libsmx] cleared stale caller error: %s\n", hipGetErrorString(e));
#else
  (void)e;
#endif
}

#define SMX_LAUNCH(...)               \
  do {                                \
    smx_clear_stale_error();          \
    hipLaunchKernelGGL(__VA_ARGS__);  \
  } while (0)

static inline int smx_launch_status() {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return SMX_OK;
  fprintf(stderr, "[libsmx] kernel launch failed: %s\n", hipGetErrorString(e));
  return SMX_ELAUNCH;
}

// tuning table (tuning.hip): plain ints read by the launchers; set via smx_set_tuning / SMX_* env at first use
enum { SMX_TUNE_WINO_NW = 0, SMX_TUNE_WINO_ABLATE, SMX_TUNE_GEMM_VARIANT, SMX_TUNE_GEMM_XCD_SWIZZLE,
       SMX_TUNE_WARP_ROWS, SMX_TUNE_WARP_REORDER, SMX_TUNE_ATTN16, SMX_TUNE_WINO_WIDE, SMX_TUNE_WINO_NT, SMX_TUNE_ATTN4_MFMA, SMX_TUNE_CONV16_SLAB, SMX_TUNE_ATTN_BWD_MFMA, SMX_TUNE_VQ_SPLIT, SMX_TUNE_WARP_NT, SMX_TUNE_WGRAD_REGION, SMX_TUNE_WGRAD_SLOTS, SMX_TUNE_WINO_STAGGER, SMX_TUNE_WINO_WS, SMX_TUNE_GEMM_LOADER, SMX_TUNE_WINO_XCD, SMX_TUNE_WINO_BF3_SHAPE, SMX_TUNE_ATTN_BF3, SMX_TUNE_COUNT };
int smx_tune(int key);

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE property of a kernel: set it once per (kernel, device) of this translation unit,
// under a lock (a process that moves to another GPU, or two threads racing on a first call, must not launch without it)
#include <mutex>
static inline hipError_t smx_max_dynamic_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static struct { const void* fn; int dev; } done[64];
  static int n = 0;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> g(mu);
  for (int i = 0; i < n; ++i) if (done[i].fn == fn && done[i].dev == dev) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && n < 64) { done[n].fn = fn; done[n].dev = dev; ++n; }
  return e;
}

static inline int smx_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// region form of the 3x3 / s1 / p1 weight gradient (train_wgrad_region.hip), dispatched by train_gemm.hip's wgrad_launch
bool smx_wgrad_region_shape_ok(int nb, int Cout, int Cin, int Hin, int Win, int Ho, int Wo, int kh, int kw, int stride, int pad_t, int pad_l, int up2);
int smx_wgrad_region_split(int M, int Cout, int Cin, int Ho, int Wo, int* uper_out);
int smx_wgrad_region_launch(bool bf16, const float* dy, int ldy, const float* x, int ldx, int M, int Cout, int Hin, int Win, int Cin, int Ho, int Wo,
                            int up2, float* ws, float* bias_ws, int msplit, void* stream);
