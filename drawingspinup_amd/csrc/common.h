// Shared host/device helpers for libdsu_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/dsu_hip.h"

#define DSU_WAVE 64

// max(x, 0).  fmaxf() is llvm.maxnum, and in IEEE mode the backend quiets a possible signalling NaN
// first (`v_max_f32 v, v, v`) whenever it cannot see where the value came from: a second VALU slot
// per ReLU / Softplus.  A translation unit WITHOUT matrix instructions may define DSU_RELU_ONE_VMAX
// before including this header and get the single instruction (same result for every non-NaN
// input; hashgrid.hip does: forward 0.130 -> 0.122 ms).  NOT where the operand can be an MFMA
// result: the wait states between an MFMA and the first VALU read of its result are inserted by
// the compiler, which does not look inside an asm statement — the texture forward read
// accumulators early that way (caught by test_forward_matches_reference_forward_fixture).
__device__ __forceinline__ float dsu_relu(float x) {
#if defined(DSU_RELU_ONE_VMAX) && !defined(DSU_RELU_FMAXF)
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
#else
  return fmaxf(x, 0.0f);
#endif
}

#define DSU_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return DSU_ELAUNCH;               \
  } while (0)

// A/B switches (alternative kernels, placement experiments, debug output) exist only in variant
// builds: `python -m drawingspinup_amd.build --variant ab -DDSU_AB_SWITCHES` compiles the
// environment look-ups in, the product library takes the defaults as constants (tools/ drives
// the variants through DSU_HIP_LIB).
#ifdef DSU_AB_SWITCHES
#include <stdlib.h>
#include <string.h>
static inline int dsu_ab_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
static inline bool dsu_ab_is(const char* name, const char* value) {
  const char* e = getenv(name);
  return e && strcmp(e, value) == 0;
}
#else
static inline int dsu_ab_int(const char*, int dflt) { return dflt; }
static inline bool dsu_ab_is(const char*, const char*) { return false; }
#endif

// Workgroups of the two one-wave-per-SIMD kernels of the NSR step (dsu_set_onewave_grid_cap, capi.hip)
extern "C" int32_t dsu_onewave_grid_cap_value;
// Priority of the NSR driver's side stream (dsu_set_nsr_side_stream_priority, capi.hip; nsr_driver.hip has the story)
#ifndef DSU_NSR_SIDE_PRIO_DEFAULT
#define DSU_NSR_SIDE_PRIO_DEFAULT 1
#endif
extern "C" int32_t dsu_nsr_side_priority_value;
extern "C" int32_t dsu_nsr_side_pool_value;
extern "C" int32_t dsu_scatter_grid_cap_value;
static inline int dsu_onewave_blocks(int64_t n, int threads, int max_blocks) {
  int cap = dsu_onewave_grid_cap_value;
  if (cap < 1 || cap > max_blocks) cap = max_blocks;
  int64_t b = (n + threads - 1) / threads;
  if (b < 1) b = 1;
  return (int)(b > cap ? cap : b);
}

static inline int dsu_blocks_for(int64_t n, int threads) {
  return (int)((n + threads - 1) / threads);
}

// Memory-bound launches are capped at 256 CUs x 8 blocks and grid-stride the rest.
static inline int dsu_capped_blocks(int64_t n, int threads, int cap = 2048) {
  int64_t b = (n + threads - 1) / threads;
  if (b < 1) b = 1;
  return (int)(b > cap ? cap : b);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) costs ~55 us of host time per call (measured with
// rocprofv3 --hip-runtime-trace): raise the limit once per kernel and process, and again only if a
// larger size is ever requested.
#define DSU_ENSURE_DYN_LDS(kernel, bytes)                                                   \
  do {                                                                                      \
    static int dsu_lds_set__ = 0;                                                           \
    if (dsu_lds_set__ < (int)(bytes)) {                                                     \
      if (hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)(bytes)) != hipSuccess)                                  \
        return DSU_ELAUNCH;                                                                 \
      dsu_lds_set__ = (int)(bytes);                                                         \
    }                                                                                       \
  } while (0)
