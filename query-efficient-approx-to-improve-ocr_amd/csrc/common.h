// Internal helpers shared by the HIP translation units of libqea_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/qea_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define QEA_WAVE 64

void qea_set_error(const char* fmt, ...);

#define QEA_REQUIRE(cond, ...)                \
  do {                                        \
    if (!(cond)) {                            \
      qea_set_error(__VA_ARGS__);             \
      return QEA_ERR_INVALID;                 \
    }                                         \
  } while (0)

#define QEA_CHECK_LAUNCH()                                            \
  do {                                                                \
    hipError_t e__ = hipGetLastError();                               \
    if (e__ != hipSuccess) {                                          \
      qea_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,    \
                    hipGetErrorString(e__));                          \
      return QEA_ERR_LAUNCH;                                          \
    }                                                                 \
  } while (0)

static inline int qea_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Bijective XCD-aware remap of a 1-D block id: blocks that land on the same XCD
// (bid % 8 equal under round-robin dispatch) get a contiguous chunk of tile ids, so
// neighbouring tiles share that XCD's L2.  Speed only; any placement is correct.
__device__ __forceinline__ int qea_xcd_swizzle(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, k = bid >> 3;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + k;
}

// wave-level reductions over 64 lanes
__device__ __forceinline__ float qea_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double qea_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float qea_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// event-bracketed timing of one kernel class (bench.py roofline leg)
void qea_prof_begin(int klass, hipStream_t s);
void qea_prof_end(int klass, hipStream_t s, double flops, double bytes);
