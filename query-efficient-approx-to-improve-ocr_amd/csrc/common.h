// Internal helpers shared by the HIP translation units of libqea_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/qea_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// x = h + m + l with three bf16 values, |x - (h+m+l)| <= 2^-24 |x| (each residual is exact in fp32).  A product of two
// such triples keeps hh, hm, mh, hl, lh, mm (six bf16 MFMAs, fp32 accumulate); the dropped ml, lm, ll are < 2^-23 |ab|.
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ __forceinline__ void qea_split3(const f32x4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const __bf16 hk = (__bf16)v[k];
    const float r1 = v[k] - (float)hk;
    const __bf16 mk = (__bf16)r1;
    const float r2 = r1 - (float)mk;
    h[k] = hk;
    m[k] = mk;
    l[k] = (__bf16)r2;
  }
}
#endif

// TWO-way fp16 split of a SCALED operand (round 3): xs = x * s with s a power of two chosen from the tensor's largest finite
// magnitude m so that m * s is in [2^14, 2^15) (qea_f16_scale); h = f16(xs) and l = f16(xs - h) carry 11 + 11 significant bits
// (+ the sign of l): |xs - (h + l)| <= 2^-24 |xs| — as good as the three bf16 pieces — for every element down to about 2^-16 of m
// (|xs| >= 1/2: the residual's last fp32 bit is then still >= 2^-24, the spacing of the SUBNORMAL fp16 values l falls into once
// |xs| < ~2^-3), and an ABSOLUTE error below 2^-25 in scaled units = 2^-40 m for the smaller ones (e.g. ~2^-11 relative at 2^-29 of
// m: negligible against the rounding of the large terms such an element is summed with).  A product keeps hh,
// hl, lh (three v_mfma_f32_32x32x16_f16, fp32 accumulate); the dropped ll is < 2^-22 |ab|.  Non-finite elements do not enter m
// and stay non-finite in h (inf) / l (NaN).
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ __forceinline__ void qea_split2_f16(const f32x4 v, float s, f16x4& h, f16x4& l) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float xs = v[k] * s;
    const _Float16 hk = (_Float16)xs;
    h[k] = hk;
    l[k] = (_Float16)(xs - (float)hk);
  }
}
// s = 2^(15 - e) for m = f * 2^e, f in [0.5, 1): bits of the scale and of its inverse from the exponent field of m (m >= 0, finite)
__device__ __forceinline__ void qea_f16_scale(float m, float& s, float& inv) {
  const unsigned E = (__float_as_uint(m) >> 23) & 0xffu;     // biased exponent; m = 1.x * 2^(E - 127), floor(log2 m) = E - 127
  int se = 14 - ((int)E - 127);                                // s = 2^se puts m * s into [2^14, 2^15)
  if (m == 0.f || E == 0) se = 0;                              // all-zero (or subnormal-only) tensor: nothing to scale
  se = se > 126 ? 126 : (se < -126 ? -126 : se);
  s = __uint_as_float((unsigned)(se + 127) << 23);
  inv = __uint_as_float((unsigned)(127 - se) << 23);
}
#endif

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

// ---- abs-max carried by the PRODUCER of a tensor (round 3): the two-way fp16 split needs the largest finite magnitude of its
// operands; a kernel that writes a tensor folds |v| of what it stores into a running maximum and commits it with one atomicMax
// per wave (non-negative floats order like their bit patterns).  The caller zero-fills the 4-byte slot; several launches (a
// BatchNorm applied per group, the two halves of a concat buffer) may share one slot.
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ __forceinline__ float qea_amax_acc(float m, float v) {
  const float a = fabsf(v);
  return (a <= 3.4028234663852886e38f && a > m) ? a : m;      // false for NaN; inf excluded
}
// every lane of the wave must reach this call (no early returns before it)
__device__ __forceinline__ void qea_amax_commit(float m, float* out) {
  if (!out) return;
  m = qea_wave_max(m);
  // thousands of waves meet on ONE address: an atomic only when this wave would raise the value it can see (a stale read costs a
  // redundant atomic, never a wrong result) — same-address atomics from every wave cost 0.1 ms per launch
  if ((threadIdx.x & 63) == 0 && m > 0.f) {
    unsigned* slot = reinterpret_cast<unsigned*>(out);
    const unsigned mine = __float_as_uint(m);
    if (mine > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, mine);
  }
}
// The same for the elementwise kernels (256-thread workgroups, every thread reaches the call): ONE gated access per workgroup.  Per-wave
// commits made a 26 MB BatchNorm apply take 64 us instead of 6 (16 k waves queueing on one L2 address: the gate's load alone).
__device__ __forceinline__ void qea_amax_commit_block(float m, float* out) {
  if (!out) return;
  __shared__ float qea_amax_sm[16];
  m = qea_wave_max(m);
  if ((threadIdx.x & 63) == 0) qea_amax_sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float b = qea_amax_sm[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) b = fmaxf(b, qea_amax_sm[w]);
    if (b > 0.f) {
      unsigned* slot = reinterpret_cast<unsigned*>(out);
      const unsigned mine = __float_as_uint(b);
      if (mine > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, mine);
    }
  }
  __syncthreads();                                         // (a kernel may commit two maxima through the same staging array)
}
#endif

// event-bracketed timing of one kernel class (bench.py roofline leg)
// QEA_MFMA=f32 keeps every product on v_mfma_f32_32x32x2_f32; anything else (default) allows the split-bf16 kernels
bool qea_split_bf16_enabled();
void qea_prof_begin(int klass, hipStream_t s);
// split: 0 = fp32 MFMA, 1 = three-way bf16 split (six MFMAs per product), 2 = two-way fp16 split (three)
void qea_prof_end(int klass, hipStream_t s, double flops, double bytes, int split = 0, int tag = 0);
void qea_prof_abort(int klass);
