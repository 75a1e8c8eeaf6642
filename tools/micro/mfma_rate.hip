// Developer micro-benchmark: sustained MFMA rate on random operands (registers only), fp32 32x32x2 vs bf16 32x32x16,
// to price an fp32 emulation by 6 bf16 MFMAs (3-way mantissa split) under the clock the chip actually holds.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k_f32(const float* in, float* out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 0xffff]; b[i] = in[(t * 8 + 4 + i) & 0xffff]; }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[1], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(s + 1) & 3], b[2], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(s + 1) & 3], b[3], acc[3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[t] = s;
}

__global__ __launch_bounds__(256) void k_bf16(const float* in, float* out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)in[(t * 64 + i * 8 + e) & 0xffff]; b[i][e] = (__bf16)in[(t * 64 + 32 + i * 8 + e) & 0xffff]; }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[1], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + 1) & 3], b[2], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + 1) & 3], b[3], acc[3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[t] = s;
}

int main() {
  const int blocks = 256 * 2, threads = 256;  // 8 waves per CU
  std::vector<float> h(65536);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *din, *dout;
  hipMalloc(&din, h.size() * 4);
  hipMalloc(&dout, (size_t)blocks * threads * 4);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    const int iters = (which == 0 ? 20000 : 40000) * (getenv("LONG") ? 40 : 1);   // LONG=1: ~0.8 s per launch, the clock settles
    for (int rep = 0; rep < (getenv("LONG") ? 4 : 3); ++rep) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k_f32, dim3(blocks), dim3(threads), 0, 0, din, dout, iters);
      else hipLaunchKernelGGL(k_bf16, dim3(blocks), dim3(threads), 0, 0, din, dout, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      const double waves = (double)blocks * threads / 64;
      const double mfma = waves * iters * 16.0;
      const double flops = mfma * (which == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16);
      printf("%s: %.1f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)%s\n", which == 0 ? "f32 32x32x2 " : "bf16 32x32x16", ms,
             flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (mfma / 1024.0), which == 1 ? "  -> /6 = fp32-equivalent" : "");
      if (which == 1) printf("    bf16x3 (6 MFMAs per product) ceiling: %.1f TFLOP/s fp32-equivalent\n", flops / ms / 1e9 / 6.0);
    }
  }
  return 0;
}
