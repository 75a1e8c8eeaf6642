// Developer probe: how does v_mfma_f32_32x32x16_bf16 round D = A.B + C?  One wave, random bf16 A [32x16], B [16x32] with O(1)
// products, C = s * N(0,1) for a range of scales s: error of the hardware result against the exact sum (fp64 on the host;
// every bf16 product is exact in fp64), in ulps of the fp32 result: mean (a non-zero mean = truncation bias) and rms
// (round-to-nearest gives mean 0, rms 0.29).  Same for v_mfma_f32_32x32x2f32 (C + 2 products per op).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_bias.hip -o tools/micro/mfma_bias.bin && tools/micro/mfma_bias.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void k_bf16(const float* A, const float* B, const float* C, float* D) {   // A [32][16], B [16][32], C/D [32][32]
  const int lane = threadIdx.x, fr = lane & 31, fh = lane >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[fr * 16 + 8 * fh + j]; b[j] = (__bf16)B[(8 * fh + j) * 32 + fr]; }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * fh) * 32 + fr];
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * fh) * 32 + fr] = c[r];
}

__global__ void k_f32(const float* A, const float* B, const float* C, float* D) {    // A [32][2], B [2][32]
  const int lane = threadIdx.x, fr = lane & 31, fh = lane >> 5;
  const float a = A[fr * 2 + fh], b = B[fh * 32 + fr];
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * fh) * 32 + fr];
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * fh) * 32 + fr] = c[r];
}

static double gauss() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }
static float to_bf16(float x) { float y; unsigned u; memcpy(&u, &x, 4); u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000u; memcpy(&y, &u, 4); return y; }

int main() {
  float *dA, *dB, *dC, *dD;
  hipMalloc(&dA, 32 * 16 * 4); hipMalloc(&dB, 16 * 32 * 4); hipMalloc(&dC, 1024 * 4); hipMalloc(&dD, 1024 * 4);
  srand(3);
  for (int kind = 0; kind < 2; ++kind) {
    const int K = kind == 0 ? 16 : 2;
    for (int e = -20; e <= 20; e += 4) {
      const double s = ldexp(1.0, e);
      double sum = 0, sq = 0, sum_abs = 0; long n = 0; double worst = 0;
      for (int rep = 0; rep < 200; ++rep) {
        std::vector<float> A(32 * K), B(K * 32), C(1024), D(1024);
        for (auto& v : A) v = kind == 0 ? to_bf16((float)gauss()) : (float)gauss();
        for (auto& v : B) v = kind == 0 ? to_bf16((float)gauss()) : (float)gauss();
        for (auto& v : C) v = (float)(s * gauss());
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
        if (kind == 0) hipLaunchKernelGGL(k_bf16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        else hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
          double ex = C[i * 32 + j];
          for (int k = 0; k < K; ++k) ex += (double)A[i * K + k] * (double)B[k * 32 + j];
          const double got = D[i * 32 + j];
          int ee; frexp(got != 0 ? got : ex, &ee);
          const double ulp = ldexp(1.0, ee - 24);
          const double err = (got - ex) / ulp;
          sum += err; sq += err * err; sum_abs += fabs(err); ++n; if (fabs(err) > worst) worst = fabs(err);
        }
      }
      printf("%s  C scale 2^%+3d : error in ulps of the result: mean %+.4f  rms %.4f  max %.3f\n", kind == 0 ? "bf16 32x32x16" : "f32  32x32x2 ", e, sum / n, sqrt(sq / n), worst);
    }
  }
  return 0;
}
