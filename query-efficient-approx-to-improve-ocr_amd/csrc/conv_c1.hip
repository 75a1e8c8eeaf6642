// The single-channel ends of the two networks — HBM-bound direct kernels (no MFMA: K = 9).
//   * 3x3 pad-1 convolution with C_in = 1 (UNet enc1conv1, models/model_unet.py:13;
//     CRNN conv1, models/model_crnn.py:37,48): forward, weight(+bias) gradient, input gradient
//   * UNet head: 1x1 conv C -> 1 + sigmoid (models/model_unet.py:45,76): forward, backward
// Output channels are handled as float4 columns: `cols = C/4` adjacent lanes share a pixel.
#include "common.h"

namespace {

__global__ void conv_c1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                   float* __restrict__ y, int ldy, int B, int H, int W, int Co, int relu) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [9][Co] tap-major
  for (int i = threadIdx.x; i < Co * 9; i += blockDim.x) {
    const int co = i / 9, tap = i - co * 9;
    sw[tap * Co + co] = w[i];
  }
  __syncthreads();
  const int cols = Co / 4;
  const long long n = (long long)B * H * W * cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int ct = (int)(i % cols);
    const long long pix = i / cols;
    const int pw = (int)(pix % W);
    const int ph = (int)((pix / W) % H);
    const float* xb = x + (pix - (long long)ph * W - pw);  // image base
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (bias) acc = *reinterpret_cast<const f32x4*>(bias + ct * 4);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = ph + kh - 1;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = pw + kw - 1;
        float xv = 0.f;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) xv = xb[ih * W + iw];
        const f32x4 wv = *reinterpret_cast<const f32x4*>(sw + (kh * 3 + kw) * Co + ct * 4);
        acc += wv * xv;
      }
    }
    if (relu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = fmaxf(acc[k], 0.f);
    }
    *reinterpret_cast<f32x4*>(y + pix * ldy + ct * 4) = acc;
  }
}

// dW[co][tap] = sum_p dy[p][co] * x[p + tap]; db[co] = sum_p dy[p][co].
// thread (rt, ct): 4 channels x (9 taps + bias) fp32 partials over <= ROWS_PER_THREAD pixels,
// fp32 LDS tree over rt, fp64 across blocks (second kernel).
constexpr int C1_ROWS_PER_THREAD = 32;

__global__ __launch_bounds__(256) void conv_c1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int lddy, int B, int H,
                                                            int W, int Co, int rt_n, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float sacc[];  // [rt_n][10][Co]
  const int cols = Co / 4;
  const int ct = threadIdx.x % cols, rt = threadIdx.x / cols;
  const long long M = (long long)B * H * W;
  f32x4 acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (rt < rt_n) {
    const long long r0 = (long long)blockIdx.x * rt_n * C1_ROWS_PER_THREAD;
    for (int j = 0; j < C1_ROWS_PER_THREAD; ++j) {
      const long long pix = r0 + (long long)j * rt_n + rt;
      if (pix >= M) break;
      const int pw = (int)(pix % W);
      const int ph = (int)((pix / W) % H);
      const float* xb = x + (pix - (long long)ph * W - pw);
      const f32x4 g = *reinterpret_cast<const f32x4*>(dy + pix * lddy + ct * 4);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int ih = ph + kh - 1;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int iw = pw + kw - 1;
          float xv = 0.f;
          if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) xv = xb[ih * W + iw];
          acc[kh * 3 + kw] += g * xv;
        }
      }
      acc[9] += g;
    }
#pragma unroll
    for (int t = 0; t < 10; ++t) *reinterpret_cast<f32x4*>(sacc + ((size_t)rt * 10 + t) * Co + ct * 4) = acc[t];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 10 * Co; e += blockDim.x) {
    float s = 0.f;
    for (int r = 0; r < rt_n; ++r) s += sacc[(size_t)r * 10 * Co + e];
    ws[(size_t)blockIdx.x * 10 * Co + e] = s;  // e = tap*Co + co
  }
}

__global__ void conv_c1_wgrad_finalize_kernel(const float* __restrict__ ws, int nblk, int Co, float* dw, float* db, int accumulate) {
  // one wave per output element: lanes stride over the partial blocks
  const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // tap*Co + co
  if (e >= 10 * Co) return;
  double s = 0;
  for (int b = threadIdx.x & 63; b < nblk; b += 64) s += (double)ws[(size_t)b * 10 * Co + e];
  s = qea_wave_sum_d(s);
  if ((threadIdx.x & 63) != 0) return;
  const int tap = e / Co, co = e - tap * Co;
  if (tap < 9) {
    float* d = dw + co * 9 + tap;
    *d = accumulate ? *d + (float)s : (float)s;
  } else if (db) {
    db[co] = accumulate ? db[co] + (float)s : (float)s;
  }
}

// dx[p] = sum_{kh,kw,co} dy[p + (1-kh, 1-kw)][co] * w[co][kh][kw]
template <int COLS>
__global__ __launch_bounds__(256) void conv_c1_dgrad_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ w,
                                                            float* __restrict__ dx, int B, int H, int W, int accumulate) {
  constexpr int Co = COLS * 4;
  const int ct = threadIdx.x % COLS;
  f32x4 wr[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) wr[t][k] = w[(ct * 4 + k) * 9 + t];
  const long long M = (long long)B * H * W;
  const long long n = M * COLS;
  const long long stride = (long long)gridDim.x * blockDim.x;
  // all lanes of a pixel group iterate together (n is a multiple of COLS and so is the stride)
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n + stride - 1) / stride * stride; i += stride) {
    const bool live = i < n;
    const long long pix = live ? i / COLS : 0;
    const int pw = (int)(pix % W);
    const int ph = (int)((pix / W) % H);
    float s = 0.f;
    if (live) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int oh = ph + 1 - kh;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ow = pw + 1 - kw;
          if ((unsigned)oh < (unsigned)H && (unsigned)ow < (unsigned)W) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(dy + (pix + (long long)(oh - ph) * W + (ow - pw)) * lddy + ct * 4);
            const f32x4 wv = wr[kh * 3 + kw];
            s += g[0] * wv[0] + g[1] * wv[1] + g[2] * wv[2] + g[3] * wv[3];
          }
        }
      }
    }
#pragma unroll
    for (int o = COLS / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (live && ct == 0) dx[pix] = accumulate ? dx[pix] + s : s;
  }
  (void)Co;
}

// ---- UNet head ----
template <int COLS>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ y, long long M) {
  const int ct = threadIdx.x % COLS;
  const f32x4 wv = *reinterpret_cast<const f32x4*>(w + ct * 4);
  const float bias = b[0];
  const long long n = M * COLS;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n + stride - 1) / stride * stride; i += stride) {
    const bool live = i < n;
    const long long pix = live ? i / COLS : 0;
    float s = 0.f;
    if (live) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + pix * ldx + ct * 4);
      s = xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
    }
#pragma unroll
    for (int o = COLS / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (live && ct == 0) y[pix] = 1.f / (1.f + expf(-(s + bias)));
  }
}

// dz = dyy * y * (1 - y); dx = dz * w; dw += dz * x; db += dz.  Partials in fp64 per thread.
template <int COLS>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y,
                                                       const float* __restrict__ dyy, const float* __restrict__ w, float* __restrict__ dx,
                                                       int lddx, long long M, double* __restrict__ ws) {
  constexpr int C = COLS * 4;
  constexpr int RT = 256 / COLS;
  __shared__ double sred[RT][C + 1];
  const int ct = threadIdx.x % COLS, rt = threadIdx.x / COLS;
  const f32x4 wv = *reinterpret_cast<const f32x4*>(w + ct * 4);
  double aw[4] = {0, 0, 0, 0}, ab = 0;
  const long long n = M * COLS;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long pix = i / COLS;
    const float yy = y[pix];
    const float dz = dyy[pix] * yy * (1.f - yy);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + pix * ldx + ct * 4);
    *reinterpret_cast<f32x4*>(dx + pix * lddx + ct * 4) = wv * dz;
#pragma unroll
    for (int k = 0; k < 4; ++k) aw[k] += (double)(dz * xv[k]);
    if (ct == 0) ab += (double)dz;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) sred[rt][ct * 4 + k] = aw[k];
  if (ct == 0) sred[rt][C] = ab;
  __syncthreads();
  for (int e = threadIdx.x; e <= C; e += blockDim.x) {
    double s = 0;
    for (int r = 0; r < RT; ++r) s += sred[r][e];
    ws[(size_t)blockIdx.x * (C + 1) + e] = s;
  }
}

__global__ void head_bwd_finalize_kernel(const double* __restrict__ ws, int nblk, int C, float* dw, float* db, int accumulate) {
  const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (e > C) return;
  double s = 0;
  for (int b = threadIdx.x & 63; b < nblk; b += 64) s += ws[(size_t)b * (C + 1) + e];
  s = qea_wave_sum_d(s);
  if ((threadIdx.x & 63) != 0) return;
  float* d = (e < C) ? dw + e : db;
  *d = accumulate ? *d + (float)s : (float)s;
}

int grid_for(long long n, int cap = 4096) {
  long long g = (n + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int qea_conv_c1_fwd(const float* x, const float* w, const float* bias, float* y, int32_t ldy, int32_t B, int32_t H, int32_t W,
                               int32_t Co, int32_t relu, void* stream) {
  QEA_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0 && Co > 0 && Co % 4 == 0 && ldy % 4 == 0 && ldy >= Co, "qea_conv_c1_fwd: bad arguments");
  const long long n = (long long)B * H * W * (Co / 4);
  hipLaunchKernelGGL(conv_c1_fwd_kernel, dim3(grid_for(n, 8192)), dim3(256), (size_t)Co * 9 * sizeof(float), (hipStream_t)stream, x, w, bias, y,
                     ldy, B, H, W, Co, relu);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

static int c1_wgrad_geom(long long M, int Co, int* rt_n) {
  const int cols = Co / 4;
  *rt_n = 256 / cols;
  const long long per_block = (long long)(*rt_n) * C1_ROWS_PER_THREAD;
  return (int)((M + per_block - 1) / per_block);
}

extern "C" size_t qea_conv_c1_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Co) {
  if (B <= 0 || H <= 0 || W <= 0 || Co <= 0 || Co % 4 || Co / 4 > 256) return 0;
  int rt;
  const int grid = c1_wgrad_geom((long long)B * H * W, Co, &rt);
  return (size_t)grid * 10 * Co * sizeof(float);
}

extern "C" int qea_conv_c1_wgrad(const float* x, const float* dy, int32_t lddy, float* dw, float* db, int32_t B, int32_t H, int32_t W,
                                 int32_t Co, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  QEA_REQUIRE(x && dy && dw && B > 0 && H > 0 && W > 0 && Co > 0 && Co % 4 == 0 && Co / 4 <= 256 && lddy % 4 == 0, "qea_conv_c1_wgrad: bad arguments");
  int rt;
  const int grid = c1_wgrad_geom((long long)B * H * W, Co, &rt);
  QEA_REQUIRE(workspace && workspace_bytes >= (size_t)grid * 10 * Co * sizeof(float), "qea_conv_c1_wgrad: workspace too small");
  const size_t lds = (size_t)rt * 10 * Co * sizeof(float);
  QEA_REQUIRE(lds <= 160 * 1024, "qea_conv_c1_wgrad: Co too large for LDS");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_c1_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(conv_c1_wgrad_kernel, dim3(grid), dim3(256), lds, s, x, dy, lddy, B, H, W, Co, rt, (float*)workspace);
  hipLaunchKernelGGL(conv_c1_wgrad_finalize_kernel, dim3(qea_cdiv(10 * Co, 4)), dim3(256), 0, s, (const float*)workspace, grid, Co, dw, db,
                     accumulate);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_conv_c1_dgrad(const float* dy, int32_t lddy, const float* w, float* dx, int32_t B, int32_t H, int32_t W, int32_t Co,
                                 int32_t accumulate, void* stream) {
  QEA_REQUIRE(dy && w && dx && B > 0 && H > 0 && W > 0 && lddy % 4 == 0, "qea_conv_c1_dgrad: bad arguments");
  const long long n = (long long)B * H * W * (Co / 4);
  const int grid = grid_for(n, 8192);
  hipStream_t s = (hipStream_t)stream;
  switch (Co) {
    case 32: hipLaunchKernelGGL(conv_c1_dgrad_kernel<8>, dim3(grid), dim3(256), 0, s, dy, lddy, w, dx, B, H, W, accumulate); break;
    case 64: hipLaunchKernelGGL(conv_c1_dgrad_kernel<16>, dim3(grid), dim3(256), 0, s, dy, lddy, w, dx, B, H, W, accumulate); break;
    case 128: hipLaunchKernelGGL(conv_c1_dgrad_kernel<32>, dim3(grid), dim3(256), 0, s, dy, lddy, w, dx, B, H, W, accumulate); break;
    default: qea_set_error("qea_conv_c1_dgrad: Co=%d not in {32,64,128}", Co); return QEA_ERR_INVALID;
  }
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_head_fwd(const float* x, int32_t ldx, const float* w, const float* b, float* y, int64_t M, int32_t C, void* stream) {
  QEA_REQUIRE(x && w && b && y && M > 0 && ldx % 4 == 0, "qea_head_fwd: bad arguments");
  const int grid = grid_for(M * (C / 4), 8192);
  hipStream_t s = (hipStream_t)stream;
  switch (C) {
    case 32: hipLaunchKernelGGL(head_fwd_kernel<8>, dim3(grid), dim3(256), 0, s, x, ldx, w, b, y, (long long)M); break;
    case 64: hipLaunchKernelGGL(head_fwd_kernel<16>, dim3(grid), dim3(256), 0, s, x, ldx, w, b, y, (long long)M); break;
    default: qea_set_error("qea_head_fwd: C=%d not in {32,64}", C); return QEA_ERR_INVALID;
  }
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" size_t qea_head_bwd_workspace_bytes(int64_t M, int32_t C) {
  if (M <= 0 || C <= 0) return 0;
  return (size_t)grid_for(M * (C / 4), 1024) * (C + 1) * sizeof(double);
}

extern "C" int qea_head_bwd(const float* x, int32_t ldx, const float* y, const float* dyy, const float* w, float* dx, int32_t lddx,
                            float* dw, float* db, int32_t accumulate, int64_t M, int32_t C, void* workspace, size_t workspace_bytes,
                            void* stream) {
  QEA_REQUIRE(x && y && dyy && w && dx && dw && db && M > 0 && ldx % 4 == 0 && lddx % 4 == 0, "qea_head_bwd: bad arguments");
  const int grid = grid_for(M * (C / 4), 1024);
  QEA_REQUIRE(workspace && workspace_bytes >= (size_t)grid * (C + 1) * sizeof(double), "qea_head_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  switch (C) {
    case 32: hipLaunchKernelGGL(head_bwd_kernel<8>, dim3(grid), dim3(256), 0, s, x, ldx, y, dyy, w, dx, lddx, (long long)M, (double*)workspace); break;
    case 64: hipLaunchKernelGGL(head_bwd_kernel<16>, dim3(grid), dim3(256), 0, s, x, ldx, y, dyy, w, dx, lddx, (long long)M, (double*)workspace); break;
    default: qea_set_error("qea_head_bwd: C=%d not in {32,64}", C); return QEA_ERR_INVALID;
  }
  hipLaunchKernelGGL(head_bwd_finalize_kernel, dim3(qea_cdiv(C + 1, 4)), dim3(256), 0, s, (const double*)workspace, grid, C, dw, db, accumulate);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}
