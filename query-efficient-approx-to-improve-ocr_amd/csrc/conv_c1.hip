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

// Forward, fast path (W % 4 == 0, fewer than 2^31 pixels): a thread makes FOUR consecutive pixels x 4 channels from one 3 x 6 input
// window, filter column and bias in registers, 32-bit index arithmetic.  (The one-pixel kernel above spends most of its time on
// 64-bit divisions and on nine bounds-checked loads per 16 output bytes: 2.9 TB/s of stores on the CRNN's first layer.)
template <int COLS>
__global__ __launch_bounds__(256) void conv_c1_fwd4_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ y, int ldy, int B, int H, int W, int relu) {
  constexpr int GPB = 256 / COLS;                          // pixel groups per block and iteration
  const int ct = threadIdx.x % COLS;
  f32x4 wr[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) wr[t][k] = w[(ct * 4 + k) * 9 + t];
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv = *reinterpret_cast<const f32x4*>(bias + ct * 4);
  const unsigned W4 = (unsigned)W >> 2;
  const unsigned groups = (unsigned)B * (unsigned)H * W4;
  const unsigned gstride = gridDim.x * GPB;
  for (unsigned g = blockIdx.x * GPB + threadIdx.x / COLS; g < groups; g += gstride) {
    const unsigned row = g / W4;                           // b * H + ph
    const int pw0 = (int)(g - row * W4) * 4;
    const int ph = (int)(row % (unsigned)H);
    const float* xr = x + (size_t)row * W;                 // this image row
    float xv[3][6];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const bool rok = (unsigned)(ph + kh - 1) < (unsigned)H;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const int iw = pw0 + c - 1;
        xv[kh][c] = (rok && (unsigned)iw < (unsigned)W) ? xr[(kh - 1) * W + iw] : 0.f;
      }
    }
    float* yo = y + ((size_t)row * W + pw0) * ldy + ct * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 acc = bv;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc += wr[kh * 3 + kw] * xv[kh][j + kw];     // (tap order of the one-pixel kernel: same bits)
      if (relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = fmaxf(acc[k], 0.f);
      }
      *reinterpret_cast<f32x4*>(yo + (size_t)j * ldy) = acc;
    }
  }
}

// The fast-path forward with the 2x2 max-pool that follows it (CRNN conv1 -> ReLU -> max_pool2d(2,2), model_crnn.py:48): a thread makes
// 2 rows x 4 pixels x 4 channels from one 4 x 6 input window, stores the eight activations (the pool's backward needs them) and the two
// pooled pixels — same tap order, scan order and NaN rule as conv_c1_fwd4_kernel + maxpool_fwd_kernel (bit-identical), without the
// second read of the full-resolution tensor (2.1 GB at B = 2048: 557 us).
template <int COLS>
__global__ __launch_bounds__(256) void conv_c1_fwd4_pool_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                float* __restrict__ y, int ldy, float* __restrict__ pooled, int ldp, int B, int H,
                                                                int W, int relu, float* __restrict__ amax_p) {
  constexpr int GPB = 256 / COLS;
  const int ct = threadIdx.x % COLS;
  f32x4 wr[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) wr[t][k] = w[(ct * 4 + k) * 9 + t];
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv = *reinterpret_cast<const f32x4*>(bias + ct * 4);
  const unsigned W4 = (unsigned)W >> 2, H2 = (unsigned)H >> 1;
  const unsigned groups = (unsigned)B * H2 * W4;
  const unsigned gstride = gridDim.x * GPB;
  float pm = 0.f;
  for (unsigned g = blockIdx.x * GPB + threadIdx.x / COLS; g < groups; g += gstride) {
    const unsigned prow = g / W4;                          // b * (H / 2) + pooled row
    const int pw0 = (int)(g - prow * W4) * 4;
    const int ph = (int)(prow % H2) * 2;                   // first of the two image rows
    const size_t row = (size_t)prow * 2;                   // b * H + ph
    const float* xr = x + row * W;
    float xv[4][6];
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
      const bool rok = (unsigned)(ph + kh - 1) < (unsigned)H;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const int iw = pw0 + c - 1;
        xv[kh][c] = (rok && (unsigned)iw < (unsigned)W) ? xr[(kh - 1) * W + iw] : 0.f;
      }
    }
    f32x4 o[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 acc = bv;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) acc += wr[kh * 3 + kw] * xv[r + kh][j + kw];
        if (relu) {
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] = fmaxf(acc[k], 0.f);
        }
        o[r][j] = acc;
        if (y) *reinterpret_cast<f32x4*>(y + ((row + r) * W + pw0 + j) * ldy + ct * 4) = acc;
      }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float v = o[r][2 * q + j][k];
            m[k] = (v > m[k] || v != v) ? v : m[k];
          }
      *reinterpret_cast<f32x4*>(pooled + ((size_t)prow * (W >> 1) + (pw0 >> 1) + q) * ldp + ct * 4) = m;
#pragma unroll
      for (int k = 0; k < 4; ++k) pm = qea_amax_acc(pm, m[k]);
    }
  }
  qea_amax_commit_block(pm, amax_p);
}

// dW[co][tap] = sum_p dy[p][co] * x[p + tap]; db[co] = sum_p dy[p][co].
// thread (rt, ct): 4 channels x (9 taps + bias) fp32 partials over <= ROWS_PER_THREAD pixels,
// fp32 LDS tree over rt, fp64 across blocks (second kernel).
// pixels per thread: 32 ... 128, chosen by c1_wgrad_geom so that large launches amortise the per-block LDS tree and small ones keep the grid full

__global__ __launch_bounds__(256) void conv_c1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int lddy, int B, int H,
                                                            int W, int Co, int rt_n, int rows_per_thread, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float sacc[];  // [rt_n][10][Co]
  const int cols = Co / 4;
  const int ct = threadIdx.x % cols, rt = threadIdx.x / cols;
  const long long M = (long long)B * H * W;
  f32x4 acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (rt < rt_n) {
    const long long r0 = (long long)blockIdx.x * rt_n * rows_per_thread;
    // (ph, pw) of the first pixel by division, then advanced by rt_n pixels per step (no 64-bit divisions in the loop)
    int pw = (int)((r0 + rt) % W);
    int ph = (int)(((r0 + rt) / W) % H);
    const int dph = rt_n / W, dpw = rt_n - dph * W;
    for (int j = 0; j < rows_per_thread; ++j) {
      const long long pix = r0 + (long long)j * rt_n + rt;
      if (pix >= M) break;
      if (j) {
        pw += dpw;
        ph += dph;
        if (pw >= W) {
          pw -= W;
          ++ph;
        }
        ph = ph >= H ? ph % H : ph;
      }
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

// ---------------------------------------------------------------------------------------------
// Round 4 (ABI v8): the WHOLE backward of conv1 -> ReLU -> max_pool2d(2, 2) (models/model_crnn.py:37-38,47-48) from the pooled tensor's
// gradient and the 1-channel INPUT alone.  The full-resolution activation a1 [B, 32, 128, 64] (2.1 GB at B = 2048) was written by the
// forward only for this backward, read by qea_maxpool_bwd, whose 2.1 GB output (three quarters of it zeros) was then read by the
// weight gradient and again by the input gradient: 2.4 ms of a step.  With C_in = 1 the activation costs nine multiply-adds per
// element to rebuild from x (33 MB), so:
//   c1_pool_route_kernel   one lane per 2 x 2 window: rebuilds the four pre-activations per channel with the forward's own fused
//                          multiply-add chain (bias first, taps in kh, kw order: the bits the forward stored and pooled), picks the
//                          winner as the pool did (first maximum in scan order, NaN wins), masks by the ReLU, overwrites the pooled
//                          gradient with the masked value and leaves the winner's position (one byte per (window, channel)); for the
//                          input gradient it also leaves T[tap][pixel] = sum_c dy[pixel][c] w[c][tap] (the per-tap channel sums of
//                          conv_c1_dgrad_tile_kernel) for its four pixels;
//   c1_pool_wgrad_kernel   one lane per channel walks windows: dW[c][tap] += g x[winner + tap], db[c] += g — the x patch of a window is
//                          uniform across the wave (scalar loads), no cross-lane reduction;
//   c1_pool_dx_kernel      dx[p] = sum over the nine taps of T[tap][p + (1 - kh, 1 - kw)].
// ---------------------------------------------------------------------------------------------
template <bool NEED_T>
__global__ __launch_bounds__(256) void c1_pool_route_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ g, int ldg, unsigned* __restrict__ idx, float* __restrict__ T, int B,
                                                            int H, int W) {
  constexpr int Co = 64;
  __shared__ __attribute__((aligned(16))) float Wl[Co * 12];      // [channel][9 taps, bias, 2 pad]
  for (int e = threadIdx.x; e < Co * 12; e += 256) {
    const int c = e / 12, t = e - c * 12;
    Wl[e] = t < 9 ? w[c * 9 + t] : (t == 9 && bias ? bias[c] : 0.f);
  }
  __syncthreads();
  const int OH = H >> 1, OW = W >> 1;
  const long long NW = (long long)B * OH * OW;
  const long long M = (long long)B * H * W;
  for (long long wi = (long long)blockIdx.x * 256 + threadIdx.x; wi < NW; wi += (long long)gridDim.x * 256) {
    const int ow = (int)(wi % OW);
    const int oh = (int)((wi / OW) % OH);
    const long long b = wi / ((long long)OW * OH);
    const int wy = oh * 2, wx = ow * 2;
    const float* xb = x + b * H * W;
    float xp[4][4];                                                // x[wy - 1 .. wy + 2][wx - 1 .. wx + 2], zero outside the image
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int iy = wy + r - 1, ix = wx + c - 1;
        xp[r][c] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? xb[iy * W + ix] : 0.f;
      }
    float Tacc[4][9];
    if (NEED_T) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) Tacc[j][t] = 0.f;
    }
    float* grow = g + wi * ldg;
    unsigned* irow = idx + wi * (Co / 4);
#pragma unroll 2
    for (int c4 = 0; c4 < Co / 4; ++c4) {
      f32x4 gv = *reinterpret_cast<const f32x4*>(grow + c4 * 4);
      unsigned packed = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float* wc = Wl + (c4 * 4 + k) * 12;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wc), w1 = *reinterpret_cast<const f32x4*>(wc + 4), w2 = *reinterpret_cast<const f32x4*>(wc + 8);
        const float wt[9] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0]};
        float m = -INFINITY;
        int arg = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                              // scan order of the pool: (0,0), (0,1), (1,0), (1,1)
          const int jr = j >> 1, jc = j & 1;
          float acc = w2[1];                                       // bias first, then the taps in kh, kw order: the forward's chain
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc = __builtin_fmaf(wt[kh * 3 + kw], xp[jr + kh][jc + kw], acc);
          const float v = fmaxf(acc, 0.f);
          if (v > m || v != v) {
            m = v;
            arg = j;
          }
        }
        const float gm = (m > 0.f) ? gv[k] : 0.f;                  // (qea_maxpool_bwd with relu_mask: a NaN winner routes nothing)
        gv[k] = gm;
        packed |= (unsigned)arg << (8 * k);
        if (NEED_T) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float mj = (arg == j) ? gm : 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) Tacc[j][t] = __builtin_fmaf(mj, wt[t], Tacc[j][t]);
          }
        }
      }
      *reinterpret_cast<f32x4*>(grow + c4 * 4) = gv;
      irow[c4] = packed;
    }
    if (NEED_T) {
      // T[tap][pixel]: consecutive lanes hold consecutive windows of a row -> pairs of floats 8 bytes apart: coalesced float2 stores
      const long long p0 = (b * H + wy) * (long long)W + wx;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        float* Tp = T + (long long)t * M;
        *reinterpret_cast<float2*>(Tp + p0) = make_float2(Tacc[0][t], Tacc[1][t]);
        *reinterpret_cast<float2*>(Tp + p0 + W) = make_float2(Tacc[2][t], Tacc[3][t]);
      }
    }
  }
}

// ws[(block * 4 + wave)][tap * 64 + c] (tap 9 = bias): partial sums of one wave's windows; conv_c1_wgrad_finalize_kernel sums them in fp64.
// A wave walks its windows FOUR AT A TIME (four neighbours of one window row: W / 2 is a multiple of 4): their 4 x 10 input patch is
// uniform over the wave (scalar loads), the four gradient / winner loads of the NEXT group are in flight under this group's arithmetic
// (one window per trip left every trip waiting for its own loads: 1.15 ms at B = 2048 against 0.2 ms of arithmetic).
__global__ __launch_bounds__(256) void c1_pool_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ g, int ldg,
                                                            const unsigned char* __restrict__ idx, int B, int H, int W, int win_per_wave,
                                                            float* __restrict__ ws) {
  constexpr int Co = 64;
  const int c = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int OH = H >> 1, OW = W >> 1;
  const long long NW = (long long)B * OH * OW;
  const long long w0 = ((long long)blockIdx.x * 4 + wave) * win_per_wave;   // (win_per_wave is a multiple of 4)
  long long w1 = w0 + win_per_wave;
  if (w1 > NW) w1 = NW;
  float acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = 0.f;
  float gq[4];
  int aq[4];
  auto load_group = [&](long long wi, float* gv, int* av) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gv[k] = g[(wi + k) * ldg + c];
      av[k] = idx[(wi + k) * Co + c];
    }
  };
  if (w0 < w1) load_group(w0, gq, aq);
  for (long long wi = w0; wi < w1; wi += 4) {                      // (uniform over the wave)
    float gn[4] = {0.f, 0.f, 0.f, 0.f};
    int an[4] = {0, 0, 0, 0};
    if (wi + 4 < w1) load_group(wi + 4, gn, an);
    const int ow = (int)(wi % OW);
    const int oh = (int)((wi / OW) % OH);
    const long long b = wi / ((long long)OW * OH);
    const int wy = oh * 2, wx = ow * 2;
    const float* xb = x + b * H * W;
    float xr[4][10];                                               // rows wy - 1 .. wy + 2, columns wx - 1 .. wx + 8
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int cc = 0; cc < 10; ++cc) {
        const int iy = wy + r - 1, ix = wx + cc - 1;
        xr[r][cc] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? xb[iy * W + ix] : 0.f;
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gm = gq[k];
      const int arg = aq[k];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float mj = (arg == j) ? gm : 0.f;
        const int jr = j >> 1, jc = j & 1;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = __builtin_fmaf(mj, xr[jr + kh][2 * k + jc + kw], acc[kh * 3 + kw]);
      }
      acc[9] += gm;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gq[k] = gn[k];
      aq[k] = an[k];
    }
  }
  float* o = ws + ((size_t)blockIdx.x * 4 + wave) * 10 * Co;
#pragma unroll
  for (int t = 0; t < 10; ++t) o[t * Co + c] = acc[t];
}

__global__ void c1_pool_dx_kernel(const float* __restrict__ T, float* __restrict__ dx, int B, int H, int W, int accumulate) {
  const long long M = (long long)B * H * W;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < M; p += (long long)gridDim.x * blockDim.x) {
    const int px = (int)(p % W);
    const int py = (int)((p / W) % H);
    float s = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int qy = py + 1 - kh;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int qx = px + 1 - kw;
        if ((unsigned)qy < (unsigned)H && (unsigned)qx < (unsigned)W) s += T[(long long)(kh * 3 + kw) * M + p + (long long)(1 - kh) * W + (1 - kw)];
      }
    }
    dx[p] = accumulate ? dx[p] + s : s;
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

// Input gradient, tiled form: a workgroup owns an 8 x 64 pixel tile.  Phase 1: every pixel of the 10 x 66 halo reads its dy row ONCE
// (COLS lanes x float4) and leaves the nine per-tap channel sums T[pixel][tap] = sum_co dy[pixel][co] w[co][tap] in LDS; phase 2: an
// output pixel adds its nine neighbours' T.  (The per-pixel kernel above reads every dy row nine times through L1 and divides 64-bit
// indices per element: 1.2 ms for the CRNN's first layer at B = 2048 against 0.45 ms of HBM time.)
constexpr int C1D_TH = 8, C1D_TW = 64, C1D_HW = C1D_TW + 2, C1D_HP = (C1D_TH + 2) * C1D_HW;
template <int COLS>
__global__ __launch_bounds__(256) void conv_c1_dgrad_tile_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ w,
                                                                 float* __restrict__ dx, int H, int W, int tiles_x, int tiles_y, int accumulate) {
  __shared__ float T[C1D_HP * 9];
  __shared__ __attribute__((aligned(16))) float Wl[COLS * 9 * 4];   // [float4 column][tap][4 channels]
  for (int e = threadIdx.x; e < COLS * 36; e += 256) {
    const int k = e & 3, t = (e >> 2) % 9, c4 = e / 36;
    Wl[e] = w[(c4 * 4 + k) * 9 + t];
  }
  int bid = blockIdx.x;
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int x0 = tx * C1D_TW, y0 = ty * C1D_TH;
  const float* dyb = dy + (size_t)b * H * W * lddy;
  __syncthreads();
  // ONE LANE PER HALO PIXEL: it reads the pixel's whole dy row (COLS float4; the 64 rows of a wave are one contiguous 16 KB range
  // that stays in L1 across the COLS loads) against filter columns broadcast from LDS — no cross-lane reduction (the first tiled
  // form gave a pixel to COLS lanes and all-reduced nine sums through 36 shuffles: 1003 us, hardly better than the 1193 of the
  // per-pixel kernel)
  for (int hq = threadIdx.x; hq < C1D_HP; hq += 256) {
    const int hy = hq / C1D_HW, hx = hq - hy * C1D_HW;
    const int iy = y0 + hy - 1, ix = x0 + hx - 1;
    float d[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) d[t] = 0.f;
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
      const float* row = dyb + ((size_t)iy * W + ix) * lddy;
#pragma unroll 4
      for (int c4 = 0; c4 < COLS; ++c4) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(row + c4 * 4);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(Wl + (c4 * 9 + t) * 4);
          d[t] += g[0] * wv[0] + g[1] * wv[1] + g[2] * wv[2] + g[3] * wv[3];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) T[hq * 9 + t] = d[t];
  }
  __syncthreads();
  float* dxb = dx + (size_t)b * H * W;
  for (int q = threadIdx.x; q < C1D_TH * C1D_TW; q += 256) {
    const int qy = q / C1D_TW, qx = q - qy * C1D_TW;
    const int oy = y0 + qy, ox = x0 + qx;
    if (oy >= H || ox >= W) continue;
    float sacc = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) sacc += T[((qy + 2 - kh) * C1D_HW + qx + 2 - kw) * 9 + kh * 3 + kw];   // halo pixel (qy + 1 + (1 - kh), qx + 1 + (1 - kw))
    float* o = dxb + (size_t)oy * W + ox;
    *o = accumulate ? *o + sacc : sacc;
  }
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
  if (W % 4 == 0 && (long long)B * H * W < 0x7fffffffLL && (Co == 32 || Co == 64 || Co == 128) && ((uintptr_t)y & 15) == 0 &&
      (!bias || ((uintptr_t)bias & 15) == 0)) {
    const int cols = Co / 4;
    const long long groups = (long long)B * H * (W / 4);
    const int grid = (int)((groups + 256 / cols - 1) / (256 / cols) > 16384 ? 16384 : (groups + 256 / cols - 1) / (256 / cols));
    hipStream_t s = (hipStream_t)stream;
    if (Co == 32) hipLaunchKernelGGL(conv_c1_fwd4_kernel<8>, dim3(grid), dim3(256), 0, s, x, w, bias, y, ldy, B, H, W, relu);
    else if (Co == 64) hipLaunchKernelGGL(conv_c1_fwd4_kernel<16>, dim3(grid), dim3(256), 0, s, x, w, bias, y, ldy, B, H, W, relu);
    else hipLaunchKernelGGL(conv_c1_fwd4_kernel<32>, dim3(grid), dim3(256), 0, s, x, w, bias, y, ldy, B, H, W, relu);
    QEA_CHECK_LAUNCH();
    return QEA_OK;
  }
  hipLaunchKernelGGL(conv_c1_fwd_kernel, dim3(grid_for(n, 8192)), dim3(256), (size_t)Co * 9 * sizeof(float), (hipStream_t)stream, x, w, bias, y,
                     ldy, B, H, W, Co, relu);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_conv_c1_fwd_pool(const float* x, const float* w, const float* bias, float* y, int32_t ldy, float* pooled, int32_t ldp, int32_t B,
                                    int32_t H, int32_t W, int32_t Co, int32_t relu, float* absmax_pooled, void* stream) {
  QEA_REQUIRE(x && w && pooled && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 4 == 0 && (Co == 32 || Co == 64 || Co == 128) && ldy % 4 == 0 &&
                  ldy >= Co && ldp % 4 == 0 && ldp >= Co && (long long)B * H * W < 0x7fffffffLL,
              "qea_conv_c1_fwd_pool: needs H %% 2 == 0, W %% 4 == 0, Co in {32, 64, 128}, fewer than 2^31 pixels");
  QEA_REQUIRE(((uintptr_t)y & 15) == 0 && ((uintptr_t)pooled & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0), "qea_conv_c1_fwd_pool: 16-byte alignment");
  const int cols = Co / 4;
  const long long groups = (long long)B * (H / 2) * (W / 4);
  const long long blocks = (groups + 256 / cols - 1) / (256 / cols);
  const int grid = (int)(blocks > 16384 ? 16384 : blocks);
  hipStream_t s = (hipStream_t)stream;
  if (Co == 32) hipLaunchKernelGGL(conv_c1_fwd4_pool_kernel<8>, dim3(grid), dim3(256), 0, s, x, w, bias, y, ldy, pooled, ldp, B, H, W, relu, absmax_pooled);
  else if (Co == 64) hipLaunchKernelGGL(conv_c1_fwd4_pool_kernel<16>, dim3(grid), dim3(256), 0, s, x, w, bias, y, ldy, pooled, ldp, B, H, W, relu, absmax_pooled);
  else hipLaunchKernelGGL(conv_c1_fwd4_pool_kernel<32>, dim3(grid), dim3(256), 0, s, x, w, bias, y, ldy, pooled, ldp, B, H, W, relu, absmax_pooled);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

static int c1_wgrad_geom(long long M, int Co, int* rt_n, int* rows_per_thread) {
  const int cols = Co / 4;
  *rt_n = 256 / cols;
  long long rpt = M / ((long long)(*rt_n) * 4096);         // about 4096 blocks when there is that much work
  rpt = rpt < 32 ? 32 : (rpt > 128 ? 128 : rpt);
  *rows_per_thread = (int)rpt;
  const long long per_block = (long long)(*rt_n) * rpt;
  return (int)((M + per_block - 1) / per_block);
}

extern "C" size_t qea_conv_c1_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Co) {
  if (B <= 0 || H <= 0 || W <= 0 || Co <= 0 || Co % 4 || Co / 4 > 256) return 0;
  int rt, rpt;
  const int grid = c1_wgrad_geom((long long)B * H * W, Co, &rt, &rpt);
  return (size_t)grid * 10 * Co * sizeof(float);
}

extern "C" int qea_conv_c1_wgrad(const float* x, const float* dy, int32_t lddy, float* dw, float* db, int32_t B, int32_t H, int32_t W,
                                 int32_t Co, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  QEA_REQUIRE(x && dy && dw && B > 0 && H > 0 && W > 0 && Co > 0 && Co % 4 == 0 && Co / 4 <= 256 && lddy % 4 == 0, "qea_conv_c1_wgrad: bad arguments");
  int rt, rpt;
  const int grid = c1_wgrad_geom((long long)B * H * W, Co, &rt, &rpt);
  QEA_REQUIRE(workspace && workspace_bytes >= (size_t)grid * 10 * Co * sizeof(float), "qea_conv_c1_wgrad: workspace too small");
  const size_t lds = (size_t)rt * 10 * Co * sizeof(float);
  QEA_REQUIRE(lds <= 160 * 1024, "qea_conv_c1_wgrad: Co too large for LDS");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_c1_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(conv_c1_wgrad_kernel, dim3(grid), dim3(256), lds, s, x, dy, lddy, B, H, W, Co, rt, rpt, (float*)workspace);
  hipLaunchKernelGGL(conv_c1_wgrad_finalize_kernel, dim3(qea_cdiv(10 * Co, 4)), dim3(256), 0, s, (const float*)workspace, grid, Co, dw, db,
                     accumulate);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

// workspace: [winner bytes: B*(H/2)*(W/2)*64, rounded up to 256][T: 9*B*H*W floats when dx is wanted][wgrad partials: blocks*4 rows of 640 floats]
static long long c1pb_win_per_wave(long long NW) {
  long long wpw = (NW + 4 * 2048 - 1) / (4 * 2048);                // about 2048 workgroups of four waves
  wpw = wpw < 16 ? 16 : wpw;
  return (wpw + 3) & ~3LL;                                         // whole groups of four windows
}

extern "C" size_t qea_conv_c1_pool_bwd_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Co, int32_t need_dx) {
  if (B <= 0 || H <= 0 || W <= 0 || Co != 64 || H % 2 || W % 8) return 0;
  const long long NW = (long long)B * (H / 2) * (W / 2);
  const long long wpw = c1pb_win_per_wave(NW);
  const long long blocks = (NW + 4 * wpw - 1) / (4 * wpw);
  size_t b = ((size_t)NW * 64 + 255) & ~(size_t)255;
  if (need_dx) b += (size_t)9 * B * H * W * sizeof(float);
  b += (size_t)blocks * 4 * 640 * sizeof(float);
  return b;
}

extern "C" int qea_conv_c1_pool_bwd(const float* x, const float* w, const float* bias, float* dpool, int32_t lddp, float* dw, float* db, float* dx,
                                    int32_t B, int32_t H, int32_t W, int32_t Co, int32_t accumulate, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  QEA_REQUIRE(x && w && dpool && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 8 == 0 && Co == 64 && lddp % 4 == 0 && lddp >= Co &&
                  ((uintptr_t)dpool & 15) == 0 && ((uintptr_t)x & 7) == 0 && (long long)B * H * W < 0x7fffffffLL,
              "qea_conv_c1_pool_bwd: needs Co == 64, even H, W %% 8 == 0, 16-byte aligned dpool rows, fewer than 2^31 pixels");
  const size_t need = qea_conv_c1_pool_bwd_workspace_bytes(B, H, W, Co, dx != nullptr);
  QEA_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace & 255) == 0, "qea_conv_c1_pool_bwd: workspace of %zu bytes (256-byte aligned) required", need);
  const long long NW = (long long)B * (H / 2) * (W / 2);
  const long long wpw = c1pb_win_per_wave(NW);
  const long long blocks = (NW + 4 * wpw - 1) / (4 * wpw);
  char* wsb = (char*)workspace;
  unsigned* idx = (unsigned*)wsb;
  wsb += ((size_t)NW * 64 + 255) & ~(size_t)255;
  float* T = nullptr;
  if (dx) {
    T = (float*)wsb;
    wsb += (size_t)9 * B * H * W * sizeof(float);
  }
  float* part = (float*)wsb;
  hipStream_t s = (hipStream_t)stream;
  const int rgrid = grid_for(NW, 8192);
  if (dx) hipLaunchKernelGGL(c1_pool_route_kernel<true>, dim3(rgrid), dim3(256), 0, s, x, w, bias, dpool, lddp, idx, T, B, H, W);
  else hipLaunchKernelGGL(c1_pool_route_kernel<false>, dim3(rgrid), dim3(256), 0, s, x, w, bias, dpool, lddp, idx, T, B, H, W);
  if (dw) {
    hipLaunchKernelGGL(c1_pool_wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, (const float*)dpool, lddp, (const unsigned char*)idx, B, H, W,
                       (int)wpw, part);
    hipLaunchKernelGGL(conv_c1_wgrad_finalize_kernel, dim3(qea_cdiv(10 * Co, 4)), dim3(256), 0, s, (const float*)part, (int)(blocks * 4), Co, dw, db,
                       accumulate);
  }
  if (dx) hipLaunchKernelGGL(c1_pool_dx_kernel, dim3(grid_for((long long)B * H * W, 8192)), dim3(256), 0, s, (const float*)T, dx, B, H, W, 0);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_conv_c1_dgrad(const float* dy, int32_t lddy, const float* w, float* dx, int32_t B, int32_t H, int32_t W, int32_t Co,
                                 int32_t accumulate, void* stream) {
  QEA_REQUIRE(dy && w && dx && B > 0 && H > 0 && W > 0 && lddy % 4 == 0, "qea_conv_c1_dgrad: bad arguments");
  const long long n = (long long)B * H * W * (Co / 4);
  const int grid = grid_for(n, 8192);
  hipStream_t s = (hipStream_t)stream;
  const int tiles_x = qea_cdiv(W, C1D_TW), tiles_y = qea_cdiv(H, C1D_TH);
  const long long tiles = (long long)B * tiles_x * tiles_y;
  if (tiles < 0x7fffffffLL && (Co == 32 || Co == 64 || Co == 128) && ((uintptr_t)dy & 15) == 0) {
    if (Co == 32) hipLaunchKernelGGL(conv_c1_dgrad_tile_kernel<8>, dim3((unsigned)tiles), dim3(256), 0, s, dy, lddy, w, dx, H, W, tiles_x, tiles_y, accumulate);
    else if (Co == 64) hipLaunchKernelGGL(conv_c1_dgrad_tile_kernel<16>, dim3((unsigned)tiles), dim3(256), 0, s, dy, lddy, w, dx, H, W, tiles_x, tiles_y, accumulate);
    else hipLaunchKernelGGL(conv_c1_dgrad_tile_kernel<32>, dim3((unsigned)tiles), dim3(256), 0, s, dy, lddy, w, dx, H, W, tiles_x, tiles_y, accumulate);
    QEA_CHECK_LAUNCH();
    return QEA_OK;
  }
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
