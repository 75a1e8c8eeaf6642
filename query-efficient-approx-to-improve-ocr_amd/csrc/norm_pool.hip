// HBM-bound NHWC kernels between the convolutions: BatchNorm (batch statistics in fp64,
// apply + ReLU, backward), max-pool forward/backward (+ fused ReLU mask), per-channel column
// sums (bias gradients), 2-D transposes / filter flips for the gradient GEMMs.
// All of them stream float4 per lane over [pixels][channels] rows; roofline = HBM (~6.3 TB/s
// achievable), algorithmic bytes = 4 * (tensors read + tensors written) * M * C.
#include "common.h"

namespace {

constexpr int RED_THREADS = 256;

// ------------------------------------------------------------------ column reductions
// Each block owns a contiguous range of rows; thread (rt, ct) accumulates 4 channels
// (one float4 column) over rows rt, rt+RT, ... in fp64; LDS tree over rt; partial -> ws.
struct ColGeom {
  int cols;  // C/4 float4 columns
  int rt;    // row-threads per block
  int grid;
  int rows_per_block;
};

ColGeom col_geom(long long M, int C) {
  ColGeom g;
  g.cols = C / 4;
  g.rt = RED_THREADS / g.cols;
  if (g.rt < 1) g.rt = 1;
  long long want = (M + (long long)g.rt * 16 - 1) / ((long long)g.rt * 16);
  if (want > 1024) want = 1024;
  if (want < 1) want = 1;
  g.grid = (int)want;
  g.rows_per_block = (int)((M + g.grid - 1) / g.grid);
  g.grid = (int)((M + g.rows_per_block - 1) / g.rows_per_block);
  return g;
}

// The BatchNorm output before the ReLU, evaluated in ONE place so that the forward (bn_apply) and the mask the
// backward recomputes from y agree bit for bit.
__device__ __forceinline__ float bn_affine(float v, float sc, float sh) { return __fmaf_rn(v, sc, sh); }

// MODE 0: s0 = sum x, s1 = sum x^2                                 (BN statistics)
// MODE 1: dz = da * relu'(.); s0 = sum dz, s1 = sum dz * (y - mean) * invstd   (BN backward)
//         relu' from a > 0 when a is given, else from bn_affine(y, msc, msh) > 0 when msc is given (no read of a)
// MODE 2: s0 = sum x                                               (bias gradient)
template <int MODE>
__global__ __launch_bounds__(RED_THREADS) void colreduce_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ a,
                                                                  int lda, const float* __restrict__ y, int ldy,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  const double* __restrict__ stat64, const float* __restrict__ msc,
                                                                  const float* __restrict__ msh, long long M, int C,
                                                                  int rows_per_block, int rt_n, double* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) double sred[];  // [rt_n][cols*4][2]
  const int cols = C / 4;
  const int ct = threadIdx.x % cols, rt = threadIdx.x / cols;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  if (rt < rt_n) {
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    double mu[4] = {0, 0, 0, 0}, is[4] = {0, 0, 0, 0};
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        mu[k] = stat64 ? stat64[ct * 4 + k] : (double)mean[ct * 4 + k];
        is[k] = stat64 ? stat64[C + ct * 4 + k] : (double)invstd[ct * 4 + k];
      }
      if (!a && msc) {
        sc = *reinterpret_cast<const f32x4*>(msc + ct * 4);
        sh = *reinterpret_cast<const f32x4*>(msh + ct * 4);
      }
    }
#pragma unroll 2
    for (long long r = r0 + rt; r < r1; r += rt_n) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + ct * 4);
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s0[k] += (double)v[k];
          s1[k] += (double)v[k] * (double)v[k];
        }
      } else if (MODE == 1) {
        f32x4 dz = v;
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + r * ldy + ct * 4);
        if (a) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(a + r * lda + ct * 4);
#pragma unroll
          for (int k = 0; k < 4; ++k) dz[k] = av[k] > 0.f ? dz[k] : 0.f;
        } else if (msc) {
#pragma unroll
          for (int k = 0; k < 4; ++k) dz[k] = bn_affine(yv[k], sc[k], sh[k]) > 0.f ? dz[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s0[k] += (double)dz[k];
          s1[k] += (double)dz[k] * (((double)yv[k] - mu[k]) * is[k]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) s0[k] += (double)v[k];
      }
    }
  }
  if (rt < rt_n) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sred[((size_t)rt * C + ct * 4 + k) * 2 + 0] = s0[k];
      sred[((size_t)rt * C + ct * 4 + k) * 2 + 1] = s1[k];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double t0 = 0, t1 = 0;
    for (int r = 0; r < rt_n; ++r) {
      t0 += sred[((size_t)r * C + c) * 2 + 0];
      t1 += sred[((size_t)r * C + c) * 2 + 1];
    }
    ws[((size_t)blockIdx.x * C + c) * 2 + 0] = t0;
    ws[((size_t)blockIdx.x * C + c) * 2 + 1] = t1;
  }
}

// Finalize kernels: ONE WAVE PER CHANNEL.  Lane l sums partial blocks l, l+64, ... then a wave
// reduction; a single thread walking up to 1024 dependent L2 loads per channel took ~200 us.
__device__ __forceinline__ void partial_sums(const double* __restrict__ ws, int nblk, int C, int c, double& s, double& q) {
  const int lane = threadIdx.x & 63;
  double a = 0, b = 0;
  for (int k = lane; k < nblk; k += 64) {
    a += ws[((size_t)k * C + c) * 2 + 0];
    b += ws[((size_t)k * C + c) * 2 + 1];
  }
  s = qea_wave_sum_d(a);
  q = qea_wave_sum_d(b);
}

__global__ void bn_stats_finalize_kernel(const double* __restrict__ ws, int nblk, int C, long long M, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, float eps, float momentum, float* running_mean,
                                         float* running_var, float* mean_out, float* invstd_out, float* scale_out, float* shift_out,
                                         double* stat64) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= C) return;
  double s, q;
  partial_sums(ws, nblk, C, c, s, q);
  if ((threadIdx.x & 63) != 0) return;
  const double mean = s / (double)M;
  double var = q / (double)M - mean * mean;  // biased (normalisation)
  if (var < 0) var = 0;
  const double invstd_d = 1.0 / sqrt(var + (double)eps);
  if (stat64) {
    stat64[c] = mean;
    stat64[C + c] = invstd_d;
  }
  const float invstd = (float)invstd_d;
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  mean_out[c] = (float)mean;
  invstd_out[c] = invstd;
  scale_out[c] = g * invstd;
  shift_out[c] = b - (float)mean * g * invstd;
  if (running_mean) {
    const double unb = (M > 1) ? var * (double)M / (double)(M - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

__global__ void bn_eval_coeff_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                                     const float* __restrict__ conv_bias, float* mean_out, float* invstd_out, float* scale_out,
                                     float* shift_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(rvar[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  mean_out[c] = rmean[c];
  invstd_out[c] = invstd;
  scale_out[c] = g * invstd;
  shift_out[c] = b + ((conv_bias ? conv_bias[c] : 0.f) - rmean[c]) * g * invstd;
}

__global__ void colsum_finalize_kernel(const double* __restrict__ ws, int nblk, int C, float* out, int accumulate) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= C) return;
  double s, q;
  partial_sums(ws, nblk, C, c, s, q);
  if ((threadIdx.x & 63) != 0) return;
  out[c] = accumulate ? out[c] + (float)s : (float)s;
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ ws, int nblk, int C, long long M, const float* __restrict__ gamma,
                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                       const double* __restrict__ stat64, int training, float* dgamma, float* dbeta, int accumulate,
                                       double* k0, double* k1, double* k2) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= C) return;
  double s, q;
  partial_sums(ws, nblk, C, c, s, q);
  if ((threadIdx.x & 63) != 0) return;
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)q : (float)q;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
  // dy = g*invstd * (dz - mean(dz) - xhat * mean(dz*xhat)),  xhat = (y - mu) * invstd
  //    = k0 * dz + k1 * y + k2     (fp64 per-channel constants, see bn_bwd_apply_kernel)
  const double g = gamma ? (double)gamma[c] : 1.0;
  const double is = stat64 ? stat64[C + c] : (double)invstd[c];
  const double mu = stat64 ? stat64[c] : (double)mean[c];
  const double m0 = training ? s / (double)M : 0.0;
  const double m1 = training ? q / (double)M : 0.0;
  k0[c] = g * is;
  k1[c] = -g * is * is * m1;
  k2[c] = g * is * (mu * is * m1 - m0);
}

// ------------------------------------------------------------------ elementwise
__global__ void bn_apply_kernel(const float* __restrict__ y, int ldy, float* __restrict__ a, int lda, long long M, int C,
                                const float* __restrict__ scale, const float* __restrict__ shift, int relu, float* __restrict__ amax) {
  const int cols = C / 4;
  const long long n = M * cols;
  float am = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols;
    const int ct = (int)(i - r * cols);
    const f32x4 v = *reinterpret_cast<const f32x4*>(y + r * ldy + ct * 4);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + ct * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + ct * 4);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[k] = bn_affine(v[k], sc[k], sh[k]);
      if (relu) o[k] = fmaxf(o[k], 0.f);
      am = qea_amax_acc(am, o[k]);
    }
    *reinterpret_cast<f32x4*>(a + r * lda + ct * 4) = o;
  }
  qea_amax_commit_block(am, amax);
}

__global__ __launch_bounds__(RED_THREADS) void bn_bwd_apply_kernel(const float* __restrict__ da, int ldda, const float* __restrict__ a, int lda,
                                                                    const float* __restrict__ y, int ldy, float* __restrict__ dy, int lddy,
                                                                    long long M, int C, const float* __restrict__ msc,
                                                                    const float* __restrict__ msh, const double* __restrict__ k0,
                                                                    const double* __restrict__ k1, const double* __restrict__ k2,
                                                                    int rows_per_block, int rt_n, float* __restrict__ amax) {
  // dy = k0*dz + k1*y + k2 in fp64 with fp64 per-channel constants: mean(dz), mean(dz*xhat), mean and invstd are
  // common to all pixels of a channel, so rounding them to fp32 puts a CORRELATED error into dy which the
  // per-channel sums of the next layer amplify by the pixel count (measured 7e-4 on dbeta at M = 8192); ATen's
  // CPU kernel also runs this in its fp64 accumulate type.  A thread keeps ONE float4 channel column (constants in
  // registers) and walks rows, two per iteration to keep more bytes in flight.
  const int cols = C / 4;
  const int ct = threadIdx.x % cols, rt = threadIdx.x / cols;
  const bool idle = rt >= rt_n;                            // (no early return: every lane meets in qea_amax_commit)
  float am = 0.f;
  double c0[4], c1[4], c2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c0[k] = k0[ct * 4 + k];
    c1[k] = k1[ct * 4 + k];
    c2[k] = k2[ct * 4 + k];
  }
  f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
  const bool remask = !a && msc;
  if (remask) {
    sc = *reinterpret_cast<const f32x4*>(msc + ct * 4);
    sh = *reinterpret_cast<const f32x4*>(msh + ct * 4);
  }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  auto one = [&](long long r, f32x4 dz, const f32x4 yv, const f32x4 av) {
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float d = dz[k];
      if (a) d = av[k] > 0.f ? d : 0.f;
      else if (remask) d = bn_affine(yv[k], sc[k], sh[k]) > 0.f ? d : 0.f;
      o[k] = (float)(c0[k] * (double)d + (c1[k] * (double)yv[k] + c2[k]));
      am = qea_amax_acc(am, o[k]);
    }
    *reinterpret_cast<f32x4*>(dy + r * lddy + ct * 4) = o;
  };
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  long long r = idle ? r1 : r0 + rt;
  for (; r + rt_n < r1; r += 2 * rt_n) {
    const long long rb = r + rt_n;
    const f32x4 dza = *reinterpret_cast<const f32x4*>(da + r * ldda + ct * 4);
    const f32x4 dzb = *reinterpret_cast<const f32x4*>(da + rb * ldda + ct * 4);
    const f32x4 ya = *reinterpret_cast<const f32x4*>(y + r * ldy + ct * 4);
    const f32x4 yb = *reinterpret_cast<const f32x4*>(y + rb * ldy + ct * 4);
    f32x4 aa = zero, ab = zero;
    if (a) {
      aa = *reinterpret_cast<const f32x4*>(a + r * lda + ct * 4);
      ab = *reinterpret_cast<const f32x4*>(a + rb * lda + ct * 4);
    }
    one(r, dza, ya, aa);
    one(rb, dzb, yb, ab);
  }
  if (r < r1) {
    const f32x4 dza = *reinterpret_cast<const f32x4*>(da + r * ldda + ct * 4);
    const f32x4 ya = *reinterpret_cast<const f32x4*>(y + r * ldy + ct * 4);
    const f32x4 aa = a ? *reinterpret_cast<const f32x4*>(a + r * lda + ct * 4) : zero;
    one(r, dza, ya, aa);
  }
  qea_amax_commit_block(am, amax);
}

// ---------------------------------------------------------------------------------------------
// Round 4 (ABI v8): the max-pool backward INSIDE the BatchNorm(+ReLU) backward that follows it.  A block's output a = relu(bn(y)) goes to
// a kh x kw max-pool (kh = 2, kw in {1, 2}; model_unet.py:52-59, model_crnn.py:50-54) and, in the UNet, to a skip connection: the
// gradient that reaches the BatchNorm is da = dskip + maxpool_bwd(dpool).  qea_maxpool_bwd made that sum in a pass of its own (read a,
// read dpool, read + write dskip: 3.25 tensor passes); here both kernels of the BatchNorm backward rebuild it per WINDOW: the four
// (two) activations of a window are recomputed from y with the forward's own fused multiply-add (bn_affine + max(., 0): bit for bit
// what bn_apply stored and what the pool saw), the winner is the first maximum in (kh, kw) scan order, NaN wins (PyTorch's rule, as
// maxpool_fwd / maxpool_bwd), and dz = relu'(a) * (da + [winner] dpool).  A thread keeps one float4 channel column and walks windows.
// Values: identical to qea_maxpool_bwd(accumulate) followed by qea_bn_bwd (same fp32 add, same mask); the two fp64 reductions visit the
// pixels in another order (equal to ~1e-16 relative).
// ---------------------------------------------------------------------------------------------
template <int KW, bool APPLY>
__global__ __launch_bounds__(RED_THREADS) void bn_bwd_pool_kernel(const float* __restrict__ da, int ldda, const float* __restrict__ dpool, int lddp,
                                                                   const float* __restrict__ y, int ldy, float* __restrict__ dy, int lddy, int B, int H,
                                                                   int W, int C, const float* __restrict__ msc, const float* __restrict__ msh,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   const double* __restrict__ stat64, const double* __restrict__ k0,
                                                                   const double* __restrict__ k1, const double* __restrict__ k2,
                                                                   int win_per_block, int rt_n, double* __restrict__ ws, float* __restrict__ amax) {
  extern __shared__ __attribute__((aligned(16))) double sred[];  // reduction form: [rt_n][C][2]
  constexpr int NP = 2 * KW;
  const int cols = C / 4;
  const int ct = threadIdx.x % cols, rt = threadIdx.x / cols;
  const int OH = H / 2, OW = W / KW;
  const long long NWIN = (long long)B * OH * OW;
  const bool idle = rt >= rt_n;
  float am = 0.f;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  double mu[4] = {0, 0, 0, 0}, is[4] = {0, 0, 0, 0}, c0[4] = {0, 0, 0, 0}, c1[4] = {0, 0, 0, 0}, c2[4] = {0, 0, 0, 0};
  const f32x4 sc = *reinterpret_cast<const f32x4*>(msc + ct * 4);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(msh + ct * 4);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (APPLY) {
      c0[k] = k0[ct * 4 + k];
      c1[k] = k1[ct * 4 + k];
      c2[k] = k2[ct * 4 + k];
    } else {
      mu[k] = stat64 ? stat64[ct * 4 + k] : (double)mean[ct * 4 + k];
      is[k] = stat64 ? stat64[C + ct * 4 + k] : (double)invstd[ct * 4 + k];
    }
  }
  const long long w0 = (long long)blockIdx.x * win_per_block;
  long long w1 = w0 + win_per_block;
  if (w1 > NWIN) w1 = NWIN;
  for (long long wi = idle ? w1 : w0 + rt; wi < w1; wi += rt_n) {
    const int ow = (int)(wi % OW);
    const int oh = (int)((wi / OW) % OH);
    const int b = (int)(wi / ((long long)OW * OH));
    const long long r00 = ((long long)b * H + 2 * oh) * W + (long long)ow * KW;
    long long rows[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) rows[j] = r00 + (j / KW) * (long long)W + (j % KW);
    f32x4 yv[NP], dv[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      yv[j] = *reinterpret_cast<const f32x4*>(y + rows[j] * ldy + ct * 4);
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      dv[j] = da ? *reinterpret_cast<const f32x4*>(da + rows[j] * ldda + ct * 4) : zero;
    }
    const f32x4 g = *reinterpret_cast<const f32x4*>(dpool + wi * lddp + ct * 4);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int arg[4] = {0, 0, 0, 0};
    f32x4 av[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v = fmaxf(bn_affine(yv[j][k], sc[k], sh[k]), 0.f);
        av[j][k] = v;
        if (v > m[k] || v != v) {
          m[k] = v;
          arg[k] = j;
        }
      }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float d = (arg[k] == j ? g[k] : 0.f) + dv[j][k];      // (qea_maxpool_bwd's accumulate: routed gradient + what the skip path left)
        d = av[j][k] > 0.f ? d : 0.f;
        if (APPLY) {
          o[k] = (float)(c0[k] * (double)d + (c1[k] * (double)yv[j][k] + c2[k]));
          am = qea_amax_acc(am, o[k]);
        } else {
          s0[k] += (double)d;
          s1[k] += (double)d * (((double)yv[j][k] - mu[k]) * is[k]);
        }
      }
      if (APPLY) *reinterpret_cast<f32x4*>(dy + rows[j] * lddy + ct * 4) = o;
    }
  }
  if constexpr (APPLY) {
    qea_amax_commit_block(am, amax);
  } else {
    if (!idle) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sred[((size_t)rt * C + ct * 4 + k) * 2 + 0] = s0[k];
        sred[((size_t)rt * C + ct * 4 + k) * 2 + 1] = s1[k];
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      double t0 = 0, t1 = 0;
      for (int r = 0; r < rt_n; ++r) {
        t0 += sred[((size_t)r * C + c) * 2 + 0];
        t1 += sred[((size_t)r * C + c) * 2 + 1];
      }
      ws[((size_t)blockIdx.x * C + c) * 2 + 0] = t0;
      ws[((size_t)blockIdx.x * C + c) * 2 + 1] = t1;
    }
  }
}

// max-pool with window == stride (2x2 or 2x1), PyTorch tie rule: first maximum in (kh,kw) scan order
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int B, int H, int W, int C,
                                   int kh, int kw, float* __restrict__ amax) {
  const int OH = H / kh, OW = W / kw, cols = C / 4;
  const long long n = (long long)B * OH * OW * cols;
  float am = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int ct = (int)(i % cols);
    const long long op = i / cols;
    const int ow = (int)(op % OW);
    const int oh = (int)((op / OW) % OH);
    const int b = (int)(op / ((long long)OW * OH));
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int i2 = 0; i2 < kh; ++i2)
      for (int j2 = 0; j2 < kw; ++j2) {
        const long long ip = ((long long)b * H + oh * kh + i2) * W + ow * kw + j2;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + ip * ldx + ct * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = (v[k] > m[k] || v[k] != v[k]) ? v[k] : m[k];
      }
    *reinterpret_cast<f32x4*>(y + op * ldy + ct * 4) = m;
#pragma unroll
    for (int k = 0; k < 4; ++k) am = qea_amax_acc(am, m[k]);
  }
  qea_amax_commit_block(am, amax);
}

// BatchNorm apply (+ReLU) AND the max-pool that follows it, one pass (model_unet.py:51-59: enc -> pool): a thread owns
// one pooled pixel x 4 channels, reads its kh x kw window of y once, stores the kh*kw activations (the skip tensor / the operand of the
// pool's backward) and their maximum.  Same fused multiply-add as bn_apply_kernel, same scan order and NaN rule as maxpool_fwd_kernel:
// both outputs are bit-identical to the two separate passes, which read the full-resolution activation a second time.
__global__ void bn_apply_pool_kernel(const float* __restrict__ y, int ldy, float* __restrict__ a, int lda, float* __restrict__ pooled, int ldp,
                                     int B, int H, int W, int C, const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                     int kh, int kw, float* __restrict__ amax_a, float* __restrict__ amax_p) {
  const int OH = H / kh, OW = W / kw, cols = C / 4;
  const long long n = (long long)B * OH * OW * cols;
  float am = 0.f, pm = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int ct = (int)(i % cols);
    const long long op = i / cols;
    const int ow = (int)(op % OW);
    const long long orow = op / OW;                         // b * OH + oh
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + ct * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + ct * 4);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int i2 = 0; i2 < kh; ++i2)
      for (int j2 = 0; j2 < kw; ++j2) {
        const long long ip = (orow * kh + i2) * W + ow * kw + j2;
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + ip * ldy + ct * 4);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          o[k] = bn_affine(v[k], sc[k], sh[k]);
          if (relu) o[k] = fmaxf(o[k], 0.f);
          am = qea_amax_acc(am, o[k]);
          m[k] = (o[k] > m[k] || o[k] != o[k]) ? o[k] : m[k];
        }
        *reinterpret_cast<f32x4*>(a + ip * lda + ct * 4) = o;
      }
    *reinterpret_cast<f32x4*>(pooled + op * ldp + ct * 4) = m;
#pragma unroll
    for (int k = 0; k < 4; ++k) pm = qea_amax_acc(pm, m[k]);
  }
  qea_amax_commit_block(am, amax_a);
  qea_amax_commit_block(pm, amax_p);
}

__global__ void maxpool_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy, float* __restrict__ dx,
                                   int lddx, int B, int H, int W, int C, int kh, int kw, int relu_mask, int accumulate,
                                   float* __restrict__ amax) {
  const int OH = H / kh, OW = W / kw, cols = C / 4;
  const long long n = (long long)B * OH * OW * cols;
  float am = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int ct = (int)(i % cols);
    const long long op = i / cols;
    const int ow = (int)(op % OW);
    const int oh = (int)((op / OW) % OH);
    const int b = (int)(op / ((long long)OW * OH));
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int arg[4] = {0, 0, 0, 0};
    for (int i2 = 0; i2 < kh; ++i2)
      for (int j2 = 0; j2 < kw; ++j2) {
        const long long ip = ((long long)b * H + oh * kh + i2) * W + ow * kw + j2;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + ip * ldx + ct * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (v[k] > m[k] || v[k] != v[k]) {
            m[k] = v[k];
            arg[k] = i2 * kw + j2;
          }
      }
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + op * lddy + ct * 4);
    for (int i2 = 0; i2 < kh; ++i2)
      for (int j2 = 0; j2 < kw; ++j2) {
        const long long ip = ((long long)b * H + oh * kh + i2) * W + ow * kw + j2;
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float t = (arg[k] == i2 * kw + j2) ? g[k] : 0.f;
          if (relu_mask && !(m[k] > 0.f)) t = 0.f;
          o[k] = t;
        }
        f32x4* dst = reinterpret_cast<f32x4*>(dx + ip * lddx + ct * 4);
        if (accumulate) o += *dst;
        *dst = o;
#pragma unroll
        for (int k = 0; k < 4; ++k) am = qea_amax_acc(am, o[k]);
      }
  }
  qea_amax_commit_block(am, amax);
}

// out[c][r] = in[r][c]  (32x32 LDS tiles)
__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc) {
  __shared__ float t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    t[j][tx] = (r < R && c < Cc) ? in[(size_t)r * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;
    if (c < Cc && r < R) out[(size_t)c * R + r] = t[tx][j];
  }
}

// conv weight [Co][KH][KW][Ci] -> input-gradient filter [Ci][KH][KW][Co] with taps flipped
__global__ void flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int Co, int Ci, int KH, int KW) {
  const long long n = (long long)Co * KH * KW * Ci;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    // i indexes wt: (ci, kh', kw', co), co fastest
    const int co = (int)(i % Co);
    long long r = i / Co;
    const int kw2 = (int)(r % KW);
    r /= KW;
    const int kh2 = (int)(r % KH);
    const int ci = (int)(r / KH);
    wt[i] = w[(((size_t)co * KH + (KH - 1 - kh2)) * KW + (KW - 1 - kw2)) * Ci + ci];
  }
}

int grid_for(long long n) {
  long long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

int check_nc(const char* who, long long M, int C, size_t ws_bytes, void* ws, const ColGeom& g) {
  QEA_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C / 4 <= RED_THREADS, "%s: need C %% 4 == 0 and C <= %d (C=%d)", who, 4 * RED_THREADS, C);
  QEA_REQUIRE(ws && ws_bytes >= (size_t)g.grid * C * 2 * sizeof(double), "%s: workspace too small", who);
  return QEA_OK;
}

}  // namespace

extern "C" size_t qea_colreduce_workspace_bytes(int64_t M, int32_t C) {
  if (M <= 0 || C <= 0 || C % 4) return 0;
  const ColGeom g = col_geom(M, C);
  return (size_t)g.grid * C * 2 * sizeof(double) + 3 * (size_t)C * sizeof(double);
}

extern "C" int qea_bn_train_stats(const float* y, int32_t ldy, int64_t M, int32_t C, const float* gamma, const float* beta, float eps,
                                  float momentum, float* running_mean, float* running_var, float* mean_out, float* invstd_out,
                                  float* scale_out, float* shift_out, double* stat64, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  QEA_REQUIRE(y && mean_out && invstd_out && scale_out && shift_out, "qea_bn_train_stats: null pointer");
  const ColGeom g = col_geom(M, C);
  int rc = check_nc("qea_bn_train_stats", M, C, workspace_bytes, workspace, g);
  if (rc) return rc;
  QEA_REQUIRE(ldy % 4 == 0 && ((uintptr_t)y & 15) == 0, "qea_bn_train_stats: alignment");
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)g.rt * C * 2 * sizeof(double);
  hipLaunchKernelGGL(colreduce_kernel<0>, dim3(g.grid), dim3(RED_THREADS), lds, s, y, ldy, nullptr, 0, nullptr, 0, nullptr, nullptr,
                     nullptr, nullptr, nullptr, (long long)M, C, g.rows_per_block, g.rt, (double*)workspace);
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(qea_cdiv(C, 4)), dim3(256), 0, s, (const double*)workspace, g.grid, C,
                     (long long)M, gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, scale_out, shift_out,
                     stat64);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

// First stage for MANY partial blocks (the LDS-halo conv kernels leave one per workgroup row: 131072 at B = 2048 on the
// 32x128 level): workgroup g sums the contiguous block range [g*per, (g+1)*per) for all channels — thread (c2, stripe) walks
// its stripe of the range with 16-byte loads that are contiguous over the channels, an LDS tree joins the stripes — and writes
// row g of `out`.  Fixed ranges and a fixed tree: bit-reproducible.  The wave-per-channel finalize kernel then sums <= 256 rows
// instead of walking thousands of strided loads per lane (250 us -> a few us per BatchNorm layer).
__global__ __launch_bounds__(256) void partials_reduce_kernel(const double* __restrict__ ws, int nblk, int C, int per, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double sp[];   // [stripes][C][2]
  const int stripes = 256 / C > 0 ? 256 / C : 1;               // C <= 256 per pass
  const int b0 = blockIdx.x * per;
  const int b1 = min(nblk, b0 + per);
  for (int cbase = 0; cbase < C; cbase += 256) {
    const int cw = min(256, C - cbase);
    const int st = cw < 256 ? 256 / cw : 1;
    const int c = threadIdx.x % cw, stripe = threadIdx.x / cw;
    double a = 0, b = 0;
    if (stripe < st) {
      for (int k = b0 + stripe; k < b1; k += st) {
        const double* src = ws + ((size_t)k * C + cbase + c) * 2;
        a += src[0];
        b += src[1];
      }
      sp[((size_t)stripe * cw + c) * 2 + 0] = a;
      sp[((size_t)stripe * cw + c) * 2 + 1] = b;
    }
    __syncthreads();
    if (threadIdx.x < cw) {
      double t0 = 0, t1 = 0;
      for (int r = 0; r < st; ++r) {
        t0 += sp[((size_t)r * cw + threadIdx.x) * 2 + 0];
        t1 += sp[((size_t)r * cw + threadIdx.x) * 2 + 1];
      }
      out[((size_t)blockIdx.x * C + cbase + threadIdx.x) * 2 + 0] = t0;
      out[((size_t)blockIdx.x * C + cbase + threadIdx.x) * 2 + 1] = t1;
    }
    __syncthreads();
  }
  (void)stripes;
}

extern "C" int qea_bn_train_stats_from_partials(const double* partials, int32_t blocks, int64_t M, int32_t C, const float* gamma,
                                                const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                                float* mean_out, float* invstd_out, float* scale_out, float* shift_out, double* stat64,
                                                void* stream) {
  QEA_REQUIRE(partials && blocks > 0 && M > 0 && C > 0 && mean_out && invstd_out && scale_out && shift_out,
              "qea_bn_train_stats_from_partials: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const double* src = partials;
  int nblk = blocks;
  if (blocks > QEA_BN_PARTIAL_SCRATCH_ROWS * 2) {
    // two stages: QEA_BN_PARTIAL_SCRATCH_ROWS ranges -> the scratch rows the caller left behind the partials
    const int per = qea_cdiv(blocks, QEA_BN_PARTIAL_SCRATCH_ROWS);
    nblk = qea_cdiv(blocks, per);
    double* scratch = const_cast<double*>(partials) + (size_t)blocks * C * 2;
    hipLaunchKernelGGL(partials_reduce_kernel, dim3(nblk), dim3(256), (size_t)256 * 2 * sizeof(double), s, partials, blocks, C, per, scratch);
    src = scratch;
  }
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(qea_cdiv(C, 4)), dim3(256), 0, s, src, nblk, C, (long long)M,
                     gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, scale_out, shift_out, stat64);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_bn_eval_coeff(int32_t C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                 float eps, const float* conv_bias, float* mean_out, float* invstd_out, float* scale_out,
                                 float* shift_out, void* stream) {
  QEA_REQUIRE(C > 0 && running_mean && running_var && mean_out && invstd_out && scale_out && shift_out, "qea_bn_eval_coeff: null pointer");
  hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3(qea_cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, C, gamma, beta, running_mean,
                     running_var, eps, conv_bias, mean_out, invstd_out, scale_out, shift_out);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_bn_apply(const float* y, int32_t ldy, float* a, int32_t lda, int64_t M, int32_t C, const float* scale,
                            const float* shift, int32_t relu, float* absmax_out, void* stream) {
  QEA_REQUIRE(y && a && scale && shift && M > 0 && C > 0 && C % 4 == 0 && ldy % 4 == 0 && lda % 4 == 0, "qea_bn_apply: bad arguments");
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, y, ldy, a, lda, (long long)M, C,
                     scale, shift, relu, absmax_out);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_bn_bwd(const float* da, int32_t ldda, const float* a, int32_t lda, const float* relu_scale, const float* relu_shift,
                          const float* y, int32_t ldy, int64_t M, int32_t C, const float* gamma, const float* mean, const float* invstd, const double* stat64, int32_t training,
                          float* dgamma, float* dbeta, int32_t accumulate_param_grads, float* dy, int32_t lddy, void* workspace,
                          size_t workspace_bytes, float* absmax_out, void* stream) {
  QEA_REQUIRE(da && y && mean && invstd && dy, "qea_bn_bwd: null pointer");
  const ColGeom g = col_geom(M, C);
  int rc = check_nc("qea_bn_bwd", M, C, workspace_bytes, workspace, g);
  if (rc) return rc;
  QEA_REQUIRE(workspace_bytes >= qea_colreduce_workspace_bytes(M, C), "qea_bn_bwd: workspace too small");
  QEA_REQUIRE(ldda % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && (!a || lda % 4 == 0), "qea_bn_bwd: strides must be multiples of 4");
  QEA_REQUIRE(!a || !relu_scale, "qea_bn_bwd: give the ReLU mask either as a or as relu_scale/relu_shift, not both");
  QEA_REQUIRE((relu_scale == nullptr) == (relu_shift == nullptr), "qea_bn_bwd: relu_scale and relu_shift go together");
  hipStream_t s = (hipStream_t)stream;
  double* ws = (double*)workspace;
  double* k0 = ws + (size_t)g.grid * C * 2;
  double* k1 = k0 + C;
  double* k2 = k1 + C;
  const size_t lds = (size_t)g.rt * C * 2 * sizeof(double);
  hipLaunchKernelGGL(colreduce_kernel<1>, dim3(g.grid), dim3(RED_THREADS), lds, s, da, ldda, a, lda, y, ldy, mean, invstd, stat64,
                     relu_scale, relu_shift, (long long)M, C, g.rows_per_block, g.rt, ws);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(qea_cdiv(C, 4)), dim3(256), 0, s, (const double*)ws, g.grid, C, (long long)M, gamma,
                     mean, invstd, stat64, training, dgamma, dbeta, accumulate_param_grads, k0, k1, k2);
  // elementwise pass: same column-per-thread geometry, but up to 4096 workgroups (no partials to merge afterwards)
  long long agrid = (M + (long long)g.rt * 8 - 1) / ((long long)g.rt * 8);
  if (agrid > 4096) agrid = 4096;
  if (agrid < 1) agrid = 1;
  const int arows = (int)((M + agrid - 1) / agrid);
  agrid = (M + arows - 1) / arows;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)agrid), dim3(RED_THREADS), 0, s, da, ldda, a, lda, y, ldy, dy, lddy, (long long)M,
                     C, relu_scale, relu_shift, (const double*)k0, (const double*)k1, (const double*)k2, arows, g.rt, absmax_out);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_bn_bwd_pool(const float* da, int32_t ldda, const float* dpool, int32_t lddp, int32_t kw, const float* relu_scale,
                               const float* relu_shift, const float* y, int32_t ldy, int32_t B, int32_t H, int32_t W, int32_t C,
                               const float* gamma, const float* mean, const float* invstd, const double* stat64, int32_t training, float* dgamma,
                               float* dbeta, int32_t accumulate_param_grads, float* dy, int32_t lddy, void* workspace, size_t workspace_bytes,
                               float* absmax_out, void* stream) {
  QEA_REQUIRE(dpool && y && mean && invstd && dy && relu_scale && relu_shift, "qea_bn_bwd_pool: null pointer (relu_scale / relu_shift are required)");
  QEA_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && (kw == 1 || kw == 2) && W % kw == 0, "qea_bn_bwd_pool: 2 x kw windows, kw in {1, 2}, H and W multiples");
  const long long M = (long long)B * H * W;
  const long long NWIN = M / (2 * kw);
  // (a window is 2 * kw rows of work: the reduction's blocks are sized as for that many rows, so that a small tensor — the reference's own
  // batch sizes — still spreads over the chip instead of walking its windows in 16 workgroups)
  ColGeom g = col_geom(M, C);
  g.rows_per_block = (g.rows_per_block + 2 * kw - 1) / (2 * kw);
  if (g.rows_per_block < 1) g.rows_per_block = 1;
  g.grid = (int)((NWIN + g.rows_per_block - 1) / g.rows_per_block);
  int rc = check_nc("qea_bn_bwd_pool", M, C, workspace_bytes, workspace, g);
  if (rc) return rc;
  QEA_REQUIRE(workspace_bytes >= qea_colreduce_workspace_bytes(M, C) && g.grid <= col_geom(M, C).grid,
              "qea_bn_bwd_pool: workspace too small (qea_colreduce_workspace_bytes(B * H * W, C))");
  QEA_REQUIRE(ldy % 4 == 0 && lddy % 4 == 0 && lddp % 4 == 0 && (!da || ldda % 4 == 0), "qea_bn_bwd_pool: strides must be multiples of 4");
  hipStream_t s = (hipStream_t)stream;
  double* ws = (double*)workspace;
  // (the constants sit behind the partial rows of qea_bn_bwd's geometry for M rows, which has at least as many blocks)
  const ColGeom gm = col_geom(M, C);
  double* k0 = ws + (size_t)(gm.grid > g.grid ? gm.grid : g.grid) * C * 2;
  double* k1 = k0 + C;
  double* k2 = k1 + C;
  const size_t lds = (size_t)g.rt * C * 2 * sizeof(double);
  auto red = kw == 2 ? bn_bwd_pool_kernel<2, false> : bn_bwd_pool_kernel<1, false>;
  hipLaunchKernelGGL(red, dim3(g.grid), dim3(RED_THREADS), lds, s, da, ldda, dpool, lddp, y, ldy, (float*)nullptr, 0, B, H, W, C, relu_scale, relu_shift,
                     mean, invstd, stat64, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, g.rows_per_block, g.rt, ws,
                     (float*)nullptr);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(qea_cdiv(C, 4)), dim3(256), 0, s, (const double*)ws, g.grid, C, M, gamma, mean, invstd, stat64, training,
                     dgamma, dbeta, accumulate_param_grads, k0, k1, k2);
  long long agrid = (NWIN + (long long)g.rt * 4 - 1) / ((long long)g.rt * 4);
  if (agrid > 4096) agrid = 4096;
  if (agrid < 1) agrid = 1;
  const int awin = (int)((NWIN + agrid - 1) / agrid);
  agrid = (NWIN + awin - 1) / awin;
  auto app = kw == 2 ? bn_bwd_pool_kernel<2, true> : bn_bwd_pool_kernel<1, true>;
  hipLaunchKernelGGL(app, dim3((unsigned)agrid), dim3(RED_THREADS), 0, s, da, ldda, dpool, lddp, y, ldy, dy, lddy, B, H, W, C, relu_scale, relu_shift, mean,
                     invstd, stat64, (const double*)k0, (const double*)k1, (const double*)k2, awin, g.rt, (double*)nullptr, absmax_out);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_bn_bwd_from_partials(const double* partials, int32_t blocks, const float* da, int32_t ldda, const float* relu_scale,
                                        const float* relu_shift, const float* y, int32_t ldy, int64_t M, int32_t C, const float* gamma,
                                        const float* mean, const float* invstd, const double* stat64, int32_t training, float* dgamma,
                                        float* dbeta, int32_t accumulate_param_grads, float* dy, int32_t lddy, void* workspace,
                                        size_t workspace_bytes, float* absmax_out, void* stream) {
  QEA_REQUIRE(partials && blocks > 0 && da && y && mean && invstd && dy && stat64 && relu_scale && relu_shift && M > 0 && C > 0 && C % 4 == 0,
              "qea_bn_bwd_from_partials: null pointer / bad size (stat64 and relu_scale / relu_shift are required)");
  QEA_REQUIRE(ldda % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0, "qea_bn_bwd_from_partials: strides must be multiples of 4");
  QEA_REQUIRE(workspace && workspace_bytes >= (size_t)3 * C * sizeof(double), "qea_bn_bwd_from_partials: workspace too small (3 * C doubles)");
  const ColGeom g = col_geom(M, C);
  hipStream_t s = (hipStream_t)stream;
  double* k0 = (double*)workspace;
  double* k1 = k0 + C;
  double* k2 = k1 + C;
  const double* src = partials;
  int nblk = blocks;
  if (blocks > QEA_BN_PARTIAL_SCRATCH_ROWS * 2) {          // two stages, as qea_bn_train_stats_from_partials
    const int per = qea_cdiv(blocks, QEA_BN_PARTIAL_SCRATCH_ROWS);
    nblk = qea_cdiv(blocks, per);
    double* scratch = const_cast<double*>(partials) + (size_t)blocks * C * 2;
    hipLaunchKernelGGL(partials_reduce_kernel, dim3(nblk), dim3(256), (size_t)256 * 2 * sizeof(double), s, partials, blocks, C, per, scratch);
    src = scratch;
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(qea_cdiv(C, 4)), dim3(256), 0, s, src, nblk, C, (long long)M, gamma, mean, invstd, stat64,
                     training, dgamma, dbeta, accumulate_param_grads, k0, k1, k2);
  long long agrid = (M + (long long)g.rt * 8 - 1) / ((long long)g.rt * 8);
  if (agrid > 4096) agrid = 4096;
  if (agrid < 1) agrid = 1;
  const int arows = (int)((M + agrid - 1) / agrid);
  agrid = (M + arows - 1) / arows;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)agrid), dim3(RED_THREADS), 0, s, da, ldda, (const float*)nullptr, 0, y, ldy, dy, lddy,
                     (long long)M, C, relu_scale, relu_shift, (const double*)k0, (const double*)k1, (const double*)k2, arows, g.rt, absmax_out);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_colsum(const float* x, int32_t ldx, int64_t M, int32_t C, float* out, int32_t accumulate, void* workspace,
                          size_t workspace_bytes, void* stream) {
  QEA_REQUIRE(x && out, "qea_colsum: null pointer");
  const ColGeom g = col_geom(M, C);
  int rc = check_nc("qea_colsum", M, C, workspace_bytes, workspace, g);
  if (rc) return rc;
  QEA_REQUIRE(ldx % 4 == 0, "qea_colsum: ldx must be a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)g.rt * C * 2 * sizeof(double);
  hipLaunchKernelGGL(colreduce_kernel<2>, dim3(g.grid), dim3(RED_THREADS), lds, s, x, ldx, nullptr, 0, nullptr, 0, nullptr, nullptr,
                     nullptr, nullptr, nullptr, (long long)M, C, g.rows_per_block, g.rt, (double*)workspace);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(qea_cdiv(C, 4)), dim3(256), 0, s, (const double*)workspace, g.grid, C, out, accumulate);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_maxpool_fwd(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t kh,
                               int32_t kw, float* absmax_out, void* stream) {
  QEA_REQUIRE(x && y && B > 0 && C > 0 && C % 4 == 0 && kh > 0 && kw > 0 && H % kh == 0 && W % kw == 0 && ldx % 4 == 0 && ldy % 4 == 0,
              "qea_maxpool_fwd: bad arguments (H,W must be multiples of the window; C, ld multiples of 4)");
  const long long n = (long long)B * (H / kh) * (W / kw) * (C / 4);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, B, H, W, C, kh, kw, absmax_out);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_bn_apply_pool(const float* y, int32_t ldy, float* a, int32_t lda, float* pooled, int32_t ldp, int32_t B, int32_t H, int32_t W,
                                 int32_t C, const float* scale, const float* shift, int32_t relu, int32_t kh, int32_t kw, float* absmax_a,
                                 float* absmax_pooled, void* stream) {
  QEA_REQUIRE(y && a && pooled && scale && shift && B > 0 && C > 0 && C % 4 == 0 && kh > 0 && kw > 0 && H > 0 && W > 0 && H % kh == 0 && W % kw == 0 &&
                  ldy % 4 == 0 && lda % 4 == 0 && ldp % 4 == 0,
              "qea_bn_apply_pool: bad arguments (H,W must be multiples of the window; C, ld multiples of 4)");
  const long long n = (long long)B * (H / kh) * (W / kw) * (C / 4);
  hipLaunchKernelGGL(bn_apply_pool_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, ldy, a, lda, pooled, ldp, B, H, W, C, scale, shift,
                     relu, kh, kw, absmax_a, absmax_pooled);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_maxpool_bwd(const float* x, int32_t ldx, const float* dy, int32_t lddy, float* dx, int32_t lddx, int32_t B, int32_t H,
                               int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t relu_mask, int32_t accumulate, float* absmax_out,
                               void* stream) {
  QEA_REQUIRE(x && dy && dx && B > 0 && C > 0 && C % 4 == 0 && kh > 0 && kw > 0 && H % kh == 0 && W % kw == 0 && ldx % 4 == 0 &&
                  lddy % 4 == 0 && lddx % 4 == 0,
              "qea_maxpool_bwd: bad arguments");
  const long long n = (long long)B * (H / kh) * (W / kw) * (C / 4);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, ldx, dy, lddy, dx, lddx, B, H, W, C, kh,
                     kw, relu_mask, accumulate, absmax_out);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_transpose2d(const float* in, float* out, int32_t R, int32_t Cc, void* stream) {
  QEA_REQUIRE(in && out && R > 0 && Cc > 0, "qea_transpose2d: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3(qea_cdiv(Cc, 32), qea_cdiv(R, 32)), dim3(256), 0, (hipStream_t)stream, in, out, R, Cc);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_filter_flip_transpose(const float* w, float* wt, int32_t Co, int32_t Ci, int32_t KH, int32_t KW, void* stream) {
  QEA_REQUIRE(w && wt && Co > 0 && Ci > 0 && KH > 0 && KW > 0, "qea_filter_flip_transpose: bad arguments");
  hipLaunchKernelGGL(flip_transpose_kernel, dim3(grid_for((long long)Co * Ci * KH * KW)), dim3(256), 0, (hipStream_t)stream, w, wt, Co, Ci,
                     KH, KW);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}
