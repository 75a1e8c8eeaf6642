// log_softmax (+ backward with the reference's NaN scrub) and the CTC loss / gradient.
//
// CTC follows ATen's native recursion (Graves 2006 eq. 6-16 in log space).  The recursion, the
// per-sample nll and the gradient formula run in fp64 (inputs/outputs fp32): with |log-likelihood|
// ~ 1e2 an fp32 recursion carries ~1e-5 absolute error into exp(alpha+beta+nll-lp), i.e. a 1e-5..1e-4
// RELATIVE error common to every (t,c) of a sample, which per-channel sums downstream amplify
// (measured: bias gradients 4-7x further from the exact value than ATen's fp32 path).  fp64 costs
// nothing here (latency-bound scan, 2*T*S values per sample).
//   alpha_t(s) = lse(alpha_{t-1}(s), alpha_{t-1}(s-1), [alpha_{t-1}(s-2)]) + lp[t, l'_s]
// one workgroup per sample, one thread per extended-label state (S = 2L+1 <= 256),
// one barrier per time step; alpha and beta go to a caller-owned workspace and a second kernel
// forms   grad[t,n,c] = (exp(lp) - exp(lse_{s: l'_s = c}(alpha+beta) + nll - lp)) * grad_out[n]
// exactly as ATen does (reference call sites: train_nn_patch.py:143,178,294).
// Latency-bound (31-step scan); bytes: 2 * T*N*C*4 in/out + 2 * N*T*S*4 workspace.
#include "common.h"

namespace {

constexpr int CTC_MAX_S = 256;

#define NEG_INF_D (-(double)INFINITY)
__device__ __forceinline__ double lse3(double a, double b, double c) {
  const double m = fmax(fmax(a, b), c);
  if (m == NEG_INF_D) return NEG_INF_D;
  return m + log(exp(a - m) + exp(b - m) + exp(c - m));
}
__device__ __forceinline__ double lse2(double a, double b) {
  const double m = fmax(a, b);
  if (m == NEG_INF_D) return NEG_INF_D;
  return m + log(exp(a - m) + exp(b - m));
}

// one wave per row
__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, long long M,
                                                              int C) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + row * ldx;
  float mx = -INFINITY;
  for (int c = lane; c < C; c += 64) mx = fmaxf(mx, xr[c]);
  mx = qea_wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += expf(xr[c] - mx);
  s = qea_wave_sum(s);
  const float lz = mx + logf(s);
  for (int c = lane; c < C; c += 64) y[row * ldy + c] = xr[c] - lz;
}

// dx = g - exp(lp) * sum_c g ; optional NaN -> 0 (models/model_crnn.py:30-32); pad columns -> 0
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ lp, int ldlp,
                                                              float* __restrict__ dx, int lddx, long long M, int C, int Cpad, int scrub) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += g[row * ldg + c];
  s = qea_wave_sum(s);
  for (int c = lane; c < Cpad; c += 64) {
    float v = 0.f;
    if (c < C) {
      v = g[row * ldg + c] - expf(lp[row * ldlp + c]) * s;
      if (scrub && v != v) v = 0.f;
    }
    dx[row * lddx + c] = v;
  }
}

// alpha (blockIdx.y == 0) and beta (blockIdx.y == 1) recursions: one workgroup per (sample, pass), one thread per state
__global__ __launch_bounds__(CTC_MAX_S) void ctc_alpha_beta_kernel(const float* __restrict__ lp, int ld_t, int ld_n,
                                                                    const int* __restrict__ targets, const long long* __restrict__ tg_off,
                                                                    const int* __restrict__ in_len, const int* __restrict__ tg_len, int T,
                                                                    int blank, double* __restrict__ alpha, double* __restrict__ beta,
                                                                    float* __restrict__ nll, double* __restrict__ nll64, int S_max) {
  __shared__ double prev[2][CTC_MAX_S + 2];
  const int n = blockIdx.x;
  const bool is_beta = blockIdx.y == 1;
  const int s = threadIdx.x;
  const int L = tg_len[n];
  const int Tn = min(in_len[n], T);
  const int S = 2 * L + 1;
  const int* tg = targets + tg_off[n];
  const float* lpn = lp + (size_t)n * ld_n;
  double* out = (is_beta ? beta : alpha) + (size_t)n * T * S_max;

  int ch = blank;      // l'_s
  bool skip = false;   // alpha: may come from s-2 ; beta: may go to s+2
  if (s < S && (s & 1)) {
    ch = tg[s >> 1];
    if (!is_beta) skip = (s >= 2) && (tg[(s >> 1) - 1] != ch);
    else skip = (s + 2 < S) && (tg[(s >> 1) + 1] != ch);
  }
  if (Tn <= 0 || S > S_max) {  // degenerate / longer than the caller sized for: infeasible
    if (!is_beta && s == 0) {
      nll[n] = INFINITY;
      nll64[n] = (double)INFINITY;
    }
    return;
  }

  // prev rows are padded by 2 on the side the recursion reaches into
  double cur = NEG_INF_D;
  if (!is_beta) {
    if (s == 0) cur = (double)lpn[blank];
    else if (s == 1 && S > 1) cur = (double)lpn[ch];
  } else {
    const float* lpt = lpn + (size_t)(Tn - 1) * ld_t;
    if (s == S - 1) cur = (double)lpt[blank];
    else if (s == S - 2 && S > 1) cur = (double)lpt[ch];
  }
  int buf = 0;
  if (s < 2) {
    prev[0][is_beta ? CTC_MAX_S + s : s] = NEG_INF_D;  // padding cells
    prev[1][is_beta ? CTC_MAX_S + s : s] = NEG_INF_D;
  }
  // storage index: alpha uses prev[.][s+2] (reads s+1, s), beta uses prev[.][s] (reads s+1, s+2)
  const int off = is_beta ? 0 : 2;
  if (s < S) out[(size_t)(is_beta ? Tn - 1 : 0) * S_max + s] = cur;
  prev[buf][s + off] = (s < S) ? cur : NEG_INF_D;
  __syncthreads();
  for (int step = 1; step < Tn; ++step) {
    const int t = is_beta ? Tn - 1 - step : step;
    double v = NEG_INF_D;
    if (s < S) {
      double a0, a1, a2;
      if (!is_beta) {
        a0 = prev[buf][s + 2];
        a1 = prev[buf][s + 1];
        a2 = skip ? prev[buf][s] : NEG_INF_D;
      } else {
        a0 = prev[buf][s];
        a1 = (s + 1 < S) ? prev[buf][s + 1] : NEG_INF_D;
        a2 = skip ? prev[buf][s + 2] : NEG_INF_D;
      }
      v = lse3(a0, a1, a2) + (double)lpn[(size_t)t * ld_t + ch];
      out[(size_t)t * S_max + s] = v;
    }
    buf ^= 1;
    prev[buf][s + off] = v;
    __syncthreads();
  }
  if (!is_beta) {
    // prev[buf] holds alpha_{Tn-1}
    if (s == 0) {
      const double a = prev[buf][(S - 1) + 2];
      const double b = (S > 1) ? prev[buf][(S - 2) + 2] : NEG_INF_D;
      const double v = -lse2(a, b);
      nll64[n] = v;
      nll[n] = (float)v;
    }
  }
}

// grad for one (n, t) row: thread c scans the states carrying character c
__global__ __launch_bounds__(128) void ctc_grad_kernel(const float* __restrict__ lp, int ld_t, int ld_n, const int* __restrict__ targets,
                                                       const long long* __restrict__ tg_off, const int* __restrict__ in_len,
                                                       const int* __restrict__ tg_len, int T, int C, int blank,
                                                       const double* __restrict__ alpha, const double* __restrict__ beta,
                                                       const double* __restrict__ nll, const float* __restrict__ grad_out,
                                                       float* __restrict__ grad, int gld_t, int gld_n, int S_max) {
  __shared__ double ab[CTC_MAX_S];
  __shared__ int ext[CTC_MAX_S];
  const int n = blockIdx.x, t = blockIdx.y;
  const int L = tg_len[n], Tn = min(in_len[n], T);
  const int S = 2 * L + 1;
  float* grow = grad + (size_t)n * gld_n + (size_t)t * gld_t;
  if (t >= Tn || S > S_max) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) grow[c] = 0.f;
    return;
  }
  const int* tg = targets + tg_off[n];
  const double* al = alpha + ((size_t)n * T + t) * S_max;
  const double* be = beta + ((size_t)n * T + t) * S_max;
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    ab[s] = al[s] + be[s];
    ext[s] = (s & 1) ? tg[s >> 1] : blank;
  }
  __syncthreads();
  const double nl = nll[n];
  const double go = (double)grad_out[n];
  const float* lpr = lp + (size_t)n * ld_n + (size_t)t * ld_t;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double res = NEG_INF_D;
    for (int s = 0; s < S; ++s)
      if (ext[s] == c) res = lse2(res, ab[s]);
    const double l = (double)lpr[c];
    grow[c] = (float)((exp(l) - exp(res + nl - l)) * go);
  }
}

__global__ void ctc_reduce_kernel(const double* __restrict__ nll, const int* __restrict__ tg_len, int N, int reduction, float scale_in,
                                  float* __restrict__ loss, float* __restrict__ grad_out) {
  // reduction 1 = mean: loss = mean_n(nll_n / max(len_n,1)), grad_out_n = scale_in / (N * max(len_n,1))
  // reduction 0 = none: grad_out_n = scale_in (caller multiplies by its own upstream gradient)
  __shared__ double sred[256];
  double acc = 0;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const float tl = (float)max(tg_len[n], 1);
    if (reduction == 1) {
      acc += nll[n] / (double)tl;
      grad_out[n] = scale_in / ((float)N * tl);
    } else {
      acc += nll[n];
      grad_out[n] = scale_in;
    }
  }
  sred[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sred[threadIdx.x] += sred[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (reduction == 1) ? (float)(sred[0] / (double)N) : (float)sred[0];
}

}  // namespace

extern "C" int qea_log_softmax_fwd(const float* x, int32_t ldx, float* y, int32_t ldy, int64_t M, int32_t C, void* stream) {
  QEA_REQUIRE(x && y && M > 0 && C > 0, "qea_log_softmax_fwd: bad arguments");
  hipLaunchKernelGGL(log_softmax_fwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, (long long)M, C);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_log_softmax_bwd(const float* g, int32_t ldg, const float* lp, int32_t ldlp, float* dx, int32_t lddx, int64_t M, int32_t C,
                                   int32_t Cpad, int32_t nan_scrub, void* stream) {
  QEA_REQUIRE(g && lp && dx && M > 0 && C > 0 && Cpad >= C && lddx >= Cpad, "qea_log_softmax_bwd: bad arguments");
  hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g, ldg, lp, ldlp, dx, lddx,
                     (long long)M, C, Cpad, nan_scrub);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" size_t qea_ctc_workspace_bytes(int32_t T, int32_t N, int32_t S_max) {
  if (T <= 0 || N <= 0 || S_max <= 0) return 0;
  return (size_t)2 * N * T * S_max * sizeof(double) + (size_t)N * sizeof(double) + (size_t)N * sizeof(float);
}

extern "C" int qea_ctc_loss(const float* lp, int32_t ld_t, int32_t ld_n, const int32_t* targets, const int64_t* target_offsets,
                            const int32_t* input_lengths, const int32_t* target_lengths, int32_t T, int32_t N, int32_t C, int32_t blank,
                            int32_t S_max, int32_t reduction, float grad_scale, float* nll, float* loss, float* grad, int32_t gld_t,
                            int32_t gld_n, void* workspace, size_t workspace_bytes, void* stream) {
  QEA_REQUIRE(lp && targets && target_offsets && input_lengths && target_lengths && nll && loss, "qea_ctc_loss: null pointer");
  QEA_REQUIRE(T > 0 && N > 0 && C > 0 && blank >= 0 && blank < C, "qea_ctc_loss: bad dimensions");
  QEA_REQUIRE(S_max >= 1 && S_max <= CTC_MAX_S, "qea_ctc_loss: S_max=%d must be in [1,%d] (targets up to %d chars)", S_max, CTC_MAX_S,
              (CTC_MAX_S - 1) / 2);
  QEA_REQUIRE(workspace && workspace_bytes >= qea_ctc_workspace_bytes(T, N, S_max), "qea_ctc_loss: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  double* alpha = (double*)workspace;
  double* beta = alpha + (size_t)N * T * S_max;
  double* nll64 = beta + (size_t)N * T * S_max;
  float* gout = (float*)(nll64 + N);
  const int threads = ((S_max + 63) / 64) * 64;
  hipLaunchKernelGGL(ctc_alpha_beta_kernel, dim3(N, grad ? 2 : 1), dim3(threads), 0, s, lp, ld_t, ld_n, targets,
                     (const long long*)target_offsets, input_lengths, target_lengths, T, blank, alpha, beta, nll, nll64, S_max);
  hipLaunchKernelGGL(ctc_reduce_kernel, dim3(1), dim3(256), 0, s, (const double*)nll64, target_lengths, N, reduction, grad_scale, loss, gout);
  if (grad) {
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(N, T), dim3(128), 0, s, lp, ld_t, ld_n, targets, (const long long*)target_offsets, input_lengths,
                       target_lengths, T, C, blank, (const double*)alpha, (const double*)beta, (const double*)nll64, (const float*)gout, grad,
                       gld_t, gld_n, S_max);
  }
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}
