// Small kernels around the networks: fused multi-tensor Adam over flat buffers, per-image
// Gaussian jitter (Philox4x32-10 + Box-Muller) with replicas fused into the batch dimension,
// stable descending top-k (TopKCER), crop+pad gather / scatter-add, greedy CTC decode.
// All HBM- or latency-bound; bytes noted per entry point in include/qea_hip.h.
#include "common.h"

static __device__ __forceinline__ int qea_floordiv2(int d) { return d >= 0 ? d / 2 : -((-d + 1) / 2); }

namespace {

// ------------------------------------------------------------------ Adam
// coef (optional, device): {step_size, bc2_sqrt} written by adam_coeff_kernel — the capturable form, whose launch
// arguments do not depend on the step count
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                            float beta1, float beta2, float eps, float weight_decay, float step_size, float bc2_sqrt, float grad_scale,
                            const float* __restrict__ coef) {
  if (coef) {
    step_size = coef[0];
    bc2_sqrt = coef[1];
  }
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gr = gv[k] * grad_scale;
      if (weight_decay != 0.f) gr += weight_decay * pv[k];
      mv[k] = mv[k] + (gr - mv[k]) * (1.f - beta1);          // lerp, as torch's exp_avg.lerp_
      vv[k] = vv[k] * beta2 + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
      pv[k] = pv[k] - step_size * (mv[k] / denom);
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  // tail
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float gr = g[i] * grad_scale;
    if (weight_decay != 0.f) gr += weight_decay * p[i];
    const float mm = m[i] + (gr - m[i]) * (1.f - beta1);
    const float vv = v[i] * beta2 + (1.f - beta2) * gr * gr;
    m[i] = mm;
    v[i] = vv;
    p[i] = p[i] - step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
  }
}

// ------------------------------------------------------------------ Philox4x32-10
struct U4 { unsigned x, y, z, w; };
__device__ __forceinline__ U4 philox4x32(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c.z;
    U4 n;
    n.x = (unsigned)(p1 >> 32) ^ c.y ^ k0;
    n.y = (unsigned)p1;
    n.z = (unsigned)(p0 >> 32) ^ c.w ^ k1;
    n.w = (unsigned)p0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ float u01(unsigned x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)

// out[r*K + k][i] = clamp(img[k][i] - coef * sigma[r*K + k] * z, 0, 1), 4 pixels per thread
__global__ void jitter_kernel(const float* __restrict__ img, const float* __restrict__ sigma, float* __restrict__ out,
                              float* __restrict__ noise_out, int K, int R, int HW, float coef, unsigned long long seed,
                              unsigned long long offset) {
  const long long quads = (long long)R * K * (HW / 4);
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (long long)gridDim.x * blockDim.x) {
    const long long im = q / (HW / 4);
    const int i4 = (int)(q - im * (HW / 4));
    const int k = (int)(im % K);
    U4 ctr;
    ctr.x = (unsigned)q;
    ctr.y = (unsigned)(q >> 32);
    ctr.z = (unsigned)offset;
    ctr.w = (unsigned)(offset >> 32);
    const U4 rnd = philox4x32(ctr, (unsigned)seed, (unsigned)(seed >> 32));
    const float r0 = sqrtf(-2.f * logf(u01(rnd.x))), r1 = sqrtf(-2.f * logf(u01(rnd.z)));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u01(rnd.y), &s0, &c0);
    sincosf(6.283185307179586f * u01(rnd.w), &s1, &c1);
    const float sg = sigma[im];
    f32x4 z = {r0 * c0 * sg, r0 * s0 * sg, r1 * c1 * sg, r1 * s1 * sg};
    const f32x4 x = *reinterpret_cast<const f32x4*>(img + (size_t)k * HW + i4 * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fminf(fmaxf(x[e] - coef * z[e], 0.f), 1.f);
    *reinterpret_cast<f32x4*>(out + (size_t)im * HW + i4 * 4) = o;
    if (noise_out) *reinterpret_cast<f32x4*>(noise_out + (size_t)im * HW + i4 * 4) = z;
  }
}

__global__ void jitter_apply_kernel(const float* __restrict__ img, const float* __restrict__ noise, float* __restrict__ out, int K, int R,
                                    int HW, float coef) {
  const long long n = (long long)R * K * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long im = i / HW;
    const int px = (int)(i - im * HW);
    const int k = (int)(im % K);
    out[i] = fminf(fmaxf(img[(size_t)k * HW + px] - coef * noise[i], 0.f), 1.f);
  }
}

// ------------------------------------------------------------------ top-k (bitonic, one workgroup)
// total order: larger key first; equal keys -> smaller index first (stable descending)
__device__ __forceinline__ bool before(float ka, int ia, float kb, int ib) { return (ka > kb) || (ka == kb && ia < ib); }

__global__ __launch_bounds__(1024) void topk_kernel(const float* __restrict__ keys, int n, int npow2, int k, long long* __restrict__ idx_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sk = reinterpret_cast<float*>(smem_raw);
  int* si = reinterpret_cast<int*>(sk + npow2);
  for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
    sk[i] = (i < n) ? keys[i] : -INFINITY;
    si[i] = (i < n) ? i : 0x7fffffff;
  }
  __syncthreads();
  for (int size = 2; size <= npow2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < npow2 / 2; t += blockDim.x) {
        const int lo = (t / stride) * (stride << 1) + (t % stride);
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);  // ascending position order = "before" order
        const float ka = sk[lo], kb = sk[hi];
        const int ia = si[lo], ib = si[hi];
        const bool swap = up ? before(kb, ib, ka, ia) : before(ka, ia, kb, ib);
        if (swap) {
          sk[lo] = kb; sk[hi] = ka;
          si[lo] = ib; si[hi] = ia;
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < k; i += blockDim.x) idx_out[i] = (long long)si[i];
}

// ------------------------------------------------------------------ crop + pad
struct Box { int x0, y0, x1, y1; };

__global__ void crop_pad_gather_kernel(const float* __restrict__ img, int H, int W, const int* __restrict__ boxes, int N, int OH, int OW,
                                       float* __restrict__ out) {
  const long long n_el = (long long)N * OH * OW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int oy = (int)((i / OW) % OH);
    const int n = (int)(i / ((long long)OW * OH));
    const int x0 = boxes[n * 4 + 0], y0 = boxes[n * 4 + 1], x1 = boxes[n * 4 + 2], y1 = boxes[n * 4 + 3];
    const int cw = x1 - x0, ch = y1 - y0;
    // Python floor division as in padder (utils.py:118-126): a crop larger than the target by an odd amount loses the
    // EXTRA pixel on the left/top (negative padding), where C's truncation would take it from the right/bottom
    const int left = qea_floordiv2(OW - cw), top = qea_floordiv2(OH - ch);
    const int sx = ox - left, sy = oy - top;
    float v = 1.f;
    if (sx >= 0 && sx < cw && sy >= 0 && sy < ch) v = img[(size_t)(y0 + sy) * W + x0 + sx];
    out[i] = v;
  }
}

__global__ void crop_pad_scatter_kernel(const float* __restrict__ dout, const int* __restrict__ boxes, int N, int OH, int OW, float* dimg,
                                        int H, int W) {
  const long long n_el = (long long)N * OH * OW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % OW);
    const int oy = (int)((i / OW) % OH);
    const int n = (int)(i / ((long long)OW * OH));
    const int x0 = boxes[n * 4 + 0], y0 = boxes[n * 4 + 1], x1 = boxes[n * 4 + 2], y1 = boxes[n * 4 + 3];
    const int cw = x1 - x0, ch = y1 - y0;
    // Python floor division as in padder (utils.py:118-126): a crop larger than the target by an odd amount loses the
    // EXTRA pixel on the left/top (negative padding), where C's truncation would take it from the right/bottom
    const int left = qea_floordiv2(OW - cw), top = qea_floordiv2(OH - ch);
    const int sx = ox - left, sy = oy - top;
    if (sx >= 0 && sx < cw && sy >= 0 && sy < ch) atomicAdd(dimg + (size_t)(y0 + sy) * W + x0 + sx, dout[i]);
  }
}

// ------------------------------------------------------------------ greedy CTC decode
// one wave per sample: argmax over C per step (first maximum), collapse repeats, drop blank
__global__ __launch_bounds__(64) void greedy_decode_kernel(const float* __restrict__ scores, int ld_t, int ld_n, int T, int N, int C, int blank,
                                                           int* __restrict__ tokens, int* __restrict__ lengths) {
  const int n = blockIdx.x, lane = threadIdx.x;
  int prev = -1, len = 0;
  for (int t = 0; t < T; ++t) {
    const float* row = scores + (size_t)t * ld_t + (size_t)n * ld_n;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
      const float v = row[c];
      if (v > best || (v == best && c < bi) || (v != v && best == best)) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    // reference quirk (utils.py:85-88): the first emitted char does not look at the previous index
    if (bi != blank && (len == 0 || bi != prev)) {
      if (lane == 0) tokens[(size_t)n * T + len] = bi;
      ++len;
    }
    prev = bi;
  }
  if (lane == 0) lengths[n] = len;
}

// ------------------------------------------------------------------ edit distance (CER numerator)
// one thread per sample, the DP row in LDS column-major (row[j*64 + lane]: conflict-free), unit costs
constexpr int ED_MAX = 128;
__global__ __launch_bounds__(64) void edit_distance_kernel(const int* __restrict__ pred, int ldp, const int* __restrict__ pred_len,
                                                           const int* __restrict__ gt, const long long* __restrict__ gt_off,
                                                           const int* __restrict__ gt_len, int N, int* __restrict__ out) {
  __shared__ int row[(ED_MAX + 1) * 64];
  const int lane = threadIdx.x;
  const int n = blockIdx.x * 64 + lane;
  if (n >= N) return;
  const int lp = min(pred_len[n], ED_MAX), lg = gt_len[n];
  const int* p = pred + (size_t)n * ldp;
  const int* g = gt + gt_off[n];
  for (int j = 0; j <= lp; ++j) row[j * 64 + lane] = j;
  for (int i = 1; i <= lg; ++i) {
    const int gc = g[i - 1];
    int diag = row[lane];  // D[i-1][0]
    row[lane] = i;
    int left = i;
    for (int j = 1; j <= lp; ++j) {
      const int up = row[j * 64 + lane];
      const int v = min(min(up + 1, left + 1), diag + (p[j - 1] != gc));
      row[j * 64 + lane] = v;
      diag = up;
      left = v;
    }
  }
  out[n] = row[lp * 64 + lane];
}

int grid_for(long long n, int cap = 4096) {
  long long g = (n + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// step += 1;  coef = {lr / (1 - beta1^step), sqrt(1 - beta2^step)}   (same fp64 formulas as the host path)
__global__ void adam_coeff_kernel(float* step, float* coef, float lr, float beta1, float beta2) {
  const double t = (double)step[0] + 1.0;
  step[0] = (float)t;
  coef[0] = (float)((double)lr / (1.0 - pow((double)beta1, t)));
  coef[1] = (float)sqrt(1.0 - pow((double)beta2, t));
}

}  // namespace

extern "C" int qea_adam_step_capturable(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                        float eps, float weight_decay, float* step, float* coef, float grad_scale, void* stream) {
  QEA_REQUIRE(p && g && m && v && step && coef && n > 0, "qea_adam_step_capturable: bad arguments");
  QEA_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "qea_adam_step_capturable: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_coeff_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, coef, lr, beta1, beta2);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4 + 1, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, beta1, beta2, eps,
                     weight_decay, 0.f, 1.f, grad_scale, (const float*)coef);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int64_t step, float grad_scale, void* stream) {
  QEA_REQUIRE(p && g && m && v && n > 0 && step >= 1, "qea_adam_step: bad arguments");
  QEA_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "qea_adam_step: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4 + 1, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, beta1, beta2, eps,
                     weight_decay, (float)((double)lr / bc1), (float)sqrt(bc2), grad_scale, (const float*)nullptr);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_jitter(const float* img, const float* sigma, float* out, float* noise_out, int32_t K, int32_t R, int32_t HW, float coef,
                          uint64_t seed, uint64_t offset, void* stream) {
  QEA_REQUIRE(img && sigma && out && K > 0 && R > 0 && HW > 0 && HW % 4 == 0, "qea_jitter: bad arguments (HW must be a multiple of 4)");
  hipLaunchKernelGGL(jitter_kernel, dim3(grid_for((long long)R * K * (HW / 4))), dim3(256), 0, (hipStream_t)stream, img, sigma, out, noise_out,
                     K, R, HW, coef, (unsigned long long)seed, (unsigned long long)offset);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_jitter_apply(const float* img, const float* noise, float* out, int32_t K, int32_t R, int32_t HW, float coef, void* stream) {
  QEA_REQUIRE(img && noise && out && K > 0 && R > 0 && HW > 0, "qea_jitter_apply: bad arguments");
  hipLaunchKernelGGL(jitter_apply_kernel, dim3(grid_for((long long)R * K * HW)), dim3(256), 0, (hipStream_t)stream, img, noise, out, K, R, HW,
                     coef);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_topk_desc_stable(const float* keys, int32_t n, int32_t k, int64_t* idx_out, void* stream) {
  QEA_REQUIRE(keys && idx_out && n > 0 && k > 0 && k <= n, "qea_topk_desc_stable: bad arguments");
  QEA_REQUIRE(n <= 16384, "qea_topk_desc_stable: n=%d > 16384 not supported", n);
  int npow2 = 2;
  while (npow2 < n) npow2 <<= 1;
  const size_t lds = (size_t)npow2 * 8;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
    attr = true;
  }
  hipLaunchKernelGGL(topk_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, keys, n, npow2, k, (long long*)idx_out);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_crop_pad_gather(const float* img, int32_t H, int32_t W, const int32_t* boxes, int32_t N, int32_t OH, int32_t OW, float* out,
                                   void* stream) {
  QEA_REQUIRE(img && boxes && out && H > 0 && W > 0 && N > 0 && OH > 0 && OW > 0, "qea_crop_pad_gather: bad arguments");
  hipLaunchKernelGGL(crop_pad_gather_kernel, dim3(grid_for((long long)N * OH * OW)), dim3(256), 0, (hipStream_t)stream, img, H, W, boxes, N,
                     OH, OW, out);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_crop_pad_scatter(const float* dout, const int32_t* boxes, int32_t N, int32_t OH, int32_t OW, float* dimg, int32_t H,
                                    int32_t W, void* stream) {
  QEA_REQUIRE(dout && boxes && dimg && H > 0 && W > 0 && N > 0 && OH > 0 && OW > 0, "qea_crop_pad_scatter: bad arguments");
  hipLaunchKernelGGL(crop_pad_scatter_kernel, dim3(grid_for((long long)N * OH * OW)), dim3(256), 0, (hipStream_t)stream, dout, boxes, N, OH, OW,
                     dimg, H, W);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_greedy_decode(const float* scores, int32_t ld_t, int32_t ld_n, int32_t T, int32_t N, int32_t C, int32_t blank,
                                 int32_t* tokens, int32_t* lengths, void* stream) {
  QEA_REQUIRE(scores && tokens && lengths && T > 0 && N > 0 && C > 0, "qea_greedy_decode: bad arguments");
  hipLaunchKernelGGL(greedy_decode_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, scores, ld_t, ld_n, T, N, C, blank, tokens, lengths);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_edit_distance(const int32_t* pred_tokens, int32_t ldp, const int32_t* pred_len, const int32_t* gt_tokens,
                                 const int64_t* gt_offsets, const int32_t* gt_len, int32_t N, int32_t* out, void* stream) {
  QEA_REQUIRE(pred_tokens && pred_len && gt_tokens && gt_offsets && gt_len && out && N > 0 && ldp > 0 && ldp <= ED_MAX,
              "qea_edit_distance: bad arguments (prediction length <= %d)", ED_MAX);
  hipLaunchKernelGGL(edit_distance_kernel, dim3(qea_cdiv(N, 64)), dim3(64), 0, (hipStream_t)stream, pred_tokens, ldp, pred_len, gt_tokens,
                     (const long long*)gt_offsets, gt_len, N, out);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}


// ---------------------------------------------------------------------------------------------
// largest finite magnitude of a strided [M][C] tensor (the scale source of the two-way fp16 split, ABI v6).  Non-negative floats
// order like their bit patterns, so the block results meet in one atomicMax on the bits; NaN / inf are skipped.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int ld, long long M, int C, unsigned* __restrict__ out) {
  const int cols = C / 4;
  const long long n = M * cols;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols;
    const int ct = (int)(i - r * cols);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ld + ct * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = fabsf(v[k]);
      m = (a <= 3.4028234663852886e38f && a > m) ? a : m;      // the comparison is false for NaN; inf is excluded by the bound
    }
  }
  qea_amax_commit_block(m, reinterpret_cast<float*>(out));   // one gated access per workgroup
}

extern "C" int qea_absmax(const float* x, int32_t ld, int64_t M, int32_t C, float* out, void* stream) {
  QEA_REQUIRE(x && out && M > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 && ld >= C, "qea_absmax: bad arguments (C and ld multiples of 4)");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, sizeof(float), s) != hipSuccess) {
    qea_set_error("qea_absmax: memset failed");
    return QEA_ERR_LAUNCH;
  }
  const long long n = (long long)M * (C / 4);
  const int grid = (int)(n / 256 / 8 > 2048 ? 2048 : (n / 256 / 8 < 1 ? 1 : n / 256 / 8));
  hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, s, x, ld, (long long)M, C, (unsigned*)out);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}
