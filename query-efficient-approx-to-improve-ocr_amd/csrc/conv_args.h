// ConvArgs + the shared epilogue of the MFMA conv / GEMM kernels (conv_igemm.hip, gemm1x1.hip).
#pragma once
#include "common.h"

namespace qea_conv {

struct ConvArgs {
  const float* x;
  const float* w;
  float* y;
  const float* scale;
  const float* bias;
  const float* mask;
  int B, H, W, Cin, OH, OW, N, KH, KW, pad_h, pad_w, stride_h, stride_w;
  int ldx, ldy, ldmask, relu, accumulate, out_mode;
  int M, K, n_tiles, m_tiles;
  // pre-split operands (P3 format, see qea_split_planes): byte pointers + the byte offset of each buffer's zero chunk
  const char* xp;
  const char* wp;
  unsigned xp_zero, wp_zero;
  // fused BatchNorm batch statistics (STATS kernels): per (M-tile, wave row) partial column sums [blocks][N][2] in fp64
  double* stats;
  // two-way fp16 split (QEA_MFMA_SPLIT_F16): largest finite |x| of the input tensor (device scalar) — the filter's is in its planes
  const float* xmax;
  // producer-carried abs-max of the stored outputs (qea_conv_desc.y_absmax), or null
  float* yamax;
  // fused max-pool of the stored outputs (LDS-halo kernel only): pooled tensor, its pixel stride, window width (height 2), abs-max slot
  float* pool_y;
  int ldpool, pool_kw;
  float* pool_amax;
  // BatchNorm-backward partial sums of the stored tensor (LDS-halo kernel only; the partials go to `stats`): the BatchNorm's input
  // (pre-BN conv output), its pixel stride, [2][N] fp64 mean / invstd, the forward's scale / shift (ReLU mask)
  const float* bst_y;
  int ldbst;
  const double* bst64;
  const float* bst_scale;
  const float* bst_shift;
};

// the 16x16x32 form of the LDS-halo conv lives in conv_halo16.hip
int launch_halo_m16_any(int n_sel, int sm, const ConvArgs& a, hipStream_t s);

// Shared epilogue of the MFMA conv kernels.  C/D map of a 32x32 accumulator tile: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Order: v = acc*scale + bias; relu; mask; accumulate; store (out_mode remaps).
// STATS: additionally accumulate, per output column, sum and sum of squares of the STORED values in fp64 (rows past M hold
// zero accumulators and add nothing) and write one partial per (M-tile, wave row): the batch statistics of the BatchNorm
// that follows (models/model_unet.py:78-109) without a second pass over the conv output.
template <int MI, int NJ, int TM, int TN, bool STATS = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, const f32x16 (&acc)[MI][NJ], int m0, int n0, int wm, int wn, int fr, int fh,
                                              int stats_block = 0) {
  const int ohw = p.OH * p.OW;
  float am = 0.f;
  double st0[NJ], st1[NJ];
  if (STATS) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) st0[j] = st1[j] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      if (m >= p.M) continue;
      size_t orow;  // output row (pixel) index for QEA_OUT_NHWC / TBC
      int cb = 0, ch = 0, cw = 0;
      if (p.out_mode == QEA_OUT_NHWC) {
        orow = (size_t)m;
      } else if (p.out_mode == QEA_OUT_TBC) {
        const int b = m / p.OW;
        const int ow = m - b * p.OW;
        orow = (size_t)ow * p.B + b;
      } else {
        cb = m / ohw;
        const int rem = m - cb * ohw;
        ch = rem / p.OW;
        cw = rem - ch * p.OW;
        orow = 0;
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wn * TN + j * 32 + fr;
        if (n >= p.N) continue;
        float v = acc[i][j][r];
        size_t o;
        int nb = n;
        if (p.out_mode == QEA_OUT_CONVT) {
          const int co_n = p.N >> 2;
          const int ab = n / co_n;
          nb = n - ab * co_n;
          const size_t opix = ((size_t)cb * (2 * p.OH) + 2 * ch + (ab >> 1)) * (2 * p.OW) + 2 * cw + (ab & 1);
          o = opix * p.ldy + nb;
        } else {
          o = orow * p.ldy + n;
        }
        if (p.scale && p.bias) v = __fmaf_rn(v, p.scale[n], p.bias[nb]);  // the very fma qea_bn_apply evaluates
        else if (p.scale) v *= p.scale[n];
        else if (p.bias) v += p.bias[nb];
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.mask) {
          const size_t mo = (p.out_mode == QEA_OUT_CONVT) ? (o / p.ldy) * p.ldmask + nb : orow * p.ldmask + n;
          v = (p.mask[mo] > 0.f) ? v : 0.f;
        }
        if (p.accumulate) v += p.y[o];
        p.y[o] = v;
        am = qea_amax_acc(am, v);
        if (STATS) {
          st0[j] += (double)v;
          st1[j] += (double)v * (double)v;
        }
      }
    }
  }
  qea_amax_commit_block(am, p.yamax);                      // (every lane of the workgroup runs the epilogue to its end: one gated access per workgroup)
  if (STATS) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      // the two lane halves hold different rows of the same column
      const double a = st0[j] + __shfl_xor(st0[j], 32, 64);
      const double b = st1[j] + __shfl_xor(st1[j], 32, 64);
      const int n = n0 + wn * TN + j * 32 + fr;
      if (fh == 0 && n < p.N) {
        double* dst = p.stats + ((size_t)stats_block * p.N + n) * 2;
        dst[0] = a;
        dst[1] = b;
      }
    }
  }
}

// gemm1x1.hip: the 1x1 / transposed-conv GEMM on a 128-row LDS tile (two-way fp16 split; filter in fragment-order planes)
__attribute__((visibility("hidden"))) int launch_gemm1x1_f16(const ConvArgs& a, hipStream_t s);

}  // namespace qea_conv
