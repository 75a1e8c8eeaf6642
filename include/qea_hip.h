/*
 * qea_hip.h — C ABI of libqea_hip.so, the MI355X (gfx950) kernel library under the
 * preprocessor-training inner loop of tataganesh/Query-Efficient-Approx-to-improve-OCR.
 *
 * The reference has no FFI of its own: every op on the path is a stock torch.nn call
 * (SURVEY.md §2.3).  Each entry point below therefore cites the reference line that
 * ISSUES the op it replaces (paths relative to the reference root).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless named host_*;
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *    synchronises, nothing is allocated (caller owns outputs and workspaces);
 *  - return 0 on success, <0 on error (qea_last_error() gives the message); never throws;
 *  - activations are NHWC fp32: element (b,h,w,c) of a tensor with pixel stride `ld`
 *    lives at ((b*H+h)*W+w)*ld + c, so channel slices of wider buffers (UNet skip
 *    concatenations) are addressed with a base pointer + ld and no copy;
 *  - conv weights are [Cout][KH][KW][Cin] (= the checkpoint's OIHW tensor in
 *    torch.channels_last memory format), ConvTranspose2d weights are
 *    [Cin][KH][KW][Cout] (= IOHW in channels_last); LSTM/Linear weights are torch's
 *    row-major [out][in].
 */
#ifndef QEA_HIP_H
#define QEA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QEA_OK 0
#define QEA_ERR_INVALID (-1)
#define QEA_ERR_LAUNCH (-2)
#define QEA_ERR_WORKSPACE (-3)

/* ABI version; bumped whenever a struct below changes or entry points are added. */
int qea_version(void);
const char* qea_last_error(void);

/* Event-bracketed timing of one kernel class, for bench.py's roofline leg: while enabled,
 * every launch of that class is bracketed by hipEvents on its own stream; read() syncs
 * the events and returns summed milliseconds, algorithmic flops / bytes and launch count. */
#define QEA_PROF_CONV_IGEMM 0
#define QEA_PROF_CONV_WGRAD 1
#define QEA_PROF_LSTM_STEP 2
#define QEA_PROF_NCLASS 3
int qea_prof_enable(int klass, int on);
int qea_prof_reset(void);
int qea_prof_read(int klass, double* ms, double* flops, double* bytes, int64_t* launches);
/* per-launch view of the same records (in launch order): up to `capacity` entries, *count = recorded launches */
int qea_prof_read_launches(int klass, double* ms, double* flops, int64_t capacity, int64_t* count);
/* The same sums restricted to the launches one kernel took.  Tags of QEA_PROF_CONV_IGEMM: the tile id, except for the
 * split-bf16 LDS-halo kernel, whose template instantiation conv3x3_halo_bf3_kernel<CIN, COUT, .., STATS> is
 * QEA_PROF_TAG_HALO_BF3(CIN, COUT, STATS) — the name rocprofv3 lists it under, so that bench.py's roofline figures of the
 * dominant kernel can be checked against the kernel trace. */
#define QEA_PROF_TAG_HALO_BF3(cin_chunk, cout_group, stats) (24000 + ((cin_chunk) == 64 ? 1000 : 0) + (cout_group) + ((stats) ? 500 : 0))
/* (+ 20 * image width for the small-image instantiations, + 5 for the two-way fp16 instantiation: each is its own kernel in a trace); + 2000 * pool_kw for the instantiations with the fused max-pool, ABI v7) */
int qea_prof_read_tagged(int klass, int32_t tag, double* ms, double* flops, double* bytes, int64_t* launches);
/* The part of a class's algorithmic flops that ran through the split-bf16 kernels (six bf16 MFMAs per fp32
 * multiply-add): bench.py blends the fp32 and the bf16/6 matrix peaks with it. */
int qea_prof_read_split_bf16(int klass, double* flops);
/* ABI v6: the part that ran through the two-way fp16 split (three fp16 MFMAs per fp32 multiply-add) */
int qea_prof_read_split_f16(int klass, double* flops);

/* Which matrix instruction the GEMM-class launches use: 0 = split-bf16 tiles where the dispatcher prefers them
 * (default), 1 = every product on v_mfma_f32_32x32x2_f32.  The initial value comes from QEA_MFMA=f32 in the
 * environment; tests and bench.py's native-fp32 leg switch it at run time.  Returns the previous mode, or
 * QEA_ERR_INVALID for a mode other than 0/1 (mode -1 only queries).
 *
 * PROCESS-GLOBAL STATE, one of two exceptions to "no global mutable state" (the other: the profiling registry above, which
 * only records).  The mode is read by every GEMM-class entry point at LAUNCH time and selects the kernel, i.e. the summation
 * order of the results: a caller that flips it while another host thread of the same process is launching changes that
 * thread's results (both forms are fp32-class — see DESIGN.md §4 — but not bit-identical).  The library is used with one host
 * thread per GPU process; a process that needs both forms concurrently must serialise the switch with its
 * launches itself.  A single launch can be pinned regardless of the mode through its descriptor's `tile` field (tiles 1-9: fp32
 * instruction, 20-25: split-bf16). */
#define QEA_MFMA_SPLIT_BF16 0
#define QEA_MFMA_F32 1
int qea_set_mfma_mode(int mode);

/* ------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on the matrix cores with fp32-class accuracy: v_mfma_f32_32x32x2_f32, or — by default
 * for N >= 64-128 output channels and K >= 256 (tiles 20-23) — the split-bf16 form: fp32 operands split on the fly into
 * three bf16 values x = h + m + l, six v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation (at least as close to
 * the fp64 result as the fp32 instruction on long reductions).  QEA_MFMA=f32 in the environment keeps every launch on
 * the fp32 instruction.
 *   y[b,oh,ow,n] = epilogue( sum_{kh,kw,c} x[b, oh*sh+kh-ph, ow*sw+kw-pw, c] * w[n,kh,kw,c] )
 * Replaces nn.Conv2d forward at models/model_unet.py:78-109 (3x3 p1, bias=False),
 * models/model_crnn.py:38-45,49-55 (3x3 p1 and the 2x2 p0 conv7); run on
 * flipped/transposed weights it is also their input-gradient; with KH=KW=2, stride 2 it
 * is the input-gradient of nn.ConvTranspose2d (model_unet.py:25-41); with out_mode
 * QEA_OUT_CONVT it is ConvTranspose2d forward; with KH=KW=1 it is the plain GEMM
 * y = x * w^T of nn.LSTM's input projection and nn.Linear (model_crnn.py:9-10,19-20).
 * Cin must be a multiple of 32 (the C_in = 1 layers use qea_conv_c1_*).
 * Epilogue order: v = acc * scale[n] + bias[n]; relu; mask (v = mask[row,n] > 0 ? v : 0);
 * accumulate (y += v).
 * ---------------------------------------------------------------------------------- */
#define QEA_OUT_NHWC 0  /* row m=(b,oh,ow) -> y + m*ldy                                  */
#define QEA_OUT_TBC 1   /* OH==1: row (b,ow) -> y + (ow*B+b)*ldy  (CRNN map_to_sequence,  */
                        /* models/model_crnn.py:23-28)                                    */
#define QEA_OUT_CONVT 2 /* N = 4*Cout', n=(a,bb,co): y[b,2oh+a,2ow+bb,co]; bias[co]        */

typedef struct qea_conv_desc {
  const float* x;     /* input, NHWC, pixel stride ldx                                   */
  const float* w;     /* [N][KH*KW*Cin]                                                   */
  float* y;           /* output                                                           */
  const float* scale; /* [N] or NULL                                                      */
  const float* bias;  /* [N] or NULL  (QEA_OUT_CONVT: [N/4])                              */
  const float* mask;  /* same row/channel indexing as y with stride ldmask, or NULL       */
  int32_t B, H, W, Cin;
  int32_t OH, OW, N;
  int32_t KH, KW, pad_h, pad_w, stride_h, stride_w;
  int32_t ldx, ldy, ldmask;
  int32_t relu;       /* 0/1                                                              */
  int32_t accumulate; /* 0/1                                                              */
  int32_t out_mode;   /* QEA_OUT_*                                                        */
  int32_t tile;       /* 0 = auto; else forced tile config id (tests / tuning)            */
  /* ABI v2: optional PRE-SPLIT operands of the split-bf16 tiles (NULL: split on the fly).  Both in the P3 format written
   * by qea_split_planes: x_planes from x (M = B*H*W rows, Cin channels), w_planes from w (N rows, K = KH*KW*Cin).   */
  const void* x_planes;
  const void* w_planes;
  /* ABI v3: fused BatchNorm batch statistics.  When non-NULL the launch also writes, per partial block k and output column
   * n, stats[(k*N + n)*2 + {0,1}] = fp64 sum / sum of squares of the stored outputs of that block's rows;
   * qea_conv_igemm_stats_blocks(d) gives the block count (0: this launch has no such epilogue — run qea_bn_train_stats).
   * qea_bn_train_stats_from_partials turns the partials into the BatchNorm coefficients. */
  double* stats;
  /* ABI v4: filter of a 3x3 pad-1 stride-1 layer as bf16 planes in MFMA-fragment order (qea_pack_frag_planes): operand of the
   * split-bf16 LDS-halo kernel (tile 24: Cin = 32 or 64k <= 512, N in {32, 64, 128k}; W % 32 == 0 and H % 4 == 0 (8 for Cin = 32),
   * or — round 3, no ABI change — whole small images per tile: 4x16 / 2x8 pixel images with Cin = 64k, N = 128k);
   * qea_conv_igemm_wants_frag_planes(d) tells.  Without it such a launch runs on the fp32 halo / generic tiles. */
  const void* w_frag_planes;
  /* ABI v6: TWO-WAY fp16 split of the LDS-halo kernel (tile 24).  When non-NULL: device pointer to ONE float = the largest finite
   * |x| of the input tensor (qea_absmax), and w_frag_planes must then hold the FP16 planes of qea_pack_frag_planes_f16.  Both
   * operands are scaled by powers of two into fp16's range and split into h + l (11 + 11 bits); a product is three
   * v_mfma_f32_32x32x16_f16 (lh, hl, hh) instead of six bf16 ones, the accumulators are un-scaled in the epilogue (exact).
   * Same accuracy class as the three-way bf16 split, half the matrix instructions, two thirds of the LDS.
   * The hybrid tiles (20-23, 25 with w_planes and no x_planes) take the same switch: w_planes must then come from
   * qea_split_planes_f16.  NULL: the bf16 form. */
  const float* x_absmax;
  /* ABI v6: zero-filled float slot receiving max |v| of the finite values this launch STORES (see qea_bn_apply's absmax_out);
   * honoured by the LDS-halo kernel (tile 24) and by the generic split tiles 20-23 / 25 (incl. QEA_OUT_CONVT); NULL = off. */
  float* y_absmax;
  /* ABI v7: the max-pool that FOLLOWS the layer, fused into the epilogue of the LDS-halo kernel (nothing in the reference: fn.max_pool2d
   * after relu(conv), models/model_crnn.py:49,51; self.poolN(encN), models/model_unet.py:52-59 in the inference pass).  pool_y non-NULL:
   * besides y the launch writes pool_y [B, H/2, W/pool_kw, N] (pixel stride ldpool) = max over 2 x pool_kw windows of the stored
   * values, first maximum in scan order / NaN propagating as qea_maxpool_fwd — bit-identical to that call on y — and folds its
   * abs-max into pool_absmax (zero-filled slot, may be NULL).  Only where qea_conv_igemm_can_pool(d, pool_kw) returns 1 and the fp16
   * operands (x_absmax, w_frag_planes of qea_pack_frag_planes_f16) are given; anything else is QEA_ERR_INVALID. */
  float* pool_y;
  int32_t ldpool;
  int32_t pool_kw;
  float* pool_absmax;
  /* ABI v7: the launch is an INPUT GRADIENT (3x3 dgrad on the LDS-halo kernel, fp16 operands) whose output da feeds the backward of a
   * train-mode BatchNorm(+ReLU) (models/model_unet.py:78-109 under autograd).  bst_y non-NULL (with `stats`): instead of the forward
   * statistics, `stats` receives per partial block the fp64 sums of dz = da * [bst_scale * bst_y + bst_shift > 0] and of
   * dz * (bst_y - mean) * invstd, mean / invstd = bst_stat64[0..N) / [N..2N) — the reductions qea_bn_bwd otherwise makes in a pass of
   * its own over da and bst_y; qea_bn_bwd_from_partials consumes them.  Block count: qea_conv_igemm_stats_blocks. */
  const float* bst_y;
  int32_t ldbst;
  const double* bst_stat64;
  const float* bst_scale;
  const float* bst_shift;
} qea_conv_desc;

int qea_conv_igemm(const qea_conv_desc* d, void* stream);
/* 1 when qea_conv_igemm would run this launch on a split-bf16 tile (so that pre-split operands pay), else 0 */
int qea_conv_igemm_uses_split_bf16(const qea_conv_desc* d);
int qea_conv_igemm_stats_blocks(const qea_conv_desc* d);
/* ABI v7: 1 when this launch (pool_y aside) has an LDS-halo instance with the fused 2 x kw max-pool (kw = 1 or 2) */
int qea_conv_igemm_can_pool(const qea_conv_desc* d, int32_t kw);
/* 1: the launch would run on the LDS-halo 3x3 kernel (tile 24) given w_frag_planes = qea_pack_frag_planes / _f16 of the filter;
 * 2 (ABI v7): it would run on the 1x1 LDS tile (tile 26) given x_absmax and w_frag_planes = qea_pack_frag_planes_f16_1x1; 0: neither */
int qea_conv_igemm_wants_frag_planes(const qea_conv_desc* d);
/* ABI v7 (additive).  Filter [N][K] of a 1x1 stride-1 GEMM (N % 128 == 0, K % 64 == 0) as two fp16 planes in the fragment order of
 * tile 26, scaled from wmax[0] like qea_pack_frag_planes_f16, followed by one float = the inverse scale.  Tile 26 replaces the
 * generic split tiles for the launches behind nn.ConvTranspose2d forward (/root/reference/models/model_unet.py:25-44, 61-73), the BiLSTM
 * input projections and their input gradients (/root/reference/models/model_crnn.py:9) when the two-way fp16 split is on. */
size_t qea_pack_frag_planes_f16_1x1_bytes(int32_t N, int32_t K);
int qea_pack_frag_planes_f16_1x1(const float* w, int32_t N, int32_t K, const float* wmax, void* planes, void* stream);
/* ABI v6.  out[0] = max |x[r*ld + c]| over r < M, c < C with NaN / inf elements ignored (a non-finite element must not decide the
 * scale of the finite ones: it stays non-finite in the products it enters).  One pass over the tensor; `out` is overwritten. */
int qea_absmax(const float* x, int32_t ld, int64_t M, int32_t C, float* out, void* stream);
/* ABI v6.  The filter [N][9][Cin] as two fp16 planes in the fragment order of qea_pack_frag_planes, scaled by the power of two that
 * qea_f16_scale derives from wmax[0] (= qea_absmax of the filter), followed by one float holding the inverse scale.
 * qea_pack_frag_planes_f16_bytes gives the buffer size. */
size_t qea_pack_frag_planes_f16_bytes(int32_t N, int32_t Cin);
/* ABI v6.  The row format of qea_split_planes with TWO fp16 planes per value, scaled by the power of two of xmax[0] (= qea_absmax of
 * x), the 128-byte zero tail, then one float = the inverse scale: the filter operand (w_planes) of the hybrid tiles 20-23 / 25
 * when qea_conv_desc.x_absmax is given. */
size_t qea_split_planes_f16_bytes(int64_t M, int32_t C);
int qea_split_planes_f16(const float* x, int32_t ld, int64_t M, int32_t C, const float* xmax, void* planes, void* stream);
int qea_pack_frag_planes_f16(const float* w, int32_t N, int32_t Cin, const float* wmax, void* planes, void* stream);
size_t qea_pack_frag_planes_bytes(int32_t N, int32_t Cin);
int qea_pack_frag_planes(const float* w, int32_t N, int32_t Cin, void* planes, void* stream);

/* P3 format of a fp32 matrix [M][ld] with C used columns (C % 16 == 0): planes[row][C/16][3][16] bf16 — per 16-column
 * slice the three bf16 planes h, m, l of x = h + m + l (|x - (h+m+l)| <= 2^-24 |x|), 96 contiguous bytes — followed by
 * a 128-byte zero tail (the source of out-of-image taps).  One HBM pass: 4 B read + 6 B written per element, against
 * ~5.5 VALU operations per element for EVERY tap and N-tile when the conv kernel splits on the fly.  Used for the
 * activations / gradients entering nn.Conv2d forward, input-gradient (models/model_unet.py:78-109,
 * models/model_crnn.py:38-45) and for their filters. */
size_t qea_split_planes_bytes(int64_t M, int32_t C);
int qea_split_planes(const float* x, int32_t ld, int64_t M, int32_t C, void* planes, void* stream);

/* ------------------------------------------------------------------------------------
 * Weight gradient on the matrix cores (fp32 MFMA, or the split-bf16 form of qea_conv_igemm for R, C >= 64 with one
 * of them >= 128: tiles 20-22); reduction over pixels, split over blocks, order-fixed second pass: bit-reproducible.
 *   dw[r][kh][kw][c] (+)= sum_{b,ph,pw} p[b,ph,pw][r] * q[b, ph*sh+kh-pad_h, pw*sw+kw-pad_w][c]
 * nn.Conv2d weight gradient (autograd of models/model_unet.py:78-109, model_crnn.py:38-45):
 *   p = dY, q = X.  nn.ConvTranspose2d (model_unet.py:25-41): p = X, q = dY, KH=KW=2,
 *   stride 2, pad 0 -> dw in [Cin][2][2][Cout].  nn.Linear / nn.LSTM weights: KH=KW=1.
 * R, C, ldp, ldq multiples of 4.  Workspace: qea_conv_wgrad_workspace_bytes(d).
 * ---------------------------------------------------------------------------------- */
typedef struct qea_wgrad_desc {
  const float* p;   /* [B*PH*PW][ldp], R channels used                                    */
  const float* q;   /* NHWC image [B,QH,QW] with pixel stride ldq, C channels used        */
  float* dw;        /* [R][KH*KW][C]                                                      */
  void* workspace;
  size_t workspace_bytes;
  int32_t B, PH, PW, QH, QW, R, C;
  int32_t KH, KW, pad_h, pad_w, stride_h, stride_w;
  int32_t ldp, ldq;
  int32_t accumulate; /* dw += result                                                     */
  int32_t splits;     /* 0 = auto                                                         */
  int32_t tile;       /* 0 = auto; 23 = the nine-tap LDS-halo kernel (with both abs-max pointers and R, C multiples of 64: its
                       * producer / consumer form, round 4), 29 = the nine-tap kernel in the round-3 form (every wave stages) */
  /* ABI v6: when BOTH are non-NULL (device pointers to one float each: qea_absmax of p and of q) a launch that runs on a split tile —
   * the nine-tap LDS-halo kernel (3x3 pad 1 stride 1, PW in {16, 32k}, R and C multiples of 32) or tiles 20-22 — takes the TWO-way
   * fp16 split: three MFMAs per product instead of six (see qea_conv_desc.x_absmax).  Other launches ignore them. */
  const float* p_absmax;
  const float* q_absmax;
  /* ABI v8 (the descriptor grew at its end: v7 callers must re-compile).  dbias [R] (NULL = off): the bias gradient = column sums of p
   * (`nn.Conv2d(bias=True)` under autograd, models/model_crnn.py:38-45), accumulated with the same `accumulate` flag as dw.  Taken only by the
   * producer / consumer nine-tap form, where every p element passes through the staging waves' registers anyway (no pass of its own,
   * no qea_colsum launch): ask qea_conv_wgrad_fuses_bias(d) first — with dbias set, any other form refuses the call. */
  float* dbias;
} qea_wgrad_desc;

int qea_conv_wgrad_fuses_bias(const qea_wgrad_desc* d);
size_t qea_conv_wgrad_workspace_bytes(const qea_wgrad_desc* d);
int qea_conv_wgrad(const qea_wgrad_desc* d, void* stream);

/* ------------------------------------------------------------------------------------
 * BatchNorm2d over NHWC rows [M = B*H*W][C] (nn.BatchNorm2d at models/model_unet.py:92,105 and
 * models/model_crnn.py:42,44; mode policy train_nn_patch.py:226-227,312-314, utils.py:113-115).
 * Statistics are accumulated in fp64.  a = y*scale + shift (+ReLU) with
 * scale = gamma*invstd, shift = beta - mean*scale.
 * qea_bn_train_stats : batch mean / biased var -> mean, invstd, scale, shift; running stats
 *                      updated in place with momentum and the UNBIASED variance.
 * qea_bn_eval_coeff  : the same four vectors from the running statistics (conv_bias, if given,
 *                      is folded into shift).
 * qea_bn_bwd         : dz = da * relu'; dgamma = sum dz*xhat, dbeta = sum dz;
 *                      training: dy = scale*(dz - mean(dz) - xhat*mean(dz*xhat)), eval: dy = scale*dz.
 *                      dy may alias da.  relu' comes from a > 0 when a is given, or — without reading
 *                      a — from y*relu_scale + relu_shift > 0 (the very fused multiply-add qea_bn_apply
 *                      evaluates, hence the identical mask); both NULL: no ReLU.
 * stat64 (optional, [2][C] doubles): the batch mean and invstd unrounded; handing it back to
 * qea_bn_bwd keeps the per-channel constants of the backward in fp64 — they are common to every
 * pixel, so fp32 rounding of them is a CORRELATED error that later per-channel sums amplify by M.
 * Workspace for stats / bwd / colsum: qea_colreduce_workspace_bytes(M, C).
 * ---------------------------------------------------------------------------------- */
size_t qea_colreduce_workspace_bytes(int64_t M, int32_t C);
int qea_bn_train_stats(const float* y, int32_t ldy, int64_t M, int32_t C, const float* gamma, const float* beta,
                       float eps, float momentum, float* running_mean, float* running_var, float* mean_out,
                       float* invstd_out, float* scale_out, float* shift_out, double* stat64, void* workspace,
                       size_t workspace_bytes, void* stream);
/* the same, from the per-block partial sums a convolution's fused-statistics epilogue wrote (qea_conv_desc.stats).  The
 * buffer must hold QEA_BN_PARTIAL_SCRATCH_ROWS more rows ([C][2] doubles each) behind the `blocks` partial rows: with many
 * blocks the reduction runs in two order-fixed stages through them. */
#define QEA_BN_PARTIAL_SCRATCH_ROWS 256
int qea_bn_train_stats_from_partials(const double* partials, int32_t blocks, int64_t M, int32_t C, const float* gamma,
                                     const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                     float* mean_out, float* invstd_out, float* scale_out, float* shift_out, double* stat64,
                                     void* stream);
int qea_bn_eval_coeff(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, const float* conv_bias, float* mean_out,
                      float* invstd_out, float* scale_out, float* shift_out, void* stream);
/* absmax_out (ABI v6; here and in qea_bn_bwd / qea_maxpool_fwd / qea_maxpool_bwd; NULL = off): a zero-filled float slot into which the
 * kernel folds max |v| of the finite values it stores (atomicMax on the bits) — the scale source of the two-way fp16 split for the
 * conv / wgrad launch that consumes the tensor (qea_conv_desc.x_absmax, qea_wgrad_desc.p_absmax / q_absmax), carried by the
 * producer instead of a qea_absmax pass.  Several launches may share one slot (per-group BatchNorm, the halves of a concat). */
int qea_bn_apply(const float* y, int32_t ldy, float* a, int32_t lda, int64_t M, int32_t C, const float* scale,
                 const float* shift, int32_t relu, float* absmax_out, void* stream);
int qea_bn_bwd(const float* da, int32_t ldda, const float* a, int32_t lda, const float* relu_scale,
               const float* relu_shift, const float* y, int32_t ldy, int64_t M, int32_t C, const float* gamma, const float* mean, const float* invstd, const double* stat64,
               int32_t training, float* dgamma, float* dbeta, int32_t accumulate_param_grads, float* dy,
               int32_t lddy, void* workspace, size_t workspace_bytes, float* absmax_out, void* stream);
/* ABI v8 (additive).  qea_maxpool_bwd(accumulate) + qea_bn_bwd in one: the BatchNorm(+ReLU) backward of a block whose output went to a
 * 2 x kw max-pool (kw in {1, 2}; model_unet.py:52-59 pool1-4, model_crnn.py:53-54 the (2,1) pools) and possibly to a skip connection.
 * da [B*H*W][ldda] = the gradient from the skip path (NULL: none), dpool [B*(H/2)*(W/kw)][lddp] = the gradient of the pooled tensor;
 * the pool's winners and the ReLU mask are recomputed from y with relu_scale / relu_shift (required), so neither the activation nor
 * the summed gradient is read or written.  Same values as the two calls (the fp64 reductions visit the pixels in another order).
 * Workspace: qea_colreduce_workspace_bytes(B*H*W, C). */
int qea_bn_bwd_pool(const float* da, int32_t ldda, const float* dpool, int32_t lddp, int32_t kw, const float* relu_scale,
                    const float* relu_shift, const float* y, int32_t ldy, int32_t B, int32_t H, int32_t W, int32_t C, const float* gamma,
                    const float* mean, const float* invstd, const double* stat64, int32_t training, float* dgamma, float* dbeta,
                    int32_t accumulate_param_grads, float* dy, int32_t lddy, void* workspace, size_t workspace_bytes, float* absmax_out,
                    void* stream);
/* ABI v7 (additive).  qea_bn_bwd with the two per-channel reductions taken from `partials` [blocks (+ 256 scratch rows)][C][2] fp64, as
 * written by the producing input-gradient launch (qea_conv_desc.bst_y), instead of a pass over da and y: finalize + the elementwise
 * pass only.  stat64 and relu_scale / relu_shift are required (what the producer used); workspace: 3 * C doubles. */
int qea_bn_bwd_from_partials(const double* partials, int32_t blocks, const float* da, int32_t ldda, const float* relu_scale,
                             const float* relu_shift, const float* y, int32_t ldy, int64_t M, int32_t C, const float* gamma,
                             const float* mean, const float* invstd, const double* stat64, int32_t training, float* dgamma, float* dbeta,
                             int32_t accumulate_param_grads, float* dy, int32_t lddy, void* workspace, size_t workspace_bytes,
                             float* absmax_out, void* stream);
/* out[c] (+)= sum_m x[m][c]  — conv / linear / LSTM bias gradients */
int qea_colsum(const float* x, int32_t ldx, int64_t M, int32_t C, float* out, int32_t accumulate, void* workspace,
               size_t workspace_bytes, void* stream);

/* Max-pool, window == stride (nn.MaxPool2d(2,2) model_unet.py:14-20; fn.max_pool2d (2,2)/(2,1)
 * model_crnn.py:48-54).  First maximum in (kh,kw) scan order wins, as ATen.  bwd recomputes the
 * arg-max from x; relu_mask additionally zeroes the gradient where the maximum is <= 0 (pool of
 * a ReLU output); accumulate: dx += (UNet skip connections). */
int qea_maxpool_fwd(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t B, int32_t H, int32_t W, int32_t C,
                    int32_t kh, int32_t kw, float* absmax_out, void* stream);
/* ABI v7 (additive).  qea_bn_apply followed by qea_maxpool_fwd in ONE pass over y [B,H,W,C]: a = relu?(y*scale + shift) at full
 * resolution (pixel stride lda) and pooled = max-pool(a) (pixel stride ldp), both bit-identical to the two separate calls, which
 * read the activation a second time (model_unet.py:51-59: encoder block -> MaxPool2d(2,2)); absmax_a / absmax_pooled as absmax_out. */
int qea_bn_apply_pool(const float* y, int32_t ldy, float* a, int32_t lda, float* pooled, int32_t ldp, int32_t B, int32_t H, int32_t W,
                      int32_t C, const float* scale, const float* shift, int32_t relu, int32_t kh, int32_t kw, float* absmax_a,
                      float* absmax_pooled, void* stream);
int qea_maxpool_bwd(const float* x, int32_t ldx, const float* dy, int32_t lddy, float* dx, int32_t lddx, int32_t B,
                    int32_t H, int32_t W, int32_t C, int32_t kh, int32_t kw, int32_t relu_mask, int32_t accumulate,
                    float* absmax_out, void* stream);

/* Filter re-layouts for the gradient GEMMs (run once per optimiser step):
 * out[c][r] = in[r][c];  wt[ci][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][ci]. */
int qea_transpose2d(const float* in, float* out, int32_t R, int32_t Cc, void* stream);
int qea_filter_flip_transpose(const float* w, float* wt, int32_t Co, int32_t Ci, int32_t KH, int32_t KW, void* stream);

/* 3x3 pad-1 convolution with ONE input channel (UNet enc1conv1 model_unet.py:13; CRNN conv1
 * model_crnn.py:37,48): x [B,H,W], w [Co][9], y NHWC.  wgrad also yields the bias gradient
 * (db may be NULL); dgrad: dx[B,H,W] (+)= sum_{tap,co} dy * w. */
int qea_conv_c1_fwd(const float* x, const float* w, const float* bias, float* y, int32_t ldy, int32_t B, int32_t H,
                    int32_t W, int32_t Co, int32_t relu, void* stream);
/* ABI v7 (additive).  qea_conv_c1_fwd followed by the 2x2 max-pool in one pass (CRNN conv1 -> ReLU -> max_pool2d(2,2),
 * model_crnn.py:48): y (full resolution: the pool's backward reads it) and pooled [B,H/2,W/2,Co], bit-identical to the two calls.
 * Needs H % 2 == 0, W % 4 == 0, Co in {32, 64, 128}; absmax_pooled as qea_maxpool_fwd's absmax_out. */
int qea_conv_c1_fwd_pool(const float* x, const float* w, const float* bias, float* y, int32_t ldy, float* pooled, int32_t ldp, int32_t B,
                         int32_t H, int32_t W, int32_t Co, int32_t relu, float* absmax_pooled, void* stream);
/* ABI v8 (additive).  The whole backward of conv1 -> ReLU -> max_pool2d(2, 2) (models/model_crnn.py:37-38,47-48) from the POOLED tensor's
 * gradient and the 1-channel input: the full-resolution activation is rebuilt from x (nine multiply-adds per element, the forward's own
 * chain, so the pool's winners and the ReLU mask are the forward's bit for bit), neither it nor its gradient is read or written —
 * qea_conv_c1_fwd_pool may therefore be given y = NULL.  dpool [B*(H/2)*(W/2)][lddp] is OVERWRITTEN with its ReLU-masked values;
 * dw [Co][3][3], db [Co] (accumulate as in qea_conv_c1_wgrad; dw NULL: no parameter gradients), dx [B][H][W] (NULL: not wanted).
 * Co = 64.  Same values as qea_maxpool_bwd(relu_mask) + qea_conv_c1_wgrad + qea_conv_c1_dgrad up to the order of the fp32 partial sums. */
size_t qea_conv_c1_pool_bwd_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Co, int32_t need_dx);
int qea_conv_c1_pool_bwd(const float* x, const float* w, const float* bias, float* dpool, int32_t lddp, float* dw, float* db, float* dx,
                         int32_t B, int32_t H, int32_t W, int32_t Co, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);
size_t qea_conv_c1_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Co);
int qea_conv_c1_wgrad(const float* x, const float* dy, int32_t lddy, float* dw, float* db, int32_t B, int32_t H,
                      int32_t W, int32_t Co, int32_t accumulate, void* workspace, size_t workspace_bytes,
                      void* stream);
int qea_conv_c1_dgrad(const float* dy, int32_t lddy, const float* w, float* dx, int32_t B, int32_t H, int32_t W,
                      int32_t Co, int32_t accumulate, void* stream);

/* UNet head: y[m] = sigmoid(<x[m,:], w> + b)  (nn.Conv2d 1x1 + torch.sigmoid, model_unet.py:45,76)
 * bwd: dz = dyy*y*(1-y); dx = dz*w; dw (+)= sum dz*x; db (+)= sum dz. */
int qea_head_fwd(const float* x, int32_t ldx, const float* w, const float* b, float* y, int64_t M, int32_t C,
                 void* stream);
size_t qea_head_bwd_workspace_bytes(int64_t M, int32_t C);
int qea_head_bwd(const float* x, int32_t ldx, const float* y, const float* dyy, const float* w, float* dx,
                 int32_t lddx, float* dw, float* db, int32_t accumulate, int64_t M, int32_t C, void* workspace,
                 size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * One bidirectional LSTM layer, hidden 256 (nn.LSTM(512,256,2,bidirectional=True),
 * models/model_crnn.py:9,19; gate order i,f,g,o; zero initial state).
 * Buffers (seq-first, both directions side by side):
 *   gates [T][B][2][4][256]  in: x*W_ih^T + b_ih + b_hh (from qea_conv_igemm), out: activations
 *   c     [T][B][2][256], y [T][B][2][256] (= the layer output)
 * qea_lstm_pack_whh re-orders one direction's W_hh [1024][256] into per-lane MFMA fragment
 * order for the forward (h*W_hh^T) and backward (dgates*W_hh) step GEMMs; the two directions'
 * packed copies must be contiguous ([2][1024*256]).
 * qea_lstm_layer_bwd turns `gates` in place into pre-activation gate gradients given dy;
 * dX, dW_ih, dW_hh, db then follow from qea_conv_igemm / qea_conv_wgrad / qea_colsum.
 * ---------------------------------------------------------------------------------- */
int qea_lstm_pack_whh(const float* w_hh, float* packed_fwd, float* packed_bwd, void* stream);
int qea_lstm_layer_fwd(float* gates, float* c, float* y, const float* packed_fwd, int32_t T, int32_t B, void* stream);
int qea_lstm_layer_bwd(float* gates, const float* c, const float* dy, const float* packed_bwd, float* dc_scratch,
                       int32_t T, int32_t B, void* stream);
/* ABI v5: the same layer with the recurrent GEMMs in split-bf16 form (three bf16 planes per fp32 value, six
 * v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate: the arithmetic of the split convolution kernels).
 * qea_lstm_pack_whh_split writes one direction's W_hh as pre-split planes in MFMA-fragment order
 * (qea_lstm_pack_whh_split_bytes() bytes for each of the forward and backward forms; the two directions' copies
 * contiguous, as above); h_prev / dgates rows are split on the fly.  The forward step of batches of >= 1 536 rows runs
 * 128-row workgroups whose weight fragments pass through LDS once per workgroup (LDS-DMA); everything else the 32-row
 * shape of the fp32 step. */
size_t qea_lstm_pack_whh_split_bytes(void);
int qea_lstm_pack_whh_split(const float* w_hh, void* planes_fwd, void* planes_bwd, void* stream);
int qea_lstm_layer_fwd_split(float* gates, float* c, float* y, const void* planes_fwd, int32_t T, int32_t B, void* stream);
int qea_lstm_layer_bwd_split(float* gates, const float* c, const float* dy, const void* planes_bwd, float* dc_scratch,
                             int32_t T, int32_t B, void* stream);

/* ABI v9: the same layer in ONE launch per pass (csrc/lstm_seq.hip): the time loop runs inside the kernel, a workgroup keeps its
 * W_hh slice in LDS for all T steps as two fp16 planes (qea_lstm_seq_pack: qea_lstm_seq_pack_bytes() bytes per direction and form,
 * the two directions' copies contiguous; w_absmax: a device float >= the direction's largest |W_hh|, e.g. qea_absmax's — the scale
 * source of the planes; the layer calls take the two directions' values as [2] floats), c / dc stay in registers, and the eight workgroups of a (row block, direction) exchange h[t] (forward) or the gate
 * gradients (backward) through global memory with write-through stores, one arrival counter per group and sc1 loads — no
 * grid-wide barrier, no residency requirement beyond the group's own eight workgroups, every spin bounded (on a timeout the outputs
 * are NaN).  Three v_mfma_f32_32x32x16_f16 per product (|h| < 1 and W_hh by its abs-max; the gate gradients by a per-row,
 * per-64-column-chunk scale taken in the kernel).  workspace: qea_lstm_seq_workspace_bytes(B) bytes, zeroed by the call itself.
 * Buffers and results as qea_lstm_layer_fwd / _bwd (no dc_scratch).  y_absmax / dgates_absmax: NULL, or a ZEROED device float that
 * receives the largest finite |y| / |gate gradient| of the pass (the scale source of the GEMMs that read the tensor next: no separate
 * abs-max pass over it). */
size_t qea_lstm_seq_pack_bytes(void);
size_t qea_lstm_seq_workspace_bytes(int32_t B);
int qea_lstm_seq_pack(const float* w_hh, void* planes_fwd, void* planes_bwd, const float* w_absmax, void* stream);
int qea_lstm_seq_fwd(float* gates, float* c, float* y, const void* planes_fwd, const float* w_absmax, int32_t T, int32_t B,
                     void* workspace, float* y_absmax, void* stream);
int qea_lstm_seq_bwd(float* gates, const float* c, const float* dy, const void* planes_bwd, const float* w_absmax, int32_t T,
                     int32_t B, void* workspace, float* dgates_absmax, void* stream);

/* ABI v9 (additive).  Several derived weight forms in ONE launch (a model has ~110 of them per optimiser step, a few microseconds each):
 * kind 0 = qea_filter_flip_transpose (a, b, c, d = Co, Ci, KH, KW; amax unused), kind 1 = qea_pack_frag_planes_f16 (a, b = N, Cin),
 * kind 2 = qea_pack_frag_planes_f16_1x1 (a, b = N, K); same bytes as the single calls.  `jobs` is a HOST array of 1..64 jobs; a job
 * must not read what another job of the same call writes (flip first, then pack the flipped filters in a second call). */
typedef struct qea_wform_job {
  const float* src;
  void* dst;
  const float* amax;
  int32_t kind, a, b, c, d;
} qea_wform_job;
int qea_weight_forms_multi(const qea_wform_job* jobs, int32_t n, void* stream);

/* log_softmax over the last dim (fn.log_softmax(.., 2), model_crnn.py:20) and its backward
 * dx = g - exp(lp)*sum(g), with the reference's NaN scrub (CRNN.backward_hook,
 * model_crnn.py:30-32, registered at train_nn_patch.py:94) when nan_scrub != 0; columns
 * [C, Cpad) of dx are zeroed. */
int qea_log_softmax_fwd(const float* x, int32_t ldx, float* y, int32_t ldy, int64_t M, int32_t C, void* stream);
int qea_log_softmax_bwd(const float* g, int32_t ldg, const float* lp, int32_t ldlp, float* dx, int32_t lddx, int64_t M,
                        int32_t C, int32_t Cpad, int32_t nan_scrub, void* stream);

/* CTC loss (torch.nn.CTCLoss, blank 0, zero_infinity=False; train_nn_patch.py:143,178,294,
 * train_nn_area.py:146-147,265).  lp element (t,n,c) at t*ld_t + n*ld_n + c.  targets:
 * concatenated int32 (device), target_offsets[n] = start of sample n.  reduction 1 = 'mean'
 * (mean_n nll_n/max(len_n,1)), 0 = 'none' (loss = sum, nll[] per sample).  grad (may be NULL)
 * = d(reduced loss * grad_scale)/d lp exactly as ATen forms it (NaN for infeasible samples).
 * S_max >= 2*max(target_len)+1, <= 256. */
size_t qea_ctc_workspace_bytes(int32_t T, int32_t N, int32_t S_max);
int qea_ctc_loss(const float* lp, int32_t ld_t, int32_t ld_n, const int32_t* targets, const int64_t* target_offsets,
                 const int32_t* input_lengths, const int32_t* target_lengths, int32_t T, int32_t N, int32_t C,
                 int32_t blank, int32_t S_max, int32_t reduction, float grad_scale, float* nll, float* loss,
                 float* grad, int32_t gld_t, int32_t gld_n, void* workspace, size_t workspace_bytes, void* stream);

/* torch.optim.Adam step over one flat fp32 buffer (train_nn_patch.py:146-152,308-309,343-345):
 * g' = g*grad_scale + wd*p; m,v update; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps). */
int qea_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int64_t step, float grad_scale, void* stream);
/* The same update with the step count held on the device (torch.optim.Adam(capturable=True)): step[0] (a float,
 * as torch keeps it) is incremented by a one-thread kernel which also writes the two bias-correction
 * coefficients to coef[0..1]; no launch argument depends on the step count, so the pair of launches can be
 * recorded into a hipGraph and replayed. */
int qea_adam_step_capturable(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, float* step, float* coef, float grad_scale,
                             void* stream);

/* AddGaussianNoice (transform_helper.py:33-45) for R replicas of K images fused into the batch
 * dim: out[r*K+k] = clamp(img[k] - coef*sigma[r*K+k]*N(0,1), 0, 1); Philox4x32-10 keyed by
 * (seed, offset); noise_out (optional) receives sigma*N(0,1).  qea_jitter_apply takes the
 * noise as an input instead (parity tests; train_nn_area.py:184-191 returns the noise). */
int qea_jitter(const float* img, const float* sigma, float* out, float* noise_out, int32_t K, int32_t R, int32_t HW,
               float coef, uint64_t seed, uint64_t offset, void* stream);
int qea_jitter_apply(const float* img, const float* noise, float* out, int32_t K, int32_t R, int32_t HW, float coef,
                     void* stream);

/* TopKCERSampler.query ranking (selection_utils.py:144-151): indices of the k largest keys,
 * descending, equal keys in ascending index order (stable); n <= 16384. */
int qea_topk_desc_stable(const float* keys, int32_t n, int32_t k, int64_t* idx_out, void* stream);

/* utils.get_text_stack / padder (utils.py:118-141): crop boxes (x0,y0,x1,y1, already clipped to
 * the image) out of a [H][W] image, centre on a white OH x OW canvas; scatter is its backward
 * (adds into dimg). */
int qea_crop_pad_gather(const float* img, int32_t H, int32_t W, const int32_t* boxes, int32_t N, int32_t OH, int32_t OW,
                        float* out, void* stream);
int qea_crop_pad_scatter(const float* dout, const int32_t* boxes, int32_t N, int32_t OH, int32_t OW, float* dimg,
                         int32_t H, int32_t W, void* stream);

/* utils.pred_to_string (utils.py:74-92): per-step argmax (first maximum), collapse repeats, drop
 * blank -> tokens [N][T] int32 + lengths [N]; scores element (t,n,c) at t*ld_t + n*ld_n + c. */
int qea_greedy_decode(const float* scores, int32_t ld_t, int32_t ld_n, int32_t T, int32_t N, int32_t C, int32_t blank,
                      int32_t* tokens, int32_t* lengths, void* stream);

/* Unit-cost edit distance between the greedy decode of each sample (pred_tokens [N][ldp], pred_len)
 * and its ground-truth token string (concatenated gt_tokens, gt_offsets, gt_len): the numerator of
 * utils.compare_labels' CER (utils.py:95-110, python-Levenshtein `distance`); the host divides by
 * max(1, len(gt)) in float64 as the reference does.  ldp <= 128. */
int qea_edit_distance(const int32_t* pred_tokens, int32_t ldp, const int32_t* pred_len, const int32_t* gt_tokens,
                      const int64_t* gt_offsets, const int32_t* gt_len, int32_t N, int32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QEA_HIP_H */
