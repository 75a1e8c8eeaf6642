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

/* ABI version; bumped whenever a struct below changes. */
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

/* ------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
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
} qea_conv_desc;

int qea_conv_igemm(const qea_conv_desc* d, void* stream);

/* ------------------------------------------------------------------------------------
 * Weight gradient on the fp32 matrix cores (reduction over pixels, split over blocks,
 * order-fixed second pass: bit-reproducible).
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
  int32_t tile;       /* 0 = auto                                                         */
} qea_wgrad_desc;

size_t qea_conv_wgrad_workspace_bytes(const qea_wgrad_desc* d);
int qea_conv_wgrad(const qea_wgrad_desc* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QEA_HIP_H */
