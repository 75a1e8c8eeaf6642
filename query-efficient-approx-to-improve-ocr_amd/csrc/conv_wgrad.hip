// Weight gradient of the implicit-GEMM convolution on the fp32 matrix cores:
//
//   dW[r][tap][c] = sum_{m = (b,ph,pw)}  P[m][r] * Q[b, ph*sh + kh - pad_h, pw*sw + kw - pad_w][c]
//
// conv2d:          P = dY (r = output channel), Q = X (c = input channel)
// convTranspose2d: P = X  (r = input channel),  Q = dY (c = output channel), stride 2, pad 0
// linear / LSTM:   KH = KW = 1
//
// The reduction index is the pixel m, the slow dimension of both operands, so tiles are
// staged pixel-major ([16 or 32 pixels][channels]) and MFMA fragments are read column-wise
// (lane l: channels {MI*(l&31) + i} of pixel 2s + (l>>5) at MFMA step s, one ds_read_b32/b64).
// The pixel range is cut into splits sized so that tiles x splits fills the chip a whole number
// of times; the (split, tile) work list is handed to the XCDs in contiguous runs; partial
// results go to a workspace slab per split and a second, order-fixed pass sums them
// (bit-reproducible, no float atomics).
//
// Roofline: MFMA-bound for wide layers (2*M*R*taps*C flops), L2/HBM-leaning for the 32/64-
// channel UNet levels where each pixel carries only 2*R*C*taps flops per (R+C)*4 bytes.
#include "common.h"
#include <type_traits>

namespace {

struct WgArgs {
  const float* p;
  const float* q;
  float* out;  // workspace slab base (splits > 1) or dW
  int B, PH, PW, QH, QW, R, C, KH, KW, pad_h, pad_w, stride_h, stride_w, ldp, ldq;
  int M, chunk, splits, accumulate;
  int r_tiles, c_tiles, tiles;  // tiles = r_tiles * taps * c_tiles
  long long slab;               // floats per split slab (R*taps*C)
  const float* pmax;            // abs-max of P / Q (two-way fp16 split, ABI v6) or null
  const float* qmax;
};

constexpr int BKP_MAX = 32;

// Per-thread operand fetch shared by the fp32 and the split-bf16 kernel.  The 256 threads cover the BKP pixel rows of a
// stage with TPR threads per row; a thread keeps ONE pixel row and loads P_LD float4 chunks of P and Q_LD of Q from it
// (chunk gl + i*TPR), so a single pixel -> (image, row, column) split serves all its loads.  The split is done once and
// then advanced by BKP pixels per stage with two conditional carries: no divisions and no branches inside the main loop
// (out-of-range slots load from a safe address; p_ok / q_ok tell the staging code which slots to zero).
template <int P_LD, int Q_LD, int BKP>
struct WgGather {
  static constexpr int TPR = 256 / BKP;
  int grow, gl;
  int g_img, g_ph, g_pw;       // image offset (pixels of q), output row / column of this thread's current pixel
  int qhw, dr, er, eq;         // BKP pixels = eq images + er rows + dr columns
  int kh, kw, m_end;
  unsigned colp_ok, colq_ok;   // bit i: chunk i of this thread lies inside R (C)
  unsigned p_ok, q_ok;         // bit i: slot i holds real data
  const float* p_ptr;
  const float* q_col;
  f32x4 p_reg[P_LD], q_reg[Q_LD];

  __device__ __forceinline__ void init(const WgArgs& a, int tid, int m_begin, int m_end_, int r0, int c0, int kh_, int kw_) {
    grow = tid / TPR;
    gl = tid % TPR;
    kh = kh_;
    kw = kw_;
    m_end = m_end_;
    qhw = a.QH * a.QW;
    const int dq = BKP / a.PW;
    dr = BKP - dq * a.PW;
    eq = dq / a.PH;
    er = dq - eq * a.PH;
    const int phw = a.PH * a.PW;
    const int m = m_begin + grow;
    const int b = m / phw;
    const int rem = m - b * phw;
    g_ph = rem / a.PW;
    g_pw = rem - g_ph * a.PW;
    g_img = b * qhw;
    colp_ok = colq_ok = p_ok = q_ok = 0;
#pragma unroll
    for (int i = 0; i < P_LD; ++i) colp_ok |= (unsigned)(r0 + (gl + i * TPR) * 4 < a.R) << i;
#pragma unroll
    for (int i = 0; i < Q_LD; ++i) colq_ok |= (unsigned)(c0 + (gl + i * TPR) * 4 < a.C) << i;
    p_ptr = a.p + (size_t)(m_begin + grow) * a.ldp + r0 + gl * 4;
    q_col = a.q + c0 + gl * 4;
  }

  // issue the loads of the stage that starts at pixel mbase, then advance the pixel by BKP
  __device__ __forceinline__ void fetch(const WgArgs& a, int mbase) {
    const bool row_ok = mbase + grow < m_end;
    const int qh = g_ph * a.stride_h + kh - a.pad_h;
    const int qw = g_pw * a.stride_w + kw - a.pad_w;
    const bool pix_ok = row_ok && (unsigned)qh < (unsigned)a.QH && (unsigned)qw < (unsigned)a.QW;
    const float* q_ptr = q_col + (size_t)(pix_ok ? g_img + qh * a.QW + qw : 0) * a.ldq;
    p_ok = row_ok ? colp_ok : 0u;
    q_ok = pix_ok ? colq_ok : 0u;
#pragma unroll
    for (int i = 0; i < P_LD; ++i) p_reg[i] = *reinterpret_cast<const f32x4*>(((p_ok >> i) & 1) ? p_ptr + i * TPR * 4 : a.p);
#pragma unroll
    for (int i = 0; i < Q_LD; ++i) q_reg[i] = *reinterpret_cast<const f32x4*>(((q_ok >> i) & 1) ? q_ptr + i * TPR * 4 : a.q);
    p_ptr += (size_t)BKP * a.ldp;
    int pw = g_pw + dr, ph = g_ph + er, img = g_img + eq * qhw;
    const bool cw = pw >= a.PW;
    pw -= cw ? a.PW : 0;
    ph += cw ? 1 : 0;
    const bool chh = ph >= a.PH;
    ph -= chh ? a.PH : 0;
    img += chh ? qhw : 0;
    g_pw = pw;
    g_ph = ph;
    g_img = img;
  }
};  // pixels per stage (split chunks are rounded to this)

template <int BR, int BC, int WR, int WC, int WK, int BKP = 32>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgArgs a) {
  constexpr int TR = BR / WR, TCc = BC / WC;
  constexpr int MI = TR / 32, NJ = TCc / 32;
  constexpr int P_LD = BKP * BR / 4 / 256;  // float4 loads per thread per stage (P)
  constexpr int Q_LD = BKP * BC / 4 / 256;
  constexpr int STEPS = (BKP / 2) / WK;  // MFMA k-steps per wave per stage
  static_assert(WR * WC * WK == 4, "4 waves");
  static_assert(P_LD >= 1 && Q_LD >= 1 && STEPS >= 1, "stage too small for 256 threads");
  static_assert(MI <= 2 && NJ <= 2, "fragment reads are b32 / b64");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ps = smem;                 // [2][BKP][BR]
  float* Qs = smem + 2 * BKP * BR;  // [2][BKP][BC]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / (WR * WC);
  const int wrc = wave % (WR * WC);
  const int wr = wrc / WC, wc = wrc % WC;

  // block -> (split, tile): every XCD (blocks with equal id % 8 under round-robin dispatch) takes a contiguous run
  // of the (split-major) work list, so the r x tap x c tiles of one pixel chunk run on ONE XCD at about the same
  // time and its P / Q rows are fetched into that L2 once instead of once per tap and XCD (PMC: 8x the algorithmic
  // fetch on the 64/128-channel levels with the plain blockIdx order)
  const int vid = qea_xcd_swizzle(blockIdx.x, a.tiles * a.splits);
  const int split = vid / a.tiles;
  const int tile = vid - split * a.tiles;
  const int taps = a.KH * a.KW;
  const int c_tile = tile % a.c_tiles;
  const int tap = (tile / a.c_tiles) % taps;
  const int r_tile = tile / (a.c_tiles * taps);
  const int r0 = r_tile * BR, c0 = c_tile * BC;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;

  const int m_begin = split * a.chunk;
  const int m_end = min(a.M, m_begin + a.chunk);

  constexpr int TPR = 256 / BKP;
  static_assert(P_LD * TPR * 4 == BR && Q_LD * TPR * 4 == BC, "chunks per thread");
  WgGather<P_LD, Q_LD, BKP> g;
  g.init(a, tid, m_begin, m_end, r0, c0, kh, kw);
  auto gather = [&](int mbase) { g.fetch(a, mbase); };
  auto stage = [&](int buf) {
    float* pd = Ps + buf * BKP * BR + g.grow * BR + g.gl * 4;
    float* qd = Qs + buf * BKP * BC + g.grow * BC + g.gl * 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < P_LD; ++i) *reinterpret_cast<f32x4*>(pd + i * TPR * 4) = ((g.p_ok >> i) & 1) ? g.p_reg[i] : zero;
#pragma unroll
    for (int i = 0; i < Q_LD; ++i) *reinterpret_cast<f32x4*>(qd + i * TPR * 4) = ((g.q_ok >> i) & 1) ? g.q_reg[i] : zero;
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  const int nstage = (m_end > m_begin) ? (m_end - m_begin + BKP - 1) / BKP : 0;

  if (nstage > 0) {
    gather(m_begin);
    stage(0);
  }
  __syncthreads();
  // fragment reads: a lane takes MI (NJ) CONSECUTIVE channels with one ds_read_b32/b64 and feeds them to
  // MI (NJ) different accumulator tiles, i.e. tile i holds channels {MI*row + i} of the wave's block
  // (interleaved, undone in the epilogue) — half the LDS instructions of one read per tile.  The fragments of
  // k-step s+1 are read under the MFMAs of k-step s (register double buffer; hipcc does not pipeline them itself).
  auto read_frag = [&](const float* ps, const float* qs, int s, float (&af)[MI], float (&bf)[NJ]) {
    if constexpr (MI == 2) {
      const float2 t = *reinterpret_cast<const float2*>(ps + s * 2 * BR);
      af[0] = t.x;
      af[1] = t.y;
    } else {
      af[0] = ps[s * 2 * BR];
    }
    if constexpr (NJ == 2) {
      const float2 t = *reinterpret_cast<const float2*>(qs + s * 2 * BC);
      bf[0] = t.x;
      bf[1] = t.y;
    } else {
      bf[0] = qs[s * 2 * BC];
    }
  };
  for (int st = 0; st < nstage; ++st) {
    const int cur = st & 1;
    if (st + 1 < nstage) gather(m_begin + (st + 1) * BKP);
    const float* ps = Ps + cur * BKP * BR + (wk * (BKP / WK) + fh) * BR + wr * TR + fr * MI;
    const float* qs = Qs + cur * BKP * BC + (wk * (BKP / WK) + fh) * BC + wc * TCc + fr * NJ;
    float af[2][MI], bf[2][NJ];
    read_frag(ps, qs, 0, af[0], bf[0]);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int c = s & 1, n = c ^ 1;
      if (s + 1 < STEPS) read_frag(ps, qs, s + 1, af[n], bf[n]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i], bf[c][j], acc[i][j], 0, 0, 0);
      if (s + 1 < STEPS) {
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);        // the two fragment reads of step s+1 ...
        __builtin_amdgcn_sched_group_barrier(0x008, MI * NJ, 0);  // ... ahead of the MFMAs of step s
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, MI * NJ, 0);
      }
    }
    if (st + 1 < nstage) stage(cur ^ 1);
    __syncthreads();
  }

  // ---- cross-wave reduction of the K-split (through LDS), then store ----
  float* outp = a.out + (size_t)split * a.slab;
  const int ktot = taps * a.C;
  if constexpr (WK > 1) {
    float* red = smem;  // [WK][BR][BC]   (staging buffers are dead after the last barrier)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = wr * TR + MI * ((r & 3) + 8 * (r >> 2) + 4 * fh) + i;
          const int cc = wc * TCc + NJ * fr + j;
          red[(wk * BR + rr) * BC + cc] = acc[i][j][r];
        }
    __syncthreads();
    for (int e = tid; e < BR * BC; e += 256) {
      const int rr = e / BC, cc = e - rr * BC;
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < WK; ++k) v += red[(k * BR + rr) * BC + cc];
      if (r0 + rr < a.R && c0 + cc < a.C) {
        const size_t o = (size_t)(r0 + rr) * ktot + tap * a.C + c0 + cc;
        if (a.accumulate) v += outp[o];
        outp[o] = v;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = r0 + wr * TR + MI * ((r & 3) + 8 * (r >> 2) + 4 * fh) + i;
          const int cc = c0 + wc * TCc + NJ * fr + j;
          if (rr < a.R && cc < a.C) {
            const size_t o = (size_t)rr * ktot + tap * a.C + cc;
            float v = acc[i][j][r];
            if (a.accumulate) v += outp[o];
            outp[o] = v;
          }
        }
  }
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 form of the kernel above (tiles 20-22): P and Q are split into three bf16 planes while they are
// staged (x = h + m + l, qea_split3) and every product is formed by six v_mfma_f32_32x32x16_bf16 with fp32
// accumulation.  The LDS planes keep the natural [pixel][channel] order; the MFMA operands (32 channels x 16 pixels,
// 8 consecutive pixels per lane) come out of them through the gfx950 transposing read ds_read_b64_tr_b16: a 16-lane
// group reads a 4-pixel x 16-channel block and lane i receives channel i of the 4 pixels, two reads per operand.
// Row stride = channels*2 + 64 bytes: the four pixel rows of a read fall on four different 64-byte bank ranges.
// ---------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 tr_frag(const __bf16* base, int row_stride) {
  // base: element address of (pixel 0 of this lane's 4-pixel block, this lane's 4-channel chunk); second block 4 pixels further
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + 4 * row_stride));
  // whole-vector reinterpretation (an element-by-element bit_cast of the result was folded into a splat of element 0)
  return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}

// NPL = 2 (round 3, ABI v6): the two-way fp16 split — P and Q scaled by the powers of two of their abs-max (a.pmax / a.qmax),
// three MFMAs per product, the block un-scaled when it is written (exact).
template <int BR, int BC, int WR, int WC, int NPL = 3>
__global__ __launch_bounds__(256) void wgrad_bf3_kernel(const WgArgs a) {
  constexpr bool F16 = NPL == 2;
  typedef typename std::conditional<F16, f16x8, bf16x8>::type frag_t;
  constexpr int BKP = 16;
  constexpr int TR = BR / WR, TCc = BC / WC;
  constexpr int MI = TR / 32, NJ = TCc / 32;
  constexpr int TPR = 256 / BKP;
  constexpr int P_LD = BR / (4 * TPR), Q_LD = BC / (4 * TPR);
  // LDS rows = pixels.  A transposing read takes 4 consecutive pixels x one 64-byte (32-channel) chunk: with 128+ channels
  // per row the rows are unpadded and the 64-byte chunks of pixel p are stored at chunk index c ^ (p & 3), so the 4 rows
  // of a read fall on 4 different bank ranges; 64-channel rows (2 chunks) keep a 64-byte pad instead.
  constexpr int PR = BR >= 128 ? BR : BR + 32, QR = BC >= 128 ? BC : BC + 32;
  constexpr bool SWP = BR >= 128, SWQ = BC >= 128;
  static_assert(WR * WC == 4 && P_LD >= 1 && Q_LD >= 1, "4 waves, at least one chunk per thread");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* Ps = reinterpret_cast<__bf16*>(smem);  // [2][NPL][BKP][PR] (16-bit elements)
  __bf16* Qs = Ps + 2 * NPL * BKP * PR;          // [2][NPL][BKP][QR]
  float sp = 1.f, sq = 1.f, inv_p = 1.f, inv_q = 1.f;
  if constexpr (F16) {
    qea_f16_scale(a.pmax[0], sp, inv_p);
    qea_f16_scale(a.qmax[0], sq, inv_q);
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int vid = qea_xcd_swizzle(blockIdx.x, a.tiles * a.splits);
  const int split = vid / a.tiles;
  const int tile = vid - split * a.tiles;
  const int taps = a.KH * a.KW;
  const int c_tile = tile % a.c_tiles;
  const int tap = (tile / a.c_tiles) % taps;
  const int r_tile = tile / (a.c_tiles * taps);
  const int r0 = r_tile * BR, c0 = c_tile * BC;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int m_begin = split * a.chunk;
  const int m_end = min(a.M, m_begin + a.chunk);

  WgGather<P_LD, Q_LD, BKP> g;
  g.init(a, tid, m_begin, m_end, r0, c0, kh, kw);
  auto gather = [&](int mbase) { g.fetch(a, mbase); };
  auto stage = [&](int buf) {
    __bf16* pd = Ps + (size_t)buf * NPL * BKP * PR + g.grow * PR;
    __bf16* qd = Qs + (size_t)buf * NPL * BKP * QR + g.grow * QR;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < P_LD; ++i) {
      const f32x4 pv = ((g.p_ok >> i) & 1) ? g.p_reg[i] : zero;
      const int e0 = (g.gl + i * TPR) * 4;  // first channel of this chunk
      const int eo = SWP ? ((((e0 >> 5) ^ (g.grow & 3)) << 5) | (e0 & 31)) : e0;
      if constexpr (F16) {
        f16x4 h, l;
        qea_split2_f16(pv, sp, h, l);
        *reinterpret_cast<f16x4*>(pd + eo) = h;
        *reinterpret_cast<f16x4*>(pd + BKP * PR + eo) = l;
      } else {
        bf16x4 h, m, l;
        qea_split3(pv, h, m, l);
        *reinterpret_cast<bf16x4*>(pd + eo) = h;
        *reinterpret_cast<bf16x4*>(pd + BKP * PR + eo) = m;
        *reinterpret_cast<bf16x4*>(pd + 2 * BKP * PR + eo) = l;
      }
    }
#pragma unroll
    for (int i = 0; i < Q_LD; ++i) {
      const f32x4 qv = ((g.q_ok >> i) & 1) ? g.q_reg[i] : zero;
      const int e0 = (g.gl + i * TPR) * 4;
      const int eo = SWQ ? ((((e0 >> 5) ^ (g.grow & 3)) << 5) | (e0 & 31)) : e0;
      if constexpr (F16) {
        f16x4 h, l;
        qea_split2_f16(qv, sq, h, l);
        *reinterpret_cast<f16x4*>(qd + eo) = h;
        *reinterpret_cast<f16x4*>(qd + BKP * QR + eo) = l;
      } else {
        bf16x4 h, m, l;
        qea_split3(qv, h, m, l);
        *reinterpret_cast<bf16x4*>(qd + eo) = h;
        *reinterpret_cast<bf16x4*>(qd + BKP * QR + eo) = m;
        *reinterpret_cast<bf16x4*>(qd + 2 * BKP * QR + eo) = l;
      }
    }
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // this lane's place in the transposing reads: 16-lane group g16 -> channels (g16&1)*16.., pixels (g16>>1)*8..;
  // inside the group lane 4q+pp addresses pixel q, channel chunk pp
  const int g16 = lane >> 4, tq = (lane & 15) >> 2, tpp = lane & 3;
  // (pixel & 3) of both 4-pixel blocks of a lane is tq: the chunk swizzle of a read is the same for its two halves
  const int t_row_p = ((g16 >> 1) * 8 + tq) * PR, t_row_q = ((g16 >> 1) * 8 + tq) * QR;
  const int t_col = (g16 & 1) * 16 + tpp * 4;  // inside the 32-channel chunk
  const int fr = lane & 31, fh = lane >> 5;
  const int nstage = (m_end > m_begin) ? (m_end - m_begin + BKP - 1) / BKP : 0;

  if (nstage > 0) {
    gather(m_begin);
    stage(0);
  }
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    const int cur = st & 1;
    if (st + 1 < nstage) gather(m_begin + (st + 1) * BKP);
    const __bf16* ps = Ps + (size_t)cur * NPL * BKP * PR + t_row_p + t_col;
    const __bf16* qs = Qs + (size_t)cur * NPL * BKP * QR + t_row_q + t_col;
    frag_t af[NPL][MI], bf[NPL][NJ];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int ch = (wr * TR) / 32 + i;  // 32-channel chunk of this tile
        af[pl][i] = __builtin_bit_cast(frag_t, tr_frag(ps + pl * BKP * PR + ((SWP ? (ch ^ tq) : ch) << 5), PR));
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int ch = (wc * TCc) / 32 + j;
        bf[pl][j] = __builtin_bit_cast(frag_t, tr_frag(qs + pl * BKP * QR + ((SWQ ? (ch ^ tq) : ch) << 5), QR));
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if constexpr (F16) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
        }
      }
    if (st + 1 < nstage) stage(cur ^ 1);
    __syncthreads();
  }

  float* outp = a.out + (size_t)split * a.slab;
  const int ktot = taps * a.C;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = r0 + wr * TR + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        const int cc = c0 + wc * TCc + j * 32 + fr;
        if (rr < a.R && cc < a.C) {
          const size_t o = (size_t)rr * ktot + tap * a.C + cc;
          float v = F16 ? (acc[i][j][r] * inv_p) * inv_q : acc[i][j][r];
          if (a.accumulate) v += outp[o];
          outp[o] = v;
        }
      }
}

// out[g][i] = sum_{k in group g} ws[k][i]   (float4 elements; group g = blockIdx.y covers `gs` slabs).
// With gridDim.y == 1 and gs >= splits this is the plain final reduction into dW.
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, long long n4, int splits, int gs,
                                     long long slab4, int accumulate) {
  const f32x4* w4 = reinterpret_cast<const f32x4*>(ws);
  f32x4* d4 = reinterpret_cast<f32x4*>(out) + (size_t)blockIdx.y * slab4;
  const int k0 = blockIdx.y * gs;
  const int k1 = min(splits, k0 + gs);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 s = w4[i + (size_t)k0 * slab4];
    for (int k = k0 + 1; k < k1; ++k) s += w4[i + (size_t)k * slab4];
    if (accumulate) s += d4[i];
    d4[i] = s;
  }
}

constexpr int REDUCE_GROUP = 32;

// order-fixed reduction of `splits` slabs into dw; more than 64 slabs go through one level of group partials
// (stored behind the slabs in the workspace) so that no thread walks thousands of dependent loads
void reduce_slabs(float* ws, float* dw, long long n4, int splits, int accumulate, hipStream_t s) {
  int grid = (int)((n4 + 255) / 256);
  if (grid > 2048) grid = 2048;
  if (splits <= 64) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, (const float*)ws, dw, n4, splits, splits, n4, accumulate);
    return;
  }
  const int groups = (splits + REDUCE_GROUP - 1) / REDUCE_GROUP;
  float* ws2 = ws + (size_t)splits * n4 * 4;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid, groups), dim3(256), 0, s, (const float*)ws, ws2, n4, splits, REDUCE_GROUP, n4, 0);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, (const float*)ws2, dw, n4, groups, groups, n4, accumulate);
}

size_t slab_workspace_bytes(size_t slab_floats, int splits) {
  size_t b = (size_t)splits * slab_floats * sizeof(float);
  if (splits > 64) b += (size_t)((splits + REDUCE_GROUP - 1) / REDUCE_GROUP) * slab_floats * sizeof(float);
  return b;
}

int tile_slots(int tile);  // workgroups of this tile's kernel the whole chip holds at once

struct Plan {
  int tile;  // 1/9: 128x128  2: 64x64 (K-split)  3: 32x32 (K-split)  4: 32x64  5: 64x32  7/10: 128x64  8/11: 64x128
  int br, bc;
  int splits, chunk, tiles, r_tiles, c_tiles;
};

Plan make_plan(const qea_wgrad_desc* d) {
  Plan p;
  const int R = d->R, C = d->C;
  int tile = d->tile;
  if (tile == 0) {
    // 16-pixel stages (tiles 9-11): half the LDS of the 32-pixel ones, more resident workgroups, +3-5 % (MI355X);
    // tiles 20-22 are their split-bf16 forms (default unless QEA_MFMA=f32)
    const bool bf3 = qea_split_bf16_enabled();
    if (R >= 128 && C >= 128) tile = bf3 ? 20 : 9;
    else if (R >= 128 && C == 64) tile = bf3 ? 21 : 10;
    else if (R == 64 && C >= 128) tile = bf3 ? 22 : 11;
    else if (R <= 32 && C <= 32) tile = 3;
    else if (R <= 32) tile = 4;
    else if (C <= 32) tile = 5;
    else tile = 2;
  }
  p.tile = tile;
  switch (tile) {
    case 1: p.br = 128; p.bc = 128; break;
    case 2: p.br = 64; p.bc = 64; break;
    case 3: p.br = 32; p.bc = 32; break;
    case 4: p.br = 32; p.bc = 64; break;
    case 7: case 10: case 21: p.br = 128; p.bc = 64; break;
    case 8: case 11: case 22: p.br = 64; p.bc = 128; break;
    case 9: case 20: p.br = 128; p.bc = 128; break;
    default: p.br = 64; p.bc = 32; break;
  }
  p.r_tiles = qea_cdiv(R, p.br);
  p.c_tiles = qea_cdiv(C, p.bc);
  p.tiles = p.r_tiles * p.c_tiles * d->KH * d->KW;
  const long long M = (long long)d->B * d->PH * d->PW;
  int splits = d->splits;
  if (splits <= 0) {
    // fill the chip a whole number of times: the workgroups are equally long, so a grid just past a multiple of
    // the resident-workgroup count leaves a nearly empty last round (2052 workgroups on 512 slots: 20 % idle)
    const int slots = tile_slots(tile);
    const int rounds = (2048 + slots / 2) / slots > 0 ? (2048 + slots / 2) / slots : 1;
    splits = rounds * slots / p.tiles;
    const long long max_by_m = (M + 4 * BKP_MAX - 1) / (4 * BKP_MAX);  // at least 4 stages per split
    if (splits > max_by_m) splits = (int)max_by_m;
    if (splits < 1) splits = 1;
    if (splits > 4096) splits = 4096;
  }
  long long chunk = (M + splits - 1) / splits;
  chunk = (chunk + BKP_MAX - 1) / BKP_MAX * BKP_MAX;
  p.chunk = (int)chunk;
  p.splits = (int)((M + chunk - 1) / chunk);
  if (p.splits < 1) p.splits = 1;
  return p;
}

template <int BR, int BC, int WR, int WC, int WK, int BKP>
constexpr size_t lds_bytes() {
  const size_t stage = (size_t)2 * BKP * (BR + BC) * sizeof(float);
  const size_t red = (WK > 1) ? (size_t)WK * BR * BC * sizeof(float) : 0;
  return red > stage ? red : stage;
}

template <int BR, int BC, int WR, int WC, int WK, int BKP = 32>
void launch(const WgArgs& a, hipStream_t s) {
  constexpr size_t lds = lds_bytes<BR, BC, WR, WC, WK, BKP>();
  auto kern = wgrad_kernel<BR, BC, WR, WC, WK, BKP>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)a.tiles * (unsigned)a.splits), dim3(256), lds, s, a);
}

template <int BR, int BC, int WR, int WC, int WK, int BKP = 32>
int slots_of() {
  static int slots = 0;
  if (slots == 0) {
    int per_cu = 0, dev = 0, cus = 0;
    constexpr size_t lds = lds_bytes<BR, BC, WR, WC, WK, BKP>();
    auto kern = wgrad_kernel<BR, BC, WR, WC, WK, BKP>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
      cus = 256;
    (void)hipGetLastError();
    slots = per_cu * cus;
  }
  return slots;
}

template <int BR, int BC, int NPL = 3>
constexpr size_t lds_bytes_bf3() {
  return (size_t)2 * NPL * 16 * ((BR >= 128 ? BR : BR + 32) + (BC >= 128 ? BC : BC + 32)) * 2;
}

template <int BR, int BC, int WR, int WC, int NPL>
void launch_bf3_(const WgArgs& a, hipStream_t s) {
  constexpr size_t lds = lds_bytes_bf3<BR, BC, NPL>();
  auto kern = wgrad_bf3_kernel<BR, BC, WR, WC, NPL>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)a.tiles * (unsigned)a.splits), dim3(256), lds, s, a);
}

template <int BR, int BC, int WR, int WC>
void launch_bf3(const WgArgs& a, hipStream_t s) {
  if (a.pmax && a.qmax) launch_bf3_<BR, BC, WR, WC, 2>(a, s);   // both abs-max values given: the two-way fp16 split
  else launch_bf3_<BR, BC, WR, WC, 3>(a, s);
}

template <int BR, int BC, int WR, int WC>
int slots_of_bf3() {
  static int slots = 0;
  if (slots == 0) {
    int per_cu = 0, dev = 0, cus = 0;
    constexpr size_t lds = lds_bytes_bf3<BR, BC>();
    auto kern = wgrad_bf3_kernel<BR, BC, WR, WC>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
      cus = 256;
    (void)hipGetLastError();
    slots = per_cu * cus;
  }
  return slots;
}

int tile_slots(int tile) {
  switch (tile) {
    case 20: return slots_of_bf3<128, 128, 2, 2>();
    case 21: return slots_of_bf3<128, 64, 2, 2>();
    case 22: return slots_of_bf3<64, 128, 2, 2>();
    case 1: return slots_of<128, 128, 2, 2, 1>();
    case 2: return slots_of<64, 64, 1, 1, 4>();
    case 3: return slots_of<32, 32, 1, 1, 4>();
    case 4: return slots_of<32, 64, 1, 1, 4>();
    case 5: return slots_of<64, 32, 1, 1, 4>();
    case 7: return slots_of<128, 64, 2, 2, 1>();
    case 8: return slots_of<64, 128, 2, 2, 1>();
    case 9: return slots_of<128, 128, 2, 2, 1, 16>();
    case 10: return slots_of<128, 64, 2, 2, 1, 16>();
    case 11: return slots_of<64, 128, 2, 2, 1, 16>();
    default: return 512;
  }
}

int validate(const qea_wgrad_desc* d, const char* who) {
  QEA_REQUIRE(d && d->p && d->q && d->dw, "%s: null pointer", who);
  QEA_REQUIRE(d->B > 0 && d->PH > 0 && d->PW > 0 && d->QH > 0 && d->QW > 0 && d->R > 0 && d->C > 0, "%s: bad dims", who);
  QEA_REQUIRE(d->R % 4 == 0 && d->C % 4 == 0, "%s: R=%d and C=%d must be multiples of 4", who, d->R, d->C);
  QEA_REQUIRE(d->ldp % 4 == 0 && d->ldq % 4 == 0 && d->ldp >= d->R && d->ldq >= d->C, "%s: bad ldp/ldq", who);
  QEA_REQUIRE(((uintptr_t)d->p & 15) == 0 && ((uintptr_t)d->q & 15) == 0 && ((uintptr_t)d->dw & 15) == 0, "%s: 16-byte alignment", who);
  QEA_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride_h > 0 && d->stride_w > 0, "%s: bad filter", who);
  QEA_REQUIRE((long long)d->B * d->PH * d->PW < 0x7fffffffLL && (long long)d->B * d->QH * d->QW < 0x7fffffffLL, "%s: pixel count overflows int32", who);
  return QEA_OK;
}


// ---------------------------------------------------------------------------------------------
// Weight gradient of the NARROW 3x3 layers (R, C in {32, 64}: UNet levels 1-2).  The per-tap kernel above
// re-reads P and Q nine times for 2*R*C flops per (R+C)*4 bytes — the PMC pass shows 4.7x the algorithmic
// HBM traffic.  Here a workgroup walks TH x 32 pixel tiles: P tile and the (TH+2) x 34 Q halo go to LDS
// ONCE and every wave accumulates ALL NINE taps (9 accumulator tiles) from it: waves split the (R/32 x
// C/32) channel blocks and, when those are fewer than 4, the tile's rows.  Each (workgroup, row-split)
// writes one partial slab; the order-fixed splitk_reduce pass sums them.
// ---------------------------------------------------------------------------------------------
template <int R, int C, int TH>
__global__ __launch_bounds__(256) void wgrad_halo_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                         float* __restrict__ ws, int B, int H, int W, int ldp, int ldq,
                                                         int n_tiles) {
  constexpr int WR = R / 32, WC = C / 32, WK = 4 / (WR * WC), ROWS = TH / WK;
  constexpr int HW_ = 34, HH = TH + 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ps = smem;                  // [TH*32][R]
  float* Qs = smem + TH * 32 * R;    // [HH*34][C]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / (WR * WC), wrc = wave % (WR * WC), wr = wrc / WC, wc = wrc % WC;
  const int fr = lane & 31, fh = lane >> 5;
  const int tiles_x = W / 32, tiles_y = H / TH;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // tile t+1's P tile and Q halo are fetched into registers while the MFMAs of tile t run (single LDS buffer)
  constexpr int NP = (TH * 32 * (R / 4) + 255) / 256;
  constexpr int NQ = (HH * HW_ * (C / 4) + 255) / 256;
  f32x4 preg[NP], qreg[NQ];
  auto fetch = [&](int tile) {
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x0 = tx * 32, y0 = ty * TH;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int e = tid + 256 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (e < TH * 32 * (R / 4)) {
        const int c4 = e % (R / 4), pix = e / (R / 4);
        v = *reinterpret_cast<const f32x4*>(p + ((size_t)(b * H + y0 + (pix >> 5)) * W + x0 + (pix & 31)) * ldp + c4 * 4);
      }
      preg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int e = tid + 256 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (e < HH * HW_ * (C / 4)) {
        const int c4 = e % (C / 4), hq = e / (C / 4);
        const int iy = y0 + hq / HW_ - 1, ix = x0 + hq % HW_ - 1;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
          v = *reinterpret_cast<const f32x4*>(q + ((size_t)(b * H + iy) * W + ix) * ldq + c4 * 4);
      }
      qreg[i] = v;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int e = tid + 256 * i;
      if (e < TH * 32 * (R / 4)) *reinterpret_cast<f32x4*>(Ps + e * 4) = preg[i];
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int e = tid + 256 * i;
      if (e < HH * HW_ * (C / 4)) *reinterpret_cast<f32x4*>(Qs + e * 4) = qreg[i];
    }
  };

  int tile = blockIdx.x;
  if (tile < n_tiles) fetch(tile);
  for (; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();  // previous tile fully consumed
    stage();
    __syncthreads();
    if (tile + (int)gridDim.x < n_tiles) fetch(tile + gridDim.x);
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      const int py = wk * ROWS + rr;
      const float* pa = Ps + (py * 32 + fh) * R + wr * 32 + fr;
      const float* qb = Qs + (py * HW_ + fh) * C + wc * 32 + fr;
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const float af = pa[(2 * s) * R];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float bf = qb[((kh * HW_) + 2 * s + kw) * C];
            acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[kh * 3 + kw], 0, 0, 0);
          }
      }
    }
  }
  // partial slab of this (workgroup, row-split): [R][9][C]
  float* out = ws + ((size_t)blockIdx.x * WK + wk) * (R * 9 * C);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      out[((size_t)rr * 9 + t) * C + wc * 32 + fr] = acc[t][r];
    }
}

struct HaloPlan {
  bool ok;
  int th, wk, n_tiles, grid;
};

HaloPlan halo_plan(const qea_wgrad_desc* d) {
  HaloPlan h = {false, 0, 0, 0, 0};
  const bool ch = (d->R == 32 || d->R == 64) && (d->C == 32 || d->C == 64);
  if (!ch || d->KH != 3 || d->KW != 3 || d->pad_h != 1 || d->pad_w != 1 || d->stride_h != 1 || d->stride_w != 1 || d->PH != d->QH ||
      d->PW != d->QW || d->PW % 32)
    return h;
  h.th = (d->R == 32 && d->C == 32) ? 8 : (d->R + d->C <= 96) ? 4 : 2;
  if (d->PH % h.th) return h;
  h.wk = 4 / ((d->R / 32) * (d->C / 32));
  const long long nt = (long long)d->B * (d->PH / h.th) * (d->PW / 32);
  if (nt > 0x7fffffffLL) return h;
  h.n_tiles = (int)nt;
  h.grid = h.n_tiles < 768 ? h.n_tiles : 768;
  h.ok = true;
  return h;
}

template <int R, int C, int TH>
void launch_halo(const qea_wgrad_desc* d, const HaloPlan& h, hipStream_t s) {
  constexpr size_t lds = ((size_t)TH * 32 * R + (size_t)(TH + 2) * 34 * C) * sizeof(float);
  auto kern = wgrad_halo_kernel<R, C, TH>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL(kern, dim3(h.grid), dim3(256), lds, s, d->p, d->q, (float*)d->workspace, d->B, d->PH, d->PW, d->ldp, d->ldq, h.n_tiles);
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 weight gradient of the WIDE 3x3 layers (R, C multiples of 64; tile 23) with all nine taps per workgroup.
// wgrad_bf3_kernel treats every tap as its own GEMM: each pixel of dY and X is gathered, split (5.5 VALU operations per
// element) and staged NINE times per channel-tile pair — the PMC pass showed 2.6x the algorithmic traffic and the matrix
// pipe 45 % busy.  Here a workgroup owns a 64 x 64 (dY-channel, X-channel) block for ALL nine taps (each wave a 32 x 32
// sub-block: nine accumulator tiles) and walks 64-pixel tiles (TH rows x SW columns, SW = 16 or 32): the dY tile and the
// (TH+2) x (SW+2) X halo are gathered, split and written to LDS ONCE per tile (the next tile's loads fly under this
// tile's 216 MFMAs per wave) and the nine taps read their X fragments from the halo at nine shifts through
// ds_read_b64_tr_b16.  Per staged element the matrix work is 9 x what the per-tap kernel gets.  Each workgroup writes its
// block into the partial slab of its pixel split; the order-fixed splitk_reduce pass sums the slabs: bit-reproducible.
// LDS rows = pixels x 64 channels (128 B), the two 64-byte chunks of pixel p stored at chunk ^ ((p >> 1) & 1): the four
// pixel rows of a transposing read then fall on four different 64-byte bank ranges.
// ---------------------------------------------------------------------------------------------
struct Halo9Plan {
  bool ok;
  int sw, th, tiles_x, tiles_y, n_tiles, r_blks, c_blks, splits, rb, cb, wk;
  int spec;   // the producer / consumer form (wgrad_halo9_spec_kernel): two-way fp16 split, 64 x 64 blocks, tile 0 / 23
};

constexpr int H9_HP_MAX = 136;   // halo pixels: (2+2) x (32+2) or (4+2) x (16+2) = 108

// RB / CB = channel block of dY / X per workgroup (64, or 32 for the 32-channel level): 2x2 waves of 32x32 sub-blocks for
// 64x64; with fewer sub-blocks the spare waves split the tile's four k-steps (WK = 4 / sub-blocks) and write separate slabs.
// NPL = 3: three bf16 planes, six MFMAs per product.  NPL = 2 (round 3, ABI v6): two fp16 planes of the operands scaled by the
// powers of two qea_f16_scale derives from their abs-max (pmax, qmax), three MFMAs per product (lh, hl, hh), two thirds of the LDS;
// the block is un-scaled when it is written to its slab (exact).  The four registers a tap's fragments no longer need pay for
// reading the X fragments one tap ahead in the 64 x 64 form too.
template <int SW, int RB, int CB, int NPL = 3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void wgrad_halo9_bf3_kernel(const float* __restrict__ p, const float* __restrict__ q, float* __restrict__ ws, int B, int H, int W, int R, int C,
                            int ldp, int ldq, Halo9Plan hp, const float* __restrict__ pmax, const float* __restrict__ qmax) {
  constexpr bool F16 = NPL == 2;
  typedef typename std::conditional<F16, f16x8, bf16x8>::type frag_t;
  constexpr int TH = 64 / SW, HW_ = SW + 2, HH = TH + 2, HP = HH * HW_;
  constexpr int WR = RB / 32, WC = CB / 32, WK = 4 / (WR * WC);
  constexpr int P_PLANE = 64 * RB, Q_PLANE = H9_HP_MAX * CB;    // 16-bit elements per plane
  constexpr int RC4 = RB / 4, CC4 = CB / 4;                     // float4 chunks per pixel
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* Ps = reinterpret_cast<__bf16*>(smem);            // [NPL][64 px][RB ch]
  __bf16* Qs = Ps + NPL * P_PLANE;                          // [NPL][HP_MAX px][CB ch]
  float sp = 1.f, sq = 1.f, inv_p = 1.f, inv_q = 1.f;
  if constexpr (F16) {
    qea_f16_scale(pmax[0], sp, inv_p);
    qea_f16_scale(qmax[0], sq, inv_q);
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / (WR * WC), wrc = wave % (WR * WC);
  const int wr = wrc / WC, wc = wrc % WC;
  // the r_blks x c_blks workgroups of one split walk the SAME pixel tiles: the XCD swizzle gives them consecutive slots of one
  // XCD, so the tiles come out of that XCD's L2 instead of being fetched by all eight (PMC: 2.1 GB -> see DESIGN.md per launch)
  int bid = qea_xcd_swizzle(blockIdx.x, gridDim.x);
  const int c_blk = bid % hp.c_blks;
  bid /= hp.c_blks;
  const int r_blk = bid % hp.r_blks;
  const int split = bid / hp.r_blks;
  const int r0 = r_blk * RB, c0 = c_blk * CB;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // gather: element e = tid + 256 i -> pixel e / (channels / 4), float4 chunk e % (channels / 4)
  constexpr int NP = 64 * RC4 / 256;                        // 4 (RB 64) / 2 (RB 32)
  constexpr int NQ = (HP * CC4 + 255) / 256;
  f32x4 preg[NP], qreg[NQ];
  auto fetch = [&](int tile) {
    const int tx = tile % hp.tiles_x;
    const int ty = (tile / hp.tiles_x) % hp.tiles_y;
    const int b = tile / (hp.tiles_x * hp.tiles_y);
    const int x0 = tx * SW, y0 = ty * TH;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int e = tid + 256 * i;
      const int c4 = e % RC4, pix = e / RC4;
      const int py = pix / SW, px = pix - py * SW;
      preg[i] = *reinterpret_cast<const f32x4*>(p + ((size_t)(b * H + y0 + py) * W + x0 + px) * ldp + r0 + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int e = tid + 256 * i;
      const int c4 = e % CC4, hq = e / CC4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (hq < HP) {
        const int hy = hq / HW_, hx = hq - hy * HW_;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = *reinterpret_cast<const f32x4*>(q + ((size_t)(b * H + iy) * W + ix) * ldq + c0 + c4 * 4);
      }
      qreg[i] = v;
    }
  };
  // element offset of float4 chunk c4 of LDS pixel row pix: 64-channel rows swap their two 64-byte chunks by (pix >> 1) & 1;
  // 32-channel rows are one chunk (four pixel rows of a transposing read already fall on four 64-byte bank ranges)
  auto row_off = [](int pix, int c4, int chw) { return chw == 64 ? pix * 64 + ((((c4 >> 3) ^ (pix >> 1)) & 1) << 5) + (c4 & 7) * 4 : pix * 32 + c4 * 4; };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int e = tid + 256 * i;
      const int c4 = e % RC4, pix = e / RC4;
      const int o = row_off(pix, c4, RB);
      if constexpr (F16) {
        f16x4 h, l;
        qea_split2_f16(preg[i], sp, h, l);
        *reinterpret_cast<f16x4*>(Ps + o) = h;
        *reinterpret_cast<f16x4*>(Ps + P_PLANE + o) = l;
      } else {
        bf16x4 h, m, l;
        qea_split3(preg[i], h, m, l);
        *reinterpret_cast<bf16x4*>(Ps + o) = h;
        *reinterpret_cast<bf16x4*>(Ps + P_PLANE + o) = m;
        *reinterpret_cast<bf16x4*>(Ps + 2 * P_PLANE + o) = l;
      }
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int e = tid + 256 * i;
      const int c4 = e % CC4, hq = e / CC4;
      if (hq < HP) {
        const int o = row_off(hq, c4, CB);
        if constexpr (F16) {
          f16x4 h, l;
          qea_split2_f16(qreg[i], sq, h, l);
          *reinterpret_cast<f16x4*>(Qs + o) = h;
          *reinterpret_cast<f16x4*>(Qs + Q_PLANE + o) = l;
        } else {
          bf16x4 h, m, l;
          qea_split3(qreg[i], h, m, l);
          *reinterpret_cast<bf16x4*>(Qs + o) = h;
          *reinterpret_cast<bf16x4*>(Qs + Q_PLANE + o) = m;
          *reinterpret_cast<bf16x4*>(Qs + 2 * Q_PLANE + o) = l;
        }
      }
    }
  };
  // transposing-read geometry of this lane (see tr_frag): pixel (g16 >> 1) * 8 + tq (+ 4 for the second read) of the
  // 16-pixel k-step, channels (g16 & 1) * 16 + tpp * 4 of the wave's 32-channel chunk
  const int g16 = lane >> 4, tq = (lane & 15) >> 2, tpp = lane & 3;
  const int l_pix = (g16 >> 1) * 8 + tq;
  const int l_ch = (g16 & 1) * 16 + tpp * 4;
  auto frag_p = [&](const __bf16* plane, int pix) -> frag_t {   // pix = LDS pixel row of this lane's first 4-pixel block
    return __builtin_bit_cast(frag_t, RB == 64 ? tr_frag(plane + pix * 64 + (((wr ^ (pix >> 1)) & 1) << 5) + l_ch, 64) : tr_frag(plane + pix * 32 + l_ch, 32));
  };
  auto frag_q = [&](const __bf16* plane, int pix) -> frag_t {   // ((pix + 4) >> 1 has the parity of pix >> 1: both reads share the swap)
    return __builtin_bit_cast(frag_t, CB == 64 ? tr_frag(plane + pix * 64 + (((wc ^ (pix >> 1)) & 1) << 5) + l_ch, 64) : tr_frag(plane + pix * 32 + l_ch, 32));
  };
  // the products of one tap: smallest terms first (the ll-class terms are dropped)
  auto mma = [&](f32x16& a9, const frag_t* af, const frag_t* bf) {
    if constexpr (F16) {
      a9 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], bf[0], a9, 0, 0, 0);
      a9 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[1], a9, 0, 0, 0);
      a9 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[0], a9, 0, 0, 0);
    } else {
      a9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], a9, 0, 0, 0);
      a9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], a9, 0, 0, 0);
      a9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], a9, 0, 0, 0);
      a9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], a9, 0, 0, 0);
      a9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], a9, 0, 0, 0);
      a9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], a9, 0, 0, 0);
    }
  };

  int tile = split;
  if (tile < hp.n_tiles) fetch(tile);
  for (; tile < hp.n_tiles; tile += hp.splits) {
    __syncthreads();                                          // the previous tile's fragments are all read
    stage();
    __syncthreads();
    if (tile + hp.splits < hp.n_tiles) fetch(tile + hp.splits);   // in flight under the MFMAs below
#pragma unroll 1
    for (int ks = wk; ks < 4; ks += WK) {                     // k-step = 16 consecutive pixels of one tile row (not unrolled: 144
                                                              // accumulator + prefetch registers leave no room for hoisted fragments)
      const int py = (ks * 16) / SW, px0 = (ks * 16) % SW;
      frag_t af[NPL];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) af[pl] = frag_p(Ps + pl * P_PLANE, ks * 16 + l_pix);
      if constexpr (RB == 32 || F16) {
        // X fragments one tap ahead of their MFMAs: the transposing reads of tap t + 1 are issued before the MFMAs of tap t (read
        // just in time, every tap started with an exposed LDS round trip).  The 64 x 64 three-plane form has no registers for the
        // second fragment set (it spills inside this loop: 512 x 512 249 -> 214 TFLOP/s; the 32-wide blocks go 152 -> 169 and
        // 172 -> 184); the two-plane form has.
        auto read_q = [&](int t, frag_t* bfr) {
          const int hq = (py + t / 3) * HW_ + px0 + t % 3 + l_pix;
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) bfr[pl] = frag_q(Qs + pl * Q_PLANE, hq);
        };
        frag_t bq2[2][NPL];
        read_q(0, bq2[0]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const frag_t* bf = bq2[t & 1];
          if (t + 1 < 9) read_q(t + 1, bq2[(t + 1) & 1]);
          mma(acc[t], af, bf);
          if (t + 1 < 9) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NPL, 0);          // the LDS reads of tap t + 1 ...
            __builtin_amdgcn_sched_group_barrier(0x008, F16 ? 3 : 6, 0);      // ... ahead of the MFMAs of tap t
          }
        }
      } else {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int hq = (py + kh) * HW_ + px0 + kw + l_pix;
            frag_t bf[NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) bf[pl] = frag_q(Qs + pl * Q_PLANE, hq);
            mma(acc[kh * 3 + kw], af, bf);
          }
      }
    }
  }
  // this wave's 32 x 9 x 32 block of the partial slab of (its split, its k-step share): [R][9][C]
  float* out = ws + ((size_t)split * WK + wk) * R * 9 * C;
  const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = r0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      out[((size_t)rr * 9 + t) * C + c0 + wc * 32 + fr] = F16 ? (acc[t][r] * inv_p) * inv_q : acc[t][r];
    }
}

// ---------------------------------------------------------------------------------------------
// Round 4: the nine-tap kernel as PRODUCER / CONSUMER waves (two-way fp16 split, 64 x 64 channel blocks; tools/micro/wgrad_lab.hip
// measured it: 299 -> 381 TFLOP/s at 256 x 256 channels, 8 x 32 pixels, B = 2048).  In the kernel above every wave gathers, splits,
// stores, waits at two barriers and only then multiplies: staging was 46 % of a tile's MFMA time and the matrix pipe 0.41 busy.  Here a
// workgroup has EIGHT waves: waves 0-3 (one per SIMD) only run MFMAs — each a 32 x 32 sub-block, nine accumulator tiles — and waves
// 4-7 (their SIMD partners) gather, split and write the NEXT 64-pixel tile into the other of two LDS buffers (50 KB each) while the
// consumers read this one: ONE barrier per tile, the split's VALU work beside the MFMAs instead of in front of them.  One workgroup
// per CU (100 KB of LDS), the pixel tiles dealt over 256 / (channel blocks) splits; slabs and their fixed-order reduction as above.
// Same LDS rows (pixels x 64 channels, 64-byte chunks swapped by (pixel >> 1) & 1), same products in the same order per tile as the
// kernel above — the slab layout (one per split) and the number of splits differ, so the two forms agree to fp32 rounding, not bit
// for bit; each is bit-reproducible.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f16x8 tr_pair128(const char* base) {   // this lane's 4-pixel block and the block 4 pixel rows (512 bytes) further
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + 4 * 128));
  return __builtin_shufflevector(__builtin_bit_cast(f16x4, lo), __builtin_bit_cast(f16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}

// out[c] (+)= sum over the nblk partial rows of ws[k][c]: one wave per channel, lanes stride the rows, fixed order
__global__ void bias_finalize_kernel(const double* __restrict__ ws, int nblk, int R, float* __restrict__ out, int accumulate) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (c >= R) return;
  const int lane = threadIdx.x & 63;
  double a = 0;
  for (int k = lane; k < nblk; k += 64) a += ws[(size_t)k * R + c];
  a = qea_wave_sum_d(a);
  if (lane == 0) out[c] = accumulate ? out[c] + (float)a : (float)a;
}

template <int SW>
constexpr size_t halo9_spec_lds() { return (size_t)2 * (2 * 64 * 64 * 2 + 2 * ((64 / SW + 2) * (SW + 2)) * 64 * 2); }

template <int SW>
__global__ __launch_bounds__(512) void wgrad_halo9_spec_kernel(const float* __restrict__ p, const float* __restrict__ q, float* __restrict__ ws, int B,
                                                               int H, int W, int R, int C, int ldp, int ldq, Halo9Plan hp,
                                                               const float* __restrict__ pmax, const float* __restrict__ qmax,
                                                               double* __restrict__ bias_ws) {
  constexpr int TH = 64 / SW, HWD = SW + 2, HH = TH + 2, HP = HH * HWD;
  constexpr int P_PLANE_B = 64 * 64 * 2, Q_PLANE_B = HP * 64 * 2;            // bytes per plane
  constexpr int BUF_B = 2 * P_PLANE_B + 2 * Q_PLANE_B;                        // one buffer: P h, P l, Q h, Q l
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);
  float sp, sq, inv_p, inv_q;
  qea_f16_scale(pmax[0], sp, inv_p);
  qea_f16_scale(qmax[0], sq, inv_q);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = qea_xcd_swizzle(blockIdx.x, gridDim.x);
  const int c_blk = bid % hp.c_blks;
  bid /= hp.c_blks;
  const int r_blk = bid % hp.r_blks;
  const int split = bid / hp.r_blks;
  const int r0 = r_blk * 64, c0 = c_blk * 64;
  const int ntl = (hp.n_tiles - split + hp.splits - 1) / hp.splits;          // tiles of this workgroup: split, split + splits, ...

  if (wave >= 4) {
    // ---------------------------------------------------------------- producers
    const int pt = tid - 256;
    constexpr int NP = 64 * 16 / 256, NQ = (HP * 16 + 255) / 256;
    f32x4 preg[NP], qreg[NQ];
    // bias gradient (round 4): the column sums of dY ride with the staging — every dY element passes through these registers exactly once
    // per channel-block column, so the workgroups of column block 0 add what they stage (fp64, channels pt % 16 * 4 ... + 3 of this thread)
    const bool do_bias = bias_ws != nullptr && c_blk == 0;
    double bsum[4] = {0.0, 0.0, 0.0, 0.0};
    auto fetch = [&](int tile) {
      const int tx = tile % hp.tiles_x;
      const int ty = (tile / hp.tiles_x) % hp.tiles_y;
      const int b = tile / (hp.tiles_x * hp.tiles_y);
      const int x0 = tx * SW, y0 = ty * TH;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int e = pt + 256 * i;
        const int c4 = e % 16, pix = e / 16;
        const int py = pix / SW, px = pix - py * SW;
        preg[i] = *reinterpret_cast<const f32x4*>(p + ((size_t)(b * H + y0 + py) * W + x0 + px) * ldp + r0 + c4 * 4);
      }
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int e = pt + 256 * i;
        const int c4 = e % 16, hq = e / 16;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (hq < HP) {
          const int hy = hq / HWD, hx = hq - hy * HWD;
          const int iy = y0 + hy - 1, ix = x0 + hx - 1;
          if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = *reinterpret_cast<const f32x4*>(q + ((size_t)(b * H + iy) * W + ix) * ldq + c0 + c4 * 4);
        }
        qreg[i] = v;
      }
    };
    auto row_off = [](int pix, int c4) { return pix * 64 + ((((c4 >> 3) ^ (pix >> 1)) & 1) << 5) + (c4 & 7) * 4; };   // 16-bit elements
    auto stage = [&](char* buf) {
      _Float16* Ps = reinterpret_cast<_Float16*>(buf);
      _Float16* Qs = reinterpret_cast<_Float16*>(buf + 2 * P_PLANE_B);
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int e = pt + 256 * i;
        const int o = row_off(e / 16, e % 16);
        if (do_bias) {
#pragma unroll
          for (int k = 0; k < 4; ++k) bsum[k] += (double)preg[i][k];
        }
        f16x4 h, l;
        qea_split2_f16(preg[i], sp, h, l);
        *reinterpret_cast<f16x4*>(Ps + o) = h;
        *reinterpret_cast<f16x4*>(Ps + P_PLANE_B / 2 + o) = l;
      }
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int e = pt + 256 * i;
        const int hq = e / 16;
        if (hq < HP) {
          const int o = row_off(hq, e % 16);
          f16x4 h, l;
          qea_split2_f16(qreg[i], sq, h, l);
          *reinterpret_cast<f16x4*>(Qs + o) = h;
          *reinterpret_cast<f16x4*>(Qs + Q_PLANE_B / 2 + o) = l;
        }
      }
    };
    if (ntl > 0) {
      fetch(split);
      stage(lds);
      if (ntl > 1) fetch(split + hp.splits);
    }
    __syncthreads();                                         // tile 0 staged
    for (int t = 0; t < ntl; ++t) {
      if (t + 1 < ntl) stage(lds + ((t + 1) & 1) * BUF_B);
      if (t + 2 < ntl) fetch(split + (t + 2) * hp.splits);
      __syncthreads();                                       // consumers done with buffer t & 1, buffer (t + 1) & 1 complete
    }
    if (do_bias) {
      // lanes l, l + 16, l + 32, l + 48 of a wave hold the same four channels (pixel rows 4 apart): wave sums, then one partial row per
      // (split, producer wave): [splits * 4][R] doubles, summed in a fixed order by bias_finalize_kernel (fp64 to the end, as qea_colsum:
      // in front of a batch-statistics BatchNorm this gradient is an exact zero made of cancelling terms)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double v = bsum[k];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        bsum[k] = v;
      }
      if (lane < 16) {
        double* dst = bias_ws + (size_t)(split * 4 + (wave - 4)) * R + r0 + lane * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = bsum[k];
      }
    }
    return;
  }

  // ------------------------------------------------------------------ consumers
  const int wr = wave >> 1, wc = wave & 1;
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // transposing-read geometry: pixel (g16 >> 1) * 8 + tq (+ 4 for the second read) of a 16-pixel k-step, channels (g16 & 1) * 16 + tpp * 4
  const int g16 = lane >> 4, tq = (lane & 15) >> 2, tpp = lane & 3;
  const int l_pix = (g16 >> 1) * 8 + tq;
  const int l_ch = (g16 & 1) * 16 + tpp * 4;
  // P fragment of k-step ks: LDS pixel row ks * 16 + l_pix (ks * 16 is a multiple of 4: the chunk swap only depends on l_pix)
  const int p_off = l_pix * 128 + ((((wr ^ (l_pix >> 1)) & 1) << 5) + l_ch) * 2;
  // Q fragment at halo pixel c + l_pix (c a compile-time tap offset): the chunk swap follows the parity of (c + l_pix) >> 1 -> four
  // loop-invariant bases by (c & 1, (c >> 1) & 1) and an immediate c * 128
  int TQ[2][2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int hs = (par ? (l_pix + 1) >> 1 : l_pix >> 1) + b;
      TQ[par][b] = l_pix * 128 + ((((wc ^ hs) & 1) << 5) + l_ch) * 2;
    }
  __syncthreads();                                           // tile 0 staged
  for (int t = 0; t < ntl; ++t) {
    const char* buf = lds + (t & 1) * BUF_B;
    const char* Pb = buf + p_off;
    const char* Qb = buf + 2 * P_PLANE_B;
    auto read_p = [&](int ks, f16x8* af) {
      af[0] = tr_pair128(Pb + ks * 16 * 128);
      af[1] = tr_pair128(Pb + P_PLANE_B + ks * 16 * 128);
    };
    auto read_q = [&](int f, f16x8* bf) {                   // f = ks * 9 + tap
      const int ks = f / 9, tap = f % 9;
      const int py = (ks * 16) / SW, px0 = (ks * 16) % SW;
      const int c = (py + tap / 3) * HWD + px0 + tap % 3;
      const char* src = Qb + TQ[c & 1][(c >> 1) & 1] + c * 128;
      bf[0] = tr_pair128(src);
      bf[1] = tr_pair128(src + Q_PLANE_B);
    };
    f16x8 af[2][2], bq[2][2];
    read_p(0, af[0]);
    read_q(0, bq[0]);
#pragma unroll
    for (int f = 0; f < 36; ++f) {
      const int ks = f / 9, tap = f % 9;
      const f16x8* a = af[ks & 1];
      const f16x8* b = bq[f & 1];
      if (f + 1 < 36) read_q(f + 1, bq[(f + 1) & 1]);
      if (tap == 0 && ks + 1 < 4) read_p(ks + 1, af[(ks + 1) & 1]);
      // smallest terms first (ll is dropped): lh, hl, hh
      acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc[tap], 0, 0, 0);
      acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc[tap], 0, 0, 0);
      acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc[tap], 0, 0, 0);
      if (f + 1 < 36) {
        if (tap == 0 && ks + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        else __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      }
    }
    __syncthreads();
  }
  float* out = ws + (size_t)split * R * 9 * C;
  const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = r0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      out[((size_t)rr * 9 + t) * C + c0 + wc * 32 + fr] = (acc[t][r] * inv_p) * inv_q;
    }
}

// ---------------------------------------------------------------------------------------------
// The producer / consumer form for 32-pixel-wide tiles with the X halo in a RING (round 4, tools/micro/wgrad_lab.hip: 380 -> 399 TFLOP/s at
// 256 x 256 channels and 8 x 32 pixels, 297 -> 335 at 64 x 64 and 16 x 64, 397 -> 404 at 512 x 512 and 4 x 32).  A 2-row tile has a 4-row
// halo: walking the tiles of one image column top to bottom, two of the four rows are the previous tile's.  A workgroup therefore takes
// whole column strips (contiguous ranges of (image, column tile) pairs) and keeps the halo in a ring of 8 pixel rows: a tile reads ring
// rows g .. g + 3, the producers meanwhile write the next tile's NEW rows — g + 4, g + 5 inside a strip, g + 4 .. g + 7 for the first tile
// of the next strip — so they gather and split 68 instead of 136 halo pixels per tile.  g is even and a ring row holds an even number of
// pixels, so the chunk-swap parity of a tap only needs the row's index inside the halo: a tap's address is one of 16 per-tile registers
// (4 rows x the 4 table entries) + an immediate.  Everything else as wgrad_halo9_spec_kernel; the tile -> split assignment differs (strips,
// not a stride), so the two agree to fp32 rounding and each is bit-reproducible.
// ---------------------------------------------------------------------------------------------
constexpr int H9R_ROWS = 8, H9R_HWD = 34;
constexpr size_t halo9_ring_lds() { return (size_t)2 * (2 * 64 * 64 * 2) + (size_t)2 * H9R_ROWS * H9R_HWD * 64 * 2; }

__global__ __launch_bounds__(512) void wgrad_halo9_ring_kernel(const float* __restrict__ p, const float* __restrict__ q, float* __restrict__ ws, int B, int H,
                                                               int W, int R, int C, int ldp, int ldq, Halo9Plan hp, const float* __restrict__ pmax,
                                                               const float* __restrict__ qmax, double* __restrict__ bias_ws) {
  constexpr int SW = 32, TH = 2, HWD = H9R_HWD;
  constexpr int P_PLANE_B = 64 * 64 * 2, QR_PLANE_B = H9R_ROWS * HWD * 64 * 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const Pbase = reinterpret_cast<char*>(smem);                          // [2 buffers][h, l][64 px][64 ch]
  char* const Qbase = Pbase + 2 * (2 * P_PLANE_B);                            // [h, l][8 ring rows][34][64 ch]
  float sp, sq, inv_p, inv_q;
  qea_f16_scale(pmax[0], sp, inv_p);
  qea_f16_scale(qmax[0], sq, inv_q);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = qea_xcd_swizzle(blockIdx.x, gridDim.x);
  const int c_blk = bid % hp.c_blks;
  bid /= hp.c_blks;
  const int r_blk = bid % hp.r_blks;
  const int split = bid / hp.r_blks;
  const int r0 = r_blk * 64, c0 = c_blk * 64;
  const int n_strips = B * hp.tiles_x;                                        // (hp.splits <= n_strips)
  const int s0 = (int)((long long)n_strips * split / hp.splits), s1 = (int)((long long)n_strips * (split + 1) / hp.splits);
  const int ntl = (s1 - s0) * hp.tiles_y;                                     // this workgroup's tiles, in (strip, tile row) order
  auto ring_of = [&](int t) { return ((t + t / hp.tiles_y) * 2) & 7; };      // + 2 per tile inside a strip, + 4 across strips

  if (wave >= 4) {
    // ---------------------------------------------------------------- producers
    const int pt = tid - 256;
    constexpr int NP = 64 * 16 / 256, NQ = (4 * HWD * 16 + 255) / 256;        // 4, 9 (a whole four-row halo at a strip start)
    f32x4 preg[NP], qreg[NQ];
    int q_rows = 0, q_ring0 = 0;                                              // what qreg holds: halo rows and their first ring row
    const bool do_bias = bias_ws != nullptr && c_blk == 0;
    double bsum[4] = {0.0, 0.0, 0.0, 0.0};
    auto fetch = [&](int t) {
      const int st = s0 + t / hp.tiles_y, ty = t % hp.tiles_y;
      const int b = st / hp.tiles_x, x0 = (st % hp.tiles_x) * SW, y0 = ty * TH;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int e = pt + 256 * i;
        const int c4 = e % 16, pix = e / 16;
        const int py = pix / SW, px = pix - py * SW;
        preg[i] = *reinterpret_cast<const f32x4*>(p + ((size_t)(b * H + y0 + py) * W + x0 + px) * ldp + r0 + c4 * 4);
      }
      const int hy0 = ty == 0 ? 0 : 2;                                        // all four halo rows at a strip start, else the two new ones
      q_rows = 4 - hy0;
      q_ring0 = (ring_of(t) + hy0) & 7;
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int e = pt + 256 * i;
        const int c4 = e % 16, hq = e / 16;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (hq < q_rows * HWD) {
          const int hy = hy0 + hq / HWD, hx = hq % HWD;
          const int iy = y0 + hy - 1, ix = x0 + hx - 1;
          if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = *reinterpret_cast<const f32x4*>(q + ((size_t)(b * H + iy) * W + ix) * ldq + c0 + c4 * 4);
        }
        qreg[i] = v;
      }
    };
    auto row_off = [](int pix, int c4) { return pix * 64 + ((((c4 >> 3) ^ (pix >> 1)) & 1) << 5) + (c4 & 7) * 4; };   // 16-bit elements
    auto stage = [&](int pbuf) {
      _Float16* Ps = reinterpret_cast<_Float16*>(Pbase + pbuf * (2 * P_PLANE_B));
      _Float16* Qs = reinterpret_cast<_Float16*>(Qbase);
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int e = pt + 256 * i;
        const int o = row_off(e / 16, e % 16);
        if (do_bias) {
#pragma unroll
          for (int k = 0; k < 4; ++k) bsum[k] += (double)preg[i][k];
        }
        f16x4 h, l;
        qea_split2_f16(preg[i], sp, h, l);
        *reinterpret_cast<f16x4*>(Ps + o) = h;
        *reinterpret_cast<f16x4*>(Ps + P_PLANE_B / 2 + o) = l;
      }
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int e = pt + 256 * i;
        const int hq = e / 16;
        if (hq < q_rows * HWD) {
          const int rr = (q_ring0 + hq / HWD) & 7;
          const int o = row_off(rr * HWD + hq % HWD, e % 16);
          f16x4 h, l;
          qea_split2_f16(qreg[i], sq, h, l);
          *reinterpret_cast<f16x4*>(Qs + o) = h;
          *reinterpret_cast<f16x4*>(Qs + QR_PLANE_B / 2 + o) = l;
        }
      }
    };
    if (ntl > 0) {
      fetch(0);
      stage(0);
      if (ntl > 1) fetch(1);
    }
    __syncthreads();                                         // tile 0 staged
    for (int t = 0; t < ntl; ++t) {
      if (t + 1 < ntl) stage((t + 1) & 1);
      if (t + 2 < ntl) fetch(t + 2);
      __syncthreads();                                       // consumers done with tile t, tile t + 1 complete
    }
    if (do_bias) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double v = bsum[k];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        bsum[k] = v;
      }
      if (lane < 16) {
        double* dst = bias_ws + (size_t)(split * 4 + (wave - 4)) * R + r0 + lane * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = bsum[k];
      }
    }
    return;
  }

  // ------------------------------------------------------------------ consumers
  const int wr = wave >> 1, wc = wave & 1;
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int g16 = lane >> 4, tq = (lane & 15) >> 2, tpp = lane & 3;
  const int l_pix = (g16 >> 1) * 8 + tq;
  const int l_ch = (g16 & 1) * 16 + tpp * 4;
  const int p_off = l_pix * 128 + ((((wr ^ (l_pix >> 1)) & 1) << 5) + l_ch) * 2;
  int TQ[2][2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int hs = (par ? (l_pix + 1) >> 1 : l_pix >> 1) + b;
      TQ[par][b] = l_pix * 128 + ((((wc ^ hs) & 1) << 5) + l_ch) * 2;
    }
  __syncthreads();                                           // tile 0 staged
  for (int t = 0; t < ntl; ++t) {
    const char* Pb = Pbase + (t & 1) * (2 * P_PLANE_B) + p_off;
    const int g = ring_of(t);
    int QB[4][2][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rowoff = ((g + k) & 7) * (HWD * 128);
#pragma unroll
      for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) QB[k][par][bb] = TQ[par][bb] + rowoff;
    }
    auto read_p = [&](int ks, f16x8* af) {
      af[0] = tr_pair128(Pb + ks * 16 * 128);
      af[1] = tr_pair128(Pb + P_PLANE_B + ks * 16 * 128);
    };
    auto read_q = [&](int f, f16x8* bf) {                   // f = ks * 9 + tap
      const int ks = f / 9, tap = f % 9;
      const int py = (ks * 16) / SW, px0 = (ks * 16) % SW;
      const int k = py + tap / 3, col = px0 + tap % 3;
      const char* src = Qbase + QB[k][col & 1][((col >> 1) + k) & 1] + col * 128;
      bf[0] = tr_pair128(src);
      bf[1] = tr_pair128(src + QR_PLANE_B);
    };
    f16x8 af[2][2], bq[2][2];
    read_p(0, af[0]);
    read_q(0, bq[0]);
#pragma unroll
    for (int f = 0; f < 36; ++f) {
      const int ks = f / 9, tap = f % 9;
      const f16x8* a = af[ks & 1];
      const f16x8* b = bq[f & 1];
      if (f + 1 < 36) read_q(f + 1, bq[(f + 1) & 1]);
      if (tap == 0 && ks + 1 < 4) read_p(ks + 1, af[(ks + 1) & 1]);
      acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc[tap], 0, 0, 0);
      acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc[tap], 0, 0, 0);
      acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc[tap], 0, 0, 0);
      if (f + 1 < 36) {
        if (tap == 0 && ks + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        else __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      }
    }
    __syncthreads();
  }
  float* out = ws + (size_t)split * R * 9 * C;
  const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = r0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      out[((size_t)rr * 9 + t) * C + c0 + wc * 32 + fr] = (acc[t][r] * inv_p) * inv_q;
    }
}

Halo9Plan halo9_plan(const qea_wgrad_desc* d) {
  Halo9Plan h = {false, 0, 0, 0, 0, 0, 0, 0, 0, 64, 64, 1, 0};
  if (d->KH != 3 || d->KW != 3 || d->pad_h != 1 || d->pad_w != 1 || d->stride_h != 1 || d->stride_w != 1 || d->PH != d->QH || d->PW != d->QW) return h;
  if (d->R % 32 || d->C % 32) return h;
  h.sw = (d->PW % 32 == 0) ? 32 : (d->PW == 16 ? 16 : 0);
  if (!h.sw) return h;
  h.th = 64 / h.sw;
  if (d->PH % h.th) return h;
  h.tiles_x = d->PW / h.sw;
  h.tiles_y = d->PH / h.th;
  const long long nt = (long long)d->B * h.tiles_x * h.tiles_y;
  if (nt > 0x7fffffffLL) return h;
  h.n_tiles = (int)nt;
  // dY / X channel block per workgroup: 64 x 64, or 32 x 32 as soon as one side is not a multiple of 64 (the mixed 32 x 64
  // shapes ran at 128-134 TFLOP/s — their 9 + 2 prefetch registers per lane spill in the MFMA loop — against 172 for two
  // 32 x 32 blocks of the same layer)
  h.rb = h.cb = (d->R % 64 || d->C % 64) ? 32 : 64;
  h.wk = 4 / ((h.rb / 32) * (h.cb / 32));                    // k-step shares (separate slabs)
  h.r_blks = d->R / h.rb;
  h.c_blks = d->C / h.cb;
  // tile 29 keeps the round-3 form (every wave stages, then multiplies) where the producer / consumer form would run
  h.spec = (h.rb == 64 && d->p_absmax && d->q_absmax && d->tile != 29) ? 1 : 0;
  // two workgroups per CU (77 KB of LDS each): about 512 workgroups, each walking at least 8 tiles; the producer / consumer form:
  // one 8-wave workgroup per CU (100 KB)
  int splits = d->splits > 0 ? d->splits : (h.spec ? 256 : 512) / (h.r_blks * h.c_blks);
  if (splits > h.n_tiles / 8) splits = h.n_tiles / 8;
  if (h.spec && h.sw == 32 && splits > d->B * h.tiles_x) splits = d->B * h.tiles_x;   // the ring form deals whole column strips
  if (splits < 1) splits = 1;
  h.splits = splits;
  h.ok = true;
  return h;
}

// the 32- and 64-channel layers are taken by both halo kernels: the split-bf16 nine-tap form wins unless QEA_MFMA=f32
bool prefer_halo9(const qea_wgrad_desc* d) { return qea_split_bf16_enabled() && halo9_plan(d).ok; }

template <int SW, int RB, int CB, int NPL = 3>
int launch_halo9_(const qea_wgrad_desc* d, const Halo9Plan& h, hipStream_t s) {
  constexpr size_t lds = (size_t)NPL * (64 * RB + H9_HP_MAX * CB) * 2;
  const long long grid = (long long)h.r_blks * h.c_blks * h.splits;
  auto kern = wgrad_halo9_bf3_kernel<SW, RB, CB, NPL>;
  static int attr_rc = (int)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr_rc != (int)hipSuccess) {
    qea_set_error("qea_conv_wgrad: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString((hipError_t)attr_rc));
    return QEA_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, d->p, d->q, (float*)d->workspace, d->B, d->PH, d->PW, d->R, d->C, d->ldp, d->ldq, h,
                     d->p_absmax, d->q_absmax);
  return QEA_OK;
}

template <int SW, int RB, int CB>
int launch_halo9_any(const qea_wgrad_desc* d, const Halo9Plan& h, hipStream_t s) {
  // both abs-max pointers given (ABI v6): the two-way fp16 split
  return (d->p_absmax && d->q_absmax) ? launch_halo9_<SW, RB, CB, 2>(d, h, s) : launch_halo9_<SW, RB, CB, 3>(d, h, s);
}

template <int SW>
int launch_halo9_spec(const qea_wgrad_desc* d, const Halo9Plan& h, hipStream_t s, double* bias_ws) {
  constexpr size_t lds = halo9_spec_lds<SW>();
  const long long grid = (long long)h.r_blks * h.c_blks * h.splits;
  auto kern = wgrad_halo9_spec_kernel<SW>;
  static int attr_rc = (int)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr_rc != (int)hipSuccess) {
    qea_set_error("qea_conv_wgrad: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString((hipError_t)attr_rc));
    return QEA_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, s, d->p, d->q, (float*)d->workspace, d->B, d->PH, d->PW, d->R, d->C, d->ldp, d->ldq, h,
                     d->p_absmax, d->q_absmax, bias_ws);
  return QEA_OK;
}

int launch_halo9_ring(const qea_wgrad_desc* d, const Halo9Plan& h, hipStream_t s, double* bias_ws) {
  constexpr size_t lds = halo9_ring_lds();
  const long long grid = (long long)h.r_blks * h.c_blks * h.splits;
  static int attr_rc = (int)hipFuncSetAttribute((const void*)wgrad_halo9_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr_rc != (int)hipSuccess) {
    qea_set_error("qea_conv_wgrad: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString((hipError_t)attr_rc));
    return QEA_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(wgrad_halo9_ring_kernel, dim3((unsigned)grid), dim3(512), lds, s, d->p, d->q, (float*)d->workspace, d->B, d->PH, d->PW, d->R, d->C,
                     d->ldp, d->ldq, h, d->p_absmax, d->q_absmax, bias_ws);
  return QEA_OK;
}

int launch_halo9(const qea_wgrad_desc* d, const Halo9Plan& h, hipStream_t s, double* bias_ws = nullptr) {
  if (h.spec) return h.sw == 32 ? launch_halo9_ring(d, h, s, bias_ws) : launch_halo9_spec<16>(d, h, s, bias_ws);
  if (h.sw == 32) return h.rb == 64 ? launch_halo9_any<32, 64, 64>(d, h, s) : launch_halo9_any<32, 32, 32>(d, h, s);
  return h.rb == 64 ? launch_halo9_any<16, 64, 64>(d, h, s) : launch_halo9_any<16, 32, 32>(d, h, s);
}

}  // namespace

extern "C" int qea_conv_wgrad_fuses_bias(const qea_wgrad_desc* d) {
  if (!d || d->R <= 0 || d->C <= 0 || d->B <= 0) return 0;
  if (!((d->tile == 0 && qea_split_bf16_enabled()) || d->tile == 23)) return 0;
  const Halo9Plan h9 = halo9_plan(d);
  return (h9.ok && h9.spec) ? 1 : 0;
}

extern "C" size_t qea_conv_wgrad_workspace_bytes(const qea_wgrad_desc* d) {
  if (!d || d->R <= 0 || d->C <= 0 || d->B <= 0) return 0;
  if ((d->tile == 0 && !prefer_halo9(d)) || d->tile == 6) {
    const HaloPlan h = halo_plan(d);
    if (h.ok) return slab_workspace_bytes((size_t)d->R * 9 * d->C, h.grid * h.wk);
  }
  if ((d->tile == 0 && qea_split_bf16_enabled()) || d->tile == 23 || d->tile == 29) {
    const Halo9Plan h9 = halo9_plan(d);
    if (h9.ok) return slab_workspace_bytes((size_t)d->R * 9 * d->C, h9.splits * h9.wk) + ((h9.spec && d->dbias) ? (size_t)h9.splits * 4 * d->R * sizeof(double) : 0);
  }
  const Plan p = make_plan(d);
  if (p.splits <= 1) return 0;
  return slab_workspace_bytes((size_t)d->R * d->KH * d->KW * d->C, p.splits);
}

extern "C" int qea_conv_wgrad(const qea_wgrad_desc* d, void* stream) {
  int rc = validate(d, "qea_conv_wgrad");
  if (rc != QEA_OK) return rc;
  if ((d->tile == 0 && !prefer_halo9(d)) || d->tile == 6) {
    const HaloPlan h = halo_plan(d);
    if (h.ok) {
      const size_t slab = (size_t)d->R * 9 * d->C;
      const size_t need_h = slab_workspace_bytes(slab, h.grid * h.wk);
      QEA_REQUIRE(d->workspace && d->workspace_bytes >= need_h && ((uintptr_t)d->workspace & 15) == 0,
                  "qea_conv_wgrad: workspace of %zu bytes required, %zu given", need_h, (size_t)d->workspace_bytes);
      hipStream_t hs = (hipStream_t)stream;
      qea_prof_begin(QEA_PROF_CONV_WGRAD, hs);
      if (d->R == 32 && d->C == 32) launch_halo<32, 32, 8>(d, h, hs);
      else if (d->R == 32 && d->C == 64) launch_halo<32, 64, 4>(d, h, hs);
      else if (d->R == 64 && d->C == 32) launch_halo<64, 32, 4>(d, h, hs);
      else launch_halo<64, 64, 2>(d, h, hs);
      reduce_slabs((float*)d->workspace, d->dw, (long long)slab / 4, h.grid * h.wk, d->accumulate, hs);
      // algorithmic bytes: dY once + X once + dW once
      qea_prof_end(QEA_PROF_CONV_WGRAD, hs, 2.0 * d->B * d->PH * (double)d->PW * (double)slab,
                   4.0 * ((double)d->B * d->PH * d->PW * d->R + (double)d->B * d->QH * d->QW * d->C + (double)slab));
      QEA_CHECK_LAUNCH();
      return QEA_OK;
    }
    QEA_REQUIRE(d->tile == 0, "qea_conv_wgrad: tile 6 (LDS-halo) needs a 3x3 pad-1 stride-1 conv with R,C in {32,64}, PW %% 32 == 0");
  }
  if ((d->tile == 0 && qea_split_bf16_enabled()) || d->tile == 23 || d->tile == 29) {
    const Halo9Plan h9 = halo9_plan(d);
    if (h9.ok) {
      const size_t slab = (size_t)d->R * 9 * d->C;
      const size_t need_dw = slab_workspace_bytes(slab, h9.splits * h9.wk);
      const bool fused_bias = h9.spec && d->dbias;
      const size_t need9 = need_dw + (fused_bias ? (size_t)h9.splits * 4 * d->R * sizeof(double) : 0);
      QEA_REQUIRE(d->workspace && d->workspace_bytes >= need9 && ((uintptr_t)d->workspace & 15) == 0,
                  "qea_conv_wgrad: workspace of %zu bytes required, %zu given", need9, (size_t)d->workspace_bytes);
      QEA_REQUIRE(!d->dbias || h9.spec, "qea_conv_wgrad: dbias is taken by the producer / consumer nine-tap form only (ask qea_conv_wgrad_fuses_bias first)");
      hipStream_t hs = (hipStream_t)stream;
      qea_prof_begin(QEA_PROF_CONV_WGRAD, hs);
      double* bias_ws = fused_bias ? reinterpret_cast<double*>(reinterpret_cast<char*>(d->workspace) + need_dw) : nullptr;
      rc = launch_halo9(d, h9, hs, bias_ws);
      if (rc != QEA_OK) {
        qea_prof_abort(QEA_PROF_CONV_WGRAD);
        return rc;
      }
      reduce_slabs((float*)d->workspace, d->dw, (long long)slab / 4, h9.splits * h9.wk, d->accumulate, hs);
      if (fused_bias) hipLaunchKernelGGL(bias_finalize_kernel, dim3(qea_cdiv(d->R, 4)), dim3(256), 0, hs, (const double*)bias_ws, h9.splits * 4, d->R, d->dbias, d->accumulate);
      qea_prof_end(QEA_PROF_CONV_WGRAD, hs, 2.0 * d->B * d->PH * (double)d->PW * (double)slab,
                   4.0 * ((double)d->B * d->PH * d->PW * d->R + (double)d->B * d->QH * d->QW * d->C + (double)slab),
                   (d->p_absmax && d->q_absmax) ? 2 : 1);
      QEA_CHECK_LAUNCH();
      return QEA_OK;
    }
    QEA_REQUIRE(d->tile == 0, "qea_conv_wgrad: tile 23 (nine-tap split-bf16) needs a 3x3 pad-1 stride-1 conv, R,C multiples of 64, PW in {16, 32k}");
  }
  QEA_REQUIRE(!d->dbias, "qea_conv_wgrad: dbias is taken by the producer / consumer nine-tap form only (ask qea_conv_wgrad_fuses_bias first)");
  const Plan p = make_plan(d);
  const size_t need = (p.splits > 1) ? slab_workspace_bytes((size_t)d->R * d->KH * d->KW * d->C, p.splits) : 0;
  QEA_REQUIRE(need == 0 || (d->workspace && d->workspace_bytes >= need && ((uintptr_t)d->workspace & 15) == 0),
              "qea_conv_wgrad: workspace of %zu bytes required, %zu given", need, (size_t)d->workspace_bytes);

  WgArgs a;
  a.p = d->p; a.q = d->q;
  a.B = d->B; a.PH = d->PH; a.PW = d->PW; a.QH = d->QH; a.QW = d->QW; a.R = d->R; a.C = d->C;
  a.KH = d->KH; a.KW = d->KW; a.pad_h = d->pad_h; a.pad_w = d->pad_w; a.stride_h = d->stride_h; a.stride_w = d->stride_w;
  a.ldp = d->ldp; a.ldq = d->ldq;
  a.M = d->B * d->PH * d->PW;
  a.chunk = p.chunk; a.splits = p.splits;
  a.r_tiles = p.r_tiles; a.c_tiles = p.c_tiles; a.tiles = p.tiles;
  a.slab = (long long)d->R * d->KH * d->KW * d->C;
  a.out = (p.splits > 1) ? (float*)d->workspace : d->dw;
  a.accumulate = (p.splits > 1) ? 0 : d->accumulate;
  a.pmax = d->p_absmax;
  a.qmax = d->q_absmax;

  hipStream_t s = (hipStream_t)stream;
  qea_prof_begin(QEA_PROF_CONV_WGRAD, s);
  switch (p.tile) {
    case 1: launch<128, 128, 2, 2, 1>(a, s); break;
    case 2: launch<64, 64, 1, 1, 4>(a, s); break;
    case 3: launch<32, 32, 1, 1, 4>(a, s); break;
    case 4: launch<32, 64, 1, 1, 4>(a, s); break;
    case 5: launch<64, 32, 1, 1, 4>(a, s); break;
    case 7: launch<128, 64, 2, 2, 1>(a, s); break;
    case 8: launch<64, 128, 2, 2, 1>(a, s); break;
    case 9: launch<128, 128, 2, 2, 1, 16>(a, s); break;   // 16-pixel stages: half the LDS per workgroup
    case 10: launch<128, 64, 2, 2, 1, 16>(a, s); break;
    case 11: launch<64, 128, 2, 2, 1, 16>(a, s); break;
    case 20: launch_bf3<128, 128, 2, 2>(a, s); break;      // split-bf16 forms
    case 21: launch_bf3<128, 64, 2, 2>(a, s); break;
    case 22: launch_bf3<64, 128, 2, 2>(a, s); break;
    default:
      qea_prof_abort(QEA_PROF_CONV_WGRAD);
      qea_set_error("qea_conv_wgrad: unknown tile %d", p.tile);
      return QEA_ERR_INVALID;
  }
  if (p.splits > 1) reduce_slabs((float*)d->workspace, d->dw, a.slab / 4, p.splits, d->accumulate, s);
  qea_prof_end(QEA_PROF_CONV_WGRAD, s, 2.0 * a.M * (double)a.slab,
               4.0 * ((double)a.M * d->R + (double)d->B * d->QH * d->QW * d->C + (double)a.slab),
               p.tile >= 20 ? ((a.pmax && a.qmax) ? 2 : 1) : 0);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}
