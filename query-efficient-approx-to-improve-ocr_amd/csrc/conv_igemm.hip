// Implicit-GEMM convolution / GEMM on the gfx950 fp32 matrix cores.
//
//   C[m][n] = sum_k A[m][k] * Wt[n][k]      m = (b,oh,ow) output pixel, n = output channel,
//                                            k = (kh,kw,c) filter tap x input channel.
// A is never materialised: every K-slice (16 or 32 channels of one filter tap; channel slice
// outermost, taps innermost) is gathered straight from the NHWC input (zero outside the image)
// into an LDS tile, the filter slice into a second one, and each of the 4 or 8 waves runs
// v_mfma_f32_32x32x2_f32 over its 64x64 (or 64x32) accumulator block.  K is consumed in a permuted order inside every group of 8 so that a
// lane fetches its 4 A (B) values of four consecutive MFMAs with one ds_read_b128:
// lane (r = l&31, h = l>>5), MFMA step s  ->  k = 4h + s  on both operands.
//
// Roofline: MFMA-bound (fp32 matrix peak 157.3 TFLOP/s); 2*M*N*K algorithmic flops/launch.
#include "conv_args.h"
#include <type_traits>
#include <stdlib.h>
#include <string.h>

using qea_conv::ConvArgs;
using qea_conv::conv_epilogue;

namespace {

// Per-thread operand fetch shared by the fp32 and the split-bf16 kernel: thread (lrow, kc) gathers float4 number kc of
// the K slice for A_LD rows of the im2col matrix (zero outside the image / past M) and B_LD rows of the filter.
// K order: channel slice outermost, the KH*KW taps innermost.  The taps of one slice re-read (shifted) the same few KB
// of input, which then come from L1/L2; tap-outermost order re-streamed the whole input tile per tap and the PMC pass
// showed 5x the algorithmic HBM fetch on the 512-channel layers.
template <int A_LD, int B_LD, int BK, int RPP>
struct IgemmGather {
  int lrow, kc, ntaps;
  int a_pix[A_LD];             // b*H*W, or -1 when the row is past M
  int a_ih0[A_LD], a_iw0[A_LD];
  const float* b_ptr[B_LD > 0 ? B_LD : 1];    // filter row of slot j (+ kc*4), or null past N
  f32x4 a_reg[A_LD], b_reg[B_LD > 0 ? B_LD : 1];

  __device__ __forceinline__ void init(const ConvArgs& p, int tid, int m0, int n0) {
    constexpr int KCH = BK / 4;
    lrow = tid / KCH;
    kc = tid % KCH;
    ntaps = p.KH * p.KW;
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int m = m0 + lrow + RPP * i;
      if (m < p.M) {
        const int b = m / ohw;
        const int rem = m - b * ohw;
        const int oh = rem / p.OW;
        const int ow = rem - oh * p.OW;
        a_pix[i] = b * p.H * p.W;
        a_ih0[i] = oh * p.stride_h - p.pad_h;
        a_iw0[i] = ow * p.stride_w - p.pad_w;
      } else {
        a_pix[i] = -1;
        a_ih0[i] = 0;
        a_iw0[i] = 0;
      }
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const int n = n0 + lrow + RPP * j;
      b_ptr[j] = (n < p.N) ? (p.w + (size_t)n * p.K + kc * 4) : nullptr;
    }
  }

  __device__ __forceinline__ void fetch(const ConvArgs& p, int kt) {
    const int cs = kt / ntaps;
    const int tap = kt - cs * ntaps;
    const int c0 = cs * BK;
    const int kh = tap / p.KW;
    const int kw = tap - kh * p.KW;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
      const bool ok = (a_pix[i] >= 0) && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(p.x + (size_t)(a_pix[i] + ih * p.W + iw) * p.ldx + c0 + kc * 4);
      a_reg[i] = v;
    }
    const int koff = tap * p.Cin + c0;
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (b_ptr[j]) v = *reinterpret_cast<const f32x4*>(b_ptr[j] + koff);
      b_reg[j] = v;
    }
  }
};

// Cin must be a multiple of 32 (checked by the entry point); the K-slice BK is 16 or 32

template <int BM, int BN, int WGM, int WGN, int BK>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_igemm_kernel(const ConvArgs p) {
  constexpr int NT = WGM * WGN * 64;           // threads per workgroup (4 or 8 waves)
  constexpr int LDS_LD = BK + 4;               // padded LDS row (floats): keeps b128 reads conflict-free
  constexpr int KCH = BK / 4;                  // float4 per row of a K-slice
  constexpr int RPP = NT / KCH;                // rows covered per pass of the workgroup's threads
  constexpr int TM = BM / WGM, TN = BN / WGN;  // wave tile
  constexpr int MI = TM / 32, NJ = TN / 32;    // 32x32 accumulator tiles per wave
  constexpr int A_LD = BM / RPP;               // float4 gathers per thread per stage (A)
  constexpr int B_LD = BN / RPP;               // (B)
  static_assert(WGM * WGN == 4 || WGM * WGN == 8, "4 or 8 waves per workgroup");
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the rows staged per pass");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                    // [2][BM][LDS_LD]
  float* Bs = smem + 2 * BM * LDS_LD;  // [2][BN][LDS_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;

  const int nwg = p.m_tiles * p.n_tiles;
  const int tile = qea_xcd_swizzle(blockIdx.x, nwg);
  const int tile_m = tile / p.n_tiles;
  const int tile_n = tile % p.n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  IgemmGather<A_LD, B_LD, BK, RPP> g;
  g.init(p, tid, m0, n0);
  const int lrow = g.lrow, kc = g.kc;
  const int KT = p.KH * p.KW * (p.Cin / BK);
  auto gather = [&](int kt) { g.fetch(p, kt); };
  auto stage = [&](int buf) {
    float* a_dst = As + buf * BM * LDS_LD;
    float* b_dst = Bs + buf * BN * LDS_LD;
#pragma unroll
    for (int i = 0; i < A_LD; ++i)
      *reinterpret_cast<f32x4*>(a_dst + (lrow + RPP * i) * LDS_LD + kc * 4) = g.a_reg[i];
#pragma unroll
    for (int j = 0; j < B_LD; ++j)
      *reinterpret_cast<f32x4*>(b_dst + (lrow + RPP * j) * LDS_LD + kc * 4) = g.b_reg[j];
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31;  // fragment row (A: m, B: n)
  const int fh = lane >> 5;  // K half inside a group of 8

  gather(0);
  stage(0);
  __syncthreads();

  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) gather(kt + 1);  // global loads in flight under the MFMAs below

    const float* a_src = As + cur * BM * LDS_LD + (wm * TM + fr) * LDS_LD + fh * 4;
    const float* b_src = Bs + cur * BN * LDS_LD + (wn * TN + fr) * LDS_LD + fh * 4;
    // fragment registers double-buffered over the four K-groups of the slice: the ds_reads of group
    // kb+1 are in flight under the MFMAs of group kb (hipcc does not pipeline them on its own)
    f32x4 af[2][MI], bf[2][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(a_src + i * 32 * LDS_LD);
#pragma unroll
    for (int j = 0; j < NJ; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(b_src + j * 32 * LDS_LD);
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      const int c = kb & 1, n = c ^ 1;
      if (kb + 1 < BK / 8) {
#pragma unroll
        for (int i = 0; i < MI; ++i) af[n][i] = *reinterpret_cast<const f32x4*>(a_src + i * 32 * LDS_LD + (kb + 1) * 8);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[n][j] = *reinterpret_cast<const f32x4*>(b_src + j * 32 * LDS_LD + (kb + 1) * 8);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i][s], bf[c][j][s], acc[i][j], 0, 0, 0);
      // pin the interleave: one LDS fragment read of the next group per (4*MI*NJ / (MI+NJ)) MFMAs of this one
      if (kb + 1 < BK / 8) {
#pragma unroll
        for (int r = 0; r < MI + NJ; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, (4 * MI * NJ) / (MI + NJ), 0);
        }
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * MI * NJ, 0);
      }
    }

    if (kt + 1 < KT) stage(cur ^ 1);
    __syncthreads();
  }

  conv_epilogue<MI, NJ, TM, TN>(p, acc, m0, n0, wm, wn, fr, fh);
}


// ---------------------------------------------------------------------------------------------
// Split-bf16 form (tile ids 20-23, 25; operands split on the fly — the fallback when no pre-split filter planes are given, see
// conv_igemm_bf3w_kernel / conv_igemm_p3_kernel below): the same implicit GEMM with every fp32 operand split
// into three bf16 planes x = h + m + l (|x - (h+m+l)| <= 2^-24 |x|) and each product formed by SIX bf16 MFMAs
// (hh, hm, mh, hl, lh, mm; the dropped ml, lm, ll terms are < 2^-23 |ab|) accumulated in fp32:
// v_mfma_f32_32x32x16_bf16 sustains 1800 TFLOP/s on this part (tools/micro/mfma_rate.hip), i.e. a 300 TFLOP/s
// fp32-equivalent ceiling against 154 for v_mfma_f32_32x32x2_f32.  One K stage = 16 channels of one tap = one MFMA
// k-step; LDS holds [plane][row][16 bf16] (32-byte rows, halves of rows 8..15 mod 16 swapped: conflict-free ds_read_b128).
// ---------------------------------------------------------------------------------------------
// (the 8-wave tiles must stay within 128 VGPRs: two workgroups, i.e. four waves per SIMD, share a CU)
template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(WGM * WGN * 64) __attribute__((amdgpu_waves_per_eu(WGM * WGN == 8 ? 4 : 1)))
void conv_igemm_bf3_kernel(const ConvArgs p) {
  constexpr int BK = 16;
  constexpr int NT = WGM * WGN * 64;
  constexpr int ROWB = 16;                     // bf16 elements per LDS row: 32 bytes, unpadded; the two 16-byte halves of rows
                                               // 8..15 (mod 16) are swapped, which keeps ds_read_b128 conflict-free (see frag)
  constexpr int KCH = BK / 4;                  // float4 per row of a K-slice
  constexpr int RPP = NT / KCH;                // rows covered per pass of the workgroup's threads
  constexpr int TM = BM / WGM, TN = BN / WGN;  // wave tile
  constexpr int MI = TM / 32, NJ = TN / 32;
  constexpr int A_LD = BM / RPP, B_LD = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the rows staged per pass");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* As = reinterpret_cast<__bf16*>(smem);  // [2][3][BM][ROWB]
  __bf16* Bs = As + 2 * 3 * BM * ROWB;           // [2][3][BN][ROWB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int nwg = p.m_tiles * p.n_tiles;
  const int tile = qea_xcd_swizzle(blockIdx.x, nwg);
  const int tile_m = tile / p.n_tiles, tile_n = tile % p.n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  IgemmGather<A_LD, B_LD, BK, RPP> g;
  g.init(p, tid, m0, n0);
  const int lrow = g.lrow, kc = g.kc;
  const int KT = p.KH * p.KW * (p.Cin / BK);
  auto gather = [&](int kt) { g.fetch(p, kt); };
  auto stage = [&](int buf) {
    __bf16* a_dst = As + (size_t)buf * 3 * BM * ROWB;
    __bf16* b_dst = Bs + (size_t)buf * 3 * BN * ROWB;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      bf16x4 h, m, l;
      qea_split3(g.a_reg[i], h, m, l);
      const int ro = lrow + RPP * i;
      const int o = ro * ROWB + (((kc >> 1) ^ ((ro >> 3) & 1)) << 3) + (kc & 1) * 4;
      *reinterpret_cast<bf16x4*>(a_dst + o) = h;
      *reinterpret_cast<bf16x4*>(a_dst + BM * ROWB + o) = m;
      *reinterpret_cast<bf16x4*>(a_dst + 2 * BM * ROWB + o) = l;
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      bf16x4 h, m, l;
      qea_split3(g.b_reg[j], h, m, l);
      const int ro = lrow + RPP * j;
      const int o = ro * ROWB + (((kc >> 1) ^ ((ro >> 3) & 1)) << 3) + (kc & 1) * 4;
      *reinterpret_cast<bf16x4*>(b_dst + o) = h;
      *reinterpret_cast<bf16x4*>(b_dst + BN * ROWB + o) = m;
      *reinterpret_cast<bf16x4*>(b_dst + 2 * BN * ROWB + o) = l;
    }
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  gather(0);
  stage(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) gather(kt + 1);
    // lane (fr, fh) reads the 8 k-values of half fh of row fr; rows 8..15 (mod 16) keep their halves swapped
    const int hsw = (fh ^ ((fr >> 3) & 1)) * 8;   // wm*TM, wn*TN and i*32 are multiples of 16: the swap depends on fr only
    const __bf16* a_src = As + (size_t)cur * 3 * BM * ROWB + (wm * TM + fr) * ROWB + hsw;
    const __bf16* b_src = Bs + (size_t)cur * 3 * BN * ROWB + (wn * TN + fr) * ROWB + hsw;
    bf16x8 af[3][MI], bf[3][NJ];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int i = 0; i < MI; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(a_src + pl * BM * ROWB + i * 32 * ROWB);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(b_src + pl * BN * ROWB + j * 32 * ROWB);
    }
    // smallest terms first (ll-class terms are dropped): lh, hl, mm, mh, hm, hh
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
      }
    if (kt + 1 < KT) stage(cur ^ 1);
    __syncthreads();
  }

  conv_epilogue<MI, NJ, TM, TN>(p, acc, m0, n0, wm, wn, fr, fh);
}

// ---------------------------------------------------------------------------------------------
// The split-bf16 implicit GEMM on PRE-SPLIT operands ("P3" format, written once per tensor by split_planes_kernel):
//   planes[row][C/16][3][16]  bf16 — for every 16-channel slice of a pixel (or of a filter row) the h, m, l planes side by
//   side, 96 contiguous bytes.
// conv_igemm_bf3_kernel splits every gathered fp32 element again for each of the 9 taps and each N-tile that uses it
// (~5.5 VALU operations per element and stage: the PMC pass showed the loop bound by instruction issue, matrix pipe 47 %
// busy).  Here a K stage is pure data movement: every wave issues its share of 1 KiB LDS-DMA pieces (global_load_lds_dwordx4:
// 32 rows x 32 bytes of ONE plane per wave-instruction, per-lane source address = the im2col gather; lanes of rows outside
// the image / past M or N read a zero chunk kept behind the planes), no VGPR staging, no conversions, no ds_write.
// The LDS image, the fragment reads and the MFMA sequence are those of conv_igemm_bf3_kernel, so the results are
// bit-identical to it (tests/test_conv_igemm_gpu.py::test_presplit_planes_bit_identical).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void qea_lds_void;
typedef __attribute__((address_space(1))) const void qea_glob_void;

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(WGM * WGN * 64) __attribute__((amdgpu_waves_per_eu(WGM * WGN == 8 ? 4 : 1)))
void conv_igemm_p3_kernel(const ConvArgs p) {
  constexpr int NW = WGM * WGN;
  constexpr int ROWB = 16;                     // bf16 per LDS row (32 bytes); halves of rows 8..15 (mod 16) swapped, as in bf3
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MI = TM / 32, NJ = TN / 32;
  constexpr int NA = 3 * BM / 32, NB = 3 * BN / 32;            // 1 KiB DMA pieces per stage (A, B)
  constexpr int IA = (NA + NW - 1) / NW, IB = (NB + NW - 1) / NW;   // per wave

  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* As = reinterpret_cast<__bf16*>(smem);  // [2][3][BM][ROWB]
  __bf16* Bs = As + 2 * 3 * BM * ROWB;           // [2][3][BN][ROWB]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int nwg = p.m_tiles * p.n_tiles;
  const int tile = qea_xcd_swizzle(blockIdx.x, nwg);
  const int tile_m = tile / p.n_tiles, tile_n = tile % p.n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int ntaps = p.KH * p.KW;
  const int cslices = p.Cin >> 4;
  const int KT = ntaps * cslices;

  // ---- per-lane DMA state: lane (row = l>>1, half = l&1) of piece j fills LDS row rb*32+row, 16-byte half `half`
  const int l_row = lane >> 1, l_half = lane & 1;
  int a_off[IA];
  unsigned a_mask[IA];
  int b_off[IB];
  {
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const int j = wave + NW * i;
      a_off[i] = 0;
      a_mask[i] = 0;
      if (j < NA) {
        const int rb = j / 3, plane = j - rb * 3;
        const int row = rb * 32 + l_row;
        const int hsrc = l_half ^ ((row >> 3) & 1);
        const int m = m0 + row;
        if (m < p.M) {
          const int b = m / ohw;
          const int rem = m - b * ohw;
          const int oh = rem / p.OW;
          const int ow = rem - oh * p.OW;
          const int ih0 = oh * p.stride_h - p.pad_h, iw0 = ow * p.stride_w - p.pad_w;
          a_off[i] = ((b * p.H + ih0) * p.W + iw0) * cslices * 96 + plane * 32 + hsrc * 16;
          unsigned mk = 0;
          for (int kh = 0; kh < p.KH; ++kh)
            for (int kw = 0; kw < p.KW; ++kw)
              if ((unsigned)(ih0 + kh) < (unsigned)p.H && (unsigned)(iw0 + kw) < (unsigned)p.W) mk |= 1u << (kh * p.KW + kw);
          a_mask[i] = mk;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int j = wave + NW * i;
      b_off[i] = -1;
      if (j < NB) {
        const int rb = j / 3, plane = j - rb * 3;
        const int row = rb * 32 + l_row;
        const int hsrc = l_half ^ ((row >> 3) & 1);
        const int n = n0 + row;
        if (n < p.N) b_off[i] = n * (p.K >> 4) * 96 + plane * 32 + hsrc * 16;
      }
    }
  }
  char* const lds_a = reinterpret_cast<char*>(As);
  char* const lds_b = reinterpret_cast<char*>(Bs);
  auto dma = [&](int kt, int buf) {
    const int cs = kt / ntaps;
    const int tap = kt - cs * ntaps;
    const int kh = tap / p.KW;
    const int kw = tap - kh * p.KW;
    const int dA = ((kh * p.W + kw) * cslices + cs) * 96;
    const int dB = (tap * cslices + cs) * 96;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const int j = wave + NW * i;
      if (j < NA) {
        const int rb = j / 3, plane = j - rb * 3;
        const unsigned off = ((a_mask[i] >> tap) & 1u) ? (unsigned)(a_off[i] + dA) : p.xp_zero;
        char* dst = lds_a + ((buf * 3 + plane) * BM + rb * 32) * (ROWB * 2);
        __builtin_amdgcn_global_load_lds((qea_glob_void*)(p.xp + off), (qea_lds_void*)dst, 16, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int j = wave + NW * i;
      if (j < NB) {
        const int rb = j / 3, plane = j - rb * 3;
        const unsigned off = (b_off[i] >= 0) ? (unsigned)(b_off[i] + dB) : p.wp_zero;
        char* dst = lds_b + ((buf * 3 + plane) * BN + rb * 32) * (ROWB * 2);
        __builtin_amdgcn_global_load_lds((qea_glob_void*)(p.wp + off), (qea_lds_void*)dst, 16, 0, 0);
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

  const int fr = lane & 31, fh = lane >> 5;
  const int hsw = (fh ^ ((fr >> 3) & 1)) * 8;
  dma(0, 0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) dma(kt + 1, cur ^ 1);      // lands under the MFMAs below; __syncthreads() waits for it (vmcnt)
    const __bf16* a_src = As + (size_t)cur * 3 * BM * ROWB + (wm * TM + fr) * ROWB + hsw;
    const __bf16* b_src = Bs + (size_t)cur * 3 * BN * ROWB + (wn * TN + fr) * ROWB + hsw;
    bf16x8 af[3][MI], bf[3][NJ];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int i = 0; i < MI; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(a_src + pl * BM * ROWB + i * 32 * ROWB);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(b_src + pl * BN * ROWB + j * 32 * ROWB);
    }
    // smallest terms first (ll-class terms are dropped): lh, hl, mm, mh, hm, hh — the order of conv_igemm_bf3_kernel
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
      }
    // hipcc otherwise hoists the barrier (and the vmcnt(0) in front of it) above the register-only MFMAs, which exposes
    // the whole DMA latency of the next stage instead of hiding it under this stage's matrix work
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
  conv_epilogue<MI, NJ, TM, TN>(p, acc, m0, n0, wm, wn, fr, fh);
}

// fp32 rows [M][ld] (C channels used, C % 16 == 0) -> P3 planes + a 128-byte zero tail.  One thread per 8 channels.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, int ld, long long M, int C, __bf16* __restrict__ planes) {
  const int c8n = C >> 3;
  const long long total = M * c8n;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e < 8) {                                    // zero tail (128 bytes = 64 bf16) behind the planes
    bf16x8 z;
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = (__bf16)0.f;
    *reinterpret_cast<bf16x8*>(planes + M * C * 3 + e * 8) = z;
  }
  if (e >= total) return;
  const long long row = e / c8n;
  const int c8 = (int)(e - row * c8n);
  const float* src = x + row * ld + c8 * 8;
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(src);
  const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + 4);
  bf16x4 h0, m0, l0, h1, m1, l1;
  qea_split3(v0, h0, m0, l0);
  qea_split3(v1, h1, m1, l1);
  bf16x8 h, m, l;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    h[k] = h0[k]; h[k + 4] = h1[k];
    m[k] = m0[k]; m[k + 4] = m1[k];
    l[k] = l0[k]; l[k + 4] = l1[k];
  }
  __bf16* dst = planes + ((row * (C >> 4) + (c8 >> 1)) * 3) * 16 + (c8 & 1) * 8;
  *reinterpret_cast<bf16x8*>(dst) = h;
  *reinterpret_cast<bf16x8*>(dst + 16) = m;
  *reinterpret_cast<bf16x8*>(dst + 32) = l;
}

// the two-plane fp16 row format of the same tensor: [row][C/16][plane h, l][16 fp16] of x * s (s from xmax, qea_f16_scale), 128 zero
// bytes, then one float = 1 / s
__global__ void split_planes_f16_kernel(const float* __restrict__ x, int ld, long long M, int C, const float* __restrict__ xmax,
                                        _Float16* __restrict__ planes) {
  const int c8n = C >> 3;
  const long long total = M * c8n;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float sx, inv;
  qea_f16_scale(xmax[0], sx, inv);
  if (e < 32) reinterpret_cast<float*>(planes + M * C * 2)[e] = 0.f;      // the 128-byte zero tail
  if (e == 0) reinterpret_cast<float*>(planes + M * C * 2)[32] = inv;
  if (e >= total) return;
  const long long row = e / c8n;
  const int c8 = (int)(e - row * c8n);
  const float* src = x + row * ld + c8 * 8;
  f16x4 h0, l0, h1, l1;
  qea_split2_f16(*reinterpret_cast<const f32x4*>(src), sx, h0, l0);
  qea_split2_f16(*reinterpret_cast<const f32x4*>(src + 4), sx, h1, l1);
  f16x8 h, l;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    h[k] = h0[k]; h[k + 4] = h1[k];
    l[k] = l0[k]; l[k + 4] = l1[k];
  }
  _Float16* dst = planes + ((row * (C >> 4) + (c8 >> 1)) * 2) * 16 + (c8 & 1) * 8;
  *reinterpret_cast<f16x8*>(dst) = h;
  *reinterpret_cast<f16x8*>(dst + 16) = l;
}

// ---------------------------------------------------------------------------------------------
// Hybrid: activations split on the fly (conv_igemm_bf3_kernel's gather), FILTER from pre-split planes by LDS-DMA
// (conv_igemm_p3_kernel's path).  Pre-splitting an ACTIVATION costs a 10-byte-per-element HBM pass for a tensor that one
// launch consumes (measured at B = 2048: the passes cost the 8 ms the all-DMA kernel gains), but the filter's planes are
// made once per optimiser step (once per run for the frozen CRNN in Phase B) and shared by every M-tile: the filter share
// of the staging work — a third of it on the 256x128 tile, two thirds on 128x256 — leaves the VALU for free.
// Same LDS image and MFMA sequence as the other two kernels: bit-identical results.
// ---------------------------------------------------------------------------------------------
// NPL = 2 (round 3, ABI v6): the two-way fp16 split — activations scaled by the power of two of their abs-max (p.xmax) and split
// h + l on the fly, filter planes in the two-plane row format of qea_split_planes_f16 (64 B per row and K chunk, its inverse scale
// behind the zero tail), three MFMAs per product, accumulators un-scaled before the epilogue.
template <int BM, int BN, int WGM, int WGN, bool STATS, int NPL = 3>
__global__ __launch_bounds__(WGM * WGN * 64) __attribute__((amdgpu_waves_per_eu(WGM * WGN == 8 ? 4 : 1)))
void conv_igemm_bf3w_kernel(const ConvArgs p) {
  constexpr bool F16 = NPL == 2;
  typedef typename std::conditional<F16, f16x8, bf16x8>::type frag_t;
  constexpr int BK = 16;
  constexpr int NW = WGM * WGN;
  constexpr int NT = NW * 64;
  constexpr int ROWB = 16;
  constexpr int KCH = BK / 4;
  constexpr int RPP = NT / KCH;
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MI = TM / 32, NJ = TN / 32;
  constexpr int A_LD = BM / RPP;
  constexpr int NB = NPL * BN / 32;
  constexpr int IB = (NB + NW - 1) / NW;
  static_assert(BM % RPP == 0, "tile rows must be a multiple of the rows staged per pass");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* As = reinterpret_cast<__bf16*>(smem);  // [2][NPL][BM][ROWB] (16-bit elements)
  __bf16* Bs = As + 2 * NPL * BM * ROWB;         // [2][NPL][BN][ROWB]
  float sx = 1.f, inv_x = 1.f, inv_w = 1.f;
  if constexpr (F16) {
    qea_f16_scale(p.xmax[0], sx, inv_x);
    inv_w = reinterpret_cast<const float*>(p.wp + p.wp_zero + 128)[0];
  }

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int nwg = p.m_tiles * p.n_tiles;
  const int tile = qea_xcd_swizzle(blockIdx.x, nwg);
  const int tile_m = tile / p.n_tiles, tile_n = tile % p.n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int ntaps = p.KH * p.KW;
  const int cslices = p.Cin >> 4;

  IgemmGather<A_LD, 0, BK, RPP> g;               // activations only
  g.init(p, tid, m0, n0);
  const int lrow = g.lrow, kc = g.kc;
  const int KT = ntaps * cslices;
  const int l_row = lane >> 1, l_half = lane & 1;
  int b_off[IB];
#pragma unroll
  for (int i = 0; i < IB; ++i) {
    const int j = wave + NW * i;
    b_off[i] = -1;
    if (j < NB) {
      const int rb = j / NPL, plane = j - rb * NPL;
      const int row = rb * 32 + l_row;
      const int hsrc = l_half ^ ((row >> 3) & 1);
      const int n = n0 + row;
      if (n < p.N) b_off[i] = n * (p.K >> 4) * (NPL * 32) + plane * 32 + hsrc * 16;
    }
  }
  char* const lds_b = reinterpret_cast<char*>(Bs);
  auto dma_b = [&](int kt, int buf) {
    const int cs = kt / ntaps;
    const int tap = kt - cs * ntaps;
    const int dB = (tap * cslices + cs) * (NPL * 32);
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int j = wave + NW * i;
      if (j < NB) {
        const int rb = j / NPL, plane = j - rb * NPL;
        const unsigned off = (b_off[i] >= 0) ? (unsigned)(b_off[i] + dB) : p.wp_zero;
        char* dst = lds_b + ((buf * NPL + plane) * BN + rb * 32) * (ROWB * 2);
        __builtin_amdgcn_global_load_lds((qea_glob_void*)(p.wp + off), (qea_lds_void*)dst, 16, 0, 0);
      }
    }
  };
  auto stage_a = [&](int buf) {
    __bf16* a_dst = As + (size_t)buf * NPL * BM * ROWB;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int ro = lrow + RPP * i;
      const int o = ro * ROWB + (((kc >> 1) ^ ((ro >> 3) & 1)) << 3) + (kc & 1) * 4;
      if constexpr (F16) {
        f16x4 h, l;
        qea_split2_f16(g.a_reg[i], sx, h, l);
        *reinterpret_cast<f16x4*>(a_dst + o) = h;
        *reinterpret_cast<f16x4*>(a_dst + BM * ROWB + o) = l;
      } else {
        bf16x4 h, m, l;
        qea_split3(g.a_reg[i], h, m, l);
        *reinterpret_cast<bf16x4*>(a_dst + o) = h;
        *reinterpret_cast<bf16x4*>(a_dst + BM * ROWB + o) = m;
        *reinterpret_cast<bf16x4*>(a_dst + 2 * BM * ROWB + o) = l;
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

  const int fr = lane & 31, fh = lane >> 5;
  const int hsw = (fh ^ ((fr >> 3) & 1)) * 8;
  dma_b(0, 0);
  g.fetch(p, 0);
  stage_a(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) {
      dma_b(kt + 1, cur ^ 1);
      g.fetch(p, kt + 1);
    }
    const __bf16* a_src = As + (size_t)cur * NPL * BM * ROWB + (wm * TM + fr) * ROWB + hsw;
    const __bf16* b_src = Bs + (size_t)cur * NPL * BN * ROWB + (wn * TN + fr) * ROWB + hsw;
    frag_t af[NPL][MI], bf[NPL][NJ];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
      for (int i = 0; i < MI; ++i) af[pl][i] = *reinterpret_cast<const frag_t*>(a_src + pl * BM * ROWB + i * 32 * ROWB);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[pl][j] = *reinterpret_cast<const frag_t*>(b_src + pl * BN * ROWB + j * 32 * ROWB);
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
    if (kt + 1 < KT) stage_a(cur ^ 1);
    __syncthreads();
  }
  if constexpr (F16) {                                      // un-scale: exact (powers of two)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] * inv_x) * inv_w;
  }
  conv_epilogue<MI, NJ, TM, TN, STATS>(p, acc, m0, n0, wm, wn, fr, fh, tile_m * WGM + wm);
}

template <int BM, int BN, int WGM, int WGN, bool STATS, int NPL = 3>
int launch_bf3w_(const ConvArgs& a, hipStream_t s) {
  ConvArgs p = a;
  p.m_tiles = qea_cdiv(p.M, BM);
  p.n_tiles = qea_cdiv(p.N, BN);
  const size_t lds = (size_t)2 * NPL * (BM + BN) * 16 * 2;
  auto kern = conv_igemm_bf3w_kernel<BM, BN, WGM, WGN, STATS, NPL>;
  static int attr_rc = (int)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr_rc != (int)hipSuccess) {
    qea_set_error("qea_conv_igemm: cannot reserve %zu bytes of LDS for the %dx%d tile: %s", lds, BM, BN, hipGetErrorString((hipError_t)attr_rc));
    return QEA_ERR_LAUNCH;
  }
  const long long grid = (long long)p.m_tiles * p.n_tiles;
  if (grid <= 0 || grid > 0x7fffffffLL) {
    qea_set_error("qea_conv_igemm: grid %lld out of range", grid);
    return QEA_ERR_INVALID;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WGM * WGN * 64), lds, s, p);
  return QEA_OK;
}

template <int BM, int BN, int WGM, int WGN>
int launch_bf3w(const ConvArgs& a, hipStream_t s) {
  if (a.xmax)                                              // two-way fp16 split: fp16 filter planes + the input's abs-max (ABI v6)
    return a.stats ? launch_bf3w_<BM, BN, WGM, WGN, true, 2>(a, s) : launch_bf3w_<BM, BN, WGM, WGN, false, 2>(a, s);
  return a.stats ? launch_bf3w_<BM, BN, WGM, WGN, true>(a, s) : launch_bf3w_<BM, BN, WGM, WGN, false>(a, s);
}

template <int BM, int BN, int WGM, int WGN>
int launch_p3(const ConvArgs& a, hipStream_t s) {
  ConvArgs p = a;
  p.m_tiles = qea_cdiv(p.M, BM);
  p.n_tiles = qea_cdiv(p.N, BN);
  const size_t lds = (size_t)2 * 3 * (BM + BN) * 16 * 2;
  auto kern = conv_igemm_p3_kernel<BM, BN, WGM, WGN>;
  static int attr_rc = (int)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr_rc != (int)hipSuccess) {
    qea_set_error("qea_conv_igemm: cannot reserve %zu bytes of LDS for the %dx%d tile: %s", lds, BM, BN, hipGetErrorString((hipError_t)attr_rc));
    return QEA_ERR_LAUNCH;
  }
  const long long grid = (long long)p.m_tiles * p.n_tiles;
  if (grid <= 0 || grid > 0x7fffffffLL) {
    qea_set_error("qea_conv_igemm: grid %lld out of range", grid);
    return QEA_ERR_INVALID;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WGM * WGN * 64), lds, s, p);
  return QEA_OK;
}

template <int BM, int BN, int WGM, int WGN>
int launch_bf3(const ConvArgs& a, hipStream_t s) {
  ConvArgs p = a;
  p.m_tiles = qea_cdiv(p.M, BM);
  p.n_tiles = qea_cdiv(p.N, BN);
  const size_t lds = (size_t)2 * 3 * (BM + BN) * 16 * 2;
  auto kern = conv_igemm_bf3_kernel<BM, BN, WGM, WGN>;
  static int attr_rc = (int)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr_rc != (int)hipSuccess) {
    qea_set_error("qea_conv_igemm: cannot reserve %zu bytes of LDS for the %dx%d split-bf16 tile: %s", lds, BM, BN, hipGetErrorString((hipError_t)attr_rc));
    return QEA_ERR_LAUNCH;
  }
  const long long grid = (long long)p.m_tiles * p.n_tiles;
  if (grid <= 0 || grid > 0x7fffffffLL) {
    qea_set_error("qea_conv_igemm: grid %lld out of range", grid);
    return QEA_ERR_INVALID;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WGM * WGN * 64), lds, s, p);
  return QEA_OK;
}

template <int BM, int BN, int WGM, int WGN, int BK>
int launch(const ConvArgs& a, hipStream_t s) {
  ConvArgs p = a;
  p.m_tiles = qea_cdiv(p.M, BM);
  p.n_tiles = qea_cdiv(p.N, BN);
  const size_t lds = (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float);
  auto kern = conv_igemm_kernel<BM, BN, WGM, WGN, BK>;
  static int attr_rc = (int)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr_rc != (int)hipSuccess) {
    qea_set_error("qea_conv_igemm: cannot reserve %zu bytes of LDS for the %dx%d tile: %s", lds, BM, BN, hipGetErrorString((hipError_t)attr_rc));
    return QEA_ERR_LAUNCH;
  }
  const long long grid = (long long)p.m_tiles * p.n_tiles;
  if (grid <= 0 || grid > 0x7fffffffLL) {
    qea_set_error("qea_conv_igemm: grid %lld out of range", grid);
    return QEA_ERR_INVALID;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WGM * WGN * 64), lds, s, p);
  return QEA_OK;
}


// ---------------------------------------------------------------------------------------------
// 3x3 / pad 1 / stride 1 convolution for the NARROW layers (C_in, C_out in {32, 64}: UNet levels 1-2,
// forward and input-gradient).  With N = 32..64 the generic kernel re-gathers every input pixel 9x
// through LDS for only 32..64 output channels (16-32 flop per gathered byte: L2-bound).  Here a
// workgroup owns a TH x 32 pixel tile of one image: the (TH+2) x 34 input halo is loaded ONCE into LDS
// (pixel stride C_in+4 floats: conflict-free ds_read_b128) and all 9 taps read their A fragments from it
// with shifted addresses; filter fragments come straight from global memory (<= 147 KB, L1/L2-resident),
// so after the single barrier the four waves never synchronise again.
// ---------------------------------------------------------------------------------------------
template <int CIN, int COUT, int TH, bool STATS>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ y, int B, int H, int W, int ldx, int ldy,
                                                           const float* __restrict__ scale, const float* __restrict__ bias, int relu,
                                                           double* __restrict__ stats, float* __restrict__ yamax) {
  constexpr int TW = 32, PS = CIN + 4, HW_ = TW + 2, HH = TH + 2;
  constexpr int MI = TH / 4, NJ = COUT / 32, KC = CIN / 32;
  extern __shared__ __attribute__((aligned(16))) float halo[];  // [HH][HW_][PS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = W / TW, tiles_y = H / TH;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;

  // ---- halo load (zero outside the image): every global load of the thread is issued before the first LDS write
  // (a load -> wait -> write loop serialises ~12 memory latencies, as long as the whole MFMA phase of the tile) ----
  constexpr int C4 = CIN / 4;
  constexpr int NLD = (HH * HW_ * C4 + 255) / 256;
  const float* xb = x + (size_t)b * H * W * ldx;
  f32x4 hv[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = tid + 256 * i;
    const int c4 = e % C4;
    const int q = e / C4;
    const int hx = q % HW_, hy = q / HW_;
    const int iy = y0 + hy - 1, ix = x0 + hx - 1;
    const bool ok = e < HH * HW_ * C4 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? xb + ((size_t)iy * W + ix) * ldx + c4 * 4 : x);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    hv[i] = ok ? v : zero;
  }
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = tid + 256 * i;
    if (e < HH * HW_ * C4) *reinterpret_cast<f32x4*>(halo + (e / C4) * PS + (e % C4) * 4) = hv[i];
  }
  __syncthreads();

  const int fr = lane & 31, fh = lane >> 5;
  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // filter rows of this lane: w[n][tap][c], n = j*32 + fr
  const float* wl = w + (size_t)fr * 9 * CIN + fh * 4;
  constexpr int KSTEPS = 9 * KC;
  f32x4 bq[2][4][NJ];  // [buffer][kb][j]
  auto load_b = [&](int ks, int buf) {
    const int tap = ks / KC, c0 = (ks % KC) * 32;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        bq[buf][kb][j] = *reinterpret_cast<const f32x4*>(wl + (size_t)j * 32 * 9 * CIN + tap * CIN + c0 + kb * 8);
  };
  load_b(0, 0);
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const int cur = ks & 1;
    if (ks + 1 < KSTEPS) load_b(ks + 1, cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the next step's filter loads AHEAD of this step's MFMAs (hipcc sinks them to just-in-time)
    const int tap = ks / KC, c0 = (ks % KC) * 32;
    const int kh = tap / 3, kw = tap % 3;
    const float* a_src = halo + ((wave * MI + kh) * HW_ + fr + kw) * PS + c0 + fh * 4;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      f32x4 af[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const f32x4*>(a_src + i * HW_ * PS + kb * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bq[cur][kb][j][s], acc[i][j], 0, 0, 0);
    }
  }

  // ---- store: row (= pixel x) = (r&3) + 8*(r>>2) + 4*fh, col (= channel) = fr; optional per-channel
  // scale/bias (+ReLU) epilogue = eval-mode BatchNorm folded into the conv ----
  float esc[NJ], ebi[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    esc[j] = scale ? scale[j * 32 + fr] : 1.f;
    ebi[j] = bias ? bias[j * 32 + fr] : 0.f;
  }
  double st0[NJ], st1[NJ];
  if (STATS) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) st0[j] = st1[j] = 0.0;
  }
  float am = 0.f;                                // abs-max of what this workgroup stores (qea_conv_desc.y_absmax, as every other tile)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    float* yrow = y + ((size_t)(b * H + y0 + wave * MI + i) * W + x0) * ldy;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int px = (r & 3) + 8 * (r >> 2) + 4 * fh;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float v = acc[i][j][r];
        if (scale && bias) v = __fmaf_rn(v, esc[j], ebi[j]);
        else if (scale) v *= esc[j];
        else if (bias) v += ebi[j];
        if (relu) v = fmaxf(v, 0.f);
        yrow[(size_t)px * ldy + j * 32 + fr] = v;
        am = qea_amax_acc(am, v);
        if (STATS) {
          st0[j] += (double)v;
          st1[j] += (double)v * (double)v;
        }
      }
    }
  }
  qea_amax_commit_block(am, yamax);
  if (STATS) {                                   // one partial per (workgroup, wave): [blocks][COUT][2] fp64 column sums
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const double a = st0[j] + __shfl_xor(st0[j], 32, 64);
      const double c = st1[j] + __shfl_xor(st1[j], 32, 64);
      if (fh == 0) {
        double* dst = stats + ((size_t)(blockIdx.x * 4 + wave) * COUT + j * 32 + fr) * 2;
        dst[0] = a;
        dst[1] = c;
      }
    }
  }
}

template <int CIN, int COUT, int TH, bool STATS>
int launch_halo_(const ConvArgs& a, hipStream_t s) {
  constexpr size_t lds = (size_t)(TH + 2) * 34 * (CIN + 4) * sizeof(float);
  auto kern = conv3x3_halo_kernel<CIN, COUT, TH, STATS>;
  static int attr_rc = (int)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr_rc != (int)hipSuccess) {
    qea_set_error("qea_conv_igemm(halo): cannot reserve %zu bytes of LDS: %s", (size_t)lds, hipGetErrorString((hipError_t)attr_rc));
    return QEA_ERR_LAUNCH;
  }
  const long long grid = (long long)a.B * (a.H / TH) * (a.W / 32);
  if (grid <= 0 || grid > 0x7fffffffLL) {
    qea_set_error("qea_conv_igemm(halo): grid %lld out of range", grid);
    return QEA_ERR_INVALID;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, a.x, a.w, a.y, a.B, a.H, a.W, a.ldx, a.ldy, a.scale, a.bias, a.relu, a.stats, a.yamax);
  return QEA_OK;
}

template <int CIN, int COUT, int TH>
int launch_halo(const ConvArgs& a, hipStream_t s) {
  return a.stats ? launch_halo_<CIN, COUT, TH, true>(a, s) : launch_halo_<CIN, COUT, TH, false>(a, s);
}

// ---------------------------------------------------------------------------------------------
// Split-bf16 form of conv3x3_halo_kernel for the NARROW layers (C_in, C_out in {32, 64}; tile 24).  Seventeen per cent of
// a B = 2048 step still ran on the fp32 MFMA (105-120 TFLOP/s) or on the 256x64 generic split tile (150), which gathers
// and splits every input element once per tap.  Here the (TH+2) x 34 input halo of a TH x 32 pixel tile is gathered ONCE,
// split ONCE into the three bf16 planes and kept in LDS for all nine taps and all channel slices; the filter comes from its
// cached planes in MFMA-fragment order (qea_pack_frag_planes: one coalesced 1 KiB wave load per fragment, L1/L2-resident,
// next step's fragments in flight under this step's MFMAs); six v_mfma_f32_32x32x16_bf16 per product in the order of the
// other split kernels.  LDS rows = pixels x C_in bf16, the 16-byte slots of pixel p stored at slot ^ ((p >> s) & m) so
// that the 16 pixels of a ds_read_b128 lane group fall on distinct slots of the 256-byte bank row.
// ---------------------------------------------------------------------------------------------
// CIN is the channel CHUNK held in LDS at a time (32, or 64 for every wider input: `chunks` = C_in / 64 passes over the same
// pixel tile, the next chunk's halo gathered into registers under this chunk's MFMAs); COUT in {32, 64, 128}.
// IMW = 0: the tile is a TH x 32 window of ONE image (W % 32 == 0, H % TH == 0).  IMW = 16 or 8: SMALL IMAGES — the image is IMW
// pixels wide and IMH = IMW / 4 high (UNet levels 4 and 5 of a 32x128 strip: 4x16 and 2x8), and a tile holds 32 / IMW images side
// by side and TH / IMH on top of each other, every image with its own zero columns left and right; rows above / below an image
// are all zero, so ONE shared zero row (stored row TH) stands for them (the row a tap reads is a compile-time function of (row,
// kh)).  These levels used to run in the generic implicit-GEMM tile, which gathers and splits every input element once per tap.
// NPL = 3: three bf16 planes, six MFMAs per product.  NPL = 2 (round 3, QEA_MFMA_SPLIT_F16): two fp16 planes of the SCALED
// operands (qea_split2_f16; scales from the input's abs-max `xmax` and from the tail of the filter planes), three MFMAs per
// product, two thirds of the LDS; the accumulators are un-scaled in the epilogue (exact: powers of two).
#ifndef QEA_HALO_NARROW_WGS
#define QEA_HALO_NARROW_WGS 3
#endif
// workgroups per CU: two where the LDS allows no more; three for the narrow fp16 instances (<= 52 KB of LDS each), whose tiles are
// bound by the latency of their halo gather and stores, not by the MFMA pipe
constexpr int halo_bf3_wgs(int cin, int cout, int npl, bool bst = false) {
  return (npl == 2 && cout <= 64 && (cin == 32 || cout == 32) && !(bst && cout == 64) && QEA_HALO_NARROW_WGS == 3) ? 3 : 2;   // (the 32->64 instance with the BatchNorm-backward sums spills at 168 registers)
}

// PKW != 0 (round 3): the max-pool that follows the layer leaves with the epilogue — window 2 x PKW over the STORED values (after scale /
// bias / ReLU), scan order and NaN rule of maxpool_fwd_kernel, written to `pooled` (pixel stride ldp) next to the full-resolution
// output; a row pair and, for PKW = 2, a pixel pair sit in one lane's accumulators (needs MI even).  No mask, no statistics.
// BST (round 3): the launch is an INPUT GRADIENT whose output da feeds a train-mode BatchNorm(+ReLU) backward: the epilogue also leaves,
// per (pixel tile, wave row) and output column, the fp64 partial sums of dz = da * [scale * yref + shift > 0] and of dz * (yref - mean) *
// invstd — what colreduce_kernel<1> computes in a pass of its own over da and yref (the partials' layout is the STATS one).
template <int CIN, int COUT, int TH, bool STATS, int IMW = 0, int NPL = 3, int PKW = 0, bool BST = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(halo_bf3_wgs(CIN, COUT, NPL, BST), halo_bf3_wgs(CIN, COUT, NPL, BST)))) void conv3x3_halo_bf3_kernel(const float* __restrict__ x, const __bf16* __restrict__ wf, float* __restrict__ y,
                                                               int B, int H, int W, int ldx, int ldy, const float* __restrict__ scale,
                                                               const float* __restrict__ bias, int relu, double* __restrict__ stats, int chunks,
                                                               int Ntot, const float* __restrict__ mask, int ldmask, int total,
                                                               const float* __restrict__ xmax, float* __restrict__ yamax,
                                                               float* __restrict__ pooled, int ldp, float* __restrict__ pamax,
                                                               const float* __restrict__ yref, int ldyref, const double* __restrict__ bst64,
                                                               const float* __restrict__ bsc, const float* __restrict__ bsh) {
  constexpr bool F16 = NPL == 2;
  typedef typename std::conditional<F16, f16x8, bf16x8>::type frag_t;
  constexpr bool SMALL = IMW != 0;
  constexpr int IMH = SMALL ? IMW / 4 : 1;               // image height in small-image mode
  constexpr int IPX = SMALL ? 32 / IMW : 1, IPY = SMALL ? TH / IMH : 1;   // images per tile, across and down
  constexpr int TW = 32, HW_ = SMALL ? IPX * (IMW + 2) : TW + 2, HH = SMALL ? TH + 1 : TH + 2, HP = HH * HW_;
  static_assert(!SMALL || (TH % IMH == 0 && COUT == 128), "small-image tiles: whole images per tile, one wave row");
  constexpr int WN = COUT / 32, WM = 4 / WN;            // waves across output channels / across tile rows
  constexpr int MI = TH / WM;                           // 32-pixel rows per wave
  constexpr int KS = CIN / 16;                          // 16-channel MFMA k-steps per tap
  constexpr int SLOTS = CIN / 8;                        // 16-byte slots per pixel row (4 or 8)
  constexpr int PLANE = HP * CIN;                       // bf16 elements per plane
  static_assert(MI >= 1 && TH % WM == 0, "tile rows must split over the waves");
  static_assert(PKW == 0 || (MI % 2 == 0 && !STATS && (PKW == 1 || PKW == 2)), "fused pooling: row pairs inside one wave, no statistics");
  static_assert(!BST || (!STATS && PKW == 0), "BatchNorm-backward sums: their own instances");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* As = reinterpret_cast<__bf16*>(smem);         // [NPL][HP][CIN] (16-bit elements: bf16, or fp16 when NPL == 2)
  float sx = 1.f, inv_x = 1.f, inv_w = 1.f;
  if constexpr (F16) {
    qea_f16_scale(xmax[0], sx, inv_x);
    // the filter's inverse scale sits behind its planes (qea_pack_frag_planes_f16)
    inv_w = reinterpret_cast<const float*>(wf + (size_t)Ntot * 9 * chunks * CIN * NPL)[0];
  }

  const int tiles_x = W / TW, tiles_y = H / TH;
  // Work item = (pixel tile, 128-channel group), `total` of them.  The groups of one tile read the same input halo and get
  // consecutive slots of one XCD (swizzle).  The grid is PERSISTENT (at most two workgroups per CU, what the LDS allows):
  // a workgroup walks items blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8 or == total, so all its items stay on
  // its XCD's contiguous range) and gathers the NEXT item's first halo chunk into registers under the current item's last
  // MFMA phase — the 2-4 us of HBM latency at the head of every tile were 30-40 % of a narrow layer's tile time.
  const int nblk = Ntot / COUT;
  struct Item { int nb, tile_id, b, x0, y0; };
  auto decode = [&](int vb) {
    const int lid = qea_xcd_swizzle(vb, total);
    Item it;
    it.nb = lid % nblk;
    it.tile_id = lid / nblk;
    if (SMALL) {                                          // tile_id = group of IPX * IPY consecutive images
      it.b = it.tile_id * (IPX * IPY);
      it.x0 = it.y0 = 0;
      return it;
    }
    int bid = it.tile_id;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    it.b = bid / tiles_y;
    it.x0 = tx * TW;
    it.y0 = ty * TH;
    return it;
  };

  // slot swizzle: 64-byte rows (CIN 32) put 4 pixels in a bank row -> key (p >> 2) & 3; 128-byte rows (CIN 64): 2 pixels -> (p >> 1) & 7
  auto swz = [](int p, int slot) { return SLOTS == 4 ? (slot ^ ((p >> 2) & 3)) : (slot ^ ((p >> 1) & 7)); };

  // ---- halo of one channel chunk: gather (zero outside the image), split once, three planes into LDS
  constexpr int C4 = CIN / 4;
  constexpr int NLD = (HP * C4 + 255) / 256;
  f32x4 hv[NLD];
  // element e = tid + 256 i of the halo: float4 chunk c4 = e % C4 (the same for every i: 256 % C4 == 0), pixel q = e / C4 walks
  // in steps of 256 / C4 (< 34), so (hx, hy) advance without divisions
  constexpr int QS = 256 / C4;
  auto gather = [&](const Item& it, int chunk, int tid) {
    const float* xb = x + (size_t)it.b * H * W * ldx + chunk * CIN + (tid % C4) * 4;
    int q = tid / C4;
    int hy = q / HW_, hx = q - hy * HW_;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int iy, ix;
      bool ok;
      if (SMALL) {                                        // stored row hy < TH = image row hy % IMH of image row-group hy / IMH; row TH = zeros
        const int gx = hx / (IMW + 2);
        const int img = (hy / IMH) * IPX + gx;
        ix = hx - gx * (IMW + 2) - 1;
        iy = (img * IMH + hy % IMH);                      // rows of consecutive images are consecutive in memory (W == IMW, H == IMH)
        ok = q < HP && hy < TH && (unsigned)ix < (unsigned)IMW && it.b + img < B;
      } else {
        iy = it.y0 + hy - 1;
        ix = it.x0 + hx - 1;
        ok = q < HP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      }
      const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? xb + ((size_t)iy * W + ix) * ldx : x);
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      hv[i] = ok ? v : zero;
      q += QS;
      hx += QS;
      if (hx >= HW_) {
        hx -= HW_;
        ++hy;
      }
    }
  };
  auto stage = [&](int tid) {
    const int c4 = tid % C4;
    int q = tid / C4;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      if (q < HP) {
        const int o = q * CIN + swz(q, c4 >> 1) * 8 + (c4 & 1) * 4;
        if constexpr (F16) {
          f16x4 h, l;
          qea_split2_f16(hv[i], sx, h, l);
          *reinterpret_cast<f16x4*>(As + o) = h;
          *reinterpret_cast<f16x4*>(As + PLANE + o) = l;
        } else {
          bf16x4 h, m, l;
          qea_split3(hv[i], h, m, l);
          *reinterpret_cast<bf16x4*>(As + o) = h;
          *reinterpret_cast<bf16x4*>(As + PLANE + o) = m;
          *reinterpret_cast<bf16x4*>(As + 2 * PLANE + o) = l;
        }
      }
      q += QS;
    }
  };

  // filter fragments: wf[n-block][chunk][step = tap*KS + cs][plane][nj][lane][8]; this wave's nj = wn
  constexpr int STEPS = 9 * KS;
  frag_t bq[2][NPL];
  auto load_b = [&](int nb, int gst, int buf, int tid) {   // gst = chunk * STEPS + step
    const frag_t* wl = reinterpret_cast<const frag_t*>(wf) + ((tid >> 6) % WN) * 64 + (tid & 63) + (size_t)nb * chunks * STEPS * NPL * WN * 64;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) bq[buf][pl] = wl[(size_t)(gst * NPL + pl) * WN * 64];
  };

  int vb = blockIdx.x;
  Item cur = decode(vb);
  gather(cur, 0, threadIdx.x);
  load_b(cur.nb, 0, 0, threadIdx.x);
  bool first = true;
  // abs-max of everything this workgroup stores, committed ONCE after its last item: a commit per item put an L2 round trip (the
  // gate's load of the shared slot) at the end of every tile and 262 k same-address accesses into one launch of the 32x128 level
  float am = 0.f, pm = 0.f;
  while (true) {
    const int nvb = vb + gridDim.x;
    const bool has_next = nvb < total;
    const Item nxt = decode(has_next ? nvb : vb);
    f32x16 acc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    for (int chunk = 0; chunk < chunks; ++chunk) {
      // the thread id as an opaque value per chunk: every gather / staging / fragment address below is a function of it, i.e.
      // invariant over the chunk and item loops (the XOR swizzle makes each of the 36 x MI fragment addresses its own
      // register) — hoisted they cost > 100 live registers (the one-item kernel needed 278 = ONE workgroup per CU);
      // recomputed per chunk they are ~700 VALU ops against 860 MFMAs per wave
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      const int lane = tid & 63, wave = tid >> 6;
      const int wm = wave / WN;
      const int fr = lane & 31, fh = lane >> 5;
      if (!first) __syncthreads();                      // every wave has read the previous planes
      first = false;
      stage(tid);
      __syncthreads();
      if (chunk + 1 < chunks) gather(cur, chunk + 1, tid);   // in flight under the MFMAs below
      else if (has_next) gather(nxt, 0, tid);                // ... or the next item's first chunk
      // A fragments of (step, row) — flat index f = st * MI + i — are read ONE fragment row ahead of their MFMAs, across step
      // boundaries too: the three ds_read_b128 of row f + 1 are issued before the six MFMAs of row f (hipcc otherwise reads just
      // in time and every row starts with an exposed LDS round trip: 239-250 -> 261-269 TFLOP/s on the 512-channel layers)
      auto read_a = [&](int st, int i, frag_t* a) {
        const int tap = st / KS, cs = st % KS;
        const int kh = tap / 3, kw = tap % 3;
        int hp;
        if (SMALL) {                                      // WM == 1: the tile row is i (compile time), so is the stored row of tap kh
          const int rr = i % IMH + kh - 1;
          const int srow = (rr >= 0 && rr < IMH) ? (i / IMH) * IMH + rr : TH;
          hp = srow * HW_ + fr + 2 * (fr / IMW) + kw;     // lane's pixel x = fr: image fr / IMW, its columns start after the zero column
        } else {
          hp = (wm * MI + i + kh) * HW_ + fr + kw;
        }
        const __bf16* src = As + hp * CIN + swz(hp, cs * 2 + fh) * 8;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) a[pl] = *reinterpret_cast<const frag_t*>(src + pl * PLANE);
      };
      frag_t ar[2][NPL];
      read_a(0, 0, ar[0]);
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        const int cb = st & 1;                          // STEPS is even for KS = 2, 4: the buffer parity carries over chunks and items
        if (st + 1 < STEPS || chunk + 1 < chunks) load_b(cur.nb, chunk * STEPS + st + 1, cb ^ 1, tid);
        else if (has_next) load_b(nxt.nb, 0, cb ^ 1, tid);
        __builtin_amdgcn_sched_barrier(0);              // keep the next step's filter loads AHEAD of this step's MFMAs
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int f = st * MI + i;
          const frag_t* a = ar[f & 1];
          const bool more = f + 1 < STEPS * MI;
          if (more) read_a((f + 1) / MI, (f + 1) % MI, ar[(f + 1) & 1]);
          if constexpr (F16) {
            // smallest terms first (ll is dropped): lh, hl, hh
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], bq[cb][0], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bq[cb][1], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bq[cb][0], acc[i], 0, 0, 0);
            if (more) {
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // the two LDS reads of row f + 1 ...
              __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);   // ... ahead of the three MFMAs of row f
            }
          }
          else {
            // smallest terms first (ll-class terms are dropped): lh, hl, mm, mh, hm, hh
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bq[cb][0], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[cb][2], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bq[cb][1], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bq[cb][0], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[cb][1], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[cb][0], acc[i], 0, 0, 0);
            if (more) {
              __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // the three LDS reads of row f + 1 ...
              __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);   // ... ahead of the six MFMAs of row f
            }
          }
        }
      }
    }

    // ---- store: row (= pixel x) = (r&3) + 8*(r>>2) + 4*fh, col (= channel) = nb*COUT + wn*32 + fr
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fh = lane >> 5;
    const int n = cur.nb * COUT + wn * 32 + fr;
    const float esc = scale ? scale[n] : 1.f, ebi = bias ? bias[n] : 0.f;
    float msc = 0.f, msh = 0.f;
    double bmu = 0.0, bis = 0.0;
    if constexpr (BST) {
      msc = bsc[n];
      msh = bsh[n];
      bmu = bst64[n];
      bis = bst64[Ntot + n];
    }
    double st0 = 0.0, st1 = 0.0;
    if constexpr (PKW != 0) {
#pragma unroll
      for (int i = 0; i < MI; i += 2) {
        float vv[2][16];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int px = (r & 3) + 8 * (r >> 2) + 4 * fh;
            size_t prow;
            bool live = true;
            if (SMALL) {
              const int img = ((i + ii) / IMH) * IPX + px / IMW;
              prow = ((size_t)(cur.b + img) * IMH + (i + ii) % IMH) * IMW + px % IMW;
              live = cur.b + img < B;
            } else {
              prow = (size_t)(cur.b * H + cur.y0 + wm * MI + i + ii) * W + cur.x0 + px;
            }
            float v = acc[i + ii][r];
            if constexpr (F16) v = (v * inv_x) * inv_w;
            if (scale && bias) v = __fmaf_rn(v, esc, ebi);
            else if (scale) v *= esc;
            else if (bias) v += ebi;
            if (relu) v = fmaxf(v, 0.f);
            vv[ii][r] = v;
            if (!live) continue;
            y[prow * ldy + n] = v;
            am = qea_amax_acc(am, v);
          }
#pragma unroll
        for (int r = 0; r < 16; r += PKW) {
          const int px = (r & 3) + 8 * (r >> 2) + 4 * fh;     // even when PKW == 2 (r even)
          float m = -INFINITY;
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int jj = 0; jj < PKW; ++jj) {
              const float v = vv[ii][r + jj];
              m = (v > m || v != v) ? v : m;
            }
          size_t prow;
          bool live = true;
          if (SMALL) {
            const int img = (i / IMH) * IPX + px / IMW;
            prow = ((size_t)(cur.b + img) * (IMH / 2) + (i % IMH) / 2) * (IMW / PKW) + (px % IMW) / PKW;
            live = cur.b + img < B;
          } else {
            prow = ((size_t)cur.b * (H / 2) + (cur.y0 + wm * MI + i) / 2) * (W / PKW) + (cur.x0 + px) / PKW;
          }
          if (!live) continue;
          pooled[prow * ldp + n] = m;
          pm = qea_amax_acc(pm, m);
        }
      }
    } else if constexpr (BST) {
      // (the instances with the BatchNorm-backward sums keep the per-element form: in the hoisted form below their fp64 sums and the
      //  second row base spill — 819 -> 1251 us on the 32x128 level)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = (r & 3) + 8 * (r >> 2) + 4 * fh;
          size_t prow;
          bool live = true;
          if (SMALL) {
            const int img = (i / IMH) * IPX + px / IMW;
            prow = ((size_t)(cur.b + img) * IMH + i % IMH) * IMW + px % IMW;
            live = cur.b + img < B;
          } else {
            prow = (size_t)(cur.b * H + cur.y0 + wm * MI + i) * W + cur.x0 + px;
          }
          float v = acc[i][r];
          if constexpr (F16) v = (v * inv_x) * inv_w;
          if (scale && bias) v = __fmaf_rn(v, esc, ebi);
          else if (scale) v *= esc;
          else if (bias) v += ebi;
          if (relu) v = fmaxf(v, 0.f);
          if (!live) continue;
          if (mask) v = (mask[prow * ldmask + n] > 0.f) ? v : 0.f;
          y[prow * ldy + n] = v;
          am = qea_amax_acc(am, v);
          const float yv = yref[prow * ldyref + n];       // (the very mask and the very terms of colreduce_kernel<1>)
          const float dz = __fmaf_rn(yv, msc, msh) > 0.f ? v : 0.f;
          st0 += (double)dz;
          st1 += (double)dz * (((double)yv - bmu) * bis);
        }
      }
    } else {
      // The option tests (scale / bias / ReLU / mask: the same for every element) sit OUTSIDE the element loops, the accumulators are
      // finished in place, and a tile row's addresses are a uniform 64-bit row base + a 32-bit lane / column offset.  (The first form
      // tested the options and multiplied 64-bit pixel indices per element: 2900 instructions and 316 branches for 64 stores, with
      // spilled scalar registers read back lane by lane — 12-35 % on top of a tile's MFMA time.)
      if constexpr (F16) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] = (acc[i][r] * inv_x) * inv_w;   // un-scale: exact (powers of two), one factor at a time
      }
      if (scale && bias) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] = __fmaf_rn(acc[i][r], esc, ebi);
      } else if (scale) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] *= esc;
      } else if (bias) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] += ebi;
      }
      if (relu) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] = fmaxf(acc[i][r], 0.f);
      }
      auto store_rows = [&](auto has_mask) {
        constexpr bool HAS_MASK = decltype(has_mask)::value;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          // non-SMALL: pixel (tile row wm * MI + i, column px) = row base (uniform over the wave) + px, px = c(r) + 4 fh
          const int rowpix = (cur.b * H + cur.y0 + wm * MI + i) * W + cur.x0;
          float* yb = y + (size_t)rowpix * ldy;
          const float* mb = HAS_MASK ? mask + (size_t)rowpix * ldmask : nullptr;
          const int lo = 4 * fh * ldy + n, lom = 4 * fh * ldmask + n;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2);
            float v = acc[i][r];
            size_t prow = 0;
            if (SMALL) {                                  // several images per tile: per-element image / row / column
              const int px = c + 4 * fh;
              const int img = (i / IMH) * IPX + px / IMW;
              prow = ((size_t)(cur.b + img) * IMH + i % IMH) * IMW + px % IMW;
              if (cur.b + img >= B) continue;             // the last tile of an image count that is no multiple of IPX * IPY
              if (HAS_MASK) v = (mask[prow * ldmask + n] > 0.f) ? v : 0.f;
              y[prow * ldy + n] = v;
            } else {
              if (HAS_MASK) v = (mb[lom + c * ldmask] > 0.f) ? v : 0.f;   // ReLU mask of another tensor (input gradient through a bare ReLU)
              yb[lo + c * ldy] = v;
            }
            am = qea_amax_acc(am, v);
            if (STATS) {
              st0 += (double)v;
              st1 += (double)v * (double)v;
            }
          }
        }
      };
      if (mask) store_rows(std::true_type{});
      else store_rows(std::false_type{});
    }
    if (STATS || BST) {                                        // one partial per (pixel tile, wave row): [blocks][Ntot][2]
      const double sa = st0 + __shfl_xor(st0, 32, 64);
      const double sc = st1 + __shfl_xor(st1, 32, 64);
      if (fh == 0) {
        double* dst = stats + ((size_t)(cur.tile_id * WM + wm) * Ntot + n) * 2;
        dst[0] = sa;
        dst[1] = sc;
      }
    }
    if (!has_next) break;
    cur = nxt;
    vb = nvb;
  }
  qea_amax_commit_block(am, yamax);
  if constexpr (PKW != 0) qea_amax_commit_block(pm, pamax);
}

// w [N][9][Cin] fp32 -> fragment-ordered planes [chunk][step = tap*KS + cs][plane][nj][lane][8 bf16], chunk width CW (32 or 64
// channels), KS = CW / 16: lane (n = nj*32 + (lane & 31), half = lane >> 5) holds channels chunk*CW + cs*16 + 8*half + j of filter
// row n at `tap`
__global__ void pack_frag_planes_kernel(const float* __restrict__ w, __bf16* __restrict__ dst, int N, int Cin, int CW) {
  const int NB = N > 128 ? 128 : N;                            // output channels per n-block (one workgroup column)
  const int KSr = CW / 16, WNr = NB / 32, chunks = Cin / CW;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;         // (n-block, chunk, step, nj, lane)
  if (i >= (N / NB) * chunks * 9 * KSr * WNr * 64) return;
  const int lane = i & 63;
  const int nj = (i >> 6) % WNr;
  const int gst = (i >> 6) / WNr;                              // (n-block, chunk, step) flattened
  const int nbk = gst / (chunks * 9 * KSr);
  const int chunk = (gst / (9 * KSr)) % chunks, st = gst % (9 * KSr);
  const int tap = st / KSr, cs = st % KSr;
  const int n = nbk * NB + nj * 32 + (lane & 31);
  const float* src = w + ((size_t)n * 9 + tap) * Cin + chunk * CW + cs * 16 + 8 * (lane >> 5);
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
  bf16x4 h0, m0, l0, h1, m1, l1;
  qea_split3(v0, h0, m0, l0);
  qea_split3(v1, h1, m1, l1);
  bf16x8 pl[3];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    pl[0][k] = h0[k]; pl[0][k + 4] = h1[k];
    pl[1][k] = m0[k]; pl[1][k + 4] = m1[k];
    pl[2][k] = l0[k]; pl[2][k + 4] = l1[k];
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(dst + ((((size_t)gst * 3 + p) * WNr + nj) * 64 + lane) * 8) = pl[p];
}

// the fp16 two-plane form of pack_frag_planes_kernel: [n-block][chunk][step][plane h, l][nj][lane][8 fp16] of the filter SCALED by
// s_w (qea_f16_scale of the filter's abs-max `wmax`), followed — at element offset N * 9 * Cin * 2 — by one float: 1 / s_w
__global__ void pack_frag_planes_f16_kernel(const float* __restrict__ w, _Float16* __restrict__ dst, int N, int Cin, int CW,
                                            const float* __restrict__ wmax) {
  const int NB = N > 128 ? 128 : N;
  const int KSr = CW / 16, WNr = NB / 32, chunks = Cin / CW;
  float sw, inv;
  qea_f16_scale(wmax[0], sw, inv);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;         // (n-block, chunk, step, nj, lane)
  if (i == 0) reinterpret_cast<float*>(dst + (size_t)N * 9 * Cin * 2)[0] = inv;
  if (i >= (N / NB) * chunks * 9 * KSr * WNr * 64) return;
  const int lane = i & 63;
  const int nj = (i >> 6) % WNr;
  const int gst = (i >> 6) / WNr;
  const int nbk = gst / (chunks * 9 * KSr);
  const int chunk = (gst / (9 * KSr)) % chunks, st = gst % (9 * KSr);
  const int tap = st / KSr, cs = st % KSr;
  const int n = nbk * NB + nj * 32 + (lane & 31);
  const float* src = w + ((size_t)n * 9 + tap) * Cin + chunk * CW + cs * 16 + 8 * (lane >> 5);
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
  f16x4 h0, l0, h1, l1;
  qea_split2_f16(v0, sw, h0, l0);
  qea_split2_f16(v1, sw, h1, l1);
  f16x8 pl[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    pl[0][k] = h0[k]; pl[0][k + 4] = h1[k];
    pl[1][k] = l0[k]; pl[1][k + 4] = l1[k];
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) *reinterpret_cast<f16x8*>(dst + ((((size_t)gst * 2 + p) * WNr + nj) * 64 + lane) * 8) = pl[p];
}

// ... and for 64-channel chunks (C_in % 64 == 0) in the order of conv3x3_halo_m16_kernel: [n-block][chunk][step = tap * 2 + ks][plane][16-channel
// group][lane][8 fp16]: lane l of group g holds filter row n-block * NB + g * 16 + (l & 15), channels chunk * 64 + ks * 32 + 8 (l >> 4) + j
__global__ void pack_frag_planes_f16_m16_kernel(const float* __restrict__ w, _Float16* __restrict__ dst, int N, int Cin, const float* __restrict__ wmax) {
  const int NB = N > 128 ? 128 : N;
  const int NGr = NB / 16, chunks = Cin / 64;
  float sw, inv;
  qea_f16_scale(wmax[0], sw, inv);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;         // (n-block, chunk, step, group, lane)
  if (i == 0) reinterpret_cast<float*>(dst + (size_t)N * 9 * Cin * 2)[0] = inv;
  if (i >= (N / NB) * chunks * 18 * NGr * 64) return;
  const int lane = i & 63;
  const int ng = (i >> 6) % NGr;
  const int gst = (i >> 6) / NGr;                              // (n-block, chunk, step) flattened
  const int nbk = gst / (chunks * 18);
  const int chunk = (gst / 18) % chunks, st = gst % 18;
  const int tap = st / 2, ks = st % 2;
  const int n = nbk * NB + ng * 16 + (lane & 15);
  const float* src = w + ((size_t)n * 9 + tap) * Cin + chunk * 64 + ks * 32 + 8 * (lane >> 4);
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
  f16x4 h0, l0, h1, l1;
  qea_split2_f16(v0, sw, h0, l0);
  qea_split2_f16(v1, sw, h1, l1);
  f16x8 pl[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    pl[0][k] = h0[k]; pl[0][k + 4] = h1[k];
    pl[1][k] = l0[k]; pl[1][k + 4] = l1[k];
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) *reinterpret_cast<f16x8*>(dst + ((((size_t)gst * 2 + p) * NGr + ng) * 64 + lane) * 8) = pl[p];
}

template <int CIN, int COUT, int TH, bool STATS, int IMW = 0, int NPL = 3, int PKW = 0, bool BST = false>
int launch_halo_bf3_(const ConvArgs& a, hipStream_t s) {
  constexpr size_t lds = IMW ? (size_t)NPL * (TH + 1) * (32 / IMW) * (IMW + 2) * CIN * 2 : (size_t)NPL * (TH + 2) * 34 * CIN * 2;
  auto kern = conv3x3_halo_bf3_kernel<CIN, COUT, TH, STATS, IMW, NPL, PKW, BST>;
  static int attr_rc = (int)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr_rc != (int)hipSuccess) {
    qea_set_error("qea_conv_igemm(halo bf3): cannot reserve %zu bytes of LDS: %s", (size_t)lds, hipGetErrorString((hipError_t)attr_rc));
    return QEA_ERR_LAUNCH;
  }
  const long long total = IMW ? (long long)qea_cdiv(a.B, (32 / IMW) * (TH / (IMW / 4))) * (a.N / COUT)
                              : (long long)a.B * (a.H / TH) * (a.W / 32) * (a.N / COUT);
  if (total <= 0 || total > 0x7fffffffLL) {
    qea_set_error("qea_conv_igemm(halo bf3): grid %lld out of range", total);
    return QEA_ERR_INVALID;
  }
  // persistent: two workgroups per CU (the LDS bound) once there are more items than that
  static const int resident = [] {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return halo_bf3_wgs(CIN, COUT, NPL, BST) * (cus & ~7);
  }();
  // (32-channel outputs keep one item per workgroup: two persistent workgroups of a CU fall into lockstep there — both staging,
  // then both in their MFMA phase — and lose the overlap that staggered dispatch gives: 150-159 vs 159-166 TFLOP/s measured)
  // (re-measured with the fp16 split and three workgroups per CU: persistent 0.779 / 1.431 ms against 0.750 / 1.331 one item each)
  const unsigned grid = (total > resident && COUT > 32) ? (unsigned)resident : (unsigned)total;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a.x, (const __bf16*)a.wp, a.y, a.B, a.H, a.W, a.ldx, a.ldy,
                     a.scale, a.bias, a.relu, a.stats, a.Cin / CIN, a.N, a.mask, a.ldmask, (int)total, a.xmax, a.yamax, a.pool_y, a.ldpool, a.pool_amax,
                     a.bst_y, a.ldbst, a.bst64, a.bst_scale, a.bst_shift);
  return QEA_OK;
}

// the instances with the fused max-pool (fp16 form, no statistics): the layers that are followed by a pool — CRNN conv2 (2x2) and conv4
// (2x1), the second conv of the UNet encoder levels in the inference pass (BatchNorm + ReLU in the epilogue)
template <int CIN, int COUT, int TH, int IMW = 0>
int launch_halo_bf3_pool(const ConvArgs& a, hipStream_t s) {
  if (a.pool_kw == 1) return launch_halo_bf3_<CIN, COUT, TH, false, IMW, 2, 1>(a, s);
  return launch_halo_bf3_<CIN, COUT, TH, false, IMW, 2, 2>(a, s);
}

template <int CIN, int COUT, int TH, int IMW = 0>
int launch_halo_bf3(const ConvArgs& a, hipStream_t s) {
  if (a.xmax && a.bst_y)                                   // ... with the BatchNorm-backward sums of the tensor it writes (fp16 form only)
    return launch_halo_bf3_<CIN, COUT, TH, false, IMW, 2, 0, true>(a, s);
  if (a.xmax)                                              // two-way fp16 split: the caller gave the input's abs-max and fp16 filter planes
    return a.stats ? launch_halo_bf3_<CIN, COUT, TH, true, IMW, 2>(a, s) : launch_halo_bf3_<CIN, COUT, TH, false, IMW, 2>(a, s);
  return a.stats ? launch_halo_bf3_<CIN, COUT, TH, true, IMW>(a, s) : launch_halo_bf3_<CIN, COUT, TH, false, IMW>(a, s);
}

bool halo_eligible(const qea_conv_desc* d) {
  const bool ch = (d->Cin == 32 || d->Cin == 64) && (d->N == 32 || d->N == 64);
  const int th = d->Cin == 32 ? 8 : 4;
  return ch && d->KH == 3 && d->KW == 3 && d->pad_h == 1 && d->pad_w == 1 && d->stride_h == 1 && d->stride_w == 1 && d->OH == d->H &&
         d->OW == d->W && d->W % 32 == 0 && d->H % th == 0 && d->out_mode == QEA_OUT_NHWC && !d->mask && !d->accumulate;
}

// the split-bf16 LDS-halo kernel takes C_in = 32 or a multiple of 64 (up to 256) and C_out in {32, 64, 128}
// small-image tiles (several whole images per 4 x 32 tile): 4x16 and 2x8 images, C_in a multiple of 64, C_out a multiple of 128
int halo_bf3_small(const qea_conv_desc* d) {
  if (d->Cin % 64 || d->Cin > 512 || d->N % 128) return 0;
  if (d->W == 16 && d->H == 4) return 16;
  if (d->W == 8 && d->H == 2) return 8;
  return 0;
}

bool halo_bf3_eligible(const qea_conv_desc* d) {
  const bool cin = d->Cin == 32 || (d->Cin % 64 == 0 && d->Cin <= 512);
  const bool cout = d->N == 32 || d->N == 64 || d->N % 128 == 0;
  const int th = d->Cin == 32 ? 8 : 4;
  const bool shape = (d->W % 32 == 0 && d->H % th == 0) || halo_bf3_small(d) != 0;
  return cin && cout && d->KH == 3 && d->KW == 3 && d->pad_h == 1 && d->pad_w == 1 && d->stride_h == 1 && d->stride_w == 1 && d->OH == d->H &&
         d->OW == d->W && shape && d->out_mode == QEA_OUT_NHWC && !d->accumulate;
}

// pixel tiles of a launch on the split-bf16 LDS-halo kernel (each leaves halo_bf3_wm() rows of fused BatchNorm partials)
long long halo_bf3_tiles(const qea_conv_desc* d) {
  const int sm = halo_bf3_small(d);
  if (sm) return qea_cdiv(d->B, (32 / sm) * (4 / (sm / 4)));
  return (long long)d->B * (d->H / (d->Cin == 32 ? 8 : 4)) * (d->W / 32);
}

// 1 when a launch of this shape has an instance with the fused max-pool (window 2 x kw) — given the fp16 operands
bool halo_bf3_pool_shape(const qea_conv_desc* d, int kw) {
  if (!halo_bf3_eligible(d) || d->mask || d->stats || d->H % 2 || (kw != 1 && kw != 2) || (kw == 2 && d->W % 2)) return false;
  const int sm = halo_bf3_small(d);
  if (sm == 16) return kw == 2;                          // <64,128,4,16>: the 2 x 2 window
  if (sm) return false;
  if (d->Cin == 32) return d->N == 32 && kw == 2;        // <32,32,8>
  if (d->N == 64) return kw == 2;                        // <64,64,4>
  return d->N % 128 == 0;                                // <64,128,4>: both windows
}

int launch_halo_bf3_any(const qea_conv_desc* d, const ConvArgs& a, hipStream_t s) {
  const int sm = halo_bf3_small(d);
  if (a.xmax && d->Cin % 64 == 0) {                        // two-way fp16 split, 64-channel chunks: the 16x16x32 kernel (round 4)
    return qea_conv::launch_halo_m16_any(d->N == 32 ? 32 : (d->N == 64 ? 64 : 128), sm, a, s);   // (conv_halo16.hip)
  }
  if (a.pool_y) {
    if (sm == 16) return launch_halo_bf3_<64, 128, 4, false, 16, 2, 2>(a, s);
    if (d->Cin == 32) return launch_halo_bf3_<32, 32, 8, false, 0, 2, 2>(a, s);
    if (d->N == 64) return launch_halo_bf3_<64, 64, 4, false, 0, 2, 2>(a, s);
    return launch_halo_bf3_pool<64, 128, 4>(a, s);
  }
  if (sm == 16) return launch_halo_bf3<64, 128, 4, 16>(a, s);
  if (sm == 8) return launch_halo_bf3<64, 128, 4, 8>(a, s);
  if (d->Cin == 32) {
    if (d->N == 32) return launch_halo_bf3<32, 32, 8>(a, s);
    if (d->N == 64) return launch_halo_bf3<32, 64, 8>(a, s);
    return launch_halo_bf3<32, 128, 8>(a, s);                // (N = 128k: one workgroup column per 128 output channels)
  }
  if (d->N == 32) return launch_halo_bf3<64, 32, 4>(a, s);
  if (d->N == 64) return launch_halo_bf3<64, 64, 4>(a, s);
  return launch_halo_bf3<64, 128, 4>(a, s);
}

// workgroup-rows of the statistics partials: halo fp32 = 4 waves over rows; halo bf3 = WM = 4 / (COUT / 32), COUT = min(N, 128)
int halo_bf3_wm(const qea_conv_desc* d) { return 4 / ((d->N > 128 ? 128 : d->N) / 32); }

int launch_halo_any(const qea_conv_desc* d, const ConvArgs& a, hipStream_t s) {
  if (d->Cin == 32 && d->N == 32) return launch_halo<32, 32, 8>(a, s);
  if (d->Cin == 32 && d->N == 64) return launch_halo<32, 64, 8>(a, s);
  if (d->Cin == 64 && d->N == 32) return launch_halo<64, 32, 4>(a, s);
  return launch_halo<64, 64, 4>(a, s);
}

// The split-bf16 LDS-halo kernel beats the generic split tiles on EVERY shape it takes (tools/bench_conv.py, B = 512:
// 32->32 168 vs 95 fp32-halo; 128->64 211 vs 140; 128->128 208 vs 186; 256->256 226 vs 215; 512->512 238 vs 226 TFLOP/s).
bool halo_bf3_wins(const qea_conv_desc* d) { return halo_bf3_eligible(d); }

// tile 26 (gemm1x1.hip): 1x1 stride-1 GEMMs on the 128-row LDS tile, two-way fp16 split only (needs x_absmax + the planes of
// qea_pack_frag_planes_f16_1x1): K a multiple of 64, N a multiple of 128, bias / ReLU epilogue, NHWC / TBC / transposed-conv store
bool gemm1x1_eligible(const qea_conv_desc* d) {
  return d->KH == 1 && d->KW == 1 && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 0 && d->pad_w == 0 && d->OH == d->H && d->OW == d->W &&
         d->Cin % 64 == 0 && d->N % 128 == 0 && !d->scale && !d->mask && !d->accumulate && !d->stats &&
         (long long)d->N * d->Cin * 4 < 0x7fffffffLL &&
         (d->out_mode != QEA_OUT_CONVT || (long long)d->B * d->H * d->W * 4 < 0x7fffffffLL);   // (output pixel rows are kept as int32)
}

// Tile choice for tile == 0 (measured on MI355X with tools/bench_conv.py)
int pick_tile(const qea_conv_desc* d, const ConvArgs& a) {
  int tile = 0;
  // measured on MI355X (tools/bench_conv.py): the 16-deep K-slice (half the LDS, 4 workgroups per CU) wins
  // only when the grid is large enough to keep all of them busy
  const long long tiles128 = (long long)qea_cdiv(a.M, 128) * qea_cdiv(d->N, 128);
  const long long tiles7 = (long long)qea_cdiv(a.M, 256) * qea_cdiv(d->N, 128);   // 256x128, 8 waves
  const long long tiles8 = (long long)qea_cdiv(a.M, 128) * qea_cdiv(d->N, 256);   // 128x256, 8 waves
  // 8-wave workgroups (twice the tile, 16-deep K-slice, 4 waves per SIMD) reach 124-135 TFLOP/s once the grid
  // holds at least two of them per CU; below that the 4-wave tiles win (tools/bench_conv.py on MI355X)
  // split-bf16 tiles (20-23) for every layer of 64+ output channels unless QEA_MFMA=f32 asks for the native fp32 MFMA:
  // 160-186 TFLOP/s against 110-134, and closer to the fp64 result than the fp32 instruction (fewer accumulator roundings)
  const long long tiles21 = (long long)qea_cdiv(a.M, 256) * qea_cdiv(d->N, 128);
  const long long tiles22 = (long long)qea_cdiv(a.M, 128) * qea_cdiv(d->N, 256);
  // (short-K launches — the transposed convs — are bound by their output traffic, and N < 128 wastes the tile)
  const bool bf3 = qea_split_bf16_enabled() && d->N >= 128 && a.K >= 256;
  // 33..64 output channels: the split-bf16 256x64 tile beats both the fp32 256x64 tile (132 vs 107 TFLOP/s at Cin = 128) and
  // the fp32 LDS-halo kernel at Cin = 64 (120 vs 111); the halo kernel keeps Cin = 32 (K = 288: 102 vs 91)
  if (qea_split_bf16_enabled() && d->tile != -1 && gemm1x1_eligible(d)) tile = 26;     // 1x1 / transposed-conv GEMMs: 128-row LDS tile (fp16 split)
  else if (qea_split_bf16_enabled() && d->tile != -1 && halo_bf3_wins(d)) tile = 24;   // narrow layers: split-bf16 LDS-halo kernel (168-208 vs 95-120 TFLOP/s)
  else if (qea_split_bf16_enabled() && d->N > 32 && d->N <= 64 && a.K >= 256 && d->Cin >= 64) tile = 23;
  else if (halo_eligible(d)) tile = 4;
  // transposed convolutions (forward scatter / stride-2 input gradient) with K <= 256: traffic-bound launches of 4-16 K
  // stages; the 8-wave 128x128 split tile has the shortest prologue per output byte (tools/bench_convt.py at B = 2048:
  // 64->32 fwd 970 -> 821 us, dgrad 708 -> 573; 128->64 604 -> 470 / 320 -> 266; 256->128 343 -> 315 / 224 -> 216)
  else if (qea_split_bf16_enabled() && (d->out_mode == QEA_OUT_CONVT || (d->stride_h == 2 && d->stride_w == 2)) && d->N >= 64 && a.K >= 64 &&
           a.K <= 256 && d->Cin % 16 == 0)
    tile = 25;
  else if (d->N <= 32) tile = 3;
  else if (d->N <= 64) tile = 9;  // 16-deep slice: 51 KB of LDS, three workgroups per CU (107 vs 80 TFLOP/s at 32-deep; the split-bf16 256x64 tile is slower here)
  else if (bf3 && d->N % 256 == 0 && tiles22 >= 512) tile = 22;
  else if (bf3 && tiles21 >= 512) tile = 21;
  else if (bf3) tile = 25;  // small grids: 8 waves on a 128x128 tile (146-166 vs 110-150 TFLOP/s for the 4-wave tile 20)
  else if (d->N % 256 == 0 && tiles8 >= 1024) tile = 8;
  else if (tiles7 >= 512) tile = 7;
  else if (tiles128 >= 2048) tile = 5;
  else if (tiles128 < 192) tile = 6;
  else tile = 1;
  return tile;
}

// the tile a launch runs on: forced, or the automatic choice — which falls back to the fp32 halo / generic split tile when the
// narrow-layer split kernel was chosen but the caller did not supply the fragment-order filter planes
int resolve_tile(const qea_conv_desc* d, const ConvArgs& a) {
  int tile = d->tile ? d->tile : pick_tile(d, a);
  if (!d->tile && ((tile == 24 && !d->w_frag_planes) || (tile == 26 && !(d->w_frag_planes && d->x_absmax)))) {
    qea_conv_desc e = *d;
    e.tile = -1;                                           // the choice without tiles 24 / 26
    tile = pick_tile(&e, a);
  }
  return tile;
}

// Partial blocks the fused-statistics epilogue of this launch writes, 0 when the chosen kernel has none: the hybrid
// split-bf16 tiles (one block per M-tile and wave row) and the fp32 LDS-halo kernel (one per workgroup and wave); the output
// must be the plain conv result (no scale / bias / ReLU / mask / accumulate, NHWC).
int stats_blocks_for(const qea_conv_desc* d, const ConvArgs& a, int tile, bool wp3) {
  if (d->scale || d->bias || d->mask || d->relu || d->accumulate || d->out_mode != QEA_OUT_NHWC) return 0;
  if (tile == 4 && halo_eligible(d)) return d->B * (d->H / (d->Cin == 32 ? 8 : 4)) * (d->W / 32) * 4;
  if (tile == 24 && halo_bf3_eligible(d)) return (int)halo_bf3_tiles(d) * halo_bf3_wm(d);
  if (!wp3) return 0;
  switch (tile) {
    case 21: return qea_cdiv(a.M, 256) * 4;
    case 22: return qea_cdiv(a.M, 128) * 2;
    case 23: return qea_cdiv(a.M, 256) * 4;
    case 25: return qea_cdiv(a.M, 128) * 4;
    default: return 0;
  }
}

}  // namespace

extern "C" int qea_conv_igemm_stats_blocks(const qea_conv_desc* d) {
  if (!d || d->Cin <= 0 || d->Cin % 32 || d->B <= 0) return 0;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.M = d->B * d->OH * d->OW;
  a.N = d->N;
  a.K = d->KH * d->KW * d->Cin;
  const int tile = resolve_tile(d, a);
  return stats_blocks_for(d, a, tile, tile >= 20 && !d->x_planes && d->w_planes);
}

extern "C" int qea_conv_igemm(const qea_conv_desc* d, void* stream) {
  QEA_REQUIRE(d && d->x && d->w && d->y, "qea_conv_igemm: null pointer");
  QEA_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->OH > 0 && d->OW > 0 && d->N > 0,
              "qea_conv_igemm: non-positive dimension");
  QEA_REQUIRE(d->Cin > 0 && d->Cin % 32 == 0, "qea_conv_igemm: Cin=%d must be a multiple of 32", d->Cin);
  QEA_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride_h > 0 && d->stride_w > 0, "qea_conv_igemm: bad filter/stride");
  QEA_REQUIRE(d->ldx >= d->Cin && d->ldx % 4 == 0, "qea_conv_igemm: ldx=%d must be >= Cin and a multiple of 4", d->ldx);
  QEA_REQUIRE(((uintptr_t)d->x & 15) == 0 && ((uintptr_t)d->w & 15) == 0, "qea_conv_igemm: x/w must be 16-byte aligned");
  if (d->out_mode == QEA_OUT_TBC) QEA_REQUIRE(d->OH == 1, "qea_conv_igemm: QEA_OUT_TBC needs OH == 1");
  if (d->out_mode == QEA_OUT_CONVT) {
    QEA_REQUIRE(d->N % 4 == 0 && d->OH == d->H && d->OW == d->W && d->KH == 1 && d->KW == 1,
                "qea_conv_igemm: QEA_OUT_CONVT needs a 1x1 GEMM with N = 4*Cout");
    QEA_REQUIRE(d->ldy >= d->N / 4, "qea_conv_igemm: ldy too small");
  } else {
    QEA_REQUIRE(d->ldy >= d->N, "qea_conv_igemm: ldy=%d < N=%d", d->ldy, d->N);
  }
  QEA_REQUIRE((long long)d->B * d->H * d->W < 0x7fffffffLL && (long long)d->B * d->OH * d->OW < 0x7fffffffLL,
              "qea_conv_igemm: pixel count overflows int32");

  ConvArgs a;
  a.x = d->x; a.w = d->w; a.y = d->y; a.scale = d->scale; a.bias = d->bias; a.mask = d->mask;
  a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.OH = d->OH; a.OW = d->OW; a.N = d->N;
  a.KH = d->KH; a.KW = d->KW; a.pad_h = d->pad_h; a.pad_w = d->pad_w; a.stride_h = d->stride_h; a.stride_w = d->stride_w;
  a.ldx = d->ldx; a.ldy = d->ldy; a.ldmask = d->ldmask; a.relu = d->relu; a.accumulate = d->accumulate; a.out_mode = d->out_mode;
  a.xmax = nullptr;
  a.yamax = d->y_absmax;
  a.M = d->B * d->OH * d->OW;
  a.K = d->KH * d->KW * d->Cin;
  a.m_tiles = a.n_tiles = 0;
  a.xp = (const char*)d->x_planes;
  a.wp = (const char*)d->w_planes;
  a.xp_zero = a.wp_zero = 0;
  a.stats = nullptr;
  a.pool_y = nullptr;
  a.ldpool = a.pool_kw = 0;
  a.pool_amax = nullptr;
  a.bst_y = nullptr;
  a.ldbst = 0;
  a.bst64 = nullptr;
  a.bst_scale = a.bst_shift = nullptr;

  hipStream_t s = (hipStream_t)stream;
  int tile = resolve_tile(d, a);
  if (d->pool_y) {                                         // fused max-pool: only where qea_conv_igemm_can_pool says so, fp16 operands given
    QEA_REQUIRE(tile == 24 && halo_bf3_pool_shape(d, d->pool_kw) && d->x_absmax && d->w_frag_planes && d->ldpool >= d->N && d->ldpool % 4 == 0,
                "qea_conv_igemm: pool_y needs a launch qea_conv_igemm_can_pool accepts (tile 24, fp16 operands), ldpool >= N");
    a.pool_y = d->pool_y;
    a.ldpool = d->ldpool;
    a.pool_kw = d->pool_kw;
    a.pool_amax = d->pool_absmax;
  }
  if (tile == 4 && !halo_eligible(d)) {
    qea_set_error("qea_conv_igemm: tile 4 (LDS-halo 3x3) needs Cin,N in {32,64}, 3x3 pad 1 stride 1, W %% 32 == 0, no mask / accumulate");
    return QEA_ERR_INVALID;
  }
  const bool p3 = tile >= 20 && d->x_planes && d->w_planes;
  const bool wp3 = tile >= 20 && !d->x_planes && d->w_planes;
  if (p3 || wp3) {
    const bool f16w = wp3 && d->x_absmax;                  // hybrid tile on the two-way fp16 split: two planes per value
    const unsigned long long xb = p3 ? (unsigned long long)d->B * d->H * d->W * d->Cin * 6 : 0, wb = (unsigned long long)d->N * a.K * (f16w ? 4 : 6);
    if (f16w) a.xmax = d->x_absmax;
    QEA_REQUIRE(d->Cin % 16 == 0 && xb + 128 < 0x7fffffffULL && wb + 128 < 0x7fffffffULL,
                "qea_conv_igemm: pre-split operands need Cin %% 16 == 0 and planes below 2 GiB (%llu, %llu bytes)", xb, wb);
    QEA_REQUIRE(((uintptr_t)d->x_planes & 15) == 0 && ((uintptr_t)d->w_planes & 15) == 0, "qea_conv_igemm: planes must be 16-byte aligned");
    a.xp_zero = (unsigned)xb;
    a.wp_zero = (unsigned)wb;
  }
  if (d->stats) {
    const int blocks = stats_blocks_for(d, a, tile, wp3);
    QEA_REQUIRE(blocks > 0, "qea_conv_igemm: this launch cannot produce fused statistics (ask qea_conv_igemm_stats_blocks first)");
    a.stats = d->stats;
  }
  if (d->bst_y) {                                          // BatchNorm-backward sums instead of the forward statistics: LDS-halo kernel, fp16 form
    QEA_REQUIRE(d->stats && tile == 24 && d->x_absmax && !d->pool_y && d->bst_stat64 && d->bst_scale && d->bst_shift && d->ldbst >= d->N,
                "qea_conv_igemm: bst_y needs stats (the partials), tile 24 with fp16 operands, bst_stat64 / bst_scale / bst_shift, ldbst >= N");
    a.bst_y = d->bst_y;
    a.ldbst = d->ldbst;
    a.bst64 = d->bst_stat64;
    a.bst_scale = d->bst_scale;
    a.bst_shift = d->bst_shift;
  }
  if (tile == 24 && (!halo_bf3_eligible(d) || !d->w_frag_planes)) {
    qea_set_error("qea_conv_igemm: tile 24 needs Cin = 32 or 64k <= 512, N in {32,64,128k}, 3x3 pad 1 stride 1, W %% 32 == 0 (or 4x16 / 2x8 images with Cin = 64k, N = 128k), no accumulate, and w_frag_planes");
    return QEA_ERR_INVALID;
  }
  if (tile == 26 && (!gemm1x1_eligible(d) || !d->w_frag_planes || !d->x_absmax)) {
    qea_set_error("qea_conv_igemm: tile 26 needs a 1x1 stride-1 GEMM with Cin %% 64 == 0, N %% 128 == 0, no scale / mask / accumulate / stats, x_absmax and the w_frag_planes of qea_pack_frag_planes_f16_1x1");
    return QEA_ERR_INVALID;
  }
  qea_prof_begin(QEA_PROF_CONV_IGEMM, s);
  int rc;
  switch (tile) {
    case 4: rc = launch_halo_any(d, a, s); break;
    case 26:                                               // 1x1 / transposed-conv GEMM on the 128-row LDS tile (gemm1x1.hip)
      a.wp = (const char*)d->w_frag_planes;
      a.xmax = d->x_absmax;
      rc = qea_conv::launch_gemm1x1_f16(a, s);
      break;
    case 24:                                               // split-bf16 LDS-halo kernel of the narrow layers: filter in fragment-order planes
      a.wp = (const char*)d->w_frag_planes;
      a.xmax = d->x_absmax;                                // non-NULL: fp16 planes + scales (ABI v6)
      rc = launch_halo_bf3_any(d, a, s);
      break;
    case 1: rc = launch<128, 128, 2, 2, 32>(a, s); break;
    case 2: rc = launch<256, 64, 4, 1, 32>(a, s); break;
    case 3: rc = launch<256, 32, 4, 1, 32>(a, s); break;
    case 5: rc = launch<128, 128, 2, 2, 16>(a, s); break;
    case 6: rc = launch<128, 64, 2, 2, 32>(a, s); break;   // small grids: twice the workgroups of tile 1
    case 7: rc = launch<256, 128, 4, 2, 16>(a, s); break;  // 8 waves
    case 8: rc = launch<128, 256, 2, 4, 16>(a, s); break;  // 8 waves
    case 9: rc = launch<256, 64, 4, 1, 16>(a, s); break;   // tile 2 with half the LDS (3 workgroups per CU)
    // split-bf16 forms; with pre-split operands (x_planes, w_planes) the LDS-DMA kernel, else the split-on-the-fly kernel
    // (x_planes + w_planes: all-DMA kernel; w_planes only: hybrid — activations split on the fly, filter by DMA)
    case 20: rc = p3 ? launch_p3<128, 128, 2, 2>(a, s) : wp3 ? launch_bf3w<128, 128, 2, 2>(a, s) : launch_bf3<128, 128, 2, 2>(a, s); break;
    case 21: rc = p3 ? launch_p3<256, 128, 4, 2>(a, s) : wp3 ? launch_bf3w<256, 128, 4, 2>(a, s) : launch_bf3<256, 128, 4, 2>(a, s); break;
    case 22: rc = p3 ? launch_p3<128, 256, 2, 4>(a, s) : wp3 ? launch_bf3w<128, 256, 2, 4>(a, s) : launch_bf3<128, 256, 2, 4>(a, s); break;
    case 23: rc = p3 ? launch_p3<256, 64, 4, 1>(a, s) : wp3 ? launch_bf3w<256, 64, 4, 1>(a, s) : launch_bf3<256, 64, 4, 1>(a, s); break;
    case 25: rc = p3 ? launch_p3<128, 128, 4, 2>(a, s) : wp3 ? launch_bf3w<128, 128, 4, 2>(a, s) : launch_bf3<128, 128, 4, 2>(a, s); break;  // 8 waves on a 128x128 tile: small grids
    default: qea_set_error("qea_conv_igemm: unknown tile id %d", tile); rc = QEA_ERR_INVALID; break;
  }
  if (rc != QEA_OK) {
    qea_prof_abort(QEA_PROF_CONV_IGEMM);
    return rc;
  }
  // algorithmic bytes: input once + filter once + output once
  const double abytes = 4.0 * ((double)d->B * d->H * d->W * d->Cin + (double)d->N * a.K + (double)a.M * d->N);
  // tag = the kernel that ran: QEA_PROF_TAG_CONV(tile, input-channel chunk, output-channel group, fused statistics) for the
  // LDS-halo split kernel (its template instantiation), the tile id otherwise
  // (the small-image instantiations of the halo kernel are their own kernels in a trace: + 20 * image width)
  // (... and so are the instances with the fused max-pool: + 2000 * window width)
  const int tag = tile == 24 ? QEA_PROF_TAG_HALO_BF3(d->Cin == 32 ? 32 : 64, d->N > 128 ? 128 : d->N, a.stats != nullptr) + 20 * halo_bf3_small(d) +
                                   (a.pool_y ? 2000 * a.pool_kw : 0)
                             : tile;
  qea_prof_end(QEA_PROF_CONV_IGEMM, s, 2.0 * a.M * (double)a.N * a.K, abytes, tile >= 20 ? (a.xmax ? 2 : 1) : 0, tag + (tile == 24 && a.xmax ? 5 : 0));
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_conv_igemm_can_pool(const qea_conv_desc* d, int32_t kw) {
  if (!d || d->Cin <= 0 || d->Cin % 32 || d->B <= 0) return 0;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.M = d->B * d->OH * d->OW;
  a.N = d->N;
  a.K = d->KH * d->KW * d->Cin;
  const int tile = d->tile ? d->tile : pick_tile(d, a);
  return (tile == 24 && halo_bf3_pool_shape(d, kw)) ? 1 : 0;
}

extern "C" int qea_conv_igemm_uses_split_bf16(const qea_conv_desc* d) {
  if (!d || d->Cin <= 0 || d->Cin % 32 || d->B <= 0) return 0;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.M = d->B * d->OH * d->OW;
  a.N = d->N;
  a.K = d->KH * d->KW * d->Cin;
  const int tile = d->tile ? d->tile : pick_tile(d, a);
  return tile >= 20 ? 1 : 0;
}

extern "C" size_t qea_split_planes_bytes(int64_t M, int32_t C) { return (size_t)M * (size_t)C * 6 + 128; }

extern "C" int qea_split_planes(const float* x, int32_t ld, int64_t M, int32_t C, void* planes, void* stream) {
  QEA_REQUIRE(x && planes && M > 0 && C > 0 && C % 16 == 0 && ld >= C && ld % 4 == 0, "qea_split_planes: bad arguments (C=%d must be a multiple of 16)", C);
  QEA_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)planes & 15) == 0, "qea_split_planes: pointers must be 16-byte aligned");
  const long long total = M * (C >> 3);
  const long long blocks = (total + 255) / 256;
  QEA_REQUIRE(blocks < 0x7fffffffLL, "qea_split_planes: too large");
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ld, (long long)M, C, (__bf16*)planes);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" size_t qea_split_planes_f16_bytes(int64_t M, int32_t C) { return (size_t)M * (size_t)C * 4 + 128 + 16; }

extern "C" int qea_split_planes_f16(const float* x, int32_t ld, int64_t M, int32_t C, const float* xmax, void* planes, void* stream) {
  QEA_REQUIRE(x && planes && xmax && M > 0 && C > 0 && C % 16 == 0 && ld >= C && ld % 4 == 0, "qea_split_planes_f16: bad arguments (C=%d must be a multiple of 16)", C);
  QEA_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)planes & 15) == 0, "qea_split_planes_f16: pointers must be 16-byte aligned");
  const long long total = M * (C >> 3);
  const long long blocks = (total + 255) / 256;
  QEA_REQUIRE(blocks < 0x7fffffffLL, "qea_split_planes_f16: too large");
  hipLaunchKernelGGL(split_planes_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ld, (long long)M, C, xmax, (_Float16*)planes);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" size_t qea_pack_frag_planes_bytes(int32_t N, int32_t Cin) { return (size_t)N * 9 * Cin * 6; }

extern "C" int qea_pack_frag_planes(const float* w, int32_t N, int32_t Cin, void* planes, void* stream) {
  QEA_REQUIRE(w && planes && (N == 32 || N == 64 || (N > 0 && N % 128 == 0)) && (Cin == 32 || (Cin % 64 == 0 && Cin <= 512)),
              "qea_pack_frag_planes: N in {32, 64, 128k}, Cin = 32 or a multiple of 64 up to 512");
  const int total = 9 * (Cin / 16) * (N / 32) * 64;
  hipLaunchKernelGGL(pack_frag_planes_kernel, dim3(qea_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)planes, N, Cin,
                     Cin == 32 ? 32 : 64);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" size_t qea_pack_frag_planes_f16_bytes(int32_t N, int32_t Cin) { return (size_t)N * 9 * Cin * 4 + 16; }

extern "C" int qea_pack_frag_planes_f16(const float* w, int32_t N, int32_t Cin, const float* wmax, void* planes, void* stream) {
  QEA_REQUIRE(w && planes && wmax && (N == 32 || N == 64 || (N > 0 && N % 128 == 0)) && (Cin == 32 || (Cin % 64 == 0 && Cin <= 512)),
              "qea_pack_frag_planes_f16: N in {32, 64, 128k}, Cin = 32 or a multiple of 64 up to 512");
  if (Cin % 64 == 0) {                                     // 64-channel chunks: the order of conv3x3_halo_m16_kernel (16x16x32 MFMA)
    const int total16 = 9 * (Cin / 32) * (N / 16) * 64;
    hipLaunchKernelGGL(pack_frag_planes_f16_m16_kernel, dim3(qea_cdiv(total16, 256)), dim3(256), 0, (hipStream_t)stream, w, (_Float16*)planes, N, Cin, wmax);
    QEA_CHECK_LAUNCH();
    return QEA_OK;
  }
  const int total = 9 * (Cin / 16) * (N / 32) * 64;
  hipLaunchKernelGGL(pack_frag_planes_f16_kernel, dim3(qea_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, (_Float16*)planes, N, Cin,
                     Cin == 32 ? 32 : 64, wmax);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

/* 1 when qea_conv_igemm would pick the LDS-halo kernel (tile 24: wants the w_frag_planes of qea_pack_frag_planes[_f16]); 2 when it
 * would pick the 1x1 LDS tile (tile 26: wants those of qea_pack_frag_planes_f16_1x1 AND x_absmax — without them the launch runs on
 * the generic tiles); else 0 */
extern "C" int qea_conv_igemm_wants_frag_planes(const qea_conv_desc* d) {
  if (!d || d->Cin <= 0 || d->Cin % 32 || d->B <= 0) return 0;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.M = d->B * d->OH * d->OW;
  a.N = d->N;
  a.K = d->KH * d->KW * d->Cin;
  const int tile = d->tile ? d->tile : pick_tile(d, a);
  return tile == 24 ? 1 : (tile == 26 ? 2 : 0);
}
