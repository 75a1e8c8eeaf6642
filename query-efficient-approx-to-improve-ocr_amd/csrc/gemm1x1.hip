// 1x1 convolutions / plain GEMMs / the transposed convolution's forward GEMM on a 128-row LDS tile (round 3, two-way fp16 split).
//
//   Y[m][n] = sum_k X[m][k] * Wt[n][k] (+ bias)      m = pixel / sequence row, K = C_in (a multiple of 64), N a multiple of 128
//
// The generic implicit-GEMM tile (conv_igemm_bf3w_kernel) steps K sixteen channels at a time with one barrier and ONE 16-byte load
// per thread in flight per step: the short-K launches of the step (transposed convs: K = 64..512, 1.6 GB of traffic for 34 GFLOP)
// ran at 1.8 TB/s and the BiLSTM projections (K = 512, N = 1024) at 200 TFLOP/s.  Here a workgroup takes 128 rows x 64 channels at
// a time: eight 16-byte loads per thread in flight, split ONCE into the two fp16 planes in LDS (rows of 128 B, 16-byte slots
// XOR-swizzled as in the LDS-halo conv), the next chunk — or the next work item's first chunk — gathered into registers under this
// chunk's MFMAs; the filter comes from its cached fragment-order planes (qea_pack_frag_planes_f16_1x1, L2-resident), one step
// ahead of its MFMAs.  Four waves side by side over NB = 128 or 256 output channels, each wave all four 32-row blocks; three
// v_mfma_f32_32x32x16_f16 per product (lh, hl, hh), fp32 accumulate, exact un-scaling, then the shared epilogue (bias, ReLU,
// NHWC / TBC / transposed-conv scatter, producer-carried abs-max).  Persistent grid, two workgroups per CU.
//
// Roofline: MFMA-bound for K >= 256 (fp16 dense / 3 = 838.9 TFLOP/s fp32-equivalent); HBM-bound for the K = 64 / 128 transposed
// convs (input once + output once).  Algorithmic bytes per launch: 4 (M K + N K + M N).
#include "conv_args.h"

using qea_conv::ConvArgs;
using qea_conv::conv_epilogue;

namespace {

// WM waves down the 128 rows x 4 / WM waves across the columns; each wave MI = 4 / WM row blocks x NJ column blocks of 32
// CIN = channels staged per chunk (64, or 128: half the barriers per MFMA, twice the bytes in flight, two workgroups per CU)
constexpr int gemm1x1_wgs(int cin, int wm, int nj) { return (cin == 64 && (4 / wm) * nj <= 4) ? 3 : 2; }

template <int CIN, int WM, int NJ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(gemm1x1_wgs(CIN, WM, NJ), gemm1x1_wgs(CIN, WM, NJ)))) void gemm1x1_f16_kernel(const ConvArgs p, const _Float16* __restrict__ wf,
                                                                                                                            int chunks, int total) {
  constexpr int BM = 128, KS = CIN / 16, MI = 4 / WM, WN = 4 / WM;
  constexpr int C4 = CIN / 4, RP = 256 / C4, NLD = BM / RP;   // float4 per row, rows per gather pass, passes
  constexpr int NB = WN * NJ * 32;                         // output channels per work item; wave column wn takes [wn*32*NJ, +32*NJ)
  constexpr int PLANE = BM * CIN;                          // fp16 elements per plane
  extern __shared__ __attribute__((aligned(16))) float smem[];
  _Float16* As = reinterpret_cast<_Float16*>(smem);        // [2][BM][CIN]
  int* rowpix = reinterpret_cast<int*>(As + 2 * PLANE);    // [2][BM]: output pixel row of each tile row (-1 past M), per item parity
  float sx, inv_x;
  qea_f16_scale(p.xmax[0], sx, inv_x);
  const float inv_w = reinterpret_cast<const float*>(wf + (size_t)p.N * p.K * 2)[0];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int nblk = p.N / NB;
  const int c4 = tid % C4, grow = tid / C4;                // gather: rows grow + RP i, float4 number c4 of the chunk
  // LDS element offsets.  CIN 64: rows of 128 B, 16-byte slot s of row r stored at s ^ ((r >> 1) & 7) (two rows share a 256-byte bank
  // row); CIN 128: rows of 256 B, slot s at s ^ (r & 15).  The key of row grow + RP i: the same for every i (CIN 64), two values (128).
  auto key = [](int r) { return CIN == 64 ? ((r >> 1) & 7) : (r & 15); };
  const int st_off0 = grow * CIN + (((c4 >> 1) ^ key(grow)) << 3) + (c4 & 1) * 4;
  const int st_off1 = (grow + RP) * CIN + (((c4 >> 1) ^ key(grow + RP)) << 3) + (c4 & 1) * 4;
  int a_off[KS];
#pragma unroll
  for (int cs = 0; cs < KS; ++cs) a_off[cs] = (wm * MI * 32 + fr) * CIN + (((cs * 2 + fh) ^ key(fr)) << 3);   // + 32 i rows (same key)
  const float* xg = p.x + (size_t)grow * p.ldx + c4 * 4;

  f32x4 hv[NLD];
  auto gather = [&](int tile_m, int chunk) {
    const int r0 = tile_m * BM + grow;
    const float* src = xg + (size_t)tile_m * BM * p.ldx + chunk * CIN;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const bool ok = r0 + RP * i < p.M;
      const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? src + (size_t)(RP * i) * p.ldx : p.x);
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      hv[i] = ok ? v : zero;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      f16x4 h, l;
      qea_split2_f16(hv[i], sx, h, l);
      const int o = ((i & 1) ? st_off1 : st_off0) + (i >> 1) * 2 * RP * CIN;
      *reinterpret_cast<f16x4*>(As + o) = h;
      *reinterpret_cast<f16x4*>(As + PLANE + o) = l;
    }
  };

  // filter fragments: wf[128-column block][chunk][cs][plane][nj4][lane][8]
  f16x8 bq[2][2][NJ];
  auto load_b = [&](int nb, int gst, int buf) {            // gst = chunk * KS + cs
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col32 = nb * (NB / 32) + wn * NJ + j;      // 32-column block index in [0, N / 32)
      const f16x8* wl = reinterpret_cast<const f16x8*>(wf) + ((size_t)(col32 >> 2) * chunks * KS * 2 * 4 + (col32 & 3)) * 64 + lane;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) bq[buf][pl][j] = wl[(size_t)(gst * 2 + pl) * 4 * 64];
    }
  };
  // output pixel row of tile row `tid` (threads 0..127): plain row, (t, b) -> (b, t) transpose, or the even-even pixel of the 2x2 scatter
  auto write_rowpix = [&](int tile_m, int par) {
    if (tid < BM) {
      const int m = tile_m * BM + tid;
      int v = -1;
      if (m < p.M) {
        if (p.out_mode == QEA_OUT_NHWC) {
          v = m;
        } else if (p.out_mode == QEA_OUT_TBC) {
          const int b = m / p.OW;
          v = (m - b * p.OW) * p.B + b;
        } else {
          const int ohw = p.OH * p.OW;
          const int cb = m / ohw, rem = m - cb * ohw;
          const int ch = rem / p.OW, cw = rem - ch * p.OW;
          v = (cb * (2 * p.OH) + 2 * ch) * (2 * p.OW) + 2 * cw;
        }
      }
      rowpix[par * BM + tid] = v;
    }
  };

  int vb = blockIdx.x;
  auto decode = [&](int v, int& nb, int& tile_m) {
    const int lid = qea_xcd_swizzle(v, total);
    nb = lid % nblk;
    tile_m = lid / nblk;
  };
  int cur_nb, cur_tm, par = 0;
  decode(vb, cur_nb, cur_tm);
  gather(cur_tm, 0);
  load_b(cur_nb, 0, 0);
  bool first = true;
  float am = 0.f;
  while (true) {
    const int nvb = vb + gridDim.x;
    const bool has_next = nvb < total;
    int nxt_nb, nxt_tm;
    decode(has_next ? nvb : vb, nxt_nb, nxt_tm);
    write_rowpix(cur_tm, par);                            // read after the barriers of the chunk loop below
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int chunk = 0; chunk < chunks; ++chunk) {
      if (!first) __syncthreads();                        // every wave has read the previous planes
      first = false;
      stage();
      __syncthreads();
      if (chunk + 1 < chunks) gather(cur_tm, chunk + 1);  // in flight under the MFMAs below
      else if (has_next) gather(nxt_tm, 0);
      auto read_a = [&](int cs, int i, f16x8* a) {
        const _Float16* src = As + a_off[cs] + i * 32 * CIN;
        a[0] = *reinterpret_cast<const f16x8*>(src);
        a[1] = *reinterpret_cast<const f16x8*>(src + PLANE);
      };
      f16x8 ar[2][2];
      read_a(0, 0, ar[0]);
#pragma unroll
      for (int cs = 0; cs < KS; ++cs) {
        const int cb = cs & 1;                            // KS is even: the buffer parity carries over chunks and items
        if (cs + 1 < KS || chunk + 1 < chunks) load_b(cur_nb, chunk * KS + cs + 1, cb ^ 1);
        else if (has_next) load_b(nxt_nb, 0, cb ^ 1);
        __builtin_amdgcn_sched_barrier(0);                // the next step's filter loads stay AHEAD of this step's MFMAs
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int f = cs * MI + i;
          const f16x8* a = ar[f & 1];
          const bool more = f + 1 < KS * MI;
          if (more) read_a((f + 1) / MI, (f + 1) % MI, ar[(f + 1) & 1]);
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            // smallest terms first (ll is dropped): lh, hl, hh
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], bq[cb][0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bq[cb][1][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bq[cb][0][j], acc[i][j], 0, 0, 0);
          }
          if (more) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);        // the two LDS reads of row f + 1 ...
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * NJ, 0);   // ... ahead of the MFMAs of row f
          }
        }
      }
    }

    // ---- epilogue: v = acc / (s_x s_w) + bias; ReLU; store.  Accumulator row = (r & 3) + 8 (r >> 2) + 4 fh of block i, column fr.
    // The option tests sit outside the element loops and the accumulators are finished in place (as in the LDS-halo conv).
    const int* rp = rowpix + par * BM;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = cur_nb * NB + (wn * NJ + j) * 32 + fr;
      int nbias = n, dpix = 0, ncol = n;
      if (p.out_mode == QEA_OUT_CONVT) {                   // column n = (a, b, c): output pixel (2h + a, 2w + b), channel c
        const int co_n = p.N >> 2;
        const int ab = n / co_n;
        ncol = nbias = n - ab * co_n;
        dpix = (ab >> 1) * (2 * p.OW) + (ab & 1);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] * inv_x) * inv_w;   // un-scale: exact (powers of two), one factor at a time
      if (p.bias) {
        const float ebi = p.bias[nbias];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += ebi;
      }
      if (p.relu) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = fmaxf(acc[i][j][r], 0.f);
      }
      float* yc = p.y + (size_t)dpix * p.ldy + ncol;       // this lane's column (+ the scatter's pixel shift)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int pix = rp[(wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh];
          if (pix < 0) continue;                          // row past M
          const float v = acc[i][j][r];
          yc[(size_t)pix * p.ldy] = v;
          am = qea_amax_acc(am, v);
        }
      }
    }
    if (!has_next) break;
    cur_nb = nxt_nb;
    cur_tm = nxt_tm;
    vb = nvb;
    par ^= 1;
  }
  qea_amax_commit_block(am, p.yamax);                      // once per workgroup, after its last item (not an L2 round trip per item)
}

template <int CIN, int WM, int NJ>
int launch_(const ConvArgs& a, hipStream_t s) {
  constexpr int NB = (4 / WM) * NJ * 32;
  constexpr size_t lds = (size_t)2 * 128 * CIN * 2 + 2 * 128 * sizeof(int);
  auto kern = gemm1x1_f16_kernel<CIN, WM, NJ>;
  static int attr_rc = (int)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr_rc != (int)hipSuccess) {
    qea_set_error("qea_conv_igemm(1x1 tile): cannot reserve %zu bytes of LDS: %s", (size_t)lds, hipGetErrorString((hipError_t)attr_rc));
    return QEA_ERR_LAUNCH;
  }
  const long long total = (long long)qea_cdiv(a.M, 128) * (a.N / NB);
  if (total <= 0 || total > 0x7fffffffLL) {
    qea_set_error("qea_conv_igemm(1x1 tile): grid %lld out of range", total);
    return QEA_ERR_INVALID;
  }
  static const int resident = [] {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return gemm1x1_wgs(CIN, WM, NJ) * (cus & ~7);
  }();
  const unsigned grid = total > resident ? (unsigned)resident : (unsigned)total;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a, (const _Float16*)a.wp, a.K / CIN, (int)total);
  return QEA_OK;
}

// w [N][K] fp32 -> [128-column block][chunk][cs][plane h, l][nj4][lane][8 fp16] of the filter scaled by s_w (qea_f16_scale of `wmax`),
// followed, at element offset N * K * 2, by one float: 1 / s_w.  Lane (n = 32 nj + (lane & 31), half = lane >> 5) holds channels
// 64 chunk + 16 cs + 8 half + 0..7 of filter row n.
__global__ void pack_frag_planes_f16_1x1_kernel(const float* __restrict__ w, _Float16* __restrict__ dst, int N, int K, const float* __restrict__ wmax) {
  const int chunks = K / 64;
  float sw, inv;
  qea_f16_scale(wmax[0], sw, inv);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;       // (column block, chunk, cs, nj, lane)
  if (i == 0) reinterpret_cast<float*>(dst + (size_t)N * K * 2)[0] = inv;
  if (i >= (N / 128) * chunks * 4 * 4 * 64) return;
  const int lane = i & 63;
  const int nj = (i >> 6) & 3;
  const int gst = i >> 8;                                    // (column block, chunk, cs) flattened
  const int nbk = gst / (chunks * 4);
  const int chunk = (gst >> 2) % chunks, cs = gst & 3;
  const int n = nbk * 128 + nj * 32 + (lane & 31);
  const float* src = w + (size_t)n * K + chunk * 64 + cs * 16 + 8 * (lane >> 5);
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
  for (int q = 0; q < 2; ++q) *reinterpret_cast<f16x8*>(dst + ((((size_t)gst * 2 + q) * 4 + nj) * 64 + lane) * 8) = pl[q];
}

}  // namespace

int qea_conv::launch_gemm1x1_f16(const ConvArgs& a, hipStream_t s) {
  // 256 columns per item halve the split work per MFMA; taken when the launch still has enough items to fill the card twice.
  // (measured, kernel time at M = 63488: N 1024 K 512 282.6 us against 291.7 for 128 columns and 314.9 for the generic 128x256 tile;
  //  N 512 K 2048 488.2 / 533.4 / 521.1; 128-channel chunks and a 2 x 2 wave layout were slower: profiles/r03_gemm1x1_experiment.txt)
  if (a.N % 256 == 0 && (long long)qea_cdiv(a.M, 128) * (a.N / 256) >= 1024) return launch_<64, 1, 2>(a, s);
  return launch_<64, 1, 1>(a, s);
}

extern "C" size_t qea_pack_frag_planes_f16_1x1_bytes(int32_t N, int32_t K) { return (size_t)N * K * 4 + 16; }

extern "C" int qea_pack_frag_planes_f16_1x1(const float* w, int32_t N, int32_t K, const float* wmax, void* planes, void* stream) {
  QEA_REQUIRE(w && planes && wmax && N > 0 && N % 128 == 0 && K > 0 && K % 64 == 0 && (long long)N * K * 4 < 0x7fffffffLL,
              "qea_pack_frag_planes_f16_1x1: N a multiple of 128, K a multiple of 64");
  QEA_REQUIRE(((uintptr_t)w & 15) == 0 && ((uintptr_t)planes & 15) == 0, "qea_pack_frag_planes_f16_1x1: pointers must be 16-byte aligned");
  const int total = (N / 128) * (K / 64) * 4 * 4 * 64;
  hipLaunchKernelGGL(pack_frag_planes_f16_1x1_kernel, dim3(qea_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, (_Float16*)planes, N, K, wmax);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}
