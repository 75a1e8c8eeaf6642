// Developer lab for the nine-tap LDS-halo weight gradient (two-way fp16 split, 64 x 64 channel blocks, 2 x 32 pixel tiles): the
// library's kernel (through qea_conv_wgrad) beside a PRODUCER / CONSUMER form on the same random data, interleaved rounds in one
// process.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/wgrad_lab.hip -Iinclude -Lquery-efficient-approx-to-improve-ocr_amd -lqea_hip \
//         -Wl,-rpath,/root/repo/query-efficient-approx-to-improve-ocr_amd -o tools/micro/wgrad_lab.bin
//   tools/micro/wgrad_lab.bin B H W R C
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <type_traits>
#include "qea_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e__ = (x);                                                             \
    if (e__ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__));      \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

__device__ __forceinline__ void split2_f16(const f32x4 v, float s, f16x4& h, f16x4& l) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float xs = v[k] * s;
    const _Float16 hk = (_Float16)xs;
    h[k] = hk;
    l[k] = (_Float16)(xs - (float)hk);
  }
}
__device__ __forceinline__ void f16_scale(float m, float& s, float& inv) {
  const unsigned E = (__float_as_uint(m) >> 23) & 0xffu;
  int se = 14 - ((int)E - 127);
  if (m == 0.f || E == 0) se = 0;
  se = se > 126 ? 126 : (se < -126 ? -126 : se);
  s = __uint_as_float((unsigned)(se + 127) << 23);
  inv = __uint_as_float((unsigned)(127 - se) << 23);
}
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, k = bid >> 3;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + k;
}
// two transposing reads: this lane's 4-pixel block and the block 4 pixel rows (512 bytes) further
__device__ __forceinline__ f16x8 tr_pair(const char* base) {
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + 4 * 128));
  return __builtin_shufflevector(__builtin_bit_cast(f16x4, lo), __builtin_bit_cast(f16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}

struct Plan { int tiles_x, tiles_y, n_tiles, r_blks, c_blks, splits; };

enum { V_PRIO = 1, V_STAMP = 2, V_NOPROD = 4 };
constexpr int SW = 32, TH = 2, HWD = SW + 2, HH = TH + 2, HP = HH * HWD;      // 64-pixel tile, 136 halo pixels
constexpr int P_PLANE_B = 64 * 64 * 2, Q_PLANE_B = HP * 64 * 2;              // bytes per plane
constexpr int BUF_B = 2 * P_PLANE_B + 2 * Q_PLANE_B;                          // one buffer: P h, P l, Q h, Q l

// Producer / consumer form: waves 0-3 only run MFMAs (one per SIMD: wave w and wave w + 4 share a SIMD), waves 4-7 gather, split and
// write the NEXT tile into the other LDS buffer while the consumers read this one; ONE barrier per tile.
template <int VAR>
__global__ __launch_bounds__(512) void wgrad_spec_kernel(const float* __restrict__ p, const float* __restrict__ q, float* __restrict__ ws, int B, int H, int W,
                                                         int R, int C, int ldp, int ldq, Plan hp, const float* __restrict__ pmax,
                                                         const float* __restrict__ qmax, unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float sp, sq, inv_p, inv_q;
  f16_scale(pmax[0], sp, inv_p);
  f16_scale(qmax[0], sq, inv_q);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  int bid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int c_blk = bid % hp.c_blks;
  bid /= hp.c_blks;
  const int r_blk = bid % hp.r_blks;
  const int split = bid / hp.r_blks;
  const int r0 = r_blk * 64, c0 = c_blk * 64;
  const int ntl = (hp.n_tiles - split + hp.splits - 1) / hp.splits;          // tiles of this workgroup: split, split + splits, ...

  if (producer) {
    const int pt = tid - 256;
    constexpr int NP = 64 * 16 / 256, NQ = (HP * 16 + 255) / 256;            // 4, 9 float4 per thread
    f32x4 preg[NP], qreg[NQ];
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
    auto row_off = [](int pix, int c4) { return pix * 64 + ((((c4 >> 3) ^ (pix >> 1)) & 1) << 5) + (c4 & 7) * 4; };   // elements
    auto stage = [&](char* buf) {
      _Float16* Ps = reinterpret_cast<_Float16*>(buf);
      _Float16* Qs = reinterpret_cast<_Float16*>(buf + 2 * P_PLANE_B);
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int e = pt + 256 * i;
        const int o = row_off(e / 16, e % 16);
        f16x4 h, l;
        split2_f16(preg[i], sp, h, l);
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
          split2_f16(qreg[i], sq, h, l);
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
      if ((VAR & V_NOPROD) == 0 || t == 0) {
        if (t + 1 < ntl) stage(lds + ((t + 1) & 1) * BUF_B);
        if (t + 2 < ntl) fetch(split + (t + 2) * hp.splits);
      }
      __syncthreads();                                       // consumers done with buffer t & 1, buffer (t + 1) & 1 complete
    }
    return;
  }

  // ---------------------------------------------------------------- consumers
  if constexpr ((VAR & V_PRIO) != 0) __builtin_amdgcn_s_setprio(1);
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
  // Q fragment at halo pixel c + l_pix: the chunk swap follows the parity of (c + l_pix) >> 1 -> four bases by (c & 1, (c >> 1) & 1)
  int TQ[2][2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int hs = (par ? (l_pix + 1) >> 1 : l_pix >> 1) + b;
      TQ[par][b] = l_pix * 128 + ((((wc ^ hs) & 1) << 5) + l_ch) * 2;
    }
  __syncthreads();                                           // tile 0 staged
  unsigned long long t_wait = 0, t_loop = 0, t0 = 0;
  for (int t = 0; t < ntl; ++t) {
    const char* buf = lds + (t & 1) * BUF_B;
    const char* Pb = buf + p_off;
    const char* Qb = buf + 2 * P_PLANE_B;
    auto read_p = [&](int ks, f16x8* af) {
      af[0] = tr_pair(Pb + ks * 16 * 128);
      af[1] = tr_pair(Pb + P_PLANE_B + ks * 16 * 128);
    };
    auto read_q = [&](int f, f16x8* bf) {                   // f = ks * 9 + tap
      const int ks = f / 9, tap = f % 9;
      const int py = (ks * 16) / SW, px0 = (ks * 16) % SW;
      const int c = (py + tap / 3) * HWD + px0 + tap % 3;
      const char* src = Qb + TQ[c & 1][(c >> 1) & 1] + c * 128;
      bf[0] = tr_pair(src);
      bf[1] = tr_pair(src + Q_PLANE_B);
    };
    if constexpr ((VAR & V_STAMP) != 0) t0 = __builtin_amdgcn_s_memtime();
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
    if constexpr ((VAR & V_STAMP) != 0) {
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      t_loop += t1 - t0;
      __syncthreads();
      t_wait += __builtin_amdgcn_s_memtime() - t1;
    } else {
      __syncthreads();
    }
  }
  if constexpr ((VAR & V_STAMP) != 0) {
    if (lane == 0 && wave == 0) {
      stamps[blockIdx.x * 4 + 0] = t_loop;
      stamps[blockIdx.x * 4 + 1] = t_wait;
      stamps[blockIdx.x * 4 + 2] = (unsigned long long)ntl;
    }
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

// ---- RING variant: a workgroup walks whole column strips (all the 2-row tiles of one image column, top to bottom) and keeps the X halo
// in a RING of 8 pixel rows: a tile's halo is ring rows g .. g + 3, the next tile of the strip needs rows g + 2 .. g + 5, i.e. only TWO new
// rows (the first tile of the next strip: four, at g + 4 .. g + 7) — the producers gather and split 68 instead of 136 halo pixels per tile.
constexpr int RING_ROWS = 8;
constexpr int QR_PLANE_B = RING_ROWS * HWD * 64 * 2;                          // bytes per plane of the ring
constexpr int RING_LDS = 2 * (2 * P_PLANE_B) + 2 * QR_PLANE_B;                // two P buffers (h, l each) + the ring (h, l)

template <int VAR>
__global__ __launch_bounds__(512) void wgrad_ring_kernel(const float* __restrict__ p, const float* __restrict__ q, float* __restrict__ ws, int B, int H, int W,
                                                         int R, int C, int ldp, int ldq, Plan hp, const float* __restrict__ pmax,
                                                         const float* __restrict__ qmax) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const Pbase = lds;                                                    // [2 buffers][h, l][64 px][64 ch]
  char* const Qbase = lds + 2 * (2 * P_PLANE_B);                              // [h, l][8 ring rows][34][64 ch]
  float sp, sq, inv_p, inv_q;
  f16_scale(pmax[0], sp, inv_p);
  f16_scale(qmax[0], sq, inv_q);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  int bid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int c_blk = bid % hp.c_blks;
  bid /= hp.c_blks;
  const int r_blk = bid % hp.r_blks;
  const int split = bid / hp.r_blks;
  const int r0 = r_blk * 64, c0 = c_blk * 64;
  // strips of this workgroup: a contiguous range of (image, column tile) pairs
  const int n_strips = B * hp.tiles_x;
  const int s0 = (int)((long long)n_strips * split / hp.splits), s1 = (int)((long long)n_strips * (split + 1) / hp.splits);
  const int ntl = (s1 - s0) * hp.tiles_y;                                     // tiles, in (strip, ty) order
  auto tile_of = [&](int t, int& b, int& x0, int& y0, bool& first) {
    const int st = s0 + t / hp.tiles_y, ty = t % hp.tiles_y;
    b = st / hp.tiles_x;
    x0 = (st % hp.tiles_x) * SW;
    y0 = ty * TH;
    first = ty == 0;
  };
  // ring base (first halo row) of tile t: + 2 per tile inside a strip, + 4 across strips -> ((t + t / tiles_y) * 2) & 7
  auto ring_of = [&](int t) { return ((t + t / hp.tiles_y) * 2) & 7; };

  if (producer) {
    const int pt = tid - 256;
    constexpr int NP = 64 * 16 / 256, NQ = (4 * HWD * 16 + 255) / 256;        // 4, 9 (a whole four-row halo at a strip start)
    f32x4 preg[NP], qreg[NQ];
    int q_rows = 0, q_ring0 = 0;                                              // what qreg holds: rows of the halo and their first ring row
    auto fetch = [&](int t) {
      int b, x0, y0;
      bool first;
      tile_of(t, b, x0, y0, first);
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int e = pt + 256 * i;
        const int c4 = e % 16, pix = e / 16;
        const int py = pix / SW, px = pix - py * SW;
        preg[i] = *reinterpret_cast<const f32x4*>(p + ((size_t)(b * H + y0 + py) * W + x0 + px) * ldp + r0 + c4 * 4);
      }
      // halo rows to bring: all four (hy = 0 .. 3) for the first tile of a strip, else the two new ones (hy = 2, 3)
      const int hy0 = first ? 0 : 2;
      q_rows = 4 - hy0;
      q_ring0 = (ring_of(t) + hy0) & 7;
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int e = pt + 256 * i;
        const int c4 = e % 16, hq = e / 16;                                   // pixel of the rows being fetched
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (hq < q_rows * HWD) {
          const int hy = hy0 + hq / HWD, hx = hq % HWD;
          const int iy = y0 + hy - 1, ix = x0 + hx - 1;
          if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = *reinterpret_cast<const f32x4*>(q + ((size_t)(b * H + iy) * W + ix) * ldq + c0 + c4 * 4);
        }
        qreg[i] = v;
      }
    };
    auto row_off = [](int pix, int c4) { return pix * 64 + ((((c4 >> 3) ^ (pix >> 1)) & 1) << 5) + (c4 & 7) * 4; };   // elements
    auto stage = [&](int pbuf) {
      _Float16* Ps = reinterpret_cast<_Float16*>(Pbase + pbuf * (2 * P_PLANE_B));
      _Float16* Qs = reinterpret_cast<_Float16*>(Qbase);
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int e = pt + 256 * i;
        const int o = row_off(e / 16, e % 16);
        f16x4 h, l;
        split2_f16(preg[i], sp, h, l);
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
          split2_f16(qreg[i], sq, h, l);
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
      __syncthreads();
    }
    return;
  }

  // ---------------------------------------------------------------- consumers
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
    // the four halo rows of this tile sit at ring rows (g + k) & 7, g even: the chunk-swap parity of a tap's pixel offset only needs k & 1
    // (a row holds an even number of pixels), so a tap's address is one of 16 per-tile registers + an immediate column offset
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
    {
      auto read_p = [&](int ks, f16x8* af) {
        af[0] = tr_pair(Pb + ks * 16 * 128);
        af[1] = tr_pair(Pb + P_PLANE_B + ks * 16 * 128);
      };
      auto read_q = [&](int f, f16x8* bf) {                   // f = ks * 9 + tap
        const int ks = f / 9, tap = f % 9;
        const int py = (ks * 16) / SW, px0 = (ks * 16) % SW;
        const int k = py + tap / 3, col = px0 + tap % 3;
        const char* src = Qbase + QB[k][col & 1][((col >> 1) + k) & 1] + col * 128;
        bf[0] = tr_pair(src);
        bf[1] = tr_pair(src + QR_PLANE_B);
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

__global__ void reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, long long n, int splits) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float s = ws[i];
    for (int k = 1; k < splits; ++k) s += ws[i + (size_t)k * n];
    out[i] = s;
  }
}

__global__ void fill_normal(float* p, size_t n, unsigned seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned a = (unsigned)(i * 2654435761u) ^ seed, b = (unsigned)((i >> 32) * 40503u + i * 2246822519u) ^ (seed * 3266489917u);
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    b ^= b >> 16; b *= 0x7feb352du; b ^= b >> 15; b *= 0x846ca68bu; b ^= b >> 16;
    const float u1 = ((a >> 8) + 1) * (1.f / 16777217.f), u2 = (b >> 8) * (1.f / 16777216.f);
    p[i] = scale * sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
  }
}

struct Variant {
  const char* name;
  int var;
  void (*kern)(const float*, const float*, float*, int, int, int, int, int, int, int, Plan, const float*, const float*, unsigned long long*);
};
#define VARIANT(name, v) {name, v, wgrad_spec_kernel<v>}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 2048, H = argc > 2 ? atoi(argv[2]) : 8, W = argc > 3 ? atoi(argv[3]) : 32;
  const int R = argc > 4 ? atoi(argv[4]) : 256, C = argc > 5 ? atoi(argv[5]) : 256;
  const int rounds = argc > 6 ? atoi(argv[6]) : 5;
  if (R % 64 || C % 64 || W % 32 || H % 2) { fprintf(stderr, "shape not taken\n"); return 1; }
  const size_t M = (size_t)B * H * W;
  float *p, *q, *dwref, *dw, *pmax, *qmax;
  CK(hipMalloc(&p, M * R * 4));
  CK(hipMalloc(&q, M * C * 4));
  CK(hipMalloc(&dwref, (size_t)R * 9 * C * 4));
  CK(hipMalloc(&dw, (size_t)R * 9 * C * 4));
  CK(hipMalloc(&pmax, 4));
  CK(hipMalloc(&qmax, 4));
  hipLaunchKernelGGL(fill_normal, dim3(4096), dim3(256), 0, 0, p, M * R, 4242u, 0.01f);
  hipLaunchKernelGGL(fill_normal, dim3(4096), dim3(256), 0, 0, q, M * C, 99u, 1.0f);
  CK(hipDeviceSynchronize());
  if (qea_absmax(p, R, (int64_t)M, R, pmax, nullptr) || qea_absmax(q, C, (int64_t)M, C, qmax, nullptr)) { fprintf(stderr, "absmax: %s\n", qea_last_error()); return 1; }

  qea_wgrad_desc d;
  memset(&d, 0, sizeof(d));
  d.p = p; d.q = q; d.dw = dwref; d.B = B; d.PH = d.QH = H; d.PW = d.QW = W; d.R = R; d.C = C; d.KH = d.KW = 3; d.pad_h = d.pad_w = 1;
  d.stride_h = d.stride_w = 1; d.ldp = R; d.ldq = C; d.p_absmax = pmax; d.q_absmax = qmax;
  const size_t wsb = qea_conv_wgrad_workspace_bytes(&d);
  void* wsl;
  CK(hipMalloc(&wsl, wsb ? wsb : 16));
  d.workspace = wsl;
  d.workspace_bytes = wsb;
  if (qea_conv_wgrad(&d, nullptr)) { fprintf(stderr, "wgrad: %s\n", qea_last_error()); return 1; }
  CK(hipDeviceSynchronize());
  const size_t nd = (size_t)R * 9 * C;
  std::vector<float> href(nd), hy(nd);
  CK(hipMemcpy(href.data(), dwref, nd * 4, hipMemcpyDeviceToHost));

  Plan hp;
  hp.tiles_x = W / SW;
  hp.tiles_y = H / TH;
  hp.n_tiles = B * hp.tiles_x * hp.tiles_y;
  hp.r_blks = R / 64;
  hp.c_blks = C / 64;
  int cus = 256;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  int splits = cus / (hp.r_blks * hp.c_blks);                 // ONE workgroup (8 waves, 100 KB of LDS) per CU
  if (splits > hp.n_tiles / 8) splits = hp.n_tiles / 8;
  if (splits < 1) splits = 1;
  hp.splits = splits;
  const int grid = hp.r_blks * hp.c_blks * splits;
  float* ws;
  CK(hipMalloc(&ws, (size_t)splits * nd * 4));
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, (size_t)grid * 4 * 8));
  const size_t ldsb = 2 * BUF_B;

  std::vector<Variant> vs = {VARIANT("spec", 0), VARIANT("spec prio", V_PRIO), VARIANT("spec abl no producer work", V_NOPROD), VARIANT("spec stamp", V_STAMP),
                             {"ring", 4096, nullptr}};
  CK(hipFuncSetAttribute((const void*)wgrad_ring_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RING_LDS));
  for (auto& v : vs) if (v.kern) CK(hipFuncSetAttribute((const void*)v.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  auto launch = [&](const Variant& v) {
    if (v.var == 4096) {
      Plan hr = hp;
      if (hr.splits > B * hr.tiles_x) hr.splits = B * hr.tiles_x;
      hipLaunchKernelGGL(wgrad_ring_kernel<0>, dim3(hr.r_blks * hr.c_blks * hr.splits), dim3(512), (size_t)RING_LDS, 0, p, q, ws, B, H, W, R, C, R, C, hr, pmax, qmax);
      hipLaunchKernelGGL(reduce_kernel, dim3(std::min<long long>(2048, (nd + 255) / 256)), dim3(256), 0, 0, ws, dw, (long long)nd, hr.splits);
      return;
    }
    hipLaunchKernelGGL(v.kern, dim3(grid), dim3(512), ldsb, 0, p, q, ws, B, H, W, R, C, R, C, hp, pmax, qmax, stamps);
    hipLaunchKernelGGL(reduce_kernel, dim3(std::min<long long>(2048, (nd + 255) / 256)), dim3(256), 0, 0, ws, dw, (long long)nd, splits);
  };
  const double flops = 2.0 * M * R * 9.0 * C;
  printf("shape B%d H%d W%d R%d C%d  tiles %d  grid %d (splits %d)  %.1f GFLOP  lds %zu\n", B, H, W, R, C, hp.n_tiles, grid, splits, flops / 1e9, ldsb);
  for (auto& v : vs) {
    if (v.var & V_NOPROD) continue;
    CK(hipMemset(dw, 0xff, nd * 4));
    launch(v);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hy.data(), dw, nd * 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxr = 0;
    size_t nbit = 0;
    for (size_t i = 0; i < nd; ++i) {
      const double dd = fabs((double)hy[i] - href[i]);
      if (!(dd <= maxd)) maxd = dd;
      maxr = std::max(maxr, fabs((double)href[i]));
      nbit += memcmp(&hy[i], &href[i], 4) != 0;
    }
    printf("check %-28s max|d| %.3e (max|ref| %.3e, rel %.2e)  bit-different %zu of %zu\n", v.name, maxd, maxr, maxd / maxr, nbit, nd);
  }
  std::vector<std::vector<float>> ms(vs.size() + 1);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 4;
  for (int r = 0; r < rounds + 1; ++r) {
    for (size_t k = 0; k <= vs.size(); ++k) {
      CK(hipEventRecord(e0, 0));
      for (int qq = 0; qq < reps; ++qq) {
        if (k == vs.size()) qea_conv_wgrad(&d, nullptr);
        else launch(vs[k]);
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      if (r > 0) ms[k].push_back(t / reps);
    }
  }
  for (size_t k = 0; k <= vs.size(); ++k) {
    std::sort(ms[k].begin(), ms[k].end());
    const float med = ms[k][ms[k].size() / 2], mn = ms[k][0];
    printf("time  %-28s median %8.1f us  min %8.1f us   %7.1f TF (median, incl. slab reduction)\n", k == vs.size() ? "LIBRARY (auto tile)" : vs[k].name, med * 1e3,
           mn * 1e3, flops / med / 1e9);
  }
  for (auto& v : vs) {
    if (!(v.var & V_STAMP)) continue;
    CK(hipMemset(stamps, 0, (size_t)grid * 4 * 8));
    launch(v);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hs((size_t)grid * 4);
    CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> lp, wt;
    for (int g = 0; g < grid; ++g)
      if (hs[g * 4 + 2]) {
        lp.push_back((double)hs[g * 4] / hs[g * 4 + 2]);
        wt.push_back((double)hs[g * 4 + 1] / hs[g * 4 + 2]);
      }
    std::sort(lp.begin(), lp.end());
    std::sort(wt.begin(), wt.end());
    printf("stamps %s: consumer cycles per tile: MFMA loop median %.0f (108 MFMAs = 3456), barrier wait median %.0f\n", v.name, lp[lp.size() / 2], wt[wt.size() / 2]);
  }
  return 0;
}
