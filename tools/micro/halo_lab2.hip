// Developer lab (round 4): the LDS-halo 3x3 conv as a CONSUMER OF PRODUCER-WRITTEN fp16 PLANES.  The activation tensor arrives as two
// NHWC fp16 planes h, l of x * s (what a producer's epilogue would store next to / instead of the fp32 tensor); the halo of a
// 32-channel sub-chunk goes global -> LDS by LDS-DMA (no registers, no VALU split, no staging phase), double-buffered, one barrier per
// sub-chunk.  Compared on the same random data with the library's tile 24 (which gathers fp32, splits on the fly and stages through
// registers), interleaved rounds in ONE process.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/halo_lab2.hip -Iinclude -Lquery-efficient-approx-to-improve-ocr_amd -lqea_hip \
//         -Wl,-rpath,/root/repo/query-efficient-approx-to-improve-ocr_amd -o tools/micro/halo_lab2.bin
//   tools/micro/halo_lab2.bin B H W Cin Cout [rounds]
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
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glob_void;

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

__device__ __forceinline__ void dma16_asm(const void* base, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
}
enum { V_WG3 = 1, V_NOB = 2, V_NOEPI = 4, V_NODMA = 8, V_TH8 = 16, V_DMAFIRST = 32, V_N256 = 64, V_DMASPREAD = 128, V_ASM = 256, V_BCONST = 512, V_DMACONST = 1024, V_NB3 = 2048, V_WG1 = 4096 };
// V_NB3: three filter-fragment buffers (loads two taps ahead).  V_WG1: the same kernel launched with ONE workgroup per CU (one wave per SIMD):
// its MFMA-busy is the share of a wave's own time that is MFMA issue.
// V_BCONST / V_DMACONST: the filter loads / the DMA pieces are ISSUED but always read the same few KiB (L1 / L2 hits): separates what the
// instruction stream costs from what the bytes cost.  (V_NODMA keeps the very first fill of both buffers: random data under the MFMAs.)
// V_ASM: the DMA instruction as inline asm — hipcc puts an s_waitcnt vmcnt(0) behind every second global_load_lds builtin (it cannot tell the
// in-flight piece from the LDS reads that follow), which exposes the latency of the filter loads just issued
// V_TH8: 8 x 32 pixel tile, EIGHT waves (two wave rows of four), one workgroup per CU: the two wave rows read the same filter fragments (the second from L1)
// and the halo overlap drops from 1.59 to 1.33.  V_N256: a wave owns 64 output channels (four 16-channel groups), the workgroup 256: half the halo
// traffic and half the LDS fragment reads per MFMA.

// ---- the consumer.  Tile = TH x 32 pixels x 128 output channels, 4 waves side by side over the channels (32 each), every wave all TH rows.
// LDS: [buffer 2][plane 2][HPA pixels][32 channels] fp16 (64-byte pixel rows; the four 16-byte slots of pixel row p rotated by p >> 1:
// conflict-free ds_read_b128 for the 16x16x32 operand layout under every tap offset).
template <int VAR>
__global__ __launch_bounds__((VAR & V_TH8) ? 512 : 256) __attribute__((amdgpu_waves_per_eu((VAR & V_WG3) ? 3 : 2, (VAR & V_WG3) ? 3 : 2))) void planes_kernel(
    const _Float16* __restrict__ xh, const _Float16* __restrict__ xl, const _Float16* __restrict__ wf, float* __restrict__ y, int B, int H, int W,
    int C, int ldy, int nsc, int Ntot, int total, const float* __restrict__ xmax, unsigned zero_off) {
  constexpr bool NOB = (VAR & V_NOB) != 0, NOEPI = (VAR & V_NOEPI) != 0, NODMA = (VAR & V_NODMA) != 0;
  constexpr int TH = (VAR & V_TH8) ? 8 : 4, TW = 32, HWD = 34, HH = TH + 2, HP = HH * HWD, NPC = (HP + 15) / 16, HPA = NPC * 16;
  constexpr int NJ = (VAR & V_N256) ? 4 : 2, COUT = 64 * NJ, WN = 4, NWAVE = (VAR & V_TH8) ? 8 : 4, MI = 4, NG = COUT / 16;
  constexpr int ROWB = 64, PLANE_B = HPA * ROWB, BUF_B = 2 * PLANE_B;
  constexpr int NPW = (NPC + NWAVE - 1) / NWAVE;                      // pieces per wave and plane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float sx, inv_x;
  f16_scale(xmax[0], sx, inv_x);
  const float inv_w = reinterpret_cast<const float*>(wf + (size_t)Ntot * 9 * nsc * 32 * 2)[0];

  const int tiles_x = W / TW, tiles_y = H / TH;
  const int nblk = Ntot / COUT;
  struct Item { int nb, b, x0, y0; };
  auto decode = [&](int vb) {
    const int lid = xcd_swizzle(vb, total);
    Item it;
    it.nb = lid % nblk;
    int bid = lid / nblk;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    it.b = bid / tiles_y;
    it.x0 = tx * TW;
    it.y0 = ty * TH;
    return it;
  };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave % WN, wm = wave / WN;
  // DMA source offsets (bytes into a plane) of this lane for the pieces j = wave + 4 i of an item: piece j covers halo pixels 16 j ... 16 j + 15,
  // lane -> pixel 16 j + (lane >> 2), physical slot lane & 3 = logical slot rotated by pixel >> 1
  unsigned off[NPW], offn[NPW], off0[NPW];
  auto offsets = [&](const Item& it, unsigned* off) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int j = min(wave + NWAVE * i, NPC - 1);     // (a piece past the last is the last once more: same bytes to the same place, no branch)
      const int q = j * 16 + (lane >> 2);
      const int hy = q / HWD, hx = q - hy * HWD;
      const int iy = it.y0 + hy - 1, ix = it.x0 + hx - 1;
      const bool ok = q < HP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const int logical = ((lane & 3) - (q >> 1)) & 3;
      off[i] = (ok ? (unsigned)(((it.b * H + iy) * W + ix) * C * 2) : zero_off) + logical * 16;
    }
  };
  auto dma = [&](int sc, int buf, bool init = false) {
    if (NODMA && !init) return;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int j = min(wave + NWAVE * i, NPC - 1);
      {
        char* dst = smem + buf * BUF_B + j * 1024;
        const unsigned o = (VAR & V_DMACONST) ? off0[i] : off[i] + sc * 64;
        __builtin_amdgcn_global_load_lds((glob_void*)(reinterpret_cast<const char*>(xh) + o), (lds_void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glob_void*)(reinterpret_cast<const char*>(xl) + o), (lds_void*)(dst + PLANE_B), 16, 0, 0);
      }
    }
  };

  // one DMA instruction: number k = piece i = k >> 1 of this wave, plane k & 1
  auto dma1 = [&](int sc, int buf, int k) {
    if (NODMA) return;
    const int i = k >> 1, pl_ = k & 1;
    if (i < NPW) {                                        // (compile time)
      const int j = min(wave + NWAVE * i, NPC - 1);
      char* dst = smem + buf * BUF_B + j * 1024 + pl_ * PLANE_B;
      const unsigned o = (VAR & V_DMACONST) ? off0[i] : (sc < 0 ? offn[i] : off[i] + sc * 64);   // sc < 0: the next item's first sub-chunk
      if constexpr ((VAR & V_ASM) != 0) {
        const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)dst);
        dma16_asm(pl_ ? xl : xh, o, la);
      } else {
        __builtin_amdgcn_global_load_lds((glob_void*)(reinterpret_cast<const char*>(pl_ ? xl : xh) + o), (lds_void*)dst, 16, 0, 0);
      }
    }
  };
  // A-fragment addresses: lane (pixel p16 of a 16-pixel group, 8-channel slot g4), pixel offset c (compile time):
  // byte = (p16 + c) * 64 + ((g4 + ((p16 + c) >> 1)) & 3) * 16 = T[c & 1][(c >> 1) & 3] + c * 64
  const int p16 = lane & 15, g4 = lane >> 4;
  int T[2][4];
#pragma unroll
  for (int par = 0; par < 2; ++par)
#pragma unroll
    for (int k = 0; k < 4; ++k) T[par][k] = p16 * ROWB + (((k + g4 + (p16 >> 1) + (par & p16 & 1)) & 3) << 4);

  constexpr int NBQ = (VAR & V_NB3) ? 3 : 2;
  f16x8 bq[NBQ][NJ][2];                                      // [buffer][channel group of the wave][plane]
  auto load_b = [&](int nb, int gst, int buf) {
    const f16x8* wl = reinterpret_cast<const f16x8*>(wf) + (size_t)nb * nsc * 9 * 2 * NG * 64 + (wn * NJ) * 64 + lane;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int g2 = 0; g2 < NJ; ++g2) bq[buf][g2][pl] = wl[(size_t)(((((VAR & V_BCONST) ? (gst & 1) : gst)) * 2 + pl) * NG + g2) * 64];
  };

  int vb = blockIdx.x;
  Item cur = decode(vb);
  offsets(cur, off);
  {
    Item z;
    z.nb = 0; z.b = (blockIdx.x & 7); z.x0 = 0; z.y0 = 0;
    offsets(z, off0);
  }
  dma(0, 0, true);
  if (NODMA) dma(1, 1, true);
  load_b(cur.nb, 0, 0);
  if constexpr (NBQ == 3) load_b(cur.nb, 1, 1);
  while (true) {
    const int nvb = vb + gridDim.x;
    const bool has_next = nvb < total;
    const Item nxt = decode(has_next ? nvb : vb);
    offsets(nxt, offn);
    f32x4 acc[MI][2][NJ];                                  // [tile row][half row][channel group]
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int xh_ = 0; xh_ < 2; ++xh_)
#pragma unroll
        for (int g2 = 0; g2 < NJ; ++g2)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][xh_][g2][r] = 0.f;
    for (int sc2 = 0; sc2 < nsc; sc2 += 2) {
      // two sub-chunks per trip: nine taps flip the parity of the filter buffer, and the LDS buffer becomes a compile-time constant
      auto half = [&](auto par_) {
      constexpr int PAR = decltype(par_)::value;
      const int sc = sc2 + PAR;
      constexpr int gbuf = PAR;
      // the DMA of this sub-chunk (issued one sub-chunk ago) has landed for every wave, and every wave is done reading the other buffer
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // vmcnt retires in order: the first filter load issued AFTER the DMA pieces cannot be waited for without waiting for them too, so
      // the pieces go out behind tap 0's filter load (they then have two taps of MFMAs to land, not one)
      auto next_dma = [&]() {
        if (sc + 1 < nsc) dma(sc + 1, gbuf ^ 1);
        else if (has_next) {
#pragma unroll
          for (int i = 0; i < NPW; ++i) off[i] = offn[i];
          dma(0, gbuf ^ 1);
        }
      };
      if constexpr ((VAR & V_DMAFIRST) != 0) next_dma();
      const char* As = smem + gbuf * BUF_B;
      constexpr int GR = MI * 2;
      auto read_a = [&](int tap, int g, f16x8* a) {
        const int kh = tap / 3, kw = tap % 3;
        const int i = g / 2, xh_ = g % 2;
        const int c = (i + kh) * HWD + kw + xh_ * 16;
        const char* Aw = As + wm * (MI * HWD * ROWB);
        const char* src = Aw + T[c & 1][(c >> 1) & 3] + c * ROWB;
        a[0] = *reinterpret_cast<const f16x8*>(src);
        a[1] = *reinterpret_cast<const f16x8*>(src + PLANE_B);
      };
      f16x8 ar[2][2];
      read_a(0, 0, ar[0]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int cb = NBQ == 3 ? tap % 3 : (tap + PAR) & 1;
        if (!NOB) {
          constexpr int D = NBQ - 1;
          const int nbuf = NBQ == 3 ? (tap + 2) % 3 : cb ^ 1;
          if (tap + D < 9 || sc + 1 < nsc) load_b(cur.nb, sc * 9 + tap + D, nbuf);
          else if (has_next) load_b(nxt.nb, tap + D - 9, nbuf);
        }
        if constexpr ((VAR & V_DMASPREAD) != 0) {
          // one DMA instruction per tap (vmcnt retires in order: every piece then has two taps to land and none of the filter waits
          // sees more than one piece in front of it)
          static_assert(2 * NPW <= 8, "eight taps carry the pieces");
          if constexpr ((VAR & V_ASM) == 0) {
            if (tap < 8) dma1(sc + 1 < nsc ? sc + 1 : -1, gbuf ^ 1, tap);   // (no next item: this item's first sub-chunk once more, unused)
          }
        } else if constexpr ((VAR & V_DMAFIRST) == 0) {
          if (tap == 0) next_dma();
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < GR; ++g) {
          const int f = tap * GR + g;
          const f16x8* a = ar[f & 1];
          const bool more = f + 1 < 9 * GR;
          if (more) read_a((f + 1) / GR, (f + 1) % GR, ar[(f + 1) & 1]);
          const int bb = NOB ? 0 : cb;
          const int i = g / 2, xh_ = g % 2;
          if constexpr ((VAR & V_ASM) != 0 && (VAR & V_DMASPREAD) != 0) {
            // hipcc does not count this instruction: it goes out in the MIDDLE of the tap, behind the waits for this tap's filter fragments, so that
            // it is never among the youngest operations a counted wait lets stay in flight
            if (g == GR / 2 && tap < 8) dma1(sc + 1 < nsc ? sc + 1 : -1, gbuf ^ 1, tap);
          }
#pragma unroll
          for (int g2 = 0; g2 < NJ; ++g2) acc[i][xh_][g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[bb][g2][0], a[1], acc[i][xh_][g2], 0, 0, 0);
#pragma unroll
          for (int g2 = 0; g2 < NJ; ++g2) acc[i][xh_][g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[bb][g2][1], a[0], acc[i][xh_][g2], 0, 0, 0);
#pragma unroll
          for (int g2 = 0; g2 < NJ; ++g2) acc[i][xh_][g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[bb][g2][0], a[0], acc[i][xh_][g2], 0, 0, 0);
          if (more) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
          }
        }
      }
      };
      half(std::integral_constant<int, 0>{});
      half(std::integral_constant<int, 1>{});
    }
    // epilogue: filter = A operand (rows = channels), pixels = B operand: a lane's four registers are channels n0 + 4 g4 ... + 3 of pixel p16
    if constexpr (NOEPI) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int xh_ = 0; xh_ < 2; ++xh_)
#pragma unroll
          for (int g2 = 0; g2 < NJ; ++g2) asm volatile("" ::"v"(acc[i][xh_][g2]));
    } else {
      const int n = cur.nb * COUT + wn * (16 * NJ) + 4 * g4;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int xh_ = 0; xh_ < 2; ++xh_) {
          const int pix = (cur.b * H + cur.y0 + wm * MI + i) * W + cur.x0 + xh_ * 16 + p16;
          float* yb = y + (size_t)pix * ldy + n;
#pragma unroll
          for (int g2 = 0; g2 < NJ; ++g2) {
            f32x4 v = acc[i][xh_][g2];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (v[r] * inv_x) * inv_w;
            *reinterpret_cast<f32x4*>(yb + g2 * 16) = v;
          }
        }
    }
    if (!has_next) break;
#pragma unroll
    for (int i = 0; i < NPW; ++i) off[i] = offn[i];
    cur = nxt;
    vb = nvb;
  }
}

// fp32 NHWC [M][C] -> two fp16 planes [M + 1][C] of x * s (the extra row: zeros, the halo's out-of-image source)
__global__ void make_planes_kernel(const float* __restrict__ x, size_t M, int C, const float* __restrict__ xmax, _Float16* __restrict__ ph,
                                   _Float16* __restrict__ pl) {
  float s, inv;
  f16_scale(xmax[0], s, inv);
  const size_t n4 = (M + 1) * C / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i * 4 < M * C) v = reinterpret_cast<const f32x4*>(x)[i];
    f16x4 h, l;
    split2_f16(v, s, h, l);
    reinterpret_cast<f16x4*>(ph)[i] = h;
    reinterpret_cast<f16x4*>(pl)[i] = l;
  }
}

// filter [N][9][Cin] fp32 -> [n-block][sub-chunk][tap][plane][ng = 8 groups of 16 channels][lane][8]: lane (n = ng * 16 + (l & 15),
// k = sc * 32 + 8 (l >> 4) + j) + the inverse scale behind
__global__ void pack_sc_kernel(const float* __restrict__ w, _Float16* __restrict__ dst, int N, int Cin, const float* __restrict__ wmax, int NGR) {
  const int nsc = Cin / 32;
  float sw, inv;
  f16_scale(wmax[0], sw, inv);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (n-block, sub-chunk, tap, ng, lane)
  if (i == 0) reinterpret_cast<float*>(dst + (size_t)N * 9 * Cin * 2)[0] = inv;
  if (i >= (N / 16) * nsc * 9 * 64) return;
  const int lane = i & 63;
  const int ng = (i >> 6) % NGR;
  const int gst = (i >> 6) / NGR;
  const int nbk = gst / (nsc * 9);
  const int sc = (gst / 9) % nsc, tap = gst % 9;
  const int n = nbk * (NGR * 16) + ng * 16 + (lane & 15);
  const float* src = w + ((size_t)n * 9 + tap) * Cin + sc * 32 + 8 * (lane >> 4);
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
  f16x4 h0, l0, h1, l1;
  split2_f16(v0, sw, h0, l0);
  split2_f16(v1, sw, h1, l1);
  f16x8 pl[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    pl[0][k] = h0[k]; pl[0][k + 4] = h1[k];
    pl[1][k] = l0[k]; pl[1][k + 4] = l1[k];
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) *reinterpret_cast<f16x8*>(dst + ((((size_t)gst * 2 + p) * NGR + ng) * 64 + lane) * 8) = pl[p];
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
  void (*kern)(const _Float16*, const _Float16*, const _Float16*, float*, int, int, int, int, int, int, int, int, const float*, unsigned);
};
#define VARIANT(name, v) {name, v, planes_kernel<v>}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 2048, H = argc > 2 ? atoi(argv[2]) : 8, W = argc > 3 ? atoi(argv[3]) : 32;
  const int Cin = argc > 4 ? atoi(argv[4]) : 256, N = argc > 5 ? atoi(argv[5]) : 256;
  const int rounds = argc > 6 ? atoi(argv[6]) : 5;
  if (Cin % 64 || N % 128 || W % 32 || H % 4) { fprintf(stderr, "shape not taken by this instance\n"); return 1; }
  const size_t M = (size_t)B * H * W;
  float *x, *w, *yref, *y, *xmax, *wmax;
  _Float16 *ph, *pl;
  CK(hipMalloc(&x, M * Cin * 4));
  CK(hipMalloc(&ph, (M + 1) * Cin * 2));
  CK(hipMalloc(&pl, (M + 1) * Cin * 2));
  CK(hipMalloc(&w, (size_t)N * 9 * Cin * 4));
  CK(hipMalloc(&yref, M * N * 4));
  CK(hipMalloc(&y, M * N * 4));
  CK(hipMalloc(&xmax, 4));
  CK(hipMalloc(&wmax, 4));
  hipLaunchKernelGGL(fill_normal, dim3(4096), dim3(256), 0, 0, x, M * Cin, 12345u, 1.0f);
  hipLaunchKernelGGL(fill_normal, dim3(1024), dim3(256), 0, 0, w, (size_t)N * 9 * Cin, 777u, 0.05f);
  CK(hipDeviceSynchronize());
  if (qea_absmax(x, Cin, (int64_t)M, Cin, xmax, nullptr) || qea_absmax(w, 9 * Cin, N, 9 * Cin, wmax, nullptr)) { fprintf(stderr, "absmax: %s\n", qea_last_error()); return 1; }
  void *flib, *fsc, *fsc256;
  const size_t fpb = qea_pack_frag_planes_f16_bytes(N, Cin);
  CK(hipMalloc(&flib, fpb));
  CK(hipMalloc(&fsc, fpb));
  CK(hipMalloc(&fsc256, fpb));
  if (qea_pack_frag_planes_f16(w, N, Cin, wmax, flib, nullptr)) { fprintf(stderr, "pack: %s\n", qea_last_error()); return 1; }
  {
    const int totalp = (N / 16) * (Cin / 32) * 9 * 64;
    hipLaunchKernelGGL(pack_sc_kernel, dim3((totalp + 255) / 256), dim3(256), 0, 0, w, (_Float16*)fsc, N, Cin, wmax, 8);
    if (N % 256 == 0) hipLaunchKernelGGL(pack_sc_kernel, dim3((totalp + 255) / 256), dim3(256), 0, 0, w, (_Float16*)fsc256, N, Cin, wmax, 16);
  }
  hipLaunchKernelGGL(make_planes_kernel, dim3(4096), dim3(256), 0, 0, x, M, Cin, xmax, ph, pl);
  CK(hipDeviceSynchronize());

  qea_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.x = x; d.w = w; d.y = yref; d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.OH = H; d.OW = W; d.N = N; d.KH = d.KW = 3; d.pad_h = d.pad_w = 1;
  d.stride_h = d.stride_w = 1; d.ldx = Cin; d.ldy = N; d.tile = 24; d.w_frag_planes = flib; d.x_absmax = xmax;
  if (qea_conv_igemm(&d, nullptr)) { fprintf(stderr, "conv: %s\n", qea_last_error()); return 1; }
  CK(hipDeviceSynchronize());
  std::vector<float> href(M * N), hy(M * N);
  CK(hipMemcpy(href.data(), yref, M * N * 4, hipMemcpyDeviceToHost));

  int cus = 256;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  std::vector<Variant> vs = {
      VARIANT("planes wg2", 0),
      VARIANT("planes wg3", V_WG3),
      VARIANT("planes wg2 dmafirst", V_DMAFIRST),
      VARIANT("planes wg2 spread", V_DMASPREAD),
      VARIANT("planes wg3 spread", V_DMASPREAD | V_WG3),
      VARIANT("planes wg2 spread asm", V_DMASPREAD | V_ASM),
      VARIANT("planes wg3 spread asm", V_DMASPREAD | V_ASM | V_WG3),
      VARIANT("planes th8 spread asm", V_TH8 | V_DMASPREAD | V_ASM),
      VARIANT("planes wg2 spread asm Bconst", V_DMASPREAD | V_ASM | V_BCONST),
      VARIANT("planes wg2 spread asm DMAconst", V_DMASPREAD | V_ASM | V_DMACONST),
      VARIANT("planes wg2 spread asm Bconst DMAconst", V_DMASPREAD | V_ASM | V_DMACONST | V_BCONST),
      VARIANT("planes wg2 spread asm noepi", V_DMASPREAD | V_ASM | V_NOEPI),
      VARIANT("planes wg2 spread asm nb3", V_DMASPREAD | V_ASM | V_NB3),
      VARIANT("planes th8 spread asm nb3", V_DMASPREAD | V_ASM | V_NB3 | V_TH8),
      VARIANT("planes WG1 spread asm", V_DMASPREAD | V_ASM | V_WG1),
      VARIANT("planes WG1 spread asm noB", V_DMASPREAD | V_ASM | V_WG1 | V_NOB),
      VARIANT("planes WG1 noB noepi nodma", V_WG1 | V_NOB | V_NOEPI | V_NODMA),
      VARIANT("planes th8 (8 waves)", V_TH8),
      VARIANT("planes th8 spread", V_TH8 | V_DMASPREAD),
      VARIANT("planes n256", V_N256),
      VARIANT("planes th8 n256", V_TH8 | V_N256),
      VARIANT("planes th8 noB noepi nodma", V_TH8 | V_NOB | V_NOEPI | V_NODMA),
      VARIANT("planes n256 noB noepi nodma", V_N256 | V_NOB | V_NOEPI | V_NODMA),
      VARIANT("planes wg2 noB", V_NOB),
      VARIANT("planes wg2 noepi", V_NOEPI),
      VARIANT("planes wg2 nodma", V_NODMA),
      VARIANT("planes wg2 noB noepi nodma", V_NOB | V_NOEPI | V_NODMA),
      VARIANT("planes wg3 noB noepi nodma", V_WG3 | V_NOB | V_NOEPI | V_NODMA),
  };
  auto lds_of = [&](int var) { const int th = (var & V_TH8) ? 8 : 4; const int hp = (th + 2) * 34; return (size_t)2 * 2 * ((hp + 15) / 16 * 16) * 64; };
  for (auto& v : vs) CK(hipFuncSetAttribute((const void*)v.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(v.var)));
  auto launch = [&](const Variant& v) {
    const int th = (v.var & V_TH8) ? 8 : 4;
    const int cout = (v.var & V_N256) ? 256 : 128;
    if (N % cout || H % th) return;
    const long long total = (long long)B * (H / th) * (W / 32) * (N / cout);
    const int wgs = (v.var & (V_TH8 | V_WG1)) ? 1 : (v.var & V_WG3) ? 3 : 2;
    const int resident = wgs * (cus & ~7);
    const unsigned grid = total > resident ? (unsigned)resident : (unsigned)total;
    hipLaunchKernelGGL(v.kern, dim3(grid), dim3((v.var & V_TH8) ? 512 : 256), lds_of(v.var), 0, ph, pl, (const _Float16*)((v.var & V_N256) ? fsc256 : fsc), y, B, H, W, Cin, N, Cin / 32, N, (int)total, xmax,
                       (unsigned)(M * Cin * 2));
  };
  const double flops = 2.0 * M * N * 9.0 * Cin;
  printf("shape B%d H%d W%d Cin%d N%d  %.1f GFLOP\n", B, H, W, Cin, N, flops / 1e9);
  for (auto& v : vs) {
    if (v.var & (V_NOB | V_NOEPI | V_NODMA | V_BCONST | V_DMACONST)) continue;
    if (N % ((v.var & V_N256) ? 256 : 128) || H % ((v.var & V_TH8) ? 8 : 4)) continue;
    CK(hipMemset(y, 0xff, M * N * 4));
    launch(v);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hy.data(), y, M * N * 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxr = 0, se = 0, sr = 0;
    for (size_t i = 0; i < M * N; ++i) {
      const double dd = fabs((double)hy[i] - href[i]);
      if (!(dd <= maxd)) maxd = dd;
      maxr = std::max(maxr, fabs((double)href[i]));
      se += dd * dd;
      sr += (double)href[i] * href[i];
    }
    printf("check %-28s max|d| %.3e (max|ref| %.3e)  l2 rel %.3e\n", v.name, maxd, maxr, sqrt(se / sr));
  }
  std::vector<std::vector<float>> ms(vs.size() + 1);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 4;
  for (int r = 0; r < rounds + 1; ++r) {
    for (size_t k = 0; k <= vs.size(); ++k) {
      CK(hipEventRecord(e0, 0));
      for (int q = 0; q < reps; ++q) {
        if (k == vs.size()) qea_conv_igemm(&d, nullptr);
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
    printf("time  %-28s median %8.1f us  min %8.1f us   %7.1f TF (median)\n", k == vs.size() ? "LIBRARY tile 24" : vs[k].name, med * 1e3, mn * 1e3, flops / med / 1e9);
  }
  return 0;
}
