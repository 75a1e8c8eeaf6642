// Developer lab for the dominant LDS-halo 3x3 conv (two-way fp16 split, 64-channel chunks, 128 output channels, 4 x 32 pixel
// tiles): variants of the kernel side by side with the library's instance on the same random data, interleaved rounds in ONE
// process (cdna_hip_programming.md rule 24), checked against the library's output.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/halo_lab.hip -Iinclude -Lquery-efficient-approx-to-improve-ocr_amd -lqea_hip \
//         -Wl,-rpath,/root/repo/query-efficient-approx-to-improve-ocr_amd -o tools/micro/halo_lab.bin
//   tools/micro/halo_lab.bin B H W Cin Cout
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "qea_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

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
__device__ const float lab_zero16[4] = {0.f, 0.f, 0.f, 0.f};
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, k = bid >> 3;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + k;
}

// variant bits
// V_ZP: out-of-image halo pixels load from a zero-filled 16-byte buffer instead of being selected to zero AFTER the load (the select
// makes hipcc wait for the whole gather right behind its issue: the prefetch under the MFMAs never happens)
// V_NB(n): n filter-fragment buffers, i.e. the filter loads run n - 1 steps ahead and the gather is issued behind step 0's filter load: vmcnt
// retires in order, so every filter load issued after the gather waits for the whole gather; a deeper filter pipeline hides that much more of it
#define V_NB(n) ((n) << 16)
// V_UNSW: 16x16x32 form with the pixels as the A operand (accumulator: column = channel on the lane, rows = 4 consecutive pixels)
// V_INSPLIT: the fp16 split of the prefetched halo is done IN the MFMA loop (one float4 per step, in place), the stage phase only writes LDS
enum { V_UNSW = 1024, V_INSPLIT = 2048, V_ZP = 512, V_ROT = 1, V_M16 = 2, V_NOSTAGE = 4, V_NOB = 8, V_NOEPI = 16, V_STAMP = 32, V_EPI4 = 64, V_NOA = 128, V_WG3 = 256, V_ROTP = 4096, V_NODEEP = 8192 };
// V_NODEEP: with more than two filter buffers, keep the halo gather where the two-buffer form has it (behind the barrier, before the loop)
// V_ROTP (16x16x32 form): slots of pixel row p rotated by p, not p >> 1 — conflict-free for ds_read_b128's 16-lane groups by the guide's bank model
// (the p >> 1 rotation puts lanes {12..15} and {24..27} of a group on the same banks), and only 8 table registers

constexpr int CIN = 64, COUT = 128, TH = 4, TW = 32, HWD = 34, HH = 6, HP = HH * HWD, WN = 4, MI = 4, KS = 4, PLANE = HP * CIN;
constexpr int PLANE_B = PLANE * 2;
constexpr int NSTAMP = 64;

template <int VAR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((VAR & V_WG3) ? 3 : 2, (VAR & V_WG3) ? 3 : 2))) void lab_kernel(
    const float* __restrict__ x, const _Float16* __restrict__ wf, float* __restrict__ y, int B, int H, int W, int ldx, int ldy, int chunks,
    int Ntot, int total, const float* __restrict__ xmax, unsigned long long* __restrict__ stamps) {
  constexpr bool ROT = (VAR & V_ROT) != 0, M16 = (VAR & V_M16) != 0, NOSTAGE = (VAR & V_NOSTAGE) != 0, NOB = (VAR & V_NOB) != 0;
  constexpr bool NOEPI = (VAR & V_NOEPI) != 0, STAMP = (VAR & V_STAMP) != 0, EPI4 = (VAR & V_EPI4) != 0, NOA = (VAR & V_NOA) != 0;
  static_assert(!M16 || ROT, "the 16x16x32 form is written on the rotation swizzle");
  constexpr int NB = (VAR >> 16) ? (VAR >> 16) : 2;
  constexpr bool DEEP = (VAR >> 16) != 0 && (VAR & V_NODEEP) == 0;
  constexpr bool UNSW = (VAR & V_UNSW) != 0, INSPLIT = (VAR & V_INSPLIT) != 0;              // gather issued inside the loop (behind step 0's filter load)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  _Float16* As = reinterpret_cast<_Float16*>(smem);
  float sx, inv_x, inv_w;
  f16_scale(xmax[0], sx, inv_x);
  inv_w = reinterpret_cast<const float*>(wf + (size_t)Ntot * 9 * chunks * CIN * 2)[0];

  const int tiles_x = W / TW, tiles_y = H / TH;
  const int nblk = Ntot / COUT;
  struct Item { int nb, tile_id, b, x0, y0; };
  auto decode = [&](int vb) {
    const int lid = xcd_swizzle(vb, total);
    Item it;
    it.nb = lid % nblk;
    it.tile_id = lid / nblk;
    int bid = it.tile_id;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    it.b = bid / tiles_y;
    it.x0 = tx * TW;
    it.y0 = ty * TH;
    return it;
  };
  constexpr bool ROTP = (VAR & V_ROTP) != 0;
  auto swz = [](int p, int slot) { return ROTP ? ((slot + p) & 7) : ROT ? ((slot + (p >> 1)) & 7) : (slot ^ ((p >> 1) & 7)); };

  constexpr int C4 = CIN / 4, NLD = (HP * C4 + 255) / 256, QS = 256 / C4;
  f32x4 hv[NLD];
  auto gather = [&](const Item& it, int chunk, int tid) {
    const float* xb = x + (size_t)it.b * H * W * ldx + chunk * CIN + (tid % C4) * 4;
    int q = tid / C4;
    int hy = q / HWD, hx = q - hy * HWD;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int iy = it.y0 + hy - 1, ix = it.x0 + hx - 1;
      const bool ok = q < HP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      if constexpr ((VAR & V_ZP) != 0) {
        typedef const __attribute__((address_space(1))) f32x4* gptr;   // (a generic pointer would make these flat loads: lgkmcnt too)
        const gptr pz = (gptr)(const void*)lab_zero16;
        const gptr pv = (gptr)(const void*)(xb + ((size_t)iy * W + ix) * ldx);
        hv[i] = *(ok ? pv : pz);
      } else {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? xb + ((size_t)iy * W + ix) * ldx : x);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        hv[i] = ok ? v : zero;
      }
      q += QS;
      hx += QS;
      if (hx >= HWD) {
        hx -= HWD;
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
        f16x4 h, l;
        if constexpr (INSPLIT) {                          // already split in place: {h, l} in the four registers
          typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
          const f16x8v hl = __builtin_bit_cast(f16x8v, hv[i]);
          h = f16x4{hl[0], hl[1], hl[2], hl[3]};
          l = f16x4{hl[4], hl[5], hl[6], hl[7]};
        } else {
          split2_f16(hv[i], sx, h, l);
        }
        *reinterpret_cast<f16x4*>(As + o) = h;
        *reinterpret_cast<f16x4*>(As + PLANE + o) = l;
      }
      q += QS;
    }
  };
  auto split_one = [&](int i) {
    f16x4 h, l;
    split2_f16(hv[i], sx, h, l);
    const f16x8 hl = {h[0], h[1], h[2], h[3], l[0], l[1], l[2], l[3]};
    hv[i] = __builtin_bit_cast(f32x4, hl);
  };

  // ---- rotation-swizzle address table: byte offset of (lane pixel + c, slot) = T[c & 1][(slot0 + (c >> 1)) & 7] + c * 128, slot0 the
  // lane-independent part of the slot (the lane half rides in q)
  int T[2][8];
  {
    const int lane = threadIdx.x & 63;
    const int px = M16 ? (lane & 15) : (lane & 31), g = M16 ? (lane >> 4) : (lane >> 5);
    const int q0 = (px >> 1) + g, q1 = ((px + 1) >> 1) + g;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      T[0][k] = px * (CIN * 2) + (((k + (ROTP ? px + g : q0)) & 7) << 4);
      T[1][k] = px * (CIN * 2) + (((k + (ROTP ? px + g : q1)) & 7) << 4);
    }
  }

  int vb = blockIdx.x;
  Item cur = decode(vb);
  gather(cur, 0, threadIdx.x);
  if constexpr (INSPLIT) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) split_one(i);
  }
  bool first = true;
  unsigned long long* st_out = stamps + (size_t)blockIdx.x * NSTAMP;
  int nst = 0;
  auto stamp = [&]() {
    if constexpr (STAMP) {
      if (threadIdx.x == 0 && nst < NSTAMP - 4) st_out[nst] = __builtin_amdgcn_s_memtime();
      ++nst;
    }
  };
  if constexpr (STAMP) {
    if (threadIdx.x == 0) {
      st_out[NSTAMP - 4] = __builtin_amdgcn_s_memtime();
      st_out[NSTAMP - 3] = __builtin_amdgcn_s_memrealtime();
    }
  }

  if constexpr (!M16) {
    // =================================================== 32x32x16 form (the library's) =============================================
    constexpr int STEPS = 9 * KS;
    static_assert(STEPS % NB == 0, "buffer index must be a compile-time function of the step");
    f16x8 bq[NB][2];
    auto load_b = [&](int nb, int gst, int buf, int tid) {
      const f16x8* wl = reinterpret_cast<const f16x8*>(wf) + ((tid >> 6) % WN) * 64 + (tid & 63) + (size_t)nb * chunks * STEPS * 2 * WN * 64;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) bq[buf][pl] = wl[(size_t)(gst * 2 + pl) * WN * 64];
    };
#pragma unroll
    for (int d = 0; d < NB - 1; ++d) load_b(cur.nb, d, d, threadIdx.x);   // (every item has >= 36 steps)
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
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int fr = lane & 31, fh = lane >> 5;
        stamp();                                          // 0: chunk start
        if (!NOSTAGE || first) {
          if (!first) __syncthreads();
          stage(tid);
          stamp();                                        // 1: staged
          __syncthreads();
          stamp();                                        // 2: barrier passed
          if (!DEEP) {
            if (chunk + 1 < chunks) gather(cur, chunk + 1, tid);
            else if (has_next) gather(nxt, 0, tid);
          }
        } else {
          stamp();
          stamp();
        }
        const bool do_gather = DEEP && (!NOSTAGE || first);
        first = false;
        auto read_a = [&](int st, int i, f16x8* a) {
          const int tap = st / KS, cs = st % KS;
          const int kh = tap / 3, kw = tap % 3;
          if constexpr (ROT) {
            const int c = (i + kh) * HWD + kw;
            const char* src = reinterpret_cast<const char*>(As) + T[c & 1][(cs * 2 + (c >> 1)) & 7] + c * (CIN * 2);
            a[0] = *reinterpret_cast<const f16x8*>(src);
            a[1] = *reinterpret_cast<const f16x8*>(src + PLANE_B);
          } else {
            const int hp = (i + kh) * HWD + fr + kw;
            const _Float16* src = As + hp * CIN + swz(hp, cs * 2 + fh) * 8;
            a[0] = *reinterpret_cast<const f16x8*>(src);
            a[1] = *reinterpret_cast<const f16x8*>(src + PLANE);
          }
        };
        f16x8 ar[2][2];
        read_a(0, 0, ar[0]);
        stamp();                                          // 3: loop start
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
          const int cb = st % NB;
          if (!NOB) {
            constexpr int D = NB - 1;
            if (st + D < STEPS || chunk + 1 < chunks) load_b(cur.nb, chunk * STEPS + st + D, (st + D) % NB, tid);
            else if (has_next) load_b(nxt.nb, st + D - STEPS, (st + D) % NB, tid);
          }
          if (st == 0 && do_gather) {
            if (chunk + 1 < chunks) gather(cur, chunk + 1, tid);
            else if (has_next) gather(nxt, 0, tid);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int f = st * MI + i;
            const f16x8* a = ar[NOA ? 0 : (f & 1)];
            const bool more = f + 1 < STEPS * MI;
            if (more && !NOA) read_a((f + 1) / MI, (f + 1) % MI, ar[(f + 1) & 1]);
            const int bb = NOB ? 0 : cb;
            if constexpr (EPI4) {                          // operands swapped: D rows = channels, columns = pixels
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bq[bb][0], a[1], acc[i], 0, 0, 0);
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bq[bb][1], a[0], acc[i], 0, 0, 0);
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bq[bb][0], a[0], acc[i], 0, 0, 0);
            } else {
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], bq[bb][0], acc[i], 0, 0, 0);
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bq[bb][1], acc[i], 0, 0, 0);
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bq[bb][0], acc[i], 0, 0, 0);
            }
            if (more && !NOA) {
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            }
          }
        }
        stamp();                                          // 4: loop end
      }
      // ---- epilogue
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      const int lane = tid & 63, wave = tid >> 6;
      const int wn = wave % WN;
      const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = (acc[i][r] * inv_x) * inv_w;
      if constexpr (NOEPI) {
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(acc[i]));
      } else if constexpr (EPI4) {
        // D[row = channel (r&3) + 8 (r>>2) + 4 fh][col = pixel fr]: four consecutive channels per register quad -> float4 stores
        const int n = cur.nb * COUT + wn * 32;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int rowpix = (cur.b * H + cur.y0 + i) * W + cur.x0 + fr;
          float* yb = y + (size_t)rowpix * ldy + n + 4 * fh;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v = {acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
            *reinterpret_cast<f32x4*>(yb + 8 * g) = v;
          }
        }
      } else {
        const int n = cur.nb * COUT + wn * 32 + fr;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int rowpix = (cur.b * H + cur.y0 + i) * W + cur.x0;
          float* yb = y + (size_t)rowpix * ldy;
          const int lo = 4 * fh * ldy + n;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2);
            yb[lo + c * ldy] = acc[i][r];
          }
        }
      }
      stamp();                                            // 5: stored
      if (!has_next) break;
      cur = nxt;
      vb = nvb;
    }
  } else {
    // =================================================== 16x16x32 form ===============================================================
    // filter planes [n-block][chunk][step = tap * 2 + ks][plane][ng = 8 groups of 16 channels][lane][8]: lane (n = ng*16 + (l & 15),
    // k = ks*32 + 8 (l >> 4) + j); the FILTER is the A operand (rows = channels), the pixels the B operand (columns): a lane's four
    // accumulator registers are four consecutive channels of one pixel -> float4 stores.
    constexpr int STEPS = 9 * 2, NG = 8;
    static_assert(STEPS % NB == 0, "buffer index must be a compile-time function of the step");
    f16x8 bq[NB][2][2];                                   // [buffer][channel group of the wave][plane]
    auto load_b = [&](int nb, int gst, int buf, int tid) {
      const f16x8* wl = reinterpret_cast<const f16x8*>(wf) + (size_t)nb * chunks * STEPS * 2 * NG * 64 + (((tid >> 6) % WN) * 2) * 64 + (tid & 63);
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) bq[buf][g2][pl] = wl[(size_t)((gst * 2 + pl) * NG + g2) * 64];
    };
#pragma unroll
    for (int d = 0; d < NB - 1; ++d) load_b(cur.nb, d, d, threadIdx.x);
    while (true) {
      const int nvb = vb + gridDim.x;
      const bool has_next = nvb < total;
      const Item nxt = decode(has_next ? nvb : vb);
      f32x4 acc[MI][2][2];                                // [tile row][half row][channel group]
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int xh = 0; xh < 2; ++xh)
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][xh][g2][r] = 0.f;
      for (int chunk = 0; chunk < chunks; ++chunk) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        stamp();
        if (!NOSTAGE || first) {
          if (!first) __syncthreads();
          stage(tid);
          stamp();
          __syncthreads();
          stamp();
          if (!DEEP) {
            if (chunk + 1 < chunks) gather(cur, chunk + 1, tid);
            else if (has_next) gather(nxt, 0, tid);
          }
        } else {
          stamp();
          stamp();
        }
        const bool do_gather = DEEP && (!NOSTAGE || first);
        const bool first_chunk_flag = first;
        first = false;
        constexpr int GR = MI * 2;                        // 16-pixel groups per step
        auto read_a = [&](int st, int g, f16x8* a) {
          const int tap = st / 2, ks = st % 2;
          const int kh = tap / 3, kw = tap % 3;
          const int i = g / 2, xh = g % 2;
          const int c = (i + kh) * HWD + kw + xh * 16;
          const char* src = reinterpret_cast<const char*>(As) + (ROTP ? T[0][(ks * 4 + c) & 7] : T[c & 1][(ks * 4 + (c >> 1)) & 7]) + c * (CIN * 2);
          a[0] = *reinterpret_cast<const f16x8*>(src);
          a[1] = *reinterpret_cast<const f16x8*>(src + PLANE_B);
        };
        f16x8 ar[2][2];
        read_a(0, 0, ar[0]);
        stamp();
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
          const int cb = st % NB;
          if (!NOB) {
            constexpr int D = NB - 1;
            if (st + D < STEPS || chunk + 1 < chunks) load_b(cur.nb, chunk * STEPS + st + D, (st + D) % NB, tid);
            else if (has_next) load_b(nxt.nb, st + D - STEPS, (st + D) % NB, tid);
          }
          if (st == 0 && do_gather) {
            if (chunk + 1 < chunks) gather(cur, chunk + 1, tid);
            else if (has_next) gather(nxt, 0, tid);
          }
          if constexpr (INSPLIT) {
            if (st >= STEPS - NLD && (do_gather || (!DEEP && (!NOSTAGE || first_chunk_flag)))) split_one(st - (STEPS - NLD));
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < GR; ++g) {
            const int f = st * GR + g;
            const f16x8* a = ar[f & 1];
            const bool more = f + 1 < STEPS * GR;
            if (more) read_a((f + 1) / GR, (f + 1) % GR, ar[(f + 1) & 1]);
            const int bb = NOB ? 0 : cb;
            const int i = g / 2, xh = g % 2;
            if constexpr (UNSW) {
#pragma unroll
              for (int g2 = 0; g2 < 2; ++g2) acc[i][xh][g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], bq[bb][g2][0], acc[i][xh][g2], 0, 0, 0);
#pragma unroll
              for (int g2 = 0; g2 < 2; ++g2) acc[i][xh][g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bq[bb][g2][1], acc[i][xh][g2], 0, 0, 0);
#pragma unroll
              for (int g2 = 0; g2 < 2; ++g2) acc[i][xh][g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bq[bb][g2][0], acc[i][xh][g2], 0, 0, 0);
            } else {
#pragma unroll
              for (int g2 = 0; g2 < 2; ++g2) acc[i][xh][g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[bb][g2][0], a[1], acc[i][xh][g2], 0, 0, 0);
#pragma unroll
              for (int g2 = 0; g2 < 2; ++g2) acc[i][xh][g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[bb][g2][1], a[0], acc[i][xh][g2], 0, 0, 0);
#pragma unroll
              for (int g2 = 0; g2 < 2; ++g2) acc[i][xh][g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[bb][g2][0], a[0], acc[i][xh][g2], 0, 0, 0);
            }
            if (more) {
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
          }
        }
        stamp();
      }
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      const int lane = tid & 63, wave = tid >> 6;
      const int wn = wave % WN;
      const int p16 = lane & 15, g4 = lane >> 4;
      if constexpr (NOEPI) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int xh = 0; xh < 2; ++xh)
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) asm volatile("" ::"v"(acc[i][xh][g2]));
      } else if constexpr (UNSW) {
        // D[row = pixel 4 g4 + r][col = channel p16]
        const int n = cur.nb * COUT + wn * 32 + p16;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int rowpix = (cur.b * H + cur.y0 + i) * W + cur.x0;
          float* yb = y + (size_t)rowpix * ldy + n + (4 * g4) * ldy;
#pragma unroll
          for (int xh = 0; xh < 2; ++xh)
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
              for (int r = 0; r < 4; ++r) yb[(xh * 16 + r) * ldy + g2 * 16] = (acc[i][xh][g2][r] * inv_x) * inv_w;
        }
      } else {
        const int n = cur.nb * COUT + wn * 32 + 4 * g4;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int xh = 0; xh < 2; ++xh) {
            const int pix = (cur.b * H + cur.y0 + i) * W + cur.x0 + xh * 16 + p16;
            float* yb = y + (size_t)pix * ldy + n;
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
              f32x4 v = acc[i][xh][g2];
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = (v[r] * inv_x) * inv_w;
              *reinterpret_cast<f32x4*>(yb + g2 * 16) = v;
            }
          }
      }
      stamp();
      if (!has_next) break;
      cur = nxt;
      vb = nvb;
    }
  }
  if constexpr (STAMP) {
    if (threadIdx.x == 0) {
      st_out[NSTAMP - 2] = __builtin_amdgcn_s_memtime();
      st_out[NSTAMP - 1] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

// filter [N][9][Cin] fp32 -> the 16x16x32 fragment order above (two fp16 planes of the scaled filter) + the inverse scale behind
__global__ void pack16_kernel(const float* __restrict__ w, _Float16* __restrict__ dst, int N, int Cin, const float* __restrict__ wmax) {
  const int chunks = Cin / 64;
  float sw, inv;
  f16_scale(wmax[0], sw, inv);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (n-block, chunk, step, ng, lane)
  if (i == 0) reinterpret_cast<float*>(dst + (size_t)N * 9 * Cin * 2)[0] = inv;
  if (i >= (N / 128) * chunks * 18 * 8 * 64) return;
  const int lane = i & 63;
  const int ng = (i >> 6) % 8;
  const int gst = (i >> 6) / 8;
  const int nbk = gst / (chunks * 18);
  const int chunk = (gst / 18) % chunks, st = gst % 18;
  const int tap = st / 2, ks = st % 2;
  const int n = nbk * 128 + ng * 16 + (lane & 15);
  const float* src = w + ((size_t)n * 9 + tap) * Cin + chunk * 64 + ks * 32 + 8 * (lane >> 4);
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
  for (int p = 0; p < 2; ++p) *reinterpret_cast<f16x8*>(dst + ((((size_t)gst * 2 + p) * 8 + ng) * 64 + lane) * 8) = pl[p];
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
  void (*kern)(const float*, const _Float16*, float*, int, int, int, int, int, int, int, int, const float*, unsigned long long*);
};
#define VARIANT(name, v) {name, v, lab_kernel<v>}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 2048, H = argc > 2 ? atoi(argv[2]) : 8, W = argc > 3 ? atoi(argv[3]) : 32;
  const int Cin = argc > 4 ? atoi(argv[4]) : 256, N = argc > 5 ? atoi(argv[5]) : 256;
  const int rounds = argc > 6 ? atoi(argv[6]) : 5;
  if (Cin % 64 || N % 128 || W % 32 || H % 4) { fprintf(stderr, "shape not taken by this instance\n"); return 1; }
  const size_t M = (size_t)B * H * W;
  float *x, *w, *yref, *y, *xmax, *wmax;
  CK(hipMalloc(&x, M * Cin * 4));
  CK(hipMalloc(&w, (size_t)N * 9 * Cin * 4));
  CK(hipMalloc(&yref, M * N * 4));
  CK(hipMalloc(&y, M * N * 4));
  CK(hipMalloc(&xmax, 4));
  CK(hipMalloc(&wmax, 4));
  hipLaunchKernelGGL(fill_normal, dim3(4096), dim3(256), 0, 0, x, M * Cin, 12345u, 1.0f);
  hipLaunchKernelGGL(fill_normal, dim3(1024), dim3(256), 0, 0, w, (size_t)N * 9 * Cin, 777u, 0.05f);
  CK(hipDeviceSynchronize());
  if (qea_absmax(x, Cin, (int64_t)M, Cin, xmax, nullptr) || qea_absmax(w, 9 * Cin, N, 9 * Cin, wmax, nullptr)) { fprintf(stderr, "absmax: %s\n", qea_last_error()); return 1; }
  void *fp32p, *fp16p;
  const size_t fpb = qea_pack_frag_planes_f16_bytes(N, Cin);
  CK(hipMalloc(&fp32p, fpb));
  CK(hipMalloc(&fp16p, fpb));
  if (qea_pack_frag_planes_f16(w, N, Cin, wmax, fp32p, nullptr)) { fprintf(stderr, "pack: %s\n", qea_last_error()); return 1; }
  {
    const int totalp = (N / 128) * (Cin / 64) * 18 * 8 * 64;
    hipLaunchKernelGGL(pack16_kernel, dim3((totalp + 255) / 256), dim3(256), 0, 0, w, (_Float16*)fp16p, N, Cin, wmax);
  }
  CK(hipDeviceSynchronize());

  qea_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.x = x; d.w = w; d.y = yref; d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.OH = H; d.OW = W; d.N = N; d.KH = d.KW = 3; d.pad_h = d.pad_w = 1;
  d.stride_h = d.stride_w = 1; d.ldx = Cin; d.ldy = N; d.tile = 24; d.w_frag_planes = fp32p; d.x_absmax = xmax;
  if (qea_conv_igemm(&d, nullptr)) { fprintf(stderr, "conv: %s\n", qea_last_error()); return 1; }
  CK(hipDeviceSynchronize());
  std::vector<float> href(M * N), hy(M * N);
  CK(hipMemcpy(href.data(), yref, M * N * 4, hipMemcpyDeviceToHost));

  const long long total = (long long)B * (H / TH) * (W / 32) * (N / COUT);
  int cus = 256;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const size_t lds = (size_t)2 * HP * CIN * 2;
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, (size_t)3 * 256 * NSTAMP * 8));

  std::vector<Variant> vs = {
      VARIANT("base(xor)", 0),
      VARIANT("base+zp", V_ZP),
      VARIANT("rot", V_ROT),
      VARIANT("rot+zp", V_ROT | V_ZP),
      VARIANT("rot+zp+epi4", V_ROT | V_ZP | V_EPI4),
      VARIANT("m16", V_ROT | V_M16),
      VARIANT("m16+zp", V_ROT | V_M16 | V_ZP),
      VARIANT("m16+zp rotp", V_ROT | V_M16 | V_ZP | V_ROTP),
      VARIANT("m16+zp rotp nb3", V_ROT | V_M16 | V_ZP | V_ROTP | V_NB(3)),
      VARIANT("m16+zp nb3", V_ROT | V_M16 | V_ZP | V_NB(3)),
      VARIANT("m16+zp rotp nb3 nodeep", V_ROT | V_M16 | V_ZP | V_ROTP | V_NB(3) | V_NODEEP),
      VARIANT("m16+zp rotp nb6", V_ROT | V_M16 | V_ZP | V_ROTP | V_NB(6)),
      VARIANT("m16+zp unsw", V_ROT | V_M16 | V_ZP | V_UNSW),
      VARIANT("m16+zp insplit", V_ROT | V_M16 | V_ZP | V_INSPLIT),
      VARIANT("m16+zp unsw insplit", V_ROT | V_M16 | V_ZP | V_UNSW | V_INSPLIT),
      VARIANT("abl nostage", V_ROT | V_ZP | V_NOSTAGE),
      VARIANT("abl noB", V_ROT | V_ZP | V_NOB),
      VARIANT("abl noA", V_ROT | V_ZP | V_NOA),
      VARIANT("abl noepi", V_ROT | V_ZP | V_NOEPI),
      VARIANT("abl nostage+noB+noepi", V_ROT | V_ZP | V_NOSTAGE | V_NOB | V_NOEPI),
      VARIANT("m16 abl nostage+noB+noepi", V_ROT | V_ZP | V_M16 | V_NOSTAGE | V_NOB | V_NOEPI),
      VARIANT("stamp base", V_STAMP),
      VARIANT("stamp rot+zp", V_ROT | V_ZP | V_STAMP),
      VARIANT("stamp m16+zp", V_ROT | V_ZP | V_M16 | V_STAMP),
  };
  for (auto& v : vs) CK(hipFuncSetAttribute((const void*)v.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto launch = [&](const Variant& v) {
    const int wgs = (v.var & V_WG3) ? 3 : 2;   // (V_WG3 variants spill: not in the list)
    const int resident = wgs * (cus & ~7);
    const unsigned grid = total > resident ? (unsigned)resident : (unsigned)total;
    hipLaunchKernelGGL(v.kern, dim3(grid), dim3(256), lds, 0, x, (const _Float16*)((v.var & V_M16) ? fp16p : fp32p), y, B, H, W, Cin, N, Cin / CIN, N,
                       (int)total, xmax, stamps);
  };
  const double flops = 2.0 * M * N * 9.0 * Cin;
  printf("shape B%d H%d W%d Cin%d N%d  items %lld  %.1f GFLOP\n", B, H, W, Cin, N, total, flops / 1e9);
  // correctness of the complete variants
  for (auto& v : vs) {
    if (v.var & (V_NOSTAGE | V_NOB | V_NOEPI | V_NOA)) continue;
    CK(hipMemset(y, 0xff, M * N * 4));
    launch(v);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hy.data(), y, M * N * 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxr = 0;
    size_t nbit = 0;
    for (size_t i = 0; i < M * N; ++i) {
      const double dd = fabs((double)hy[i] - href[i]);
      if (!(dd <= maxd)) maxd = dd;                       // (NaN-propagating)
      maxr = std::max(maxr, fabs((double)href[i]));
      nbit += memcmp(&hy[i], &href[i], 4) != 0;
    }
    printf("check %-28s max|d| %.3e (max|ref| %.3e, rel %.2e)  bit-different %zu of %zu\n", v.name, maxd, maxr, maxd / maxr, nbit, M * N);
  }
  // timing: interleaved rounds
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
  // stamps: per-phase medians over workgroups for the stamped variants
  for (auto& v : vs) {
    if (!(v.var & V_STAMP)) continue;
    CK(hipMemset(stamps, 0, (size_t)3 * 256 * NSTAMP * 8));
    for (int q = 0; q < 3; ++q) launch(v);
    CK(hipDeviceSynchronize());
    const int grid = (int)std::min<long long>(total, 2 * (cus & ~7));
    std::vector<unsigned long long> hs((size_t)grid * NSTAMP);
    CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    const int chunks = Cin / CIN;
    const int per_item = 5 * chunks + 1;
    // phases within a chunk: 0->1 stage, 1->2 barrier wait, 2->3 gather issue + first A read, 3->4 MFMA loop; after the last chunk: 4->5 epilogue
    const char* names[5] = {"stage(+pre-barrier)", "barrier", "gather issue", "mfma loop", "epilogue"};
    std::vector<std::vector<double>> ph(5);
    std::vector<double> item_cyc, clk;
    for (int g = 0; g < grid; ++g) {
      const unsigned long long* s = &hs[(size_t)g * NSTAMP];
      const int nitems_rec = (NSTAMP - 4) / per_item;
      if (s[NSTAMP - 1] > s[NSTAMP - 3]) clk.push_back((double)(s[NSTAMP - 2] - s[NSTAMP - 4]) / (double)(s[NSTAMP - 1] - s[NSTAMP - 3]) * 0.1);
      for (int it = 1; it < nitems_rec; ++it) {           // skip the first item (cold)
        const unsigned long long* b = s + it * per_item;
        if (b[per_item - 1] == 0) break;
        for (int c = 0; c < chunks; ++c) {
          for (int p = 0; p < 4; ++p) ph[p].push_back((double)(b[c * 5 + p + 1] - b[c * 5 + p]));
        }
        ph[4].push_back((double)(b[per_item - 1] - b[per_item - 2]));
        item_cyc.push_back((double)(b[per_item - 1] - b[0]));
      }
    }
    printf("stamps %-12s (cycles, median over workgroups; %d chunks per item)\n", v.name, chunks);
    for (int p = 0; p < 5; ++p) {
      if (ph[p].empty()) continue;
      std::sort(ph[p].begin(), ph[p].end());
      printf("   %-22s median %9.0f   p10 %9.0f  p90 %9.0f  (%s)\n", names[p], ph[p][ph[p].size() / 2], ph[p][ph[p].size() / 10], ph[p][ph[p].size() * 9 / 10],
             p < 4 ? "per chunk" : "per item");
    }
    if (!clk.empty()) {
      std::sort(clk.begin(), clk.end());
      printf("   in-kernel clock (memtime / memrealtime x 100 MHz): median %.3f GHz\n", clk[clk.size() / 2]);
    }
    if (!item_cyc.empty()) {
      std::sort(item_cyc.begin(), item_cyc.end());
      printf("   item total median %9.0f cycles; MFMA-only time of an item = %d cycles per wave (x2 waves per SIMD)\n", item_cyc[item_cyc.size() / 2],
             chunks * 432 * 32);
    }
  }
  return 0;
}
