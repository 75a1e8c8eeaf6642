// The LDS-halo 3x3 convolution in the two-way fp16 form on v_mfma_f32_16x16x32_f16 (tile 24 with fp16 operands and C_in a multiple of 64):
// the kernel, its instances and their launchers — a translation unit of its own so that it compiles beside conv_igemm.hip (the two were one
// file and one five-minute compile).  Dispatch: launch_halo_bf3_any (conv_igemm.hip) -> qea_conv::launch_halo_m16_any.
#include "conv_args.h"
#include <type_traits>

using qea_conv::ConvArgs;

namespace {

// ---------------------------------------------------------------------------------------------
// Round 4: the two-way fp16 form of the LDS-halo kernel for 64-channel chunks on v_mfma_f32_16x16x32_f16 (tile 24 with fp16
// operands and C_in a multiple of 64; the 32-channel-input instances and the three-way bf16 form stay on the kernel above).
// What tools/micro/halo_lab.hip measured on the dominant instance (B = 2048, 256 -> 256 channels at 8 x 32; in-kernel stamps):
// the loop is not stall-bound but CLOCK-bound — 0.76 MFMA-busy at an in-kernel 1.65 GHz — and the 16x16x32 shape holds 1.76-1.79 GHz
// on the same tile at equal cycles per flop (MI355X_MICROARCH.md, DVFS give-back item 7): 420 -> 454 / 437 -> 470 / 352 -> 365
// TFLOP/s on 256->256 / 512->512 / 64->128 channels.  Same tiling as above (TH x 32 pixel tile, one wave = MI rows x 32 output
// channels = MI * 2 pixel groups x 2 channel groups of 16 x 16 accumulator tiles), same LDS image and staging, but:
//  * the 16-byte slots of pixel row p are ROTATED by p, (slot + p) & 7, instead of XOR-ed: a rotation commutes with the
//    compile-time pixel offset of a tap, so all fragment addresses of a lane are 8 table registers + an immediate — no address
//    arithmetic in the loop and no opaque thread-id trick.  (The first form rotated by p >> 1: 16 table registers and, by the guide's
//    16-lane ds_read_b128 groups {0-3, 12-15, 20-27}, a two-way conflict between lanes 12-15 and 24-27 — PMC 0.135 conflict cycles
//    per wave cycle, 0.0 with this one: profiles/r04_halo_lab_pmc.json);
//  * out-of-image halo pixels are LOADED from a zero-filled 16 bytes instead of selected to zero behind the load: the select made
//    hipcc wait for the whole gather right after issuing it;
//  * filter planes in the order [n-block][chunk][step = tap * 2 + ks][plane][16-channel group][lane][8] (qea_pack_frag_planes_f16
//    writes this order for C_in % 64 == 0): lane l of group g holds filter row g * 16 + (l & 15), channels ks * 32 + 8 (l >> 4) + j.
// Accumulator map: tile (row i, half row xh, channel group g2), register r of lane l = pixel xh * 16 + 4 (l >> 4) + r of tile row i,
// channel g2 * 16 + (l & 15) of the wave's 32.  Element index e = (xh * 2 + g2) * 4 + r below.
// ---------------------------------------------------------------------------------------------
__device__ const float qea_zero16[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int halo_m16_wgs(int cout) { return cout == 32 ? 3 : 2; }

template <int COUT, bool STATS, int IMW = 0, int PKW = 0, bool BST = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(halo_m16_wgs(COUT), halo_m16_wgs(COUT)))) void conv3x3_halo_m16_kernel(
    const float* __restrict__ x, const _Float16* __restrict__ wf, float* __restrict__ y, int B, int H, int W, int ldx, int ldy,
    const float* __restrict__ scale, const float* __restrict__ bias, int relu, double* __restrict__ stats, int chunks, int Ntot,
    const float* __restrict__ mask, int ldmask, int total, const float* __restrict__ xmax, float* __restrict__ yamax, float* __restrict__ pooled,
    int ldp, float* __restrict__ pamax, const float* __restrict__ yref, int ldyref, const double* __restrict__ bst64, const float* __restrict__ bsc,
    const float* __restrict__ bsh) {
  constexpr int CIN = 64, TH = 4;
  constexpr bool SMALL = IMW != 0;
  constexpr int IMH = SMALL ? IMW / 4 : 1;
  constexpr int IPX = SMALL ? 32 / IMW : 1, IPY = SMALL ? TH / IMH : 1;
  constexpr int TW = 32, HW_ = SMALL ? IPX * (IMW + 2) : TW + 2, HH = SMALL ? TH + 1 : TH + 2, HP = HH * HW_;
  static_assert(!SMALL || COUT == 128, "small-image tiles: one wave row");
  constexpr int WN = COUT / 32, WM = 4 / WN, MI = TH / WM;
  constexpr int NG = COUT / 16;                           // 16-channel groups per n-block
  constexpr int PLANE = HP * CIN, PLANE_B = PLANE * 2;
  constexpr int STEPS = 18;                               // 9 taps x two 32-channel k-steps per chunk
  static_assert(PKW == 0 || (MI % 2 == 0 && !STATS && (PKW == 1 || PKW == 2)), "fused pooling: row pairs inside one wave, no statistics");
  static_assert(!BST || (!STATS && PKW == 0), "BatchNorm-backward sums: their own instances");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  _Float16* As = reinterpret_cast<_Float16*>(smem);       // [2 planes][HP][64]
  float sx, inv_x;
  qea_f16_scale(xmax[0], sx, inv_x);
  const float inv_w = reinterpret_cast<const float*>(wf + (size_t)Ntot * 9 * chunks * CIN * 2)[0];

  const int tiles_x = W / TW, tiles_y = H / TH;
  const int nblk = Ntot / COUT;
  struct Item { int nb, tile_id, b, x0, y0; };
  auto decode = [&](int vb) {
    const int lid = qea_xcd_swizzle(vb, total);
    Item it;
    it.nb = lid % nblk;
    it.tile_id = lid / nblk;
    if (SMALL) {
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
  auto rot = [](int p, int slot) { return (slot + p) & 7; };

  constexpr int C4 = CIN / 4, NLD = (HP * C4 + 255) / 256, QS = 256 / C4;
  f32x4 hv[NLD];
  auto gather = [&](const Item& it, int chunk, int tid) {
    const float* xb = x + (size_t)it.b * H * W * ldx + chunk * CIN + (tid % C4) * 4;
    int q = tid / C4;
    int hy = q / HW_, hx = q - hy * HW_;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int iy, ix;
      bool ok;
      if (SMALL) {
        const int gx = hx / (IMW + 2);
        const int img = (hy / IMH) * IPX + gx;
        ix = hx - gx * (IMW + 2) - 1;
        iy = (img * IMH + hy % IMH);
        ok = q < HP && hy < TH && (unsigned)ix < (unsigned)IMW && it.b + img < B;
      } else {
        iy = it.y0 + hy - 1;
        ix = it.x0 + hx - 1;
        ok = q < HP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      }
      typedef const __attribute__((address_space(1))) f32x4* gptr;   // (a generic pointer would make these flat loads: lgkmcnt too)
      const gptr pz = (gptr)(const void*)qea_zero16;
      const gptr pv = (gptr)(const void*)(xb + ((size_t)iy * W + ix) * ldx);
      hv[i] = *(ok ? pv : pz);
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
        const int o = q * CIN + rot(q, c4 >> 1) * 8 + (c4 & 1) * 4;
        f16x4 h, l;
        qea_split2_f16(hv[i], sx, h, l);
        *reinterpret_cast<f16x4*>(As + o) = h;
        *reinterpret_cast<f16x4*>(As + PLANE + o) = l;
      }
      q += QS;
    }
  };

  // lane geometry: pixel p16 of a 16-pixel group, 8-channel slot g4 of a 32-channel k-step
  const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
  const int p16 = lane_ & 15, g4 = lane_ >> 4;
  const int wm = wave_ / WN, wn = wave_ % WN;
  // byte offset of (pixel = W0 + lpx + c, slot = ks * 4 + g4) = T[(ks * 4 + c) & 7] + c * 128, c a compile-time pixel offset,
  // W0 = this wave's first tile row, lpx = the lane's pixel inside the group (small 8-pixel images: + the zero columns)
  int T[8];
  {
    const int lpx = (IMW == 8) ? p16 + 2 * (p16 >> 3) : p16;
    const int W0 = SMALL ? 0 : wm * MI * HW_;
#pragma unroll
    for (int k = 0; k < 8; ++k) T[k] = (lpx + W0) * (CIN * 2) + (((k + g4 + lpx + W0) & 7) << 4);
  }

  f16x8 bq[2][2][2];                                      // [buffer][channel group of the wave][plane]
  auto load_b = [&](int nb, int gst, int buf, int tid) {
    const f16x8* wl = reinterpret_cast<const f16x8*>(wf) + (size_t)nb * chunks * STEPS * 2 * NG * 64 + (((tid >> 6) % WN) * 2) * 64 + (tid & 63);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) bq[buf][g2][pl] = wl[(size_t)((gst * 2 + pl) * NG + g2) * 64];
  };

  int vb = blockIdx.x;
  Item cur = decode(vb);
  gather(cur, 0, threadIdx.x);
  load_b(cur.nb, 0, 0, threadIdx.x);
  bool first = true;
  float am = 0.f, pm = 0.f;
  while (true) {
    const int nvb = vb + gridDim.x;
    const bool has_next = nvb < total;
    const Item nxt = decode(has_next ? nvb : vb);
    f32x4 acc[MI][4];                                     // [tile row][half row * 2 + channel group]
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][t][r] = 0.f;

    for (int chunk = 0; chunk < chunks; ++chunk) {
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));                       // (the gather / staging addresses are recomputed per chunk, not kept in registers)
      if (!first) __syncthreads();
      first = false;
      stage(tid);
      __syncthreads();
      if (chunk + 1 < chunks) gather(cur, chunk + 1, tid);
      else if (has_next) gather(nxt, 0, tid);
      constexpr int GR = MI * 2;                          // 16-pixel groups per step
      auto read_a = [&](int st, int g, f16x8* a) {
        const int tap = st / 2, ks = st % 2;
        const int kh = tap / 3, kw = tap % 3;
        const int i = g / 2, xh = g % 2;
        int c;
        if (SMALL) {
          const int rr = i % IMH + kh - 1;
          const int srow = (rr >= 0 && rr < IMH) ? (i / IMH) * IMH + rr : TH;
          c = srow * HW_ + kw + (IMW == 16 ? xh * 18 : xh * 20);
        } else {
          c = (i + kh) * HW_ + kw + xh * 16;
        }
        const char* src = reinterpret_cast<const char*>(As) + T[(ks * 4 + c) & 7] + c * (CIN * 2);
        a[0] = *reinterpret_cast<const f16x8*>(src);
        a[1] = *reinterpret_cast<const f16x8*>(src + PLANE_B);
      };
      f16x8 ar[2][2];
      read_a(0, 0, ar[0]);
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        const int cb = st & 1;
        if (st + 1 < STEPS || chunk + 1 < chunks) load_b(cur.nb, chunk * STEPS + st + 1, cb ^ 1, tid);
        else if (has_next) load_b(nxt.nb, 0, cb ^ 1, tid);
        __builtin_amdgcn_sched_barrier(0);                // keep the next step's filter loads AHEAD of this step's MFMAs
#pragma unroll
        for (int g = 0; g < GR; ++g) {
          const int f = st * GR + g;
          const f16x8* a = ar[f & 1];
          const bool more = f + 1 < STEPS * GR;
          if (more) read_a((f + 1) / GR, (f + 1) % GR, ar[(f + 1) & 1]);
          const int i = g / 2, xh = g % 2;
          // smallest terms first (ll is dropped): lh, hl, hh — the two channel groups interleaved
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2) acc[i][xh * 2 + g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], bq[cb][g2][0], acc[i][xh * 2 + g2], 0, 0, 0);
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2) acc[i][xh * 2 + g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bq[cb][g2][1], acc[i][xh * 2 + g2], 0, 0, 0);
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2) acc[i][xh * 2 + g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bq[cb][g2][0], acc[i][xh * 2 + g2], 0, 0, 0);
          if (more) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // the two LDS reads of group f + 1 ...
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);   // ... ahead of the six MFMAs of group f
          }
        }
      }
    }

    // ---- epilogue.  Element e of tile row i: accumulator tile e >> 2 = xh * 2 + g2, register r = e & 3:
    //      pixel x = xh * 16 + 4 g4 + r, channel = nb * COUT + wn * 32 + g2 * 16 + p16
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int lp = lane & 15, lg = lane >> 4;
    const int nbase = cur.nb * COUT + wn * 32 + lp;
    float esc[2], ebi[2];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
      esc[g2] = scale ? scale[nbase + g2 * 16] : 1.f;
      ebi[g2] = bias ? bias[nbase + g2 * 16] : 0.f;
    }
    double st0[2] = {0.0, 0.0}, st1[2] = {0.0, 0.0};
    auto e_px = [&](int e) { return (e >> 3) * 16 + 4 * lg + (e & 3); };
    if constexpr (PKW != 0) {
#pragma unroll
      for (int i = 0; i < MI; i += 2) {
        float vv[2][16];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int g2 = (e >> 2) & 1, n = nbase + g2 * 16;
            const int px = e_px(e);
            size_t prow;
            bool live = true;
            if (SMALL) {
              const int img = ((i + ii) / IMH) * IPX + px / IMW;
              prow = ((size_t)(cur.b + img) * IMH + (i + ii) % IMH) * IMW + px % IMW;
              live = cur.b + img < B;
            } else {
              prow = (size_t)(cur.b * H + cur.y0 + wm * MI + i + ii) * W + cur.x0 + px;
            }
            float v = (acc[i + ii][e >> 2][e & 3] * inv_x) * inv_w;
            if (scale && bias) v = __fmaf_rn(v, esc[g2], ebi[g2]);
            else if (scale) v *= esc[g2];
            else if (bias) v += ebi[g2];
            if (relu) v = fmaxf(v, 0.f);
            vv[ii][e] = v;
            if (!live) continue;
            y[prow * ldy + n] = v;
            am = qea_amax_acc(am, v);
          }
#pragma unroll
        for (int e = 0; e < 16; e += PKW) {
          const int g2 = (e >> 2) & 1, n = nbase + g2 * 16;
          const int px = e_px(e);                             // even when PKW == 2 (r even)
          float m = -INFINITY;
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int jj = 0; jj < PKW; ++jj) {
              const float v = vv[ii][e + jj];
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
      float msc[2], msh[2];
      double bmu[2], bis[2];
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        msc[g2] = bsc[nbase + g2 * 16];
        msh[g2] = bsh[nbase + g2 * 16];
        bmu[g2] = bst64[nbase + g2 * 16];
        bis[g2] = bst64[Ntot + nbase + g2 * 16];
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int g2 = (e >> 2) & 1, n = nbase + g2 * 16;
          const int px = e_px(e);
          size_t prow;
          bool live = true;
          if (SMALL) {
            const int img = (i / IMH) * IPX + px / IMW;
            prow = ((size_t)(cur.b + img) * IMH + i % IMH) * IMW + px % IMW;
            live = cur.b + img < B;
          } else {
            prow = (size_t)(cur.b * H + cur.y0 + wm * MI + i) * W + cur.x0 + px;
          }
          float v = (acc[i][e >> 2][e & 3] * inv_x) * inv_w;
          if (scale && bias) v = __fmaf_rn(v, esc[g2], ebi[g2]);
          else if (scale) v *= esc[g2];
          else if (bias) v += ebi[g2];
          if (relu) v = fmaxf(v, 0.f);
          if (!live) continue;
          if (mask) v = (mask[prow * ldmask + n] > 0.f) ? v : 0.f;
          y[prow * ldy + n] = v;
          am = qea_amax_acc(am, v);
          const float yv = yref[prow * ldyref + n];         // (the very mask and the very terms of colreduce_kernel<1>)
          const float dz = __fmaf_rn(yv, msc[g2], msh[g2]) > 0.f ? v : 0.f;
          st0[g2] += (double)dz;
          st1[g2] += (double)dz * (((double)yv - bmu[g2]) * bis[g2]);
        }
      }
    } else {
      // option tests outside the element loops, accumulators finished in place, uniform row bases (see the kernel above)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][t][r] = (acc[i][t][r] * inv_x) * inv_w;   // un-scale: exact (powers of two), one factor at a time
      if (scale && bias) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][t][r] = __fmaf_rn(acc[i][t][r], esc[t & 1], ebi[t & 1]);
      } else if (scale) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][t][r] *= esc[t & 1];
      } else if (bias) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][t][r] += ebi[t & 1];
      }
      if (relu) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][t][r] = fmaxf(acc[i][t][r], 0.f);
      }
      auto store_rows = [&](auto has_mask) {
        constexpr bool HAS_MASK = decltype(has_mask)::value;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int rowpix = (cur.b * H + cur.y0 + wm * MI + i) * W + cur.x0;
          float* yb = y + (size_t)rowpix * ldy;
          const float* mb = HAS_MASK ? mask + (size_t)rowpix * ldmask : nullptr;
          const int lo = 4 * lg * ldy + nbase, lom = 4 * lg * ldmask + nbase;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int g2 = (e >> 2) & 1;
            const int c = (e >> 3) * 16 + (e & 3);          // pixel = c + 4 lg
            float v = acc[i][e >> 2][e & 3];
            if (SMALL) {
              const int px = c + 4 * lg;
              const int img = (i / IMH) * IPX + px / IMW;
              const size_t prow = ((size_t)(cur.b + img) * IMH + i % IMH) * IMW + px % IMW;
              if (cur.b + img >= B) continue;
              const int n = nbase + g2 * 16;
              if (HAS_MASK) v = (mask[prow * ldmask + n] > 0.f) ? v : 0.f;
              y[prow * ldy + n] = v;
            } else {
              if (HAS_MASK) v = (mb[lom + c * ldmask + g2 * 16] > 0.f) ? v : 0.f;
              yb[lo + c * ldy + g2 * 16] = v;
            }
            am = qea_amax_acc(am, v);
            if (STATS) {
              st0[g2] += (double)v;
              st1[g2] += (double)v * (double)v;
            }
          }
        }
      };
      if (mask) store_rows(std::true_type{});
      else store_rows(std::false_type{});
    }
    if (STATS || BST) {                                        // one partial per (pixel tile, wave row): [blocks][Ntot][2]
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        double sa = st0[g2], sc = st1[g2];                     // the four lane groups hold different pixels of the same column
        sa += __shfl_xor(sa, 16, 64);
        sc += __shfl_xor(sc, 16, 64);
        sa += __shfl_xor(sa, 32, 64);
        sc += __shfl_xor(sc, 32, 64);
        if (lg == 0) {
          double* dst = stats + ((size_t)(cur.tile_id * WM + wm) * Ntot + nbase + g2 * 16) * 2;
          dst[0] = sa;
          dst[1] = sc;
        }
      }
    }
    if (!has_next) break;
    cur = nxt;
    vb = nvb;
  }
  qea_amax_commit_block(am, yamax);
  if constexpr (PKW != 0) qea_amax_commit_block(pm, pamax);
}

template <int COUT, bool STATS, int IMW = 0, int PKW = 0, bool BST = false>
int launch_halo_m16_(const ConvArgs& a, hipStream_t s) {
  constexpr size_t lds = IMW ? (size_t)2 * (4 + 1) * (32 / IMW) * (IMW + 2) * 64 * 2 : (size_t)2 * (4 + 2) * 34 * 64 * 2;
  auto kern = conv3x3_halo_m16_kernel<COUT, STATS, IMW, PKW, BST>;
  static int attr_rc = (int)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr_rc != (int)hipSuccess) {
    qea_set_error("qea_conv_igemm(halo m16): cannot reserve %zu bytes of LDS: %s", (size_t)lds, hipGetErrorString((hipError_t)attr_rc));
    return QEA_ERR_LAUNCH;
  }
  const long long total = IMW ? (long long)qea_cdiv(a.B, (32 / IMW) * (4 / (IMW / 4))) * (a.N / COUT) : (long long)a.B * (a.H / 4) * (a.W / 32) * (a.N / COUT);
  if (total <= 0 || total > 0x7fffffffLL) {
    qea_set_error("qea_conv_igemm(halo m16): grid %lld out of range", total);
    return QEA_ERR_INVALID;
  }
  static const int resident = [] {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return halo_m16_wgs(COUT) * (cus & ~7);
  }();
  const unsigned grid = (total > resident && COUT > 32) ? (unsigned)resident : (unsigned)total;   // (32-channel outputs: one item per workgroup, see above)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a.x, (const _Float16*)a.wp, a.y, a.B, a.H, a.W, a.ldx, a.ldy, a.scale, a.bias, a.relu, a.stats,
                     a.Cin / 64, a.N, a.mask, a.ldmask, (int)total, a.xmax, a.yamax, a.pool_y, a.ldpool, a.pool_amax, a.bst_y, a.ldbst, a.bst64, a.bst_scale,
                     a.bst_shift);
  return QEA_OK;
}

// every variant of one (COUT, image width) of the 16x16x32 kernel: fused pool / BatchNorm-backward sums / forward statistics / plain
template <int COUT, int IMW = 0>
int launch_halo_m16(const ConvArgs& a, hipStream_t s) {
  if (a.pool_y) {
    if constexpr (COUT != 32 && IMW != 8) {
      if (a.pool_kw == 1) {
        if constexpr (COUT == 128 && IMW == 0) return launch_halo_m16_<COUT, false, IMW, 1>(a, s);
      } else {
        return launch_halo_m16_<COUT, false, IMW, 2>(a, s);
      }
    }
    qea_set_error("qea_conv_igemm(halo m16): no fused-pool instance for this shape");
    return QEA_ERR_INVALID;
  }
  if (a.bst_y) return launch_halo_m16_<COUT, false, IMW, 0, true>(a, s);
  return a.stats ? launch_halo_m16_<COUT, true, IMW>(a, s) : launch_halo_m16_<COUT, false, IMW>(a, s);
}

}  // namespace

namespace qea_conv {

// n_sel: output-channel group of the instance (32, 64, 128); sm: small-image width (0, 16, 8)
int launch_halo_m16_any(int n_sel, int sm, const ConvArgs& a, hipStream_t s) {
  if (sm == 16) return launch_halo_m16<128, 16>(a, s);
  if (sm == 8) return launch_halo_m16<128, 8>(a, s);
  if (n_sel == 32) return launch_halo_m16<32>(a, s);
  if (n_sel == 64) return launch_halo_m16<64>(a, s);
  return launch_halo_m16<128>(a, s);
}

}  // namespace qea_conv
