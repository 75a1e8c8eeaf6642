// Developer lab (round 4): the LDS-halo 3x3 conv (two-way fp16 split, fp32 input gathered and split on the fly, 16x16x32 MFMA) with an
// EIGHT-ROW tile per wave: 8 x 32 pixels x 32 output channels per wave (128 accumulator registers), 32-channel sub-chunks so that the
// LDS image stays at 43.5 KB (two workgroups per CU).  Half the filter-fragment loads per MFMA of the 4-row tile — the largest seam the
// PMC ablations of halo_lab2 found — and a halo overlap of 1.33 instead of 1.59.  Same filter-plane order as the library's 16x16x32
// kernel (qea_pack_frag_planes_f16: [n-block][64-channel chunk][step = tap * 2 + ks][plane][16-channel group][lane][8]); the loop walks it
// ks-major.  Compared with the library's tile 24 on the same random data, interleaved rounds in one process.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/halo_lab3.hip -Iinclude -Lquery-efficient-approx-to-improve-ocr_amd -lqea_hip \
//         -Wl,-rpath,/root/repo/query-efficient-approx-to-improve-ocr_amd -o tools/micro/halo_lab3.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
#include "qea_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
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

enum { V_NOB = 1, V_NOEPI = 2 };

template <int VAR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void th8_kernel(
    const float* __restrict__ x, const _Float16* __restrict__ wf, float* __restrict__ y, int B, int H, int W, int ldx, int ldy, int chunks, int Ntot,
    int total, const float* __restrict__ xmax) {
  constexpr bool NOB = (VAR & V_NOB) != 0, NOEPI = (VAR & V_NOEPI) != 0;
  constexpr int TH = 8, TW = 32, HW_ = 34, HH = TH + 2, HP = HH * HW_, SC = 32, COUT = 128, WN = 4, MI = TH, NG = COUT / 16;
  constexpr int ROWH = SC, PLANE = HP * ROWH, PLANE_B = PLANE * 2;          // halfs per pixel row / per plane
  extern __shared__ __attribute__((aligned(16))) float smem[];
  _Float16* As = reinterpret_cast<_Float16*>(smem);                         // [2 planes][HP][32]
  float sx, inv_x;
  f16_scale(xmax[0], sx, inv_x);
  const float inv_w = reinterpret_cast<const float*>(wf + (size_t)Ntot * 9 * chunks * 64 * 2)[0];
  const int nsc = chunks * 2;

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
  auto rot = [](int p, int slot) { return (slot + (p >> 1)) & 3; };

  constexpr int C4 = SC / 4, NLD = (HP * C4 + 255) / 256, QS = 256 / C4;     // 8 float4 per pixel, 11 per thread, 32 pixels per pass
  f32x4 hv[NLD];
  auto gather = [&](const Item& it, int sc, int tid) {
    const float* xb = x + (size_t)it.b * H * W * ldx + (sc >> 1) * 64 + (sc & 1) * 32 + (tid % C4) * 4;
    int q = tid / C4;
    int hy = q / HW_, hx = q - hy * HW_;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int iy = it.y0 + hy - 1, ix = it.x0 + hx - 1;
      const bool ok = q < HP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      typedef const __attribute__((address_space(1))) f32x4* gptr;
      const gptr pz = (gptr)(const void*)lab_zero16;
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
        const int o = q * ROWH + rot(q, c4 >> 1) * 8 + (c4 & 1) * 4;
        f16x4 h, l;
        split2_f16(hv[i], sx, h, l);
        *reinterpret_cast<f16x4*>(As + o) = h;
        *reinterpret_cast<f16x4*>(As + PLANE + o) = l;
      }
      q += QS;
    }
  };

  const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
  const int p16 = lane_ & 15, g4 = lane_ >> 4;
  const int wn = wave_ % WN;
  // byte offset of (pixel p16 + c, slot g4) = T[c & 1][(c >> 1) & 3] + c * 64
  int T[2][4];
#pragma unroll
  for (int par = 0; par < 2; ++par)
#pragma unroll
    for (int k = 0; k < 4; ++k) T[par][k] = p16 * (ROWH * 2) + (((k + g4 + (p16 >> 1) + (par & p16 & 1)) & 3) << 4);

  f16x8 bq[2][2][2];                                      // [buffer][channel group of the wave][plane]
  auto load_b = [&](int nb, int sc, int tap, int buf, int tid) {
    const int gst = (sc >> 1) * 18 + tap * 2 + (sc & 1);   // the library's order: [chunk][tap * 2 + ks]
    const f16x8* wl = reinterpret_cast<const f16x8*>(wf) + (size_t)nb * chunks * 18 * 2 * NG * 64 + (((tid >> 6) % WN) * 2) * 64 + (tid & 63);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) bq[buf][g2][pl] = wl[(size_t)((gst * 2 + pl) * NG + g2) * 64];
  };

  int vb = blockIdx.x;
  Item cur = decode(vb);
  gather(cur, 0, threadIdx.x);
  load_b(cur.nb, 0, 0, 0, threadIdx.x);
  bool first = true;
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

    for (int sc2 = 0; sc2 < nsc; sc2 += 2) {
      auto half = [&](auto par_) {
        constexpr int PAR = decltype(par_)::value;
        const int sc = sc2 + PAR;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                     // (gather / staging addresses recomputed per sub-chunk, not kept in registers)
        if (!first) __syncthreads();
        first = false;
        stage(tid);
        __syncthreads();
        if (sc + 1 < nsc) gather(cur, sc + 1, tid);
        else if (has_next) gather(nxt, 0, tid);
        constexpr int GR = MI * 2;                        // 16-pixel groups per tap
        auto read_a = [&](int tap, int g, f16x8* a) {
          const int kh = tap / 3, kw = tap % 3;
          const int i = g / 2, xh = g % 2;
          const int c = (i + kh) * HW_ + kw + xh * 16;
          const char* src = reinterpret_cast<const char*>(As) + T[c & 1][(c >> 1) & 3] + c * (ROWH * 2);
          a[0] = *reinterpret_cast<const f16x8*>(src);
          a[1] = *reinterpret_cast<const f16x8*>(src + PLANE_B);
        };
        f16x8 ar[2][2];
        read_a(0, 0, ar[0]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int cb = (tap + PAR) & 1;
          if (!NOB) {
            if (tap + 1 < 9) load_b(cur.nb, sc, tap + 1, cb ^ 1, tid);
            else if (sc + 1 < nsc) load_b(cur.nb, sc + 1, 0, cb ^ 1, tid);
            else if (has_next) load_b(nxt.nb, 0, 0, cb ^ 1, tid);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < GR; ++g) {
            const int f = tap * GR + g;
            const f16x8* a = ar[f & 1];
            const bool more = f + 1 < 9 * GR;
            if (more) read_a((f + 1) / GR, (f + 1) % GR, ar[(f + 1) & 1]);
            const int bb = NOB ? 0 : cb;
            const int i = g / 2, xh = g % 2;
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) acc[i][xh * 2 + g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], bq[bb][g2][0], acc[i][xh * 2 + g2], 0, 0, 0);
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) acc[i][xh * 2 + g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bq[bb][g2][1], acc[i][xh * 2 + g2], 0, 0, 0);
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) acc[i][xh * 2 + g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], bq[bb][g2][0], acc[i][xh * 2 + g2], 0, 0, 0);
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
    // epilogue: element e of tile row i: accumulator tile e >> 2 = xh * 2 + g2, register r = e & 3: pixel xh * 16 + 4 g4 + r, channel wn * 32 + g2 * 16 + p16
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int lp = lane & 15, lg = lane >> 4;
    const int nbase = cur.nb * COUT + wn * 32 + lp;
    if constexpr (NOEPI) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(acc[i][t]));
    } else {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int rowpix = (cur.b * H + cur.y0 + i) * W + cur.x0;
        float* yb = y + (size_t)rowpix * ldy;
        const int lo = 4 * lg * ldy + nbase;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int g2 = (e >> 2) & 1;
          const int c = (e >> 3) * 16 + (e & 3);
          yb[lo + c * ldy + g2 * 16] = (acc[i][e >> 2][e & 3] * inv_x) * inv_w;
        }
      }
    }
    if (!has_next) break;
    cur = nxt;
    vb = nvb;
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
  void (*kern)(const float*, const _Float16*, float*, int, int, int, int, int, int, int, int, const float*);
};
#define VARIANT(name, v) {name, v, th8_kernel<v>}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 2048, H = argc > 2 ? atoi(argv[2]) : 8, W = argc > 3 ? atoi(argv[3]) : 32;
  const int Cin = argc > 4 ? atoi(argv[4]) : 256, N = argc > 5 ? atoi(argv[5]) : 256;
  const int rounds = argc > 6 ? atoi(argv[6]) : 5;
  if (Cin % 64 || N % 128 || W % 32 || H % 8) { fprintf(stderr, "shape not taken by this instance\n"); return 1; }
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
  void* flib;
  const size_t fpb = qea_pack_frag_planes_f16_bytes(N, Cin);
  CK(hipMalloc(&flib, fpb));
  if (qea_pack_frag_planes_f16(w, N, Cin, wmax, flib, nullptr)) { fprintf(stderr, "pack: %s\n", qea_last_error()); return 1; }
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
  std::vector<Variant> vs = {VARIANT("th8", 0), VARIANT("th8 noB", V_NOB), VARIANT("th8 noepi", V_NOEPI), VARIANT("th8 noB noepi", V_NOB | V_NOEPI)};
  const size_t lds = (size_t)2 * 340 * 32 * 2;
  for (auto& v : vs) CK(hipFuncSetAttribute((const void*)v.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long total = (long long)B * (H / 8) * (W / 32) * (N / 128);
  auto launch = [&](const Variant& v) {
    const int resident = 2 * (cus & ~7);
    const unsigned grid = total > resident ? (unsigned)resident : (unsigned)total;
    hipLaunchKernelGGL(v.kern, dim3(grid), dim3(256), lds, 0, x, (const _Float16*)flib, y, B, H, W, Cin, N, Cin / 64, N, (int)total, xmax);
  };
  const double flops = 2.0 * M * N * 9.0 * Cin;
  printf("shape B%d H%d W%d Cin%d N%d  %.1f GFLOP  items %lld\n", B, H, W, Cin, N, flops / 1e9, total);
  for (auto& v : vs) {
    if (v.var) continue;
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
