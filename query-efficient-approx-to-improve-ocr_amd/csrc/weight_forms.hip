// The derived weight forms of a model in ONE launch per kind group (round 4): after every optimiser step each 3x3 layer needs its
// filter as fp16 fragment planes (forward) and its flip-transposed filter as fragment planes (input gradient), each 1x1 / GEMM layer
// its fragment planes — about 110 launches of a few microseconds each per training step (a tenth of a B = 32 step).  A job table in
// the kernel arguments (<= 64 jobs), blockIdx.y = job, grid-stride over the job's items; the bodies are those of
// flip_transpose_kernel (norm_pool.hip), pack_frag_planes_f16(_m16)_kernel (conv_igemm.hip) and pack_frag_planes_f16_1x1_kernel
// (gemm1x1.hip): tests/test_kernels_gpu.py::test_weight_forms_multi_equals_the_single_launches checks bit equality.
#include "common.h"
#include "../../include/qea_hip.h"

namespace {

constexpr int MAXJ = 64;
struct JobTable {
  qea_wform_job j[MAXJ];
};

__device__ __forceinline__ void split_store2(const float* src, float sw, _Float16* d0, _Float16* d1) {
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
  f16x4 h0, l0, h1, l1;
  qea_split2_f16(v0, sw, h0, l0);
  qea_split2_f16(v1, sw, h1, l1);
  f16x8 ph, pl;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ph[k] = h0[k]; ph[k + 4] = h1[k];
    pl[k] = l0[k]; pl[k + 4] = l1[k];
  }
  *reinterpret_cast<f16x8*>(d0) = ph;
  *reinterpret_cast<f16x8*>(d1) = pl;
}

__global__ __launch_bounds__(256) void weight_forms_multi_kernel(const JobTable t) {
  const qea_wform_job& jb = t.j[blockIdx.y];
  const float* w = jb.src;
  const long long stride = (long long)gridDim.x * 256;
  const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
  if (jb.kind == 0) {
    // [Co][KH][KW][Ci] -> [Ci][KH][KW][Co], taps flipped
    const int Co = jb.a, Ci = jb.b, KH = jb.c, KW = jb.d;
    float* wt = (float*)jb.dst;
    const long long n = (long long)Co * KH * KW * Ci;
    for (long long i = i0; i < n; i += stride) {
      const int co = (int)(i % Co);
      long long r = i / Co;
      const int kw2 = (int)(r % KW);
      r /= KW;
      const int kh2 = (int)(r % KH);
      const int ci = (int)(r / KH);
      wt[i] = w[(((size_t)co * KH + (KH - 1 - kh2)) * KW + (KW - 1 - kw2)) * Ci + ci];
    }
    return;
  }
  float sw, inv;
  qea_f16_scale(jb.amax[0], sw, inv);
  _Float16* dst = (_Float16*)jb.dst;
  if (jb.kind == 1) {
    const int N = jb.a, Cin = jb.b;
    const int NB = N > 128 ? 128 : N;
    if (i0 == 0) reinterpret_cast<float*>(dst + (size_t)N * 9 * Cin * 2)[0] = inv;
    if (Cin % 64 == 0) {                                       // the order of conv3x3_halo_m16_kernel
      const int NGr = NB / 16, chunks = Cin / 64;
      const long long total = (long long)(N / NB) * chunks * 18 * NGr * 64;
      for (long long ii = i0; ii < total; ii += stride) {
        const int i = (int)ii, lane = i & 63;
        const int ng = (i >> 6) % NGr;
        const int gst = (i >> 6) / NGr;
        const int nbk = gst / (chunks * 18);
        const int chunk = (gst / 18) % chunks, st = gst % 18;
        const int tap = st / 2, ks = st % 2;
        const int n = nbk * NB + ng * 16 + (lane & 15);
        const float* src = w + ((size_t)n * 9 + tap) * Cin + chunk * 64 + ks * 32 + 8 * (lane >> 4);
        _Float16* o = dst + ((((size_t)gst * 2) * NGr + ng) * 64 + lane) * 8;
        split_store2(src, sw, o, o + (size_t)NGr * 64 * 8);
      }
    } else {                                                   // 32-channel layers: the order of conv3x3_halo_bf3_kernel's fp16 form
      const int CW = Cin == 32 ? 32 : 64;
      const int KSr = CW / 16, WNr = NB / 32, chunks = Cin / CW;
      const long long total = (long long)(N / NB) * chunks * 9 * KSr * WNr * 64;
      for (long long ii = i0; ii < total; ii += stride) {
        const int i = (int)ii, lane = i & 63;
        const int nj = (i >> 6) % WNr;
        const int gst = (i >> 6) / WNr;
        const int nbk = gst / (chunks * 9 * KSr);
        const int chunk = (gst / (9 * KSr)) % chunks, st = gst % (9 * KSr);
        const int tap = st / KSr, cs = st % KSr;
        const int n = nbk * NB + nj * 32 + (lane & 31);
        const float* src = w + ((size_t)n * 9 + tap) * Cin + chunk * CW + cs * 16 + 8 * (lane >> 5);
        _Float16* o = dst + ((((size_t)gst * 2) * WNr + nj) * 64 + lane) * 8;
        split_store2(src, sw, o, o + (size_t)WNr * 64 * 8);
      }
    }
    return;
  }
  // kind 2: [N][K] -> the planes of gemm1x1_f16_kernel
  const int N = jb.a, K = jb.b, chunks = K / 64;
  if (i0 == 0) reinterpret_cast<float*>(dst + (size_t)N * K * 2)[0] = inv;
  const long long total = (long long)(N / 128) * chunks * 4 * 4 * 64;
  for (long long ii = i0; ii < total; ii += stride) {
    const int i = (int)ii, lane = i & 63;
    const int nj = (i >> 6) & 3;
    const int gst = i >> 8;
    const int nbk = gst / (chunks * 4);
    const int chunk = (gst >> 2) % chunks, cs = gst & 3;
    const int n = nbk * 128 + nj * 32 + (lane & 31);
    const float* src = w + (size_t)n * K + chunk * 64 + cs * 16 + 8 * (lane >> 5);
    _Float16* o = dst + ((((size_t)gst * 2) * 4 + nj) * 64 + lane) * 8;
    split_store2(src, sw, o, o + (size_t)4 * 64 * 8);
  }
}

}  // namespace

extern "C" int qea_weight_forms_multi(const qea_wform_job* jobs, int32_t n, void* stream) {
  QEA_REQUIRE(jobs && n > 0 && n <= MAXJ, "qea_weight_forms_multi: 1 to 64 jobs");
  JobTable t;
  long long most = 0;
  for (int i = 0; i < n; ++i) {
    const qea_wform_job& j = jobs[i];
    QEA_REQUIRE(j.src && j.dst && j.kind >= 0 && j.kind <= 2 && (j.kind == 0 || j.amax), "qea_weight_forms_multi: null pointer or unknown kind in a job");
    QEA_REQUIRE(((uintptr_t)j.src & 15) == 0 && ((uintptr_t)j.dst & 15) == 0, "qea_weight_forms_multi: pointers must be 16-byte aligned");
    long long items;
    if (j.kind == 0) {
      QEA_REQUIRE(j.a > 0 && j.b > 0 && j.c > 0 && j.d > 0, "qea_weight_forms_multi: bad flip-transpose shape");
      items = (long long)j.a * j.b * j.c * j.d;
    } else if (j.kind == 1) {
      QEA_REQUIRE((j.a == 32 || j.a == 64 || (j.a > 0 && j.a % 128 == 0)) && (j.b == 32 || (j.b % 64 == 0 && j.b > 0 && j.b <= 512)),
                  "qea_weight_forms_multi: 3x3 planes need N in {32, 64, 128k}, Cin = 32 or a multiple of 64 up to 512");
      items = j.b % 64 == 0 ? 9LL * (j.b / 32) * (j.a / 16) * 64 : 9LL * (j.b / 16) * (j.a / 32) * 64;
    } else {
      QEA_REQUIRE(j.a > 0 && j.a % 128 == 0 && j.b > 0 && j.b % 64 == 0 && (long long)j.a * j.b * 4 < 0x7fffffffLL,
                  "qea_weight_forms_multi: 1x1 planes need N a multiple of 128, K a multiple of 64");
      items = (long long)(j.a / 128) * (j.b / 64) * 4 * 4 * 64;
    }
    QEA_REQUIRE(items < 0x7fffffffLL, "qea_weight_forms_multi: a job is too large");
    if (items > most) most = items;
    t.j[i] = j;
  }
  long long gx = (most + 255) / 256;
  if (gx > 512) gx = 512;                                      // grid-stride beyond: 512 x n workgroups fill the chip anyway
  hipLaunchKernelGGL(weight_forms_multi_kernel, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, t);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}
