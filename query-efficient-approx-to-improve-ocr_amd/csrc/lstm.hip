// Bidirectional LSTM time step on the fp32 matrix cores, one launch per step, both directions
// per launch (nn.LSTM(512, 256, 2, bidirectional=True), reference models/model_crnn.py:9,19).
//
// Forward step t:   gates = gx[t] + h_prev * W_hh^T ; i,f,o = sigmoid, g = tanh ;
//                   c = f*c_prev + i*g ; h = o*tanh(c)
// A workgroup owns 32 batch rows x 32 hidden units (all four gates -> four 32x32 MFMA tiles whose
// element (row, unit) sits in the SAME lane, so the gate math needs no data exchange); its four
// waves split K = 256 and meet through LDS.  W_hh is pre-packed once per training step into the
// exact per-lane fragment order (qea_lstm_pack_whh) so every weight fetch is a 1 KiB coalesced
// wave load; it is L2-resident (1 MiB per direction).
// Backward step: dh_rec = dgates[t] * W_hh (K = 1024 split over the four waves), fused with the
// gate backward of the PREVIOUS time step in the epilogue, which overwrites the saved gate
// activations of that step with their pre-activation gradients in place.
//
// Roofline: MFMA (2*B*1024*256 flops per direction per step) — latency matters as much: 31
// dependent launches per layer and direction pair.
#include "common.h"

namespace {

constexpr int HID = 256;       // hidden units per direction
constexpr int GATES = 4 * HID;  // 1024

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// Wp[u][w][kb][j][lane][4] <- Wt[n][k], n = j*n_stride_tile + u*32 + (lane&31), k = w*KW + kb*8 + (lane>>5)*4 + e
__global__ void pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int NT, int KW, int tile_stride, long long sn,
                            long long sk, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 3);
    const int lane = (int)((i >> 2) & 63);
    long long r = i >> 8;
    const int j = (int)(r % NT);
    r /= NT;
    const int kb = (int)(r % (KW / 8));
    r /= (KW / 8);
    const int w = (int)(r & 3);
    const int u = (int)(r >> 2);
    const int n = j * tile_stride + u * 32 + (lane & 31);
    const int k = w * KW + kb * 8 + (lane >> 5) * 4 + e;
    dst[i] = src[n * sn + k * sk];
  }
}

struct StepArgs {
  // per direction d: pointer = base + d * dir_stride
  const float* a;   // GEMM A operand rows [B][lda] (fwd: h_prev, bwd: dgates[t])
  long long a_dir;  // element offset between directions
  int lda;
  const float* wp;  // packed weights (fp32 kernel) / three-plane bf16 fragments (split kernel)
  long long wp_dir;  // elements (fp32 / bf16) between the directions
  int B;
  int skip_gemm;  // first step: A is implicitly zero
  // forward epilogue
  float* gates;  // gx[t] in, activations out  [B][ldg] (+ d*GATES)
  int ldg;
  const float* c_prev;  // [B][ldc] (+ d*HID) or null
  float* c_out;
  int ldc;
  float* h_out;  // y[t] [B][ldh] (+ d*HID)
  int ldh;
  // backward epilogue (acts on the step "e" = previous step in processing order)
  const float* dy;  // grad of layer output at step e [B][ldh] (+ d*HID)
  float* dgates_e;  // gates of step e, in: activations, out: pre-activation grads
  const float* c_e;       // c at step e
  const float* c_e_prev;  // c at the step before e in TIME (null -> zeros)
  float* dc;              // running dc  [B][2*HID] (+ d*HID), in/out
  int dc_init;            // 1: treat dc as zero on read
  // time offsets per direction (elements), added to the pointers above when d == 1
  long long rev_gates, rev_c_prev, rev_c_out, rev_h_out, rev_a, rev_dy, rev_dgates_e, rev_c_e, rev_c_e_prev;
  int has_c_prev_rev, has_c_e_prev_rev;  // whether the reverse direction has a predecessor state
  int has_c_prev_fwd, has_c_e_prev_fwd;
};

// gate math of one (row, unit) element given the recurrent GEMM result s[] (forward: the four gate pre-activation
// contributions; backward: s[0] = dh_rec and the gate backward of step e)
template <bool FWD>
__device__ __forceinline__ void gate_math(const StepArgs& p, int d, int row, int unit, const float* s) {
  if constexpr (FWD) {
    float* g = p.gates + (d ? GATES + p.rev_gates : 0) + (size_t)row * p.ldg + unit;
    const float gi = sigmoidf_(g[0 * HID] + s[0]);
    const float gf = sigmoidf_(g[1 * HID] + s[1]);
    const float gg = tanhf(g[2 * HID] + s[2]);
    const float go = sigmoidf_(g[3 * HID] + s[3]);
    const bool has_prev = d ? p.has_c_prev_rev : p.has_c_prev_fwd;
    float cp = 0.f;
    if (has_prev) cp = p.c_prev[(d ? HID + p.rev_c_prev : 0) + (size_t)row * p.ldc + unit];
    const float c = gf * cp + gi * gg;
    const float h = go * tanhf(c);
    g[0 * HID] = gi;
    g[1 * HID] = gf;
    g[2 * HID] = gg;
    g[3 * HID] = go;
    p.c_out[(d ? HID + p.rev_c_out : 0) + (size_t)row * p.ldc + unit] = c;
    p.h_out[(d ? HID + p.rev_h_out : 0) + (size_t)row * p.ldh + unit] = h;
  } else {
    float* g = p.dgates_e + (d ? GATES + p.rev_dgates_e : 0) + (size_t)row * p.ldg + unit;
    const float gi = g[0 * HID], gf = g[1 * HID], gg = g[2 * HID], go = g[3 * HID];
    const float dh = p.dy[(d ? HID + p.rev_dy : 0) + (size_t)row * p.ldh + unit] + s[0];
    const float c = p.c_e[(d ? HID + p.rev_c_e : 0) + (size_t)row * p.ldc + unit];
    const bool has_prev = d ? p.has_c_e_prev_rev : p.has_c_e_prev_fwd;
    float cp = 0.f;
    if (has_prev) cp = p.c_e_prev[(d ? HID + p.rev_c_e_prev : 0) + (size_t)row * p.ldc + unit];
    float* dcp = p.dc + (size_t)row * (2 * HID) + d * HID + unit;
    const float dc_in = p.dc_init ? 0.f : *dcp;
    const float tc = tanhf(c);
    const float dc = dh * go * (1.f - tc * tc) + dc_in;
    g[0 * HID] = dc * gg * gi * (1.f - gi);
    g[1 * HID] = dc * cp * gf * (1.f - gf);
    g[2 * HID] = dc * gi * (1.f - gg * gg);
    g[3 * HID] = dh * tc * go * (1.f - go);
    *dcp = dc * gf;
  }
}

template <int NT, int KW, bool FWD>
__global__ __launch_bounds__(256) void lstm_step_kernel(const StepArgs p) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][NT][16][64]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m0 = blockIdx.x * 32, u = blockIdx.y, d = blockIdx.z;

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  if (!p.skip_gemm) {
    const int row = m0 + (lane & 31);
    const float* arow = p.a + (d ? p.a_dir + p.rev_a : 0) + (size_t)row * p.lda + w * KW + (lane >> 5) * 4;
    const float* wp = p.wp + (size_t)d * p.wp_dir + ((size_t)(u * 4 + w) * (KW / 8)) * NT * 256 + lane * 4;
    const bool live = row < p.B;
#pragma unroll 4
    for (int kb = 0; kb < KW / 8; ++kb) {
      f32x4 af = {0.f, 0.f, 0.f, 0.f};
      if (live) af = *reinterpret_cast<const f32x4*>(arow + kb * 8);
      f32x4 bf[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(wp + ((size_t)kb * NT + j) * 256);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[j][s], acc[j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((w * NT + j) * 16 + r) * 64 + lane] = acc[j][r];
  __syncthreads();

  // 16*64 (row, unit) elements per workgroup, 4 per thread
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e = tid + 256 * it;
    const int r = e >> 6, l = e & 63;
    const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    const int unit = u * 32 + (l & 31);
    float s[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      s[j] = 0.f;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) s[j] += red[((ww * NT + j) * 16 + r) * 64 + l];
    }
    if (row >= p.B) continue;
    gate_math<FWD>(p, d, row, unit, s);
  }
}


// ---------------------------------------------------------------------------------------------
// Split-bf16 form of the step (the default MFMA mode): h_prev / dgates rows are split on the fly into three bf16 values per
// element, W_hh comes pre-split in fragment order (qea_lstm_pack_whh_split: [unit block][k-step][tile][plane][lane][8]), six
// v_mfma_f32_32x32x16_bf16 per product as in the convolution kernels.  Two workgroup shapes:
//   RG = 1: 32 rows, the four waves split K and meet through LDS (small batches: B / 32 x 8 x 2 workgroups);
//   RG = 4: 128 rows, every wave owns 32 rows over the full K, gate math straight from its accumulators; the weight
//           fragments of a stage (48 KiB = a quarter of K) go global -> LDS by LDS-DMA once per workgroup, double
//           buffered.  One wave per SIMD cannot hide latency by occupancy, so every load is issued a whole stage
//           ahead (weights, A rows) and the epilogue's operands before / under the GEMM.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glob_void_t;

// The epilogue's operands of one (row, unit) element, loaded BEFORE the recurrent GEMM so that their HBM latency hides under
// it (gates is read and written through the same pointer: left to the compiler, every element's loads wait for the previous
// element's stores — sixteen serial round trips per lane in the 128-row shape).
template <bool FWD>
struct GateIn {
  float g[4];       // forward: gate pre-activations x*W_ih^T + b; backward: saved gate activations of step e
  float cp;         // c of the previous step in time (0 if none)
  float dyv, c, dc_in;   // backward only
};

template <bool FWD>
__device__ __forceinline__ GateIn<FWD> gate_load(const StepArgs& p, int d, int row, int unit) {
  GateIn<FWD> in;
  if constexpr (FWD) {
    const float* g = p.gates + (d ? GATES + p.rev_gates : 0) + (size_t)row * p.ldg + unit;
#pragma unroll
    for (int j = 0; j < 4; ++j) in.g[j] = g[j * HID];
    const bool has_prev = d ? p.has_c_prev_rev : p.has_c_prev_fwd;
    in.cp = has_prev ? p.c_prev[(d ? HID + p.rev_c_prev : 0) + (size_t)row * p.ldc + unit] : 0.f;
    in.dyv = in.c = in.dc_in = 0.f;
  } else {
    const float* g = p.dgates_e + (d ? GATES + p.rev_dgates_e : 0) + (size_t)row * p.ldg + unit;
#pragma unroll
    for (int j = 0; j < 4; ++j) in.g[j] = g[j * HID];
    in.dyv = p.dy[(d ? HID + p.rev_dy : 0) + (size_t)row * p.ldh + unit];
    in.c = p.c_e[(d ? HID + p.rev_c_e : 0) + (size_t)row * p.ldc + unit];
    const bool has_prev = d ? p.has_c_e_prev_rev : p.has_c_e_prev_fwd;
    in.cp = has_prev ? p.c_e_prev[(d ? HID + p.rev_c_e_prev : 0) + (size_t)row * p.ldc + unit] : 0.f;
    in.dc_in = p.dc_init ? 0.f : p.dc[(size_t)row * (2 * HID) + d * HID + unit];
  }
  return in;
}

// the arithmetic of gate_math on pre-loaded operands (same expressions, same order: bit-identical results)
template <bool FWD>
__device__ __forceinline__ void gate_finish(const StepArgs& p, int d, int row, int unit, const float* s, const GateIn<FWD>& in) {
  if constexpr (FWD) {
    float* g = p.gates + (d ? GATES + p.rev_gates : 0) + (size_t)row * p.ldg + unit;
    const float gi = sigmoidf_(in.g[0] + s[0]);
    const float gf = sigmoidf_(in.g[1] + s[1]);
    const float gg = tanhf(in.g[2] + s[2]);
    const float go = sigmoidf_(in.g[3] + s[3]);
    const float c = gf * in.cp + gi * gg;
    const float h = go * tanhf(c);
    g[0 * HID] = gi;
    g[1 * HID] = gf;
    g[2 * HID] = gg;
    g[3 * HID] = go;
    p.c_out[(d ? HID + p.rev_c_out : 0) + (size_t)row * p.ldc + unit] = c;
    p.h_out[(d ? HID + p.rev_h_out : 0) + (size_t)row * p.ldh + unit] = h;
  } else {
    float* g = p.dgates_e + (d ? GATES + p.rev_dgates_e : 0) + (size_t)row * p.ldg + unit;
    const float gi = in.g[0], gf = in.g[1], gg = in.g[2], go = in.g[3];
    const float dh = in.dyv + s[0];
    const float tc = tanhf(in.c);
    const float dc = dh * go * (1.f - tc * tc) + in.dc_in;
    g[0 * HID] = dc * gg * gi * (1.f - gi);
    g[1 * HID] = dc * in.cp * gf * (1.f - gf);
    g[2 * HID] = dc * gi * (1.f - gg * gg);
    g[3 * HID] = dh * tc * go * (1.f - go);
    p.dc[(size_t)row * (2 * HID) + d * HID + unit] = dc * gf;
  }
}

template <int NT, bool FWD, int RG>
__global__ __launch_bounds__(256) void lstm_step_bf3_kernel(const StepArgs p) {
  constexpr int KTOT = FWD ? HID : GATES;
  constexpr int KSTEPS = KTOT / 16;                    // 16 / 64
  constexpr int KPS = 48 / (NT * 3);                   // k-steps per 48-fragment weight stage: 4 / 16
  constexpr int NSTAGE = KSTEPS / KPS;                 // 4
  static_assert(NT * 3 * KPS == 48 && NSTAGE == 4, "a stage is forty-eight 1 KiB fragments");
  extern __shared__ __attribute__((aligned(16))) float red[];   // RG 1: [4 waves][NT][16][64] floats; RG 4: [2][48][64][8] bf16
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int u = blockIdx.y, d = blockIdx.z;
  const int m0 = blockIdx.x * (32 * RG) + (RG == 4 ? w * 32 : 0);

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int row = m0 + (lane & 31);
  const float* arow = p.a + (d ? p.a_dir + p.rev_a : 0) + (size_t)(row < p.B ? row : 0) * p.lda + (lane >> 5) * 8;
  const __bf16* wbase = reinterpret_cast<const __bf16*>(p.wp) + (size_t)d * p.wp_dir + (size_t)u * (KSTEPS * NT * 3 * 512) + lane * 8;
  auto split_a = [&](const f32x4& v0, const f32x4& v1, bf16x8* af) {
    bf16x4 h0, m0_, l0, h1, m1, l1;
    qea_split3(v0, h0, m0_, l0);
    qea_split3(v1, h1, m1, l1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      af[0][k] = h0[k]; af[0][k + 4] = h1[k];
      af[1][k] = m0_[k]; af[1][k + 4] = m1[k];
      af[2][k] = l0[k]; af[2][k + 4] = l1[k];
    }
  };
  auto mfma6 = [&](f32x16& c, const bf16x8* af, const bf16x8* bf) {   // smallest terms first: lh, hl, mm, mh, hm, hh
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], c, 0, 0, 0);
  };

  if constexpr (RG == 4) {
    char* const lds = reinterpret_cast<char*>(red);
    const int unit = u * 32 + (lane & 31);
    GateIn<FWD> ein[16];
    auto load_epilogue = [&]() {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        ein[r] = gate_load<FWD>(p, d, rr < p.B ? rr : 0, unit);
      }
    };
    if (!p.skip_gemm) {
      auto dma = [&](int stage, int buf) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int f = w + 4 * i;                               // fragment of the stage, 1 KiB, this wave's DMA piece
          __builtin_amdgcn_global_load_lds((glob_void_t*)(wbase + (size_t)(stage * 48 + f) * 512), (lds_void_t*)(lds + (buf * 48 + f) * 1024), 16, 0, 0);
        }
      };
      f32x4 abuf[2][KPS][2];                                     // every buffer index below is a compile-time constant
      auto load_a = [&](int stage, int b) {
#pragma unroll
        for (int i = 0; i < KPS; ++i) {
          abuf[b][i][0] = *reinterpret_cast<const f32x4*>(arow + (stage * KPS + i) * 16);
          abuf[b][i][1] = *reinterpret_cast<const f32x4*>(arow + (stage * KPS + i) * 16 + 4);
        }
      };
      // hipcc drains every outstanding load (vmcnt(0)) at a barrier that follows an LDS-DMA, so a stage is made long enough
      // (a quarter of K: 96 MFMAs per wave, ~1.4 us) for the NEXT stage's weights and A rows to arrive under it
      dma(0, 0);
      load_a(0, 0);
      if (FWD) load_epilogue();
      __syncthreads();
#pragma unroll
      for (int st = 0; st < NSTAGE; ++st) {
        const int buf = st & 1;
        if (st + 1 < NSTAGE) {
          dma(st + 1, buf ^ 1);
          load_a(st + 1, buf ^ 1);
        }
        if (!FWD && st == NSTAGE - 1) load_epilogue();           // backward: 128 more registers, once the other A buffer is free
#pragma unroll
        for (int q = 0; q < KPS; ++q) {
          bf16x8 af[3];
          split_a(abuf[buf][q][0], abuf[buf][q][1], af);
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            bf16x8 bf[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bf[pl] = *reinterpret_cast<const bf16x8*>(lds + ((buf * 48 + (q * NT + j) * 3 + pl) * 64 + lane) * 16);
            mfma6(acc[j], af, bf);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                         // stage st + 1 has landed; buffer `buf` is free again
      }
    } else {
      load_epilogue();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (rr >= p.B) continue;
      float s[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) s[j] = acc[j][r];
      gate_finish<FWD>(p, d, rr, unit, s, ein[r]);
    }
  } else {
    // epilogue element `it` of this thread: (r, l) = ((tid + 256 it) >> 6, & 63)
    GateIn<FWD> ein[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int e = tid + 256 * it;
      const int r = e >> 6, l = e & 63;
      const int rr = m0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      ein[it] = gate_load<FWD>(p, d, rr < p.B ? rr : 0, u * 32 + (l & 31));
    }
    if (!p.skip_gemm) {
      constexpr int MYKS = KSTEPS / 4;                           // this wave's k-steps: [w * MYKS, (w + 1) * MYKS)
      constexpr int PF = MYKS < 8 ? MYKS : 8;                    // A prefetch depth (k-steps)
      f32x4 ar[PF][2];
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        ar[i][0] = *reinterpret_cast<const f32x4*>(arow + (w * MYKS + i) * 16);
        ar[i][1] = *reinterpret_cast<const f32x4*>(arow + (w * MYKS + i) * 16 + 4);
      }
#pragma unroll
      for (int q = 0; q < MYKS; ++q) {
        const int ks = w * MYKS + q;
        bf16x8 bf[NT][3];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) bf[j][pl] = *reinterpret_cast<const bf16x8*>(wbase + (size_t)((ks * NT + j) * 3 + pl) * 512);
        bf16x8 af[3];
        split_a(ar[q % PF][0], ar[q % PF][1], af);
        if (q + PF < MYKS) {
          ar[q % PF][0] = *reinterpret_cast<const f32x4*>(arow + (ks + PF) * 16);
          ar[q % PF][1] = *reinterpret_cast<const f32x4*>(arow + (ks + PF) * 16 + 4);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) mfma6(acc[j], af, bf[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((w * NT + j) * 16 + r) * 64 + lane] = acc[j][r];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int e = tid + 256 * it;
      const int r = e >> 6, l = e & 63;
      const int rr = m0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      const int unit = u * 32 + (l & 31);
      float s[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        s[j] = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) s[j] += red[((ww * NT + j) * 16 + r) * 64 + l];
      }
      if (rr >= p.B) continue;
      gate_finish<FWD>(p, d, rr, unit, s, ein[it]);
    }
  }
}

// planes[u][ks][j][plane][lane][8] <- Wt[n][k], n = j*tile_stride + u*32 + (lane & 31), k = ks*16 + (lane >> 5)*8 + e
__global__ void pack3_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int NT, int KSTEPS, int tile_stride, long long sn,
                             long long sk, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;            // (u, ks, j, lane)
  if (i >= total) return;
  const int lane = i & 63;
  const int j = (i >> 6) % NT;
  const int ks = ((i >> 6) / NT) % KSTEPS;
  const int u = ((i >> 6) / NT) / KSTEPS;
  const int n = j * tile_stride + u * 32 + (lane & 31);
  const int k0 = ks * 16 + (lane >> 5) * 8;
  f32x4 v0, v1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v0[e] = src[n * sn + (k0 + e) * sk];
    v1[e] = src[n * sn + (k0 + 4 + e) * sk];
  }
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
  __bf16* o = dst + ((size_t)(((u * KSTEPS + ks) * NT + j) * 3) * 64 + lane) * 8;
#pragma unroll
  for (int q = 0; q < 3; ++q) *reinterpret_cast<bf16x8*>(o + (size_t)q * 512) = pl[q];
}

// rows per workgroup.  Forward: 128 once the grid still fills the chip (B >= 1 536), else 32.  Backward: always 32 — its A
// operand (the dgates rows, K = 1 024) is private to a wave in the 128-row shape and is re-read by all eight unit blocks;
// measured at B = 2 048: 33.5 us per step (32 rows) vs 38.6 (128 rows) vs 37.8 (fp32 step); forward 28.4 vs 40.6 (fp32).
inline int split_row_groups(int B, bool fwd) { return fwd && B >= 1536 ? 4 : 1; }

template <bool FWD>
int launch_split_step(const StepArgs& p, int B, hipStream_t s) {
  constexpr int NT = FWD ? 4 : 1;
  if constexpr (FWD) if (split_row_groups(B, FWD) == 4) {
    constexpr int lds4 = 2 * 48 * 1024;
    auto kern = lstm_step_bf3_kernel<NT, FWD, 4>;
    static int attr_rc = (int)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
    if (attr_rc != (int)hipSuccess) {
      qea_set_error("qea_lstm_layer_*_split: cannot reserve %d bytes of LDS: %s", lds4, hipGetErrorString((hipError_t)attr_rc));
      return QEA_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((lstm_step_bf3_kernel<NT, FWD, 4>), dim3(qea_cdiv(B, 128), HID / 32, 2), dim3(256), lds4, s, p);
    return QEA_OK;
  }
  hipLaunchKernelGGL((lstm_step_bf3_kernel<NT, FWD, 1>), dim3(qea_cdiv(B, 32), HID / 32, 2), dim3(256), 4 * NT * 16 * 64 * sizeof(float), s, p);
  return QEA_OK;
}

}  // namespace

extern "C" int qea_lstm_pack_whh(const float* w_hh, float* packed_fwd, float* packed_bwd, void* stream) {
  QEA_REQUIRE(w_hh && (packed_fwd || packed_bwd), "qea_lstm_pack_whh: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)GATES * HID;
  // forward: Wt = W_hh [1024][256]; tiles j = gate (stride HID rows); K = 256 -> KW = 64
  if (packed_fwd) hipLaunchKernelGGL(pack_kernel, dim3(1024), dim3(256), 0, s, w_hh, packed_fwd, 4, 64, HID, (long long)HID, 1LL, total);
  // backward: Wt[n][k] = W_hh[k][n], n < 256, K = 1024 -> KW = 256, one tile
  if (packed_bwd) hipLaunchKernelGGL(pack_kernel, dim3(1024), dim3(256), 0, s, w_hh, packed_bwd, 1, 256, 0, 1LL, (long long)HID, total);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

static int layer_fwd_impl(float* gates, float* c, float* y, const float* packed_fwd, int32_t T, int32_t B, void* stream, bool split) {
  // gates [T][B][2*1024] (in: x*W_ih^T + b_ih + b_hh for both directions, out: gate activations)
  // c     [T][B][2*256], y [T][B][2*256]; packed_fwd: [2][1024*256] from qea_lstm_pack_whh
  QEA_REQUIRE(gates && c && y && packed_fwd && T > 0 && B > 0, "qea_lstm_layer_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  auto kern = lstm_step_kernel<4, 64, true>;
  const size_t lds = 4 * 4 * 16 * 64 * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const long long sg = (long long)B * 2 * GATES, sc = (long long)B * 2 * HID;
  for (int step = 0; step < T; ++step) {
    const int tf = step, tr = T - 1 - step;  // forward / reverse time index
    StepArgs p = {};
    p.B = B;
    p.skip_gemm = (step == 0);
    p.wp = packed_fwd;
    p.wp_dir = (long long)GATES * HID * (split ? 3 : 1);
    // forward direction pointers (d == 0) are the bases; reverse offsets are relative to them
    p.a = y + (long long)(tf - 1) * sc;  // h_prev (unused when step == 0)
    p.a_dir = HID;
    p.rev_a = (long long)(tr + 1 - (tf - 1)) * sc;
    p.lda = 2 * HID;
    p.gates = gates + (long long)tf * sg;
    p.rev_gates = (long long)(tr - tf) * sg;
    p.ldg = 2 * GATES;
    p.c_prev = c + (long long)(tf - 1) * sc;
    p.rev_c_prev = (long long)(tr + 1 - (tf - 1)) * sc;
    p.has_c_prev_fwd = p.has_c_prev_rev = (step > 0);
    p.c_out = c + (long long)tf * sc;
    p.rev_c_out = (long long)(tr - tf) * sc;
    p.ldc = 2 * HID;
    p.h_out = y + (long long)tf * sc;
    p.rev_h_out = (long long)(tr - tf) * sc;
    p.ldh = 2 * HID;
    qea_prof_begin(QEA_PROF_LSTM_STEP, s);
    int rc = QEA_OK;
    if (split) rc = launch_split_step<true>(p, B, s);
    else hipLaunchKernelGGL(kern, dim3(qea_cdiv(B, 32), HID / 32, 2), dim3(256), lds, s, p);
    qea_prof_end(QEA_PROF_LSTM_STEP, s, step ? 2.0 * 2 * B * (double)GATES * HID : 0.0, 0.0, split ? 1 : 0);
    if (rc != QEA_OK) return rc;
  }
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_lstm_layer_fwd(float* gates, float* c, float* y, const float* packed_fwd, int32_t T, int32_t B, void* stream) {
  return layer_fwd_impl(gates, c, y, packed_fwd, T, B, stream, false);
}

extern "C" int qea_lstm_layer_fwd_split(float* gates, float* c, float* y, const void* planes_fwd, int32_t T, int32_t B, void* stream) {
  return layer_fwd_impl(gates, c, y, (const float*)planes_fwd, T, B, stream, true);
}

static int layer_bwd_impl(float* gates, const float* c, const float* dy, const float* packed_bwd, float* dc_scratch, int32_t T,
                          int32_t B, void* stream, bool split) {
  // gates [T][B][2*1024]: in = saved activations, out = pre-activation gate gradients
  // c [T][B][512] saved cell states, dy [T][B][512] gradient of the layer output,
  // dc_scratch [B][512] floats.
  QEA_REQUIRE(gates && c && dy && packed_bwd && dc_scratch && T > 0 && B > 0, "qea_lstm_layer_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  auto kern = lstm_step_kernel<1, 256, false>;
  const size_t lds = 4 * 1 * 16 * 64 * sizeof(float);
  const long long sg = (long long)B * 2 * GATES, sc = (long long)B * 2 * HID;
  // processing step k handles time e_f = T-1-k (forward dir) and e_r = k (reverse dir);
  // the recurrent gradient comes from the step processed just before (time e_f+1 / e_r-1).
  for (int k = 0; k < T; ++k) {
    const int ef = T - 1 - k, er = k;
    StepArgs p = {};
    p.B = B;
    p.skip_gemm = (k == 0);
    p.wp = packed_bwd;
    p.wp_dir = (long long)GATES * HID * (split ? 3 : 1);
    p.a = gates + (long long)(ef + 1) * sg;  // dgates of the step processed before (valid when k > 0)
    p.a_dir = GATES;
    p.rev_a = (long long)((er - 1) - (ef + 1)) * sg;
    p.lda = 2 * GATES;
    p.ldg = 2 * GATES;
    p.ldc = 2 * HID;
    p.ldh = 2 * HID;
    p.dy = dy + (long long)ef * sc;
    p.rev_dy = (long long)(er - ef) * sc;
    p.dgates_e = gates + (long long)ef * sg;
    p.rev_dgates_e = (long long)(er - ef) * sg;
    p.c_e = c + (long long)ef * sc;
    p.rev_c_e = (long long)(er - ef) * sc;
    // predecessor in TIME of step e: forward dir -> ef-1, reverse dir -> er+1
    p.c_e_prev = c + (long long)(ef - 1) * sc;
    p.rev_c_e_prev = (long long)((er + 1) - (ef - 1)) * sc;
    p.has_c_e_prev_fwd = (ef > 0);
    p.has_c_e_prev_rev = (er < T - 1);
    p.dc = dc_scratch;
    p.dc_init = (k == 0);
    qea_prof_begin(QEA_PROF_LSTM_STEP, s);
    int rc = QEA_OK;
    if (split) rc = launch_split_step<false>(p, B, s);
    else hipLaunchKernelGGL(kern, dim3(qea_cdiv(B, 32), HID / 32, 2), dim3(256), lds, s, p);
    qea_prof_end(QEA_PROF_LSTM_STEP, s, k ? 2.0 * 2 * B * (double)GATES * HID : 0.0, 0.0, split ? 1 : 0);
    if (rc != QEA_OK) return rc;
  }
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_lstm_layer_bwd(float* gates, const float* c, const float* dy, const float* packed_bwd, float* dc_scratch, int32_t T,
                                  int32_t B, void* stream) {
  return layer_bwd_impl(gates, c, dy, packed_bwd, dc_scratch, T, B, stream, false);
}

extern "C" int qea_lstm_layer_bwd_split(float* gates, const float* c, const float* dy, const void* planes_bwd, float* dc_scratch, int32_t T,
                                        int32_t B, void* stream) {
  return layer_bwd_impl(gates, c, dy, (const float*)planes_bwd, dc_scratch, T, B, stream, true);
}

extern "C" size_t qea_lstm_pack_whh_split_bytes(void) { return (size_t)GATES * HID * 3 * sizeof(__bf16); }

extern "C" int qea_lstm_pack_whh_split(const float* w_hh, void* planes_fwd, void* planes_bwd, void* stream) {
  QEA_REQUIRE(w_hh && (planes_fwd || planes_bwd), "qea_lstm_pack_whh_split: null pointer");
  hipStream_t s = (hipStream_t)stream;
  // forward: Wt = W_hh [1024][256]: four gate tiles (rows j*256 + unit), K = 256 -> 16 k-steps; (8 unit blocks, 16, 4, 64 lanes)
  if (planes_fwd) hipLaunchKernelGGL(pack3_kernel, dim3(qea_cdiv(8 * 16 * 4 * 64, 256)), dim3(256), 0, s, w_hh, (__bf16*)planes_fwd, 4, 16, HID, (long long)HID, 1LL, 8 * 16 * 4 * 64);
  // backward: Wt[n][k] = W_hh[k][n], n < 256, K = 1024 -> 64 k-steps, one tile
  if (planes_bwd) hipLaunchKernelGGL(pack3_kernel, dim3(qea_cdiv(8 * 64 * 1 * 64, 256)), dim3(256), 0, s, w_hh, (__bf16*)planes_bwd, 1, 64, 0, 1LL, (long long)HID, 8 * 64 * 1 * 64);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}
