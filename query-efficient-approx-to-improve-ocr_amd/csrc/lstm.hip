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
  const float* wp;  // packed weights
  long long wp_dir;
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
      // s[0] = dh_rec for (row, unit); gate backward of step e
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

extern "C" int qea_lstm_layer_fwd(float* gates, float* c, float* y, const float* packed_fwd, int32_t T, int32_t B, void* stream) {
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
    p.wp_dir = (long long)GATES * HID;
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
    hipLaunchKernelGGL(kern, dim3(qea_cdiv(B, 32), HID / 32, 2), dim3(256), lds, s, p);
    qea_prof_end(QEA_PROF_LSTM_STEP, s, step ? 2.0 * 2 * B * (double)GATES * HID : 0.0, 0.0);
  }
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_lstm_layer_bwd(float* gates, const float* c, const float* dy, const float* packed_bwd, float* dc_scratch, int32_t T,
                                  int32_t B, void* stream) {
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
    p.wp_dir = (long long)GATES * HID;
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
    hipLaunchKernelGGL(kern, dim3(qea_cdiv(B, 32), HID / 32, 2), dim3(256), lds, s, p);
    qea_prof_end(QEA_PROF_LSTM_STEP, s, k ? 2.0 * 2 * B * (double)GATES * HID : 0.0, 0.0);
  }
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}
