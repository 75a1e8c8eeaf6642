// A whole bidirectional LSTM layer in ONE launch (round 4): the time loop runs inside the kernel, the recurrent weights stay in LDS
// (nn.LSTM(512, 256, 2, bidirectional=True), reference models/model_crnn.py:9,19; one launch here = one layer, both directions).
//
// The per-step kernels of lstm.hip move the same W_hh slice (192 KB of bf16 planes per workgroup) and run the same prologue at
// every one of the 31 steps of a layer: at B = 2048 a forward step is 28.6 us for 5 us of MFMA work, at B = 32 a step is a launch.
// Here a workgroup owns (row block, 32 hidden units, direction) for ALL steps:
//   * its W_hh slice lives in LDS for the whole launch as TWO fp16 planes in MFMA-fragment order (128 KB; h is bounded by 1 and
//     W_hh by its abs-max, so the two-way fp16 split of common.h applies with fixed scales: three v_mfma_f32_32x32x16_f16 per
//     product instead of the six bf16 ones);
//   * c (forward) / dc (backward) never leave registers;
//   * the eight workgroups that share a row block and a direction (one per unit block) exchange h[t] (forward) or the gate
//     gradients (backward) through global memory with the write-through / sc1-load hand-off: every payload store is an sc1 store,
//     every storing wave drains (s_waitcnt vmcnt(0)), a workgroup barrier, ONE lane adds to the group's arrival counter (agent
//     scope); the consumer polls that counter from one wave, a workgroup barrier, then EVERY load of the exchanged bytes is a
//     buffer_load_dwordx4 sc1 (never served by this CU's L1).  The groups are laid out so that a group's eight workgroups are
//     dealt to one XCD under the round-robin dispatch (a speed bonus only: exchanged lines then come from that XCD's L2).
//   * nothing depends on residency of the whole grid: a group only waits for its own eight workgroups, whose block ids are
//     consecutive in their XCD's dispatch queue; every spin is bounded and sets a timeout word (the outputs are then poisoned
//     with NaN: a loud failure, never a hang).
//   * what is exchanged is the MFMA operand itself: the producer splits its 32 rows x 32 units of h (its 32 x 128 gate gradients)
//     ONCE into fp16 planes and stores them in A-FRAGMENT order ([k-step][plane][lane][8 halfs], 1 KB per store instruction), so the
//     eight consumers load fragments with whole-line 16-byte loads and spend no VALU on them (the first form, every consumer splitting
//     the fp32 rows for itself, was VALU-bound: 30 us per backward step at B = 2048).  The exchange area is double-buffered by step
//     parity (a producer two steps ahead has passed the wait every reader of that buffer arrived at after its loads).
// Backward: dh_rec = dgates[e_prev] * W_hh over K = 1024 in eight chunks = the eight producers' 128 columns each, every chunk split to
// fp16 by its producer with ONE scale per 32-row tile taken from the tile itself (the gate gradients have no a-priori bound; finer
// than the one-per-tensor scale of the convolutions): acc += (chunk product) * 2^-scale(row block, producer).  K runs in (producer,
// gate, half) order; qea_lstm_seq_pack orders W_hh^T to match.
// The passes also leave the abs-max of what they produce (layer output / gate gradients: per-wave running maxima, one atomic per wave
// at the end) for the GEMMs that read those tensors next.
//
// Roofline: HBM — per step and direction the gate tensor is read and written once (B x 1024 floats each way); the MFMA work is
// 2 * B * 1024 * 256 flops per direction and step (x 3 for the split).
#include "common.h"

namespace {

constexpr int HID = 256;
constexpr int GATES = 4 * HID;
constexpr int W_PLANES_HALFS = GATES * HID * 2;                 // two fp16 planes of one direction's W_hh
constexpr int W_SLICE_BYTES = GATES * HID * 2 * 2 / 8;          // one unit block's slice: 128 KB
constexpr unsigned SPIN_LIMIT = 1u << 22;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned int gu32;

// gate non-linearities on the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each): the library forms cost ~180 VALU
// instructions per (row, unit) element, 5 us per step of a 128-row workgroup with nothing to hide them under.  Absolute error
// <= ~1.5e-7 (tanh near 0 included: (1 - e) / (1 + e) cancels to an ABSOLUTE error of one rounding of e)
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __expf(-2.f * fabsf(x));
  return copysignf((1.f - e) * __builtin_amdgcn_rcpf(1.f + e), x);
}

// planes[u][ks][j][plane][lane][8] <- s * Wt[n][k], n = j*tile_stride + u*32 + (lane & 31), k = ks*16 + (lane >> 5)*8 + e
// perm (backward form): k-step ks = (producer u' = ks >> 3, gate j = (ks >> 1) & 3, half = ks & 1) covers columns j*256 + u'*32 + half*16 ..
__global__ void pack16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, const float* __restrict__ amax, int NT, int KSTEPS,
                              int tile_stride, long long sn, long long sk, int total, int perm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;            // (u, ks, j, lane)
  if (i >= total) return;
  float s, inv;
  qea_f16_scale(amax[0], s, inv);
  const int lane = i & 63;
  const int j = (i >> 6) % NT;
  const int ks = ((i >> 6) / NT) % KSTEPS;
  const int u = ((i >> 6) / NT) / KSTEPS;
  const int n = j * tile_stride + u * 32 + (lane & 31);
  const int k0 = (perm ? ((ks >> 1) & 3) * HID + (ks >> 3) * 32 + (ks & 1) * 16 : ks * 16) + (lane >> 5) * 8;
  f32x4 v0, v1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v0[e] = src[n * sn + (k0 + e) * sk];
    v1[e] = src[n * sn + (k0 + 4 + e) * sk];
  }
  f16x4 h0, l0, h1, l1;
  qea_split2_f16(v0, s, h0, l0);
  qea_split2_f16(v1, s, h1, l1);
  f16x8 ph, pl;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ph[k] = h0[k]; ph[k + 4] = h1[k];
    pl[k] = l0[k]; pl[k + 4] = l1[k];
  }
  _Float16* o = dst + ((size_t)(((u * KSTEPS + ks) * NT + j) * 2) * 64 + lane) * 8;
  *reinterpret_cast<f16x8*>(o) = ph;
  *reinterpret_cast<f16x8*>(o + 512) = pl;
}

constexpr int SEQ_NBUF4 = 3;                                     // chunk buffers of the 128-row backward (2 .. 6 measured: 818 / 819 / 899 / 1081 / 1371 us)

struct SeqArgs {
  float* gates;            // [T][B][2 * 1024]
  float* c;                // [T][B][2 * 256]
  float* y;                // [T][B][2 * 256]   forward: out; backward: unused
  const float* dy;         // backward: gradient of the layer output
  const _Float16* planes;  // [2 directions][W_PLANES_HALFS (+ trailer)]
  long long planes_dir;    // halfs between the directions
  const float* w_amax;     // [2] abs-max of the two W_hh
  unsigned* out_amax;      // NULL, or a zeroed slot: receives the largest finite |h| (forward) / |gate gradient| (backward) as float bits
  unsigned* sync;          // [n_groups] arrival counters ... [n_groups_padded] timeout word
  char* xch;               // exchange area: [parity 2][direction 2][row block of 32][XRB bytes]
  int tmo_index;
  int nrb;                 // row blocks of 32 (padded to the workgroup's row count)
  int T, B;
};

constexpr int XRB_FWD = 16 * 2048;                 // [k-step 16][plane 2][lane 64][16 B]
constexpr int XRB_BWD = 64 * 2048 + 8 * 128;       // [producer 8][k-step 8][plane 2][lane 64][16 B], then one inverse scale per producer (128 B apart)

// this wave's / workgroup's 32 x 32 tile (accumulator order in, NE values per lane) -> LDS tile [row][32] with the 16-byte slots of a
// row rotated by the row (T4[row * 8 + (slot ^ (row & 7))]): dword writes and the 16-byte fragment reads both spread over the banks
template <int NE>
__device__ __forceinline__ void tile_put(float* T, const float (&v)[NE], int w) {
  const int lane = threadIdx.x & 63, unit = lane & 31;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int lr = NE == 16 ? (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5) : 8 * w + i + 4 * (lane >> 5);
    T[lr * 32 + ((((unit >> 2) ^ (lr & 7)) << 2) | (unit & 3))] = v[i];
  }
}
// the A fragment of k-step kk (0 / 1: units 0-15 / 16-31 of the tile) for lane (row = lane & 31, k half = lane >> 5), split with scale s
__device__ __forceinline__ void tile_frag(const float* T, int kk, float s, f16x8& ah, f16x8& al) {
  const int lane = threadIdx.x & 63, row = lane & 31, c4 = kk * 4 + (lane >> 5) * 2;
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(T + row * 32 + ((c4 ^ (row & 7)) << 2));
  const f32x4 v1 = *reinterpret_cast<const f32x4*>(T + row * 32 + (((c4 + 1) ^ (row & 7)) << 2));
  f16x4 h0, l0, h1, l1;
  qea_split2_f16(v0, s, h0, l0);
  qea_split2_f16(v1, s, h1, l1);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ah[k] = h0[k]; ah[k + 4] = h1[k];
    al[k] = l0[k]; al[k + 4] = l1[k];
  }
}
__device__ __forceinline__ void store_frag_sc1(const f16x8& ah, const f16x8& al, __amdgpu_buffer_rsrc_t dst, int soff) {   // soff: wave-uniform
  const int voff = (threadIdx.x & 63) * 16;
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ah), dst, voff, soff, 16);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, al), dst, voff, soff + 1024, 16);
}

__device__ __forceinline__ void decode_block(int& group, int& u) {
  // blocks b and b + 8 share an XCD: the eight unit blocks of a group take eight consecutive places of one XCD's queue
  const int L = blockIdx.x, NG = gridDim.x >> 3;
  const int xcd = L & 7, slot = L >> 3, full = NG & ~7;
  if (slot < full) {
    group = (slot & ~7) + xcd;
    u = slot & 7;
  } else {
    group = slot;
    u = xcd;
  }
}

// one wave polls the group's counter until `want` arrivals; false (and the timeout word set) when the spin limit is hit
template <bool ACQ>
__device__ __forceinline__ bool wait_arrivals(unsigned* ctr, unsigned want, unsigned* tmo) {
  bool ok = true;
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > SPIN_LIMIT) {
        __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
    }
  }
  if constexpr (ACQ) {
    // plain loads behind ONE agent acquire (drops this CU's L1 lines): the fence, the fencing wave's wait, the workgroup barrier the
    // caller places next, then every wave's loads
    if (threadIdx.x < 64) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  return ok;
}

__device__ __forceinline__ void publish(unsigned* ctr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// non-negative floats order like their bit patterns: one atomic per wave into a zeroed slot
__device__ __forceinline__ void publish_max(unsigned* slot, float m) {
  if (slot == nullptr) return;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(slot, __float_as_uint(m));
}

template <int AUX>
__device__ __forceinline__ f32x4 load_x(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, AUX);   // aux 16 = sc1
  return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ void store_sc1(float v, __amdgpu_buffer_rsrc_t rsrc, int byte_off) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc, byte_off, 0, 16);
}

// This wave's NE values per lane of one 32-unit column block (MFMA accumulator order: unit = lane & 31, local row lr(i)) go through
// a wave-private LDS tile [rows][32] and leave as 16-byte stores, every 128-byte line written whole by one instruction (a dword per
// lane would be one fabric write each when the store is write-through: ~6x the time per byte).  aux 16 = sc1, 0 = plain.
template <int NE, int AUX>
__device__ __forceinline__ void store_tile(float* Tw, const float (&v)[NE], __amdgpu_buffer_rsrc_t dst, int row0, int B, int ld, int col0) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int lr = NE == 16 ? (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5) : i + 4 * (lane >> 5);
    Tw[lr * 32 + (lane & 31)] = v[i];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < NE / 4; ++q) {
    const int idx = q * 64 + lane, row = idx >> 3, c4 = idx & 7;
    const u32x4 x = *reinterpret_cast<const u32x4*>(Tw + row * 32 + c4 * 4);
    if (row0 + row < B) __builtin_amdgcn_raw_buffer_store_b128(x, dst, ((row0 + row) * ld + col0 + c4 * 4) * 4, 0, AUX);
  }
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void copy_slice_to_lds(char* lds, const _Float16* src) {
  const f32x4* s = reinterpret_cast<const f32x4*>(src);
  f32x4* d = reinterpret_cast<f32x4*>(lds);
#pragma unroll 8
  for (int i = threadIdx.x; i < W_SLICE_BYTES / 16; i += 256) d[i] = s[i];
}

// ---------------------------------------------------------------------------------------------
// forward.  RG = 4: 128 rows per workgroup, a wave owns 32 rows and all four gate tiles (gate math straight from its accumulators);
//           RG = 1: 32 rows per workgroup, wave w computes gate w over the full K, the four gates meet through 16 KB of LDS.
// LDS: W slice 128 KB | RG 1: X 16 KB | per-wave store tile (RG 4: 4 KB, RG 1: 1 KB) | RG 1: shared exchange tile 4 KB
// ---------------------------------------------------------------------------------------------
template <int RG, bool ACQ>
__global__ __launch_bounds__(256) void lstm_seq_fwd_kernel(const SeqArgs p) {
  constexpr int NT = RG == 4 ? 4 : 1;                 // gate tiles per wave
  constexpr int NE = RG == 4 ? 16 : 4;                // (row, unit) elements per lane
  constexpr int LDAUX = ACQ ? 0 : 16;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const Wl = lds;                                // [ks 16][j 4][plane 2][lane 64][16 B]
  float* const X = reinterpret_cast<float*>(lds + W_SLICE_BYTES);   // RG 1: [gate 4][r 16][lane 64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const Tw = reinterpret_cast<float*>(lds + W_SLICE_BYTES + (RG == 4 ? 0 : 16384)) + w * (NE * 64);   // wave-private store tile
  float* const Te = RG == 4 ? Tw : reinterpret_cast<float*>(lds + W_SLICE_BYTES + 16384 + 4096);               // exchange tile
  int group, u;
  decode_block(group, u);
  const int d = group & 1, rg = group >> 1;
  const int m0 = rg * (32 * RG) + (RG == 4 ? w * 32 : 0);
  const int rb = RG == 4 ? rg * 4 + w : rg;             // this wave's row block of 32
  const int trow0 = RG == 4 ? m0 : m0 + 8 * w;          // first row of this wave's epilogue elements
  const int T = p.T, B = p.B;
  unsigned* const ctr = p.sync + group;
  unsigned* const tmo = p.sync + p.tmo_index;

  copy_slice_to_lds(Wl, p.planes + (size_t)d * p.planes_dir + (size_t)u * (W_SLICE_BYTES / 2));
  float sw, inv_w;
  qea_f16_scale(p.w_amax[d], sw, inv_w);
  const float sh = 16384.f, out_scale = inv_w * (1.f / 16384.f);   // |h| < 1: h * 2^14 in fp16 range

  const int unit = u * 32 + (lane & 31);
  int erow[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int r = RG == 4 ? i : 4 * w + i;
    erow[i] = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  }
  float creg[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) creg[i] = 0.f;
  float run_max = 0.f;                                 // the layer output's abs-max, carried by its producer (one atomic per wave at the end)
  bool failed = false;
  __syncthreads();                                     // the weight slice is in LDS

  // x * W_ih^T + b of a step is loaded one step ahead: behind the arrival of the step before, in front of that step's own stores of
  // y / activations / c, so that the reads and the writes of a step overlap in HBM
  float gx[NE][4];
  int voffg[NE];                                        // byte offset of (row, direction, unit) inside one time step of gates (32-bit: one
#pragma unroll                                          // register per element instead of a hoisted 64-bit address per element and array)
  for (int i = 0; i < NE; ++i) voffg[i] = (min(erow[i], B - 1) * (2 * GATES) + d * GATES + unit) * 4;
  auto load_gx = [&](int t) {
    const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(p.gates + (size_t)t * B * (2 * GATES), 0, B * (2 * GATES) * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < NE; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) gx[i][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gsrc, voffg[i], j * HID * 4, 0));
  };
  load_gx(d ? T - 1 : 0);

  for (int step = 0; step < T; ++step) {
    const int t = d ? T - 1 - step : step;

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    if (step > 0) {
      if (!failed && !wait_arrivals<ACQ>(ctr, 8u * step, tmo)) failed = true;   // after one timeout the polling lane stops waiting: the launch ends at once, NaN in its outputs
      __syncthreads();
      const __amdgpu_buffer_rsrc_t xsrc =
          __builtin_amdgcn_make_buffer_rsrc(p.xch + ((size_t)((((step - 1) & 1) * 2 + d) * p.nrb + rb)) * XRB_FWD, 0, XRB_FWD, 0x00020000);
      u32x4 a[16][2];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        a[ks][0] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, lane * 16, ks * 2048, LDAUX);        // the k-step in the scalar offset
        a[ks][1] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, lane * 16, ks * 2048 + 1024, LDAUX);
      }
      __builtin_amdgcn_sched_barrier(0);               // all 32 fragment loads in flight before the first MFMA (left alone, hipcc sinks
                                                       // every load next to its use: one exposed round trip per fragment)
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const f16x8 ah = __builtin_bit_cast(f16x8, a[ks][0]), al = __builtin_bit_cast(f16x8, a[ks][1]);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int jj = RG == 4 ? j : w;
          const char* bp = Wl + (size_t)((ks * 4 + jj) * 2) * 1024 + lane * 16;
          const f16x8 bh = *reinterpret_cast<const f16x8*>(bp);
          const f16x8 bl = *reinterpret_cast<const f16x8*>(bp + 1024);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
        }
      }
    }
    if constexpr (RG == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) X[(w * 16 + r) * 64 + lane] = acc[0][r];
      __syncthreads();
    }
    float hv[NE], go_[4][NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      float s[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] = (RG == 4 ? acc[RG == 4 ? j : 0][i] : X[(j * 16 + 4 * w + i) * 64 + lane]) * out_scale;
      const float gi = sigmoid_fast(gx[i][0] + s[0]);
      const float gf = sigmoid_fast(gx[i][1] + s[1]);
      const float gg = tanh_fast(gx[i][2] + s[2]);
      const float go = sigmoid_fast(gx[i][3] + s[3]);
      const float cc = gf * creg[i] + gi * gg;
      creg[i] = cc;
      hv[i] = failed ? __uint_as_float(0x7fc00000u) : go * tanh_fast(cc);
      if (erow[i] < B) run_max = fmaxf(run_max, fabsf(hv[i]) <= 3.4028234e38f ? fabsf(hv[i]) : 0.f);
      go_[0][i] = gi; go_[1][i] = gf; go_[2][i] = gg; go_[3][i] = go;
    }
    // the exchange first: h of this tile as the next step's A fragments (k-steps 2u, 2u + 1 of the row block), write-through, and the
    // arrival right behind it; y, the saved activations and c are nobody's input inside this launch
    if (step + 1 < T) {
      const __amdgpu_buffer_rsrc_t xdst =
          __builtin_amdgcn_make_buffer_rsrc(p.xch + ((size_t)(((step & 1) * 2 + d) * p.nrb + rb)) * XRB_FWD, 0, XRB_FWD, 0x00020000);
      tile_put<NE>(Te, hv, w);
      if constexpr (RG == 4) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          f16x8 ah, al;
          tile_frag(Te, kk, sh, ah, al);
          store_frag_sc1(ah, al, xdst, (2 * u + kk) * 2048);
        }
        __builtin_amdgcn_wave_barrier();
      } else {
        __syncthreads();
        if (w < 2) {
          f16x8 ah, al;
          tile_frag(Te, w, sh, ah, al);
          store_frag_sc1(ah, al, xdst, (2 * u + w) * 2048);
        }
      }
      publish(ctr);
      load_gx(d ? t - 1 : t + 1);
    }
    const __amdgpu_buffer_rsrc_t hdst = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)t * B * (2 * HID), 0, B * (2 * HID) * 4, 0x00020000);
    store_tile<NE, 0>(Tw, hv, hdst, trow0, B, 2 * HID, d * HID + u * 32);
    const __amdgpu_buffer_rsrc_t gdst = __builtin_amdgcn_make_buffer_rsrc(p.gates + (size_t)t * B * (2 * GATES), 0, B * (2 * GATES) * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j) store_tile<NE, 0>(Tw, go_[j], gdst, trow0, B, 2 * GATES, d * GATES + j * HID + u * 32);
    const __amdgpu_buffer_rsrc_t cdst = __builtin_amdgcn_make_buffer_rsrc(p.c + (size_t)t * B * (2 * HID), 0, B * (2 * HID) * 4, 0x00020000);
    store_tile<NE, 0>(Tw, creg, cdst, trow0, B, 2 * HID, d * HID + u * 32);
  }
  publish_max(p.out_amax, run_max);
}

// ---------------------------------------------------------------------------------------------
// backward.  Processing step k handles time e = T-1-k (forward direction) / e = k (reverse direction); its recurrent gradient
// comes from the gate gradients of the step processed just before, chunk by chunk = producer by producer.
// RG = 4: 128 rows per workgroup, a wave owns 32 rows over the whole K = 1024 (8 chunks);
// RG = 1: 32 rows per workgroup, the four waves take two chunks each and meet through 16 KB of LDS.
// LDS: W^T slice 128 KB | RG 1: X 16 KB | per-wave store tile | RG 1: shared exchange tile 4 KB | row abs-max 128 B (per wave at RG 4)
// ---------------------------------------------------------------------------------------------
template <int RG, bool ACQ>
__global__ __launch_bounds__(256) void lstm_seq_bwd_kernel(const SeqArgs p) {
  constexpr int NCH = RG == 4 ? 8 : 2;                // chunks (producers) per wave
  constexpr int NE = RG == 4 ? 16 : 4;
  constexpr int NBUF = RG == 4 ? SEQ_NBUF4 : 2;               // chunk buffers (NBUF - 1 in flight ahead of the multiply)
  constexpr int LDAUX = ACQ ? 0 : 16;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const Wl = lds;                                // [ks 64][plane 2][lane 64][16 B]
  float* const X = reinterpret_cast<float*>(lds + W_SLICE_BYTES);   // RG 1: [wave 4][r 16][lane 64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const Tw = reinterpret_cast<float*>(lds + W_SLICE_BYTES + (RG == 4 ? 0 : 16384)) + w * (NE * 64);
  float* const Te = RG == 4 ? Tw : reinterpret_cast<float*>(lds + W_SLICE_BYTES + 16384 + 4096);
  float* const Sm = reinterpret_cast<float*>(lds + W_SLICE_BYTES + (RG == 4 ? 16384 + w * 128 : 16384 + 4096 + 4096));   // [32 rows] abs-max
  int group, u;
  decode_block(group, u);
  const int d = group & 1, rg = group >> 1;
  const int m0 = rg * (32 * RG) + (RG == 4 ? w * 32 : 0);
  const int rb = RG == 4 ? rg * 4 + w : rg;
  const int trow0 = RG == 4 ? m0 : m0 + 8 * w;
  const int T = p.T, B = p.B;
  unsigned* const ctr = p.sync + group;
  unsigned* const tmo = p.sync + p.tmo_index;

  copy_slice_to_lds(Wl, p.planes + (size_t)d * p.planes_dir + (size_t)u * (W_SLICE_BYTES / 2));
  float sw, inv_w;
  qea_f16_scale(p.w_amax[d], sw, inv_w);

  const int unit = u * 32 + (lane & 31);
  int erow[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int r = RG == 4 ? i : 4 * w + i;
    erow[i] = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  }
  float run_max = 0.f;                                  // the gate gradients' abs-max over all steps (one atomic per wave at the end)
  float dcreg[NE];
  int voffg[NE], voffc[NE];                             // byte offsets of (row, direction, unit) inside one time step of gates / of c, dy
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    dcreg[i] = 0.f;
    voffg[i] = (min(erow[i], B - 1) * (2 * GATES) + d * GATES + unit) * 4;
    voffc[i] = (min(erow[i], B - 1) * (2 * HID) + d * HID + unit) * 4;
  }
  const int ch0 = RG == 4 ? 0 : 2 * w;                 // this wave's first chunk
  bool failed = false;
  __syncthreads();

  for (int k = 0; k < T; ++k) {
    const int e = d ? k : T - 1 - k;
    const int et = d ? e + 1 : e - 1;                   // predecessor of e in time
    const bool has_prev = d ? (e < T - 1) : (e > 0);
    float ga[NE][4], dyv[NE], cv[NE], cpv[NE];
    auto load_operands = [&]() {                        // the gate backward's inputs of step e: nobody's output in this launch
      const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(p.gates + (size_t)e * B * (2 * GATES), 0, B * (2 * GATES) * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t dsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy) + (size_t)e * B * (2 * HID), 0, B * (2 * HID) * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t csrc = __builtin_amdgcn_make_buffer_rsrc(p.c + (size_t)e * B * (2 * HID), 0, B * (2 * HID) * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t psrc = __builtin_amdgcn_make_buffer_rsrc(p.c + (size_t)(has_prev ? et : e) * B * (2 * HID), 0, B * (2 * HID) * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < NE; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ga[i][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gsrc, voffg[i], j * HID * 4, 0));
        dyv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dsrc, voffc[i], 0, 0));
        cv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(csrc, voffc[i], 0, 0));
        const float cp = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(psrc, voffc[i], 0, 0));
        cpv[i] = has_prev ? cp : 0.f;
      }
    };

    f32x16 tot;
#pragma unroll
    for (int r = 0; r < 16; ++r) tot[r] = 0.f;
    if (k == 0) {
      load_operands();
    } else {
      if constexpr (RG == 1) load_operands();           // 28 registers: ahead of the wait
      if (!failed && !wait_arrivals<ACQ>(ctr, 8u * k, tmo)) failed = true;
      __syncthreads();
      const __amdgpu_buffer_rsrc_t xsrc =
          __builtin_amdgcn_make_buffer_rsrc(p.xch + ((size_t)((((k - 1) & 1) * 2 + d) * p.nrb + rb)) * XRB_BWD, 0, XRB_BWD, 0x00020000);
      u32x4 a[NBUF][8][2];
      float inv[NBUF];
      auto load_chunk = [&](int ch, int b) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          a[b][q][0] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, lane * 16, (ch * 8 + q) * 2048, LDAUX);
          a[b][q][1] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, lane * 16, (ch * 8 + q) * 2048 + 1024, LDAUX);
        }
        inv[b] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xsrc, 0, 64 * 2048 + ch * 128, LDAUX));
      };
      // NBUF - 1 chunks (16 KB per wave each) in flight ahead of the one being multiplied; the schedule is pinned (left alone, hipcc
      // sinks every load next to its use: one exposed round trip per fragment, 35 us per step)
#pragma unroll
      for (int c = 0; c < NBUF - 1 && c < NCH; ++c) load_chunk(ch0 + c, c);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int b = c % NBUF;
        if (c + NBUF - 1 < NCH) load_chunk(ch0 + c + NBUF - 1, (c + NBUF - 1) % NBUF);
        if (RG == 4 && c == NCH - 1) load_operands();   // RG 4: 112 registers, once the chunk buffers start to free up
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const f16x8 ah = __builtin_bit_cast(f16x8, a[b][q][0]), al = __builtin_bit_cast(f16x8, a[b][q][1]);
          const char* bp = Wl + (size_t)(((ch0 + c) * 8 + q) * 2) * 1024 + lane * 16;
          const f16x8 bh = *reinterpret_cast<const f16x8*>(bp);
          const f16x8 bl = *reinterpret_cast<const f16x8*>(bp + 1024);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        }
        const float iv = inv[b] * inv_w;                // one scale per producer tile
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[r] += acc[r] * iv;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (RG == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) X[(w * 16 + r) * 64 + lane] = tot[r];
      __syncthreads();
    }
    float o_[4][NE];
    float tmax = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      float s;
      if constexpr (RG == 4) {
        s = tot[i];
      } else {
        const int r = 4 * w + i;
        s = ((X[(0 * 16 + r) * 64 + lane] + X[(1 * 16 + r) * 64 + lane]) + X[(2 * 16 + r) * 64 + lane]) + X[(3 * 16 + r) * 64 + lane];
      }
      const float gi = ga[i][0], gf = ga[i][1], gg = ga[i][2], go = ga[i][3];
      const float dh = dyv[i] + s;
      const float tc = tanh_fast(cv[i]);
      const float dc = dh * go * (1.f - tc * tc) + dcreg[i];
      o_[0][i] = dc * gg * gi * (1.f - gi);
      o_[1][i] = dc * cpv[i] * gf * (1.f - gf);
      o_[2][i] = dc * gi * (1.f - gg * gg);
      o_[3][i] = dh * tc * go * (1.f - go);
      dcreg[i] = dc * gf;
      float emax = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float av = fabsf(o_[j][i]);
        emax = fmaxf(emax, av <= 3.4028234e38f ? av : 0.f);
      }
      tmax = fmaxf(tmax, emax);
      if (erow[i] < B) run_max = fmaxf(run_max, emax);
      if (failed) o_[0][i] = o_[1][i] = o_[2][i] = o_[3][i] = __uint_as_float(0x7fc00000u);
    }
    if (k + 1 < T) {
      // ONE fp16 scale per producer tile (32 rows x 4 x 32 gate gradients): its abs-max over the wave (RG 1: over the four waves)
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o, 64));
      const __amdgpu_buffer_rsrc_t xdst =
          __builtin_amdgcn_make_buffer_rsrc(p.xch + ((size_t)(((k & 1) * 2 + d) * p.nrb + rb)) * XRB_BWD, 0, XRB_BWD, 0x00020000);
      if constexpr (RG == 4) {
        float sa, inv_a;
        qea_f16_scale(tmax, sa, inv_a);
        if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(inv_a), xdst, 64 * 2048 + u * 128, 0, 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          tile_put<NE>(Te, o_[j], w);
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            f16x8 ah, al;
            tile_frag(Te, kk, sa, ah, al);
            store_frag_sc1(ah, al, xdst, (u * 8 + j * 2 + kk) * 2048);
          }
          __builtin_amdgcn_wave_barrier();
        }
      } else {
        // the X area is free once every wave has summed its partials: it takes the four gate tiles (4 KB each); wave w then splits
        // and stores the two k-steps of gate w
        if (lane == 0) Sm[w] = tmax;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) tile_put<NE>(X + j * 1024, o_[j], w);
        __syncthreads();
        float sa, inv_a;
        qea_f16_scale(fmaxf(fmaxf(Sm[0], Sm[1]), fmaxf(Sm[2], Sm[3])), sa, inv_a);
        if (tid == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(inv_a), xdst, 64 * 2048 + u * 128, 0, 16);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          f16x8 ah, al;
          tile_frag(X + w * 1024, kk, sa, ah, al);
          store_frag_sc1(ah, al, xdst, (u * 8 + w * 2 + kk) * 2048);
        }
      }
      publish(ctr);
    }
    const __amdgpu_buffer_rsrc_t gdst = __builtin_amdgcn_make_buffer_rsrc(p.gates + (size_t)e * B * (2 * GATES), 0, B * (2 * GATES) * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j) store_tile<NE, 0>(Tw, o_[j], gdst, trow0, B, 2 * GATES, d * GATES + j * HID + u * 32);
  }
  publish_max(p.out_amax, run_max);
}

// exchanged rows through L1 behind an agent acquire (true) or by sc1 loads (false), per kernel shape
constexpr bool SEQ_ACQ_FWD4 = false, SEQ_ACQ_FWD1 = false, SEQ_ACQ_BWD4 = false, SEQ_ACQ_BWD1 = false;

// 128-row workgroups once the 32-row ones would not all be resident at once; the backward's 128-row shape (512 KB of fragments per
// workgroup and step) only pays above 1024 rows (B = 1024: 652 us against 2 x 278 for two rounds of 32-row workgroups)
inline int seq_row_groups(int B, bool bwd) { return B > (bwd ? 1024 : 512) ? 4 : 1; }
inline int seq_groups(int B, bool bwd) { return qea_cdiv(B, 32 * seq_row_groups(B, bwd)) * 2; }
inline size_t seq_sync_words(int B, bool bwd) { return (size_t)((seq_groups(B, bwd) + 1 + 63) & ~63); }    // the exchange area starts on a 256-byte line
inline int seq_nrb(int B, bool bwd) { return qea_cdiv(B, 32 * seq_row_groups(B, bwd)) * seq_row_groups(B, bwd); }
inline size_t seq_xch_bytes(int B, bool bwd) { return (size_t)2 * 2 * seq_nrb(B, bwd) * (bwd ? XRB_BWD : XRB_FWD); }

template <void (*KERN)(const SeqArgs), bool BWD>
int seq_launch(const char* what, SeqArgs& a, int B, void* ws, hipStream_t s, int lds) {
  const int ng = seq_groups(B, BWD);
  a.sync = (unsigned*)ws;
  a.xch = (char*)ws + seq_sync_words(B, BWD) * 4;
  a.nrb = seq_nrb(B, BWD);
  a.tmo_index = ng;
  static int attr_rc = (int)hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  int rc = attr_rc;
  if (rc != (int)hipSuccess) {
    qea_set_error("%s: cannot reserve %d bytes of LDS: %s", what, lds, hipGetErrorString((hipError_t)rc));
    return QEA_ERR_LAUNCH;
  }
  rc = (int)hipMemsetAsync(ws, 0, seq_sync_words(B, BWD) * 4, s);
  if (rc != (int)hipSuccess) {
    qea_set_error("%s: hipMemsetAsync: %s", what, hipGetErrorString((hipError_t)rc));
    return QEA_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(KERN, dim3(ng * 8), dim3(256), lds, s, a);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

}  // namespace

extern "C" size_t qea_lstm_seq_pack_bytes(void) { return (size_t)W_PLANES_HALFS * 2; }

extern "C" size_t qea_lstm_seq_workspace_bytes(int32_t B) {   // enough for either pass
  if (B <= 0) return 0;
  const size_t f = seq_sync_words(B, false) * 4 + seq_xch_bytes(B, false), b = seq_sync_words(B, true) * 4 + seq_xch_bytes(B, true);
  return f > b ? f : b;
}

extern "C" int qea_lstm_seq_pack(const float* w_hh, void* planes_fwd, void* planes_bwd, const float* w_absmax, void* stream) {
  QEA_REQUIRE(w_hh && w_absmax && (planes_fwd || planes_bwd), "qea_lstm_seq_pack: null pointer");
  hipStream_t s = (hipStream_t)stream;
  // forward: Wt = W_hh [1024][256]: four gate tiles (rows j*256 + unit), K = 256 -> 16 k-steps
  if (planes_fwd) hipLaunchKernelGGL(pack16_kernel, dim3(qea_cdiv(8 * 16 * 4 * 64, 256)), dim3(256), 0, s, w_hh, (_Float16*)planes_fwd, w_absmax, 4, 16, HID, (long long)HID, 1LL, 8 * 16 * 4 * 64, 0);
  // backward: Wt[n][k] = W_hh[k][n], n < 256, K = 1024 -> 64 k-steps in (producer, gate, half) order, one tile
  if (planes_bwd) hipLaunchKernelGGL(pack16_kernel, dim3(qea_cdiv(8 * 64 * 1 * 64, 256)), dim3(256), 0, s, w_hh, (_Float16*)planes_bwd, w_absmax, 1, 64, 0, 1LL, (long long)HID, 8 * 64 * 1 * 64, 1);
  QEA_CHECK_LAUNCH();
  return QEA_OK;
}

extern "C" int qea_lstm_seq_fwd(float* gates, float* c, float* y, const void* planes_fwd, const float* w_absmax, int32_t T, int32_t B, void* workspace,
                                float* y_absmax, void* stream) {
  QEA_REQUIRE(gates && c && y && planes_fwd && w_absmax && workspace && T > 0 && B > 0, "qea_lstm_seq_fwd: bad arguments");
  QEA_REQUIRE((long long)B * (2 * GATES) * 4 < 0x7fffffffLL, "qea_lstm_seq_fwd: B too large for one buffer descriptor per time step");
  SeqArgs a = {};
  a.gates = gates; a.c = c; a.y = y;
  a.out_amax = (unsigned*)y_absmax;
  a.planes = (const _Float16*)planes_fwd;
  a.planes_dir = W_PLANES_HALFS;
  a.w_amax = w_absmax;
  a.T = T; a.B = B;
  hipStream_t s = (hipStream_t)stream;
  qea_prof_begin(QEA_PROF_LSTM_STEP, s);
  int rc;
  if (seq_row_groups(B, false) == 4) rc = seq_launch<lstm_seq_fwd_kernel<4, SEQ_ACQ_FWD4>, false>("qea_lstm_seq_fwd", a, B, workspace, s, W_SLICE_BYTES + 16384);
  else rc = seq_launch<lstm_seq_fwd_kernel<1, SEQ_ACQ_FWD1>, false>("qea_lstm_seq_fwd", a, B, workspace, s, W_SLICE_BYTES + 16384 + 4096 + 4096);
  qea_prof_end(QEA_PROF_LSTM_STEP, s, 2.0 * 2 * B * (double)GATES * HID * (T - 1), 0.0, 1);
  return rc;
}

extern "C" int qea_lstm_seq_bwd(float* gates, const float* c, const float* dy, const void* planes_bwd, const float* w_absmax, int32_t T, int32_t B,
                                void* workspace, float* dgates_absmax, void* stream) {
  QEA_REQUIRE(gates && c && dy && planes_bwd && w_absmax && workspace && T > 0 && B > 0, "qea_lstm_seq_bwd: bad arguments");
  QEA_REQUIRE((long long)B * (2 * GATES) * 4 < 0x7fffffffLL, "qea_lstm_seq_bwd: B too large for one buffer descriptor per time step");
  SeqArgs a = {};
  a.gates = gates; a.c = const_cast<float*>(c); a.dy = dy;
  a.out_amax = (unsigned*)dgates_absmax;
  a.planes = (const _Float16*)planes_bwd;
  a.planes_dir = W_PLANES_HALFS;
  a.w_amax = w_absmax;
  a.T = T; a.B = B;
  hipStream_t s = (hipStream_t)stream;
  qea_prof_begin(QEA_PROF_LSTM_STEP, s);
  int rc;
  if (seq_row_groups(B, true) == 4) rc = seq_launch<lstm_seq_bwd_kernel<4, SEQ_ACQ_BWD4>, true>("qea_lstm_seq_bwd", a, B, workspace, s, W_SLICE_BYTES + 16384 + 512);
  else rc = seq_launch<lstm_seq_bwd_kernel<1, SEQ_ACQ_BWD1>, true>("qea_lstm_seq_bwd", a, B, workspace, s, W_SLICE_BYTES + 16384 + 4096 + 4096 + 128);
  qea_prof_end(QEA_PROF_LSTM_STEP, s, 2.0 * 2 * B * (double)GATES * HID * (T - 1), 0.0, 1);
  return rc;
}

