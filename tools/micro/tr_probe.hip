// Developer probe: what does ds_read_b64_tr_b16 deliver?  LDS image [row][64 cols] with value row*100+col (as int16);
// lane 4q+p of each 16-lane group addresses row (4*group + q), columns 4p..4p+3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)((i / 64) * 100 + (i % 64));
  __syncthreads();
  const int l = threadIdx.x, grp = l / 16, q = (l % 16) / 4, p = l % 4;
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(lds + (4 * grp + q) * 64 + 4 * p));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
