#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(short4v* out) {
  __shared__ short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  // lane i loads the 4 contiguous shorts at element offset 4*i
  short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(lds + threadIdx.x * 4));
  out[threadIdx.x] = v;
}
int main() {
  short4v* d; hipMalloc(&d, 64 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short4v h[64]; hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l][0], h[l][1], h[l][2], h[l][3]);
  return 0;
}
