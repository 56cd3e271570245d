// Lane -> element map of gfx950's LDS transpose read (ds_read_b64_tr_b16), printed from the
// device: lane i of a 16-lane group loads the 4 contiguous 16-bit elements at element offset
// 4 i; lane l receives elements (l & 15) + 16 j (+ 64 per group), j = 0..3 -- i.e. output
// element j of lane l comes from source lane 4 j + (l & 15) / 4, element (l & 15) % 4.
// conv3x3_wgrad_f16_kernel's operand addressing is built on this map.
//   hipcc -O3 --offload-arch=gfx950 tools/tr16_probe.hip -o /tmp/tr16_probe && /tmp/tr16_probe
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
