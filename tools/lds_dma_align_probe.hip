// Probe (GPU box): does `buffer_load_dwordx4 ... lds` accept per-lane addresses that are only 4-byte aligned?
//   hipcc --offload-arch=gfx950 -I include -I semi-supervised-adaptive-distillation_amd/csrc/kernels \
//       tools/lds_dma_align_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "conv_internal.h"

__global__ void probe(const float* src, int n, int shift, int stride, float* out) {
  __shared__ float lds[256];
  const int lane = threadIdx.x;
  ssad_dev::rsrc_words rs = ssad_dev::uniform_rsrc_words(src, (unsigned)n * 4);
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  ssad_dev::lds_dma<16>(rs, base, (unsigned)((shift + lane * stride) * 4), 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}

int main() {
  const int n = 4096;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 256 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  int bad_total = 0;
  for (int stride : {4, 5, 7, 18})
    for (int shift = 0; shift < 4; ++shift) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, n, shift, stride, o);
      std::vector<float> r(256);
      hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) bad += r[l * 4 + i] != (float)(shift + l * stride + i);
      printf("stride %d shift %d: %d wrong of 256 (lane 1 got %g %g %g %g)\n", stride, shift, bad, r[4], r[5], r[6], r[7]);
      bad_total += bad;
    }
  // range check at the end of the descriptor: lane whose 16 bytes straddle the end
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 254, 0, 4, o);
  std::vector<float> r(256);
  hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
  printf("descriptor of 254 floats, lane 63 (floats 252..255): %g %g %g %g\n", r[252], r[253], r[254], r[255]);
  printf("TOTAL wrong %d\n", bad_total);
  return 0;
}
