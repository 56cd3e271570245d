// mfma_peak.hip -- what the matrix cores deliver with nothing else in the way: a register-only loop of
// independent MFMAs on every SIMD of the chip (2 waves per SIMD), in a short burst and sustained.
// The roofline fractions in bench.py are quoted against the data-sheet peaks (157.3 TF/s fp32,
// 2.5 PF/s fp16 at 2.4 GHz); this is the ceiling the clocks actually held allow.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int F16>
__global__ __launch_bounds__(256, 2) void mfma_loop(float* out, int iters) {
  f16v acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  half8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f); b[e] = (_Float16)(e * 0.01f); }
  const float fa = threadIdx.x * 0.001f, fb = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (F16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}

template <int F16>
void run(const char* name, double flops_per_mfma, double peak) {
  float* out;
  hipMalloc(&out, 4);
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount, iters = 4096;
  const dim3 grid(cus * 2), block(256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double fl = (double)grid.x * 4 * iters * 8 * flops_per_mfma;
  for (int launches : {1, 1, 20, 400}) {
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(mfma_loop<F16>, grid, block, 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double tf = fl * launches / ms / 1e9;
    printf("%s: %3d launches %8.2f ms  %7.1f TF/s = %.3f of %.1f (implied MFMA clock %.2f GHz)\n", name, launches, ms,
           tf, tf / peak, peak, tf / peak * 2.4);
  }
}

int main() {
  run<1>("f16 32x32x16", 2.0 * 32 * 32 * 16, 2500.0);
  run<0>("f32 32x32x2 ", 2.0 * 32 * 32 * 2, 157.3);
  return 0;
}
