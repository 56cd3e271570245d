// lds_rate_probe.hip -- LDS read throughput per CU by instruction (8 waves per CU, conflict-free addresses):
// ds_read_b32 / b64 / b128 and gfx950's transpose read ds_read_b64_tr_b16.
//   hipcc -O3 --offload-arch=gfx950 tools/lds_rate_probe.hip -o /tmp/ldsrate && /tmp/ldsrate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int KIND>
__global__ __launch_bounds__(512, 1) void loop(float* out, int iters) {
  __shared__ float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 0.001f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bytes = KIND == 0 ? 4 : KIND == 2 ? 16 : 8;
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + wave * 4096 + lane * bytes;
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef float f4 __attribute__((ext_vector_type(4)));
  float a0 = 0; f2 b0 = {0, 0}; f4 c0 = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (KIND == 0) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a0) : "v"(addr), "n"(k * 256));
      if (KIND == 1) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b0) : "v"(addr), "n"(k * 512 % 4096));
      if (KIND == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c0) : "v"(addr), "n"(k * 1024 % 4096));
      if (KIND == 3) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b0) : "v"(addr), "n"(k * 512 % 4096));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (a0 + b0[0] + c0[0] == 123.456f) out[0] = 1.0f;
}

template <int KIND>
void run(const char* name, int bytes) {
  float* out;
  hipMalloc(&out, 4);
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount, iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(loop<KIND>, dim3(cus), dim3(512), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(loop<KIND>, dim3(cus), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double total = (double)iters * 16 * 512 * bytes;           // bytes per CU
  printf("%-22s %7.3f ms  %6.1f bytes per clock per CU (2.4 GHz)\n", name, ms, total / (ms * 1e-3 * 2.4e9));
  hipFree(out);
}

int main() {
  run<0>("ds_read_b32", 4);
  run<1>("ds_read_b64", 8);
  run<2>("ds_read_b128", 16);
  run<3>("ds_read_b64_tr_b16", 8);
  return 0;
}
