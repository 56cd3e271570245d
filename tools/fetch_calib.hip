// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns this
// repo's kernels use (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own
// access pattern before trusting an absolute").  Each kernel streams a 1 GiB buffer exactly once:
//   calib_lds_dma<4>    buffer_load_dword  ... lds  (4 B per lane: the Winograd kernels' patch staging)
//   calib_lds_dma<16>   buffer_load_dwordx4 ... lds (16 B per lane: the GEMM / fp16 kernels' staging)
//   calib_load<float4>  global 16-byte loads (the loss kernels)
//   calib_load<float>   global 4-byte loads
// FETCH_SIZE (KB) per dispatch / 1048576 KB = the factor tools/make_pmc_json.py divides by.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_calib -o t -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((address_space(3))) void* lds_ptr;
constexpr long long kBytes = 1LL << 30;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// every workgroup owns a contiguous 1 MiB; each wave-instruction moves 64 lanes x SIZE bytes into LDS
template <int SIZE>
__global__ __launch_bounds__(256) void calib_lds_dma(const char* x, float* out) {
  __shared__ __attribute__((aligned(16))) char lds[4 * 64 * SIZE];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = x + (long long)blockIdx.x * (1 << 20);
  const __amdgpu_buffer_rsrc_t rs = rsrc(base, 1u << 20);
  for (int off = wave * 64 * SIZE; off < (1 << 20); off += 4 * 64 * SIZE) {
    // the size argument must be a literal
    if constexpr (SIZE == 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + wave * 64 * 4), 4, lane * 4, off, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + wave * 64 * 16), 16, lane * 16, off, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = reinterpret_cast<float*>(lds)[0];
}
template <typename T>
__global__ __launch_bounds__(256) void calib_load(const T* x, float* out, long long n) {
  float acc = 0.0f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const T v = x[i];
    acc += reinterpret_cast<const float*>(&v)[0];
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  char* x;
  float* out;
  hipMalloc(&x, kBytes);
  hipMalloc(&out, 4096 * sizeof(float));
  hipMemset(x, 0, kBytes);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_lds_dma<4>, dim3(1024), dim3(256), 0, 0, x, out);
    hipLaunchKernelGGL(calib_lds_dma<16>, dim3(1024), dim3(256), 0, 0, x, out);
    hipLaunchKernelGGL(calib_load<float4>, dim3(2048), dim3(256), 0, 0, (const float4*)x, out, kBytes / 16);
    hipLaunchKernelGGL(calib_load<float>, dim3(2048), dim3(256), 0, 0, (const float*)x, out, kBytes / 4);
  }
  hipDeviceSynchronize();
  printf("fetch_calib: 4 kernels x 3, %lld bytes each\n", kBytes);
  return 0;
}
