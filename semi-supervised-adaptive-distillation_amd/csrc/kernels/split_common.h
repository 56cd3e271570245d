// split_common.h -- what the split-operand engines (conv3x3_split.hip, gemm_split.hip) share: the per-tensor
// power-of-two scale, the hi / lo split, the |max| pass, the NCHW fp32 -> channel-blocked hi / lo planes pass, and the
// LDS-DMA statement with the wait states an inline-asm VMEM instruction needs behind a VALU-written scalar operand.
// Internal to csrc/kernels (every definition has internal linkage).
#ifndef SSAD_SPLIT_COMMON_H_
#define SSAD_SPLIT_COMMON_H_

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_internal.h"
#include "ssad_kernels.h"

namespace ssad_split {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
using ssad_dev::uniform_rsrc;

constexpr int kThreads = 256;
constexpr unsigned kOob = 0x80000000u;
constexpr int kMaxLv = SSAD_MAX_CONV_PROBLEMS;

// The tensor's scale is 2^(15 - e) with |max| < 2^e: the largest element lands in [2^14, 2^15).  Zero, Inf and NaN
// maxima give e = 15 (scale 1); e is clamped so that both 2^(15 - e) and 2^(e - 15) are normal fp32 numbers.
__device__ __forceinline__ int split_exponent(unsigned amax_bits) {
  if (amax_bits == 0u || amax_bits >= 0x7f800000u) return 15;
  int e = (int)(amax_bits >> 23) - 126;
  return e < -110 ? -110 : e;
}
__device__ __forceinline__ float pow2f(int k) { return __uint_as_float((unsigned)(127 + k) << 23); }

__device__ __forceinline__ void split8(const float (&v)[8], float s, half8& hi, half8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float xs = v[e] * s;                 // exact (power of two) unless it lands in fp32's denormals
    const _Float16 h = (_Float16)xs;           // round to nearest even
    hi[e] = h;
    lo[e] = (_Float16)(xs - (float)h);         // the difference is exact in fp32
  }
}

// ---- 1. |max| per level ---------------------------------------------------------------------------------
struct AmaxTable {
  const float* x[kMaxLv];
  long long n[kMaxLv];
  int block_start[kMaxLv + 1];
  int count;
  unsigned* amax;
};
__global__ __launch_bounds__(kThreads) void split_absmax_kernel(const AmaxTable t) {
  int k = 0;
  for (int j = 1; j < t.count; ++j) k += (int)blockIdx.x >= t.block_start[j];
  const int nb = t.block_start[k + 1] - t.block_start[k], b = (int)blockIdx.x - t.block_start[k];
  const float* x = t.x[k];
  const long long n = t.n[k], n4 = n >> 2;
  unsigned m = 0;
  const uint4* x4 = reinterpret_cast<const uint4*>(x);
#pragma unroll 4
  for (long long i = (long long)b * kThreads + threadIdx.x; i < n4; i += (long long)nb * kThreads) {
    const uint4 v = x4[i];
    const unsigned a0 = v.x & 0x7fffffffu, a1 = v.y & 0x7fffffffu, a2 = v.z & 0x7fffffffu, a3 = v.w & 0x7fffffffu;
    const unsigned p = a0 > a1 ? a0 : a1, q = a2 > a3 ? a2 : a3;
    const unsigned r = p > q ? p : q;
    m = m > r ? m : r;
  }
  if (b == 0)
    for (long long i = n4 * 4 + threadIdx.x; i < n; i += kThreads) {
      const unsigned a = __float_as_uint(x[i]) & 0x7fffffffu;
      m = m > a ? m : a;
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned other = (unsigned)__shfl_xor((int)m, o, 64);
    m = m > other ? m : other;
  }
  // (|x| as an unsigned word orders like the float; a NaN's word is above Inf's and survives the max)
  // one atomic per workgroup: with one per wave the ~80 K same-address atomics of a tower launch took 1 ms
  __shared__ unsigned red[kThreads / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / 64; ++w) m = m > red[w] ? m : red[w];
    if (m) atomicMax(t.amax + k, m);
  }
}

// ---- 2. NCHW fp32 -> blocked hi / lo planes ---------------------------------------------------------------
struct ActTable {
  const float* x[kMaxLv];
  uint4* planes[kMaxLv];          // hi plane [N][CB][plane]; the lo plane follows it
  int N[kMaxLv];
  long long plane[kMaxLv];
  int block_start[kMaxLv + 1];
  int count, C;
  const unsigned* amax;
};
__global__ __launch_bounds__(kThreads) void split_pack_act_kernel(const ActTable t) {
  int k = 0;
  for (int j = 1; j < t.count; ++j) k += (int)blockIdx.x >= t.block_start[j];
  const int C = t.C, CB = (C + 7) >> 3;
  const long long plane = t.plane[k], total = (long long)t.N[k] * CB * plane;
  const long long i = (long long)((int)blockIdx.x - t.block_start[k]) * kThreads + threadIdx.x;
  if (i >= total) return;
  const float s = pow2f(15 - split_exponent(t.amax[k]));
  const long long px = i % plane, ncb = i / plane;
  const int cb = (int)(ncb % CB);
  const float* src = t.x[k] + ((ncb / CB) * C + cb * 8) * plane + px;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = cb * 8 + e < C ? src[e * plane] : 0.0f;
  half8 hi, lo;
  split8(v, s, hi, lo);
  t.planes[k][i] = __builtin_bit_cast(uint4, hi);
  t.planes[k][total + i] = __builtin_bit_cast(uint4, lo);
}


// ssad_dev::lds_dma<16> with the wait states a VALU-written scalar operand needs in front of a VMEM instruction the
// compiler does not see (see ring_load below): s_mov m0 + s_nop 3 = 5 wait states before the load.
__device__ __forceinline__ void dma16(const ssad_dev::rsrc_words& rsrc, unsigned lds_byte_addr, unsigned voffset, int soffset) {
  unsigned saved_m0;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 3\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(saved_m0) : "s"(lds_byte_addr), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
}


}  // namespace
}  // namespace ssad_split

#endif  // SSAD_SPLIT_COMMON_H_
