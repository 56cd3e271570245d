// Internal (not exported) declarations shared between the convolution sources.
#ifndef SSAD_CONV_INTERNAL_H_
#define SSAD_CONV_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stddef.h>

#include "ssad_kernels.h"

// Winograd F(3x3,2x2) filter-gradient engine (conv3x3_wgrad_winograd.hip).
// Used by ssad_conv3x3_wgrad for Cout >= 32, Cin >= 64; SSAD_WGRAD_ENGINE=direct /
// winograd overrides the choice.
bool ssad_wino_wgrad_eligible(int Cout, int Cin);
size_t ssad_wino_wgrad_workspace_bytes(const ssad_conv_level* lv, int n_levels, int Cout, int Cin);
int ssad_wino_wgrad_launch(const ssad_conv_level* lv, int n_levels, float* dW, int Cout, int Cin,
                           int accumulate, void* workspace, size_t workspace_bytes,
                           hipStream_t stream);

// Split-operand filter-gradient engine (conv3x3_wgrad_split.hip): |max| pass + main kernel + slab reduction on `stream`.
size_t ssad_split_wgrad_workspace_bytes(const ssad_conv_level* lv, int n_levels, int Cout, int Cin);
int ssad_split_wgrad_launch(const ssad_conv_level* lv, int n_levels, float* dW, int Cout, int Cin, int accumulate,
                            void* workspace, size_t workspace_bytes, const unsigned* x_amax, const unsigned* dy_amax,
                            hipStream_t stream);

// The ResNet stem (7x7 / stride 2 / pad 3, 3 -> 64 channels) from an LDS-staged raw patch (stem.hip); taken by
// ssad_conv_implicit_gemm for exactly that geometry when no epilogue term is asked for.
int ssad_stem7x7s2_launch(const float* x, const float* wt, int lda, int N, int H, int W, float* y, hipStream_t stream);

// Compute units of the CURRENT device (cached per device: a process may drive several GPUs, and
// workspace sizing and launch must agree on the same device's count).
inline int ssad_cu_count() {
  static int cache[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cache[dev] > 0) return cache[dev];
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
  cache[dev] = n;
  return n;
}

namespace ssad_dev {

// Buffer descriptor from values the compiler must treat as wave-uniform: every
// input goes through readfirstlane, otherwise hipcc wraps each buffer load in a
// waterfall loop that serialises the loads.  Loads past `bytes` return 0, stores
// are dropped (the kernels send out-of-image lanes to offset 0x80000000).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const unsigned n = __builtin_amdgcn_readfirstlane(bytes);
  void* q = (void*)(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, n, 0x00020000);
}

// ---- LDS-DMA the compiler's waitcnt pass cannot see -------------------------------------------
// Behind a `__builtin_amdgcn_raw_ptr_buffer_load_lds` in flight hipcc makes the NEXT LDS read of
// the kernel wait for that DMA (it cannot prove that the read and the DMA touch different
// stages): a multi-stage prefetch then stalls at every stage for the full latency of the fetch
// it was supposed to hide (seen in the ISA as `s_waitcnt vmcnt(..)` in front of the first ds_read
// after the issue).  Issued from inline assembly the DMA is invisible to that pass; the kernels
// wait for it themselves (counted `s_waitcnt vmcnt(N)` + barrier) where a stage changes hands.
// The compiler's own vmcnt bookkeeping for register loads stays safe: vector memory retires in
// order, so an untracked operation can only make its waits longer, never shorter.
typedef int rsrc_words __attribute__((ext_vector_type(4)));

__device__ __forceinline__ rsrc_words uniform_rsrc_words(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const unsigned n = __builtin_amdgcn_readfirstlane(bytes);
  return rsrc_words{(int)lo, (int)(hi & 0xffffu), (int)n, 0x00020000};
}

// lane l: LDS[lds_byte_addr + BYTES * l ...] = buffer[voffset + soffset ...]  (zeros past the descriptor;
// send a lane to offset 0x80000000 for zero fill).  lds_byte_addr and soffset must be wave-uniform.
template <int BYTES>
__device__ __forceinline__ void lds_dma(const rsrc_words& rsrc, unsigned lds_byte_addr, unsigned voffset,
                                        int soffset) {
  static_assert(BYTES == 4 || BYTES == 16, "LDS-DMA moves 4 or 16 bytes per lane");
  // M0 carries the LDS address of an LDS-DMA; it is a reserved register the compiler may be using
  // itself, so the statement saves and restores it instead of declaring a clobber
  unsigned saved_m0;
  if (BYTES == 4)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, %4 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(saved_m0) : "s"(lds_byte_addr), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(saved_m0) : "s"(lds_byte_addr), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
}

}  // namespace ssad_dev

#endif  // SSAD_CONV_INTERNAL_H_
