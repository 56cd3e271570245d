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

}  // namespace ssad_dev

#endif  // SSAD_CONV_INTERNAL_H_
