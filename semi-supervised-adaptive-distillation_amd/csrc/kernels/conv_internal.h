// Internal (not exported) entry points shared between the convolution sources.
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

#endif  // SSAD_CONV_INTERNAL_H_
