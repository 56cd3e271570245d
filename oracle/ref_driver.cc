// ref_driver.cc -- host driver around the REFERENCE's own kernel bodies.
//
// TEST INFRASTRUCTURE ONLY; built only in the container that has
// /root/reference (oracle/Makefile target `ref`).  The two __global__ kernel
// bodies of caffe2/modules/detectron/sigmoid_adaptive_distillation_loss_op.cu
// (lines 28-105) are extracted at build time into a temporary file that is
// passed as -DREF_KERNELS_INC=... and deleted afterwards; no reference text
// is stored in this repository.  The grid-stride loop macro is the reference's
// own text too (caffe2/core/common_gpu.h:246-248, extracted the same way as
// -DREF_LOOP_INC=...); what is supplied here is what CUDA itself would: the
// `__global__` keyword, the launch-geometry variables of a <<<1, 1>>> launch
// (blockIdx / blockDim / threadIdx / gridDim) and the mixed-type `max` / float
// `abs` overloads of CUDA's math headers.  The arithmetic that runs is the
// reference's, compiled by g++ for the host.  By the build rules this remains a
// host compile behind stand-ins for CUDA, not "the reference compiled here":
// the loss oracle's parity is reported as UNPINNED.
//
// What this is NOT: a build of the reference operator.  RunOnDevice's
// epilogue (math::Sum, math::Scale; .cu:135-138,167-168) needs the Caffe2
// core + CUDA and is restated in oracle/ssad_oracle.c instead.
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <cstdint>

#define __global__
// a <<<1, 1>>> launch: one thread walks the whole index range of the reference's loop macro
struct ref_dim3 { unsigned x, y, z; };
static const ref_dim3 blockIdx{0, 0, 0}, blockDim{1, 1, 1}, threadIdx{0, 0, 0}, gridDim{1, 1, 1};
#include REF_LOOP_INC

static inline float max(float a, float b) { return a > b ? a : b; }
static inline double max(float a, double b) { return (double)a > b ? (double)a : b; }
static inline double max(double a, double b) { return a > b ? a : b; }
static inline float abs(float a) { return a < 0 ? -a : a; }   // CUDA's float overload

namespace ref_kernels {
#include REF_KERNELS_INC
}  // namespace ref_kernels
// caffe2/modules/detectron/sigmoid_focal_loss_op.cu:26-109 and
// select_smooth_l1_loss_op.cu:23-86 (row f2 of SURVEY.md 8f), same treatment
namespace ref_focal {
#include REF_FOCAL_INC
}
namespace ref_smoothl1 {
#include REF_SMOOTHL1_INC
}

extern "C" {

__attribute__((visibility("default"))) void ref_distill_loss_kernel(
    int N, int D, int H, int W, int ignored_label, const float* logits,
    const float* targets, const int* gt, const float* weight_pos, float gamma,
    float alpha, float beta, int num_classes, float* losses) {
  ref_kernels::SigmoidAdaptiveDistillLossKernel(
      N, D, H, W, ignored_label, logits, targets, gt, weight_pos, gamma, alpha,
      beta, num_classes, losses);
}

__attribute__((visibility("default"))) void ref_distill_grad_kernel(
    int N, int D, int H, int W, int ignored_label, const float* logits,
    const float* targets, const int* gt, float* dX, const float* weight_pos,
    float gamma, float alpha, float beta, int num_classes,
    const float* avg_loss) {
  ref_kernels::SigmoidAdaptiveDistillLossGradientKernel(
      N, D, H, W, ignored_label, logits, targets, gt, dX, weight_pos, gamma,
      alpha, beta, num_classes, avg_loss);
}

__attribute__((visibility("default"))) void ref_focal_loss_kernel(
    int N, int D, int H, int W, const float* logits, const int* targets, const float* weight_pos,
    float gamma, float alpha, int num_classes, float* losses) {
  ref_focal::SigmoidFocalLossKernel(N, D, H, W, logits, targets, weight_pos, gamma, alpha,
                                    num_classes, losses);
}

__attribute__((visibility("default"))) void ref_focal_grad_kernel(
    int N, int D, int H, int W, const float* logits, const int* targets, float* dX,
    const float* weight_pos, float gamma, float alpha, int num_classes, const float* avg_loss) {
  ref_focal::SigmoidFocalLossGradientKernel(N, D, H, W, logits, targets, dX, weight_pos, gamma,
                                            alpha, num_classes, avg_loss);
}

__attribute__((visibility("default"))) void ref_smoothl1_kernel(
    int D, int H, int W, int M, const float* Y_hat, const float* Y, const float* L, float* out,
    const float* S, float beta) {
  ref_smoothl1::SelectSmoothL1Kernel(D, H, W, M, Y_hat, Y, L, out, S, beta);
}

__attribute__((visibility("default"))) void ref_smoothl1_grad_kernel(
    int D, int H, int W, int M, const float* Y_hat, const float* Y, const float* L, float* out,
    const float* d_loss, float norm, const float* S, float beta) {
  ref_smoothl1::SelectSmoothL1GradientKernel(D, H, W, M, Y_hat, Y, L, out, d_loss, norm, S, beta);
}

}  // extern "C"
