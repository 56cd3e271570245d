// ref_driver_hip.cc -- the REFERENCE's own loss kernels, compiled by hipcc for gfx950 and launched on the GPU.
//
// TEST INFRASTRUCTURE ONLY (a checker for tests/ -m gpu; never linked into or loaded by the product).  Built only
// in the container that has /root/reference (oracle/Makefile target `ref`); the .so travels to the GPU box,
// the reference sources do not.
//
// What is compiled: the __global__ kernels of
//     caffe2/modules/detectron/sigmoid_adaptive_distillation_loss_op.cu:28-105
//     caffe2/modules/detectron/sigmoid_focal_loss_op.cu:26-109
//     caffe2/modules/detectron/select_smooth_l1_loss_op.cu:23-86
// as their text stands in the reference (extracted at build time into a temporary file, included below, deleted
// afterwards -- no reference text is stored in this repository), with the reference's grid-stride loop macro
// (caffe2/core/common_gpu.h:246-248) and its launch geometry (common_gpu.h:274-288: 512 threads, at most 4096
// blocks) taken from the reference the same way.  Unlike oracle/ref_driver.cc (the host compile), NOTHING is
// supplied in CUDA's place here: `__global__`, blockIdx / blockDim / threadIdx / gridDim, the mixed-type `max`,
// `abs`, expf / logf / powf are the HIP toolchain's own (hipcc understands the CUDA kernel language natively; the
// device math library is ROCm's ocml where the reference ran on libdevice).  What is still NOT compiled is the
// operators' RunOnDevice (it needs the Caffe2 core and the CUDA runtime headers, which this image does not have):
// its epilogue -- math::Sum, math::Scale (.cu:135-138, 167-168) -- is restated in oracle/ssad_oracle.c.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <cstdint>

#include REF_LOOP_INC

namespace ref_geom {
#include REF_GEOM_INC
}

namespace ref_kernels {
#include REF_KERNELS_INC
}
namespace ref_focal {
#include REF_FOCAL_INC
}
namespace ref_smoothl1 {
#include REF_SMOOTHL1_INC
}

#define REF_API extern "C" __attribute__((visibility("default")))

// every pointer is a DEVICE pointer; the launch is <<<CAFFE_GET_BLOCKS(n), CAFFE_CUDA_NUM_THREADS, 0, stream>>>
// exactly as the reference's RunOnDevice issues it; returns hipGetLastError()

REF_API int ref_hip_distill_loss(int N, int D, int H, int W, int ignored_label, const float* logits,
                                 const float* targets, const int* gt, const float* weight_pos, float gamma,
                                 float alpha, float beta, int num_classes, float* losses, void* stream) {
  const int n = N * D * H * W;
  hipLaunchKernelGGL(ref_kernels::SigmoidAdaptiveDistillLossKernel, dim3(ref_geom::CAFFE_GET_BLOCKS(n)),
                     dim3(ref_geom::CAFFE_CUDA_NUM_THREADS), 0, (hipStream_t)stream, N, D, H, W, ignored_label,
                     logits, targets, gt, weight_pos, gamma, alpha, beta, num_classes, losses);
  return (int)hipGetLastError();
}

REF_API int ref_hip_distill_grad(int N, int D, int H, int W, int ignored_label, const float* logits,
                                 const float* targets, const int* gt, float* dX, const float* weight_pos,
                                 float gamma, float alpha, float beta, int num_classes, const float* avg_loss,
                                 void* stream) {
  const int n = N * D * H * W;
  hipLaunchKernelGGL(ref_kernels::SigmoidAdaptiveDistillLossGradientKernel, dim3(ref_geom::CAFFE_GET_BLOCKS(n)),
                     dim3(ref_geom::CAFFE_CUDA_NUM_THREADS), 0, (hipStream_t)stream, N, D, H, W, ignored_label,
                     logits, targets, gt, dX, weight_pos, gamma, alpha, beta, num_classes, avg_loss);
  return (int)hipGetLastError();
}

REF_API int ref_hip_focal_loss(int N, int D, int H, int W, const float* logits, const int* targets,
                               const float* weight_pos, float gamma, float alpha, int num_classes, float* losses,
                               void* stream) {
  const int n = N * D * H * W;
  hipLaunchKernelGGL(ref_focal::SigmoidFocalLossKernel, dim3(ref_geom::CAFFE_GET_BLOCKS(n)),
                     dim3(ref_geom::CAFFE_CUDA_NUM_THREADS), 0, (hipStream_t)stream, N, D, H, W, logits, targets,
                     weight_pos, gamma, alpha, num_classes, losses);
  return (int)hipGetLastError();
}

REF_API int ref_hip_focal_grad(int N, int D, int H, int W, const float* logits, const int* targets, float* dX,
                               const float* weight_pos, float gamma, float alpha, int num_classes,
                               const float* avg_loss, void* stream) {
  const int n = N * D * H * W;
  hipLaunchKernelGGL(ref_focal::SigmoidFocalLossGradientKernel, dim3(ref_geom::CAFFE_GET_BLOCKS(n)),
                     dim3(ref_geom::CAFFE_CUDA_NUM_THREADS), 0, (hipStream_t)stream, N, D, H, W, logits, targets,
                     dX, weight_pos, gamma, alpha, num_classes, avg_loss);
  return (int)hipGetLastError();
}

// (the reference sizes both smooth-L1 launches by Y_hat's element count, select_smooth_l1_loss_op.cu:125,173)
REF_API int ref_hip_smoothl1(int yhat_size, int D, int H, int W, int M, const float* Y_hat, const float* Y,
                             const float* L, float* out, const float* S, float beta, void* stream) {
  hipLaunchKernelGGL(ref_smoothl1::SelectSmoothL1Kernel, dim3(ref_geom::CAFFE_GET_BLOCKS(yhat_size)),
                     dim3(ref_geom::CAFFE_CUDA_NUM_THREADS), 0, (hipStream_t)stream, D, H, W, M, Y_hat, Y, L, out,
                     S, beta);
  return (int)hipGetLastError();
}

REF_API int ref_hip_smoothl1_grad(int yhat_size, int D, int H, int W, int M, const float* Y_hat, const float* Y,
                                  const float* L, float* out, const float* d_loss, float norm, const float* S,
                                  float beta, void* stream) {
  hipLaunchKernelGGL(ref_smoothl1::SelectSmoothL1GradientKernel, dim3(ref_geom::CAFFE_GET_BLOCKS(yhat_size)),
                     dim3(ref_geom::CAFFE_CUDA_NUM_THREADS), 0, (hipStream_t)stream, D, H, W, M, Y_hat, Y, L, out,
                     d_loss, norm, S, beta);
  return (int)hipGetLastError();
}
