// blas.h -- plain fp32 GEMM for the default convolution engine's im2col route, on this repo's
// own matrix-core kernel (kernels/gemm_general.hip); no vendor BLAS.
#ifndef C2HIP_BLAS_H_
#define C2HIP_BLAS_H_

#include <hip/hip_runtime_api.h>

#include "c2/common.h"

namespace caffe2 {

// Row-major C[M x N] = alpha * op(A) * op(B) + beta * C with leading dimensions in
// elements (math::Gemm of caffe2/utils/math_gpu.cu:33-80).
C2HIP_API void GemmRowMajor(hipStream_t stream, bool trans_a, bool trans_b, int M, int N, int K,
                            float alpha, const float* A, int lda, const float* B, int ldb,
                            float beta, float* C, int ldc);

// The same over `batch` independent problems whose operands are `stride_*` elements apart
// (the groups of a grouped convolution: math::GemmStridedBatched).
C2HIP_API void GemmRowMajorStridedBatched(hipStream_t stream, bool trans_a, bool trans_b, int M, int N,
                                          int K, float alpha, const float* A, int lda,
                                          long long stride_a, const float* B, int ldb,
                                          long long stride_b, float beta, float* C, int ldc,
                                          long long stride_c, int batch);

}  // namespace caffe2
#endif  // C2HIP_BLAS_H_
