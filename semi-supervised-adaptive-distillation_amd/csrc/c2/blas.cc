#include "c2/blas.h"

#include "ssad_kernels.h"

// math::Gemm / math::GemmStridedBatched (caffe2/utils/math_gpu.cu:33-80) for the default
// convolution engine, on this repo's own fp32-MFMA kernel (kernels/gemm_general.hip).  No vendor
// BLAS is linked or opened by the product library.
namespace caffe2 {

void GemmRowMajor(hipStream_t stream, bool trans_a, bool trans_b, int M, int N, int K, float alpha,
                  const float* A, int lda, const float* B, int ldb, float beta, float* C, int ldc) {
  const int rc = ssad_gemm_f32(trans_a, trans_b, M, N, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 1, stream);
  CAFFE_ENFORCE_EQ(rc, 0, "ssad_gemm_f32 failed");
}

void GemmRowMajorStridedBatched(hipStream_t stream, bool trans_a, bool trans_b, int M, int N, int K,
                                float alpha, const float* A, int lda, long long stride_a,
                                const float* B, int ldb, long long stride_b, float beta, float* C,
                                int ldc, long long stride_c, int batch) {
  const int rc = ssad_gemm_f32(trans_a, trans_b, M, N, K, alpha, A, lda, stride_a, B, ldb, stride_b, beta, C, ldc,
                               stride_c, batch, stream);
  CAFFE_ENFORCE_EQ(rc, 0, "ssad_gemm_f32 (strided batched) failed");
}

}  // namespace caffe2
