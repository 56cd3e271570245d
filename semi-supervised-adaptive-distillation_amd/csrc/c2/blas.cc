#include "c2/blas.h"

#include <dlfcn.h>

#include <mutex>

namespace caffe2 {
namespace {

// the few rocBLAS entry points used, resolved at run time (rocblas/rocblas.h)
typedef void* rocblas_handle_t;
typedef int (*create_handle_fn)(rocblas_handle_t*);
typedef int (*set_stream_fn)(rocblas_handle_t, hipStream_t);
typedef int (*sgemm_fn)(rocblas_handle_t, int, int, int, int, int, const float*, const float*, int,
                        const float*, int, const float*, float*, int);
typedef int (*sgemm_sb_fn)(rocblas_handle_t, int, int, int, int, int, const float*, const float*, int,
                           long long, const float*, int, long long, const float*, float*, int,
                           long long, int);
constexpr int kOpNone = 111, kOpTranspose = 112;     // rocblas_operation_none / _transpose

struct RocBlas {
  create_handle_fn create = nullptr;
  set_stream_fn set_stream = nullptr;
  sgemm_fn sgemm = nullptr;
  sgemm_sb_fn sgemm_sb = nullptr;
};

const RocBlas& Lib() {
  static RocBlas lib;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = nullptr;
    for (const char* name : {"librocblas.so", "librocblas.so.5", "librocblas.so.4",
                             "/opt/rocm/lib/librocblas.so"}) {
      h = dlopen(name, RTLD_LAZY | RTLD_LOCAL);
      if (h) break;
    }
    CAFFE_ENFORCE(h, "the default convolution engine needs rocBLAS, and librocblas.so could "
                     "not be opened: ", dlerror());
    lib.create = (create_handle_fn)dlsym(h, "rocblas_create_handle");
    lib.set_stream = (set_stream_fn)dlsym(h, "rocblas_set_stream");
    lib.sgemm = (sgemm_fn)dlsym(h, "rocblas_sgemm");
    lib.sgemm_sb = (sgemm_sb_fn)dlsym(h, "rocblas_sgemm_strided_batched");
    CAFFE_ENFORCE(lib.create && lib.set_stream && lib.sgemm && lib.sgemm_sb, "rocBLAS symbols missing");
  });
  return lib;
}

rocblas_handle_t Handle() {
  static thread_local rocblas_handle_t handle = nullptr;
  if (!handle) CAFFE_ENFORCE_EQ(Lib().create(&handle), 0, "rocblas_create_handle failed");
  return handle;
}

}  // namespace

void GemmRowMajor(hipStream_t stream, bool trans_a, bool trans_b, int M, int N, int K, float alpha,
                  const float* A, int lda, const float* B, int ldb, float beta, float* C, int ldc) {
  if (M == 0 || N == 0) return;
  rocblas_handle_t h = Handle();
  CAFFE_ENFORCE_EQ(Lib().set_stream(h, stream), 0, "rocblas_set_stream failed");
  // row-major C = op(A) op(B)  <=>  column-major C^T = op(B)^T op(A)^T
  const int rc = Lib().sgemm(h, trans_b ? kOpTranspose : kOpNone, trans_a ? kOpTranspose : kOpNone, N,
                             M, K, &alpha, B, ldb, A, lda, &beta, C, ldc);
  CAFFE_ENFORCE_EQ(rc, 0, "rocblas_sgemm failed");
}

void GemmRowMajorStridedBatched(hipStream_t stream, bool trans_a, bool trans_b, int M, int N, int K,
                                float alpha, const float* A, int lda, long long stride_a,
                                const float* B, int ldb, long long stride_b, float beta, float* C,
                                int ldc, long long stride_c, int batch) {
  if (M == 0 || N == 0 || batch == 0) return;
  rocblas_handle_t h = Handle();
  CAFFE_ENFORCE_EQ(Lib().set_stream(h, stream), 0, "rocblas_set_stream failed");
  const int rc = Lib().sgemm_sb(h, trans_b ? kOpTranspose : kOpNone, trans_a ? kOpTranspose : kOpNone,
                                N, M, K, &alpha, B, ldb, stride_b, A, lda, stride_a, &beta, C, ldc,
                                stride_c, batch);
  CAFFE_ENFORCE_EQ(rc, 0, "rocblas_sgemm_strided_batched failed");
}

}  // namespace caffe2
