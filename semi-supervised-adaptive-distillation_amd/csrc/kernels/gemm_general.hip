// gemm_general.hip -- plain fp32 GEMM, C[M x N] = alpha op(A) op(B) + beta C, row-major, any
// transposition / leading dimensions / sizes, optionally strided-batched: math::Gemm and
// math::GemmStridedBatched of caffe2/utils/math_gpu.cu:33-80 for the default convolution engine's
// im2col route (conv_op_impl.h:126-173 forward, :451-560 gradients: k x k / strided / grouped
// geometries outside the implicit-GEMM and 3x3 kernels).  Round 3: this replaces the rocBLAS
// calls the product library used to make (csrc/c2/blas.cc dlopen'ed librocblas.so).
//
// v_mfma_f32_32x32x2_f32 (exact fp32); workgroup = 4 waves = a 64 x 64 tile (wave = 32 x 32), K
// in chunks of 16 through LDS ([k][64 + 1] images: the k-major operand rows are read
// conflict-free).  Operands are gathered element-wise with bounds checks -- no alignment or
// divisibility assumption -- in the order that is contiguous in memory for the operand's
// transposition.  The layers that matter have kernels of their own; this one has to be right.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssad_kernels.h"

namespace {

constexpr int kThreads = 256;
constexpr int TM = 64, TN = 64, TK = 16, PAD = 1;
typedef float float16v __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K, lda, ldb, ldc;
  long long sa, sb, sc;
  float alpha, beta;
  int ta, tb;
};

__global__ __launch_bounds__(kThreads) void gemm_general_kernel(const GemmArgs g) {
  __shared__ float As[TK][TM + PAD];
  __shared__ float Bs[TK][TN + PAD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const float* A = g.A + (long long)blockIdx.z * g.sa;
  const float* B = g.B + (long long)blockIdx.z * g.sb;
  float* C = g.C + (long long)blockIdx.z * g.sc;
  float16v acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const int i = lane & 31, kk = lane >> 5;
  for (int k0 = 0; k0 < g.K; k0 += TK) {
#pragma unroll
    for (int e = 0; e < (TM * TK) / kThreads; ++e) {
      const int idx = e * kThreads + tid;
      // op(A)[m][k]: stored [m][k] (k contiguous) or, transposed, [k][m] (m contiguous)
      const int m = g.ta ? idx % TM : idx / TK, k = g.ta ? idx / TM : idx % TK;
      const int gm = m0 + m, gk = k0 + k;
      float v = 0.0f;
      if (gm < g.M && gk < g.K) v = g.ta ? A[(long long)gk * g.lda + gm] : A[(long long)gm * g.lda + gk];
      As[k][m] = v;
    }
#pragma unroll
    for (int e = 0; e < (TN * TK) / kThreads; ++e) {
      const int idx = e * kThreads + tid;
      // op(B)[k][n]: stored [k][n] (n contiguous) or, transposed, [n][k] (k contiguous)
      const int n = g.tb ? idx / TK : idx % TN, k = g.tb ? idx % TK : idx / TN;
      const int gn = n0 + n, gk = k0 + k;
      float v = 0.0f;
      if (gn < g.N && gk < g.K) v = g.tb ? B[(long long)gn * g.ldb + gk] : B[(long long)gk * g.ldb + gn];
      Bs[k][n] = v;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < TK; ks += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[ks + kk][wm * 32 + i], Bs[ks + kk][wn * 32 + i], acc, 0, 0, 0);
    __syncthreads();
  }
  const int gn = n0 + wn * 32 + i;
  if (gn >= g.N) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
    if (gm < g.M) {
      float* c = C + (long long)gm * g.ldc + gn;
      const float v = g.alpha * acc[r];
      *c = g.beta == 0.0f ? v : v + g.beta * *c;        // beta = 0 must not read C (may hold NaN)
    }
  }
}

}  // namespace

extern "C" int ssad_gemm_f32(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A, int lda,
                             long long stride_a, const float* B, int ldb, long long stride_b, float beta, float* C,
                             int ldc, long long stride_c, int batch, ssad_stream_t stream) {
  if (M < 0 || N < 0 || K < 0 || batch < 0 || lda < 1 || ldb < 1 || ldc < 1) return SSAD_E_BADARG;
  if (M == 0 || N == 0 || batch == 0) return 0;
  if (!A || !B || !C) return SSAD_E_BADARG;
  const long long gy = (M + TM - 1) / TM, gx = (N + TN - 1) / TN;
  if (gy > 65535 || batch > 65535) return SSAD_E_BADARG;
  GemmArgs g{A, B, C, M, N, K, lda, ldb, ldc, stride_a, stride_b, stride_c, alpha, beta, trans_a ? 1 : 0,
             trans_b ? 1 : 0};
  hipLaunchKernelGGL(gemm_general_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)batch), dim3(kThreads), 0,
                     (hipStream_t)stream, g);
  return (int)hipGetLastError();
}
