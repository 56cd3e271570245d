// conv1x1_fused.hip -- pointwise convolution with the bottleneck tail in its epilogue:
//     Y[n][m][p] = act( sum_c W[m][c] X[n][c][p] + bias[m] (+ R[n][m][p]) ),  NCHW fp32,
// i.e. Conv(kernel=1) -> AffineChannel (folded into W / bias) -> Sum with the shortcut -> Relu
// (detectron/lib/modeling/ResNet.py:176-197, :223-283) in ONE pass over the tensors.
//
// Why a kernel of its own: at the early stages (res2: 64 / 256 channels on 160 x 224, res3:
// 128 / 512 on 80 x 112) these layers are HBM bound -- 2 K flop per output element against 12 B
// of traffic -- and the library route pays the output twice: the GEMM writes Z, the fused tail
// pass reads Z and the shortcut and writes Y.  Here Z never exists: X (small) is staged through
// LDS, W stays in LDS for the whole workgroup, and the accumulators meet bias, shortcut and
// ReLU in registers.  The arithmetic is exact fp32 MFMA (v_mfma_f32_32x32x2f32).
//
// Two kernels: persistent, prefetching workgroups with W^T resident in LDS for 64 / 128 input
// channels (the HBM-bound cases this exists for), and a generic one (any multiple of 32 input
// channels): 4 waves, tile 128 output channels x 128 pixels, K in double-buffered chunks of 32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "conv_internal.h"
#include "ssad_kernels.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int TM = 128, TP = 128;

struct PwArgs {
  const float* x2;     // optional second input [N][C - C1][P]: input channels C1.. come from it
  int C1;              // channels taken from x (= C when x2 is null)
  const float* x;      // [N][C1][P]
  const float* w;      // [M][C]
  const float* bias;   // [M] or null
  const float* res;    // [N][M][P] or null
  float* y;            // [N][M][P]
  int N, C, P, M, relu;
  int ptiles;          // ceil(P / 128)
};

// ---- any other channel count (multiples of 32): K in double-buffered chunks of 32 -----------
// The next chunk's W and X blocks travel global -> registers while the current one is
// multiplied, then registers -> the other LDS buffer; one barrier per chunk.
constexpr int KP = 32;
__global__ __launch_bounds__(kThreads, 2) void conv1x1_fused_pipelined_kernel(const PwArgs a) {
  __shared__ float wl[2][KP][TM];
  __shared__ float xl[2][KP][TP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wp = wave >> 1;
  const int j = lane & 31, h = lane >> 5;
  const int pt = blockIdx.x % a.ptiles, n = blockIdx.x / a.ptiles;
  const int m0 = blockIdx.y * TM, p0 = pt * TP;
  const float* xn = a.x + (long long)n * a.C * a.P;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int p = p0 + wp * 64 + t * 32 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        acc[i][t][r] = (a.res && p < a.P) ? a.res[((long long)n * a.M + m) * a.P + p] : 0.0f;
      }
    }

  float4 wr[4], xr[4];
  const int w_m = tid & 127, w_kh = tid >> 7;              // W: row m, 16 consecutive k
  auto fetch = [&](int k0) {
    const float4* src = reinterpret_cast<const float4*>(a.w + (long long)(m0 + w_m) * a.C + k0 + w_kh * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) wr[q] = src[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + kThreads * q;
      const int k = e >> 5, p = p0 + (e & 31) * 4;
      xr[q] = p < a.P ? *reinterpret_cast<const float4*>(xn + (long long)(k0 + k) * a.P + p)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = w_kh * 16 + q * 4;
      wl[buf][k][w_m] = wr[q].x; wl[buf][k + 1][w_m] = wr[q].y;
      wl[buf][k + 2][w_m] = wr[q].z; wl[buf][k + 3][w_m] = wr[q].w;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + kThreads * q;
      *reinterpret_cast<float4*>(&xl[buf][e >> 5][(e & 31) * 4]) = xr[q];
    }
  };

  fetch(0);
  stash(0);
  __syncthreads();
  const int chunks = a.C / KP;
  for (int c = 0; c < chunks; ++c) {
    const int cur = c & 1;
    if (c + 1 < chunks) fetch((c + 1) * KP);
#pragma unroll
    for (int ks = 0; ks < KP / 2; ++ks) {
      const int k = 2 * ks + h;
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = wl[cur][k][wm * 64 + i * 32 + j];
#pragma unroll
      for (int t = 0; t < 2; ++t) bv[t] = xl[cur][k][wp * 64 + t * 32 + j];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[t], acc[i][t], 0, 0, 0);
    }
    if (c + 1 < chunks) stash(cur ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float bvv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      bvv[r] = a.bias ? a.bias[m] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int p = p0 + wp * 64 + t * 32 + j;
      if (p >= a.P) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[i][t][r] + bvv[r];
        if (a.relu) v = fmaxf(v, 0.0f);
        a.y[((long long)n * a.M + m) * a.P + p] = v;
      }
    }
  }
}

// ---- 64 / 128 input channels (the last layers of res2 / res3): persistent workgroups -------
// The layer is HBM bound (12 B against 2 C flop per output element), so what matters is bytes
// in flight: a workgroup keeps its W^T block [C][128] in LDS for its whole life and walks over
// (image, 64-pixel) tiles; while tile i is multiplied and stored, tile i+1's X block
// (-> registers -> the other LDS buffer) and shortcut block (-> the next accumulators) are
// already on their way.  C = 64: 4 waves, 64 KiB of LDS, 2 workgroups per CU; C = 128: 8 waves,
// 128 KiB, one workgroup per CU.
constexpr int PT = 64;        // pixels per tile
template <int C_, int NT>
__global__ __launch_bounds__(NT, NT == 256 ? 2 : 1) void conv1x1_fused_persistent_kernel(const PwArgs a, int tiles) {
  extern __shared__ float smem[];
  float (*wl)[TM] = reinterpret_cast<float (*)[TM]>(smem);                       // [C_][128]
  float (*xl)[C_][PT] = reinterpret_cast<float (*)[C_][PT]>(smem + C_ * TM);     // [2][C_][64]
  constexpr int CT = (PT / 32) / (NT / 256);      // 32-pixel column tiles per wave
  constexpr int XQ = C_ * PT / 4 / NT;            // float4 of the X tile per thread
  constexpr int WK = C_ / (NT / 128);             // consecutive k per thread in the W^T staging
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int wm = wave & 3, c0 = (wave >> 2) * CT;
  const int m0 = blockIdx.y * TM, mw = m0 + wm * 32;         // this wave's 32 output channels
  {   // W^T block, once
    const int m = tid & 127, kh = tid >> 7;
    const float4* src = reinterpret_cast<const float4*>(a.w + (long long)(m0 + m) * C_ + kh * WK);
#pragma unroll
    for (int q = 0; q < WK / 4; ++q) {
      const float4 v = src[q];
      const int k = kh * WK + q * 4;
      wl[k][m] = v.x; wl[k + 1][m] = v.y; wl[k + 2][m] = v.z; wl[k + 3][m] = v.w;
    }
  }
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = a.bias ? a.bias[mw + (r & 3) + 8 * (r >> 2) + 4 * h] : 0.0f;

  float4 xr[XQ];
  f32x16 acc[CT], nxt[CT];
  auto fetch_x = [&](int t) {
    const int n = t / a.ptiles, p0 = (t % a.ptiles) * PT;
    const float* xn = a.x + (long long)n * a.C1 * a.P;
    const float* xn2 = a.x2 ? a.x2 + (long long)n * (C_ - a.C1) * a.P : nullptr;
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const int e = tid + NT * q;
      const int k = e >> 4, p = p0 + (e & 15) * 4;
      const float* row = k < a.C1 ? xn + (long long)k * a.P : xn2 + (long long)(k - a.C1) * a.P;
      xr[q] = p < a.P ? *reinterpret_cast<const float4*>(row + p) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stash_x = [&](int buf) {
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const int e = tid + NT * q;
      *reinterpret_cast<float4*>(&xl[buf][e >> 4][(e & 15) * 4]) = xr[q];
    }
  };
  auto fetch_r = [&](int t, f32x16 (&dst)[CT]) {
    const int n = t / a.ptiles, p0 = (t % a.ptiles) * PT;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int p = p0 + (c0 + c) * 32 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mw + (r & 3) + 8 * (r >> 2) + 4 * h;
        dst[c][r] = (a.res && p < a.P) ? a.res[((long long)n * a.M + m) * a.P + p] : 0.0f;
      }
    }
  };

  int t = blockIdx.x;
  if (t < tiles) {
    fetch_x(t);
    fetch_r(t, acc);
    stash_x(0);
  }
  __syncthreads();
  for (int it = 0; t < tiles; t += gridDim.x, ++it) {
    const int cur = it & 1;
    const int tn = t + gridDim.x;
    if (tn < tiles) {
      fetch_x(tn);
      fetch_r(tn, nxt);
    }
#pragma unroll 8
    for (int ks = 0; ks < C_ / 2; ++ks) {
      const int k = 2 * ks + h;
      const float av = wl[k][wm * 32 + j];
#pragma unroll
      for (int c = 0; c < CT; ++c)
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xl[cur][k][(c0 + c) * 32 + j], acc[c], 0, 0, 0);
    }
    {
      const int n = t / a.ptiles, p0 = (t % a.ptiles) * PT;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int p = p0 + (c0 + c) * 32 + j;
        if (p >= a.P) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mw + (r & 3) + 8 * (r >> 2) + 4 * h;
          float v = acc[c][r] + bv[r];
          if (a.relu) v = fmaxf(v, 0.0f);
          a.y[((long long)n * a.M + m) * a.P + p] = v;
        }
      }
    }
    if (tn < tiles) {
      stash_x(cur ^ 1);
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[c] = nxt[c];
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" {

int ssad_conv1x1_bias_act2(const float* x, int C1, const float* x2, int C2, const float* w, const float* bias,
                           const float* residual, float* y, int N, int P, int M, int relu,
                           ssad_stream_t stream);

int ssad_conv1x1_bias_act(const float* x, const float* w, const float* bias, const float* residual,
                          float* y, int N, int C, int P, int M, int relu, ssad_stream_t stream) {
  return ssad_conv1x1_bias_act2(x, C, nullptr, 0, w, bias, residual, y, N, P, M, relu, stream);
}

int ssad_conv1x1_bias_act2(const float* x, int C1, const float* x2, int C2, const float* w, const float* bias,
                           const float* residual, float* y, int N, int P, int M, int relu,
                           ssad_stream_t stream) {
  const int C = C1 + C2;
  if (!x || !w || !y || N < 0 || C1 < 1 || C2 < 0 || C < KP || (C % KP) || M < TM || (M % TM) || P < 4 ||
      (P & 3))
    return SSAD_E_BADARG;
  if ((x2 != nullptr) != (C2 > 0)) return SSAD_E_BADARG;
  if (x2 && C != 128) return SSAD_E_BADARG;            // two inputs: the 128-channel persistent kernel only
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)x2) & 15) return SSAD_E_BADARG;
  if (N == 0) return 0;
  PwArgs a;
  a.x = x; a.x2 = x2; a.C1 = C1; a.w = w; a.bias = bias; a.res = residual; a.y = y;
  a.N = N; a.C = C; a.P = P; a.M = M; a.relu = relu;
  if (C == 64 || C == 128) {
    a.ptiles = (P + PT - 1) / PT;
    const long long tiles = (long long)N * a.ptiles;
    if (tiles >= (1LL << 31)) return SSAD_E_BADARG;
    const int cus = ssad_cu_count();
    const int mblocks = M / TM;
    const int per_cu = C == 64 ? 2 : 1;
    long long g = ((long long)per_cu * cus + mblocks - 1) / mblocks;
    if (g > tiles) g = tiles;
    const size_t lds = (size_t)(C * TM + 2 * C * PT) * sizeof(float);
    const dim3 grid((unsigned)g, (unsigned)mblocks);
    if (C == 64) {
      static const bool ok = hipFuncSetAttribute(
          reinterpret_cast<const void*>(conv1x1_fused_persistent_kernel<64, 256>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (64 * TM + 2 * 64 * PT) * 4) == hipSuccess;
      if (!ok) return SSAD_E_BADARG;
      hipLaunchKernelGGL((conv1x1_fused_persistent_kernel<64, 256>), grid, dim3(256), lds, (hipStream_t)stream,
                         a, (int)tiles);
    } else {
      static const bool ok = hipFuncSetAttribute(
          reinterpret_cast<const void*>(conv1x1_fused_persistent_kernel<128, 512>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (128 * TM + 2 * 128 * PT) * 4) == hipSuccess;
      if (!ok) return SSAD_E_BADARG;
      hipLaunchKernelGGL((conv1x1_fused_persistent_kernel<128, 512>), grid, dim3(512), lds, (hipStream_t)stream,
                         a, (int)tiles);
    }
    return (int)hipGetLastError();
  }
  a.ptiles = (P + TP - 1) / TP;
  if ((long long)N * a.ptiles >= (1LL << 31)) return SSAD_E_BADARG;
  hipLaunchKernelGGL(conv1x1_fused_pipelined_kernel, dim3((unsigned)(N * a.ptiles), (unsigned)(M / TM)),
                     dim3(kThreads), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // extern "C"
