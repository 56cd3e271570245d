// im2col.hip -- the data-movement kernels of the DEFAULT convolution engine
// (im2col + GEMM, caffe2/operators/conv_op_impl.h:31-202 and :358-577) that
// serves every geometry the matrix-core 3x3 engine does not (the backbone's
// 1x1, strided and 7x7 convolutions), plus MaxPool and the per-channel sum of
// the bias gradient.  The GEMMs themselves: kernels/gemm_general.hip
// (through csrc/c2/blas.cc, the math::Gemm front end).
//
//   Im2col / Col2im NCHW    caffe2/utils/math_gpu.cu im2col_gpu_kernel_nchw /
//                           col2im_gpu_kernel_nchw
//   MaxPool(Gradient)       caffe2/operators/pool_op.cu MaxPoolForwardNCHW /
//                           MaxPoolBackwardNCHW (window clipped to the image)

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ssad_kernels.h"

namespace {

constexpr int kT = 256;

inline unsigned grid_for(long long n) {
  long long b = (n + kT - 1) / kT;
  if (b > 65535) b = 65535;
  if (b < 1) b = 1;
  return (unsigned)b;
}

struct Geo {
  int C, H, W, kh, kw, dh, dw, pt, pl, sh, sw, OH, OW;
};

__global__ __launch_bounds__(kT) void im2col_kernel(const float* __restrict__ x, const Geo g,
                                                    float* __restrict__ col) {
  const long long total = (long long)g.C * g.kh * g.kw * g.OH * g.OW;
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < total;
       i += (long long)gridDim.x * kT) {
    const int ow = (int)(i % g.OW);
    long long r = i / g.OW;
    const int oh = (int)(r % g.OH);
    r /= g.OH;
    const int kj = (int)(r % g.kw);
    r /= g.kw;
    const int ki = (int)(r % g.kh);
    const int c = (int)(r / g.kh);
    const int h = oh * g.sh - g.pt + ki * g.dh, w = ow * g.sw - g.pl + kj * g.dw;
    col[i] = (h >= 0 && h < g.H && w >= 0 && w < g.W) ? x[((long long)c * g.H + h) * g.W + w] : 0.0f;
  }
}

// gather form (deterministic): each input element sums the column entries that read it
__global__ __launch_bounds__(kT) void col2im_kernel(const float* __restrict__ col, const Geo g,
                                                    float* __restrict__ x) {
  const long long total = (long long)g.C * g.H * g.W;
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < total;
       i += (long long)gridDim.x * kT) {
    const int w = (int)(i % g.W);
    const long long r = i / g.W;
    const int h = (int)(r % g.H);
    const int c = (int)(r / g.H);
    float acc = 0.0f;
    for (int ki = 0; ki < g.kh; ++ki) {
      const int hh = h + g.pt - ki * g.dh;
      if (hh < 0 || hh % g.sh) continue;
      const int oh = hh / g.sh;
      if (oh >= g.OH) continue;
      for (int kj = 0; kj < g.kw; ++kj) {
        const int ww = w + g.pl - kj * g.dw;
        if (ww < 0 || ww % g.sw) continue;
        const int ow = ww / g.sw;
        if (ow >= g.OW) continue;
        acc += col[((((long long)c * g.kh + ki) * g.kw + kj) * g.OH + oh) * g.OW + ow];
      }
    }
    x[i] = acc;
  }
}

// out[c] (+)= sum over n, p of dy[n][c][p]; one workgroup per channel, double partials
__global__ __launch_bounds__(kT) void channel_sum_kernel(const float* __restrict__ dy, int N, int C,
                                                         int HW, float* __restrict__ out,
                                                         int accumulate) {
  __shared__ double ws[kT / 64];
  const int c = blockIdx.x;
  double acc = 0.0;
  for (int n = 0; n < N; ++n) {
    const float* p = dy + ((long long)n * C + c) * HW;
    for (int i = threadIdx.x; i < HW; i += kT) acc += (double)p[i];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < kT / 64; ++i) t += ws[i];
    out[c] = accumulate ? out[c] + (float)t : (float)t;
  }
}

struct PoolGeo {
  long long planes;
  int H, W, kh, kw, sh, sw, pt, pl, OH, OW;
};

__global__ __launch_bounds__(kT) void max_pool_kernel(const float* __restrict__ x, const PoolGeo g,
                                                      float* __restrict__ y) {
  const long long total = g.planes * g.OH * g.OW;
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < total;
       i += (long long)gridDim.x * kT) {
    const int ow = (int)(i % g.OW);
    const long long r = i / g.OW;
    const int oh = (int)(r % g.OH);
    const long long p = r / g.OH;
    int h0 = oh * g.sh - g.pt, w0 = ow * g.sw - g.pl;
    const int h1 = min(h0 + g.kh, g.H), w1 = min(w0 + g.kw, g.W);
    h0 = max(h0, 0); w0 = max(w0, 0);
    float m = -__builtin_inff();
    const float* xp = x + p * g.H * g.W;
    for (int h = h0; h < h1; ++h)
      for (int w = w0; w < w1; ++w) m = fmaxf(m, xp[h * g.W + w]);
    y[i] = m;
  }
}

// The ResNet stem's pool (ResNet.py:166-168: 3x3, stride 2, pad 1) with the preceding
// AffineChannel bias and ReLU folded in: both are monotonic per channel, so
// relu(max(window) + b[c]) == max over the window of relu(x + b[c]).
// kVec (W % 4 == 0): a thread makes two adjacent outputs from one 16-byte load per
// window row; the left neighbour column comes from the previous lane's load.
template <bool kVec>
__global__ __launch_bounds__(kT) void pool3x3s2_bias_relu_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ bias,
                                                                 long long planes, int C, int H, int W,
                                                                 int OH, int OW, int relu,
                                                                 float* __restrict__ y) {
  if (kVec) {
    const int OW2 = OW >> 1;
    const long long total = planes * OH * OW2;
    const long long i = (long long)blockIdx.x * kT + threadIdx.x;   // grid covers total exactly
    const bool live = i < total;
    const long long ii = live ? i : total - 1;
    const int j = (int)(ii % OW2);
    const long long r = ii / OW2;
    const int oh = (int)(r % OH);
    const long long p = r / OH;
    const float* xp = x + p * H * W;
    const int lane = threadIdx.x & 63;
    float m0 = -__builtin_inff(), m1 = m0;
#pragma unroll
    for (int d = -1; d <= 1; ++d) {
      const int h = 2 * oh + d;
      const bool ok = h >= 0 && h < H;                // uniform per output row, not per wave
      const float* row = xp + (long long)(ok ? h : 2 * oh) * W;
      const float4 v = *reinterpret_cast<const float4*>(row + 4 * j);
      float left = __shfl_up(v.w, 1);
      if (lane == 0 && j > 0) left = row[4 * j - 1];
      if (j == 0) left = v.x;
      if (ok) {
        m0 = fmaxf(m0, fmaxf(left, fmaxf(v.x, v.y)));
        m1 = fmaxf(m1, fmaxf(v.y, fmaxf(v.z, v.w)));
      }
    }
    if (!live) return;
    if (bias) { const float b = bias[(int)(p % C)]; m0 += b; m1 += b; }
    if (relu) { m0 = fmaxf(m0, 0.0f); m1 = fmaxf(m1, 0.0f); }
    *reinterpret_cast<float2*>(y + (p * OH + oh) * OW + 2 * j) = make_float2(m0, m1);
    return;
  }
  const long long total = planes * OH * OW;
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < total;
       i += (long long)gridDim.x * kT) {
    const int ow = (int)(i % OW);
    const long long r = i / OW;
    const int oh = (int)(r % OH);
    const long long p = r / OH;
    const float* xp = x + p * H * W;
    const int w0 = 2 * ow;
    float m = -__builtin_inff();
#pragma unroll
    for (int d = -1; d <= 1; ++d) {
      const int h = 2 * oh + d;
      if (h < 0 || h >= H) continue;
      const float* row = xp + (long long)h * W;
      m = fmaxf(m, row[w0]);
      if (w0 + 1 < W) m = fmaxf(m, row[w0 + 1]);
      if (w0 > 0) m = fmaxf(m, row[w0 - 1]);
    }
    if (bias) m += bias[(int)(p % C)];
    y[i] = relu ? fmaxf(m, 0.0f) : m;
  }
}

// pool_op.cu MaxPoolBackwardNCHW: every input equal to its window's maximum receives that
// window's gradient (gather form, deterministic)
__global__ __launch_bounds__(kT) void max_pool_grad_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ y,
                                                           const float* __restrict__ dy,
                                                           const PoolGeo g, float* __restrict__ dx) {
  const long long total = g.planes * g.H * g.W;
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < total;
       i += (long long)gridDim.x * kT) {
    const int w = (int)(i % g.W);
    const long long r = i / g.W;
    const int h = (int)(r % g.H);
    const long long p = r / g.H;
    const int ph0 = (h + g.pt < g.kh) ? 0 : (h + g.pt - g.kh) / g.sh + 1;
    const int ph1 = min((h + g.pt) / g.sh + 1, g.OH);
    const int pw0 = (w + g.pl < g.kw) ? 0 : (w + g.pl - g.kw) / g.sw + 1;
    const int pw1 = min((w + g.pl) / g.sw + 1, g.OW);
    const float v = x[i];
    float acc = 0.0f;
    for (int oh = ph0; oh < ph1; ++oh)
      for (int ow = pw0; ow < pw1; ++ow) {
        const long long o = (p * g.OH + oh) * g.OW + ow;
        if (v == y[o]) acc += dy[o];
      }
    dx[i] = acc;
  }
}

}  // namespace

extern "C" {

int ssad_conv_out_size(int in, int kernel, int dilation, int pad_a, int pad_b, int stride) {
  const int eff = dilation * (kernel - 1) + 1;
  if (stride < 1 || in + pad_a + pad_b < eff) return -1;
  return (in + pad_a + pad_b - eff) / stride + 1;
}

int ssad_im2col(const float* x, int C, int H, int W, int kh, int kw, int dil_h, int dil_w, int pad_t,
                int pad_l, int pad_b, int pad_r, int stride_h, int stride_w, float* col,
                ssad_stream_t stream) {
  Geo g{C, H, W, kh, kw, dil_h, dil_w, pad_t, pad_l, stride_h, stride_w,
        ssad_conv_out_size(H, kh, dil_h, pad_t, pad_b, stride_h),
        ssad_conv_out_size(W, kw, dil_w, pad_l, pad_r, stride_w)};
  if (!x || !col || C < 1 || g.OH < 1 || g.OW < 1) return SSAD_E_BADARG;
  const long long total = (long long)C * kh * kw * g.OH * g.OW;
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total)), dim3(kT), 0, (hipStream_t)stream, x, g, col);
  return (int)hipGetLastError();
}

// Batched im2col, square kernel: blockIdx.z = (image, column-matrix row k = (c, ki, kj)) -- decoded
// once per workgroup -- and every thread writes four consecutive output pixels of that row with one
// 16-byte store (the per-image kernel above decodes six div / mod per 4-byte element: 0.66 TB/s on
// the stem's 84 MB-per-image column matrix; this one streams at the store rate).
__global__ __launch_bounds__(kT) void im2col_rows_kernel(const float* __restrict__ x, const Geo g, int rows,
                                                         float* __restrict__ col) {
  const int z = blockIdx.z;
  const int n = z / rows, k = z - n * rows;
  const int kj = k % g.kw, ki = (k / g.kw) % g.kh, c = k / (g.kw * g.kh);
  const int ow4 = g.OW >> 2;
  const int p4 = blockIdx.x * kT + threadIdx.x;
  if (p4 >= g.OH * ow4) return;
  const int oh = p4 / ow4, ox = (p4 - oh * ow4) * 4;
  const int h = oh * g.sh - g.pt + ki * g.dh;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (h >= 0 && h < g.H) {
    const float* row = x + (((long long)n * g.C + c) * g.H + h) * g.W;
    const int w0 = ox * g.sw - g.pl + kj * g.dw;
    float e[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int w = w0 + q * g.sw;
      e[q] = (w >= 0 && w < g.W) ? row[w] : 0.0f;
    }
    v = make_float4(e[0], e[1], e[2], e[3]);
  }
  reinterpret_cast<float4*>(col + ((long long)z * g.OH + oh) * g.OW)[ox >> 2] = v;
}

// the same for a whole batch: col[n][C*kh*kw][OH*OW] (the X operand of ssad_conv1x1_gemm for a
// k x k convolution: the ResNet stem's 7x7 / stride 2)
int ssad_im2col_batched(const float* x, int N, int C, int H, int W, int kernel, int stride, int pad,
                        float* col, ssad_stream_t stream) {
  Geo g{C, H, W, kernel, kernel, 1, 1, pad, pad, stride, stride,
        ssad_conv_out_size(H, kernel, 1, pad, pad, stride), ssad_conv_out_size(W, kernel, 1, pad, pad, stride)};
  if (!x || !col || N < 0 || C < 1 || g.OH < 1 || g.OW < 1) return SSAD_E_BADARG;
  const long long per = (long long)C * kernel * kernel * g.OH * g.OW;
  const int rows = C * kernel * kernel;
  if (N == 0) return 0;
  if ((g.OW & 3) == 0 && (((uintptr_t)col) & 15) == 0 && (long long)N * rows < 65536) {
    const int p4 = g.OH * (g.OW >> 2);
    hipLaunchKernelGGL(im2col_rows_kernel, dim3((p4 + kT - 1) / kT, 1, N * rows), dim3(kT), 0,
                       (hipStream_t)stream, x, g, rows, col);
    return (int)hipGetLastError();
  }
  for (int n = 0; n < N; ++n)
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(per)), dim3(kT), 0, (hipStream_t)stream,
                       x + (long long)n * C * H * W, g, col + (long long)n * per);
  return (int)hipGetLastError();
}

int ssad_col2im(const float* col, int C, int H, int W, int kh, int kw, int dil_h, int dil_w,
                int pad_t, int pad_l, int pad_b, int pad_r, int stride_h, int stride_w, float* x,
                ssad_stream_t stream) {
  Geo g{C, H, W, kh, kw, dil_h, dil_w, pad_t, pad_l, stride_h, stride_w,
        ssad_conv_out_size(H, kh, dil_h, pad_t, pad_b, stride_h),
        ssad_conv_out_size(W, kw, dil_w, pad_l, pad_r, stride_w)};
  if (!x || !col || C < 1 || g.OH < 1 || g.OW < 1) return SSAD_E_BADARG;
  hipLaunchKernelGGL(col2im_kernel, dim3(grid_for((long long)C * H * W)), dim3(kT), 0,
                     (hipStream_t)stream, col, g, x);
  return (int)hipGetLastError();
}

int ssad_channel_sum(const float* dy, int N, int C, int HW, float* out, int accumulate,
                     ssad_stream_t stream) {
  if (!dy || !out || N < 0 || C < 1 || HW < 0) return SSAD_E_BADARG;
  hipLaunchKernelGGL(channel_sum_kernel, dim3(C), dim3(kT), 0, (hipStream_t)stream, dy, N, C, HW,
                     out, accumulate);
  return (int)hipGetLastError();
}

int ssad_max_pool_forward(const float* x, int N, int C, int H, int W, int kh, int kw, int stride_h,
                          int stride_w, int pad_t, int pad_l, int pad_b, int pad_r, float* y,
                          ssad_stream_t stream) {
  PoolGeo g{(long long)N * C, H, W, kh, kw, stride_h, stride_w, pad_t, pad_l,
            ssad_conv_out_size(H, kh, 1, pad_t, pad_b, stride_h),
            ssad_conv_out_size(W, kw, 1, pad_l, pad_r, stride_w)};
  if (!x || !y || g.OH < 1 || g.OW < 1 || g.planes < 0) return SSAD_E_BADARG;
  if (g.planes == 0) return 0;
  hipLaunchKernelGGL(max_pool_kernel, dim3(grid_for(g.planes * g.OH * g.OW)), dim3(kT), 0,
                     (hipStream_t)stream, x, g, y);
  return (int)hipGetLastError();
}

int ssad_max_pool3x3s2_bias_relu(const float* x, const float* bias, int N, int C, int H, int W,
                                 int relu, float* y, ssad_stream_t stream) {
  const int OH = ssad_conv_out_size(H, 3, 1, 1, 1, 2), OW = ssad_conv_out_size(W, 3, 1, 1, 1, 2);
  if (!x || !y || N < 0 || C < 1 || OH < 1 || OW < 1) return SSAD_E_BADARG;
  const long long planes = (long long)N * C;
  if (planes == 0) return 0;
  if ((W & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
    const long long pairs = planes * OH * (OW >> 1);
    if ((pairs + kT - 1) / kT >= (1LL << 31)) return SSAD_E_BADARG;
    hipLaunchKernelGGL(pool3x3s2_bias_relu_kernel<true>, dim3((unsigned)((pairs + kT - 1) / kT)),
                       dim3(kT), 0, (hipStream_t)stream, x, bias, planes, C, H, W, OH, OW, relu, y);
  } else {
    hipLaunchKernelGGL(pool3x3s2_bias_relu_kernel<false>, dim3(grid_for(planes * OH * OW)), dim3(kT),
                       0, (hipStream_t)stream, x, bias, planes, C, H, W, OH, OW, relu, y);
  }
  return (int)hipGetLastError();
}

int ssad_max_pool_backward(const float* x, const float* y, const float* dy, int N, int C, int H,
                           int W, int kh, int kw, int stride_h, int stride_w, int pad_t, int pad_l,
                           int pad_b, int pad_r, float* dx, ssad_stream_t stream) {
  PoolGeo g{(long long)N * C, H, W, kh, kw, stride_h, stride_w, pad_t, pad_l,
            ssad_conv_out_size(H, kh, 1, pad_t, pad_b, stride_h),
            ssad_conv_out_size(W, kw, 1, pad_l, pad_r, stride_w)};
  if (!x || !y || !dy || !dx || g.OH < 1 || g.OW < 1 || g.planes < 0) return SSAD_E_BADARG;
  if (g.planes == 0) return 0;
  hipLaunchKernelGGL(max_pool_grad_kernel, dim3(grid_for(g.planes * H * W)), dim3(kT), 0,
                     (hipStream_t)stream, x, y, dy, g, dx);
  return (int)hipGetLastError();
}

}  // extern "C"
