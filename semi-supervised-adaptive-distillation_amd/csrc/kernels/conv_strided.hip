// conv_strided.hip -- gradients of a k x k convolution with a stride (group 1, fp32, NCHW): FPN's P6 / P7
// (3x3, stride 2: detectron/lib/modeling/FPN.py:193-224) at the layer's OWN size.  Until round 3 these two
// layers ran as the stride-1 layer + subsampling (4x the direct-form flops, zero-stuffed gradients).
//
// The reference's algorithm (caffe2/operators/conv_op_impl.h:358-577) is per image: im2col, dW += dY . col^T,
// dcol = W^T . dY, col2im.  Same sums here, over the whole batch at once -- a P6 / P7 map is 140 / 35 pixels per
// image, far too few columns for one GEMM tile, so the batch is FLATTENED into the column index q = n * P + p
// (rows padded with zeros to a multiple of 16 columns):
//
//   filter gradient   col[K9][Q] = im2col(x);  dyT[M][Q] = dY;   dW[m][k] (+)= sum_q dyT[m][q] col[k][q]
//                     = ssad_conv1x1_wgrad on one "image" of Q pixels (gemm_conv_nt_kernel: 288 tiles for P6,
//                     deterministic split reduction)
//   data gradient     dcol[K9][Q] = W^T[K9][M] . dyT[M][Q]   (ssad_conv1x1_gemm: the filter in its natural
//                     [M][K9] layout is the [K][M'] operand), then dx = col2im(dcol) in GATHER form: every input
//                     element sums the <= ceil(k/s)^2 column entries that read it (no atomics; ReluGradient
//                     mask and accumulation in the same pass)
//
// The forward pass of these layers is ssad_conv_implicit_gemm_ws (gemm_conv.hip: im2col gathered by the DMA,
// split-K over the 18 432-row reduction).  The three small passes below are HBM streams (P6 at bs 16: 165 MB).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_internal.h"
#include "ssad_kernels.h"

namespace {

constexpr int kThreads = 256;

struct Geo {
  int N, C, H, W, k, s, pad, OH, OW, P;
  long long Q, Qpad;
};

inline bool make_geo(int N, int C, int H, int W, int kernel, int stride, int pad, Geo* g) {
  if (N < 1 || C < 1 || H < 1 || W < 1 || kernel < 1 || stride < 1 || pad < 0) return false;
  if (H + 2 * pad < kernel || W + 2 * pad < kernel) return false;
  g->N = N; g->C = C; g->H = H; g->W = W; g->k = kernel; g->s = stride; g->pad = pad;
  g->OH = (H + 2 * pad - kernel) / stride + 1;
  g->OW = (W + 2 * pad - kernel) / stride + 1;
  g->P = g->OH * g->OW;
  g->Q = (long long)N * g->P;
  g->Qpad = (g->Q + 15) & ~15LL;
  return true;
}

// channels one workgroup walks for its 256 columns / pixels: up to 16, fewer while the grid would not fill the chip
inline int chan_per_block(long long col_blocks, int C) {
  int cpb = 16;
  while (cpb > 1 && col_blocks * ((C + cpb - 1) / cpb) < 2048) cpb >>= 1;
  return cpb;
}

// col[(c, ky, kx)][q] = x[n][c][oy * s + ky - pad][ox * s + kx - pad] (0 outside the image and for q >= Q).
// A thread owns ONE column q -- its (n, oy, ox) is decomposed once -- and walks kChanPerBlock channels x k x k taps:
// stores are coalesced along q, the index arithmetic is paid once per thread.
__global__ __launch_bounds__(kThreads) void im2col_flat_kernel(const float* __restrict__ x, const Geo g, int kChanPerBlock,
                                                               float* __restrict__ col) {
  const unsigned q = blockIdx.x * kThreads + threadIdx.x;
  if (q >= (unsigned)g.Qpad) return;
  const int c0 = blockIdx.y * kChanPerBlock;
  const int c1 = c0 + kChanPerBlock < g.C ? c0 + kChanPerBlock : g.C;
  const bool live = q < (unsigned)g.Q;
  const unsigned n = live ? q / (unsigned)g.P : 0, p = live ? q - n * (unsigned)g.P : 0;
  const int oy = (int)(p / (unsigned)g.OW), ox = (int)(p - (unsigned)oy * g.OW);
  const int iy0 = oy * g.s - g.pad, ix0 = ox * g.s - g.pad;
  const int kk = g.k * g.k;
  const size_t plane = (size_t)g.H * g.W;
  const float* xn = x + (size_t)n * g.C * plane;
  for (int c = c0; c < c1; ++c) {
    const float* xc = xn + (size_t)c * plane;
    float* out = col + (size_t)c * kk * g.Qpad + q;
    for (int ky = 0; ky < g.k; ++ky) {
      const int iy = iy0 + ky;
      const bool oky = live && (unsigned)iy < (unsigned)g.H;
      for (int kx = 0; kx < g.k; ++kx) {
        const int ix = ix0 + kx;
        float v = 0.0f;
        if (oky && (unsigned)ix < (unsigned)g.W) v = xc[iy * g.W + ix];
        out[(size_t)(ky * g.k + kx) * g.Qpad] = v;
      }
    }
  }
}

// dyT[m][q] = dy[n][m][p], q = n * P + p (0 for q >= Q)
__global__ __launch_bounds__(kThreads) void flatten_rows_kernel(const float* __restrict__ dy, int M, int P, long long Q,
                                                                long long Qpad, float* __restrict__ out) {
  const int m = blockIdx.y;
  for (long long q = (long long)blockIdx.x * kThreads + threadIdx.x; q < Qpad; q += (long long)gridDim.x * kThreads) {
    float v = 0.0f;
    if (q < Q) {
      const long long n = q / P;
      v = dy[(n * M + m) * P + (q - n * P)];
    }
    out[(long long)m * Qpad + q] = v;
  }
}

// dx[n][c][iy][ix] (+)= mask( sum over the taps (ky, kx) with (iy + pad - ky) % s == 0, (ix + pad - kx) % s == 0
//                             of dcol[(c, ky, kx)][n * P + oy * OW + ox] )
// GATHER form (no atomics).  A thread owns one input pixel (n, iy, ix): the taps that read it -- ceil(k / s)^2 at
// most, an arithmetic progression along each axis -- are worked out once, then kChanPerBlock channels are walked
// with them; stores are coalesced along the pixel index.

__global__ __launch_bounds__(kThreads) void col2im_flat_kernel(const float* __restrict__ dcol, const Geo g, int kChanPerBlock,
                                                               const float* __restrict__ mask, int accumulate,
                                                               float* __restrict__ dx) {
  const unsigned plane = (unsigned)(g.H * g.W);
  const unsigned pix = blockIdx.x * kThreads + threadIdx.x;              // n * H * W + iy * W + ix
  if (pix >= (unsigned)g.N * plane) return;
  const unsigned n = pix / plane, r = pix - n * plane;
  const int iy = (int)(r / (unsigned)g.W), ix = (int)(r - (unsigned)iy * g.W);
  // taps along y that read this row: ky = ky0 + a * s (a < ny), output row oy0 - a; same along x
  const int ty = iy + g.pad, tx = ix + g.pad;
  int ky0 = ty % g.s, kx0 = tx % g.s;
  if (ty - (g.OH - 1) * g.s > ky0) ky0 = ty - (g.OH - 1) * g.s;
  if (tx - (g.OW - 1) * g.s > kx0) kx0 = tx - (g.OW - 1) * g.s;
  const int ky1 = ty < g.k - 1 ? ty : g.k - 1, kx1 = tx < g.k - 1 ? tx : g.k - 1;
  const int ny = ky0 <= ky1 ? (ky1 - ky0) / g.s + 1 : 0, nx = kx0 <= kx1 ? (kx1 - kx0) / g.s + 1 : 0;
  const int oy0 = (ty - ky0) / g.s, ox0 = (tx - kx0) / g.s;
  const int c0 = blockIdx.y * kChanPerBlock;
  const int c1 = c0 + kChanPerBlock < g.C ? c0 + kChanPerBlock : g.C;
  const int kk = g.k * g.k;
  const size_t qn = (size_t)n * g.P;
  for (int c = c0; c < c1; ++c) {
    const float* dc = dcol + (size_t)c * kk * g.Qpad + qn;
    float v = 0.0f;
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b)
        v += dc[(size_t)((ky0 + a * g.s) * g.k + kx0 + b * g.s) * g.Qpad + (oy0 - a) * g.OW + ox0 - b];
    const size_t i = ((size_t)n * g.C + c) * plane + r;
    if (mask) v = mask[i] > 0.0f ? v : 0.0f;
    dx[i] = accumulate ? dx[i] + v : v;
  }
}

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

inline unsigned blocks_for(long long n) {
  long long b = (n + kThreads - 1) / kThreads;
  if (b > 64) b = 64;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" {

size_t ssad_conv_kxk_wgrad_workspace_bytes(int N, int C, int H, int W, int M, int kernel, int stride, int pad) {
  Geo g;
  if (M < 1 || !make_geo(N, C, H, W, kernel, stride, pad, &g)) return 0;
  const long long K9 = (long long)C * kernel * kernel;
  if (M > 65535 || K9 * g.Qpad * 4 >= (1LL << 31) || (long long)M * g.Qpad * 4 >= (1LL << 31)) return 0;
  return align256((size_t)K9 * g.Qpad * 4) + align256((size_t)M * g.Qpad * 4) +
         ssad_conv1x1_wgrad_workspace_bytes(1, (int)K9, (int)g.Qpad, M);
}

int ssad_conv_kxk_wgrad(const float* x, const float* dy, int N, int C, int H, int W, int M, int kernel, int stride,
                        int pad, float* dw, int accumulate, void* workspace, size_t workspace_bytes,
                        ssad_stream_t stream) {
  Geo g;
  if (!x || !dy || !dw || M < 1 || !make_geo(N, C, H, W, kernel, stride, pad, &g)) return SSAD_E_BADARG;
  const long long K9 = (long long)C * kernel * kernel;
  const size_t need = ssad_conv_kxk_wgrad_workspace_bytes(N, C, H, W, M, kernel, stride, pad);
  if (need == 0) return SSAD_E_BADARG;               // a buffer of 2 GiB or more, or more rows than a grid has
  if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < need) return SSAD_E_WORKSPACE;
  char* ws = (char*)workspace;
  float* col = (float*)ws;
  float* dyT = (float*)(ws + align256((size_t)K9 * g.Qpad * 4));
  char* gemm_ws = (char*)dyT + align256((size_t)M * g.Qpad * 4);
  hipStream_t s = (hipStream_t)stream;
  const long long qblocks = (g.Qpad + kThreads - 1) / kThreads;
  const int cpb = chan_per_block(qblocks, C);
  hipLaunchKernelGGL(im2col_flat_kernel, dim3((unsigned)qblocks, (unsigned)((C + cpb - 1) / cpb)), dim3(kThreads), 0, s, x, g,
                     cpb, col);
  hipLaunchKernelGGL(flatten_rows_kernel, dim3(blocks_for(g.Qpad), (unsigned)M), dim3(kThreads), 0, s, dy, M, g.P, g.Q,
                     g.Qpad, dyT);
  { const int e = (int)hipGetLastError(); if (e) return e; }
  return ssad_conv1x1_wgrad(col, dyT, 1, (int)K9, (int)g.Qpad, M, dw, accumulate, gemm_ws,
                            workspace_bytes - (size_t)(gemm_ws - ws), stream);
}

size_t ssad_conv_kxk_dgrad_workspace_bytes(int N, int C, int H, int W, int M, int kernel, int stride, int pad) {
  Geo g;
  if (M < 1 || !make_geo(N, C, H, W, kernel, stride, pad, &g)) return 0;
  const long long K9 = (long long)C * kernel * kernel;
  if (M > 65535 || K9 * g.Qpad * 4 >= (1LL << 31) || (long long)M * g.Qpad * 4 >= (1LL << 31)) return 0;
  return align256((size_t)K9 * g.Qpad * 4) + align256((size_t)M * g.Qpad * 4);
}

int ssad_conv_kxk_dgrad(const float* w, const float* dy, int N, int C, int H, int W, int M, int kernel, int stride,
                        int pad, float* dx, const float* mask, int accumulate, void* workspace,
                        size_t workspace_bytes, ssad_stream_t stream) {
  Geo g;
  if (!w || !dy || !dx || M < 1 || !make_geo(N, C, H, W, kernel, stride, pad, &g)) return SSAD_E_BADARG;
  const long long K9 = (long long)C * kernel * kernel;
  if ((K9 & 3) || ((uintptr_t)w & 15)) return SSAD_E_BADARG;          // the filter is the GEMM's [M][K9] operand
  const size_t need = ssad_conv_kxk_dgrad_workspace_bytes(N, C, H, W, M, kernel, stride, pad);
  if (need == 0) return SSAD_E_BADARG;
  if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < need) return SSAD_E_WORKSPACE;
  char* ws = (char*)workspace;
  float* dcol = (float*)ws;
  float* dyT = (float*)(ws + align256((size_t)K9 * g.Qpad * 4));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(flatten_rows_kernel, dim3(blocks_for(g.Qpad), (unsigned)M), dim3(kThreads), 0, s, dy, M, g.P, g.Q,
                     g.Qpad, dyT);
  { const int e = (int)hipGetLastError(); if (e) return e; }
  ssad_gemm_conv d;
  d.a = w; d.x = dyT; d.y = dcol; d.bias = nullptr; d.residual = nullptr; d.mask = nullptr;
  d.lda = (int)K9; d.N = 1; d.K = M; d.P = (int)g.Qpad; d.M = (int)K9; d.flags = 0;
  const int rc = ssad_conv1x1_gemm(&d, stream);
  if (rc) return rc;
  const long long pblocks = ((long long)N * H * W + kThreads - 1) / kThreads;
  const int cpb = chan_per_block(pblocks, C);
  hipLaunchKernelGGL(col2im_flat_kernel, dim3((unsigned)pblocks, (unsigned)((C + cpb - 1) / cpb)), dim3(kThreads), 0, s,
                     (const float*)dcol, g, cpb, mask, accumulate, dx);
  return (int)hipGetLastError();
}

}  // extern "C"
