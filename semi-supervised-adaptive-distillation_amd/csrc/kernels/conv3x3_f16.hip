// conv3x3_f16.hip -- the RetinaNet subnet convolution with fp16 storage and fp32
// accumulation on the gfx950 matrix cores (v_mfma_f32_32x32x16_f16): BASELINE config 5's
// precision.  The reference's only fp16 route is CudnnConvOp<float16> with fp32 math
// (caffe2/operators/conv_op_cudnn.cc:631-636); the arithmetic here is the same contract:
// fp16 operands, every product and sum in fp32, one rounding when a result is stored.
//
// Layout.  Inside the subnet pipeline activations live channel-blocked,
//     Xb[n][c/8][y][x][c%8]          (16 bytes = the 8 channels of one pixel),
// so that one 16-byte load IS an MFMA operand (a lane supplies 8 consecutive K = channels)
// for every filter tap -- NCHW would need a transpose through LDS per tap, and its +-1 pixel
// tap shifts are 2-byte misaligned for ds_read_b64_tr_b16.  Conversion from / to the
// operator interface's NCHW fp32 happens once at the FPN inputs (pack kernel below) and in
// the epilogue of the prediction layers (SSAD_F16_OUT_NCHW_F32).
//
// Kernel.  Workgroup = 4 waves, output tile 128 channels x (16 x 16) pixels; wave (wo, wp)
// owns 64 channels x 8 tile rows = 2 x 4 MFMA tiles of 32 x 32 (128 accumulator VGPRs).
// K runs over chunks of 32 input channels; a chunk's 18 x 18 x 32 halo tile is staged
// through LDS (double buffered, one barrier per chunk) and read back with ds_read_b128 at
// compile-time offsets per (tap, tile); the filter operand is read straight from the
// packed, L2-resident Wp[tap][c/8][m][c%8] (one 16-byte load per lane per 32 channels x 16 K).
// Per K-step (one tap, 16 channels): 2 filter loads + 4 LDS reads feed 8 MFMAs.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssad_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int TS = 16;                 // output tile edge
constexpr int HS = TS + 2;             // halo tile edge
constexpr int CBC = 4;                 // 8-channel blocks per K chunk (32 channels)
constexpr int SLOTS = CBC * HS * HS;   // 16-byte slots per stage (1296)
constexpr int NLD = (SLOTS + kThreads - 1) / kThreads;   // staging loads per thread (6)
constexpr int MT = 128;                // output channels per workgroup

struct F16Conv {
  const uint4* x;        // blocked fp16 input
  const uint4* w;        // packed filter [9][C/8][M] x 16 B
  const float* bias;     // [M] or null
  void* y;               // blocked fp16 or NCHW fp32
  int N, C, H, W, M;
  int tiles_x, tiles_y;
  int relu, out_nchw_f32;
};

__device__ __forceinline__ half8 as_half8(const uint4& v) {
  return __builtin_bit_cast(half8, v);
}

__global__ __launch_bounds__(kThreads) void conv3x3_f16_kernel(const F16Conv p) {
  __shared__ uint4 lds[2 * SLOTS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wo = wave & 1, wp = wave >> 1;
  const int j = lane & 31, h = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int n = t / p.tiles_y;
  const int y0 = ty * TS, x0 = tx * TS;
  const int ocb = blockIdx.y * MT;
  const int CB = (p.C + 7) >> 3;          // channel blocks; a tail block is zero padded by the packers
  const long long plane = (long long)p.H * p.W;

  // ---- staging plan: slot s = tid + 256 i  ->  (block, row, col) of the halo tile
  long long goff[NLD];
  bool gok[NLD];
  int gcb[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int s = tid + kThreads * i;
    const int cbl = s / (HS * HS), r = s % (HS * HS);
    gcb[i] = cbl;
    const int gy = y0 - 1 + r / HS, gx = x0 - 1 + r % HS;
    gok[i] = s < SLOTS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    goff[i] = ((long long)n * CB + cbl) * plane + (long long)gy * p.W + gx;
  }
  uint4 stage[NLD];
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      stage[i] = make_uint4(0u, 0u, 0u, 0u);
      if (gok[i] && chunk * CBC + gcb[i] < CB)        // beyond the last block: K padding = 0
        stage[i] = p.x[goff[i] + (long long)chunk * CBC * plane];
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int s = tid + kThreads * i;
      if (s < SLOTS) lds[buf * SLOTS + s] = stage[i];
    }
  };

  // ---- operand addressing
  // B (pixels): MFMA column j -> tile row 2 tt + (j >> 4) of this wave's 8 rows, column j & 15
  const int brow = wp * 8 + (j >> 4), bcol = j & 15;
  const int bbase = (h * HS + brow) * HS + bcol;           // + ((2 ks) * HS + 2 tt + dy) * HS + dx
  // A (filter): row = output channel, clamped into range (rows beyond M are never stored)
  int aoc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oc = ocb + wo * 64 + i * 32 + j;
    aoc[i] = oc < p.M ? oc : p.M - 1;
  }
  auto load_a = [&](int chunk, int q, half8 (&a)[2]) {     // q = ks * 9 + tap
    const int ks = q / 9, tap = q % 9;
    int cb = chunk * CBC + 2 * ks + h;                     // a block beyond the last meets B = 0
    cb = cb < CB ? cb : CB - 1;
    const long long row = ((long long)tap * CB + cb) * p.M;
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = as_half8(p.w[row + aoc[i]]);
  };

  float16v acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][tt][r] = 0.0f;

  const int nchunks = (CB + CBC - 1) / CBC;
  fetch(0);
  stash(0);
  __syncthreads();
  half8 a_cur[2], a_nxt[2];
  load_a(0, 0, a_cur);
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) fetch(c + 1);
    const uint4* tile = lds + (c & 1) * SLOTS;
#pragma unroll
    for (int q = 0; q < 18; ++q) {
      const int ks = q / 9, tap = q % 9, dy = tap / 3, dx = tap % 3;
      if (q + 1 < 18) load_a(c, q + 1, a_nxt);
      else if (more) load_a(c + 1, 0, a_nxt);
      half8 b[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
        b[tt] = as_half8(tile[bbase + ((2 * ks) * HS + 2 * tt + dy) * HS + dx]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
          acc[i][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[i], b[tt], acc[i][tt], 0, 0, 0);
      a_cur[0] = a_nxt[0];
      a_cur[1] = a_nxt[1];
    }
    if (more) stash((c + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: C/D row = (r & 3) + 8 (r >> 2) + 4 h, column = j
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oc0 = ocb + wo * 64 + i * 32;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int gy = y0 + wp * 8 + 2 * tt + (j >> 4), gx = x0 + (j & 15);
      if (gy >= p.H || gx >= p.W) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int oc = oc0 + 8 * g + 4 * h;          // first of this lane's 4 channels
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[i][tt][4 * g + e];
          if (p.bias && oc + e < p.M) v[e] += p.bias[oc + e];
          if (p.relu) v[e] = fmaxf(v[e], 0.0f);
        }
        if (p.out_nchw_f32) {
          float* yo = static_cast<float*>(p.y);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (oc + e < p.M) yo[(((long long)n * p.M + oc + e) * p.H + gy) * p.W + gx] = v[e];
        } else if (oc < p.M) {                        // M % 8 == 0 on this path
          half4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
          _Float16* yo = static_cast<_Float16*>(p.y);
          const long long slot = ((long long)n * (p.M >> 3) + (oc >> 3)) * plane + (long long)gy * p.W + gx;
          *reinterpret_cast<half4*>(yo + slot * 8 + 4 * h) = o;
        }
      }
    }
  }
}

// NCHW fp32 -> blocked fp16 (round to nearest even), one thread per 16-byte slot
__global__ __launch_bounds__(kThreads) void f16_pack_kernel(const float* __restrict__ x,
                                                            uint4* __restrict__ xb, int N, int C,
                                                            long long plane) {
  const int CB = (C + 7) >> 3;
  const long long total = (long long)N * CB * plane;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (long long)gridDim.x * kThreads) {
    const long long px = i % plane, ncb = i / plane;      // ncb = n * CB + cb
    const int cb = (int)(ncb % CB);
    const float* src = x + ((ncb / CB) * C + cb * 8) * plane + px;
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = cb * 8 + e < C ? (_Float16)src[e * plane] : (_Float16)0.0f;
    xb[i] = __builtin_bit_cast(uint4, o);
  }
}

// blocked fp16 -> NCHW fp32: one thread per (n, cb, pixel); 8 strided 4-byte stores
__global__ __launch_bounds__(kThreads) void f16_unpack_kernel(const uint4* __restrict__ xb,
                                                              float* __restrict__ x, int N, int C,
                                                              long long plane) {
  const int CB = (C + 7) >> 3;
  const long long total = (long long)N * CB * plane;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (long long)gridDim.x * kThreads) {
    const long long px = i % plane, ncb = i / plane;
    const int cb = (int)(ncb % CB);
    const half8 v = as_half8(xb[i]);
    float* dst = x + ((ncb / CB) * C + cb * 8) * plane + px;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (cb * 8 + e < C) dst[e * plane] = (float)v[e];
  }
}

// Filter [M][C][3][3] fp32 -> Wp[tap][C/8][M][8] fp16 (forward), and the data-gradient form
// Wd[tap][M/8][C][8] = W[m][c][2 - ky][2 - kx] with the roles of M and C exchanged
// (conv_op_impl.h:524-560: dX = col2im(W^T dY) = correlation of dY with the flipped filter).
__global__ __launch_bounds__(kThreads) void f16_pack_filter_kernel(const float* __restrict__ w, int M,
                                                                   int C, uint4* __restrict__ wf,
                                                                   uint4* __restrict__ wd) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  const int CB = (C + 7) >> 3, MB = (M + 7) >> 3;
  if (wf && i < 9 * CB * M) {
    const int m = i % M, cb = (i / M) % CB, tap = i / (M * CB);
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e] = cb * 8 + e < C ? (_Float16)w[((long long)m * C + cb * 8 + e) * 9 + tap] : (_Float16)0.0f;
    wf[i] = __builtin_bit_cast(uint4, o);
  }
  if (wd && i < 9 * MB * C) {
    const int c = i % C, mb = (i / C) % MB, tap = i / (C * MB);
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e] = mb * 8 + e < M ? (_Float16)w[((long long)(mb * 8 + e) * C + c) * 9 + (8 - tap)]
                            : (_Float16)0.0f;
    wd[i] = __builtin_bit_cast(uint4, o);
  }
}

inline unsigned grid_for(long long n) {
  const long long b = (n + kThreads - 1) / kThreads;
  return (unsigned)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b));
}

}  // namespace

extern "C" {

int ssad_f16_pack_activations(const float* x, int N, int C, int H, int W, void* xb,
                              ssad_stream_t stream) {
  if (!x || !xb || N < 0 || C < 1 || H < 1 || W < 1) return SSAD_E_BADARG;
  if (N == 0) return 0;
  const long long plane = (long long)H * W;
  hipLaunchKernelGGL(f16_pack_kernel, dim3(grid_for((long long)N * ((C + 7) >> 3) * plane)), dim3(kThreads),
                     0, (hipStream_t)stream, x, static_cast<uint4*>(xb), N, C, plane);
  return (int)hipGetLastError();
}

int ssad_f16_unpack_activations(const void* xb, int N, int C, int H, int W, float* x,
                                ssad_stream_t stream) {
  if (!x || !xb || N < 0 || C < 1 || H < 1 || W < 1) return SSAD_E_BADARG;
  if (N == 0) return 0;
  const long long plane = (long long)H * W;
  hipLaunchKernelGGL(f16_unpack_kernel, dim3(grid_for((long long)N * ((C + 7) >> 3) * plane)),
                     dim3(kThreads), 0, (hipStream_t)stream, static_cast<const uint4*>(xb), x, N, C,
                     plane);
  return (int)hipGetLastError();
}

size_t ssad_f16_filter_halves(int M, int C) {
  const size_t f = (size_t)9 * (size_t)((C + 7) & ~7) * (size_t)M;
  const size_t d = (size_t)9 * (size_t)((M + 7) & ~7) * (size_t)C;
  return f > d ? f : d;
}

int ssad_f16_pack_filter(const float* w, int M, int C, void* wf, void* wd, ssad_stream_t stream) {
  if (!w || M < 1 || C < 1 || (!wf && !wd)) return SSAD_E_BADARG;
  const int nf = 9 * ((C + 7) >> 3) * M, nd = 9 * ((M + 7) >> 3) * C;
  const int n = nf > nd ? nf : nd;
  hipLaunchKernelGGL(f16_pack_filter_kernel, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0,
                     (hipStream_t)stream, w, M, C, static_cast<uint4*>(wf), static_cast<uint4*>(wd));
  return (int)hipGetLastError();
}

int ssad_conv3x3_forward_f16(const void* xb, const void* wp, const float* bias, int N, int C, int H,
                             int W, int M, int flags, void* y, ssad_stream_t stream) {
  if (!xb || !wp || !y || N < 0 || M < 1 || H < 1 || W < 1) return SSAD_E_BADARG;
  if (C < 1) return SSAD_E_BADARG;
  const int nchw = (flags & SSAD_F16_OUT_NCHW_F32) != 0;
  if (!nchw && (M & 7)) return SSAD_E_BADARG;                   // blocked output: whole 8-blocks
  if (N == 0) return 0;
  F16Conv p;
  p.x = static_cast<const uint4*>(xb);
  p.w = static_cast<const uint4*>(wp);
  p.bias = bias;
  p.y = y;
  p.N = N; p.C = C; p.H = H; p.W = W; p.M = M;
  p.tiles_x = (W + TS - 1) / TS;
  p.tiles_y = (H + TS - 1) / TS;
  p.relu = (flags & SSAD_CONV_RELU) != 0;
  p.out_nchw_f32 = nchw;
  const long long tiles = (long long)N * p.tiles_x * p.tiles_y;
  if (tiles >= (1LL << 31)) return SSAD_E_BADARG;
  hipLaunchKernelGGL(conv3x3_f16_kernel, dim3((unsigned)tiles, (unsigned)((M + MT - 1) / MT)),
                     dim3(kThreads), 0, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}

}  // extern "C"
