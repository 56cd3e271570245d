// stem.hip -- the ResNet stem, 7x7 / stride 2 / pad 3 on 3 input channels, 64 outputs, fp32 on the matrix cores
// (Detectron: ResNet.py:85-130 `conv1`; caffe2/operators/conv_op_impl.h:126-173 computes it as im2col + GEMM).
//
// The general implicit GEMM (gemm_conv.hip, IMPLICIT = true) gathers the im2col view 4 bytes per lane straight from
// L2: 147 rows x 2.3 M columns = 1.35 GB of L2 -> LDS traffic for a 110 MB batch of images, 1.2 ms per stem.  Here a
// workgroup stages the RAW input patch of an 8 x 32 output tile once (3 x 21 x 69 floats, LDS-DMA) and every tap is
// a `ds_read_b32 base+imm` of it:
//   K order: (c, ky) major, the 7 kx taps padded to 4 PAIRS (kx, kx + 1); the two k rows of a
//            v_mfma_f32_32x32x2_f32 step are one pair, so lane half h reads column +h: one lane-constant base, every
//            (c, ky, pair) displacement an immediate -- no vector ALU instruction in the K loop (VALU issue is time the
//            fp32 MFMAs of a SIMD do not get, tools/coissue_probe.hip); the 8th tap's filter value is 0;
//   A:       the filter re-ordered to [84 steps][2][64] in LDS once per (persistent) workgroup;
//   wave:    64 outputs x 2 rows of 32 pixels = 4 accumulator tiles, 4 MFMAs per 4 LDS reads.
// Output without bias / activation: the pool pass that follows adds them (elementwise.hip: pool3x3s2_bias_relu).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_internal.h"
#include "ssad_kernels.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TR = 8, TC = 32;                 // output tile
constexpr int PRW = 2 * TR + 5, PCL = 2 * TC + 5;   // raw patch 21 x 69
constexpr int PCH = PRW * PCL;                 // floats per channel (1449)
constexpr int PATCH = 3 * PCH;                 // 4347
constexpr int PLOADS = (PATCH + 255) / 256;    // 17 wave-rounds of 256 lanes
constexpr int KSTEPS = 3 * 7 * 4;              // 84
constexpr int WL = KSTEPS * 2 * 64;            // filter floats in LDS
constexpr unsigned kOob = 0x80000000u;

struct StemArgs {
  const float* x;      // [N][3][H][W]
  const float* wt;     // [147][lda]: row c*49 + ky*7 + kx, 64 outputs
  float* y;            // [N][64][OH][OW]
  int N, H, W, OH, OW, lda;
  int tiles_x, tiles_y, tiles;
};

__global__ __launch_bounds__(256, 2) void stem7x7s2_kernel(const StemArgs a) {
  constexpr int PBUF = PLOADS * 256 + 8;
  __shared__ float patch[2 * PBUF];            // two buffers; linear [c][r][q], pitch 69 (the pad tap reads one float on)
  __shared__ float wl[WL];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;

  // ---- filter: wl[(ks * 2 + hh) * 64 + m], ks = (c * 7 + ky) * 4 + pair, kx = 2 pair + hh (kx = 7: zero) ----
  for (int e = tid; e < WL; e += 256) {
    const int m = e & 63, hh = (e >> 6) & 1, ks = e >> 7;
    const int pair = ks & 3, cky = ks >> 2;
    const int kx = 2 * pair + hh;
    wl[e] = kx < 7 ? a.wt[(long long)(cky * 7 + kx) * a.lda + m] : 0.0f;
  }
  if (tid < 16) patch[(tid >> 3) * PBUF + PLOADS * 256 + (tid & 7)] = 0.0f;

  // ---- patch staging map: element e = i * 256 + tid -> (c, r, q); fixed for the kernel ----
  int rq[PLOADS];                                // r << 8 | q, or -1 past the patch
  unsigned rel[PLOADS];                          // (c * H + r) * W + q, in bytes
#pragma unroll
  for (int i = 0; i < PLOADS; ++i) {
    const int e = i * 256 + tid;
    const int c = e / PCH, rem = e - c * PCH;
    const int r = rem / PCL, q = rem - r * PCL;
    rq[i] = e < PATCH ? (r << 8 | q) : -1;
    rel[i] = (unsigned)(((c * a.H + r) * a.W + q) * 4);
  }
  const unsigned patch_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)patch;
  const long long img = (long long)3 * a.H * a.W;
  const ssad_dev::rsrc_words xrs = ssad_dev::uniform_rsrc_words(a.x, (unsigned)((long long)a.N * img * 4));

  // ---- operand bases ----
  // B: pixel (row 2 wave + t, column j) of the tile, tap column +h
  const float* bb = patch + (2 * (2 * wave)) * PCL + 2 * j + h;      // + t * 2 * PCL + c * PCH + ky * PCL + 2 * pair
  const float* ab = wl + h * 64 + j;                                 // + ks * 128 + mt * 32

  // the next tile's patch is fetched (LDS-DMA, no registers) while this one is multiplied
  auto stage = [&](int t, int buf) {
    const int per = a.tiles_x * a.tiles_y;
    const int n = t / per, rem = t - n * per;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int iy0 = 2 * ty * TR - 3, ix0 = 2 * tx * TC - 3;
    // interior tile: the patch origin goes into the scalar offset, the lane offsets are the kernel-constant map;
    // border tile: the (possibly negative) origin is added per lane and out-of-image lanes get an out-of-range
    // offset (zero fill) -- a negative SCALAR offset would not wrap, the address sum is 64 bits wide
    const bool inner = iy0 >= 0 && ix0 >= 0 && iy0 + PRW <= a.H && ix0 + PCL <= a.W;
    const int origin = (iy0 * a.W + ix0) * 4;
    const int soff = __builtin_amdgcn_readfirstlane((int)((long long)n * img * 4) + (inner ? origin : 0));
#pragma unroll
    for (int i = 0; i < PLOADS; ++i) {
      unsigned vo = rel[i];
      if (!inner) {
        const int r = rq[i] >> 8, q = rq[i] & 255;
        const bool ok = rq[i] >= 0 && (unsigned)(iy0 + r) < (unsigned)a.H && (unsigned)(ix0 + q) < (unsigned)a.W;
        vo = ok ? (unsigned)((int)rel[i] + origin) : kOob;
      } else if (i == PLOADS - 1) {
        vo = rq[i] >= 0 ? rel[i] : kOob;
      }
      ssad_dev::lds_dma<4>(xrs, patch_lds + (unsigned)((buf * PBUF + i * 256 + wave * 64) * 4), vo, soff);
    }
  };

  int buf = 0;
  if ((int)blockIdx.x < a.tiles) stage((int)blockIdx.x, 0);
  for (int t = (int)blockIdx.x; t < a.tiles; t += (int)gridDim.x, buf ^= 1) {
    const int per = a.tiles_x * a.tiles_y;
    const int n = t / per, rem = t - n * per;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int oy0 = ty * TR, ox0 = tx * TC;
    // this tile's patch has landed (and the previous tile's stores have left); after the barrier everybody's part
    // has, everybody is done reading the other buffer, and (first time) the filter table is written
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + (int)gridDim.x < a.tiles) stage(t + (int)gridDim.x, buf ^ 1);
    const float* bbt = bb + buf * PBUF;

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
#pragma unroll
    for (int cky = 0; cky < 21; ++cky) {
      const int c = cky / 7, ky = cky - c * 7;
#pragma unroll
      for (int pair = 0; pair < 4; ++pair) {
        const int ks = cky * 4 + pair;
        float av[2], bv[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) av[mt] = ab[ks * 128 + mt * 32];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bv[nt] = bbt[nt * 2 * PCL + c * PCH + ky * PCL + 2 * pair];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
      }
    }

    // ---- store: register r of tile (mt, nt) = output m = mt * 32 + (r & 3) + 8 (r >> 2) + 4 h, pixel column j ----
    const long long plane = (long long)a.OH * a.OW;
    float* yn = a.y + (long long)n * 64 * plane;
    const int ox = ox0 + j;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int oy = oy0 + 2 * wave + nt;
      if (oy < a.OH && ox < a.OW) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            yn[m * plane + (long long)oy * a.OW + ox] = acc[mt][nt][r];
          }
      }
    }
  }
}

}  // namespace

// ssad_conv_implicit_gemm's fast path for the stem geometry (gemm_conv.hip): x [N][3][H][W], wt the transposed filter
// [147][lda] with M = 64, y [N][64][OH][OW], no epilogue terms
int ssad_stem7x7s2_launch(const float* x, const float* wt, int lda, int N, int H, int W, float* y, hipStream_t stream) {
  StemArgs a;
  a.x = x; a.wt = wt; a.y = y; a.N = N; a.H = H; a.W = W; a.lda = lda;
  a.OH = (H + 6 - 7) / 2 + 1; a.OW = (W + 6 - 7) / 2 + 1;
  a.tiles_x = (a.OW + TC - 1) / TC; a.tiles_y = (a.OH + TR - 1) / TR;
  const long long tiles = (long long)N * a.tiles_x * a.tiles_y;
  if (tiles == 0) return 0;
  if (tiles >= (1LL << 31)) return SSAD_E_BADARG;
  a.tiles = (int)tiles;
  const int cus = ssad_cu_count();
  const int grid = (int)(tiles < 2LL * cus ? tiles : 2LL * cus);
  hipLaunchKernelGGL(stem7x7s2_kernel, dim3(grid), dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}
