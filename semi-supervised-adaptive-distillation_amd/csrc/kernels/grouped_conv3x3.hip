// grouped_conv3x3.hip -- grouped 3x3 convolution (pad 1, stride 1 or 2), NCHW fp32, forward:
// ResNeXt's cardinality-64 bottleneck layer (detectron/lib/modeling/ResNet.py:247-258 with
// RESNETS.NUM_GROUPS = 64, WIDTH_PER_GROUP = 4, STRIDE_1X1 = False, i.e. BASELINE config 5's
// X-101-64x4d teacher; reference algorithm caffe2/operators/conv_op_impl.h:93-98,126-173: the groups
// are independent convolutions over contiguous channel blocks).
//
//   y[n][g*cg + m][oy][ox] = act( bias + sum_{c < cg} sum_{ky,kx} w[g*cg + m][c][ky][kx]
//                                                       * x[n][g*cg + c][oy*s + ky - 1][ox*s + kx - 1] )
//
// cg = channels per group = 4, 8, 16, 32 at res2..res5.  Design for gfx950: per group the layer is a
// [cg x 9cg] . [9cg x pixels] product -- M = 16 of `v_mfma_f32_16x16x4_f32` is exactly a res4 group (23
// of the 33 layers).  A wave owns one group and rows of 16 output pixels:
//  * the filter is packed once (ssad_grouped_conv3x3_pack_filter; the teacher is frozen) into MFMA
//    operand order [group][row tile][step][lane], so a wave fetches its 9cg/4 A operands with
//    coalesced loads and keeps them in REGISTERS for all its rows.  Reduction order k = tap * cg + c:
//    four consecutive k share their tap, so in the fully unrolled loop the tap is a compile-time
//    constant and the channel is (constant + lane / 16);
//  * the input tile of the workgroup's channels ((rows*s + 2) x (16*s + 2) per channel, zero outside
//    the image) is staged once in LDS; a B operand is then ONE ds_read_b32 with an immediate offset
//    ((c0 * plane + ky * pitch + kx) * 4) on a per-lane base -- no address arithmetic in the loop;
//  * bias and ReLU in registers.  One workgroup = 4 waves = 8 output rows x 16 columns of GPW
//    neighbouring groups (4 at cg = 4, 2 at cg = 8, else 1): narrow groups would otherwise give a
//    workgroup too little to do (measured 0.77 -> see DESIGN.md 3.6 for res2's layer).
// Groups narrower than 16 channels leave MFMA rows idle (res2: cg = 4, res3: cg = 8 -- seven small
// layers); cg = 32 uses two row tiles.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssad_kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256;
constexpr int TR = 8, TC = 16;             // output tile: rows x columns

struct GArgs {
  const float* x;
  const float* wp;          // packed filter
  const float* bias;
  float* y;
  int N, C, H, W, OH, OW, G, relu;
  int tiles_x, tiles_y, gblocks;
};

template <int CG>
struct Geo {
  static constexpr int MT = CG > 16 ? CG / 16 : 1;          // 16-row MFMA tiles per group
  static constexpr int KS = 9 * CG / 4;                     // MFMA steps (K = 9 cg, 4 per step)
  static constexpr int GPW = CG == 4 ? 4 : (CG == 8 ? 2 : 1);   // groups per workgroup
};

// packed[g][t][s][lane] = w[g*CG + t*16 + (lane & 15)][(4s % CG) + lane / 16][tap = 4s / CG] (0 past CG rows)
template <int CG>
__global__ __launch_bounds__(256) void grouped_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int G) {
  constexpr int MT = Geo<CG>::MT, KS = Geo<CG>::KS;
  const long long total = (long long)G * MT * KS * 64;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(e & 63);
    long long r = e >> 6;
    const int s = (int)(r % KS); r /= KS;
    const int t = (int)(r % MT);
    const int g = (int)(r / MT);
    const int m = t * 16 + (lane & 15), k0 = 4 * s, tap = k0 / CG, c = (k0 % CG) + (lane >> 4);
    out[e] = (m < CG) ? w[(((long long)g * CG + m) * CG + c) * 9 + tap] : 0.0f;
  }
}

template <int CG, int S>
__global__ __launch_bounds__(kThreads) void grouped_conv3x3_kernel(const GArgs a) {
  constexpr int MT = Geo<CG>::MT, KS = Geo<CG>::KS, GPW = Geo<CG>::GPW;
  constexpr int CH = CG * GPW;                       // channels staged per workgroup
  constexpr int IR = (TR - 1) * S + 3, IC = (TC - 1) * S + 3;     // input tile rows / columns
  constexpr int CP = IC + 1;                         // row pitch
  constexpr int PLANE = IR * CP;
  constexpr int ROWS = TR * GPW / 4;                 // output rows per wave
  __shared__ float tile[CH * PLANE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kk = lane >> 4;
  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y; b /= a.tiles_y;
  const int gb = b % a.gblocks, n = b / a.gblocks;
  const int gi = wave % GPW, g = gb * GPW + gi;
  const int oy0 = ty * TR, ox0 = tx * TC;
  const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;

  // ---- the wave's filter: 9cg/4 (x MT) coalesced loads, kept in registers ----
  float wr[MT][KS];
  {
    const float* wp = a.wp + (long long)(g < a.G ? g : 0) * MT * KS * 64 + lane;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int s = 0; s < KS; ++s) wr[t][s] = wp[(t * KS + s) * 64];
  }

  // ---- stage the workgroup's input channels (zero outside the image / past the last group) ----
  const int ch0 = gb * CH;
  const float* xg = a.x + ((long long)n * a.C + ch0) * a.H * a.W;
  for (int e = tid; e < CH * IR * IC; e += kThreads) {
    const int c = e / (IR * IC), r = (e / IC) % IR, q = e % IC;
    const int iy = iy0 + r, ix = ix0 + q;
    float v = 0.0f;
    if (ch0 + c < a.C && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = xg[((long long)c * a.H + iy) * a.W + ix];
    tile[c * PLANE + r * CP + q] = v;
  }
  __syncthreads();
  if (g >= a.G) return;

  // ---- rows of 16 output pixels ----
#pragma unroll 1
  for (int rr = 0; rr < ROWS; ++rr) {
    const int r = wave / GPW + (4 / GPW) * rr;
    const float* base = tile + (gi * CG + kk) * PLANE + (r * S) * CP + j * S;
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k0 = 4 * s, tap = k0 / CG, c0 = k0 % CG;
      const float bv = base[c0 * PLANE + (tap / 3) * CP + (tap % 3)];
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t][s], bv, acc[t], 0, 0, 0);
    }
    const int oy = oy0 + r, ox = ox0 + j;
    if (oy < a.OH && ox < a.OW) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = t * 16 + kk * 4 + q;          // C/D layout of 16x16x4: row = 4 * (lane / 16) + q
          if (m < CG) {
            const int ch = g * CG + m;
            float v = acc[t][q] + (a.bias ? a.bias[ch] : 0.0f);
            if (a.relu) v = fmaxf(v, 0.0f);
            a.y[(((long long)n * a.C + ch) * a.OH + oy) * a.OW + ox] = v;
          }
        }
    }
  }
}

template <int CG>
int launch(GArgs a, int stride, hipStream_t s) {
  a.gblocks = (a.G + Geo<CG>::GPW - 1) / Geo<CG>::GPW;
  const long long blocks = (long long)a.N * a.gblocks * a.tiles_y * a.tiles_x;
  if (blocks >= (1LL << 31)) return SSAD_E_BADARG;
  const dim3 grid((unsigned)blocks);
  if (stride == 1) hipLaunchKernelGGL((grouped_conv3x3_kernel<CG, 1>), grid, dim3(kThreads), 0, s, a);
  else hipLaunchKernelGGL((grouped_conv3x3_kernel<CG, 2>), grid, dim3(kThreads), 0, s, a);
  return (int)hipGetLastError();
}

template <int CG>
int pack(const float* w, float* out, int G, hipStream_t s) {
  const long long total = (long long)G * Geo<CG>::MT * Geo<CG>::KS * 64;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL((grouped_pack_kernel<CG>), dim3(blocks), dim3(256), 0, s, w, out, G);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

long long ssad_grouped_conv3x3_filter_floats(int C, int group) {
  if (C < 1 || group < 1 || C % group) return -1;
  const int cg = C / group;
  if (cg != 4 && cg != 8 && cg != 16 && cg != 32) return -1;
  return (long long)group * (cg > 16 ? cg / 16 : 1) * (9 * cg / 4) * 64;
}

int ssad_grouped_conv3x3_pack_filter(const float* w, int C, int group, float* packed, ssad_stream_t stream) {
  if (!w || !packed || ssad_grouped_conv3x3_filter_floats(C, group) < 0) return SSAD_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  switch (C / group) {
    case 4: return pack<4>(w, packed, group, s);
    case 8: return pack<8>(w, packed, group, s);
    case 16: return pack<16>(w, packed, group, s);
    default: return pack<32>(w, packed, group, s);
  }
}

int ssad_grouped_conv3x3_forward(const float* x, const float* packed, const float* bias, int N, int C, int H,
                                 int W, int group, int stride, int relu, float* y, ssad_stream_t stream) {
  if (!x || !packed || !y || N < 0 || H < 1 || W < 1) return SSAD_E_BADARG;
  if (ssad_grouped_conv3x3_filter_floats(C, group) < 0) return SSAD_E_BADARG;   // widths other than 4/8/16/32
  if (stride != 1 && stride != 2) return SSAD_E_BADARG;
  GArgs a;
  a.x = x; a.wp = packed; a.bias = bias; a.y = y;
  a.N = N; a.C = C; a.H = H; a.W = W; a.G = group; a.relu = relu;
  a.OH = (H - 1) / stride + 1; a.OW = (W - 1) / stride + 1;      // pad 1, kernel 3
  a.tiles_y = (a.OH + TR - 1) / TR; a.tiles_x = (a.OW + TC - 1) / TC;
  a.gblocks = 0;
  if (N == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  switch (C / group) {
    case 4: return launch<4>(a, stride, s);
    case 8: return launch<8>(a, stride, s);
    case 16: return launch<16>(a, stride, s);
    default: return launch<32>(a, stride, s);
  }
}

}  // extern "C"
