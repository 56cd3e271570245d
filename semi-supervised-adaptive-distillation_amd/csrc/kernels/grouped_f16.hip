// grouped_f16.hip -- ResNeXt's grouped 3x3 convolution (cardinality 64: detectron/lib/modeling/
// ResNet.py:247-258, conv_pool_op_base.h:45-194 `group`, conv_op_impl.h:43-47) with fp16 storage
// and fp32 accumulation, channel-blocked activations (conv3x3_f16.hip): the X-101-64x4d teacher of
// BASELINE config 5 in that config's precision.  Forward only (the teacher is frozen), stride 1
// (the three stride-2 layers run at stride 1 and keep the even positions: ssad_f16_elementwise).
//
// A group is 4 / 8 / 16 / 32 channels wide (res2..res5): far too narrow for a 32-row MFMA tile,
// and the layer moves 4 bytes per 2 * 9 * cg flops -- HBM-bound at every width.  So the kernel
// is a streaming pass whose arithmetic rides along on v_mfma_f32_16x16x32_f16:
//   * unit of work = 16 output channels ("super-group": 4 / 2 / 1 groups, or half a 32-wide
//     group) x a row segment of 16 pixels; K = 32 per instruction =
//       - cg <= 16: 2 filter taps x the super-group's own 16 input channels; the filter operand is
//         the block-diagonal [16 x 16] matrix of the groups inside (zeros across groups), packed
//         once into lane order -- 5 instructions per row segment (the 10th tap slot is zero);
//       - cg = 32: 1 tap x the group's 32 input channels -- 9 instructions;
//     a lane's B operand is ONE 16-byte LDS read: 8 channels of one pixel, shifted by the tap;
//   * workgroup = 4 waves = 64 channels (8 blocks) x an 8 x 16 pixel tile; its 10 x 18 halo of all
//     8 blocks lands in LDS by LDS-DMA (out-of-image lanes at an out-of-range offset = zero
//     padding); 23 KiB per workgroup, six resident per CU keep ~140 KiB in flight;
//   * the packed filter of a wave's super-group stays in registers (20 / 36 VGPRs);
//   * epilogue: bias (the folded AffineChannel) + ReLU, 8-byte stores.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_internal.h"
#include "ssad_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
using ssad_dev::lds_dma;
using ssad_dev::rsrc_words;
using ssad_dev::uniform_rsrc;
using ssad_dev::uniform_rsrc_words;

constexpr int kThreads = 256;
constexpr int TR = 8, TC = 16;               // output tile: rows x columns
constexpr int HR = TR + 2, HC = TC + 2;      // halo tile
constexpr int NB = 8;                        // channel blocks per workgroup (64 channels)
constexpr int SLOTS = NB * HR * HC;          // 1440
constexpr int PIECES = (SLOTS + 63) / 64;    // 23
constexpr unsigned kOob = 0x80000000u;

struct GroupedF16 {
  const uint4* x;        // blocked fp16 [N][C/8][H][W]
  const uint4* w;        // packed [C/16][MF][64] x 16 B
  const float* bias;     // [C] or null
  uint4* y;              // blocked fp16 [N][C/8][H][W]
  int N, C, H, W;
  int tiles_x, tiles_y, slabs;
  int relu;
};

__device__ __forceinline__ half8 as_half8(const uint4& v) { return __builtin_bit_cast(half8, v); }

// WIDE = false: groups of <= 16 channels (MF = 5 tap pairs); true: groups of 32 (MF = 9 taps)
template <bool WIDE>
__global__ __launch_bounds__(kThreads) void grouped_f16_kernel(const GroupedF16 p) {
  constexpr int MF = WIDE ? 9 : 5;
  __shared__ uint4 lds[PIECES * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, g4 = lane >> 4;
  int t = blockIdx.x;
  const int slab = t % p.slabs; t /= p.slabs;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int n = t / p.tiles_y;
  const int y0 = ty * TR, x0 = tx * TC;
  const int CB = p.C >> 3;
  const int plane = p.H * p.W;
  const rsrc_words xrs = uniform_rsrc_words(p.x, (unsigned)((long long)p.N * CB * plane * 16));
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) uint4*)lds);

  // ---- halo of the slab's 8 blocks: piece k = wave + 4 i covers slots [64 k, 64 k + 64)
#pragma unroll
  for (int i = 0; i < (PIECES + 3) / 4; ++i) {
    const int k = wave + 4 * i;
    if (k < PIECES) {
      const int s = 64 * k + lane;
      const int blk = s / (HR * HC), r = s % (HR * HC);
      const int gy = y0 - 1 + r / HC, gx = x0 - 1 + r % HC;
      const bool ok = s < SLOTS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      const unsigned vo = ok ? (unsigned)((((long long)n * CB + slab * NB + blk) * plane + gy * p.W + gx) * 16) : kOob;
      lds_dma<16>(xrs, lds0 + (unsigned)(64 * k) * 16u, vo, 0);
    }
  }
  // ---- this wave's filter operand: super-group sg = slab * 4 + wave
  const int sg = slab * 4 + wave;
  half8 a[MF];
  {
    const uint4* wp = p.w + ((long long)sg * MF) * 64 + lane;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) a[mf] = as_half8(wp[mf * 64]);
  }
  // B operand slot of lane (pixel i16, k-group g4) for instruction mf, row segment r:
  //   narrow: tap = 2 mf + (g4 >> 1), block = 2 wave + (g4 & 1);  wide: tap = mf, block = 4 (wave / 2) + g4
  int boff[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    int tap = WIDE ? mf : 2 * mf + (g4 >> 1);
    if (tap > 8) tap = 8;                                  // the zero tap slot: any valid address
    const int blk = WIDE ? 4 * (wave >> 1) + g4 : 2 * wave + (g4 & 1);
    boff[mf] = (blk * HR + tap / 3) * HC + i16 + tap % 3;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue constants: lane holds channels 4 g4 .. 4 g4 + 3 of its super-group for pixel i16
  const int oc = sg * 16 + 4 * g4;
  float4v bq = {0.0f, 0.0f, 0.0f, 0.0f};
  if (p.bias) bq = *reinterpret_cast<const float4v*>(p.bias + oc);
  const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y, (unsigned)((long long)p.N * CB * plane * 16));
  const int gx = x0 + i16;
  const int ocb = (oc >> 3);                               // output block of this lane
#pragma unroll
  for (int r = 0; r < TR; ++r) {
    float4v acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const half8 b = as_half8(lds[boff[mf] + r * HC]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mf], b, acc, 0, 0, 0);
    }
    acc += bq;
    half4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = acc[e];
      if (p.relu) v = fmaxf(v, 0.0f);
      o[e] = (_Float16)v;
    }
    const int gy = y0 + r;
    const unsigned vo = (gy < p.H && gx < p.W)
                            ? (unsigned)((((long long)n * CB + ocb) * plane + gy * p.W + gx) * 16 + 8 * (g4 & 1))
                            : kOob;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uint2v, o), yrs, vo, 0, 0);
  }
}

// w [C][cg][3][3] fp32 -> lane-order operands [C/16][MF][64][8] fp16 (see the kernel's header)
__global__ __launch_bounds__(kThreads) void grouped_f16_pack_kernel(const float* __restrict__ w, int C, int cg,
                                                                    uint4* __restrict__ out) {
  const bool wide = cg > 16;
  const int MF = wide ? 9 : 5;
  const int total = (C / 16) * MF * 64;
  const int idx = blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63, mf = (idx >> 6) % MF, sg = (idx >> 6) / MF;
  const int i16 = lane & 15, g4 = lane >> 4;
  const int m = sg * 16 + i16;
  const int grp = m / cg;
  half8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = 0.0f;
    if (wide) {
      const int ci = 8 * g4 + e;                           // channel inside the group (0..31)
      v = w[((long long)m * cg + ci) * 9 + mf];
    } else {
      const int tap = 2 * mf + (g4 >> 1);
      const int c = sg * 16 + (g4 & 1) * 8 + e;            // absolute input channel
      if (tap < 9 && c / cg == grp) v = w[((long long)m * cg + (c - grp * cg)) * 9 + tap];
    }
    o[e] = (_Float16)v;
  }
  out[idx] = __builtin_bit_cast(uint4, o);
}

}  // namespace

extern "C" {

size_t ssad_grouped_conv3x3_f16_filter_halves(int C, int group) {
  if (C < 1 || group < 1 || C % group || (C & 15)) return 0;
  const int cg = C / group;
  return (size_t)(C / 16) * (cg > 16 ? 9 : 5) * 64 * 8;
}

int ssad_grouped_conv3x3_f16_pack_filter(const float* w, int C, int group, void* packed, ssad_stream_t stream) {
  if (!w || !packed || C < 16 || (C & 63) || group < 1 || C % group) return SSAD_E_BADARG;
  const int cg = C / group;
  if (cg != 4 && cg != 8 && cg != 16 && cg != 32) return SSAD_E_BADARG;
  const int total = (C / 16) * (cg > 16 ? 9 : 5) * 64;
  hipLaunchKernelGGL(grouped_f16_pack_kernel, dim3((total + kThreads - 1) / kThreads), dim3(kThreads), 0,
                     (hipStream_t)stream, w, C, cg, static_cast<uint4*>(packed));
  return (int)hipGetLastError();
}

int ssad_grouped_conv3x3_f16(const void* x_blocked, const void* packed, const float* bias, int N, int C, int H,
                             int W, int group, int relu, void* y_blocked, ssad_stream_t stream) {
  if (!x_blocked || !packed || !y_blocked || N < 0 || C < 64 || (C & 63) || H < 1 || W < 1 || group < 1 || C % group)
    return SSAD_E_BADARG;
  const int cg = C / group;
  if (cg != 4 && cg != 8 && cg != 16 && cg != 32) return SSAD_E_BADARG;
  if (N == 0) return 0;
  if ((long long)N * (C >> 3) * H * W * 16 >= (1LL << 31)) return SSAD_E_BADARG;
  GroupedF16 p;
  p.x = static_cast<const uint4*>(x_blocked);
  p.w = static_cast<const uint4*>(packed);
  p.bias = bias;
  p.y = static_cast<uint4*>(y_blocked);
  p.N = N; p.C = C; p.H = H; p.W = W;
  p.tiles_x = (W + TC - 1) / TC; p.tiles_y = (H + TR - 1) / TR; p.slabs = C / 64;
  p.relu = relu;
  const long long wgs = (long long)N * p.tiles_x * p.tiles_y * p.slabs;
  if (wgs >= (1LL << 31)) return SSAD_E_BADARG;
  if (cg > 16) hipLaunchKernelGGL(grouped_f16_kernel<true>, dim3((unsigned)wgs), dim3(kThreads), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(grouped_f16_kernel<false>, dim3((unsigned)wgs), dim3(kThreads), 0, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}

}  // extern "C"
