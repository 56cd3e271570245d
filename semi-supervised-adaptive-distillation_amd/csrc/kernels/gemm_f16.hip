// gemm_f16.hip -- pointwise (1x1) convolution with fp16 storage and fp32 accumulation on the
// gfx950 matrix cores (v_mfma_f32_32x32x16_f16), channel-blocked activations: the bottleneck
// 1x1 layers, projection shortcuts and FPN laterals of the backbones in BASELINE config 5's
// precision.  Reference: CudnnConvOp<float16> with fp32 math for EVERY convolution of the net
// (caffe2/operators/conv_op_cudnn.cc:631-636; detectron/lib/modeling/ResNet.py:221-283,
// FPN.py:116-250); the algorithm for a 1x1 kernel is conv_op_impl.h:126-173 (Y[n] = W . X[n]).
// Contract as in conv3x3_f16.hip: fp16 operands, every product and sum in fp32, ONE rounding
// when a result is stored.
//
// Layout (conv3x3_f16.hip): Xb[n][c/8][y][x][c%8], 16 bytes = the 8 channels of one pixel = one
// MFMA B operand (a lane supplies 8 consecutive K = channels); filter packed Wp[c/8][m][c%8] =
// the A operand.  In this layout the layer is HBM-bound at almost every shape of the network
// (res2..res4: 50-200 flop/B against ~400 for the matrix cores), so the kernel is built around the
// stream, not the MFMAs:
//   * output pixels of the whole batch are ONE flat index q = (n, y, x); a workgroup owns PT (256
//     or 128) consecutive q x 128 output channels; wave (wo, wp) = 64 channels x PT/2 pixels;
//   * K runs over chunks of 32 input channels.  Both operands of a chunk -- the pixel tile
//     [4 blocks][PT] and the filter tile [4 blocks][128 rows] -- go HBM/L2 -> LDS by LDS-DMA
//     (buffer_load_dwordx4 ... lds: no staging registers, no ds_write) through an S-stage ring,
//     S - 1 chunks in flight across one raw s_barrier + counted vmcnt per chunk; every wave
//     fetches the same 64 pixels (rows) of every block, so a lane's offset is computed once
//     and the block is the scalar offset.  A stride-2 layer (first block of res3..res5) reads
//     its input at (2y, 2x) in the loader: no subsampled copy.
//   * both operands are read back with ds_read_b128 over 32 consecutive 16-byte slots per
//     half-wave (conflict free);
//   * epilogue in registers: bias, shortcut Sum (same size, or the 2x nearest-upsampled coarser
//     FPN level: FPN.py:283-306 without a pass of its own), ReLU, or -- data gradient -- the
//     ReluGradient mask of the layer below; 8-byte stores, a wave covers whole 16-byte slots;
//   * workgroup ids: the channel blocks of one pixel tile are 8 apart, i.e. on ONE XCD, so the
//     pixel tile leaves HBM once and the other channel blocks find it in that XCD's L2.
// The data gradient dX = W^T dY is the same kernel on the transposed pack Wd[m/8][c][m%8].
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "conv_internal.h"
#include "ssad_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
using ssad_dev::lds_dma;
using ssad_dev::rsrc_words;
using ssad_dev::uniform_rsrc;
using ssad_dev::uniform_rsrc_words;

constexpr int kThreads = 256;
constexpr int MT = 128;                    // output channels per workgroup
// (PW_CBC / PW_S256 / PW_S128: compile-time A/B of the K chunk and ring depth, tools/dbg/r5_pw_f16_cbc.sh; 64-channel
// chunks measured 0.974 against 0.718 ms over config 5's shapes: the LDS they take costs a resident workgroup)
#ifndef PW_CBC
#define PW_CBC 4
#endif
#ifndef PW_S256
#define PW_S256 3
#endif
#ifndef PW_S128
#define PW_S128 4
#endif
constexpr int CBC = PW_CBC;                // 8-channel blocks per K chunk (32 channels)
constexpr unsigned kOob = 0x80000000u;     // buffer offset past any descriptor: loads 0, stores dropped

struct PwF16 {
  const uint4* x;        // blocked fp16 input  [N][CB][Hi][Wi]
  const uint4* w;        // packed filter [CB][M] x 16 B
  const float* bias;     // [M] or null
  const uint4* res;      // blocked fp16 [N][MB][Hr][Wr] or null (Hr = Ho, or Ho / 2 with res_up)
  const uint4* mask;     // blocked fp16 like y or null: y = mask > 0 ? y : 0
  uint4* y;              // blocked fp16 [N][MB][Ho][Wo]
  int N, C, M;
  int Ho, Wo;            // output map
  int Hi, Wi, stride;    // input map; output (y, x) reads input (stride y, stride x)
  int relu, res_up;
  int mblocks;           // ceil(M / 128)
  int stages;            // LDS ring depth (2 .. Geo::S): min(chunks + 1, Geo::S) -- a short K takes less LDS,
                         // so more workgroups are resident to cover the fetch latency
  long long total;       // N * Ho * Wo
};

template <int PT>
struct Geo {
  static constexpr int WPX = PT / 2;                 // pixels per wave
  static constexpr int NT = WPX / 32;                // 32-pixel MFMA column tiles per wave
  static constexpr int B_SLOTS = CBC * PT;           // pixel-tile slots per stage
  static constexpr int A_SLOTS = CBC * MT;           // filter-tile slots per stage
  static constexpr int STAGE = B_SLOTS + A_SLOTS;    // 16-byte slots
  static constexpr int S = PT == 256 ? PW_S256 : PW_S128;        // ring depth: 72 KiB / 64 KiB per workgroup
  static constexpr int B_PIECES = B_SLOTS / 64 / 4;  // DMA instructions per wave and chunk
  static constexpr int A_PIECES = A_SLOTS / 64 / 4;
  static constexpr int OPS = B_PIECES + A_PIECES;
};

__device__ __forceinline__ half8 as_half8(const uint4& v) { return __builtin_bit_cast(half8, v); }

template <int PT, bool MASKED>
__global__ __launch_bounds__(kThreads, 2) void pw_f16_kernel(const PwF16 p) {
  using G = Geo<PT>;
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave & 1, wp = wave >> 1;
  const int j = lane & 31, h = lane >> 5;
  // workgroup -> (pixel tile, channel block): ids b, b + 8, ... share an XCD (and its L2)
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int mb = seq % p.mblocks;
  const long long t = (long long)(seq / p.mblocks) * 8 + xcd;
  const long long q0 = t * PT;
  if (q0 >= p.total) return;
  const int ocb = mb * MT;
  const int CB = (p.C + 7) >> 3, MB = p.M >> 3;
  const int plane_o = p.Ho * p.Wo, plane_i = p.Hi * p.Wi;

  const rsrc_words xrs = uniform_rsrc_words(p.x, (unsigned)((long long)p.N * CB * plane_i * 16));
  const rsrc_words wrs = uniform_rsrc_words(p.w, (unsigned)((long long)CB * p.M * 16));

  // ---- LDS-DMA lanes.  Pixel tile: slot (block, pixel); piece k = wave + 4 i covers block
  // k / (PT / 64), pixels (k % (PT / 64)) * 64 + lane: the same 64 pixels for every i.
  constexpr int PPB = PT / 64;                       // pieces per block row
  const int bpx = (wave % PPB) * 64 + lane;          // this lane's pixel of the tile
  const int bcb0 = wave / PPB;                       // first block this wave fetches (then + 4 / PPB per i)
  unsigned bvo;
  {
    const long long q = q0 + bpx;
    if (q < p.total) {
      const int n = (int)(q / plane_o), r = (int)(q % plane_o);
      const int oy = r / p.Wo, ox = r % p.Wo;
      bvo = (unsigned)((((long long)n * CB) * plane_i + (long long)(oy * p.stride) * p.Wi + ox * p.stride) * 16);
    } else {
      bvo = kOob;
    }
  }
  // filter tile: slot (block, row); piece k: block k / 2, rows (k % 2) * 64 + lane
  const int arow = (wave & 1) * 64 + lane;
  const int acb0 = wave >> 1;
  const unsigned avo = (ocb + arow < p.M) ? (unsigned)((ocb + arow) * 16) : kOob;
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) uint4*)lds);

  auto fetch = [&](int chunk, int stage) {
    const unsigned base = lds0 + (unsigned)(stage * G::STAGE) * 16u;
#pragma unroll
    for (int i = 0; i < G::B_PIECES; ++i) {
      const int cbl = bcb0 + i * (4 / PPB);
      const int cb = chunk * CBC + cbl;
      lds_dma<16>(xrs, base + (unsigned)(cbl * PT + (wave % PPB) * 64) * 16u, cb < CB ? bvo : kOob,
                  cb * plane_i * 16);
    }
#pragma unroll
    for (int i = 0; i < G::A_PIECES; ++i) {
      const int cbl = acb0 + 2 * i;
      const int cb = chunk * CBC + cbl;
      lds_dma<16>(wrs, base + (unsigned)(G::B_SLOTS + cbl * MT + (wave & 1) * 64) * 16u, cb < CB ? avo : kOob,
                  cb * p.M * 16);
    }
  };

  float16v acc[2][G::NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int tt = 0; tt < G::NT; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][tt][r] = 0.0f;

  const int nchunks = (CB + CBC - 1) / CBC;
  const int S = p.stages, D = S - 1;                       // ring depth, chunks in flight (wave-uniform)
  for (int c = 0; c < D; ++c)
    if (c < nchunks) fetch(c, c);
  int slot = 0;                                            // c % S
  for (int c = 0; c < nchunks; ++c) {
    // chunk c has landed when at most the operations of the chunks issued after it are outstanding
    // (vector memory retires in order).  Near the end fewer chunks are behind it: wait for all.
    if (c + D - 1 < nchunks && D == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G::OPS) : "memory");
    else if (c + D - 1 < nchunks && D == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::OPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // the stage of chunk c - 1 is free (every wave is past the barrier): refill it
    if (c + D < nchunks) fetch(c + D, slot == 0 ? S - 1 : slot - 1);      // (c + D) % S = (c - 1) % S
    const uint4* st = lds + slot * G::STAGE;
    slot = slot + 1 == S ? 0 : slot + 1;
#pragma unroll
    for (int ks = 0; ks < CBC / 2; ++ks) {
      half8 a[2], b[G::NT];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = as_half8(st[G::B_SLOTS + (2 * ks + h) * MT + wo * 64 + i * 32 + j]);
#pragma unroll
      for (int tt = 0; tt < G::NT; ++tt) b[tt] = as_half8(st[(2 * ks + h) * PT + wp * G::WPX + tt * 32 + j]);
#pragma unroll
      for (int tt = 0; tt < G::NT; ++tt) {
        acc[0][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[tt], acc[0][tt], 0, 0, 0);
        acc[1][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[tt], acc[1][tt], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: C/D row = (r & 3) + 8 (r >> 2) + 4 h, column = j (conv3x3_f16.hip)
  const int oc_w = ocb + wo * 64;
  unsigned pvo[G::NT], rvo[G::NT];
#pragma unroll
  for (int tt = 0; tt < G::NT; ++tt) {
    const long long q = q0 + wp * G::WPX + tt * 32 + j;
    if (q < p.total) {
      const int n = (int)(q / plane_o), r = (int)(q % plane_o);
      pvo[tt] = (unsigned)((((long long)n * MB) * plane_o + r) * 16 + 8 * h);
      if (p.res_up) {
        const int oy = r / p.Wo, ox = r % p.Wo;
        const int Wr = p.Wo >> 1, plane_r = (p.Ho >> 1) * Wr;
        rvo[tt] = (unsigned)((((long long)n * MB) * plane_r + (oy >> 1) * Wr + (ox >> 1)) * 16 + 8 * h);
      } else {
        rvo[tt] = pvo[tt];
      }
    } else {
      pvo[tt] = rvo[tt] = kOob;
    }
  }
  const unsigned ybytes = (unsigned)((long long)p.N * MB * plane_o * 16);
  const int plane_r16 = (p.res_up ? (p.Ho >> 1) * (p.Wo >> 1) : plane_o) * 16;
  const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y, ybytes);
  const __amdgpu_buffer_rsrc_t brs = uniform_rsrc(p.bias ? (const void*)p.bias : (const void*)p.y,
                                                  p.bias ? (unsigned)p.M * 4u : 0u);   // no bias: reads 0
  const __amdgpu_buffer_rsrc_t rrs = uniform_rsrc(p.res ? (const void*)p.res : (const void*)p.y,
                                                  p.res ? (unsigned)((long long)p.N * MB * plane_r16) : 0u);
  const __amdgpu_buffer_rsrc_t mrs = uniform_rsrc(MASKED ? (const void*)p.mask : (const void*)p.y,
                                                  MASKED ? ybytes : 0u);
  // A lane holds channels 4h .. 4h + 3 of each 8-channel block g for its pixel: half a 16-byte slot.
  // Blocks are handled in PAIRS (g_e, g_o): v_permlane32_swap exchanges the halves between lanes j
  // and j + 32, after which lane h = 0 holds the whole slot of g_e and lane h = 1 the whole slot of
  // g_o -- one 16-byte store (and one 16-byte residual / mask load, redistributed the same way) per
  // lane and pair instead of two 8-byte ones.
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto swap2 = [](unsigned a, unsigned b, unsigned& ra, unsigned& rb) {
    // rows of 32 lanes: ra = {a.row0, b.row0}, rb = {a.row1, b.row1}
    const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    ra = r.x; rb = r.y;
  };
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oc0 = oc_w + i * 32;
    if (oc0 >= p.M) continue;                              // wave-uniform
    float4 bq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
      bq[g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(brs, (unsigned)h * 16u,
                                                                            (oc0 + 8 * g) * 4, 0));
#pragma unroll
    for (int tt = 0; tt < G::NT; ++tt) {
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        if (oc0 + 16 * gp >= p.M) continue;                // wave-uniform
        const bool mine = oc0 + 8 * (2 * gp + h) < p.M;    // this lane's own slot (block 2 gp + h) exists
        const unsigned pbase = pvo[tt] == kOob ? kOob : pvo[tt] - 8u * h;      // slot address of the pixel
        const unsigned vo = (mine && pbase != kOob) ? pbase + (unsigned)h * (unsigned)(plane_o * 16) : kOob;
        const int so = ((oc0 >> 3) + 2 * gp) * plane_o * 16;
        unsigned re[2] = {0, 0}, ro[2] = {0, 0};           // residual halves for (g_e, 4h..), (g_o, 4h..)
        if (p.res) {                                       // wave-uniform
          const unsigned rbase = rvo[tt] == kOob ? kOob : rvo[tt] - 8u * h;
          const unsigned rv = (mine && rbase != kOob) ? rbase + (unsigned)h * (unsigned)plane_r16 : kOob;
          const u32x4 L = __builtin_amdgcn_raw_buffer_load_b128(rrs, rv, ((oc0 >> 3) + 2 * gp) * plane_r16, 0);
          swap2(L.x, L.z, re[0], ro[0]);
          swap2(L.y, L.w, re[1], ro[1]);
        }
        unsigned me[2] = {0, 0}, mo[2] = {0, 0};
        if (MASKED) {
          const u32x4 L = __builtin_amdgcn_raw_buffer_load_b128(mrs, vo, so, 0);
          swap2(L.x, L.z, me[0], mo[0]);
          swap2(L.y, L.w, me[1], mo[1]);
        }
        unsigned out[2][2];
#pragma unroll
        for (int eo = 0; eo < 2; ++eo) {
          const int g = 2 * gp + eo;
          float2v lo = float2v{acc[i][tt][4 * g], acc[i][tt][4 * g + 1]} + float2v{bq[g].x, bq[g].y};
          float2v hi = float2v{acc[i][tt][4 * g + 2], acc[i][tt][4 * g + 3]} + float2v{bq[g].z, bq[g].w};
          if (p.res) {
            const half4 rv = __builtin_bit_cast(half4, u32x2{eo ? ro[0] : re[0], eo ? ro[1] : re[1]});
            lo += float2v{(float)rv[0], (float)rv[1]};
            hi += float2v{(float)rv[2], (float)rv[3]};
          }
          half2v o01 = __builtin_convertvector(lo, half2v), o23 = __builtin_convertvector(hi, half2v);
          if (p.relu) {
            const half2v z = {(_Float16)0.0f, (_Float16)0.0f};
            o01 = __builtin_elementwise_max(o01, z);
            o23 = __builtin_elementwise_max(o23, z);
          }
          half4 o = {o01[0], o01[1], o23[0], o23[1]};
          if (MASKED) {                                    // relu_op.cu:44-53: dX = Y > 0 ? dY : 0
            const half4 m = __builtin_bit_cast(half4, u32x2{eo ? mo[0] : me[0], eo ? mo[1] : me[1]});
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = m[e] > (_Float16)0.0f ? o[e] : (_Float16)0.0f;
          }
          const u32x2 od = __builtin_bit_cast(u32x2, o);
          out[eo][0] = od.x; out[eo][1] = od.y;
        }
        u32x4 slot;
        {
          unsigned a0, b0, a1, b1;
          swap2(out[0][0], out[1][0], a0, b0);
          swap2(out[0][1], out[1][1], a1, b1);
          slot = u32x4{a0, a1, b0, b1};
        }
        __builtin_amdgcn_raw_buffer_store_b128(slot, yrs, vo, so, 0);
      }
    }
  }
}

// Filter [M][C] fp32 -> Wp[C/8][M][8] fp16 (forward) and Wd[M/8][C][8] (data gradient: the
// roles of M and C exchanged, conv_op_impl.h:524-560).
__global__ __launch_bounds__(kThreads) void pw_f16_pack_filter_kernel(const float* __restrict__ w, int M, int C,
                                                                      uint4* __restrict__ wf,
                                                                      uint4* __restrict__ wd) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  const int CB = (C + 7) >> 3, MB = (M + 7) >> 3;
  if (wf && i < CB * M) {
    const int m = i % M, cb = i / M;
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = cb * 8 + e < C ? (_Float16)w[(long long)m * C + cb * 8 + e] : (_Float16)0.0f;
    wf[i] = __builtin_bit_cast(uint4, o);
  }
  if (wd && i < MB * C) {
    const int c = i % C, mb = i / C;
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e] = mb * 8 + e < M ? (_Float16)w[(long long)(mb * 8 + e) * C + c] : (_Float16)0.0f;
    wd[i] = __builtin_bit_cast(uint4, o);
  }
}

// ---- elementwise passes on blocked fp16 tensors (one thread per 16-byte slot) ------------------
struct EwF16 {
  const uint4* a;
  const uint4* b;
  uint4* y;
  int N, CB, H, W;       // geometry of y
  int mode, stride, accumulate;
  int Ha, Wa;            // EW_SUBSAMPLE: a's own map (H stride x W stride, or an odd map one short of it)
};
enum { EW_SUBSAMPLE = 0, EW_SUBSAMPLE_GRAD = 1, EW_UPSAMPLE_GRAD = 2, EW_SUM2 = 3, EW_RELU = 4, EW_RELU_GRAD = 5 };

__device__ __forceinline__ uint4 h8_add(const uint4& u, const uint4& v) {
  const half8 a = as_half8(u), b = as_half8(v);
  half8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (_Float16)((float)a[e] + (float)b[e]);
  return __builtin_bit_cast(uint4, o);
}

__global__ __launch_bounds__(kThreads) void ew_f16_kernel(const EwF16 p) {
  const long long plane = (long long)p.H * p.W;
  const long long total = (long long)p.N * p.CB * plane;
  const uint4 zero = make_uint4(0, 0, 0, 0);
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (long long)gridDim.x * kThreads) {
    const long long ncb = i / plane;
    const int r = (int)(i % plane), y = r / p.W, x = r % p.W;
    uint4 v;
    switch (p.mode) {
      case EW_SUBSAMPLE: {          // y[.., y, x] = a[.., s y, s x]   (a is H s x W s: a strided 1x1 conv's view)
        v = p.a[(ncb * p.Ha + (long long)y * p.stride) * p.Wa + (long long)x * p.stride];
        break;
      }
      case EW_SUBSAMPLE_GRAD: {     // y[.., y, x] (+)= (y, x both multiples of s) ? a[.., y / s, x / s] : 0
        const int Ha = (p.H + p.stride - 1) / p.stride, Wa = (p.W + p.stride - 1) / p.stride;
        const bool hit = (y % p.stride) == 0 && (x % p.stride) == 0;
        v = hit ? p.a[(ncb * Ha + y / p.stride) * Wa + x / p.stride] : zero;
        if (p.accumulate) v = h8_add(p.y[i], v);
        break;
      }
      case EW_UPSAMPLE_GRAD: {      // y = sum of the 2 x 2 block of a (a is 2H x 2W): upsample_nearest_op.cu:62-151
        const int Wa = p.W * 2;
        const uint4* s = p.a + (ncb * p.H * 2 + 2LL * y) * Wa + 2 * x;
        const half8 a0 = as_half8(s[0]), a1 = as_half8(s[1]), a2 = as_half8(s[Wa]), a3 = as_half8(s[Wa + 1]);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)(((float)a0[e] + (float)a1[e]) + ((float)a2[e] + (float)a3[e]));
        v = __builtin_bit_cast(uint4, o);
        if (p.b) v = h8_add(v, p.b[i]);     // + the level's own gradient (the lateral Sum's other input)
        break;
      }
      case EW_SUM2:
        v = h8_add(p.a[i], p.b[i]);
        break;
      case EW_RELU: {
        half8 a = as_half8(p.a[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = a[e] > (_Float16)0.0f ? a[e] : (_Float16)0.0f;
        v = __builtin_bit_cast(uint4, a);
        break;
      }
      default: {                    // EW_RELU_GRAD: y = a (activation) > 0 ? b (gradient) : 0
        const half8 a = as_half8(p.a[i]);
        half8 g = as_half8(p.b[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = a[e] > (_Float16)0.0f ? g[e] : (_Float16)0.0f;
        v = __builtin_bit_cast(uint4, g);
        break;
      }
    }
    p.y[i] = v;
  }
}

// Stem tail in config 5's precision: bias + ReLU + 3x3 / stride 2 / pad 1 max pool over the fp32
// NCHW output of the 7x7 convolution, written channel-blocked fp16 (the only consumer is res2).
__global__ __launch_bounds__(kThreads) void stem_pool_f16_kernel(const float* __restrict__ z,
                                                                 const float* __restrict__ bias, int N, int C,
                                                                 int H, int W, uint4* __restrict__ y) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, CB = C >> 3;
  const long long total = (long long)N * CB * Ho * Wo;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (long long)gridDim.x * kThreads) {
    const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
    const long long ncb = i / ((long long)Wo * Ho);
    const int cb = (int)(ncb % CB);
    const long long n = ncb / CB;
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float* src = z + ((n * C + cb * 8 + e) * H) * (long long)W;
      float m = -3.0e38f;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = 2 * oy + dy, xx = 2 * ox + dx;
          if (yy >= 0 && yy < H && xx >= 0 && xx < W) m = fmaxf(m, src[(long long)yy * W + xx]);
        }
      o[e] = (_Float16)fmaxf(m + bias[cb * 8 + e], 0.0f);      // max commutes with + bias and ReLU
    }
    y[i] = __builtin_bit_cast(uint4, o);
  }
}

inline unsigned grid_for(long long n) {
  const long long b = (n + kThreads - 1) / kThreads;
  return (unsigned)(b < 1 ? 1 : (b > 65535 * 8 ? 65535 * 8 : b));
}

template <int PT>
int launch_pw(PwF16 p, hipStream_t s) {
  using G = Geo<PT>;
  const long long tiles = (p.total + PT - 1) / PT;
  const long long wgs = (tiles + 7) / 8 * 8 * p.mblocks;
  if (wgs >= (1LL << 31)) return SSAD_E_BADARG;
  const int nchunks = (((p.C + 7) >> 3) + CBC - 1) / CBC;
  p.stages = nchunks + 1 < G::S ? (nchunks + 1 < 2 ? 2 : nchunks + 1) : G::S;
  const size_t lds_bytes = (size_t)p.stages * G::STAGE * 16;
  static const bool attr = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(pw_f16_kernel<PT, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)G::S * G::STAGE * 16)) ==
               hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(pw_f16_kernel<PT, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)G::S * G::STAGE * 16)) ==
               hipSuccess;
  }();
  if (!attr) return SSAD_E_BADARG;
  if (p.mask) hipLaunchKernelGGL((pw_f16_kernel<PT, true>), dim3((unsigned)wgs), dim3(kThreads), lds_bytes, s, p);
  else hipLaunchKernelGGL((pw_f16_kernel<PT, false>), dim3((unsigned)wgs), dim3(kThreads), lds_bytes, s, p);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

size_t ssad_pw_f16_filter_halves(int M, int C) {
  const size_t f = (size_t)((C + 7) & ~7) * (size_t)M, d = (size_t)((M + 7) & ~7) * (size_t)C;
  return f > d ? f : d;
}

int ssad_pw_f16_pack_filter(const float* w, int M, int C, void* wf, void* wd, ssad_stream_t stream) {
  if (!w || M < 1 || C < 1 || (!wf && !wd)) return SSAD_E_BADARG;
  const int nf = ((C + 7) >> 3) * M, nd = ((M + 7) >> 3) * C;
  const int n = nf > nd ? nf : nd;
  hipLaunchKernelGGL(pw_f16_pack_filter_kernel, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0,
                     (hipStream_t)stream, w, M, C, static_cast<uint4*>(wf), static_cast<uint4*>(wd));
  return (int)hipGetLastError();
}

int ssad_conv1x1_f16(const ssad_pw_f16* d, ssad_stream_t stream) {
  if (!d || !d->x || !d->w || !d->y || d->N < 0 || d->C < 1 || d->M < 8 || (d->M & 7) || d->Ho < 1 || d->Wo < 1 ||
      (d->stride != 1 && d->stride != 2))
    return SSAD_E_BADARG;
  if (d->N == 0) return 0;
  PwF16 p;
  p.x = static_cast<const uint4*>(d->x);
  p.w = static_cast<const uint4*>(d->w);
  p.bias = d->bias;
  p.res = static_cast<const uint4*>(d->residual);
  p.mask = static_cast<const uint4*>(d->mask);
  p.y = static_cast<uint4*>(d->y);
  p.N = d->N; p.C = d->C; p.M = d->M; p.Ho = d->Ho; p.Wo = d->Wo;
  p.stride = d->stride;
  p.Hi = d->Hi > 0 ? d->Hi : d->Ho * d->stride;
  p.Wi = d->Wi > 0 ? d->Wi : d->Wo * d->stride;
  if ((d->Ho - 1) * d->stride >= p.Hi || (d->Wo - 1) * d->stride >= p.Wi) return SSAD_E_BADARG;
  p.relu = (d->flags & SSAD_CONV_RELU) != 0;
  p.res_up = (d->flags & SSAD_PW_F16_RES_UPSAMPLE2) != 0;
  if (p.res_up && (!p.res || (d->Ho & 1) || (d->Wo & 1))) return SSAD_E_BADARG;
  p.mblocks = (d->M + MT - 1) / MT;
  p.total = (long long)d->N * d->Ho * d->Wo;
  const int CB = (d->C + 7) >> 3;
  // buffer addressing: byte offsets below 2^31
  if ((long long)d->N * (CB + CBC) * p.Hi * p.Wi * 16 >= (1LL << 31) ||
      (long long)d->N * (d->M >> 3) * d->Ho * d->Wo * 16 >= (1LL << 31) || (long long)(CB + CBC) * d->M * 16 >= (1LL << 31))
    return SSAD_E_BADARG;
  // 256-pixel tiles unless they leave the chip under-filled (two workgroups per CU are resident)
  const int cus = ssad_cu_count();
  const long long wg256 = ((p.total + 255) / 256) * p.mblocks;
  if (wg256 >= 2LL * cus) return launch_pw<256>(p, (hipStream_t)stream);
  return launch_pw<128>(p, (hipStream_t)stream);
}

int ssad_f16_elementwise(int mode, const void* a, const void* b, void* y, int N, int C, int H, int W, int stride,
                         int accumulate, ssad_stream_t stream) {
  if (mode < EW_SUBSAMPLE || mode > EW_RELU_GRAD || !a || !y || N < 0 || C < 1 || H < 1 || W < 1) return SSAD_E_BADARG;
  if ((mode == EW_SUM2 || mode == EW_RELU_GRAD) && !b) return SSAD_E_BADARG;
  if ((mode == EW_SUBSAMPLE || mode == EW_SUBSAMPLE_GRAD) && stride < 1) return SSAD_E_BADARG;
  if (N == 0) return 0;
  EwF16 p;
  p.a = static_cast<const uint4*>(a);
  p.b = static_cast<const uint4*>(b);
  p.y = static_cast<uint4*>(y);
  p.N = N; p.CB = (C + 7) >> 3; p.H = H; p.W = W;
  p.mode = mode; p.stride = stride; p.accumulate = accumulate;
  p.Ha = H * stride; p.Wa = W * stride;
  hipLaunchKernelGGL(ew_f16_kernel, dim3(grid_for((long long)N * p.CB * H * W)), dim3(kThreads), 0,
                     (hipStream_t)stream, p);
  return (int)hipGetLastError();
}

int ssad_f16_subsample(const void* a, int N, int C, int Hi, int Wi, int stride, void* y, ssad_stream_t stream) {
  if (!a || !y || N < 0 || C < 1 || Hi < 1 || Wi < 1 || stride < 1) return SSAD_E_BADARG;
  if (N == 0) return 0;
  EwF16 p;
  p.a = static_cast<const uint4*>(a);
  p.b = nullptr;
  p.y = static_cast<uint4*>(y);
  p.N = N; p.CB = (C + 7) >> 3; p.H = (Hi - 1) / stride + 1; p.W = (Wi - 1) / stride + 1;
  p.mode = EW_SUBSAMPLE; p.stride = stride; p.accumulate = 0;
  p.Ha = Hi; p.Wa = Wi;
  hipLaunchKernelGGL(ew_f16_kernel, dim3(grid_for((long long)N * p.CB * p.H * p.W)), dim3(kThreads), 0,
                     (hipStream_t)stream, p);
  return (int)hipGetLastError();
}

int ssad_stem_pool_f16(const float* z, const float* bias, int N, int C, int H, int W, void* y,
                       ssad_stream_t stream) {
  if (!z || !bias || !y || N < 0 || C < 8 || (C & 7) || H < 1 || W < 1) return SSAD_E_BADARG;
  if (N == 0) return 0;
  const long long total = (long long)N * (C >> 3) * ((H + 1) / 2) * ((W + 1) / 2);
  hipLaunchKernelGGL(stem_pool_f16_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, z, bias, N,
                     C, H, W, static_cast<uint4*>(y));
  return (int)hipGetLastError();
}

}  // extern "C"
