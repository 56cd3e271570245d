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
#include <stdlib.h>

#include "conv_internal.h"
#include "ssad_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;
using ssad_dev::uniform_rsrc;

constexpr int kThreads = 256;
constexpr int TS = 16;                 // output tile edge
constexpr int HS = TS + 2;             // halo tile edge
constexpr int CBC = 4;                 // 8-channel blocks per K chunk (32 channels)
constexpr int SLOTS = CBC * HS * HS;   // 16-byte slots per stage (1296)
constexpr int NDMA = (SLOTS + 63) / 64;   // wave-level LDS-DMA instructions per stage (21)
constexpr int STAGE = NDMA * 64;           // stage pitch in slots (1344: the last instruction's tail)
constexpr int NLD = (NDMA + 3) / 4;        // DMA instructions per wave (6; waves 1-3 issue 5)
constexpr int AD = 6;                      // filter ring depth, K-steps
constexpr unsigned kOob = 0x80000000u;     // buffer offset past any descriptor: loads 0
constexpr int MT = 128;                // output channels per workgroup

struct F16Conv {
  const uint4* x;        // blocked fp16 input
  const uint4* w;        // packed filter [9][C/8][M] x 16 B
  const float* bias;     // [M] or null
  const _Float16* aux;   // blocked fp16 like y, or null: y = aux > 0 ? y : 0 (fused ReluGradient)
  void* y;               // blocked fp16 or NCHW fp32
  int N, C, H, W, M;
  int tiles_x, tiles_y;
  int relu, sigmoid, out_nchw_f32;
};

// every FPN level that shares the filter, in one launch (the small levels alone cannot fill
// the chip: P5..P7 are 64 / 16 / 16 workgroups at batch 16)
struct F16Levels {
  const uint4* x[SSAD_MAX_F16_LEVELS];
  void* y[SSAD_MAX_F16_LEVELS];
  const _Float16* aux[SSAD_MAX_F16_LEVELS];
  int N[SSAD_MAX_F16_LEVELS], H[SSAD_MAX_F16_LEVELS], W[SSAD_MAX_F16_LEVELS];
  int tile0[SSAD_MAX_F16_LEVELS + 1];      // first workgroup (blockIdx.x) of each level
  const uint4* w[SSAD_MAX_F16_LEVELS];     // per entry: independent problems share a launch
  const float* bias[SSAD_MAX_F16_LEVELS];
  int n_levels;
  int C, M, relu, sigmoid, out_nchw_f32;
  int mblocks;                             // 128-wide output-channel blocks
};

__device__ __forceinline__ half8 as_half8(const uint4& v) {
  return __builtin_bit_cast(half8, v);
}

#ifdef F16_TIMELINE       // debug build only (tools/f16_timeline.py): s_memtime stamps of wave 0 per workgroup
__device__ unsigned long long g_f16_dbg[4096][4];
#define STAMP(k) \
  if (threadIdx.x == 0 && blockIdx.x < 4096) g_f16_dbg[blockIdx.x][k] = __builtin_readcyclecounter()
#else
#define STAMP(k)
#endif

#ifndef F16_ABLATE        // debug builds (make EXTRA=-DF16_ABLATE=n): 1 no halo fetch, 2 no filter reload, 4 no LDS reads
#define F16_ABLATE 0
#endif
constexpr int kOutBlocked = 0, kOutMasked = 1, kOutNchw = 2;

template <int OUT>        // output form: blocked fp16, blocked fp16 under the ReluGradient mask, NCHW fp32
__global__ __launch_bounds__(kThreads, 2) void conv3x3_f16_kernel(const F16Levels q) {
  constexpr int DBG = F16_ABLATE;
  __shared__ uint4 lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  // the wave index as a scalar: everything derived from it (channel block, LDS-DMA slots) is then
  // wave-uniform for the compiler too -- otherwise each buffer access with such a scalar offset is
  // wrapped in a readfirstlane waterfall loop
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave & 1, wp = wave >> 1;
  const int j = lane & 31, h = lane >> 5;
  // Workgroup -> (tile, output-channel block).  Consecutive workgroup ids go round the 8 XCDs, so
  // ids b, b + 8, b + 16 ... share an L2: the channel blocks of one tile are laid out along that
  // sequence and fetch the tile's input from HBM once.
  STAMP(0);
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int mb = seq % q.mblocks;
  int t = (seq / q.mblocks) * 8 + xcd;
  if (t >= q.tile0[q.n_levels]) return;
  int lv = 0;
  for (int l = 1; l < q.n_levels; ++l)
    if (t >= q.tile0[l]) lv = l;
  F16Conv p;
  p.x = q.x[lv]; p.w = q.w[lv]; p.bias = q.bias[lv]; p.aux = q.aux[lv]; p.y = q.y[lv];
  p.N = q.N[lv]; p.C = q.C; p.H = q.H[lv]; p.W = q.W[lv]; p.M = q.M;
  p.tiles_x = (p.W + TS - 1) / TS; p.tiles_y = (p.H + TS - 1) / TS;
  p.relu = q.relu; p.sigmoid = q.sigmoid; p.out_nchw_f32 = q.out_nchw_f32;
  t -= q.tile0[lv];
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int n = t / p.tiles_y;
  const int y0 = ty * TS, x0 = tx * TS;
  const int ocb = mb * MT;
  const int CB = (p.C + 7) >> 3;          // channel blocks; a tail block is zero padded by the packers
  const long long plane = (long long)p.H * p.W;
  const int plane16 = (int)plane * 16;
  const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(p.x, (unsigned)((long long)p.N * CB * plane16));
  const __amdgpu_buffer_rsrc_t wrs = uniform_rsrc(p.w, (unsigned)(9LL * CB * p.M * 16));

  // ---- halo staging by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write.
  // Stage slot s = (block, row, col) of the 18 x 18 x 4-block halo tile; wave-level instruction k
  // writes slots [64 k, 64 k + 64); wave w issues k = w, w + 4, ...  Lanes outside the image (and
  // past the last channel block) go to an out-of-range offset: the DMA writes zeros for them.
  unsigned dvo[NLD];
  int dcb[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int s = 64 * (wave + 4 * i) + lane;
    const int cbl = s / (HS * HS), r = s % (HS * HS);
    const int gy = y0 - 1 + r / HS, gx = x0 - 1 + r % HS;
    const bool ok = s < SLOTS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    dcb[i] = cbl;
    dvo[i] = ok ? (unsigned)((((long long)n * CB + cbl) * plane + (long long)gy * p.W + gx) * 16) : kOob;
  }
  auto fetch = [&](int chunk, int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int k = wave + 4 * i;
      if (k < NDMA) {                                        // wave-uniform
        const unsigned vo = (chunk * CBC + dcb[i] < CB) ? dvo[i] : kOob;     // K padding = 0
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr)(lds + buf * STAGE + 64 * k), 16, vo,
                                                 chunk * CBC * plane16, 0, 0);
      }
    }
  };

  // ---- operand addressing
  // B (pixels): MFMA column j -> tile row 2 tt + (j >> 4) of this wave's 8 rows, column j & 15
  // The second row's columns are rotated by 2: ds_read_b128 is serviced in the 16-lane groups
  // {0-3,12-15,20-27} and {4-11,16-19,28-31} (+32), and with the natural order the two rows of
  // a group (18 slots apart) collide on two 16-byte bank slots -- 2x the LDS cycles.
  const int brow = wp * 8 + (j >> 4), bcol = (j & 16) ? ((j - 2) & 15) : j;
  const int bbase = (h * HS + brow) * HS + bcol;           // + ((2 ks) * HS + 2 tt + dy) * HS + dx
  // A (filter): Wp[tap][cb][m] x 16 B; lane = (row m, 8-channel half h).  Rows beyond M are
  // clamped (never stored); a block beyond the last meets B = 0 (or reads zeros past the pack).
  unsigned avo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oc = ocb + wo * 64 + i * 32 + j;
    avo[i] = (unsigned)(h * p.M + (oc < p.M ? oc : p.M - 1)) * 16u;
  }
  auto load_a = [&](int chunk, int qq, half8 (&a)[2]) {     // qq = ks * 9 + tap
    const int ks = qq / 9, tap = qq % 9;
    const int soff = ((tap * CB + chunk * CBC + 2 * ks) * p.M) * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      a[i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wrs, avo[i], soff, 0));
  };

  float16v acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][tt][r] = 0.0f;

  const int nchunks = (CB + CBC - 1) / CBC;
  // Filter ring, AD K-steps deep.  Vector-memory results return in order, so a filter load issued
  // behind the halo DMA of the next chunk (HBM latency) cannot return before it: the AD steps
  // loaded ahead of the DMA are what the MFMAs run on meanwhile.  Every phase of a step is fenced
  // (sched_barrier) -- left alone, hipcc sinks each filter load to just above its first use and
  // the ring degenerates to one step of cover.
  half8 ar[AD][2];
  fetch(0, 0);
#pragma unroll
  for (int qq = 0; qq < AD; ++qq) load_a(0, qq, ar[qq]);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * AD) : "memory");       // the DMA has landed
  __builtin_amdgcn_s_barrier();
  STAMP(1);
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    const uint4* tile = lds + (c & 1) * STAGE;
    half8 b[2][4];
    auto read_b = [&](int qq, half8 (&bb)[4]) {
      const int ks = qq / 9, tap = qq % 9, dy = tap / 3, dx = tap % 3;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
        bb[tt] = as_half8(tile[bbase + ((2 * ks) * HS + 2 * tt + dy) * HS + dx]);
    };
    read_b(0, b[0]);
    // the other buffer was last read in chunk c - 1, which every wave has left (barrier above)
    if (more && !(DBG & 1)) fetch(c + 1, (c + 1) & 1);
#pragma unroll
    for (int qq = 0; qq < 18; ++qq) {
      if (qq + 1 < 18 && !(DBG & 4)) read_b(qq + 1, b[(qq + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        acc[0][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[qq % AD][0], b[qq & 1][tt], acc[0][tt], 0, 0, 0);
        acc[1][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[qq % AD][1], b[qq & 1][tt], acc[1][tt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // refill this ring slot (its MFMAs have issued) with the filter of step qq + AD
      if (!(DBG & 2)) {
        if (qq + AD < 18) load_a(c, qq + AD, ar[qq % AD]);
        else if (more) load_a(c + 1, qq + AD - 18, ar[qq % AD]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) {
      // all but the newest 2 AD loads (the ring of the next chunk's first steps) have returned,
      // this wave's DMA among them; after the barrier everybody's has
      if (DBG & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * AD) : "memory");
      __builtin_amdgcn_s_barrier();
    }
  }

  STAMP(2);
  // ---- epilogue: C/D row = (r & 3) + 8 (r >> 2) + 4 h, column = j.  Branch-free buffer accesses:
  // a lane's pixel gives the vector offset (out-of-image lanes: an out-of-range offset, the store
  // is dropped), the channel block is wave-uniform and goes into the scalar offset.  (The first
  // version computed a 64-bit address with integer multiplies behind three run-time branches
  // per store: the epilogue took 28 % of a workgroup's life, tools/f16_timeline.py.)
  const int oc_w = ocb + wo * 64;                       // this wave's first output channel
  unsigned pvo[4];                                      // per MFMA column tile: this lane's pixel
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int gy = y0 + wp * 8 + 2 * tt + (j >> 4), gx = x0 + bcol;
    const bool okp = gy < p.H && gx < p.W;
    const long long pix = (long long)gy * p.W + gx;
    if (OUT == kOutNchw) pvo[tt] = okp ? (unsigned)((((long long)n * p.M + 4 * h) * plane + pix) * 4) : kOob;
    else pvo[tt] = okp ? (unsigned)((((long long)n * (p.M >> 3)) * plane + pix) * 16 + 8 * h) : kOob;
  }
  const unsigned ybytes = OUT == kOutNchw ? (unsigned)((long long)p.N * p.M * plane * 4)
                                          : (unsigned)((long long)p.N * (p.M >> 3) * plane16);
  const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(p.y, ybytes);
  const __amdgpu_buffer_rsrc_t brs = uniform_rsrc(p.bias ? (const void*)p.bias : (const void*)p.y,
                                                  p.bias ? (unsigned)p.M * 4u : 0u);   // no bias: reads 0
  // bias of the lane's 32 channels: 8 x 16 bytes (channels past M read 0; they are never stored)
  float4 bq[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      bq[i][g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(brs, (unsigned)h * 16u,
                                                                                (oc_w + i * 32 + 8 * g) * 4, 0));
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oc0 = oc_w + i * 32;
    uint2 mk[4][4];
    if (OUT == kOutMasked) {                            // relu_op.cu:44-53: dX = Y > 0 ? dY : 0
      const __amdgpu_buffer_rsrc_t mrs = uniform_rsrc(p.aux, ybytes);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          mk[tt][g] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(
                                                    mrs, pvo[tt], ((oc0 >> 3) + g) * plane16, 0));
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (oc0 + 8 * g >= p.M) continue;               // wave-uniform
        if (OUT == kOutNchw) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][tt][4 * g + e] + bq[i][g][e];
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
          }
          if (p.sigmoid) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = 1.0f / (1.0f + __expf(-v[e]));     // sigmoid_op.cu:25-29
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned vo = (oc0 + 8 * g + 4 * h + e < p.M) ? pvo[tt] : kOob;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), yrs, vo,
                                                  (oc0 + 8 * g + e) * ((int)plane * 4), 0);
          }
        } else {
          // packed arithmetic (v_pk_add_f32, v_cvt_pk_f16_f32, v_pk_max_f16): the epilogue is VALU
          // time taken from the MFMAs of the wave sharing the SIMD.  ReLU after the rounding gives
          // the same fp16 value as before it (rounding is monotonic and keeps the sign).
          const float2v lo = float2v{acc[i][tt][4 * g], acc[i][tt][4 * g + 1]} + float2v{bq[i][g].x, bq[i][g].y};
          const float2v hi = float2v{acc[i][tt][4 * g + 2], acc[i][tt][4 * g + 3]} + float2v{bq[i][g].z, bq[i][g].w};
          half2v o01 = __builtin_convertvector(lo, half2v), o23 = __builtin_convertvector(hi, half2v);
          if (p.relu) {
            const half2v z = {(_Float16)0.0f, (_Float16)0.0f};
            o01 = __builtin_elementwise_max(o01, z);
            o23 = __builtin_elementwise_max(o23, z);
          }
          half4 o = {o01[0], o01[1], o23[0], o23[1]};
          if (OUT == kOutMasked) {
            const half4 m = __builtin_bit_cast(half4, mk[tt][g]);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = m[e] > (_Float16)0.0f ? o[e] : (_Float16)0.0f;
          }
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uint2v, o), yrs, pvo[tt],
                                                ((oc0 >> 3) + g) * plane16, 0);
        }
      }
    }
  }
  STAMP(3);
}

// NCHW fp32 -> blocked fp16 (round to nearest even), one thread per 16-byte slot
template <typename T>   // T = float (the subnet pipeline) or _Float16 (float16 blobs of the operator surface)
__global__ __launch_bounds__(kThreads) void f16_pack_kernel(const T* __restrict__ x,
                                                            uint4* __restrict__ xb, int N, int C,
                                                            long long plane, float scale,
                                                            const float* __restrict__ scale_dev) {
  if (scale_dev) scale *= scale_dev[0];     // dynamic loss scale, kept on the device
  const int CB = (C + 7) >> 3;
  const long long total = (long long)N * CB * plane;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (long long)gridDim.x * kThreads) {
    const long long px = i % plane, ncb = i / plane;      // ncb = n * CB + cb
    const int cb = (int)(ncb % CB);
    const T* src = x + ((ncb / CB) * C + cb * 8) * plane + px;
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e] = cb * 8 + e < C ? (_Float16)((float)src[e * plane] * scale) : (_Float16)0.0f;
    xb[i] = __builtin_bit_cast(uint4, o);
  }
}

// blocked fp16 -> NCHW fp32: one thread per (n, cb, pixel); 8 strided 4-byte stores
template <typename T>
__global__ __launch_bounds__(kThreads) void f16_unpack_kernel(const uint4* __restrict__ xb,
                                                              T* __restrict__ x, int N, int C,
                                                              long long plane, float scale,
                                                              const float* __restrict__ scale_dev) {
  if (scale_dev) scale *= scale_dev[0];
  const int CB = (C + 7) >> 3;
  const long long total = (long long)N * CB * plane;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (long long)gridDim.x * kThreads) {
    const long long px = i % plane, ncb = i / plane;
    const int cb = (int)(ncb % CB);
    const half8 v = as_half8(xb[i]);
    T* dst = x + ((ncb / CB) * C + cb * 8) * plane + px;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (cb * 8 + e < C) dst[e * plane] = (T)((float)v[e] * scale);
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

// every entry of a table in one launch: the entries' 256-slot blocks are laid end to end over blockIdx.x (block_start:
// an entry's first block; a grid of [largest entry] x [entries] launched 590 K workgroups for the R-101 student, nine
// in ten of them empty: 0.51 ms); taps = 1 is the pointwise layout of gemm_f16.hip (the 3x3 layout with one tap, no flip)
struct F16PackTable {
  ssad_f16_pack_entry e[SSAD_MAX_F16_PACK_ENTRIES];
  int block_start[SSAD_MAX_F16_PACK_ENTRIES + 1];
  int n;
};
__global__ __launch_bounds__(kThreads) void f16_pack_multi_kernel(const F16PackTable t) {
  int k = 0;
  for (int j = 1; j < t.n; ++j) k += (int)blockIdx.x >= t.block_start[j];
  const ssad_f16_pack_entry e = t.e[k];
  const int M = e.M, C = e.C, taps = e.taps;
  const int CB = (C + 7) >> 3, MB = (M + 7) >> 3;
  const long long i = (long long)((int)blockIdx.x - t.block_start[k]) * kThreads + threadIdx.x;
  const float* __restrict__ w = e.w;
  if (i >= (long long)taps * CB * M && i >= (long long)taps * MB * C) return;
  if (e.wf && i < (long long)taps * CB * M) {
    const int m = (int)(i % M), cb = (int)((i / M) % CB), tap = (int)(i / ((long long)M * CB));
    half8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      o[k] = cb * 8 + k < C ? (_Float16)w[((long long)m * C + cb * 8 + k) * taps + tap] : (_Float16)0.0f;
    static_cast<uint4*>(e.wf)[i] = __builtin_bit_cast(uint4, o);
  }
  if (e.wd && i < (long long)taps * MB * C) {
    const int c = (int)(i % C), mb = (int)((i / C) % MB), tap = (int)(i / ((long long)C * MB));
    half8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      o[k] = mb * 8 + k < M ? (_Float16)w[((long long)(mb * 8 + k) * C + c) * taps + (taps - 1 - tap)] : (_Float16)0.0f;
    static_cast<uint4*>(e.wd)[i] = __builtin_bit_cast(uint4, o);
  }
}

template <typename A, typename B>
__global__ __launch_bounds__(kThreads) void cast_kernel(const A* __restrict__ in, B* __restrict__ out,
                                                        long long n) {
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads)
    out[i] = (B)(float)in[i];
}

inline unsigned grid_for(long long n) {
  const long long b = (n + kThreads - 1) / kThreads;
  return (unsigned)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b));
}

}  // namespace

extern "C" {

int ssad_f16_pack_activations_dyn(const float* x, int N, int C, int H, int W, float scale,
                                  const float* scale_dev, void* xb, ssad_stream_t stream) {
  if (!x || !xb || N < 0 || C < 1 || H < 1 || W < 1) return SSAD_E_BADARG;
  if (N == 0) return 0;
  const long long plane = (long long)H * W;
  hipLaunchKernelGGL(f16_pack_kernel<float>, dim3(grid_for((long long)N * ((C + 7) >> 3) * plane)),
                     dim3(kThreads), 0, (hipStream_t)stream, x, static_cast<uint4*>(xb), N, C, plane, scale,
                     scale_dev);
  return (int)hipGetLastError();
}

int ssad_f16_pack_activations(const float* x, int N, int C, int H, int W, float scale, void* xb,
                              ssad_stream_t stream) {
  return ssad_f16_pack_activations_dyn(x, N, C, H, W, scale, nullptr, xb, stream);
}

int ssad_f16_unpack_activations(const void* xb, int N, int C, int H, int W, float scale, float* x,
                                ssad_stream_t stream) {
  return ssad_f16_unpack_activations_dyn(xb, N, C, H, W, scale, nullptr, x, stream);
}

int ssad_f16_unpack_activations_dyn(const void* xb, int N, int C, int H, int W, float scale,
                                    const float* scale_dev, float* x, ssad_stream_t stream) {
  if (!x || !xb || N < 0 || C < 1 || H < 1 || W < 1) return SSAD_E_BADARG;
  if (N == 0) return 0;
  const long long plane = (long long)H * W;
  hipLaunchKernelGGL(f16_unpack_kernel<float>, dim3(grid_for((long long)N * ((C + 7) >> 3) * plane)),
                     dim3(kThreads), 0, (hipStream_t)stream, static_cast<const uint4*>(xb), x, N, C,
                     plane, scale, scale_dev);
  return (int)hipGetLastError();
}

int ssad_f16_block_activations(const void* x_nchw_f16, int N, int C, int H, int W, void* xb,
                               ssad_stream_t stream) {
  if (!x_nchw_f16 || !xb || N < 0 || C < 1 || H < 1 || W < 1) return SSAD_E_BADARG;
  if (N == 0) return 0;
  const long long plane = (long long)H * W;
  hipLaunchKernelGGL(f16_pack_kernel<_Float16>, dim3(grid_for((long long)N * ((C + 7) >> 3) * plane)),
                     dim3(kThreads), 0, (hipStream_t)stream, static_cast<const _Float16*>(x_nchw_f16),
                     static_cast<uint4*>(xb), N, C, plane, 1.0f, (const float*)nullptr);
  return (int)hipGetLastError();
}

int ssad_f16_unblock_activations(const void* xb, int N, int C, int H, int W, void* x_nchw_f16,
                                 ssad_stream_t stream) {
  if (!x_nchw_f16 || !xb || N < 0 || C < 1 || H < 1 || W < 1) return SSAD_E_BADARG;
  if (N == 0) return 0;
  const long long plane = (long long)H * W;
  hipLaunchKernelGGL(f16_unpack_kernel<_Float16>, dim3(grid_for((long long)N * ((C + 7) >> 3) * plane)),
                     dim3(kThreads), 0, (hipStream_t)stream, static_cast<const uint4*>(xb),
                     static_cast<_Float16*>(x_nchw_f16), N, C, plane, 1.0f, (const float*)nullptr);
  return (int)hipGetLastError();
}

int ssad_cast_f16_to_f32(const void* in, float* out, long long n, ssad_stream_t stream) {
  if (n < 0 || (n > 0 && (!in || !out))) return SSAD_E_BADARG;
  if (n == 0) return 0;
  hipLaunchKernelGGL((cast_kernel<_Float16, float>), dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream,
                     static_cast<const _Float16*>(in), out, n);
  return (int)hipGetLastError();
}

int ssad_cast_f32_to_f16(const float* in, void* out, long long n, ssad_stream_t stream) {
  if (n < 0 || (n > 0 && (!in || !out))) return SSAD_E_BADARG;
  if (n == 0) return 0;
  hipLaunchKernelGGL((cast_kernel<float, _Float16>), dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, in,
                     static_cast<_Float16*>(out), n);
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

int ssad_f16_pack_filters(const ssad_f16_pack_entry* entries, int n_entries, ssad_stream_t stream) {
  if (n_entries < 0 || (n_entries > 0 && !entries)) return SSAD_E_BADARG;
  for (int i = 0; i < n_entries; ++i) {
    const ssad_f16_pack_entry& e = entries[i];
    if (!e.w || (!e.wf && !e.wd) || e.M < 1 || e.C < 1 || (e.taps != 1 && e.taps != 9)) return SSAD_E_BADARG;
  }
  for (int base = 0; base < n_entries; base += SSAD_MAX_F16_PACK_ENTRIES) {
    const int cnt = n_entries - base < SSAD_MAX_F16_PACK_ENTRIES ? n_entries - base : SSAD_MAX_F16_PACK_ENTRIES;
    F16PackTable t;
    long long blocks = 0;
    t.n = cnt;
    for (int i = 0; i < SSAD_MAX_F16_PACK_ENTRIES; ++i) {
      t.e[i] = i < cnt ? entries[base + i] : ssad_f16_pack_entry{nullptr, nullptr, nullptr, 0, 0, 1, 0};
      t.block_start[i] = (int)blocks;
      if (i < cnt) {
        const ssad_f16_pack_entry& e = t.e[i];
        const long long nf = e.wf ? (long long)e.taps * ((e.C + 7) >> 3) * e.M : 0;
        const long long nd = e.wd ? (long long)e.taps * ((e.M + 7) >> 3) * e.C : 0;
        blocks += ((nf > nd ? nf : nd) + kThreads - 1) / kThreads;
        if (blocks >= (1LL << 31)) return SSAD_E_BADARG;
      }
    }
    t.block_start[SSAD_MAX_F16_PACK_ENTRIES] = (int)blocks;
    hipLaunchKernelGGL(f16_pack_multi_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, t);
  }
  return (int)hipGetLastError();
}

int ssad_conv3x3_forward_f16_levels(const ssad_f16_level* levels, int n_levels, const void* wp,
                                    const float* bias, int C, int M, int flags, ssad_stream_t stream) {
  if (!levels || n_levels < 1 || n_levels > SSAD_MAX_F16_LEVELS || M < 1 || C < 1) return SSAD_E_BADARG;
  const int nchw = (flags & SSAD_F16_OUT_NCHW_F32) != 0;
  const int masked = (flags & SSAD_CONV_MASK_AUX) != 0;
  if (!nchw && (M & 7)) return SSAD_E_BADARG;                   // blocked output: whole 8-blocks
  if (masked && nchw) return SSAD_E_BADARG;
  if ((flags & SSAD_CONV_SIGMOID) && !nchw) return SSAD_E_BADARG;     // probabilities leave as fp32
  F16Levels q;
  long long tiles = 0;
  for (int l = 0; l < n_levels; ++l) {
    const ssad_f16_level& L = levels[l];
    if (!L.x || !L.y || L.N < 0 || L.H < 1 || L.W < 1 || masked != (L.aux != nullptr)) return SSAD_E_BADARG;
    // buffer addressing: byte offsets below 2^31 (2 GiB of blocked fp16 per level)
    if ((long long)L.N * ((C + 7) / 8 + CBC) * L.H * L.W * 16 >= (1LL << 31)) return SSAD_E_BADARG;
    q.x[l] = static_cast<const uint4*>(L.x);
    q.y[l] = L.y;
    q.aux[l] = static_cast<const _Float16*>(L.aux);
    q.N[l] = L.N; q.H[l] = L.H; q.W[l] = L.W;
    q.w[l] = static_cast<const uint4*>(L.packed ? L.packed : wp);
    q.bias[l] = L.packed ? L.bias : bias;
    if (!q.w[l]) return SSAD_E_BADARG;
    q.tile0[l] = (int)tiles;
    tiles += (long long)L.N * ((L.W + TS - 1) / TS) * ((L.H + TS - 1) / TS);
    if (tiles >= (1LL << 31)) return SSAD_E_BADARG;
  }
  for (int l = n_levels; l <= SSAD_MAX_F16_LEVELS; ++l) q.tile0[l] = (int)tiles;
  if (tiles == 0) return 0;
  q.n_levels = n_levels;
  q.C = C; q.M = M;
  q.relu = (flags & SSAD_CONV_RELU) != 0;
  q.sigmoid = (flags & SSAD_CONV_SIGMOID) != 0;
  q.out_nchw_f32 = nchw;
  q.mblocks = (M + MT - 1) / MT;
  if (9LL * ((C + 7) / 8) * M * 16 >= (1LL << 31)) return SSAD_E_BADARG;
  const long long wgs = (tiles + 7) / 8 * 8 * q.mblocks;
  if (wgs >= (1LL << 31)) return SSAD_E_BADARG;
  const dim3 grid((unsigned)wgs);
  // buffer-addressed output (and mask): byte offsets below 2^31 per level
  for (int l = 0; l < n_levels; ++l) {
    const long long px = (long long)levels[l].N * levels[l].H * levels[l].W;
    if ((nchw ? px * M * 4 : px * (M >> 3) * 16) >= (1LL << 31)) return SSAD_E_BADARG;
  }
  if (nchw) hipLaunchKernelGGL(conv3x3_f16_kernel<kOutNchw>, grid, dim3(kThreads), 0, (hipStream_t)stream, q);
  else if (masked) hipLaunchKernelGGL(conv3x3_f16_kernel<kOutMasked>, grid, dim3(kThreads), 0, (hipStream_t)stream, q);
  else hipLaunchKernelGGL(conv3x3_f16_kernel<kOutBlocked>, grid, dim3(kThreads), 0, (hipStream_t)stream, q);
  return (int)hipGetLastError();
}

int ssad_conv3x3_forward_f16(const void* xb, const void* wp, const float* bias, const void* aux, int N,
                             int C, int H, int W, int M, int flags, void* y, ssad_stream_t stream) {
  if (((flags & SSAD_CONV_MASK_AUX) != 0) != (aux != nullptr)) return SSAD_E_BADARG;
  ssad_f16_level L;
  L.x = xb; L.y = y; L.aux = aux; L.N = N; L.H = H; L.W = W;
  L.packed = nullptr; L.bias = nullptr;
  return ssad_conv3x3_forward_f16_levels(&L, 1, wp, bias, C, M, flags, stream);
}

}  // extern "C"

// =========================================================================================
// Filter gradient, fp16 operands / fp32 accumulation (conv_op_impl.h:451-500 computes
// dW = sum_n dY[n] . col[n]^T; here the same sums as GEMMs over PIXELS per filter tap):
//     dW[m][c][ky][kx] = sum_{n,y,x} dY[n][m][y][x] * X[n][c][y + ky - 1][x + kx - 1].
// Both MFMA operands need 8 consecutive K = pixels per lane while the blocked layout keeps
// 8 CHANNELS contiguous -- exactly the case of gfx950's LDS transpose read: a 16-lane group
// of ds_read_b64_tr_b16 turns a [4 pixels][16 channels] block (each lane addressing 4
// contiguous channels of one pixel, any 8-byte-aligned address) into 4 pixels of one channel
// per lane, and a tap's +-1 pixel shift is a whole slot, so every tap stays aligned.
//
// Workgroup = 4 waves, tile 128 output x 128 input channels x the 3 taps of one filter row
// (blockIdx.z = ky); wave (wo, wc) owns 64 x 64 x 3 = 12 accumulator tiles (192 AGPRs: with
// all nine taps a wave would need 18 tiles = 288 registers and hipcc shuttles the excess
// between register files around every MFMA).  K runs over stages of 8 rows x 16 pixels; a
// stage's dY tile and X tile (rows y + ky - 1, 18 of 20 columns used) land in LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds; no staging registers, no ds_write), double buffered.  LDS
// image: two channel blocks interleaved per pixel, [block pair][row][pixel][2][8], pair pitch
// skewed by 8 slots, which makes both 32-lane halves of a transpose read conflict free.
// Per K-step (a row of 16 pixels): 4 + 12 transpose-read pairs feed 12 MFMAs.  The pixel
// range is split over gridDim.y workgroups; partial sums go to a [split][tap][M][C] fp32
// workspace that f16_wgrad_reduce_kernel folds (deterministically) into dW[M][C][3][3].
constexpr int WR = 8;                        // rows per stage
constexpr int WPX = 16;                      // pixels per row segment = one K-step
constexpr int XPW = 20;                      // X tile columns: x0 - 1 .. x0 + 18 (18 used)
constexpr int W_OT = 128, W_CT = 128;        // workgroup tile: output x input channels
constexpr int X_PAIR = WR * XPW * 2;         // slots per X block pair (320 = 5 DMA pieces)
constexpr int Y_PAIR = WR * WPX * 2;         // slots per dY block pair (256 = 4 DMA pieces)
constexpr int X_PITCH = X_PAIR + 8;          // +8 slots = +32 banks between pairs
constexpr int Y_PITCH = Y_PAIR + 8;
constexpr int X_PAIRS = W_CT / 16, Y_PAIRS = W_OT / 16;
constexpr int Y_BASE = X_PAIRS * X_PITCH;                // 2624
constexpr int W_STAGE = Y_BASE + Y_PAIRS * Y_PITCH;      // 4736 slots = 74 KiB per stage
constexpr int X_PIECES = X_PAIRS * (X_PAIR / 64);        // 40
constexpr int Y_PIECES = Y_PAIRS * (Y_PAIR / 64);        // 32

struct F16Wgrad {           // all FPN levels sharing the filter: their stages form one list
  const uint4* x[SSAD_MAX_F16_LEVELS];      // blocked fp16 [N][CB][H][W]
  const uint4* dy[SSAD_MAX_F16_LEVELS];     // blocked fp16 [N][MB][H][W]
  int N[SSAD_MAX_F16_LEVELS], H[SSAD_MAX_F16_LEVELS], W[SSAD_MAX_F16_LEVELS];
  int stage0[SSAD_MAX_F16_LEVELS + 1];      // first stage of each level
  int n_levels;
  float* part;         // [split][9][M][C]
  int C, M;
  int stages;          // sum over levels of N * row groups * row segments
  int cblocks;         // ceil(C / 128)
  int blocks;          // channel-block pairs = ceil(M / 128) * cblocks
  int splits;          // pixel splits
  int xcd_group;       // consecutive work items per XCD run (1 = round robin)
};

typedef short short4v __attribute__((ext_vector_type(4)));

// 8 consecutive pixels of one channel: two transpose reads 4 pixels (8 slots) apart.
// Inline assembly on purpose: behind an LDS-DMA in flight hipcc puts `s_waitcnt vmcnt(0)` in front
// of the first ds_read_b64_tr_b16 *intrinsic* (it cannot tell that the read and the DMA touch
// different stages), i.e. every stage waited for the NEXT stage's DMA before its first MFMA -- the
// double buffer hid nothing (MFMA busy 33 %).  The asm reads are invisible to that pass; their
// completion is waited for by hand (tr_wait) before the MFMAs that consume them.
__device__ __forceinline__ half8 tr_pair(unsigned lds_addr, int off_bytes) {
  short4v lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(lds_addr), "i"(off_bytes));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(lds_addr), "i"(off_bytes + 128));
  typedef short short8v __attribute__((ext_vector_type(8)));
  const short8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(half8, v);
}
// 4 consecutive pixels of one channel
__device__ __forceinline__ short4v tr_quad(unsigned lds_addr, int off_bytes) {
  short4v v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "i"(off_bytes));
  return v;
}
// all LDS reads issued so far have returned; the operands are tied to the statement so that no
// instruction consuming them can be scheduled above it
__device__ __forceinline__ void tr_wait(half8 (&a)[2], short4v (&x)[3]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
}

// 8 waves: wave (wo, wc) owns 64 output x 32 input channels x 3 taps = 6 accumulator tiles (round 2: 32 x 64 --
// the tap shifts are built in registers from the X operand, 4 v_perm + 4 v_mov per X tile and row, and VALU
// issue is time the MFMAs of the SIMD do not get (tools/coissue_probe.hip): one X tile per wave halves it).  With
// 4 waves of twice that (one wave per SIMD, 192 accumulator registers) nothing covered the issue
// time of the LDS-DMA instructions -- 18 per wave and stage at 100-185 cycles each beside 96 MFMAs
// of 32: removing the DMA made the kernel 1.7x faster.  Two waves per SIMD do.
//
// PW = true: the same machinery for a POINTWISE layer's filter gradient dW[m][c] = sum dY[m] X[c]
// (the backbones' 1x1 convolutions, gemm_f16.hip): one filter row (ky = 1), of which only the
// centre tap is multiplied and stored -- 2 of the 6 accumulator tiles, part[split][m][c].
constexpr int kWThreads = 512;
template <bool PW>
__global__ __launch_bounds__(kWThreads) void conv3x3_wgrad_f16_kernel(const F16Wgrad p) {
  extern __shared__ uint4 lds[];                           // 2 stages
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave & 1, wc = wave >> 1;
  // Work item v = (split, filter row, block pair), split-major.  The items of one split read the SAME
  // pixel stages (X by every output block and filter row, dY by every input block); workgroups b,
  // b + 8, ... share an XCD and its L2, so each XCD takes runs of `xcd_group` consecutive items
  // (round-3 counters: hit rate 44-52 %, 2.6x the operand bytes leaving L2 with round-robin ids).
  int v = (int)blockIdx.x;
  {
    const int G = (int)gridDim.x, xg = p.xcd_group;
    if (xg > 1 && (G & 7) == 0 && ((G >> 3) % xg) == 0) {
      const int r = v >> 3, x = v & 7;
      v = (r / xg) * (8 * xg) + x * xg + (r % xg);
    }
  }
  const int per_split = p.blocks * (PW ? 1 : 3);
  const int split = v / per_split, rest = v - split * per_split;
  const int ky = PW ? 1 : rest / p.blocks;
  const int bp = PW ? rest : rest - ky * p.blocks;
  const int ocb = (bp / p.cblocks) * W_OT, ccb = (bp % p.cblocks) * (W_CT / 8);
  const int CB = (p.C + 7) >> 3, MB = (p.M + 7) >> 3;
  // this workgroup's share of the pixel stages
  const int per = (p.stages + p.splits - 1) / p.splits;
  const int s0 = split * per, s1 = min(s0 + per, p.stages);

  constexpr unsigned kOob = 0x80000000u;
  // lane's place inside a 64-slot DMA piece: slot = 2 * pixel + (block & 1)
  const int lpix = lane >> 1, lodd = lane & 1;
  // A stage's DMA: 5 X pieces + 4 dY pieces per wave.  fetch_setup computes the nine lane offsets
  // (and the level's descriptors); fetch_piece issues ONE piece -- the loop issues a piece per row
  // of MFMAs, so that the ~150-cycle issue of an LDS-DMA instruction on one wave runs under the
  // MFMAs of the other wave of the SIMD instead of both waves issuing their nine at once.
  constexpr int kPieces = X_PIECES / 8 + Y_PIECES / 8;     // 9
  // Offsets of a stage's nine pieces.  A stage whose X window and dY tile lie inside the image ("interior": most of
  // them) differs from any other interior stage of its level only by a wave-uniform displacement -- image, first
  // row, first column -- which rides in the DMA's SCALAR offset; the lane parts (which pixel of the tile, which
  // channel block) are computed once per level (lconst).  Only border stages build per-lane offsets with their
  // image-edge masks.  (Round 2 did that for every stage: ~225 VALU instructions per wave and stage beside 48 MFMAs
  // -- removing the next stage's fetch made the kernel 1.36x faster, and it was this arithmetic, not the DMA:
  // issuing the nine instructions earlier or later moved nothing.)
  unsigned poff[kPieces], lconst[kPieces];
  int cur_lv = -1, soffx = 0, soffy = 0;
  __amdgpu_buffer_rsrc_t fxrs, fyrs;
  auto fetch_setup = [&](int s) {
    int lv = 0;
    for (int l = 1; l < p.n_levels; ++l)
      if (s >= p.stage0[l]) lv = l;
    const int H = p.H[lv], W = p.W[lv], plane = H * W;
    const int seg_x = (W + WPX - 1) / WPX, seg_y = (H + WR - 1) / WR;
    if (lv != cur_lv) {
      cur_lv = lv;
      fxrs = ssad_dev::uniform_rsrc(p.x[lv], (unsigned)((long long)p.N[lv] * CB * plane * 16));
      fyrs = ssad_dev::uniform_rsrc(p.dy[lv], (unsigned)((long long)p.N[lv] * MB * plane * 16));
#pragma unroll
      for (int i = 0; i < X_PIECES / 8; ++i) {
        const int piece = i * 8 + wave;
        const int pair = piece / (X_PAIR / 64), q = piece % (X_PAIR / 64);
        const int pix = q * 32 + lpix;
        const int cb = ccb + pair * 2 + lodd;
        lconst[i] = cb < CB ? (unsigned)((cb * plane + (pix / XPW) * W + pix % XPW) * 16) : kOob;
      }
#pragma unroll
      for (int i = 0; i < Y_PIECES / 8; ++i) {
        const int piece = i * 8 + wave;
        const int pair = piece / (Y_PAIR / 64), q = piece % (Y_PAIR / 64);
        const int pix = q * 32 + lpix;
        const int mb = (ocb >> 3) + pair * 2 + lodd;
        lconst[X_PIECES / 8 + i] = mb < MB ? (unsigned)((mb * plane + (pix / WPX) * W + pix % WPX) * 16) : kOob;
      }
    }
    int t = s - p.stage0[lv];
    const int sx = t % seg_x; t /= seg_x;
    const int sy = t % seg_y;
    const int n = t / seg_y;
    const int y0 = sy * WR, x0 = sx * WPX;
    // PW: no halo -- the X tile starts AT x0, so that the one tap of a pointwise layer is the unshifted
    // operand (kx = 0: registers as they come from the transpose reads, no v_alignbit / v_mov in the loop)
    const int gy0 = y0 + ky - 1, gx0 = x0 - (PW ? 0 : 1);
    // the X window's used columns: 18 (16 + halo) of the 20 fetched; the other two may hold anything
    const bool interior = gy0 >= 0 && gy0 + WR <= H && gx0 >= 0 && gx0 + (PW ? WPX : WPX + 2) <= W &&
                          y0 + WR <= H && x0 + WPX <= W;
    if (interior) {
      soffx = __builtin_amdgcn_readfirstlane((n * CB * plane + gy0 * W + gx0) * 16);
      soffy = __builtin_amdgcn_readfirstlane((n * MB * plane + y0 * W + x0) * 16);
#pragma unroll
      for (int k = 0; k < kPieces; ++k) poff[k] = lconst[k];
      return;
    }
    soffx = soffy = 0;
#pragma unroll
    for (int i = 0; i < X_PIECES / 8; ++i) {
      const int piece = i * 8 + wave;                       // pair = piece / 5, 32 pixels each
      const int pair = piece / (X_PAIR / 64), q = piece % (X_PAIR / 64);
      const int pix = q * 32 + lpix;                        // row * 20 + column
      const int gy = gy0 + pix / XPW, gx = gx0 + pix % XPW;
      const int cb = ccb + pair * 2 + lodd;
      unsigned off = kOob;
      if (cb < CB && gy >= 0 && gy < H && gx >= 0 && gx < W)
        off = (unsigned)(((n * CB + cb) * plane + gy * W + gx) * 16);
      poff[i] = off;
    }
#pragma unroll
    for (int i = 0; i < Y_PIECES / 8; ++i) {
      const int piece = i * 8 + wave;
      const int pair = piece / (Y_PAIR / 64), q = piece % (Y_PAIR / 64);
      const int pix = q * 32 + lpix;                        // row * 16 + column
      const int gy = y0 + pix / WPX, gx = x0 + pix % WPX;
      const int mb = (ocb >> 3) + pair * 2 + lodd;
      unsigned off = kOob;
      if (mb < MB && gy < H && gx < W) off = (unsigned)(((n * MB + mb) * plane + gy * W + gx) * 16);
      poff[X_PIECES / 8 + i] = off;
    }
  };
  auto fetch_piece = [&](int k, int buf) {
    auto* dst = (__attribute__((address_space(3))) uint4*)lds + buf * W_STAGE;
    if (k < X_PIECES / 8) {
      const int piece = k * 8 + wave;
      const int pair = piece / (X_PAIR / 64), q = piece % (X_PAIR / 64);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          fxrs, (__attribute__((address_space(3))) void*)(dst + pair * X_PITCH + q * 64), 16, poff[k], soffx, 0, 0);
    } else {
      const int piece = (k - X_PIECES / 8) * 8 + wave;
      const int pair = piece / (Y_PAIR / 64), q = piece % (Y_PAIR / 64);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          fyrs, (__attribute__((address_space(3))) void*)(dst + Y_BASE + pair * Y_PITCH + q * 64), 16, poff[k], soffy,
          0, 0);
    }
  };
  auto fetch = [&](int s, int buf) {
    fetch_setup(s);
#pragma unroll
    for (int k = 0; k < kPieces; ++k) fetch_piece(k, buf);
  };

  // ---- transpose-read addressing (tools/tr16_probe.hip): lane l of a 16-lane group addresses
  // pixel (l & 15) >> 2, channel quad l & 3 of its group's 16 channels = one block pair
  const int g = lane >> 4, i16 = lane & 15;
  const int quad = i16 & 3, pj = i16 >> 2;
  const int kpx = 8 * (g >> 1) + pj;                       // pixel within the 16-pixel K-step
  const int in_pair = (quad >> 1) * 8 + (quad & 1) * 4;    // halves: block of the pair, half slot
  // 32-channel MFMA tile = 2 pairs; group g & 1 takes the second
  const int xb_base = (((wc * 2 + (g & 1)) * X_PITCH + kpx * 2) * 8) + in_pair;
  const int ya_base = ((Y_BASE + (wo * 4 + (g & 1)) * Y_PITCH + kpx * 2) * 8) + in_pair;   // + tile * 2 pairs

  float16v acc[2][3];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][kx][r] = 0.0f;

  if (s0 < s1) fetch(s0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int s = s0; s < s1; ++s) {
    const int buf = (s - s0) & 1;
    const bool more = s + 1 < s1 && !(F16_ABLATE & 8);
    if (more) fetch_setup(s + 1);
    const unsigned stage = (unsigned)(uintptr_t)((__attribute__((address_space(3))) uint4*)lds + buf * W_STAGE);
    const unsigned xa = stage + xb_base * 2, ya = stage + ya_base * 2;        // byte addresses
    // operands of row r + 1 are fetched while the MFMAs of row r run (the transpose reads'
    // latency is otherwise exposed once per row: one wave per SIMD, nothing else to issue)
    // X operands of the three taps kx of a filter row are the same 8 pixels shifted by 0 / 1 / 2:
    // a lane reads pixels 8h .. 8h + 11 of its channel ONCE (three transpose reads) and builds the
    // shifted operands in registers -- kx = 2 is a register selection, kx = 1 four v_alignbit_b32 --
    // instead of reading each shift from LDS (12 of the 16 read pairs per row; the kernel was
    // bound by the LDS read rate: 64 KB per CU per 384 MFMA cycles).
    half8 a[2][2];                        // [buffer][output tile u]
    short4v xr[2][3];                     // [buffer][lo, hi, next]
    auto load_row = [&](int row, half8 (&aa)[2], short4v (&xx)[3]) {
#pragma unroll
      for (int u = 0; u < 2; ++u) aa[u] = tr_pair(ya, (u * 2 * Y_PITCH + row * WPX * 2) * 16);
#pragma unroll
      for (int k = 0; k < (PW ? 2 : 3); ++k) xx[k] = tr_quad(xa, (row * XPW * 2) * 16 + k * 128);
    };
    load_row(0, a[0], xr[0]);
#pragma unroll
    for (int row = 0; row < WR; ++row) {
      tr_wait(a[row & 1], xr[row & 1]);
      half8 b[3];
      {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x2 lo = __builtin_bit_cast(u32x2, xr[row & 1][0]), hi = __builtin_bit_cast(u32x2, xr[row & 1][1]),
                    nx = __builtin_bit_cast(u32x2, xr[row & 1][2]);
        // dwords d0..d4 = pixels (0,1) (2,3) (4,5) (6,7) (8,9) of this lane's 8-pixel half
        b[0] = __builtin_bit_cast(half8, u32x4{lo.x, lo.y, hi.x, hi.y});
        b[1] = __builtin_bit_cast(half8, u32x4{__builtin_amdgcn_alignbit(lo.y, lo.x, 16),
                                               __builtin_amdgcn_alignbit(hi.x, lo.y, 16),
                                               __builtin_amdgcn_alignbit(hi.y, hi.x, 16),
                                               __builtin_amdgcn_alignbit(nx.x, hi.y, 16)});
        b[2] = __builtin_bit_cast(half8, u32x4{lo.y, hi.x, hi.y, nx.x});
      }
      if (row + 1 < WR && !(F16_ABLATE & 16)) load_row(row + 1, a[(row + 1) & 1], xr[(row + 1) & 1]);
      if (more) {                                            // the next stage's DMA, a piece per row
        if (!(F16_ABLATE & 32) || row < X_PIECES / 8) fetch_piece(row, buf ^ 1);
        if (row == 0 && !(F16_ABLATE & 32)) fetch_piece(WR, buf ^ 1);
      }
      __builtin_amdgcn_sched_barrier(0);                     // the next row's reads go out first
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int kx = 0; kx < (PW ? 1 : 3); ++kx)
          acc[u][kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[row & 1][u], b[kx], acc[u][kx], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);                     // bound the operands in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- partial sums: part[split][tap][m][c], c contiguous across lanes
  const int h = lane >> 5;
  const int c = ccb * 8 + wc * 32 + (lane & 31);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (c >= p.C) continue;
#pragma unroll
    for (int kx = 0; kx < (PW ? 1 : 3); ++kx)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = ocb + wo * 64 + u * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < p.M) {
          if (PW) p.part[((long long)split * p.M + m) * p.C + c] = acc[u][kx][r];
          else p.part[(((long long)split * 9 + ky * 3 + kx) * p.M + m) * p.C + c] = acc[u][kx][r];
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------
// All NINE taps in one workgroup (round 3, end): the kernel above moves 74 KB from L2 to LDS per 12.6 MFLOP
// (170 flop per byte) and that operand stream, not latency, bounds it.  Here a workgroup owns
// 128 output x 64 input channels x 9 taps: a stage is 8 rows x 16 pixels of dY and the 10 x 20 X window around
// it (58 KB per 18.9 MFLOP = 325 flop per byte); wave (wo, wc) = 32 x 32 channels x 9 taps = 9 accumulator tiles.
// A lane keeps the three X rows a dY row meets, each in its three column shifts (9 operands, 36 registers), as a
// rolling window: per dY row ONE new X row is read (three transpose reads) and shifted (4 v_perm + 4 v_mov), one
// dY operand is read, nine MFMAs are issued.  Partial sums: part[split][tap][M][C] as above, same reduce kernel.
// ---------------------------------------------------------------------------------------------------
constexpr int N9_OT = 128, N9_CT = 64;
constexpr int X9_ROWS = WR + 2;
constexpr int X9_USED = X9_ROWS * XPW;                        // 200 pixels
constexpr int X9_PAIR = (X9_USED * 2 + 63) / 64 * 64;         // 448 slots = 7 DMA pieces per block pair
constexpr int X9_PITCH = X9_PAIR + 8;
constexpr int X9_PAIRS = N9_CT / 16, Y9_PAIRS = N9_OT / 16;   // 4, 8
constexpr int Y9_BASE = X9_PAIRS * X9_PITCH;
constexpr int W9_STAGE = Y9_BASE + Y9_PAIRS * Y_PITCH;        // 3936 slots = 61.5 KiB
constexpr int X9_PIECES = X9_PAIRS * (X9_PAIR / 64);          // 28
constexpr int Y9_PIECES = Y9_PAIRS * (Y_PAIR / 64);           // 32
constexpr int P9 = 8;                                         // pieces per wave and stage (60 of 64 used)
static_assert(X9_PIECES + Y9_PIECES <= 8 * P9 && P9 == WR, "one DMA piece per wave and row");

__device__ __forceinline__ void tr_wait9(half8& a, short4v (&x)[3]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
}
__device__ __forceinline__ void tr_wait9x(short4v (&x)[3], short4v (&y)[3]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]));
}

__global__ __launch_bounds__(kWThreads) void wgrad9_f16_kernel(const F16Wgrad p) {
  extern __shared__ uint4 lds[];                           // 2 stages
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave & 3, wc = wave >> 2;
  int v = (int)blockIdx.x;
  {
    const int G = (int)gridDim.x, xg = p.xcd_group;
    if (xg > 1 && (G & 7) == 0 && ((G >> 3) % xg) == 0) {
      const int r = v >> 3, x = v & 7;
      v = (r / xg) * (8 * xg) + x * xg + (r % xg);
    }
  }
  const int split = v / p.blocks, bp = v - split * p.blocks;
  const int ocb = (bp / p.cblocks) * N9_OT, ccb = (bp % p.cblocks) * (N9_CT / 8);
  const int CB = (p.C + 7) >> 3, MB = (p.M + 7) >> 3;
  const int per = (p.stages + p.splits - 1) / p.splits;
  const int s0 = split * per, s1 = min(s0 + per, p.stages);

  constexpr unsigned kOob = 0x80000000u;
  const int lpix = lane >> 1, lodd = lane & 1;
  // piece k of this wave: index k * 8 + wave into [28 X pieces | 32 dY pieces | 4 unused]
  unsigned poff[P9], lconst[P9];
  int cur_lv = -1, soffx = 0, soffy = 0;
  __amdgpu_buffer_rsrc_t fxrs, fyrs;
  auto lane_const = [&](int k, int plane, int W) -> unsigned {
    const int pc = k * 8 + wave;
    if (pc < X9_PIECES) {
      const int pair = pc / (X9_PAIR / 64), q = pc % (X9_PAIR / 64);
      const int pix = q * 32 + lpix;
      const int cb = ccb + pair * 2 + lodd;
      return (cb < CB && pix < X9_USED) ? (unsigned)((cb * plane + (pix / XPW) * W + pix % XPW) * 16) : kOob;
    }
    if (pc < X9_PIECES + Y9_PIECES) {
      const int py = pc - X9_PIECES;
      const int pair = py / (Y_PAIR / 64), q = py % (Y_PAIR / 64);
      const int pix = q * 32 + lpix;
      const int mb = (ocb >> 3) + pair * 2 + lodd;
      return mb < MB ? (unsigned)((mb * plane + (pix / WPX) * W + pix % WPX) * 16) : kOob;
    }
    return kOob;
  };
  auto fetch_setup = [&](int s) {
    int lv = 0;
    for (int l = 1; l < p.n_levels; ++l)
      if (s >= p.stage0[l]) lv = l;
    const int H = p.H[lv], W = p.W[lv], plane = H * W;
    const int seg_x = (W + WPX - 1) / WPX, seg_y = (H + WR - 1) / WR;
    if (lv != cur_lv) {
      cur_lv = lv;
      fxrs = ssad_dev::uniform_rsrc(p.x[lv], (unsigned)((long long)p.N[lv] * CB * plane * 16));
      fyrs = ssad_dev::uniform_rsrc(p.dy[lv], (unsigned)((long long)p.N[lv] * MB * plane * 16));
#pragma unroll
      for (int k = 0; k < P9; ++k) lconst[k] = lane_const(k, plane, W);
    }
    int t = s - p.stage0[lv];
    const int sx = t % seg_x; t /= seg_x;
    const int sy = t % seg_y;
    const int n = t / seg_y;
    const int y0 = sy * WR, x0 = sx * WPX;
    const int gy0 = y0 - 1, gx0 = x0 - 1;
    const bool interior = gy0 >= 0 && gy0 + X9_ROWS <= H && gx0 >= 0 && gx0 + WPX + 2 <= W && y0 + WR <= H &&
                          x0 + WPX <= W;
    if (interior) {
      soffx = __builtin_amdgcn_readfirstlane((n * CB * plane + gy0 * W + gx0) * 16);
      soffy = __builtin_amdgcn_readfirstlane((n * MB * plane + y0 * W + x0) * 16);
#pragma unroll
      for (int k = 0; k < P9; ++k) poff[k] = lconst[k];
      return;
    }
    soffx = soffy = 0;
#pragma unroll
    for (int k = 0; k < P9; ++k) {
      const int pc = k * 8 + wave;
      unsigned off = kOob;
      if (pc < X9_PIECES) {
        const int pair = pc / (X9_PAIR / 64), q = pc % (X9_PAIR / 64);
        const int pix = q * 32 + lpix;
        const int gy = gy0 + pix / XPW, gx = gx0 + pix % XPW;
        const int cb = ccb + pair * 2 + lodd;
        if (cb < CB && pix < X9_USED && gy >= 0 && gy < H && gx >= 0 && gx < W)
          off = (unsigned)(((n * CB + cb) * plane + gy * W + gx) * 16);
      } else if (pc < X9_PIECES + Y9_PIECES) {
        const int py = pc - X9_PIECES;
        const int pair = py / (Y_PAIR / 64), q = py % (Y_PAIR / 64);
        const int pix = q * 32 + lpix;
        const int gy = y0 + pix / WPX, gx = x0 + pix % WPX;
        const int mb = (ocb >> 3) + pair * 2 + lodd;
        if (mb < MB && gy < H && gx < W) off = (unsigned)(((n * MB + mb) * plane + gy * W + gx) * 16);
      }
      poff[k] = off;
    }
  };
  auto fetch_piece = [&](int k, int buf) {
    auto* dst = (__attribute__((address_space(3))) uint4*)lds + buf * W9_STAGE;
    const int pc = k * 8 + wave;
    if (pc < X9_PIECES) {
      const int pair = pc / (X9_PAIR / 64), q = pc % (X9_PAIR / 64);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          fxrs, (__attribute__((address_space(3))) void*)(dst + pair * X9_PITCH + q * 64), 16, poff[k], soffx, 0, 0);
    } else if (pc < X9_PIECES + Y9_PIECES) {
      const int py = pc - X9_PIECES;
      const int pair = py / (Y_PAIR / 64), q = py % (Y_PAIR / 64);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          fyrs, (__attribute__((address_space(3))) void*)(dst + Y9_BASE + pair * Y_PITCH + q * 64), 16, poff[k], soffy,
          0, 0);
    }
  };

  // transpose-read addressing as in the kernel above
  const int g = lane >> 4, i16 = lane & 15;
  const int quad = i16 & 3, pj = i16 >> 2;
  const int kpx = 8 * (g >> 1) + pj;
  const int in_pair = (quad >> 1) * 8 + (quad & 1) * 4;
  const int xb_base = (((wc * 2 + (g & 1)) * X9_PITCH + kpx * 2) * 8) + in_pair;
  const int ya_base = ((Y9_BASE + (wo * 2 + (g & 1)) * Y_PITCH + kpx * 2) * 8) + in_pair;

  float16v acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  if (s0 < s1) {
    fetch_setup(s0);
#pragma unroll
    for (int k = 0; k < P9; ++k) fetch_piece(k, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int s = s0; s < s1; ++s) {
    const int buf = (s - s0) & 1;
    const bool more = s + 1 < s1;
    if (more) fetch_setup(s + 1);
    const unsigned stage = (unsigned)(uintptr_t)((__attribute__((address_space(3))) uint4*)lds + buf * W9_STAGE);
    const unsigned xa = stage + xb_base * 2, ya = stage + ya_base * 2;        // byte addresses
    // a window row's three column shifts from its 12 pixels (lo, hi, next): dwords = pixel pairs
    auto shifts = [&](const short4v (&xx)[3], half8 (&b)[3]) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const u32x2 lo = __builtin_bit_cast(u32x2, xx[0]), hi = __builtin_bit_cast(u32x2, xx[1]),
                  nx = __builtin_bit_cast(u32x2, xx[2]);
      b[0] = __builtin_bit_cast(half8, u32x4{lo.x, lo.y, hi.x, hi.y});
      b[1] = __builtin_bit_cast(half8, u32x4{__builtin_amdgcn_alignbit(lo.y, lo.x, 16),
                                             __builtin_amdgcn_alignbit(hi.x, lo.y, 16),
                                             __builtin_amdgcn_alignbit(hi.y, hi.x, 16),
                                             __builtin_amdgcn_alignbit(nx.x, hi.y, 16)});
      b[2] = __builtin_bit_cast(half8, u32x4{lo.y, hi.x, hi.y, nx.x});
    };
    auto read_x = [&](int xrow, short4v (&xx)[3]) {
#pragma unroll
      for (int k = 0; k < 3; ++k) xx[k] = tr_quad(xa, (xrow * XPW * 2) * 16 + k * 128);
    };
    half8 bw[3][3];                       // window row (X row index % 3) x column shift
    half8 a[2];
    short4v xr[2][3];
    {
      short4v x0r[3], x1r[3];
      read_x(0, x0r);
      read_x(1, x1r);
      tr_wait9x(x0r, x1r);
      shifts(x0r, bw[0]);
      shifts(x1r, bw[1]);
    }
    a[0] = tr_pair(ya, 0);
    read_x(2, xr[0]);
#pragma unroll
    for (int row = 0; row < WR; ++row) {
      tr_wait9(a[row & 1], xr[row & 1]);
      shifts(xr[row & 1], bw[(row + 2) % 3]);
      if (row + 1 < WR) {
        a[(row + 1) & 1] = tr_pair(ya, ((row + 1) * WPX * 2) * 16);
        read_x(row + 3, xr[(row + 1) & 1]);
      }
      if (more) fetch_piece(row, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[row & 1], bw[(row + ky) % 3][kx],
                                                                    acc[ky * 3 + kx], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  const int h = lane >> 5;
  const int c = ccb * 8 + wc * 32 + (lane & 31);
  if (c < p.C) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = ocb + wo * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < p.M) p.part[(((long long)split * 9 + t) * p.M + m) * p.C + c] = acc[t][r];
      }
  }
}

// dW[m][c][tap] (+)= scale * sum_split part[split][tap][m][c], splits summed in a fixed order.
// Workgroup = one output channel m x 64 input channels: thread (tap, c) sums its element over the
// splits (four independent chains; 256-byte runs along c), the [c][9] tile is transposed through
// LDS and leaves as ONE contiguous 2304-byte run of dW -- with a thread per element writing
// dW[(m C + c) 9 + tap] directly every wave store touched 36 cache lines (88 us per tower layer,
// more than a third of the whole filter-gradient call).  Row m == M folds the bias partials:
// db[m] (+)= scale * sum dbpart[k][m].
constexpr int kDbSplits = 64;
// db[m] (+)= scale * sum_k dbpart[k][m] in a fixed order.  64 outputs per pass of a workgroup, the k range dealt to
// 4 thread groups x 4 independent chains: one thread summing its output's n_levels x 64 partials alone was a chain
// of 320 dependent loads -- 80 us, the whole duration of the reduce launch (the filter rows take 20).
__device__ __forceinline__ void bias_row_reduce(const float* __restrict__ dbpart, int dbparts, int M, float scale,
                                                int accumulate, float* __restrict__ db, float* red) {
  const int Mp = (M + 7) & ~7;
  const int ml = threadIdx.x & 63, j = threadIdx.x >> 6;           // kThreads = 256: j = 0..3
  for (int m0 = blockIdx.x * 64; m0 < M; m0 += gridDim.x * 64) {
    const int mm = m0 + ml;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (mm < M) {
      int k = j;
      for (; k + 12 < dbparts; k += 16) {
        s0 += dbpart[k * Mp + mm];
        s1 += dbpart[(k + 4) * Mp + mm];
        s2 += dbpart[(k + 8) * Mp + mm];
        s3 += dbpart[(k + 12) * Mp + mm];
      }
      for (; k < dbparts; k += 4) s0 += dbpart[k * Mp + mm];
    }
    red[j * 64 + ml] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (j == 0 && mm < M) {
      const float s = ((red[ml] + red[64 + ml]) + (red[128 + ml] + red[192 + ml])) * scale;
      db[mm] = accumulate ? db[mm] + s : s;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kThreads) void f16_wgrad_reduce_kernel(const float* __restrict__ part,
                                                                    int splits, int M, int C,
                                                                    int accumulate, float scale,
                                                                    const float* __restrict__ scale_dev,
                                                                    float* __restrict__ dw,
                                                                    const float* __restrict__ dbpart,
                                                                    int dbparts, float* __restrict__ db) {
  __shared__ float o[64 * 9];
  if (scale_dev) scale *= scale_dev[0];
  const int m = blockIdx.y, c0 = blockIdx.x * 64;
  if (m == M) {                                            // the bias row
    if (!db) return;
    bias_row_reduce(dbpart, dbparts, M, scale, accumulate, db, o);
    return;
  }
  const long long total = 9LL * M * C;
  for (int e = threadIdx.x; e < 64 * 9; e += kThreads) {
    const int tap = e >> 6, cl = e & 63, c = c0 + cl;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (c < C) {
      const float* q = part + ((long long)tap * M + m) * C + c;
      int k = 0;
      for (; k + 3 < splits; k += 4) {
        s0 += q[(long long)k * total];
        s1 += q[(long long)(k + 1) * total];
        s2 += q[(long long)(k + 2) * total];
        s3 += q[(long long)(k + 3) * total];
      }
      for (; k < splits; ++k) s0 += q[(long long)k * total];
    }
    o[cl * 9 + tap] = ((s0 + s1) + (s2 + s3)) * scale;
  }
  __syncthreads();
  const int nvalid = (C - c0 < 64 ? C - c0 : 64) * 9;
  float* dst = dw + ((long long)m * C + c0) * 9;
  for (int e = threadIdx.x; e < nvalid; e += kThreads) dst[e] = accumulate ? dst[e] + o[e] : o[e];
}

// pointwise layers: dW[m][c] (+)= scale * sum_split part[split][m][c] (fixed order); row m == M
// folds the bias partials as above.
__global__ __launch_bounds__(kThreads) void f16_wgrad_reduce_pw_kernel(const float* __restrict__ part, int splits,
                                                                       int M, int C, int accumulate, float scale,
                                                                       const float* __restrict__ scale_dev,
                                                                       float* __restrict__ dw,
                                                                       const float* __restrict__ dbpart, int dbparts,
                                                                       float* __restrict__ db) {
  if (scale_dev) scale *= scale_dev[0];
  const long long total = (long long)M * C;
  if (blockIdx.y == 1) {                                   // the bias row
    if (!db) return;
    __shared__ float red[kThreads];
    bias_row_reduce(dbpart, dbparts, M, scale, accumulate, db, red);
    return;
  }
  for (long long e = (long long)blockIdx.x * kThreads + threadIdx.x; e < total; e += (long long)gridDim.x * kThreads) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // eight independent chains, fixed order
    int k = 0;
    for (; k + 7 < splits; k += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += part[(long long)(k + u) * total + e];
    }
    for (; k < splits; ++k) a[k & 7] += part[(long long)k * total + e];
    const float v = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) * scale;
    dw[e] = accumulate ? dw[e] + v : v;
  }
}

// dbpart[level][split][m] = sum of the blocked fp16 dY over the split's share of the level's
// N * H * W pixels (grid: channel blocks x kDbSplits x levels -- one launch for all levels)
__global__ __launch_bounds__(kThreads) void f16_bias_grad_kernel(const F16Wgrad p, float* __restrict__ dbpart) {
  __shared__ float red[kThreads / 64][8];
  const int lv = blockIdx.z;
  const uint4* __restrict__ dy = p.dy[lv];
  const int N = p.N[lv], M = p.M, plane = p.H[lv] * p.W[lv];
  const int MB = (M + 7) >> 3, mb = blockIdx.x;
  const long long total = (long long)N * plane;
  const long long per = (total + kDbSplits - 1) / kDbSplits;
  const long long p0 = blockIdx.y * per, p1 = p0 + per < total ? p0 + per : total;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long long i = p0 + threadIdx.x; i < p1; i += kThreads) {
    const long long n = i / plane, px = i % plane;
    const half8 v = as_half8(dy[(n * MB + mb) * plane + px]);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = s[e];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][e] = v;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    float v = 0.0f;
    for (int w = 0; w < kThreads / 64; ++w) v += red[w][threadIdx.x];
    dbpart[((size_t)lv * kDbSplits + blockIdx.y) * (MB * 8) + mb * 8 + threadIdx.x] = v;
  }
}

namespace {
bool use_wgrad9() {
  constexpr bool on = true;
  return on;
}
int wgrad_blocks(int C, int M, bool pw) {
  if (!pw && use_wgrad9()) return ((M + N9_OT - 1) / N9_OT) * ((C + N9_CT - 1) / N9_CT);
  return ((M + W_OT - 1) / W_OT) * ((C + W_CT - 1) / W_CT);
}
int wgrad_rows(bool pw) { return (pw || use_wgrad9()) ? 1 : 3; }
int wgrad_splits(int blocks, int stages, int rows = 3) {
  // one workgroup per CU at a time (148 KiB of LDS): whole rounds only -- 516 workgroups on 256
  // CUs take three rounds where 504 take two -- and ONE round measures best (tower layer, all
  // levels: 0.424 / 0.449 / 0.485 / 0.523 ms for 1 / 2 / 3 / 4 rounds: the per-workgroup prologue
  // and the 2.4 MB of partial sums per split are not hidden at this occupancy)
  const int cus = ssad_cu_count();
  constexpr int rounds = 1;
  int s = (rounds * cus) / (rows * blocks);    // whole rounds; x 3 filter rows (1 for a pointwise layer)
  if (s > stages) s = stages;
  return s < 1 ? 1 : s;
}
}  // namespace

extern "C" {

namespace {
int wgrad_stages(const ssad_f16_wgrad_level* levels, int n_levels) {
  long long st = 0;
  for (int l = 0; l < n_levels; ++l)
    st += (long long)levels[l].N * ((levels[l].H + WR - 1) / WR) * ((levels[l].W + WPX - 1) / WPX);
  return st > 0x7fffffff ? 0x7fffffff : (int)st;
}
}  // namespace

namespace {
size_t wgrad_ws_bytes(const ssad_f16_wgrad_level* levels, int n_levels, int C, int M, bool pw) {
  if (!levels || n_levels < 1) return 0;
  const int blocks = wgrad_blocks(C, M, pw);
  const int stages = wgrad_stages(levels, n_levels);
  return ((size_t)wgrad_splits(blocks, stages > 0 ? stages : 1, wgrad_rows(pw)) * (pw ? 1 : 9) * (size_t)M * (size_t)C +
          (size_t)n_levels * kDbSplits * (size_t)((M + 7) & ~7)) * sizeof(float);
}

int wgrad_launch(const ssad_f16_wgrad_level* levels, int n_levels, int C, int M, int accumulate, float scale,
                 const float* scale_dev, float* dw, float* db, void* workspace, size_t workspace_bytes,
                 ssad_stream_t stream, bool pw) {
  if (!levels || n_levels < 1 || n_levels > SSAD_MAX_F16_LEVELS || !dw || C < 1 || M < 1) return SSAD_E_BADARG;
  F16Wgrad p;
  long long stages = 0;
  for (int l = 0; l < n_levels; ++l) {
    const ssad_f16_wgrad_level& L = levels[l];
    if (!L.x || !L.dy || L.N < 0 || L.H < 1 || L.W < 1) return SSAD_E_BADARG;
    if ((long long)L.N * (((C > M ? C : M) + 7) / 8) * L.H * L.W >= (1LL << 31)) return SSAD_E_BADARG;
    p.x[l] = static_cast<const uint4*>(L.x);
    p.dy[l] = static_cast<const uint4*>(L.dy);
    p.N[l] = L.N; p.H[l] = L.H; p.W[l] = L.W;
    p.stage0[l] = (int)stages;
    stages += (long long)L.N * ((L.H + WR - 1) / WR) * ((L.W + WPX - 1) / WPX);
    if (stages >= (1LL << 31)) return SSAD_E_BADARG;
  }
  for (int l = n_levels; l <= SSAD_MAX_F16_LEVELS; ++l) p.stage0[l] = (int)stages;
  if (workspace_bytes < wgrad_ws_bytes(levels, n_levels, C, M, pw) || !workspace) return SSAD_E_WORKSPACE;
  p.n_levels = n_levels;
  p.part = static_cast<float*>(workspace);
  p.C = C; p.M = M;
  p.stages = (int)stages;
  const bool nine = !pw && use_wgrad9();
  p.cblocks = nine ? (C + N9_CT - 1) / N9_CT : (C + W_CT - 1) / W_CT;
  const int blocks = wgrad_blocks(C, M, pw);
  const int splits = wgrad_splits(blocks, p.stages > 0 ? p.stages : 1, wgrad_rows(pw));
  hipStream_t s = (hipStream_t)stream;
  if (p.stages > 0) {
    static const bool attr = [] {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_f16_kernel<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W_STAGE * 16) == hipSuccess &&
             hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_f16_kernel<true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W_STAGE * 16) == hipSuccess &&
             hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad9_f16_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W9_STAGE * 16) == hipSuccess;
    }();
    if (!attr) return SSAD_E_BADARG;
    p.blocks = blocks; p.splits = splits;
    {
      // runs of one whole split per XCD when that divides evenly, else the largest common run length
      const int per_split = blocks * wgrad_rows(pw), total = per_split * splits;
      int g = 1;
      if ((total & 7) == 0) {
        int x = total >> 3, y = per_split;
        while (y) { const int t = x % y; x = y; y = t; }
        g = x;
      }
      p.xcd_group = g;
      if (nine) hipLaunchKernelGGL(wgrad9_f16_kernel, dim3(total), dim3(kWThreads), 2 * W9_STAGE * 16, s, p);
      else if (pw) hipLaunchKernelGGL(conv3x3_wgrad_f16_kernel<true>, dim3(total), dim3(kWThreads), 2 * W_STAGE * 16, s, p);
      else hipLaunchKernelGGL(conv3x3_wgrad_f16_kernel<false>, dim3(total), dim3(kWThreads), 2 * W_STAGE * 16, s, p);
    }
  }
  float* dbpart = p.part + (size_t)splits * (pw ? 1 : 9) * (size_t)M * (size_t)C;
  if (db) {
    p.n_levels = n_levels;
    hipLaunchKernelGGL(f16_bias_grad_kernel, dim3((M + 7) / 8, kDbSplits, n_levels), dim3(kThreads), 0, s, p, dbpart);
  }
  if (pw) {
    long long gx = ((long long)M * C + kThreads - 1) / kThreads;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(f16_wgrad_reduce_pw_kernel, dim3((unsigned)gx, 2), dim3(kThreads), 0, s, p.part,
                       p.stages > 0 ? splits : 0, M, C, accumulate, scale, scale_dev, dw, dbpart,
                       n_levels * kDbSplits, db);
  } else {
    hipLaunchKernelGGL(f16_wgrad_reduce_kernel, dim3((C + 63) / 64, M + 1), dim3(kThreads), 0, s, p.part,
                       p.stages > 0 ? splits : 0, M, C, accumulate, scale, scale_dev, dw, dbpart,
                       n_levels * kDbSplits, db);
  }
  return (int)hipGetLastError();
}
}  // namespace

size_t ssad_conv3x3_wgrad_f16_levels_workspace_bytes(const ssad_f16_wgrad_level* levels, int n_levels, int C,
                                                     int M) {
  return wgrad_ws_bytes(levels, n_levels, C, M, false);
}

int ssad_conv3x3_wgrad_f16_levels(const ssad_f16_wgrad_level* levels, int n_levels, int C, int M,
                                  int accumulate, float scale, float* dw, float* db, void* workspace,
                                  size_t workspace_bytes, ssad_stream_t stream) {
  return wgrad_launch(levels, n_levels, C, M, accumulate, scale, nullptr, dw, db, workspace, workspace_bytes,
                      stream, false);
}

int ssad_conv3x3_wgrad_f16_levels_dyn(const ssad_f16_wgrad_level* levels, int n_levels, int C, int M,
                                      int accumulate, float scale, const float* scale_dev, float* dw,
                                      float* db, void* workspace, size_t workspace_bytes,
                                      ssad_stream_t stream) {
  return wgrad_launch(levels, n_levels, C, M, accumulate, scale, scale_dev, dw, db, workspace, workspace_bytes,
                      stream, false);
}

/* pointwise layer: dW[M][C] and (optionally) db[M] from blocked fp16 X [N][C/8][H][W][8] and
 * dY [N][M/8][H][W][8] */
size_t ssad_conv1x1_wgrad_f16_workspace_bytes(int N, int C, int H, int W, int M) {
  ssad_f16_wgrad_level L;
  L.x = L.dy = nullptr; L.N = N; L.H = H; L.W = W;
  return wgrad_ws_bytes(&L, 1, C, M, true);
}

int ssad_conv1x1_wgrad_f16(const void* x_blocked, const void* dy_blocked, int N, int C, int H, int W, int M,
                           int accumulate, float scale, const float* scale_dev, float* dw, float* db,
                           void* workspace, size_t workspace_bytes, ssad_stream_t stream) {
  ssad_f16_wgrad_level L;
  L.x = x_blocked; L.dy = dy_blocked; L.N = N; L.H = H; L.W = W;
  return wgrad_launch(&L, 1, C, M, accumulate, scale, scale_dev, dw, db, workspace, workspace_bytes, stream, true);
}

size_t ssad_conv3x3_wgrad_f16_workspace_bytes(int N, int C, int H, int W, int M) {
  ssad_f16_wgrad_level L;
  L.x = L.dy = nullptr; L.N = N; L.H = H; L.W = W;
  return ssad_conv3x3_wgrad_f16_levels_workspace_bytes(&L, 1, C, M);
}

int ssad_conv3x3_wgrad_f16(const void* x_blocked, const void* dy_blocked, int N, int C, int H, int W,
                           int M, int accumulate, float scale, float* dw, float* db, void* workspace,
                           size_t workspace_bytes, ssad_stream_t stream) {
  ssad_f16_wgrad_level L;
  L.x = x_blocked; L.dy = dy_blocked; L.N = N; L.H = H; L.W = W;
  return ssad_conv3x3_wgrad_f16_levels(&L, 1, C, M, accumulate, scale, dw, db, workspace, workspace_bytes,
                                       stream);
}

}  // extern "C"

#ifdef F16_TIMELINE
extern "C" SSAD_API int ssad_f16_dbg_read(void* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_f16_dbg), sizeof(g_f16_dbg));
}
#endif
