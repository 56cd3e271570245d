// gemm_split.hip -- the pointwise (1x1) convolution and its filter gradient as split-operand GEMMs on the fp16 matrix
// pipes (round 6).
//
// Same contracts as ssad_conv1x1_gemm / ssad_conv1x1_wgrad (gemm_conv.hip; detectron/lib/modeling/ResNet.py:221-283,
// FPN.py:116-250; caffe2/operators/conv_op_impl.h:126-173, 451-500 with the identity im2col of a 1x1 kernel):
//     y[n][m][p] = act( sum_k a[k][m] x[n][k][p] + bias[m] + residual[n][m][p] ),  optional mask, optional y +=
//     dw[m][c]  (+)= sum_{n,p} dy[n][m][p] x[n][c][p]
// with the arithmetic of conv3x3_split.hip: every fp32 operand as hi + lo fp16 under a per-tensor power-of-two scale
// from the tensor's measured |max|, three v_mfma_f32_32x32x16_f16 per operand pair, fp32 accumulation, the scales
// divided out exactly in the epilogue.  The activations are split ON THE FLY inside the GEMM kernels: a first version
// with a split PASS in front of a persistent LDS-DMA kernel ran the GEMM itself at 800 TFLOP/s of fp16 products but
// lost more to the pass than it saved (profiles/r06_experiments.md section 3) and was removed.
//
// One forward call = |max| of x and a (one launch) + split of a (tiny) + gemm_fly_kernel.
#include <stdlib.h>

#include <mutex>

#include "split_common.h"

namespace {

using namespace ssad_split;

constexpr int PT = 128;                // pixels per work item
constexpr int MT = 256;                // output channels per work item
constexpr int HDR = 16;                // floats in front of the packed a: [0] = |max| bits

__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

// a[K][lda] fp32 -> [HDR][hi: Wp[K/8][M] x 16 B][lo: same]; slot (kb, m) = the 8 consecutive k of row m
__global__ __launch_bounds__(kThreads) void gsplit_pack_a_kernel(const float* __restrict__ a, int lda, int K, int M,
                                                                 const unsigned* __restrict__ amax, float* __restrict__ dst) {
  const int KB = (K + 7) >> 3;
  const long long slots = (long long)KB * M;
  const float s = pow2f(15 - split_exponent(amax[0]));
  if (blockIdx.x == 0 && threadIdx.x < HDR) reinterpret_cast<unsigned*>(dst)[threadIdx.x] = threadIdx.x == 0 ? amax[0] : 0u;
  uint4* out = reinterpret_cast<uint4*>(dst + HDR);
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < slots; i += (long long)gridDim.x * kThreads) {
    const int m = (int)(i % M), kb = (int)(i / M);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = kb * 8 + j < K ? a[(long long)(kb * 8 + j) * lda + m] : 0.0f;
    half8 hi, lo;
    split8(v, s, hi, lo);
    out[i] = __builtin_bit_cast(uint4, hi);
    out[slots + i] = __builtin_bit_cast(uint4, lo);
  }
}

// The same for a whole table of filters in three launches (a program packs every filter of a network once per step --
// trained -- or once -- frozen -- instead of inside every call): zero the headers, |max| (64 workgroups per filter,
// one atomic each), split.
struct GPackTable {
  ssad_gemm_pack_entry e[SSAD_MAX_PACK_ENTRIES];
  int count;
};
__global__ __launch_bounds__(kThreads) void gsplit_zero_kernel(const GPackTable t) {
  for (int i = (int)threadIdx.x; i < t.count * HDR; i += kThreads) reinterpret_cast<unsigned*>(t.e[i / HDR].dst)[i % HDR] = 0u;
}
__global__ __launch_bounds__(kThreads) void gsplit_amax_kernel(const GPackTable t) {
  const ssad_gemm_pack_entry& E = t.e[blockIdx.y];
  unsigned m = 0;
  for (int k = (int)blockIdx.x; k < E.K; k += (int)gridDim.x)
    for (int j = (int)threadIdx.x; j < E.M; j += kThreads) {
      const unsigned a = __float_as_uint(E.a[(long long)k * E.lda + j]) & 0x7fffffffu;
      m = m > a ? m : a;
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned other = (unsigned)__shfl_xor((int)m, o, 64);
    m = m > other ? m : other;
  }
  __shared__ unsigned red[kThreads / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / 64; ++w) m = m > red[w] ? m : red[w];
    if (m) atomicMax(reinterpret_cast<unsigned*>(E.dst), m);
  }
}
__global__ __launch_bounds__(kThreads) void gsplit_pack_multi_kernel(const GPackTable t) {
  const ssad_gemm_pack_entry& E = t.e[blockIdx.y];
  const int KB = (E.K + 7) >> 3;
  const long long slots = (long long)KB * E.M;
  const float s = pow2f(15 - split_exponent(reinterpret_cast<const unsigned*>(E.dst)[0]));
  uint4* out = reinterpret_cast<uint4*>(E.dst + HDR);
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < slots; i += (long long)gridDim.x * kThreads) {
    const int m = (int)(i % E.M), kb = (int)(i / E.M);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = kb * 8 + j < E.K ? E.a[(long long)(kb * 8 + j) * E.lda + m] : 0.0f;
    half8 hi, lo;
    split8(v, s, hi, lo);
    out[i] = __builtin_bit_cast(uint4, hi);
    out[slots + i] = __builtin_bit_cast(uint4, lo);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Forward / data gradient.  No packed copy of x exists: 8 waves (two per SIMD, so one wave's fetch, split and waits run
// under the other's MFMAs -- conv3x3_wgrad_split.hip), one 256-channel x 128-pixel item per
// workgroup, K in chunks of 32 channels through two LDS stages.  A thread fetches 8 channels of one pixel as eight
// coalesced dword loads (exactly one MFMA operand slot), splits them and writes hi / lo to LDS; the packed filter's
// slots go global -> register -> LDS unchanged.  Wave tile 64 channels x 64 pixels: per 16-channel step 8 LDS reads
// feed 12 MFMAs.
#ifndef GFLY_ABLATE    // debug builds (results wrong): 1 no fetch in the loop, 2 no split / LDS write, 4 one product, 8 no MFMA
#define GFLY_ABLATE 0
#endif
constexpr int FKC = 32;                             // channels per chunk (2 MFMA steps)
constexpr int FWG = 512;
constexpr int F_XP = (FKC / 8) * PT * 16;           // bytes of one x plane of a stage (8 KB)
constexpr int F_WP = (FKC / 8) * MT * 16;           // bytes of one filter plane of a stage (16 KB)
constexpr int F_STAGE = 2 * F_XP + 2 * F_WP;        // 48 KB
constexpr int F_LDS = 2 * F_STAGE;

struct FArgs {
  const float* x;
  const float* ap;       // packed a (header + planes)
  float* y;
  const float* bias;
  const float* residual;
  const float* mask;
  const unsigned* amax;  // [0] = x
  unsigned* amax_out;    // or null: the |max| of y is folded into this word (atomicMax; the caller zeroes it)
  int N, K, P, M, relu, accumulate;
  int ptiles, mblocks, items;
};

__global__ __launch_bounds__(FWG, 1) void gemm_fly_kernel(const FArgs q) {
  extern __shared__ __attribute__((aligned(16))) char flds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wp = wave >> 2;
  const int j = lane & 31, h = lane >> 5;
  const int K = q.K, M = q.M, P = q.P;
  const int KB = (K + 7) >> 3;
  const int nchunks = (K + FKC - 1) / FKC;

  // item: ids b, b + 8, ... share an XCD's L2: the channel blocks of one pixel tile are adjacent along that sequence
  int n, p0, ocb;
  {
    const int it = (int)blockIdx.x;
    const int xcd = it & 7, seq = it >> 3;
    const int mb = seq % q.mblocks;
    const int t = (seq / q.mblocks) * 8 + xcd;
    if (t >= q.N * q.ptiles) return;
    n = t / q.ptiles;
    p0 = (t - n * q.ptiles) * PT;
    ocb = mb * MT;
  }
  const unsigned w_lo_off = (unsigned)((long long)KB * M * 16);
  const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(q.x, (unsigned)((long long)q.N * K * P * 4));
  const __amdgpu_buffer_rsrc_t wrs = uniform_rsrc(q.ap + HDR, 2u * w_lo_off);
  const float sx = pow2f(15 - split_exponent(q.amax[0]));

  // what a thread fetches per chunk: x: pixel tid & 127, 8-channel group tid >> 7; filter: channel tid & 255, groups
  // (tid >> 8) and (tid >> 8) + 2 of both planes
  float xv[8];
  f32x4 wv[4];
  const int fpx = tid & (PT - 1), fkg = tid >> 7;
  const int fco = tid & (MT - 1), fkq = tid >> 8;
  const unsigned xbase = p0 + fpx < P ? (unsigned)((((long long)n * K + 8 * fkg) * P + p0 + fpx) * 4) : kOob;
  const int P4 = P * 4;
  auto fetch = [&](int c) {
    const int soff = __builtin_amdgcn_readfirstlane(c * FKC * P4);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      xv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
          xrs, c * FKC + 8 * fkg + e < K ? xbase : kOob, soff + e * P4, 0));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kb = c * (FKC / 8) + fkq + 2 * i;
      const unsigned vo = (kb < KB && ocb + fco < M) ? (unsigned)((kb * M + ocb + fco) * 16) : kOob;
      wv[2 * i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, vo, 0, 0));
      wv[2 * i + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, vo, (int)w_lo_off, 0));
    }
  };
  auto put = [&](char* stage) {
    half8 hi, lo;
    split8(xv, sx, hi, lo);
    char* px = stage + (fkg * PT + fpx) * 16;
    *reinterpret_cast<half8*>(px) = hi;
    *reinterpret_cast<half8*>(px + F_XP) = lo;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      char* pw = stage + 2 * F_XP + ((fkq + 2 * i) * MT + fco) * 16;
      *reinterpret_cast<f32x4*>(pw) = wv[2 * i];
      *reinterpret_cast<f32x4*>(pw + F_WP) = wv[2 * i + 1];
    }
  };

  float16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][tt][r] = 0.0f;

  auto step = [&](const char* stage, int st) {
    half8 ah[2], al[2], bh[2], bl[2];
    const int kg = 2 * st + h;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const char* pw = stage + 2 * F_XP + (kg * MT + wm * 64 + i * 32 + j) * 16;
      ah[i] = *reinterpret_cast<const half8*>(pw);
      al[i] = *reinterpret_cast<const half8*>(pw + F_WP);
      const char* px = stage + (kg * PT + wp * 64 + i * 32 + j) * 16;
      bh[i] = *reinterpret_cast<const half8*>(px);
      bl[i] = *reinterpret_cast<const half8*>(px + F_XP);
    }
#pragma unroll
    for (int pr = 0; pr < ((GFLY_ABLATE & 8) ? 0 : (GFLY_ABLATE & 4) ? 1 : 3); ++pr)   // hi.hi, lo(a).hi, hi.lo(x)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
          acc[i][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 1 ? al[i] : ah[i], pr == 2 ? bl[tt] : bh[tt],
                                                              acc[i][tt], 0, 0, 0);
    if (GFLY_ABLATE & 8) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { acc[i][0][0] += (float)ah[i][0] + (float)al[i][1]; acc[i][1][0] += (float)bh[i][0] + (float)bl[i][1]; }
    }
  };

  // chunk c multiplies from stage c & 1 while chunk c + 1 (in the registers since the previous iteration) is split into
  // the other stage, then chunk c + 2 is fetched; past the end the fetch returns zeros and the writes go to a stage
  // nobody reads
  fetch(0);
  put(flds);
  fetch(1);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const char* stage = flds + (c & 1) * F_STAGE;
    char* other = flds + ((c & 1) ^ 1) * F_STAGE;
    __builtin_amdgcn_sched_barrier(0);
    step(stage, 0);
    if (!(GFLY_ABLATE & 2)) put(other);
    __builtin_amdgcn_sched_barrier(0);
    if (!(GFLY_ABLATE & 1)) fetch(c + 2);
    __builtin_amdgcn_sched_barrier(0);
    step(stage, 1);
    __syncthreads();
  }

  // ---- epilogue: C/D row = (r & 3) + 8 (r >> 2) + 4 h, column = j
  const int oc_w = ocb + wm * 64;
  unsigned pvo[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int px = p0 + wp * 64 + 32 * tt + j;
    pvo[tt] = px < P ? (unsigned)((((long long)n * M + 4 * h) * P + px) * 4) : kOob;
  }
  const int e2 = split_exponent(q.amax[0]) + split_exponent(reinterpret_cast<const unsigned*>(q.ap)[0]) - 30;
  const bool one_scale = e2 >= -126 && e2 <= 127;
  const float sc1 = one_scale ? pow2f(e2) : pow2f(split_exponent(q.amax[0]) - 15);
  const float sc2 = one_scale ? 1.0f : pow2f(split_exponent(reinterpret_cast<const unsigned*>(q.ap)[0]) - 15);
  const unsigned ybytes = (unsigned)((long long)q.N * M * P * 4);
  const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(q.y, ybytes);
  const __amdgpu_buffer_rsrc_t rrs = uniform_rsrc(q.residual ? (const void*)q.residual : (const void*)q.y, q.residual ? ybytes : 0u);
  const __amdgpu_buffer_rsrc_t mrs = uniform_rsrc(q.mask ? (const void*)q.mask : (const void*)q.y, q.mask ? ybytes : 0u);
  const __amdgpu_buffer_rsrc_t brs = uniform_rsrc(q.bias ? (const void*)q.bias : (const void*)q.y, q.bias ? (unsigned)M * 4u : 0u);
  const bool relu = q.relu, has_res = q.residual != nullptr, has_mask = q.mask != nullptr, accum = q.accumulate;
  const bool ragged = (M & 7) != 0;
  unsigned ymax = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oc0 = oc_w + i * 32;
    float4 bq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
      bq[g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(brs, (unsigned)h * 16u, (oc0 + 8 * g) * 4, 0));
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      float rs[16], mk[16], old[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = oc0 + 8 * (r >> 2) + (r & 3);
        const unsigned vo = (ch < M && (!ragged || ch + 4 * h < M)) ? pvo[tt] : kOob;
        rs[r] = has_res ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, vo, ch * P4, 0)) : 0.0f;
        mk[r] = has_mask ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mrs, vo, ch * P4, 0)) : 1.0f;
        old[r] = accum ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, vo, ch * P4, 0)) : 0.0f;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (oc0 + 8 * g >= M) continue;               // wave-uniform
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          float v = one_scale ? fmaf(acc[i][tt][r], sc1, bq[g][e]) : acc[i][tt][r] * sc1 * sc2 + bq[g][e];
          v += rs[r];
          if (relu) v = fmaxf(v, 0.0f);
          if (has_mask) v = mk[r] > 0.0f ? v : 0.0f;
          v += old[r];
          const unsigned vo = (!ragged || oc0 + 8 * g + 4 * h + e < M) ? pvo[tt] : kOob;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, vo, (oc0 + 8 * g + e) * P4, 0);
          const unsigned av = __builtin_bit_cast(unsigned, v) & 0x7fffffffu;
          if (vo != kOob) ymax = ymax > av ? ymax : av;
        }
      }
    }
  }
  if (q.amax_out) {                                  // one atomic per workgroup
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned other = (unsigned)__shfl_xor((int)ymax, o, 64);
      ymax = ymax > other ? ymax : other;
    }
    __syncthreads();                                 // (everybody is done with the LDS stages)
    unsigned* red = reinterpret_cast<unsigned*>(flds);
    if (lane == 0) red[wave] = ymax;
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < FWG / 64; ++w) ymax = ymax > red[w] ? ymax : red[w];
      if (ymax) atomicMax(q.amax_out, ymax);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The pointwise FILTER GRADIENT the same way:  dw[m][c] (+)= sum_{n,p} dy[n][m][p] x[n][c][p]
// (conv_op_impl.h:451-500 for a 1x1 kernel).  The reduction index is the pixel, which is the contiguous index of both
// tensors: a thread fetches 8 consecutive pixels of one channel (two 16-byte loads = one MFMA operand slot), splits
// them and writes hi / lo to LDS.  Workgroup = 8 waves, a 256 x 256 block of dw for one share of the pixels, chunks of
// 32 pixels through two LDS stages; wave tile 128 x 64 (8 accumulator blocks): per 16-pixel step 12 LDS reads feed 24
// MFMAs.  Partial blocks go to a slab per share; wpoint_reduce_kernel adds them in a fixed order and divides the
// scales out.
constexpr int WT = 256;                             // channels of dy and of x per workgroup
constexpr int WPX = 32;                             // pixels per chunk (2 MFMA steps)
constexpr int W_PL = (WPX / 8) * WT * 16;           // bytes of one plane of one operand of a stage (16 KB)
constexpr int W_STAGE = 4 * W_PL;                   // dy hi, dy lo, x hi, x lo
constexpr int W_LDS = 2 * W_STAGE;                  // 128 KB

struct WPArgs {
  const float* x;
  const float* dy;
  float* slabs;
  const unsigned* xa;    // |max| word of x
  const unsigned* da;    // |max| word of dy
  int N, C, P, M;
  int mtiles, ctiles, shares, per_share, total, cpi;     // cpi = chunks per image
  int xcd_runs;
};

__global__ __launch_bounds__(FWG, 1) void wpoint_split_kernel(const WPArgs q) {
  extern __shared__ __attribute__((aligned(16))) char wlds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wc = wave >> 1;          // wave block: 128 rows (m) x 64 columns (c)
  const int j = lane & 31, h = lane >> 5;
  const int P = q.P;

  const int tiles = q.mtiles * q.ctiles;
  int share, tile;
  if (q.xcd_runs) {                                 // the tiles of a share read the same pixels: one XCD
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    share = xcd * q.xcd_runs + jj / tiles;
    tile = jj % tiles;
  } else {
    share = blockIdx.x / tiles;
    tile = blockIdx.x % tiles;
  }
  const int m0 = (tile / q.ctiles) * WT, c0 = (tile % q.ctiles) * WT;
  const int q_begin = share * q.per_share;
  const int q_end = q_begin + q.per_share < q.total ? q_begin + q.per_share : q.total;

  const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(q.x, (unsigned)((long long)q.N * q.C * P * 4));
  const __amdgpu_buffer_rsrc_t drs = uniform_rsrc(q.dy, (unsigned)((long long)q.N * q.M * P * 4));
  const float sx = pow2f(15 - split_exponent(q.xa[0]));
  const float sd = pow2f(15 - split_exponent(q.da[0]));

  // a thread's fetch per chunk: pixel group tid & 3, channels (tid >> 2) and (tid >> 2) + 128 of dy and of x
  float dv[2][8], xv[2][8];
  const int fpg = tid & 3, fch = tid >> 2;
  auto fetch = [&](int ck) {
    const int n = __builtin_amdgcn_readfirstlane(ck / q.cpi);
    const int p = (ck - n * q.cpi) * WPX + 8 * fpg;
    const bool pok = ck < q.total && p < P;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + fch + 128 * i, c = c0 + fch + 128 * i;
      const unsigned od = (pok && m < q.M) ? (unsigned)((((long long)n * q.M + m) * P + p) * 4) : kOob;
      const unsigned ox = (pok && c < q.C) ? (unsigned)((((long long)n * q.C + c) * P + p) * 4) : kOob;
      const f32x4 d0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, od, 0, 0));
      const f32x4 d1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, od, 16, 0));
      const f32x4 x0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ox, 0, 0));
      const f32x4 x1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ox, 16, 0));
#pragma unroll
      for (int e = 0; e < 4; ++e) { dv[i][e] = d0[e]; dv[i][4 + e] = d1[e]; xv[i][e] = x0[e]; xv[i][4 + e] = x1[e]; }
    }
  };
  auto put = [&](char* stage) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      half8 hi, lo;
      char* pd = stage + (fpg * WT + fch + 128 * i) * 16;
      split8(dv[i], sd, hi, lo);
      *reinterpret_cast<half8*>(pd) = hi;
      *reinterpret_cast<half8*>(pd + W_PL) = lo;
      split8(xv[i], sx, hi, lo);
      *reinterpret_cast<half8*>(pd + 2 * W_PL) = hi;
      *reinterpret_cast<half8*>(pd + 3 * W_PL) = lo;
    }
  };

  float16v acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][tt][r] = 0.0f;

  auto step = [&](const char* stage, int st) {
    const int kg = 2 * st + h;
    half8 bh[2], bl[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const char* px = stage + 2 * W_PL + (kg * WT + wc * 64 + tt * 32 + j) * 16;
      bh[tt] = *reinterpret_cast<const half8*>(px);
      bl[tt] = *reinterpret_cast<const half8*>(px + W_PL);
    }
#pragma unroll
    for (int ip = 0; ip < 2; ++ip) {                // two row blocks at a time: their operands, then 12 MFMAs
      half8 ah[2], al[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const char* pd = stage + (kg * WT + wm * 128 + (2 * ip + i) * 32 + j) * 16;
        ah[i] = *reinterpret_cast<const half8*>(pd);
        al[i] = *reinterpret_cast<const half8*>(pd + W_PL);
      }
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)                // hi.hi, lo(dy).hi, hi.lo(x)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
            acc[2 * ip + i][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 1 ? al[i] : ah[i], pr == 2 ? bl[tt] : bh[tt],
                                                                         acc[2 * ip + i][tt], 0, 0, 0);
    }
  };

  if (q_begin < q_end) {
    fetch(q_begin);
    put(wlds);
    fetch(q_begin + 1 < q_end ? q_begin + 1 : q.total);
  }
  __syncthreads();
  for (int ck = q_begin; ck < q_end; ++ck) {
    const char* stage = wlds + ((ck - q_begin) & 1) * W_STAGE;
    char* other = wlds + (((ck - q_begin) & 1) ^ 1) * W_STAGE;
    __builtin_amdgcn_sched_barrier(0);
    step(stage, 0);
    put(other);
    __builtin_amdgcn_sched_barrier(0);
    fetch(ck + 2 < q_end ? ck + 2 : q.total);      // past the share's end: zeros (nobody reads that stage)
    __builtin_amdgcn_sched_barrier(0);
    step(stage, 1);
    __syncthreads();
  }

  // ---- partial block -> slab [share][Mp][Cp]; C/D row = (r & 3) + 8 (r >> 2) + 4 h, column = j
  const int Mp = q.mtiles * WT, Cp = q.ctiles * WT;
  float* slab = q.slabs + (long long)share * Mp * Cp;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 128 + i * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        const int c = c0 + wc * 64 + tt * 32 + j;
        slab[(long long)m * Cp + c] = acc[i][tt][r];
      }
}

__global__ __launch_bounds__(256) void wpoint_reduce_kernel(const float* __restrict__ slabs, int shares, int Mp, int Cp,
                                                            int M, int C, const unsigned* __restrict__ xa,
                                                            const unsigned* __restrict__ da,
                                                            float* __restrict__ dw, int accumulate) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)M * C) return;
  const int m = (int)(e / C), c = (int)(e - (long long)m * C);
  const float* p = slabs + (long long)m * Cp + c;
  const long long stride = (long long)Mp * Cp;
  float s = 0.0f;
  for (int sh = 0; sh < shares; ++sh) s += p[sh * stride];
  s = (s * pow2f(split_exponent(xa[0]) - 15)) * pow2f(split_exponent(da[0]) - 15);
  dw[e] = accumulate ? dw[e] + s : s;
}

int wpoint_plan(int N, int C, int P, int M, WPArgs* a) {
  if (N <= 0 || C <= 0 || P <= 0 || M <= 0 || (P & 7)) return SSAD_E_BADARG;
  if ((long long)N * C * P * 4 >= (1LL << 31) || (long long)N * M * P * 4 >= (1LL << 31)) return SSAD_E_BADARG;
  a->N = N; a->C = C; a->P = P; a->M = M;
  a->mtiles = cdiv(M, WT); a->ctiles = cdiv(C, WT);
  a->cpi = cdiv(P, WPX);
  a->total = N * a->cpi;
  const int cus = ssad_cu_count(), tiles = a->mtiles * a->ctiles;
  int s = cus / tiles;
  if (s < 1) s = 1;
  if (s >= 8) s &= ~7;
  while (s > 1 && a->total / s < 4) --s;
  a->per_share = cdiv(a->total, s);
  a->shares = cdiv(a->total, a->per_share);
  a->xcd_runs = (a->shares % 8 == 0) ? a->shares / 8 : 0;
  return 0;
}

struct GPlan {
  size_t amax_off, a_off, total;
};
int make_plan(const ssad_gemm_conv* d, GPlan* p) {
  if (!d || d->N <= 0 || d->K <= 0 || d->P <= 0 || d->M <= 0 || d->lda < d->M) return SSAD_E_BADARG;
  const int KB = (d->K + 7) >> 3;
  const long long aslots = (long long)KB * d->M;
  // byte offsets are 32-bit and 2^31 means "outside": larger tensors stay on the exact-fp32 engine
  if ((long long)d->N * d->K * d->P * 4 >= (1LL << 31) || (long long)d->N * d->M * d->P * 4 >= (1LL << 31) ||
      aslots * 32 >= (1LL << 31))
    return SSAD_E_BADARG;
  p->amax_off = 0;
  p->a_off = 256;
  p->total = p->a_off + (size_t)HDR * 4 + (size_t)aslots * 32;
  return 0;
}

}  // namespace

extern "C" {

size_t ssad_conv1x1_gemm_split_workspace_bytes(const ssad_gemm_conv* d) {
  GPlan p;
  return make_plan(d, &p) ? 0 : p.total;
}

size_t ssad_gemm_split_filter_floats(int K, int M) {
  if (K <= 0 || M <= 0) return 0;
  return (size_t)HDR + (size_t)((K + 7) >> 3) * M * 8;
}

int ssad_gemm_split_pack_filters(const ssad_gemm_pack_entry* entries_host, int n_entries, ssad_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!entries_host || n_entries < 0) return SSAD_E_BADARG;
  for (int base = 0; base < n_entries; base += SSAD_MAX_PACK_ENTRIES) {
    GPackTable t;
    const int cnt = n_entries - base < SSAD_MAX_PACK_ENTRIES ? n_entries - base : SSAD_MAX_PACK_ENTRIES;
    long long most = 0;
    for (int i = 0; i < cnt; ++i) {
      t.e[i] = entries_host[base + i];
      if (!t.e[i].a || !t.e[i].dst || t.e[i].K <= 0 || t.e[i].M <= 0 || t.e[i].lda < t.e[i].M) return SSAD_E_BADARG;
      const long long slots = (long long)((t.e[i].K + 7) >> 3) * t.e[i].M;
      if (slots > most) most = slots;
    }
    t.count = cnt;
    hipLaunchKernelGGL(gsplit_zero_kernel, dim3(1), dim3(kThreads), 0, stream, t);
    hipLaunchKernelGGL(gsplit_amax_kernel, dim3(64u, (unsigned)cnt), dim3(kThreads), 0, stream, t);
    long long bx = (most + kThreads - 1) / kThreads;
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(gsplit_pack_multi_kernel, dim3((unsigned)bx, (unsigned)cnt), dim3(kThreads), 0, stream, t);
  }
  return (int)hipGetLastError();
}

int ssad_conv1x1_gemm_split(const ssad_gemm_conv* d, void* workspace, size_t workspace_bytes, ssad_stream_t stream_) {
  return ssad_conv1x1_gemm_split_amax(d, nullptr, nullptr, nullptr, workspace, workspace_bytes, stream_);
}

int ssad_conv1x1_gemm_split_amax(const ssad_gemm_conv* d, const float* packed_a, const unsigned* x_amax,
                                 unsigned* y_amax_out, void* workspace, size_t workspace_bytes, ssad_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GPlan p;
  const int rc = make_plan(d, &p);
  if (rc) return rc;
  if (!d->a || !d->x || !d->y) return SSAD_E_BADARG;
  if ((d->flags & SSAD_GEMM_ACCUMULATE) && (d->bias || d->residual)) return SSAD_E_BADARG;
  if (!workspace || workspace_bytes < p.total) return SSAD_E_WORKSPACE;
  char* ws = (char*)workspace;
  unsigned* amax = (unsigned*)(ws + p.amax_off);
  float* apk = (float*)(ws + p.a_off);
  const int KB = (d->K + 7) >> 3;
  // |max| of x (word 0) and a (word 1), one launch
  AmaxTable at;
  for (int l = 0; l < kMaxLv; ++l) { at.x[l] = nullptr; at.n[l] = 0; at.block_start[l] = 0; }
  at.count = 2;
  at.amax = amax;
  at.x[0] = d->x; at.n[0] = (long long)d->N * d->K * d->P;
  at.x[1] = d->a; at.n[1] = (long long)d->K * d->lda;
  int blocks = 0;
  for (int l = 0; l < 2; ++l) {
    at.block_start[l] = blocks;
    long long nb = (at.n[l] / 4 + kThreads * 32 - 1) / (kThreads * 32);
    blocks += (int)(nb < 1 ? 1 : nb > 512 ? 512 : nb);
  }
  for (int l = 2; l <= kMaxLv; ++l) at.block_start[l] = blocks;
  // what the caller did not hand over is measured here: x's |max| (word 0), the filter's (word 1) + its split
  if (x_amax && packed_a) {
    // nothing to do
  } else {
    if (x_amax) { at.n[0] = 0; }            // (entry 0 keeps one empty block)
    if (packed_a) { at.n[1] = 0; }
    (void)hipMemsetAsync(amax, 0, 256, stream);
    hipLaunchKernelGGL(split_absmax_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, at);
    if (!packed_a) {
      long long bx = ((long long)KB * d->M + kThreads - 1) / kThreads;
      if (bx > 1024) bx = 1024;
      hipLaunchKernelGGL(gsplit_pack_a_kernel, dim3((unsigned)bx), dim3(kThreads), 0, stream, d->a, d->lda, d->K, d->M,
                         (const unsigned*)(amax + 1), apk);
    }
  }
  FArgs f;
  f.x = d->x; f.ap = packed_a ? packed_a : apk; f.y = d->y; f.bias = d->bias; f.residual = d->residual; f.mask = d->mask;
  f.amax = x_amax ? x_amax : amax;
  f.amax_out = y_amax_out;
  f.N = d->N; f.K = d->K; f.P = d->P; f.M = d->M;
  f.relu = (d->flags & SSAD_GEMM_RELU) ? 1 : 0;
  f.accumulate = (d->flags & SSAD_GEMM_ACCUMULATE) ? 1 : 0;
  f.ptiles = cdiv(d->P, PT);
  f.mblocks = cdiv(d->M, MT);
  const long long tiles = (long long)d->N * f.ptiles;
  if (tiles * f.mblocks >= (1LL << 30)) return SSAD_E_BADARG;
  f.items = (int)(cdiv((int)tiles, 8) * 8 * f.mblocks);
  static std::once_flag lds_once;           // > 64 KiB of dynamic LDS needs the opt-in, once per process
  std::call_once(lds_once, [&] {
    (void)hipFuncSetAttribute((const void*)gemm_fly_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
  });
  hipLaunchKernelGGL(gemm_fly_kernel, dim3((unsigned)f.items), dim3(FWG), F_LDS, stream, f);
  return (int)hipGetLastError();
}

/* |max| of one tensor into a caller-owned word (the caller zeroes it) */
int ssad_split_absmax(const float* x, long long n, unsigned* word, ssad_stream_t stream_) {
  if (!x || !word || n <= 0) return SSAD_E_BADARG;
  AmaxTable at;
  for (int l = 0; l < kMaxLv; ++l) { at.x[l] = nullptr; at.n[l] = 0; at.block_start[l] = 0; }
  at.count = 1;
  at.amax = word;
  at.x[0] = x; at.n[0] = n;
  long long nb = (n / 4 + kThreads * 32 - 1) / (kThreads * 32);
  const int blocks = (int)(nb < 1 ? 1 : nb > 512 ? 512 : nb);
  for (int l = 1; l <= kMaxLv; ++l) at.block_start[l] = blocks;
  hipLaunchKernelGGL(split_absmax_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream_, at);
  return (int)hipGetLastError();
}

size_t ssad_conv1x1_wgrad_split_workspace_bytes(int N, int C, int P, int M) {
  WPArgs a;
  if (wpoint_plan(N, C, P, M, &a)) return 0;
  return 256 + sizeof(float) * (size_t)a.shares * a.mtiles * WT * a.ctiles * WT;
}

int ssad_conv1x1_wgrad_split(const float* x, const float* dy, int N, int C, int P, int M, float* dw, int accumulate,
                             void* workspace, size_t workspace_bytes, ssad_stream_t stream_) {
  return ssad_conv1x1_wgrad_split_amax(x, dy, N, C, P, M, dw, accumulate, workspace, workspace_bytes, nullptr, nullptr,
                                       stream_);
}

int ssad_conv1x1_wgrad_split_amax(const float* x, const float* dy, int N, int C, int P, int M, float* dw, int accumulate,
                                  void* workspace, size_t workspace_bytes, const unsigned* x_amax,
                                  const unsigned* dy_amax, ssad_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if ((x_amax == nullptr) != (dy_amax == nullptr)) return SSAD_E_BADARG;
  WPArgs a;
  const int rc = wpoint_plan(N, C, P, M, &a);
  if (rc) return rc;
  if (!x || !dy || !dw) return SSAD_E_BADARG;
  const size_t need = 256 + sizeof(float) * (size_t)a.shares * a.mtiles * WT * a.ctiles * WT;
  if (!workspace || workspace_bytes < need) return SSAD_E_WORKSPACE;
  unsigned* amax = (unsigned*)workspace;
  a.x = x; a.dy = dy;
  a.xa = x_amax ? x_amax : amax;
  a.da = dy_amax ? dy_amax : amax + 1;
  a.slabs = (float*)((char*)workspace + 256);
  if (!x_amax) {
  AmaxTable at;
  for (int l = 0; l < kMaxLv; ++l) { at.x[l] = nullptr; at.n[l] = 0; at.block_start[l] = 0; }
  at.count = 2;
  at.amax = amax;
  at.x[0] = x; at.n[0] = (long long)N * C * P;
  at.x[1] = dy; at.n[1] = (long long)N * M * P;
  int blocks = 0;
  for (int l = 0; l < 2; ++l) {
    at.block_start[l] = blocks;
    long long nb = (at.n[l] / 4 + kThreads * 32 - 1) / (kThreads * 32);
    blocks += (int)(nb < 1 ? 1 : nb > 512 ? 512 : nb);
  }
  for (int l = 2; l <= kMaxLv; ++l) at.block_start[l] = blocks;
  (void)hipMemsetAsync(amax, 0, 256, stream);
  hipLaunchKernelGGL(split_absmax_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, at);
  }
  static std::once_flag lds_once;
  std::call_once(lds_once, [&] {
    (void)hipFuncSetAttribute((const void*)wpoint_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
  });
  hipLaunchKernelGGL(wpoint_split_kernel, dim3((unsigned)(a.mtiles * a.ctiles * a.shares)), dim3(FWG), W_LDS, stream, a);
  const long long n = (long long)M * C;
  hipLaunchKernelGGL(wpoint_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                     (const float*)a.slabs, a.shares, a.mtiles * WT, a.ctiles * WT, M, C, a.xa, a.da, dw, accumulate);
  return (int)hipGetLastError();
}

}  // extern "C"
