// conv3x3_split.hip -- fp32 3x3 convolution (forward / data gradient) on the fp16 matrix pipes by operand
// splitting (round 6).
//
// Why.  gfx950's fp32 MFMA peaks at 157 TFLOP/s, its fp16 MFMA at 2.5 PFLOP/s (16 x), and there is no xf32 mode
// (MI355X_MICROARCH.md).  An fp32 number scaled by a per-tensor power of two s splits exactly into two fp16 numbers
//
//     x s = hi + lo,    hi = fp16(x s),   lo = fp16(x s - hi)          (22 significant bits)
//
// and a product of two such numbers is  hi hi' + lo hi' + hi lo'  up to the 2^-22 term lo lo': three
// v_mfma_f32_32x32x16_f16 with fp32 accumulation per operand pair -- 3/16 of the fp32 MFMA's pipe time for the DIRECT
// convolution, i.e. 9 * 3/16 = 1.69 fp32-MFMA equivalents per output where Winograd F(2x4) on the fp32 pipes pays 3,
// with NO Winograd transforms in the way: measured error against a float64 convolution ~3e-7 of the output scale
// (F(2x4) fp32: 1.5-2.2e-6, conv3x3_winograd24.hip).  The scales are exact powers of two taken from the tensors'
// measured |max| (so that hi < 2^15 never overflows fp16) and are divided out of the fp32 accumulators in the
// epilogue.  What is NOT fp32-like: an element smaller than 2^-29 of its tensor's |max| keeps fewer than 22 bits
// (absolute error <= 2^-40 |max|: fp16's denormal floor under the scale); a tensor holding Inf / NaN goes through
// unscaled (s = 1).
//
// Same operator contract as the other 3x3 engines (caffe2/operators/conv_op_cudnn.cc:567-617 forward, :1040-1058
// data gradient as the same kernel on the flipped / transposed pack; NCHW fp32 in, NCHW fp32 out; bias, ReLU,
// Sigmoid, fused ReluGradient mask).  One call = three launches on the caller's stream:
//   1. split_absmax_kernel      |max| of every level's input (one word per level, atomicMax on the float's bits)
//   2. split_pack_act_kernel    NCHW fp32 -> two channel-blocked fp16 planes  Xb[n][c/8][y][x][c%8]  (hi, then lo)
//   3. conv3x3_split_kernel     the convolution
// Kernel 3 is conv3x3_f16.hip's tiling (workgroup = 4 waves = 128 output channels x 16 x 16 pixels, wave = 64 x 128 =
// 2 x 4 MFMA tiles of 32 x 32, 128 accumulator VGPRs) with 16-channel K chunks: per chunk the 18 x 18 x 16 halo tile
// of BOTH planes goes to LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 20.7 KB), per filter tap 4 x 16-byte filter
// operands (hi / lo x two 32-channel halves, from the L2-resident pack through a 3-step register ring with
// hand-counted s_waitcnt vmcnt) and 8 ds_read_b128 feed 24 MFMAs -- three times the matrix work per staged byte of
// the plain fp16 kernel, which is what moves it from 0.45 of the fp16 peak towards the pipes' limit.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "conv_internal.h"
#include "ssad_kernels.h"

#include "split_common.h"

namespace {

using namespace ssad_split;

constexpr int TS = 16;                 // output tile edge
constexpr int HS = TS + 2;             // halo tile edge
constexpr int CBC = 2;                 // 8-channel blocks per K chunk (16 channels = one MFMA K)
constexpr int SLOTS = CBC * HS * HS;   // 16-byte slots per plane and stage (648)
constexpr int NLD = 3;                 // LDS-DMA instructions per wave, plane and stage
constexpr int STAGE = 4 * NLD * 64;    // stage pitch in slots (768: instruction k = 11 only lands zeros in the tail)
constexpr int AD = 9;                  // filter ring depth in taps = one chunk: the operands of (chunk c + 1, tap t) are requested
                                       // at the end of (chunk c, tap t); slot = tap, a compile-time constant
constexpr int NBUF = 3;                // LDS stages: the halo of chunk c + 2 is requested during chunk c
constexpr int MT = 128;                // output channels per workgroup
constexpr int HDR = 16;                // floats in front of a packed filter: [0] = |max| of the filter (bits)
static_assert(AD == 9 && NLD == 3, "the counted waits below are written for these");

#ifndef SPLIT_ABLATE     // debug builds (results wrong): 1 no halo DMA traffic, 2 no filter ring traffic, 4 only hi*hi
#define SPLIT_ABLATE 0
#endif

#ifdef SPLIT_TIMELINE      // debug build (tools/dbg/r6_split_timeline.sh): cycle stamps of wave 0 of workgroup 5, per item
__device__ unsigned long long g_split_dbg[64][8];
#define SDBG(itn, k) if (dbg_on && (itn) < 64) g_split_dbg[itn][k] = __builtin_readcyclecounter()
#else
#define SDBG(itn, k)
#endif

__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- filter packs ---------------------------------------------------------------------------------------
// packed = [HDR floats: |max| bits][hi: Wp[tap][K/8][M] x 16 B][lo: same]; forward: (M, K) = (Cout, Cin); data
// gradient: (Cin, Cout), W'[c][m][tap] = W[m][c][8 - tap] (conv_op_impl.h:524-560).
struct FPackTable {
  ssad_pack_entry e[SSAD_MAX_PACK_ENTRIES];
};
// header words of both packs to zero (the |max| below is an atomicMax)
__global__ void split_filter_zero_kernel(const FPackTable t) {
  const ssad_pack_entry& e = t.e[blockIdx.x];
  if (threadIdx.x < HDR) {
    if (e.packed_fwd) reinterpret_cast<unsigned*>(e.packed_fwd)[threadIdx.x] = 0u;
    if (e.packed_dgrad) reinterpret_cast<unsigned*>(e.packed_dgrad)[threadIdx.x] = 0u;
  }
}
// |max| of filter blockIdx.y: one atomic per workgroup (a single workgroup per filter took 0.77 ms per step on the
// 720 x 256 x 9 filter of cls_pred, profiles/r06_bench_heads_trace.md)
__global__ __launch_bounds__(kThreads) void split_filter_amax_kernel(const FPackTable t) {
  const ssad_pack_entry& e = t.e[blockIdx.y];
  const long long n = (long long)e.Cout * e.Cin * 9;
  unsigned m = 0;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
    const unsigned a = __float_as_uint(e.w[i]) & 0x7fffffffu;
    m = m > a ? m : a;
  }
  __shared__ unsigned red[kThreads / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned other = (unsigned)__shfl_xor((int)m, o, 64);
    m = m > other ? m : other;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned r = red[0];
    for (int w = 1; w < kThreads / 64; ++w) r = r > red[w] ? r : red[w];
    if (r) {
      if (e.packed_fwd) atomicMax(reinterpret_cast<unsigned*>(e.packed_fwd), r);
      if (e.packed_dgrad) atomicMax(reinterpret_cast<unsigned*>(e.packed_dgrad), r);
    }
  }
}
__global__ __launch_bounds__(kThreads) void split_filter_pack_kernel(const FPackTable t) {
  const ssad_pack_entry& e = t.e[blockIdx.y];
  const bool dg = blockIdx.z == 1;
  float* dst = dg ? e.packed_dgrad : e.packed_fwd;
  if (!dst) return;
  const int M = dg ? e.Cin : e.Cout, K = dg ? e.Cout : e.Cin;       // the pack's (outputs, inputs)
  const int KB = (K + 7) >> 3;
  const long long slots = 9LL * KB * M;
  const float s = pow2f(15 - split_exponent(reinterpret_cast<const unsigned*>(dst)[0]));
  uint4* out = reinterpret_cast<uint4*>(dst + HDR);
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < slots; i += (long long)gridDim.x * kThreads) {
    const int m = (int)(i % M), kb = (int)((i / M) % KB), tap = (int)(i / ((long long)M * KB));
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kb * 8 + j;
      v[j] = k >= K ? 0.0f : dg ? e.w[((long long)k * e.Cin + m) * 9 + (8 - tap)] : e.w[((long long)m * e.Cin + k) * 9 + tap];
    }
    half8 hi, lo;
    split8(v, s, hi, lo);
    out[i] = __builtin_bit_cast(uint4, hi);
    out[slots + i] = __builtin_bit_cast(uint4, lo);
  }
}

// ---- 3. the convolution -----------------------------------------------------------------------------------
struct SLevels {
  const uint4* x[kMaxLv];        // hi plane; lo plane at + N * CB * H * W slots
  float* y[kMaxLv];
  const float* aux[kMaxLv];      // SSAD_CONV_MASK_AUX: y = aux > 0 ? y : 0
  const float* w[kMaxLv];        // packed filter (header + planes)
  const float* bias[kMaxLv];
  int N[kMaxLv], H[kMaxLv], W[kMaxLv];
  int tile0[kMaxLv + 1];
  const unsigned* amax;          // [n_levels] |max| words of the inputs
  unsigned* amax_out;            // [n_levels] or null: |max| of the outputs is folded in (atomicMax; the caller zeroes)
  int n_levels, C, M, relu, sigmoid, mblocks, items;
};

// One work item = (16 x 16-pixel tile, 128-channel block).  The kernel is PERSISTENT (grid = #CUs, one workgroup of
// four waves per CU = one wave per SIMD with the whole 512-entry register file: 128 accumulators, a filter ring one
// chunk (9 taps x 4 operands = 144 registers) deep, 32 B-operand registers): a second workgroup per CU would cap a
// wave at 256 registers, i.e. a 3-tap ring, and the in-order return of vector memory then parks the L2-resident filter
// operands behind every halo fetch from HBM (measured: 0.7 of 2.1 ms).  What a second workgroup would have hidden is
// hidden by hand instead: the next item's first two halo chunks and first nine filter taps are requested BEFORE the
// current item's epilogue, three LDS stages let a halo chunk fly for more than a whole chunk of MFMAs.
template <bool MASKED>
__global__ __launch_bounds__(kThreads, 1) void conv3x3_split_kernel(const SLevels q) {
  __shared__ uint4 lds[NBUF * 2 * STAGE];        // [buffer][plane][STAGE]
  __shared__ unsigned wg_max[kThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave & 1, wp = wave >> 1;
  const int j = lane & 31, h = lane >> 5;
  const int C = q.C, M = q.M;
  const int CB = (C + 7) >> 3;
  const int nchunks = (CB + CBC - 1) / CBC;
  const unsigned w_lo_off = (unsigned)(9LL * CB * M * 16);
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr)lds;
  const int brow = wp * 8 + (j >> 4), bcol = (j & 16) ? ((j - 2) & 15) : j;
  const int bbase = (h * HS + brow) * HS + bcol;           // + (2 tt + dy) * HS + dx

  // ---- the item under the LOADER's cursor (one item ahead of the compute side between prologue and epilogue)
  struct Item {
    int lv, n, y0, x0, ocb, H, W, N;
  };
  auto decode = [&](int it) {
    // item id -> (tile, output-channel block): ids b, b + 8, ... share an XCD's L2; the channel blocks of one tile are
    // laid out along that sequence and fetch the tile's planes from HBM once (conv3x3_f16.hip)
    Item o;
    const int xcd = it & 7, seq = it >> 3;
    const int mb = seq % q.mblocks;
    int t = (seq / q.mblocks) * 8 + xcd;
    int lv = 0;
    for (int l = 1; l < q.n_levels; ++l)
      if (t >= q.tile0[l]) lv = l;
    o.lv = t < q.tile0[q.n_levels] ? lv : -1;              // (the id space is padded to a multiple of 8 tiles)
    o.N = q.N[lv]; o.H = q.H[lv]; o.W = q.W[lv];
    const int tiles_x = (o.W + TS - 1) / TS, tiles_y = (o.H + TS - 1) / TS;
    t -= q.tile0[lv];
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    o.n = t / tiles_y;
    o.y0 = ty * TS; o.x0 = tx * TS;
    o.ocb = mb * MT;
    return o;
  };
  ssad_dev::rsrc_words xrs = ssad_dev::uniform_rsrc_words(q.x[0], 0), wrs = xrs;
  unsigned dvo[NLD], avo[2], x_lo_off = 0;
  int dcb[NLD], plane16 = 0;
  auto bind = [&](const Item& I) {
    const long long plane = (long long)I.H * I.W;
    plane16 = (int)plane * 16;
    x_lo_off = (unsigned)((long long)I.N * CB * plane16);          // bytes from the hi to the lo plane
    xrs = ssad_dev::uniform_rsrc_words(q.x[I.lv], 2u * x_lo_off);
    wrs = ssad_dev::uniform_rsrc_words(q.w[I.lv] + HDR, 2u * w_lo_off);
    // halo staging by LDS-DMA: slot s = (block, row, col) of the 18 x 18 x 2-block tile of one plane; wave-level
    // instruction k writes slots [64 k, 64 k + 64) of a plane's stage; wave w issues k = w, w + 4, w + 8 per plane.
    // Lanes outside the image / past the last slot go to an out-of-range offset (zero fill).
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int s = 64 * (wave + 4 * i) + lane;
      const int cbl = s / (HS * HS), r = s % (HS * HS);
      const int gy = I.y0 - 1 + r / HS, gx = I.x0 - 1 + r % HS;
      const bool ok = s < SLOTS && gy >= 0 && gy < I.H && gx >= 0 && gx < I.W;
      dcb[i] = cbl;
      dvo[i] = ok ? (unsigned)((((long long)I.n * CB + cbl) * plane + (long long)gy * I.W + gx) * 16) : kOob;
    }
    // A: Wp[tap][cb][m] x 16 B; lane = (row m, 8-channel block h).  Rows past M are clamped (never stored).
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int oc = I.ocb + wo * 64 + i * 32 + j;
      avo[i] = (unsigned)(h * M + (oc < M ? oc : M - 1)) * 16u;
    }
  };
  // piece i (0..2) of chunk `chunk` into stage `buf`, both planes (2 instructions)
  auto dma_piece = [&](int chunk, int buf, int i, bool real) {
    const unsigned vo = (real && !(SPLIT_ABLATE & 1) && chunk * CBC + dcb[i] < CB) ? dvo[i] : kOob;
    const int soff = __builtin_amdgcn_readfirstlane(real ? chunk * CBC * plane16 : 0);
    const unsigned dst = lds_base + (unsigned)((buf * 2 * STAGE + 64 * (wave + 4 * i)) * 16);
    dma16(xrs, dst, vo, soff);
    dma16(xrs, dst + STAGE * 16, vo, soff + (int)x_lo_off);
  };
  // ring slot = 4 operands: [hi half 0, hi half 1, lo half 0, lo half 1]
  f32x4 ar[AD][4];
  // operand k of a ring slot: 0 / 1 = hi of the wave's two 32-channel halves, 2 / 3 = lo
  auto ring_load1 = [&](f32x4& dst, int chunk, int tap, int k, bool real) {
    const int soff = __builtin_amdgcn_readfirstlane(real ? ((tap * CB + chunk * CBC) * M) * 16 + (k >> 1) * (int)w_lo_off : 0);
    const unsigned vo = (real && !(SPLIT_ABLATE & 2)) ? avo[k & 1] : kOob;
    // s_nop 4: the scalar operands of an asm statement may have been written by a VALU instruction just before it
    // (v_readlane restoring a spilled SGPR, v_readfirstlane) -- "VALU writes SGPR -> VMEM reads it" needs 5 wait
    // states, which hipcc inserts for its own loads but cannot for a load it does not see (found the hard way: the
    // first load after each restore read the previous tap's offset, +-1 x one tap's contribution in one wave's half).
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(vo), "s"(wrs), "s"(soff));
  };
  auto ring_load = [&](f32x4 (&dst)[4], int chunk, int tap, bool real) {
#pragma unroll
    for (int k = 0; k < 4; ++k) ring_load1(dst[k], chunk, tap, k, real);
  };
  // everything requested so far has landed; the statement owns every ring register (tools/isa_lint.py)
  auto ring_landed = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int a = 0; a < AD; ++a) asm volatile("" : "+v"(ar[a][0]), "+v"(ar[a][1]), "+v"(ar[a][2]), "+v"(ar[a][3]));
  };
  int gbuf = 0;                                   // LDS stage of the NEXT chunk 0 to be requested
  // an item's first two halo chunks and its first nine filter taps (the loader's item must be bound)
  auto prologue = [&]() {
    int b1 = gbuf + 1; if (b1 >= NBUF) b1 -= NBUF;
#pragma unroll
    for (int i = 0; i < NLD; ++i) dma_piece(0, gbuf, i, true);
#pragma unroll
    for (int i = 0; i < NLD; ++i) dma_piece(1, b1, i, nchunks > 1);
#pragma unroll
    for (int a = 0; a < AD; ++a) ring_load(ar[a], 0, a, true);
  };

  const int G = (int)gridDim.x;
  int it = (int)blockIdx.x;
  Item cur = decode(it);
  // (ids past the padded id space never occur: grid <= items; an id inside the padding has lv = -1 and is skipped)
  while (it < q.items && cur.lv < 0) { it += G; if (it < q.items) cur = decode(it); }
  if (it >= q.items) return;
  bind(cur);
  prologue();
  ring_landed();
  __builtin_amdgcn_s_barrier();

#ifdef SPLIT_TIMELINE
  const bool dbg_on = blockIdx.x == 5 && tid == 0;
  int itn = 0;
#endif
  while (true) {
    SDBG(itn, 0);
    float16v acc[2][4];
    {
      float zero;                                  // (a per-item definition the compiler cannot hoist out of the item loop)
      asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][tt][r] = zero;
    }

    int buf = gbuf;                               // stage of chunk c
    for (int c = 0; c < nchunks; ++c) {
      const bool more = c + 1 < nchunks, more2 = c + 2 < nchunks;
      int buf2 = buf + 2; if (buf2 >= NBUF) buf2 -= NBUF;
      const uint4* tile_hi = lds + buf * 2 * STAGE;
      const uint4* tile_lo = tile_hi + STAGE;
      half8 bh[4], bl[4];
      auto read_b = [&](const uint4* tile, int tap, half8 (&bb)[4]) {
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) bb[tt] = __builtin_bit_cast(half8, tile[bbase + (2 * tt + dy) * HS + dx]);
      };
      read_b(tile_hi, 0, bh);
      read_b(tile_lo, 0, bl);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        f32x4 (&a)[4] = ar[tap];
        // This tap's operands have landed.  Vector memory retires in order; the table counts what is younger than
        // the youngest of the slot's four loads in the issue order below (tools: the generator in DESIGN 3.11):
        // lo operands refilled inside this tap's third MFMA group, hi operands (+ a DMA piece behind taps 6, 7)
        // inside the NEXT tap's first group, tap 8's at its end.
        switch (tap) {
          case 0: asm volatile("s_waitcnt vmcnt(38)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); break;
          case 7: case 8: asm volatile("s_waitcnt vmcnt(34)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); break;
          default: asm volatile("s_waitcnt vmcnt(36)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); break;
        }
        const half8 ah0 = __builtin_bit_cast(half8, a[0]), ah1 = __builtin_bit_cast(half8, a[1]);
        const half8 al0 = __builtin_bit_cast(half8, a[2]), al1 = __builtin_bit_cast(half8, a[3]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {                       // hi x hi; the PREVIOUS tap's hi refill rides along
          acc[0][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh[tt], acc[0][tt], 0, 0, 0);
          acc[1][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh[tt], acc[1][tt], 0, 0, 0);
          if (tap >= 1) {
            __builtin_amdgcn_sched_barrier(0);
            if (tt < 2) ring_load1(ar[tap - 1][tt], c + 1, tap - 1, tt, more);
            if (tt == 2 && tap - 1 >= 9 - NLD) dma_piece(c + 2, buf2, tap - 1 - (9 - NLD), more2);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(SPLIT_ABLATE & 4)) {
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) {                     // lo(filter) x hi
            acc[0][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh[tt], acc[0][tt], 0, 0, 0);
            acc[1][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh[tt], acc[1][tt], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tap + 1 < 9) read_b(tile_hi, tap + 1, bh);        // (the hi operands are free: their MFMAs have issued)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {                       // hi x lo(input); this tap's lo refill rides along
          if (!(SPLIT_ABLATE & 4)) {
            acc[0][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl[tt], acc[0][tt], 0, 0, 0);
            acc[1][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl[tt], acc[1][tt], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (tt < 2) ring_load1(a[2 + tt], c + 1, tap, 2 + tt, more);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (tap + 1 < 9) read_b(tile_lo, tap + 1, bl);
        if (tap == 8) {                                        // no next tap in this chunk to carry them
          ring_load1(a[0], c + 1, 8, 0, more);
          ring_load1(a[1], c + 1, 8, 1, more);
          dma_piece(c + 2, buf2, NLD - 1, more2);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // the halo of chunk c + 1 (requested during chunk c - 1, or by the prologue) has landed: younger are this chunk's
      // 36 ring loads and 6 DMA instructions; this wave's LDS reads are done; after the barrier everybody's are
      asm volatile("s_waitcnt vmcnt(42) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (++buf == NBUF) buf = 0;
    }
    SDBG(itn, 1);
    gbuf = buf;                                   // (= old gbuf + nchunks mod NBUF)
    // Retire the queue before anything else runs: the last chunk's out-of-range refills are still landing (zeros)
    // in ring registers that are dead values for the compiler from here to the prologue -- it reuses them for the
    // next item's address arithmetic, and a late zero then corrupts e.g. a filter-row offset for a whole item (seen:
    // one wave's 32-channel half of an item wrong, irreproducibly, at full size only).
    ring_landed();

    // ---- hand-over: the next item's first loads fly during this item's epilogue.  (Nothing real of THIS item is in
    // flight: the last chunks' refills / pieces were out of range; the zero fill they wrote into the stages the next
    // prologue targets retires before it -- same wave, same slots, in order.)
    const Item done = cur;
    int nit = it + G;
    Item nxt = done;
    while (nit < q.items) { nxt = decode(nit); if (nxt.lv >= 0) break; nit += G; }
    const bool have_next = nit < q.items;
    // what the epilogue needs of the finished item, before the loader's state moves on
    const long long plane = (long long)done.H * done.W;
    const int oc_w = done.ocb + wo * 64;
    unsigned pvo[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int gy = done.y0 + wp * 8 + 2 * tt + (j >> 4), gx = done.x0 + bcol;
      const bool okp = gy < done.H && gx < done.W;
      const long long pix = (long long)gy * done.W + gx;
      pvo[tt] = okp ? (unsigned)((((long long)done.n * M + 4 * h) * plane + pix) * 4) : kOob;
    }
    SDBG(itn, 2);
#if !(SPLIT_ABLATE & 32)
    if (have_next) {
      bind(nxt);
      prologue();
    }
#endif
    SDBG(itn, 3);

    // ---- epilogue: C/D row = (r & 3) + 8 (r >> 2) + 4 h, column = j.  y = acc * 2^(ex - 15) * 2^(ew - 15) + bias.
    {
      const int lv = done.lv;
      // the two exact power-of-two scales as ONE factor when their product is a normal fp32 number (always, short of
      // tensors at the ends of fp32's range), else applied one after the other
      const int e2 = split_exponent(q.amax[lv]) + split_exponent(reinterpret_cast<const unsigned*>(q.w[lv])[0]) - 30;
      const bool one_scale = e2 >= -126 && e2 <= 127;
      const float sc1 = one_scale ? pow2f(e2) : pow2f(split_exponent(q.amax[lv]) - 15);
      const float sc2 = one_scale ? 1.0f : pow2f(split_exponent(reinterpret_cast<const unsigned*>(q.w[lv])[0]) - 15);
      const unsigned ybytes = (unsigned)((long long)done.N * M * plane * 4);
      const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(q.y[lv], ybytes);
      const __amdgpu_buffer_rsrc_t mrs = uniform_rsrc(MASKED ? (const void*)q.aux[lv] : (const void*)q.y[lv], ybytes);
      const float* bias = q.bias[lv];
      const __amdgpu_buffer_rsrc_t brs = uniform_rsrc(bias ? (const void*)bias : (const void*)q.y[lv],
                                                      bias ? (unsigned)M * 4u : 0u);       // no bias: reads 0
      const bool relu = q.relu, sigm = q.sigmoid;
      const bool ragged = (M & 7) != 0;                    // only then can a lane's 4-channel group straddle M
      const bool want_max = q.amax_out != nullptr;
      const int plane4 = (int)plane * 4;
      unsigned vmax = 0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int oc0 = oc_w + i * 32;
        float4 bq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
          bq[g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(brs, (unsigned)h * 16u, (oc0 + 8 * g) * 4, 0));
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          float mk[16];
          if (MASKED) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int ch = oc0 + 8 * (r >> 2) + (r & 3);
              const unsigned vo = (!ragged || ch + 4 * h < M) ? pvo[tt] : kOob;
              mk[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mrs, vo, ch * plane4, 0));
            }
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if (oc0 + 8 * g >= M) continue;               // wave-uniform
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = one_scale ? fmaf(acc[i][tt][4 * g + e], sc1, bq[g][e]) : acc[i][tt][4 * g + e] * sc1 * sc2 + bq[g][e];
              if (relu) v = fmaxf(v, 0.0f);
              if (sigm) v = 1.0f / (1.0f + expf(-v));      // sigmoid_op.cu:25-29
              if (MASKED) v = mk[4 * g + e] > 0.0f ? v : 0.0f;
              const unsigned vo = (!ragged || oc0 + 8 * g + 4 * h + e < M) ? pvo[tt] : kOob;
              if (want_max) {
                const unsigned av = __float_as_uint(v) & 0x7fffffffu;
                vmax = (vo != kOob && av > vmax) ? av : vmax;
              }
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, vo, (oc0 + 8 * g + e) * plane4, 0);
            }
          }
        }
      }
      if (q.amax_out) {                                    // |max| of this level's output, for the next layer's split
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const unsigned other = (unsigned)__shfl_xor((int)vmax, o, 64);
          vmax = vmax > other ? vmax : other;
        }
        if (lane == 0) wg_max[wave] = vmax;
      }
    }
#if SPLIT_ABLATE & 32          // debug: no overlap -- the next item's first loads are requested after the epilogue
    if (have_next) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      bind(nxt);
      prologue();
    }
#endif
    if (!have_next) {
      if (q.amax_out) {
        __builtin_amdgcn_s_barrier();
        if (tid == 0) {
          unsigned m = wg_max[0];
          for (int w = 1; w < kThreads / 64; ++w) m = m > wg_max[w] ? m : wg_max[w];
          if (m) atomicMax(q.amax_out + done.lv, m);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
    SDBG(itn, 4);
    ring_landed();
    SDBG(itn, 5);
    __builtin_amdgcn_s_barrier();
    SDBG(itn, 6);
#ifdef SPLIT_TIMELINE
    ++itn;
#endif
    if (q.amax_out && tid == 0) {
      unsigned m = wg_max[0];
      for (int w = 1; w < kThreads / 64; ++w) m = m > wg_max[w] ? m : wg_max[w];
      if (m) atomicMax(q.amax_out + done.lv, m);
    }
    it = nit;
    cur = nxt;
  }
}

struct Plan {
  size_t amax_bytes, total_bytes;
  size_t plane_off[kMaxLv];       // bytes from the workspace base
  long long slots[kMaxLv];        // N * CB * H * W
};
int make_plan(const ssad_conv_level* lv, int n_levels, int Cin, Plan* p) {
  if (!lv || n_levels < 1 || n_levels > kMaxLv || Cin <= 0) return SSAD_E_BADARG;
  const int CB = (Cin + 7) >> 3;
  p->amax_bytes = 256;
  size_t off = p->amax_bytes;
  for (int l = 0; l < n_levels; ++l) {
    if (lv[l].N < 0 || lv[l].H < 0 || lv[l].W < 0) return SSAD_E_BADARG;
    const long long slots = (long long)lv[l].N * CB * lv[l].H * lv[l].W;
    if (slots * 32 >= (1LL << 32)) return SSAD_E_BADARG;          // 32-bit buffer offsets over both planes
    p->slots[l] = slots;
    p->plane_off[l] = off;
    off += (size_t)slots * 32;
    off = (off + 255) & ~(size_t)255;
  }
  p->total_bytes = off;
  return 0;
}

}  // namespace

#ifdef SPLIT_TIMELINE
extern "C" __attribute__((visibility("default"))) int ssad_split_dbg_read(void* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_split_dbg), sizeof(g_split_dbg));
}
#endif

extern "C" {

size_t ssad_conv_split_filter_floats(int M, int K) {
  return (size_t)HDR + 2 * (size_t)9 * ((K + 7) >> 3) * M * 4 + 64;
}

int ssad_conv_split_pack_filters(const ssad_pack_entry* entries_host, int n_entries, ssad_stream_t stream) {
  if (n_entries < 0 || (n_entries > 0 && !entries_host)) return SSAD_E_BADARG;
  for (int base = 0; base < n_entries; base += SSAD_MAX_PACK_ENTRIES) {
    const int cnt = n_entries - base < SSAD_MAX_PACK_ENTRIES ? n_entries - base : SSAD_MAX_PACK_ENTRIES;
    FPackTable t;
    long long smax = 0;
    bool any_dgrad = false;
    for (int i = 0; i < cnt; ++i) {
      const ssad_pack_entry& e = entries_host[base + i];
      if (e.Cout <= 0 || e.Cin <= 0 || !e.w || (!e.packed_fwd && !e.packed_dgrad)) return SSAD_E_BADARG;
      t.e[i] = e;
      const long long sf = 9LL * ((e.Cin + 7) >> 3) * e.Cout, sd = 9LL * ((e.Cout + 7) >> 3) * e.Cin;
      smax = sf > smax ? sf : smax;
      if (e.packed_dgrad) { smax = sd > smax ? sd : smax; any_dgrad = true; }
    }
    for (int i = cnt; i < SSAD_MAX_PACK_ENTRIES; ++i) t.e[i] = ssad_pack_entry{};
    hipLaunchKernelGGL(split_filter_zero_kernel, dim3((unsigned)cnt), dim3(64), 0, (hipStream_t)stream, t);
    hipLaunchKernelGGL(split_filter_amax_kernel, dim3(64u, (unsigned)cnt), dim3(kThreads), 0, (hipStream_t)stream, t);
    long long bx = (smax + kThreads - 1) / kThreads;
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(split_filter_pack_kernel, dim3((unsigned)bx, (unsigned)cnt, any_dgrad ? 2u : 1u), dim3(kThreads), 0,
                       (hipStream_t)stream, t);
  }
  return (int)hipGetLastError();
}

size_t ssad_conv3x3_split_workspace_bytes(const ssad_conv_level* lv, int n_levels, int Cin) {
  Plan p;
  if (make_plan(lv, n_levels, Cin, &p)) return 0;
  return p.total_bytes;
}

int ssad_conv3x3_forward_split(const ssad_conv_level* lv, int n_levels, const float* packed, const float* bias,
                               int Cout, int Cin, int flags, void* workspace, size_t workspace_bytes,
                               const unsigned* amax_in, unsigned* amax_out, ssad_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  Plan p;
  const int rc = make_plan(lv, n_levels, Cin, &p);
  if (rc) return rc;
  if (Cout <= 0) return SSAD_E_BADARG;
  if ((flags & SSAD_CONV_MASK_AUX) && (flags & SSAD_CONV_SIGMOID)) return SSAD_E_BADARG;
  if (!workspace || workspace_bytes < p.total_bytes) return SSAD_E_WORKSPACE;
  AmaxTable at;
  ActTable pt;
  SLevels q;
  at.count = pt.count = 0;
  at.amax = (unsigned*)workspace;
  pt.amax = q.amax = (const unsigned*)workspace;
  q.amax_out = amax_out;
  pt.C = Cin;
  int nl = 0, ablocks = 0, pblocks = 0, tiles = 0;
  for (int l = 0; l < n_levels; ++l) {
    const float* pk = lv[l].packed ? lv[l].packed : packed;
    if (!pk) return SSAD_E_BADARG;
    if ((flags & SSAD_CONV_MASK_AUX) && !lv[l].aux) return SSAD_E_BADARG;
    const long long px = (long long)lv[l].N * lv[l].H * lv[l].W;
    if (px * (Cin > Cout ? Cin : Cout) >= (1LL << 29)) return SSAD_E_BADARG;
    if (px == 0) continue;
    const long long n = px * Cin;
    at.x[nl] = pt.x[nl] = lv[l].x;
    at.n[nl] = n;
    at.block_start[nl] = ablocks;
    long long nb = (n / 4 + kThreads * 32 - 1) / (kThreads * 32);     // >= 32 x 16 bytes per thread, one atomic per block
    ablocks += (int)(nb < 1 ? 1 : nb > 512 ? 512 : nb);
    pt.planes[nl] = (uint4*)((char*)workspace + p.plane_off[l]);
    pt.N[nl] = lv[l].N;
    pt.plane[nl] = (long long)lv[l].H * lv[l].W;
    pt.block_start[nl] = pblocks;
    // a tensor read by several problems of the launch (the cls and bbox towers' first layer share the FPN levels) is
    // split once: the later problems read the first one's planes (an entry without blocks is skipped by the pass)
    int same = -1;
    for (int j = 0; j < nl && same < 0; ++j)
      if (at.x[j] == lv[l].x && q.N[j] == lv[l].N && q.H[j] == lv[l].H && q.W[j] == lv[l].W) same = j;
    if (same >= 0) pt.planes[nl] = (uint4*)q.x[same];
    else pblocks += (int)((p.slots[l] + kThreads - 1) / kThreads);
    q.x[nl] = (const uint4*)pt.planes[nl];
    q.y[nl] = lv[l].y;
    q.aux[nl] = lv[l].aux;
    q.w[nl] = pk;
    q.bias[nl] = lv[l].packed ? lv[l].bias : bias;
    q.N[nl] = lv[l].N; q.H[nl] = lv[l].H; q.W[nl] = lv[l].W;
    q.tile0[nl] = tiles;
    tiles += lv[l].N * cdiv(lv[l].H, TS) * cdiv(lv[l].W, TS);
    ++nl;
  }
  if (nl == 0) return 0;
  at.block_start[nl] = ablocks;
  pt.block_start[nl] = pblocks;
  q.tile0[nl] = tiles;
  at.count = pt.count = q.n_levels = nl;
  for (int l = nl; l < kMaxLv; ++l) {
    at.x[l] = pt.x[l] = nullptr; at.n[l] = 0; pt.planes[l] = nullptr; pt.N[l] = 0; pt.plane[l] = 0;
    q.x[l] = nullptr; q.y[l] = nullptr; q.aux[l] = nullptr; q.w[l] = nullptr; q.bias[l] = nullptr;
    q.N[l] = q.H[l] = q.W[l] = 0;
    if (l > nl) { at.block_start[l] = ablocks; pt.block_start[l] = pblocks; q.tile0[l] = tiles; }
  }
  q.C = Cin; q.M = Cout;
  q.relu = (flags & SSAD_CONV_RELU) ? 1 : 0;
  q.sigmoid = (flags & SSAD_CONV_SIGMOID) ? 1 : 0;
  q.mblocks = cdiv(Cout, MT);
  if ((amax_in || amax_out) && nl != n_levels) return SSAD_E_BADARG;      // word l belongs to level l: no empty levels then
  if (amax_in) {
    pt.amax = q.amax = amax_in;
  } else {
    (void)hipMemsetAsync(workspace, 0, p.amax_bytes, stream);
    hipLaunchKernelGGL(split_absmax_kernel, dim3((unsigned)ablocks), dim3(kThreads), 0, stream, at);
  }
  hipLaunchKernelGGL(split_pack_act_kernel, dim3((unsigned)pblocks), dim3(kThreads), 0, stream, pt);
  q.items = cdiv(tiles, 8) * 8 * q.mblocks;
  const int cus = ssad_cu_count();
  const unsigned grid = (unsigned)(q.items < cus ? q.items : cus);
  if (flags & SSAD_CONV_MASK_AUX)
    hipLaunchKernelGGL((conv3x3_split_kernel<true>), dim3(grid), dim3(kThreads), 0, stream, q);
  else
    hipLaunchKernelGGL((conv3x3_split_kernel<false>), dim3(grid), dim3(kThreads), 0, stream, q);
  return (int)hipGetLastError();
}

/* |max| of the tensors of a level table into caller-owned words (word k = problem k; the caller zeroes them -- one
 * fill per step for a whole program's table).  field 0: lv[k].x, 1: lv[k].aux; `channels` = their channel count. */
int ssad_split_absmax_levels(const ssad_conv_level* lv, int n, int channels, int field, unsigned* words,
                             ssad_stream_t stream_) {
  if (!lv || !words || n < 1 || n > kMaxLv || channels <= 0) return SSAD_E_BADARG;
  AmaxTable at;
  int blocks = 0;
  for (int l = 0; l < kMaxLv; ++l) {
    at.x[l] = nullptr; at.n[l] = 0; at.block_start[l] = blocks;
    if (l >= n) continue;
    const long long cnt = (long long)lv[l].N * lv[l].H * lv[l].W * channels;
    at.x[l] = field ? lv[l].aux : lv[l].x;
    at.n[l] = cnt;
    if (cnt && !at.x[l]) return SSAD_E_BADARG;
    long long nb = (cnt / 4 + kThreads * 32 - 1) / (kThreads * 32);
    blocks += (int)(nb < 1 ? 1 : nb > 512 ? 512 : nb);
  }
  at.block_start[kMaxLv] = blocks;
  at.count = n;
  at.amax = words;
  hipLaunchKernelGGL(split_absmax_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream_, at);
  return (int)hipGetLastError();
}

}  // extern "C"
