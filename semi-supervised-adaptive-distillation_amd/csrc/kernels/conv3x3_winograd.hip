// conv3x3_winograd.hip -- Winograd F(2x2, 3x3) engine for the subnet
// convolutions (forward and data gradient), fp32 on the gfx950 matrix cores.
//
// Same operator contract as conv3x3.hip (3x3, stride 1, pad 1, NCHW fp32;
// caffe2/operators/conv_op_cudnn.cc:567-617, :1044-1058) -- cuDNN itself picks
// Winograd for these shapes -- but 2.25x fewer multiplies than the direct
// form:   Y = A^T [ (G g G^T) (.) (B^T d B) ] A   per 2x2 output tile, with the
// 16 element-wise products summed over input channels, i.e. 16 independent
// [Cout x Cin] x [Cin x tiles] GEMMs that run on v_mfma_f32_16x16x4_f32.
//
// The data layout (8 wavefronts = 128 output channels x one
// 8x16-pixel output patch = 4 x 8 tiles; per KC-channel chunk the 10x18 raw
// patch goes to LDS, B^T d B turns it into V[xi][channel][tile], and each wave
// (16 output channels x 32 tiles x 16 xi = 128 accumulator VGPRs) issues 8*KC
// MFMAs whose A operand is a linear 16-byte-per-lane stream from the
// pre-packed filter and whose B operands are `ds_read_b64 base+imm` of V):
//   wino_conv_z_kernel  persistent workgroups, LDS-DMA staging, transform / staging / tile
//                       bookkeeping threaded through the MFMA steps (see its header).
//                       (rounds 1-5 kept a non-persistent one-workgroup-per-tile kernel beside it
//                       as a reference implementation; retired in round 6 with its switch)
// The epilogue applies A^T M A lane-locally (the 16 xi accumulators of one
// (channel, tile) sit in the same lane/register slot), then bias / ReLU /
// Sigmoid / ReLU-gradient mask as the direct kernel.
//
// fp32 Winograd F(2,3) keeps ~1e-6 relative accuracy (inputs are only added /
// subtracted, filters scaled by 1/2, 1/4); parity tests bound it at the same
// 1e-4 the direct kernel is held to.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <map>
#include <mutex>
#include <type_traits>

#include "conv_internal.h"
#include "ssad_kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef WINO_KC
#define WINO_KC 16
#endif
constexpr int KC = WINO_KC;      // input channels per chunk (multiple of 8)
constexpr int KS = KC / 4;       // MFMA k-steps (4 channels each) per chunk
constexpr int STEPS = KS * 4;    // A-stream float4 per lane per chunk
constexpr int PR = 8, PC = 16;   // output patch rows / cols (4 x 8 tiles of 2x2)
constexpr int RP = 20;           // raw LDS row pitch (18 used)
constexpr int RS = (PR + 2) * RP;        // raw floats per channel
constexpr int RAW = KC * RS;             // raw floats per buffer
constexpr int VP = 32;                   // V channel pitch: 32 tiles stored as [tile&15][tile>>4]
constexpr int VBUF = 16 * KC * VP;       // transformed floats per buffer
constexpr int kBlock = 512;
constexpr int BM = 128;                  // output channels per workgroup

__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

using ssad_dev::uniform_rsrc;

// ---------------------------------------------------------------------------
// Filter transform + packing:  U = G g G^T, stored in MFMA A-operand order
//   packed[mt][chunk][ks][xq][lane][xr],  lane = k*16 + i:
//     U[xi = 4*xq + xr][out = mt*16 + i][in = chunk*KC + ks*4 + k]
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wino_u(const float* g, int xi) {
  // rows of G g: r0 = g0, r1 = (g0+g1+g2)/2, r2 = (g0-g1+g2)/2, r3 = g2
  const int a = xi >> 2, b = xi & 3;
  float r[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
    r[j] = a == 0 ? g0 : a == 1 ? 0.5f * (g0 + g1 + g2) : a == 2 ? 0.5f * (g0 - g1 + g2) : g2;
  }
  return b == 0 ? r[0] : b == 1 ? 0.5f * (r[0] + r[1] + r[2])
                       : b == 2 ? 0.5f * (r[0] - r[1] + r[2]) : r[2];
}

__device__ __forceinline__ void wino_pack_body(const float* __restrict__ w, int Cout, int Cin,
                                               float* __restrict__ pf, float* __restrict__ pd,
                                               long long tid) {
  for (int pass = 0; pass < 2; ++pass) {
    float* dst = pass == 0 ? pf : pd;
    if (!dst) continue;
    const int M = pass == 0 ? Cout : Cin, K = pass == 0 ? Cin : Cout;
    const int mtiles = cdiv(M, 16), chunks = cdiv(K, KC);
    const long long total = (long long)mtiles * chunks * STEPS * 256;
    if (tid >= total + 1024) continue;
    float v = 0.0f;
    if (tid < total) {
      const int xr = tid & 3, lane = (tid >> 2) & 63, xq = (tid >> 8) & 3;
      long long r = tid >> 10;
      const int ks = (int)(r % KS); r /= KS;
      const int chunk = (int)(r % chunks), mt = (int)(r / chunks);
      const int out = mt * 16 + (lane & 15), in = chunk * KC + ks * 4 + (lane >> 4);
      if (out < M && in < K) {
        float g[9];
#pragma unroll
        for (int t = 0; t < 9; ++t)
          g[t] = pass == 0 ? w[((long long)out * Cin + in) * 9 + t]
                           : w[((long long)in * Cin + out) * 9 + (8 - t)];   // flipped, transposed
        v = wino_u(g, xq * 4 + xr);
      }
    }
    dst[tid] = v;
  }
}

__global__ void wino_pack_kernel(const float* __restrict__ w, int Cout, int Cin,
                                 float* __restrict__ pf, float* __restrict__ pd) {
  wino_pack_body(w, Cout, Cin, pf, pd, (long long)blockIdx.x * blockDim.x + threadIdx.x);
}

// every filter of a model in ONE launch (blockIdx.y = filter): the training step repacks all
// of its filters after each parameter update
struct PackTable {
  ssad_pack_entry e[SSAD_MAX_PACK_ENTRIES];
};
__global__ void wino_pack_multi_kernel(const PackTable t) {
  const ssad_pack_entry& e = t.e[blockIdx.y];
  const size_t nf = e.packed_fwd ? (size_t)cdiv(e.Cout, 16) * cdiv(e.Cin, KC) * STEPS * 256 + 1024 : 0;
  const size_t nd = e.packed_dgrad ? (size_t)cdiv(e.Cin, 16) * cdiv(e.Cout, KC) * STEPS * 256 + 1024 : 0;
  const long long n = (long long)(nf > nd ? nf : nd);
  for (long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x; tid < n;
       tid += (long long)gridDim.x * blockDim.x)
    wino_pack_body(e.w, e.Cout, e.Cin, e.packed_fwd, e.packed_dgrad, tid);
}

// ---------------------------------------------------------------------------
// Convolution
// ---------------------------------------------------------------------------
struct WLevel {
  const float* x;
  float* y;
  const float* aux;
  const float* packed;
  const float* bias;
  int N, H, W;
  int tiles_x, tiles_y;
  int block_start;
  // persistent variant: 8 x 8-pixel sub-patches, two per work item
  int sub_x, sub_y;       // sub-patches per row / column of one image
  int pair_start;         // first work item of this level
};


struct WArgs {
  WLevel lv[SSAD_MAX_CONV_PROBLEMS];
  int n_levels;
  int M, K, chunks, flags;
  int patches, mblocks;   // persistent variant: tiles = patches x mblocks
  int xcd_group;          // consecutive tiles per XCD inside a round of the grid (1 = round robin)
  int items;              // persistent variant: the items [0, items) of the patches x mblocks this launch walks
  // split tail (wino_conv_z_kernel<.., .., true>, a launch of its own): the items [full_items, patches x mblocks)
  // of the last, partial round of the grid, cut along the reduction into `split_parts` units each, one per
  // workgroup (see the kernel's header)
  int full_items;
  int split_parts;        // units per tail item; divides `chunks`
  float* split_ws;        // [grid][BM * 128] partial outputs, one slot per unit
  unsigned* split_tickets;   // [grid] arrival counters, zero between launches
};

constexpr int NRAW = 3;
constexpr unsigned kOOBOff = 0x80000000u;

// WINO_ABLATE (debug builds only, results are wrong): 1 = no raw-patch DMA in the loop, 2 = no input transform,
// 4 = filter operand ring not refilled, 8 = B operands not re-read, 16 = no epilogue, 32 = no per-chunk barrier
#ifndef WINO_ABLATE
#define WINO_ABLATE 0
#endif
#ifndef WINO_DMA_SPREAD
#define WINO_DMA_SPREAD 1
#endif
#ifdef WINO_TIMELINE
__device__ unsigned long long g_dbg[2][64][8];
__device__ unsigned long long g_wv[8][64][2];     // per wave of the stamped workgroup: steps done, barrier passed
#define DBG(role, it, k) \
  if (dbg_on && (it) < 64) g_dbg[role][it][k] = __builtin_readcyclecounter()
#define DBGW(it, k) \
  if (blockIdx.x == 3 && lane == 0 && (it) < 64) g_wv[wave][it][k] = __builtin_readcyclecounter()
#else
#define DBG(role, it, k)
#define DBGW(it, k)
#endif

// One work item of the persistent kernel: two 8 x 8-pixel sub-patches of one level (4 x 4 tiles each = the two
// tile groups of the MFMAs), each with its own image and origin; y0 = kNoSub marks an absent partner.
struct WTile {
  int l, mb;
  int n[2], y0[2], x0[2];
};
constexpr int SP = 8;                     // sub-patch edge (output pixels)
constexpr int kNoSub = 1 << 24;

// ---------------------------------------------------------------------------
// Variant Z: persistent, 8 symmetric waves, LDS-DMA staging, everything
// threaded through the MFMA steps.
//
// Lessons from the variants above (cycle stamps, tower layer, bs 16):
//  * the B^T d B burst is LDS-throughput bound (~1700 of 10800 cycles per
//    chunk) and helper waves cannot do it (no VALU issue next to MFMA waves);
//  * a wave alone on a SIMD is latency-bound on the filter stream, so the
//    two waves of a SIMD must both stay in their MFMA steps all the time;
//  * vmcnt retires in order, so HBM staging loads stall a wave's next filter
//    operand -- unless that operand was requested BEFORE the staging loads.
// So: no helper waves, <= 256 VGPRs (2 waves/SIMD as before), and per chunk s
//   step 0      : each wave issues its 7 `buffer_load_dword ... lds` of chunk
//                 s+3 (no VGPR data, no ds_write); the filter operands of
//                 steps 0-7 are already in flight / in registers (ring of 8
//                 float4, refilled right after use, i.e. 8 steps ahead);
//   steps 4r+1  : 4 ds_read_b64 of the raw patch of chunk s+1 (round r),
//   steps 4r+3  : 8 VALU + 4 ds_write_b32 into V[s+1];
//   every step  : 8 MFMAs, B operands single-buffered (ds_read_b64 reissued
//                 into the registers an MFMA pair just consumed);
//   one barrier.
// The raw data lives in LDS as channel pairs with row pitch 40
// (addr = (c>>1)*400 + r*40 + (c&1)*20 + q): q < 10 is the 10 x 10 window of the work item's first 8 x 8
// sub-patch, q >= 10 that of the second (round 2 staged ONE 8 x 16 patch, 18 of the 20 columns: on a
// 40 x 56 map 12.5 % of the MFMAs then ran outside the image, 6 % over the five levels of a 600 px
// batch; 8 x 8 sub-patches tile 40 x 56 exactly, pairs may span images).
// Tiles of a workgroup are decoded once, in parallel, into an LDS list.
// ---------------------------------------------------------------------------
constexpr int ZP = 40;                    // raw row pitch (two channels side by side)
constexpr int ZCP = (PR + 2) * ZP;        // 400 floats per channel pair
constexpr int ZRAW = (KC / 2) * ZCP;      // 3200 floats = 50 wave-loads
constexpr int ZL = 7;                     // wave-loads per wave per chunk (8 x 7 = 56 >= 50)
constexpr int ZRAWP = 8 * ZL * 64;        // padded raw buffer (dummy loads land in the pad)
constexpr int ZNT = 256;                  // tiles per workgroup in the LDS list
constexpr int AD = 8;                     // filter operand ring depth (steps)
static_assert(ZRAW <= ZRAWP && STEPS == 16 && AD == 8, "variant Z is written for KC = 16");

// PAIRS: the work item is two 8 x 8 sub-patches (true) or one 8 x 16 patch (false: the round-2 geometry; on
// maps that 8 x 16 patches tile exactly it is ~4 % faster -- 18 instead of 2 x 10 staged columns -- so the
// launcher takes it whenever the sub-patches would not save at least 2 % of the computed pixels).
// NHALF (Cout <= 64: res2 of the backbones, bbox_pred): a 128-channel block would leave waves 4-7 (and their
// half of every SIMD's MFMA issue) idle; instead waves w and w + 4 share output channels 16 (w & 3) .. and
// split the work item's two tile groups between them (8 accumulator quads per wave instead of 16).
// SPLIT (round 5): the tail of the persistent grid.  `total` equal-cost items on G workgroups take ceil(total / G)
// rounds; the backbones' mid-sized layers pay for that (res4, 256 -> 256 at 40 x 56 x 16: 560 items = 2.19 rounds take
// 3; res5: 1.5 take 2 -- measured 0.55 of the matrix peak where the towers' exact 39 + 9 rounds reach 0.75).  With
// the split the T = total mod G items of the partial round are cut along the REDUCTION into p <= G / T units of
// chunks / p input-channel chunks each (p a power of two dividing `chunks`), and that round is a LAUNCH OF ITS OWN
// of this instantiation, T p workgroups of one unit each, behind the launch of the full rounds: it costs ~1 / p of
// a round.  (A unit as the last item of the main kernel's workgroups was the first form: the extra scalar state
// pushed the kernel from 250 to 256 VGPRs with spills -- among them ring registers in flight, which
// tools/isa_lint.py caught; a workgroup that runs one unit and nothing else needs no tile hand-over at all.)
// A unit writes its partial result -- already through A^T . A, 4 floats per (channel, tile), no bias -- to its slot
// of the launcher's scratch with agent-scope write-through stores, bumps the item's arrival counter, and the LAST
// unit of an item to arrive sums the p slots in unit order (so the bits do not depend on who is last), adds the
// bias and applies ReLU / Sigmoid / the ReluGradient mask as the ordinary epilogue does.  Nobody waits for anybody:
// all units of an item are resident in the same round.  Cross-XCD visibility: sc1 (write-through) stores, every
// wave's s_waitcnt vmcnt(0), barrier, one relaxed agent-scope arrival; sc1 loads in the last arrival -- the
// counter form of the hand-off in cdna_hip_programming.md 5.4 / MI355X_MICROARCH.md "valid forms".
template <bool PAIRS, bool NHALF, bool SPLIT = false>
__global__ __launch_bounds__(kBlock, 1) void wino_conv_z_kernel(const WArgs args) {
  static_assert(!(SPLIT && NHALF), "the split tail is instantiated for 128-channel blocks");
  constexpr int NG = NHALF ? 1 : 2;        // tile groups per wave
  __shared__ float raw[NRAW * ZRAWP];
  __shared__ float vbuf[2 * VBUF];
  __shared__ int lv_start[32], lv_tx[32], lv_per[32];
  __shared__ int trec[ZNT * 8];
  __shared__ unsigned zvoff[8 * ZL * 64];   // per wave, per wave-load, per lane

  const int K = args.K, M = args.M;
  const int chunks = args.chunks;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wv = NHALF ? (wave & 3) : wave;          // this wave's 16-channel slice of the block
  const int g0 = NHALF ? (wave >> 2) : 0;            // its first tile group
  const int total = args.items;
  // Which tiles a workgroup walks: round i of the grid covers tiles [i G, (i + 1) G); inside a round
  // the workgroups of ONE XCD (ids b, b + 8, ...: they share an L2) take runs of `xg` CONTIGUOUS
  // tiles, i.e. patches that are neighbours in x (and, for long runs, in y).  A patch row is 18
  // floats = 72 bytes at a 64-byte stride: it straddles two 128-byte lines that its x neighbours use
  // too, and 2 of its 10 rows belong to the patch above.  With neighbours on different XCDs (round 2:
  // xg = 1, tile = id + i G) every XCD fetched its own copy of those lines -- the counters showed
  // 60 M line fills per tower launch where the input is 6 M lines (profiles/r03_tcc_classes.md).
  const int G = (int)gridDim.x;
  int slot = (int)blockIdx.x;
  {
    const int xg = args.xcd_group;
    if (xg > 1 && (G & 7) == 0 && ((G >> 3) % xg) == 0) {
      const int r = slot >> 3, x = slot & 7;
      slot = (r / xg) * (8 * xg) + x * xg + (r % xg);
    }
  }
  // SPLIT: the launch IS the partial round -- workgroup b runs unit b % sp of tail item b / sp, i.e. the
  // `nch` = chunks / sp input-channel chunks from chunk t_c0 on, and nothing else
  const int sp = SPLIT ? args.split_parts : 1;
  const int t_item = SPLIT ? (int)blockIdx.x / sp : 0;
  const int nch = SPLIT ? __builtin_amdgcn_readfirstlane(chunks / sp) : chunks;       // chunks an item runs over here
  const int t_c0 = SPLIT ? __builtin_amdgcn_readfirstlane(((int)blockIdx.x - t_item * sp) * nch) : 0;
  const int my_n = SPLIT ? 1 : (total > slot ? (total - slot + G - 1) / G : 0);
  const int S = my_n * nch;     // flattened (tile, chunk) sequence length
  // static priority for the second-dispatched half of the workgroup (MI355X_MICROARCH.md, two waves
  // per SIMD, item 4): waves 4-7 lose VALU arbitration to their older SIMD partners on every chunk;
  // one s_setprio for them, no per-segment flips

  // ---- level tables and this workgroup's work items ----
  if (tid < 32) {
    int v = 0x7fffffff, tx = 1, per = 1;
#pragma unroll
    for (int i = 0; i < SSAD_MAX_CONV_PROBLEMS; ++i)
      if (tid == i && i < args.n_levels) {
        v = PAIRS ? args.lv[i].pair_start : args.lv[i].block_start;
        tx = PAIRS ? args.lv[i].sub_x : args.lv[i].tiles_x;
        per = PAIRS ? args.lv[i].sub_x * args.lv[i].sub_y : args.lv[i].tiles_x * args.lv[i].tiles_y;
      }
    lv_start[tid] = v; lv_tx[tid] = tx; lv_per[tid] = per;
  }
  __syncthreads();
  for (int i = tid; i < my_n; i += kBlock) {
    const int t = SPLIT ? args.full_items + t_item : slot + i * G;
    const int mb = t / args.patches;
    int pid = t - mb * args.patches;
    int l = -1;
    for (int k = 0; k < args.n_levels; ++k) l += pid >= lv_start[k];
    pid -= lv_start[l];
    const int per = lv_per[l], tx = lv_tx[l];
    int* r = trec + i * 8;
    r[0] = l; r[1] = mb;
    for (int h = 0; h < 2; ++h) {
      int sid = PAIRS ? 2 * pid + h : pid;
      const int n = sid / per;
      sid -= n * per;
      const int sy = sid / tx, sx = sid - sy * tx;
      const bool there = n < args.lv[l].N;
      r[2 + 3 * h] = there ? n : 0;
      r[3 + 3 * h] = there ? sy * SP : kNoSub;
      r[4 + 3 * h] = PAIRS ? sx * SP : sx * PC + h * SP;     // (one patch = its left and right half)
    }
  }
  __syncthreads();
  auto get_tile = [&](int i) {
    const int* r = trec + i * 8;
    WTile o;
    o.l = __builtin_amdgcn_readfirstlane(r[0]);
    o.mb = __builtin_amdgcn_readfirstlane(r[1]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      o.n[h] = __builtin_amdgcn_readfirstlane(r[2 + 3 * h]);
      o.y0[h] = __builtin_amdgcn_readfirstlane(r[3 + 3 * h]);
      o.x0[h] = __builtin_amdgcn_readfirstlane(r[4 + 3 * h]);
    }
    return o;
  };

  // ---- staging: wave-load k = wave + 8j covers raw slots e = 64k + lane ----
  // The per-tile byte offsets live in LDS (zvoff), not in VGPRs: a spilled
  // offset would be reloaded from scratch with a vmcnt(0) wait in the middle
  // of the filter-operand ring.
  ssad_dev::rsrc_words xrs = ssad_dev::uniform_rsrc_words(args.lv[0].x, 0);
  int chunk_bytes = 0;
  int ld_tile = 0, ld_ch = 0, ld_buf = 0;      // load cursor over (tile, chunk), its raw buffer
  unsigned* myvoff = zvoff + wave * (ZL * 64) + lane;
  const unsigned raw_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)raw;
  // real = false: the same ZL instructions with every lane out of range (zeros into the buffer nobody
  // reads any more): a wave's vector-memory queue then has the SAME shape in every chunk, which is
  // what the counted waits of the filter ring below rely on
  // One chunk's DMA = dma_begin (cursor bookkeeping, once per tile the lane offsets), ZL x dma_issue, dma_end.
  // The main loop spreads the ZL instructions over steps 0..ZL-1 (WINO_DMA_SPREAD, default): issued as one burst
  // at step 0 by all 8 waves at once they held each wave for ~750 cycles (cycle stamps around the block), which
  // an isolated burst does not cost (tools/coissue_probe.hip) -- the texture addresser of the CU is shared.
  bool dma_real = true;
  int dma_soff = 0;
  unsigned dma_dst = 0;
  auto dma_begin = [&](bool real) {
    dma_real = real;
    if (real && ld_ch == 0) {
      const WTile Tt = get_tile(ld_tile);
      const WLevel& L = args.lv[Tt.l];
      const int H = L.H, W = L.W, HW = H * W;
      // descriptor over the level's whole batch: the two sub-patches may sit in different images
      xrs = ssad_dev::uniform_rsrc_words(L.x, (unsigned)((long long)L.N * K * HW * 4));
      chunk_bytes = KC * HW * 4;
#pragma unroll 1
      for (int j = 0; j < ZL; ++j) {
        const int e = (wave + 8 * j) * 64 + lane;
        const int p = e / ZCP, rem = e - p * ZCP;
        const int r = rem / ZP, cq = rem - r * ZP;
        const int hi = cq >= ZP / 2 ? 1 : 0;
        const int q = cq - hi * (ZP / 2);                 // 0..19: sub-patch q / 10, column q % 10
        const int sb = PAIRS && q >= SP + 2 ? 1 : 0;
        const int gy = (sb ? Tt.y0[1] : Tt.y0[0]) - 1 + r, gx = (sb ? Tt.x0[1] : Tt.x0[0]) - 1 + q - sb * (SP + 2);
        const int n = sb ? Tt.n[1] : Tt.n[0];
        const bool ok = (e < ZRAW) & (PAIRS || q < PC + 2) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
        myvoff[j * 64] = ok ? (unsigned)(((n * K + 2 * p + hi) * HW + gy * W + gx) * 4) : kOOBOff;
      }
    }
    dma_soff = __builtin_amdgcn_readfirstlane(real ? (ld_ch + t_c0) * chunk_bytes : 0);
    dma_dst = __builtin_amdgcn_readfirstlane(raw_lds + (unsigned)(ld_buf * ZRAWP + wave * 64) * 4u);
  };
  // real = false: the same instruction with every lane out of range (zeros into the buffer nobody reads any
  // more): a wave's vector-memory queue then has the SAME shape in every chunk, which is what the counted
  // waits of the filter ring below rely on
  auto dma_offset = [&](int j) { return dma_real ? myvoff[j * 64] : kOOBOff; };
  // inline assembly, not __builtin_amdgcn_raw_ptr_buffer_load_lds: behind the builtin hipcc makes the next LDS
  // read of the kernel (the transform's raw-patch read, one step later) wait for the DMA itself
  // (`s_waitcnt vmcnt(1)` in the round-2 ISA); see conv_internal.h
  auto dma_issue = [&](int j, unsigned vo) { ssad_dev::lds_dma<4>(xrs, dma_dst + j * 2048, vo, dma_soff); };
  auto dma_end = [&]() {
    if (dma_real) {
      if (++ld_ch == nch) { ld_ch = 0; ++ld_tile; }
      if (++ld_buf == NRAW) ld_buf = 0;
    }
  };
  auto dma_next = [&](bool real) {
    dma_begin(real);
    unsigned vo[ZL];
#pragma unroll
    for (int j = 0; j < ZL; ++j) vo[j] = dma_offset(j);
#pragma unroll
    for (int j = 0; j < ZL; ++j) dma_issue(j, vo[j]);
    dma_end();
  };

  // ---- transform: work item = (tile, channel, row a); wave w owns row a = w & 3 for
  //      channels rnd*4 + (w>>2)*2 + (lane>>5);  row a of B^T d is dA + sg * dB:
  //      a=0: d0-d2   a=1: d1+d2   a=2: d2-d1   a=3: d1-d3
  const int t_row = wave & 3;
  const int t_ra = t_row == 0 ? 0 : t_row == 2 ? 2 : 1;
  const int t_rb = t_row == 0 ? 2 : t_row == 1 ? 2 : t_row == 2 ? 1 : 3;
  const float t_sg = t_row == 1 ? 1.0f : -1.0f;
  const int t_tile = lane & 31, t_c = (wave >> 2) * 2 + (lane >> 5);   // channel within a round of 4
  // tile t of the work item: PAIRS: sub-patch t >> 4, row (t & 15) >> 2, column t & 3 of its 4 x 4 tiles;
  // else row t >> 3, column t & 7 of the patch's 4 x 8
  const int t_src = (t_c >> 1) * ZCP + (t_c & 1) * (ZP / 2) +
      (PAIRS ? (2 * ((t_tile & 15) >> 2)) * ZP + 2 * (t_tile & 3) + (t_tile >> 4) * (SP + 2)
             : (2 * (t_tile >> 3)) * ZP + 2 * (t_tile & 7));
  // V row = 32 tiles of one (position, channel).  128-channel blocks: slot (t & 15) * 2 + (t >> 4), i.e. the two tile
  // groups interleaved -- a wave's ds_read_b64 takes both.  NHALF: a wave reads ONE group with ds_read_b32, and
  // stride-2 floats would touch only every other bank (2-way conflicts: 4.2e7 conflict cycles per bbox_pred launch in
  // profiles/r03_pmc_classes.md against 1.9e7 for a tower launch of 30x the work); there the groups are contiguous
  // halves of the row, swapped in channel rows 2, 3 of every k group so that the four rows of an MFMA step (kq = 0..3,
  // pitch 32 floats) cover all 64 banks.  (Round 4; the launch times did not move -- bbox_pred 0.555 ms before and
  // after: a 36-wide layer is bound by the per-patch side jobs, transform and DMA, not by LDS reads.)
  const int t_dst = (t_row * 4 * KC + t_c) * VP +
      (NHALF ? ((((t_tile >> 4) ^ (t_c >> 1)) & 1) * 16 + (t_tile & 15)) : ((t_tile & 15) * 2 + (t_tile >> 4)));
  const int t_oa = t_src + t_ra * ZP, t_ob = t_src + t_rb * ZP;
  auto xf_load = [&](const float* rb0, int rnd, float2 (&d)[4]) {
    d[0] = *reinterpret_cast<const float2*>(rb0 + t_oa + rnd * 2 * ZCP);
    d[1] = *reinterpret_cast<const float2*>(rb0 + t_oa + rnd * 2 * ZCP + 2);
    d[2] = *reinterpret_cast<const float2*>(rb0 + t_ob + rnd * 2 * ZCP);
    d[3] = *reinterpret_cast<const float2*>(rb0 + t_ob + rnd * 2 * ZCP + 2);
  };
  auto xf_store = [&](float* vb0, int rnd, const float2 (&d)[4]) {
    const float t0 = fmaf(t_sg, d[2].x, d[0].x), t1 = fmaf(t_sg, d[2].y, d[0].y);
    const float t2 = fmaf(t_sg, d[3].x, d[1].x), t3 = fmaf(t_sg, d[3].y, d[1].y);
    float* o = vb0 + t_dst + rnd * 4 * VP;
    o[0 * KC * VP] = t0 - t2;
    o[1 * KC * VP] = t1 + t2;
    o[2 * KC * VP] = t2 - t1;
    o[3 * KC * VP] = t1 - t3;
  };
  auto transform = [&](const float* rb0, float* vb0) {
#pragma unroll
    for (int rnd = 0; rnd < KC / 4; ++rnd) {
      float2 d[4];
      xf_load(rb0, rnd, d);
      xf_store(vb0, rnd, d);
    }
  };

  // ---- compute-side per-lane constants ----
#ifdef WINO_TIMELINE
  const bool dbg_on = blockIdx.x == 3 && tid == 0;
#endif
  const int kq = lane >> 4, jn = lane & 15;
  const float* bbase = vbuf + kq * VP + (NHALF ? (((g0 ^ (kq >> 1)) & 1) * 16 + jn) : jn * 2);
  const int mtiles = cdiv(M, 16);
  const int stream_bytes = (mtiles * chunks * STEPS * 256 + 1024) * 4;
  const unsigned a_voff = lane * 16;
  auto stream_off = [&](const WTile& Tt) {
    int mt = Tt.mb * (BM / 16) + wv;
    if (mt >= mtiles) mt = 0;
    return __builtin_amdgcn_readfirstlane(mt * chunks * STEPS * 1024);
  };
  // The filter ring is loaded and awaited by hand.  Vector memory retires in order and hipcc does not
  // count the DMA above, so its own wait in front of a ring slot (vmcnt(7): "the 7 younger ring loads
  // may still fly") made every step that followed a DMA issue wait for one more of the 7 HBM fetches
  // queued BEHIND the slot it needed.  The queue of a wave is, per chunk: R(8) D0 R(9) D1 ... R(14) D6 R(15) ... R(23),
  // R(j) = operand of step j requested at step j - 8, Dj = one wave-load of the raw-patch DMA, issued at the end
  // of step j (burst variant: all seven behind R(8)).  Step j needs R(j): 7 ring loads are younger, plus the
  // D issued since R(j) was: j of them for j < 8, 15 - j after (burst: 7 for j = 1..8).
  //
  // IN-FLIGHT RING REGISTERS AND THE COMPILER (round 4).  An a_load's destination is "defined" for hipcc the moment
  // the statement is issued, long before the data lands; any copy the register allocator makes of it before the
  // counted wait (`"+v"` on the ring slot) copies a register the load has not written yet.  Inside the chunk loop the
  // slots stay in their registers (checked on the built library by tests/test_isa_lint.py).  Across the TILE loop
  // they did not: rounds 1-3 prefetched the next tile's first 8 operands during the last chunk and re-primed the
  // ring of a wave that had sat a tile out in an `if` at the tile's end; the merge of the two definitions made
  // hipcc copy all 32 ring registers aside before the epilogue and back after it, and a prefetch that landed in
  // between (late: HBM contention from another stream) was overwritten by the stale copy -- one wave's 16 channels
  // of one tile wrong, seen only at full size with two streams (tests/test_gpu_full_size.py; 89 of 400 launches of
  // res2's 64 -> 64 layer under load in tools/dbg/r4_wino_stress.py, where NG = 1 halves the time a load has).
  // Now nothing real is in flight when control leaves the chunk loop: the last chunk's steps 8..15 issue their ring
  // loads with every lane out of range (zeros, no memory traffic: the queue keeps the shape the counted waits
  // assume), ALL waves load the next tile's first operands in straight-line code before the epilogue, and a
  // vmcnt(0) tied to the eight slots follows the epilogue (the loads had the epilogue to land).
  auto a_load_at = [&](f32x4& dst, unsigned voff, const ssad_dev::rsrc_words& rs, int soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff));
  };
  auto a_load = [&](f32x4& dst, const ssad_dev::rsrc_words& rs, int soff) { a_load_at(dst, a_voff, rs, soff); };
  auto ring_landed = [&](f32x4 (&r)[AD]) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 :: "memory");
  };
  auto load_bias = [&](const WTile& Tt) {
    const WLevel& L = args.lv[Tt.l];
    f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f};
    if (L.bias) {
      const __amdgpu_buffer_rsrc_t brsrc = uniform_rsrc(L.bias, M * 4);
      b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
          brsrc, (unsigned)(((Tt.mb * (BM / 16) + wv) * 16 + kq * 4) * 4), 0, 0));
    }
    return b;
  };

  // ---- prologue: chunks 0..2 by DMA, chunk 0 transformed, filter ring primed ----
  WTile T = get_tile(0);
  ssad_dev::rsrc_words arsrc = ssad_dev::uniform_rsrc_words(args.lv[T.l].packed, (unsigned)stream_bytes);
  int abase = stream_off(T) + t_c0 * STEPS * 1024;
  f32x4 ar[AD];
#pragma unroll
  for (int k = 0; k < AD; ++k) a_load(ar[k], arsrc, abase + k * 1024);
  f32x4 bv = load_bias(T);
  if (SPLIT) bv = f32x4{0.f, 0.f, 0.f, 0.f};      // a unit's partial sum carries no bias
  dma_next(true);
  if (S > 1) dma_next(true);
  if (S > 2) dma_next(true);
  ring_landed(ar);
  __syncthreads();
  transform(raw, vbuf);
  __syncthreads();

  int s = 0;
  int rbuf = 1;                      // raw buffer holding chunk s+1
  WTile Tn = T;
  ssad_dev::rsrc_words nrsrc = arsrc;
  int nbase = abase;
  for (int i = 0; i < my_n; ++i) {
    const int look = nch > 1 ? 1 : 0;
    const int mt = T.mb * (BM / 16) + wv;
    const bool active = mt < mtiles;
    f32x4 acc[16][NG];
    // bias folded into the accumulators: b * u u^T, u = (1,0,0,-1), A^T u = (1,1)
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
      for (int g = 0; g < NG; ++g)
        acc[x][g] = (x == 0 || x == 15) ? bv : (x == 3 || x == 12) ? -bv : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < nch; ++ch, ++s) {
      DBG(1, s, 0);
      // Side jobs, issued from inside the MFMA steps (active waves) so that they
      // run under MFMAs instead of in front of them:
      //  * chunk s+3 by DMA into the buffer chunk s vacated.  The ring already
      //    holds / has requested the filter operands of the next 8 steps, so
      //    these HBM loads sit behind them in the in-order vmcnt queue;
      //  * once per tile, the next tile's filter stream position.
      auto side_dma = [&]() { if (!(WINO_ABLATE & 1)) dma_next(s + 3 < S); };
      auto side_look = [&]() {
        if (ch == look && i + 1 < my_n) {
          Tn = get_tile(i + 1);
          nrsrc = ssad_dev::uniform_rsrc_words(args.lv[Tn.l].packed, (unsigned)stream_bytes);
          nbase = stream_off(Tn);
        }
      };
      const bool xf = s + 1 < S && !(WINO_ABLATE & 2);
      const float* xsrc = raw + rbuf * ZRAWP;
      float* xdst = vbuf + ((s + 1) & 1) * VBUF;
      DBG(1, s, 1);
      // operands 8 steps ahead: steps 8..15 request those of the NEXT chunk's steps 0..7 -- of this tile only; in
      // the tile's last chunk the same instructions go out with every lane out of range, and the next tile's first
      // operands are requested after the chunk loop (nrsrc / nbase: side_look, chunk `look`)
      const bool last = ch == nch - 1;
      const unsigned tail_voff = last ? kOOBOff : a_voff;
      if (active) {
        const float* vb = bbase + (s & 1) * VBUF;
        // B operands are single-buffered: right after the two MFMAs of an xr
        // pair issue, the same registers take the next step's pair (a 2-step
        // ring was measured: no gain).
        float bc[4][NG];
        auto b_read = [&](float (&b)[NG], const float* src) {
          if (NHALF) {
            b[0] = src[0];
          } else {
            const float2 b2 = *reinterpret_cast<const float2*>(src);
            b[0] = b2.x; b[NG - 1] = b2.y;
          }
        };
#pragma unroll
        for (int xr = 0; xr < 4; ++xr) b_read(bc[xr], vb + (xr * KC) * VP);
        float2 xd[4];
        unsigned dma_vo = kOOBOff;
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {       // step = ks*4 + xq
          const int xq = step & 3;
          const int nks = (step + 1) >> 2, nxq = (step + 1) & 3;
          // this step's operand has landed (see a_load)
          {
            const int younger = (WINO_ABLATE & 1) ? 0
                : WINO_DMA_SPREAD ? (step < AD ? (step < ZL ? step : ZL) : (STEPS - 1 - step < ZL ? STEPS - 1 - step : ZL))
                                  : (step >= 1 && step <= AD ? ZL : 0);
            switch (younger) {      // `step` is a compile-time constant after unrolling: one case survives
              case 0: asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ar[step & (AD - 1)]) : "n"(AD - 1)); break;
              case 1: asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ar[step & (AD - 1)]) : "n"(AD)); break;
              case 2: asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ar[step & (AD - 1)]) : "n"(AD + 1)); break;
              case 3: asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ar[step & (AD - 1)]) : "n"(AD + 2)); break;
              case 4: asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ar[step & (AD - 1)]) : "n"(AD + 3)); break;
              case 5: asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ar[step & (AD - 1)]) : "n"(AD + 4)); break;
              case 6: asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ar[step & (AD - 1)]) : "n"(AD + 5)); break;
              default: asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ar[step & (AD - 1)]) : "n"(AD + 6)); break;
            }
          }
          const f32x4 a0 = ar[step & (AD - 1)];
          const float av[4] = {a0[0], a0[1], a0[2], a0[3]};
#pragma unroll
          for (int xr = 0; xr < 4; ++xr) {
            const int xi = xq * 4 + xr;
#pragma unroll
            for (int g = 0; g < NG; ++g)
              if (!((WINO_ABLATE & 64) && xq == 3))        // 64: a quarter of the MFMAs gone (what F(2x2) x F(4x4) would save)
                acc[xi][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[xr], bc[xr][g], acc[xi][g], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (step < STEPS - 1 && !(WINO_ABLATE & 8)) b_read(bc[xr], vb + ((nxq * 4 + xr) * KC + nks * 4) * VP);
            __builtin_amdgcn_sched_barrier(0);
          }
          // refill the ring slot just consumed with the operand of step + 8
          if (!(WINO_ABLATE & 4)) {
            // past the tile's last operand: the same instruction with every lane out of range (see a_load)
            a_load_at(ar[step & (AD - 1)], step + AD >= STEPS ? tail_voff : a_voff, arsrc,
                      abase + (ch * STEPS + step + AD) * 1024);
          }
          if (xf && (step & 3) == 1) xf_load(xsrc, step >> 2, xd);
          if (xf && (step & 3) == 3) xf_store(xdst, step >> 2, xd);
          if (WINO_DMA_SPREAD && !(WINO_ABLATE & 1)) {
            if (step == 0) { DBG(1, s, 6); dma_begin(s + 3 < S); dma_vo = dma_offset(0); DBG(1, s, 7); }
            if (step < ZL) dma_issue(step, dma_vo);
            if (step + 1 < ZL) dma_vo = dma_offset(step + 1);
            if (step == ZL - 1) dma_end();
          } else if (step == 0) {
            side_dma();
          }
          if (step == 6) side_look();
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        side_dma();
        side_look();
        if (xf) transform(xsrc, xdst);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      DBG(1, s, 2);
      DBGW(s, 0);
      if (++rbuf == NRAW) rbuf = 0;
      // End of chunk s: the LDS traffic of this wave (V writes, operand reads) must be done, and the DMA
      // of chunk s - 1 (the raw patch the NEXT chunk's steps transform) must have landed.  The DMA of chunk
      // s itself feeds the transform two chunks from now and may still fly: with three raw buffers nothing
      // overwrites it before.  Younger than the last DMA instruction of chunk s - 1 are the ring loads of the
      // steps after it (STEPS - ZL; burst variant: 15) and this chunk's STEPS + ZL instructions.  (Round 2 waited for vmcnt(AD): every DMA
      // had to land within its own chunk, ~3.7 k cycles after the request.)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((WINO_ABLATE & 128) ? AD : WINO_DMA_SPREAD ? (STEPS - ZL) + STEPS + ZL : 2 * (STEPS - 1) + 1 + ZL) : "memory");
      if (!(WINO_ABLATE & 32)) __builtin_amdgcn_s_barrier();
      DBG(1, s, 3);
      DBGW(s, 1);
    }
    DBG(1, s - 1, 4);
    // The next tile's first operands and bias fly during the epilogue -- every wave, active or not, and also after
    // the last tile (Tn = T then: a harmless re-read), so that the ring has ONE definition at this point (see a_load)
#pragma unroll
    for (int k = 0; k < AD; ++k) a_load(ar[k], nrsrc, nbase + k * 1024);
    if (i + 1 < my_n) bv = load_bias(Tn);
    if (SPLIT) {
      // ---- a unit of a split tail item: publish the partial output, the last arrival finishes the item ----
      // (it is the workgroup's last item: the operands just requested are the harmless re-read; let them land
      // first, so that the compiler may do what it likes with the ring registers in the code below)
      ring_landed(ar);
      constexpr int SLOT = BM * 32 * 4;                // floats per unit: 128 channels x 32 tiles x (2 x 2)
      constexpr int kSc1 = 16;                         // cache policy sc1: agent scope, written through / read past L2
      const __amdgpu_buffer_rsrc_t wrs = uniform_rsrc(args.split_ws, (unsigned)((long long)G * SLOT * 4));
      const unsigned my_off = (unsigned)(((wave * 16 + kq * 4) * 32 + jn) * 16);    // + r * 512 + g * 256 bytes
      if (active) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float m0 = acc[j][g][r], m1 = acc[4 + j][g][r], m2 = acc[8 + j][g][r], m3 = acc[12 + j][g][r];
              t[0][j] = m0 + m1 + m2;
              t[1][j] = m1 - m2 - m3;
            }
            f32x4 v;
            v[0] = t[0][0] + t[0][1] + t[0][2]; v[1] = t[0][1] - t[0][2] - t[0][3];
            v[2] = t[1][0] + t[1][1] + t[1][2]; v[3] = t[1][1] - t[1][2] - t[1][3];
            // sc1 = written through at agent scope (cdna_hip_programming.md 5.4: the counter form of the hand-off).
            // The nop keeps the data registers untouched while the store still reads them: without it hipcc put
            // the next tile's v_pk_add right behind the store and the last four lanes of every 16-lane row lost their
            // leading dwords to it -- wrong, irreproducible values in a fixed lane pattern (tools/dbg/r5_split_debug.py);
            // the guide asks the same of hand-written wide stores (5.7 item 1).
            __builtin_amdgcn_raw_buffer_store_b128(
                __builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), wrs,
                my_off + r * 512 + g * 256, (int)blockIdx.x * SLOT * 4, kSc1);
            __builtin_amdgcn_sched_barrier(0);           // (the nop must FOLLOW the store: hipcc floats it otherwise)
            asm volatile("s_nop 7" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
          }
      }
      // Publish: every wave's write-through stores have left (vmcnt counts stores on gfx9), then the barrier, then ONE
      // relaxed agent-scope arrival; the last arrival reads the slots with sc1 loads.  (An agent-scope RELEASE fence
      // here -- buffer_wbl2 -- writes back every dirty line of the XCD's L2, i.e. the megabytes of output the full
      // rounds just wrote: measured, it ate the whole gain of the split.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int& split_last = lv_per[31];                   // the level tables have 24 entries: slot 31 is free
      if (tid == 0) {
        unsigned* tk = args.split_tickets + t_item;
        const unsigned old = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        split_last = old == (unsigned)sp - 1u;
        if (split_last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero for the next launch
      }
      __syncthreads();
      if (split_last && active) {
        const WLevel& L = args.lv[T.l];
        const int H = L.H, W = L.W, HW = H * W;
        const int flags = args.flags;
        const bool relu = flags & SSAD_CONV_RELU, sigm = flags & SSAD_CONV_SIGMOID;
        const bool masked = flags & SSAD_CONV_MASK_AUX;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int sy0 = g ? T.y0[1] : T.y0[0], sx0 = g ? T.x0[1] : T.x0[0], sn = g ? T.n[1] : T.n[0];
          float* yout = L.y + (long long)sn * M * HW;
          const float* aux = masked ? L.aux + (long long)sn * M * HW : nullptr;
          const int tile = g * 16 + jn;
          const int py = PAIRS ? sy0 + 2 * (jn >> 2) : T.y0[0] + 2 * (tile >> 3);
          const int px = PAIRS ? sx0 + 2 * (jn & 3) : T.x0[0] + 2 * (tile & 7);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = mt * 16 + kq * 4 + r;
            f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int q = 0; q < sp; ++q)                  // unit order: the same bits whoever arrives last
              sum += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                  wrs, my_off + r * 512 + g * 256, (t_item * sp + q) * SLOT * 4, kSc1));
            const float bias = (L.bias && m < M) ? L.bias[m] : 0.0f;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              const int yy = py + a;
              if (m < M && yy < H) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
                  if (px + b < W) {
                    float o = sum[2 * a + b] + bias;
                    if (relu) o = o > 0.0f ? o : 0.0f;
                    if (sigm) o = 1.0f / (1.0f + expf(-o));
                    const int off = m * HW + yy * W + px + b;
                    if (aux) o = aux[off] > 0.0f ? o : 0.0f;
                    yout[off] = o;
                  }
              }
            }
          }
        }
      }
    } else
    if (active && !(WINO_ABLATE & 16)) {
      const WLevel& L = args.lv[T.l];
      const int H = L.H, W = L.W, HW = H * W;
      const int flags = args.flags;
      const bool relu = flags & SSAD_CONV_RELU, sigm = flags & SSAD_CONV_SIGMOID;
      const bool masked = flags & SSAD_CONV_MASK_AUX;
      // even W: a 2-pixel store is wholly inside or wholly outside the image, so
      // edge tiles take the same path with out-of-range lanes sent to an
      // out-of-bounds buffer offset (dropped stores, zero loads)
      const bool fast = !(W & 1) && mt * 16 + 16 <= M;
      // A^T M A for output channel row r of tile group g: v[a][b]
      auto out_tile = [&](int g, int r, float (&v)[2][2]) {
        float t[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float m0 = acc[j][g][r], m1 = acc[4 + j][g][r], m2 = acc[8 + j][g][r],
                      m3 = acc[12 + j][g][r];
          t[0][j] = m0 + m1 + m2;
          t[1][j] = m1 - m2 - m3;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          v[a][0] = t[a][0] + t[a][1] + t[a][2];
          v[a][1] = t[a][1] - t[a][2] - t[a][3];
        }
      };
      if (fast && !sigm) {
        // straight-line path: buffer stores, per-lane offset per tile group,
        // (row, channel) displacement in the scalar offset
        // (descriptors over the level's whole batch: tile group g = sub-patch g has its own image)
        const int lvl_bytes = L.N * M * HW * 4;
        const __amdgpu_buffer_rsrc_t yrsrc = uniform_rsrc(L.y, lvl_bytes);
        const __amdgpu_buffer_rsrc_t krsrc = uniform_rsrc(masked ? L.aux : L.y, lvl_bytes);
        const float lo = relu ? 0.0f : -__builtin_inff();
        unsigned vo[NG][2];            // [local tile group][row a]
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
          const int g = g0 + gi;
          const int tile = g * 16 + jn;
          const int sy0 = g ? T.y0[1] : T.y0[0], sx0 = g ? T.x0[1] : T.x0[0], sn = g ? T.n[1] : T.n[0];
          const int py = PAIRS ? sy0 + 2 * (jn >> 2) : T.y0[0] + 2 * (tile >> 3);
          const int px = PAIRS ? sx0 + 2 * (jn & 3) : T.x0[0] + 2 * (tile & 7);
#pragma unroll
          for (int a = 0; a < 2; ++a)
            vo[gi][a] = ((py + a < H) & (px < W))
                ? (unsigned)(((sn * M + mt * 16 + kq * 4) * HW + (py + a) * W + px) * 4) : kOOBOff;
        }
        // channels r, r+1 of an accumulator quad are adjacent registers: the
        // whole A^T M A runs as v_pk_add_f32 on (r, r+1) pairs; the final max
        // (ReLU clamp or -inf) doubles as the repack into (x, x+1) store pairs.
        auto emit = [&](auto has_mask) {
#pragma unroll
          for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              f32x2 kk[2][2];
              if (decltype(has_mask)::value) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                  for (int a = 0; a < 2; ++a)
                    kk[rr][a] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(
                        krsrc, vo[g][a], (2 * h + rr) * HW * 4, 0));
              }
              f32x2 t[2][4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const f32x2 m0 = h ? acc[j][g].zw : acc[j][g].xy;
                const f32x2 m1 = h ? acc[4 + j][g].zw : acc[4 + j][g].xy;
                const f32x2 m2 = h ? acc[8 + j][g].zw : acc[8 + j][g].xy;
                const f32x2 m3 = h ? acc[12 + j][g].zw : acc[12 + j][g].xy;
                t[0][j] = m0 + m1 + m2;
                t[1][j] = m1 - m2 - m3;
              }
#pragma unroll
              for (int a = 0; a < 2; ++a) {
                const f32x2 v0 = t[a][0] + t[a][1] + t[a][2];
                const f32x2 v1 = t[a][1] - t[a][2] - t[a][3];
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                  float o0 = fmaxf(rr ? v0.y : v0.x, lo), o1 = fmaxf(rr ? v1.y : v1.x, lo);
                  if (decltype(has_mask)::value) {
                    o0 = kk[rr][a].x > 0.0f ? o0 : 0.0f;
                    o1 = kk[rr][a].y > 0.0f ? o1 : 0.0f;
                  }
                  __builtin_amdgcn_raw_buffer_store_b64(
                      __builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, make_float2(o0, o1)),
                      yrsrc, vo[g][a], (2 * h + rr) * HW * 4, 0);

                }
              }
            }
        };
        if (masked) emit(std::true_type{}); else emit(std::false_type{});
      } else {
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
          const int g = g0 + gi;
          const int sy0 = g ? T.y0[1] : T.y0[0], sx0 = g ? T.x0[1] : T.x0[0], sn = g ? T.n[1] : T.n[0];
          float* yout = L.y + (long long)sn * M * HW;
          const float* aux = masked ? L.aux + (long long)sn * M * HW : nullptr;
          const int tile = g * 16 + jn;
          const int py = PAIRS ? sy0 + 2 * (jn >> 2) : T.y0[0] + 2 * (tile >> 3);
          const int px = PAIRS ? sx0 + 2 * (jn & 3) : T.x0[0] + 2 * (tile & 7);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = mt * 16 + kq * 4 + r;
            float v[2][2];
            out_tile(gi, r, v);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              const int yy = py + a;
              if (m < M && yy < H) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
                  if (px + b < W) {
                    float o = v[a][b];
                    if (relu) o = o > 0.0f ? o : 0.0f;
                    if (sigm) o = 1.0f / (1.0f + expf(-o));
                    const int off = m * HW + yy * W + px + b;
                    if (aux) o = aux[off] > 0.0f ? o : 0.0f;
                    yout[off] = o;
                  }
              }
            }
          }
        }
      }
    }
    DBG(1, s - 1, 5);
    ring_landed(ar);
    T = Tn;
    arsrc = nrsrc;
    abase = nbase;
  }
}

}  // namespace

// Scratch of the split tail: one 64 KB slot per workgroup of the grid + the arrival counters, per (device, stream)
// -- launches on different streams run concurrently and must not share it; launches on one stream are ordered.
// Allocated on a stream's first split launch (during warm-up), never freed, never grown (the size depends on the
// device's CU count only); the counters are zeroed once and left zero by every launch.
static std::atomic<int>& split_tail_setting() {
  static std::atomic<int> on([] { const char* e = getenv("SSAD_WINO_SPLIT_TAIL"); return (e && *e) ? atoi(e) : 1; }());
  return on;
}

// Units per tail item (1 = no split): the largest power of two p with  p <= 8,  tail * p <= grid,  p | chunks  and
// chunks / p >= 2.  *full = items of the whole rounds, *tail = items of the partial round.
static int split_plan(long long total, int chunks, int cus, long long* full, long long* tail) {
  *full = (total / cus) * cus;
  *tail = total - *full;
  int p = 1;
  while (*tail > 0 && p * 2 <= 8 && *tail * (p * 2) <= cus && chunks % (p * 2) == 0 && chunks / (p * 2) >= 2) p *= 2;
  return p;
}

static int split_scratch(hipStream_t stream, int units, void** ws, unsigned** tickets) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, void*> table;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  const size_t slot_bytes = (size_t)BM * 32 * 4 * sizeof(float);
  const size_t ws_bytes = (size_t)units * slot_bytes;
  std::lock_guard<std::mutex> lock(mu);
  void*& base = table[{dev, stream}];
  if (!base) {
    void* p = nullptr;
    if (hipMalloc(&p, ws_bytes + (size_t)units * sizeof(unsigned) + 256) != hipSuccess) return -1;
    if (hipMemset((char*)p + ws_bytes, 0, (size_t)units * sizeof(unsigned) + 256) != hipSuccess) return -1;
    base = p;
  }
  *ws = base;
  *tickets = (unsigned*)((char*)base + ws_bytes);
  return 0;
}

extern "C" {

size_t ssad_conv_wino_filter_floats(int M, int K) {
  return (size_t)cdiv(M, 16) * cdiv(K, KC) * STEPS * 256 + 1024;
}

int ssad_conv_wino_pack_filter(const float* w, int Cout, int Cin, float* packed_fwd,
                               float* packed_dgrad, ssad_stream_t stream) {
  if (Cout <= 0 || Cin <= 0 || !w) return SSAD_E_BADARG;
  const size_t nf = ssad_conv_wino_filter_floats(Cout, Cin);
  const size_t nd = ssad_conv_wino_filter_floats(Cin, Cout);
  const size_t n = nf > nd ? nf : nd;
  hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, Cout, Cin, packed_fwd, packed_dgrad);
  return (int)hipGetLastError();
}

int ssad_conv_wino_pack_filters(const ssad_pack_entry* entries_host, int n_entries, ssad_stream_t stream) {
  if (n_entries < 0 || (n_entries > 0 && !entries_host)) return SSAD_E_BADARG;
  for (int base = 0; base < n_entries; base += SSAD_MAX_PACK_ENTRIES) {
    const int cnt = n_entries - base < SSAD_MAX_PACK_ENTRIES ? n_entries - base : SSAD_MAX_PACK_ENTRIES;
    PackTable t;
    size_t nmax = 0;
    for (int i = 0; i < cnt; ++i) {
      const ssad_pack_entry& e = entries_host[base + i];
      if (e.Cout <= 0 || e.Cin <= 0 || !e.w) return SSAD_E_BADARG;
      t.e[i] = e;
      const size_t nf = e.packed_fwd ? ssad_conv_wino_filter_floats(e.Cout, e.Cin) : 0;
      const size_t nd = e.packed_dgrad ? ssad_conv_wino_filter_floats(e.Cin, e.Cout) : 0;
      nmax = nf > nmax ? nf : nmax;
      nmax = nd > nmax ? nd : nmax;
    }
    for (int i = cnt; i < SSAD_MAX_PACK_ENTRIES; ++i) t.e[i] = ssad_pack_entry{};
    if (nmax == 0) continue;
    size_t bx = (nmax + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(wino_pack_multi_kernel, dim3((unsigned)bx, (unsigned)cnt), dim3(256), 0,
                       (hipStream_t)stream, t);
  }
  return (int)hipGetLastError();
}

// Geometry of the persistent kernel per level: 8 x 16 patches where they tile the map exactly or the 8 x 8
// sub-patches would not save pixels (a patch stages 18 columns where two sub-patches stage 20: ~4 % faster per
// computed pixel), sub-patch pairs elsewhere.  SSAD_WINO_PAIRS=0 / 1 forces one geometry for every level.
static bool level_wants_pairs(int N, int H, int W) {
  static const int geom = [] { const char* e = getenv("SSAD_WINO_PAIRS"); return (e && *e) ? atoi(e) : -1; }();
  if (geom >= 0) return geom != 0;
  const long long by_patch = (long long)cdiv(W, PC) * cdiv(H, PR) * 128;
  const long long by_sub = (long long)cdiv(W, SP) * cdiv(H, SP) * 64;
  (void)N;
  return by_sub * 100 <= by_patch * 96;
}

// number of kernel launches one call makes (1, or 2 when both geometries occur); tools/pmc_by_class.py
// attributes counters to timing classes by launch order
int ssad_conv3x3_forward_wino_launches(const ssad_conv_level* lv, int n_levels) {
  if (!lv || n_levels < 1) return 0;
  int with_pairs = 0, with_patches = 0;
  for (int l = 0; l < n_levels; ++l) {
    if ((long long)lv[l].N * lv[l].H * lv[l].W == 0) continue;
    if (level_wants_pairs(lv[l].N, lv[l].H, lv[l].W)) ++with_pairs; else ++with_patches;
  }
  return (with_pairs > 0) + (with_patches > 0);
}

// ... and with the split tails of this (Cout, Cin): + 1 per geometry whose partial round is split
int ssad_conv3x3_forward_wino_launches_for(const ssad_conv_level* lv, int n_levels, int Cout, int Cin, int flags) {
  if (!lv || n_levels < 1 || Cout <= 0 || Cin <= 0) return 0;
  constexpr int nhalf = 1;
  const bool half = nhalf && Cout <= 64;
  const int cus = ssad_cu_count();
  int launches = 0;
  for (int pass = 0; pass < 2; ++pass) {
    long long blocks = 0, pairs = 0;
    int nl = 0;
    for (int l = 0; l < n_levels; ++l) {
      if ((long long)lv[l].N * lv[l].H * lv[l].W == 0) continue;
      if (level_wants_pairs(lv[l].N, lv[l].H, lv[l].W) != (pass == 1)) continue;
      ++nl;
      blocks += (long long)lv[l].N * cdiv(lv[l].W, PC) * cdiv(lv[l].H, PR);
      pairs += ((long long)lv[l].N * cdiv(lv[l].W, SP) * cdiv(lv[l].H, SP) + 1) / 2;
    }
    if (!nl) continue;
    const long long total = (pass == 1 ? pairs : blocks) * cdiv(Cout, BM);
    long long full = 0, tail = 0;
    int mode = split_tail_setting().load();
    if (mode && (flags & SSAD_CONV_SPLIT_TAIL)) mode = 2;
    int p = (mode && !half && total <= (long long)cus * ZNT) ? split_plan(total, cdiv(Cin, KC), cus, &full, &tail) : 1;
    if (mode < 2 && full > 0) p = 1;
    launches += p >= 2 ? (full > 0) + 1 : 1;
  }
  return launches;
}

int ssad_conv_wino_split_tail(int on) {
  const int prev = split_tail_setting().load();
  if (on >= 0) split_tail_setting().store(on > 2 ? 2 : on);
  return prev;
}

int ssad_conv3x3_forward_wino(const ssad_conv_level* lv, int n_levels, const float* packed,
                              const float* bias, int Cout, int Cin, int flags,
                              ssad_stream_t stream) {
  if (n_levels < 1 || n_levels > SSAD_MAX_CONV_PROBLEMS || Cout <= 0 || Cin <= 0)
    return SSAD_E_BADARG;
  for (int l = 0; l < n_levels; ++l) {
    if (!(lv[l].packed ? lv[l].packed : packed)) return SSAD_E_BADARG;
    if (lv[l].N < 0 || lv[l].H < 0 || lv[l].W < 0) return SSAD_E_BADARG;
    if ((long long)lv[l].N * lv[l].H * lv[l].W * (Cin > Cout ? Cin : Cout) >= (1LL << 29)) return SSAD_E_BADARG;
    if ((flags & SSAD_CONV_MASK_AUX) && !lv[l].aux) return SSAD_E_BADARG;
  }
  // pass 0: the levels staged as 8 x 16 patches; pass 1: those staged as sub-patch pairs (persistent kernel
  // only)
  for (int pass = 0; pass < 2; ++pass) {
    const bool use_pairs = pass == 1;
    WArgs a;
    a.M = Cout; a.K = Cin; a.chunks = cdiv(Cin, KC); a.flags = flags;
    a.mblocks = cdiv(Cout, BM);
    int nl = 0;
    long long blocks = 0, pairs = 0;
    for (int l = 0; l < n_levels; ++l) {
      if ((long long)lv[l].N * lv[l].H * lv[l].W == 0) continue;
      if (level_wants_pairs(lv[l].N, lv[l].H, lv[l].W) != use_pairs) continue;
      WLevel& L = a.lv[nl++];
      L.x = lv[l].x; L.y = lv[l].y; L.aux = lv[l].aux;
      L.packed = lv[l].packed ? lv[l].packed : packed;
      L.bias = lv[l].packed ? lv[l].bias : bias;
      L.N = lv[l].N; L.H = lv[l].H; L.W = lv[l].W;
      L.tiles_x = cdiv(L.W, PC); L.tiles_y = cdiv(L.H, PR);
      L.block_start = (int)blocks;
      blocks += (long long)L.N * L.tiles_x * L.tiles_y;
      L.sub_x = cdiv(L.W, SP); L.sub_y = cdiv(L.H, SP);
      L.pair_start = (int)pairs;
      pairs += ((long long)L.N * L.sub_x * L.sub_y + 1) / 2;
      if (blocks >= (1LL << 31)) return SSAD_E_BADARG;
    }
    if (nl == 0) continue;
    a.n_levels = nl;
    for (int l = nl; l < SSAD_MAX_CONV_PROBLEMS; ++l) a.lv[l] = WLevel{};
    {   // any Cin: channels past Cin read as zero (buffer range check, zero-padded filter)
      const int cus2 = ssad_cu_count();
      a.patches = (int)(use_pairs ? pairs : blocks);
      const long long total = (long long)a.patches * a.mblocks;
      if (total >= (1LL << 31)) return SSAD_E_BADARG;
      long long grid = total < cus2 ? total : cus2;
      if (grid * ZNT < total) grid = (total + ZNT - 1) / ZNT;
      // tiles per XCD run: 8 (1 = round 2's round-robin order: +4 %, profiles/r03_pmc_classes_roundrobin.md)
      constexpr int xg = 8;
      constexpr int nhalf = 1;
      a.xcd_group = xg;
      const bool half = nhalf && Cout <= 64;
      // Split tail (round 5, the kernel's header): the items of the last, partial round are cut along the
      // reduction into p units each and run as a second launch behind the full rounds.  Taken when the partial
      // round is at most half full (p >= 2); p is a power of two that divides `chunks`, <= 8 and <= chunks / 2,
      // so that a unit is at least two chunks (its fixed costs: prologue, epilogue, the partial round trip).
      // SSAD_WINO_SPLIT_TAIL=0: rounds 1-4's behaviour.
      int split_on = split_tail_setting().load();
      if (split_on && (flags & SSAD_CONV_SPLIT_TAIL)) split_on = 2;
      a.items = (int)total;
      a.full_items = a.split_parts = 0;
      a.split_ws = nullptr;
      a.split_tickets = nullptr;
      long long tail = 0;
      int parts = 0;
      if (split_on && !half && grid == (total < cus2 ? total : cus2)) {
        long long full = 0;
        const int p = split_plan(total, a.chunks, cus2, &full, &tail);
        void* ws = nullptr;
        unsigned* tk = nullptr;
        // setting 1 (default): only launches WITHOUT a full round are split -- there the units replace the launch, no
        // second kernel; setting 2: also the partial round behind full rounds, as a second launch
        if (p >= 2 && (split_on >= 2 || full == 0) &&
            split_scratch((hipStream_t)stream, cus2, &ws, &tk) == 0) {
          parts = p;
          a.items = (int)full;
          a.full_items = (int)full;
          a.split_parts = p;
          a.split_ws = (float*)ws;
          a.split_tickets = tk;
          grid = cus2;
        } else {
          tail = 0;
        }
      }
      const dim3 g3((unsigned)grid), b3(kBlock);
      if (a.items > 0) {
        if (use_pairs && half) hipLaunchKernelGGL((wino_conv_z_kernel<true, true>), g3, b3, 0, (hipStream_t)stream, a);
        else if (use_pairs) hipLaunchKernelGGL((wino_conv_z_kernel<true, false>), g3, b3, 0, (hipStream_t)stream, a);
        else if (half) hipLaunchKernelGGL((wino_conv_z_kernel<false, true>), g3, b3, 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((wino_conv_z_kernel<false, false>), g3, b3, 0, (hipStream_t)stream, a);
      }
      if (parts) {
        const dim3 t3((unsigned)(tail * parts));
        if (use_pairs) hipLaunchKernelGGL((wino_conv_z_kernel<true, false, true>), t3, b3, 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((wino_conv_z_kernel<false, false, true>), t3, b3, 0, (hipStream_t)stream, a);
      }
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

#ifdef WINO_TIMELINE
SSAD_API int ssad_dbg_read(void* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dbg), sizeof(g_dbg));
}
SSAD_API int ssad_dbg_read_waves(void* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wv), sizeof(g_wv));
}
#endif

}  // extern "C"
