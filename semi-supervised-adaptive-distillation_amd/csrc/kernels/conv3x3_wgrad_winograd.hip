// conv3x3_wgrad_winograd.hip -- Winograd F(3x3, 2x2) filter gradient of the 3x3
// convolution, fp32 on the gfx950 matrix cores.
//
// Same contract as conv3x3_wgrad_kernel (caffe2/operators/conv_op_cudnn.cc:
// 1018-1041, conv_op_impl.h:430-520: dW[m][c][ky][kx] = sum over images and
// pixels of dY[m][y][x] * X[c][y+ky-1][x+kx-1]) with 2.25x fewer multiplies:
// per 2x2 output tile
//     dU += (A dY A^T) (.) (B^T X B)          (16 products instead of 36)
//     dW  = G^T dU G                           (once, in the reduce kernel)
// with the F(2x2,3x3) matrices A (4x2), B^T (4x4), G (4x3) transposed into
// their filter-gradient roles.  The 16 products are 16 independent
// [Cout x tiles] x [tiles x Cin] GEMMs with the TILES as reduction dimension,
// on v_mfma_f32_16x16x4_f32 (k = 4 tiles per instruction).
//
// Workgroup = 8 waves = 64 output x 64 input channels x 16 products, and a
// contiguous 1/S share of all (level, image, 4x16-pixel unit) units; wave
// (wm, wc) owns 32 x 16 channels = 2 x 16 accumulator quads = 128 VGPRs.
// Neither operand is ever materialised in transformed form: per k-step a
// lane reads the RAW 2x2 dY patch of its (channel, tile) (2 ds_read_b64) and
// the raw 4x4 X window (8 ds_read_b64) from LDS and applies A . A^T (12 VALU)
// and B^T . B (32 VALU) in registers -- the 32 MFMAs of the step take 1024
// pipe cycles, the VALU work 200, and the second wave of the SIMD covers it.
// (The sign pattern of A's last row is dropped here and re-applied in the
// reduce kernel: row/column 3 of dU are negated there.)
// LDS holds two stages of raw units: dY [64][4x16 (+2)] and X [64][6x20 (+10)]
// floats.  Channel strides 66 / 130 = 2 mod 32: hipcc pairs the operand reads
// into ds_read2_b64, which is served in groups of 16 consecutive lanes on 32
// banks -- the 16 channels of a group then cover all 32 banks once (strides of
// 4 mod 64, right for single ds_read_b64, measured 46 % conflict cycles).
// The next unit is loaded through registers in
// three portions requested at k-steps 0-2 and written after k-step 3 (26 VGPRs).
// Measured (cycle stamps, tower layer): per unit 8.2 k MFMA cycles take 13 k;
// removing the global staging loads alone brings it to 8.5 k (the loads, not the
// LDS stores or the transforms, are what stalls the waves); fewer/wider loads
// with the same number of cache lines touched did not help.
// Partial dU slabs [S][16][Mp][Cp] are combined, transformed by G and scattered
// into dW[m][c][3][3] by wino_wgrad_reduce_kernel in a fixed order.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "conv_internal.h"
#include "ssad_kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int kBlock = 512;
constexpr int GM = 64, GC = 64;          // output / input channels per workgroup
constexpr int UR = 4, UC = 16;           // unit: 4 x 16 output pixels = 2 x 8 tiles
constexpr int SY = 66;                   // dY LDS channel stride (64 + 2)
constexpr int XP = 20;                   // X LDS row pitch (18 used)
constexpr int SX = 130;                  // X LDS channel stride (6 * 20 + 10)
constexpr int STAGE = GM * SY + GC * SX; // floats per stage
constexpr unsigned kOOB = 0x80000000u;

// WGRAD_ABLATE (debug builds only, results are wrong): 1 = no staging DMA inside the loop, 2 = raw values
// instead of B^T d B / A d A^T, 4 = no per-unit barrier, 8 = no operand reads inside the loop,
// 16 = the work of an F(3x3, 2x4) engine per 16 pixels emulated on this skeleton (round 6 prediction): 24 of the 32
// MFMAs of a k-step and 24 more transform operations (F(4,3)'s 6-point transforms: 160 VALU per 48 MFMAs against
// 56 per 32 here), same staging traffic per pixel -- a LOWER bound on what such an engine would take
#ifndef WGRAD_ABLATE
#define WGRAD_ABLATE 0
#endif
#ifdef WGRAD_TIMELINE   // tools/wgrad_timeline.py
__device__ unsigned long long g_wdbg[64][8];
#define WDBG(it, k) if (dbg_on && (it) < 64) g_wdbg[it][k] = __builtin_readcyclecounter()
#else
#define WDBG(it, k)
#endif

__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

using ssad_dev::uniform_rsrc;

struct GLevel {
  const float* x;
  const float* dy;
  int N, H, W;
  int ux, uy;            // units per row / column of one image
  int unit_start;
};

struct GArgs {
  GLevel lv[SSAD_MAX_LEVELS];
  int n_levels;
  int M, C;              // Cout, Cin
  int mblocks, cblocks;
  int total_units, per_split, splits;
  int xcd_group;         // consecutive (split, block pair) work items per XCD run (1 = round robin)
  float* slabs;          // [splits][16][mblocks*GM][cblocks*GC]
};

__global__ __launch_bounds__(kBlock, 1) void wino_wgrad_kernel(const GArgs args) {
  __shared__ float lds[3 * STAGE];       // three stages of raw units: the DMA runs two units ahead

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Work item v = (split, block pair), split-major.  The block pairs of one split read the SAME
  // units (X by every output block, dY by every input block): workgroups b, b + 8, ... share an XCD
  // and its L2, so each XCD takes runs of `xcd_group` consecutive items -- with round-robin ids a
  // split's 16 pairs were spread over all 8 XCDs and every XCD fetched the units for itself (round 3
  // counters: 1.47 GB leave L2 per tower-layer launch for 0.39 GB of operands).
  int v = (int)blockIdx.x;
  {
    const int G = (int)gridDim.x, xg = args.xcd_group;
    if (xg > 1 && (G & 7) == 0 && ((G >> 3) % xg) == 0) {
      const int r = v >> 3, x = v & 7;
      v = (r / xg) * (8 * xg) + x * xg + (r % xg);
    }
  }
  const int pairs = args.mblocks * args.cblocks;
  const int sp = v / pairs, pr = v - sp * pairs;
  const int mb = pr / args.cblocks, cb = pr - mb * args.cblocks;
  const int M = args.M, C = args.C;
  const int u_begin = sp * args.per_split;
  int u_end = u_begin + args.per_split;
  if (u_end > args.total_units) u_end = args.total_units;

  // ---- unit cursor of the LOADER (one unit ahead of the compute) ----
  int l = 0;
#pragma unroll
  for (int i = 1; i < SSAD_MAX_LEVELS; ++i)
    if (i < args.n_levels && u_begin >= args.lv[i].unit_start) l = i;
  int cn, cuy, cux;          // image, unit row, unit column within level l
  {
    const GLevel& L = args.lv[l];
    int r = u_begin - L.unit_start;
    const int per_img = L.ux * L.uy;
    cn = r / per_img;
    r -= cn * per_img;
    cuy = r / L.ux;
    cux = r - cuy * L.ux;
  }
  auto advance = [&]() {
    if (++cux == args.lv[l].ux) {
      cux = 0;
      if (++cuy == args.lv[l].uy) {
        cuy = 0;
        if (++cn == args.lv[l].N) {
          cn = 0;
          ++l;
          // skip empty levels
          while (l < args.n_levels && args.lv[l].N * args.lv[l].ux * args.lv[l].uy == 0) ++l;
        }
      }
    }
  };

  // ---- staging by LDS-DMA (buffer_load_dword ... lds), two units ahead ----
  // One wave-level instruction writes 64 consecutive floats of LDS:
  //   dY: the 4 x 16 patch of one channel (64 floats)       -> 64 instructions per unit,
  //   X : floats [0, 64) and [64, 128) of one channel's 6 x 20 window (pitch 20, 18 columns
  //       used; the tail lands in the channel's padding)     -> 128 instructions per unit,
  // 24 per wave.  The per-lane offset (pixel of the lane's float, out-of-image lanes at an
  // out-of-range offset = zero fill) is the same for every channel; the channel goes into the
  // scalar offset.  No staging registers (the register version held 26 and stalled on its loads:
  // cycle stamps 13.0 k per unit against 8.5 k without the loads -- HBM latency exceeds the 1-3
  // k-steps between request and use; LDS-DMA into a THIRD stage gives two whole units).
  // LDS channel slot of X: bit pairs (0,1) <-> (2,3) of the channel swapped, so that the 16
  // channels of an operand read stay distinct mod 16 (see the header).
  ssad_dev::rsrc_words dyrs, xrs;
  unsigned dy_vo = kOOB, x_vo[2] = {kOOB, kOOB};
  int cHW = 0, m_left = 0, c_left = 0;
  // describe the unit under the cursor (offsets relative to the image's channel block)
  auto setup_unit = [&]() {
    const GLevel& L = args.lv[l];
    const int H = L.H, W = L.W, HW = H * W;
    cHW = HW;
    const int y0 = cuy * UR, x0 = cux * UC;
    m_left = M - mb * GM; c_left = C - cb * GC;
    dyrs = ssad_dev::uniform_rsrc_words(L.dy + ((long long)cn * M + mb * GM) * HW,
                                        (unsigned)((m_left < GM ? m_left : GM) * HW * 4));
    xrs = ssad_dev::uniform_rsrc_words(L.x + ((long long)cn * C + cb * GC) * HW,
                                       (unsigned)((c_left < GC ? c_left : GC) * HW * 4));
    {
      const int gy = y0 + (lane >> 4), gx = x0 + (lane & 15);
      dy_vo = (gy < H && gx < W) ? (unsigned)((gy * W + gx) * 4) : kOOB;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = lane + 64 * k, row = e / XP, col = e % XP;
      const int gy = y0 - 1 + row, gx = x0 - 1 + col;
      const bool ok = row < UR + 2 && col < UC + 2 && gy >= 0 && gy < H && gx >= 0 && gx < W;
      x_vo[k] = ok ? (unsigned)((gy * W + gx) * 4) : kOOB;
    }
  };
  // quarter q of a unit's DMA (6 wave-level instructions): the loop issues one quarter per k-step so
  // that the memory instructions interleave with the MFMAs instead of holding them up in one burst
  auto issue_quarter = [&](float* st, int q) {
    const unsigned base = (unsigned)(uintptr_t)(lds_ptr)st;           // LDS byte address of the stage
    // Channels past the tensor's last (only in its last block: Cout 720 = 11.25 blocks) re-read the last
    // valid channel instead of being masked per lane: the scalar offset is not range-checked, a per-lane select
    // is a VALU instruction (v_cndmask: 8 cycles of the SIMD that the fp32 MFMAs do not get,
    // tools/coissue_probe.hip), and rows / columns of dU beyond (M, C) are never read by the reduce kernel.
#pragma unroll
    for (int i = 2 * q; i < 2 * q + 2; ++i) {
      const int m = wave * (GM / 8) + i;
      const int mm = m < m_left ? m : m_left - 1;
      ssad_dev::lds_dma<4>(dyrs, base + m * SY * 4, dy_vo, mm * cHW * 4);
    }
#pragma unroll
    for (int i = 2 * q; i < 2 * q + 2; ++i) {
      const int c = wave * (GC / 8) + i;
      const int cc = c < c_left ? c : c_left - 1;
      const int slot = (c & 0x30) | ((c & 3) << 2) | ((c >> 2) & 3);
#pragma unroll
      for (int k = 0; k < 2; ++k)
        ssad_dev::lds_dma<4>(xrs, base + (GM * SY + slot * SX + 64 * k) * 4, x_vo[k], cc * cHW * 4);
    }
  };
  auto issue_unit = [&](float* st) {
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_quarter(st, q);
  };
  constexpr int kDmaPerUnit = GM / 8 + 2 * (GC / 8);        // wave-level instructions per wave and unit (24)

  // ---- compute-side constants ----
  const int wm = wave & 1, wc = wave >> 1;
  const int ln = lane & 15, kq = lane >> 4;
  const int a_base = (wm * 32 + ln) * SY;                  // + mg*16*SY + row*UC + col
  const int b_base = GM * SY + (wc * 16 + ((ln & 3) << 2) + (ln >> 2)) * SX;   // slot of channel wc*16+ln
  f32x4 acc[16][2];
#pragma unroll
  for (int x = 0; x < 16; ++x)
#pragma unroll
    for (int g = 0; g < 2; ++g) acc[x][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: units u_begin and u_begin + 1 into stages 0 and 1 ----
  int lu = u_begin;                       // unit under the loader's cursor
  for (int k = 0; k < 2 && lu < u_end; ++k, ++lu) {
    setup_unit();
    issue_unit(lds + k * STAGE);
    advance();
  }

#ifdef WGRAD_TIMELINE
  const bool dbg_on = pr == 3 && sp == 2 && tid == 0;
#endif
  int sidx = 0;                           // stage of unit u
  for (int u = u_begin; u < u_end; ++u) {
    WDBG(u - u_begin, 0);
    // this wave's DMA of unit u has landed (that of unit u + 1 may still fly) ...
    if (u + 1 < u_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDmaPerUnit) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WDBG(u - u_begin, 5);
    // ... and after the barrier everybody's has; every wave is also done with unit u - 1,
    // whose stage the next DMA overwrites
    if (!(WGRAD_ABLATE & 4)) __builtin_amdgcn_s_barrier();
    WDBG(u - u_begin, 6);
    const float* st = lds + sidx * STAGE;
    float* nst = lds + (sidx == 0 ? 2 : sidx - 1) * STAGE;
    const bool fetch = lu < u_end;
    if (fetch) {
      setup_unit();
      advance();
      ++lu;
    }
    // raw operands of one k-step: B window d[4][4] as 8 float2, A patches 2 x 2 float2;
    // those of k-step ks+1 are requested before the MFMAs of k-step ks
    float2 rb[2][8], ra[2][4];
    auto read_raw = [&](int ks, float2 (&b8)[8], float2 (&a4)[4]) {
      const int ty = ks >> 1, tx = (ks & 1) * 4 + kq;
      const float* bp = st + b_base + (2 * ty) * XP + 2 * tx;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        b8[2 * r] = *reinterpret_cast<const float2*>(bp + r * XP);
        b8[2 * r + 1] = *reinterpret_cast<const float2*>(bp + r * XP + 2);
      }
#pragma unroll
      for (int mg = 0; mg < 2; ++mg) {
        const float* ap = st + a_base + mg * 16 * SY + (2 * ty) * UC + 2 * tx;
        a4[2 * mg] = *reinterpret_cast<const float2*>(ap);
        a4[2 * mg + 1] = *reinterpret_cast<const float2*>(ap + UC);
      }
    };
    WDBG(u - u_begin, 1);
    read_raw(0, rb[0], ra[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (fetch && !(WGRAD_ABLATE & 1)) issue_quarter(nst, ks);
      if (ks < 3 && !(WGRAD_ABLATE & 8)) read_raw(ks + 1, rb[(ks + 1) & 1], ra[(ks + 1) & 1]);
      const float2 (&b8)[8] = rb[ks & 1];
      const float2 (&a4)[4] = ra[ks & 1];
      // ---- B operand: raw 4x4 window -> V = B^T d B (16 values) ----
      float v[16];
      {
        float d[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          d[r][0] = b8[2 * r].x; d[r][1] = b8[2 * r].y; d[r][2] = b8[2 * r + 1].x; d[r][3] = b8[2 * r + 1].y;
        }
        float t[4][4];
        if (WGRAD_ABLATE & 2) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = d[i >> 2][i & 3];
        } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          t[0][j] = d[0][j] - d[2][j];
          t[1][j] = d[1][j] + d[2][j];
          t[2][j] = d[2][j] - d[1][j];
          t[3][j] = d[1][j] - d[3][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i * 4 + 0] = t[i][0] - t[i][2];
          v[i * 4 + 1] = t[i][1] + t[i][2];
          v[i * 4 + 2] = t[i][2] - t[i][1];
          v[i * 4 + 3] = t[i][1] - t[i][3];
        }
        }
      }
      if (WGRAD_ABLATE & 16) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
          for (int k = 0; k < 12; ++k) v[k] = fmaf(v[(k + 5) % 12], 0.5f, v[k]);
      }
      // ---- A operand per 16-channel group: raw 2x2 -> A d A^T without the signs of
      //      A's last row (re-applied by the reduce kernel) ----
#pragma unroll
      for (int mg = 0; mg < 2; ++mg) {
        const float2 r0 = a4[2 * mg], r1 = a4[2 * mg + 1];
        const float p[4] = {r0.x, r0.x + r1.x, r0.x - r1.x, r1.x};
        const float q[4] = {r0.y, r0.y + r1.y, r0.y - r1.y, r1.y};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float dm[4] = {p[i], (WGRAD_ABLATE & 2) ? q[i] : p[i] + q[i], (WGRAD_ABLATE & 2) ? p[i] : p[i] - q[i], q[i]};
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (!(WGRAD_ABLATE & 16) || i * 4 + j < 12)
              acc[i * 4 + j][mg] = __builtin_amdgcn_mfma_f32_16x16x4f32(dm[j], v[i * 4 + j],
                                                                        acc[i * 4 + j][mg], 0, 0, 0);
        }
      }
    }
    WDBG(u - u_begin, 2);
    if (++sidx == 3) sidx = 0;
  }

  // ---- partial dU slab: [sp][xi][Mp][Cp] ----
  const int Mp = args.mblocks * GM, Cp = args.cblocks * GC;
  float* slab = args.slabs + (long long)sp * 16 * Mp * Cp;
#pragma unroll
  for (int x = 0; x < 16; ++x)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mb * GM + wm * 32 + g * 16 + kq * 4 + r;
        const int c = cb * GC + wc * 16 + ln;
        slab[((long long)x * Mp + m) * Cp + c] = acc[x][g][r];
      }
}

// dW[m][c][ky][kx] (+)= (G^T (s s^T (.) sum_sp dU) G)[ky][kx],  s = (1,1,1,-1),
// G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].
// Workgroup = one output channel m x 64 input channels; thread (j, c) sums
// column j of dU over the splits (fixed order), applies G^T down the column,
// the row pass and the [c][9] -> linear store go through LDS.
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(
    const float* __restrict__ slabs, int splits, int Mp, int Cp, int M, int C,
    float* __restrict__ dW, int accumulate) {
  __shared__ float t[3][4][64];
  __shared__ float o[64 * 9];
  const int cl = threadIdx.x & 63, j = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 64, m = blockIdx.y;
  const int c = c0 + cl;
  float u[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < Cp) {
    const float* p = slabs + ((long long)j * Mp + m) * Cp + c;
    const long long xs = (long long)4 * Mp * Cp, ss = (long long)16 * Mp * Cp;
#pragma unroll 4
    for (int s = 0; s < splits; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) u[i] += p[s * ss + i * xs];
    }
  }
  // signs dropped by the main kernel: row 3 and column 3 (not both)
  if (j == 3) { u[0] = -u[0]; u[1] = -u[1]; u[2] = -u[2]; }
  else u[3] = -u[3];
  const float h = 0.5f * (u[1] + u[2]);
  t[0][j][cl] = u[0] + h;
  t[1][j][cl] = 0.5f * (u[1] - u[2]);
  t[2][j][cl] = h + u[3];
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 3; e += 256) {
    const int cc = e & 63, k = e >> 6;
    const float g = 0.5f * (t[k][1][cc] + t[k][2][cc]);
    o[cc * 9 + k * 3 + 0] = t[k][0][cc] + g;
    o[cc * 9 + k * 3 + 1] = 0.5f * (t[k][1][cc] - t[k][2][cc]);
    o[cc * 9 + k * 3 + 2] = g + t[k][3][cc];
  }
  __syncthreads();
  const int nvalid = (C - c0 < 64 ? C - c0 : 64) * 9;
  float* dst = dW + ((long long)m * C + c0) * 9;
  for (int e = threadIdx.x; e < nvalid; e += 256) {
    if (accumulate) dst[e] += o[e]; else dst[e] = o[e];
  }
}

int plan(const ssad_conv_level* lv, int n_levels, int Cout, int Cin, GArgs* a) {
  if (n_levels < 1 || n_levels > SSAD_MAX_LEVELS || Cout <= 0 || Cin <= 0) return SSAD_E_BADARG;
  a->n_levels = n_levels;
  a->M = Cout; a->C = Cin;
  a->mblocks = cdiv(Cout, GM); a->cblocks = cdiv(Cin, GC);
  long long units = 0;
  for (int l = 0; l < SSAD_MAX_LEVELS; ++l) {
    GLevel& L = a->lv[l];
    L = GLevel{};
    if (l >= n_levels) continue;
    L.x = lv[l].x; L.dy = lv[l].aux;
    L.N = lv[l].N; L.H = lv[l].H; L.W = lv[l].W;
    if (L.N < 0 || L.H < 0 || L.W < 0) return SSAD_E_BADARG;
    if ((long long)L.H * L.W * 64 >= (1LL << 29)) return SSAD_E_BADARG;
    L.ux = cdiv(L.W, UC); L.uy = cdiv(L.H, UR);
    L.unit_start = (int)units;
    units += (long long)L.N * L.ux * L.uy;
    if (units >= (1LL << 30)) return SSAD_E_BADARG;
  }
  a->total_units = (int)units;
  // splits: minimise rounds(S) x units-per-split(S) on the chip's CUs (all
  // workgroups cost the same), at least 8 units per split, smallest S on ties.  Up to one split per CU:
  // a 64 x 64-channel layer (res2 of the backbones) is ONE block pair -- with the round-2 cap of 32 splits its
  // filter gradient ran on 32 of the 256 CUs.  (Workspace: splits x 16 x Mp x Cp floats, 64 MB at most.)
  const int cus = ssad_cu_count();
  const int oblocks = a->mblocks * a->cblocks;
  int s = 1;
  long long best = -1;
  for (int t = 1; t <= cus; ++t) {
    if (t > 1 && units / t < 8) break;
    if (t > 32 && (long long)t * 16 * a->mblocks * GM * a->cblocks * GC * 4 > (64LL << 20)) break;
    const long long cost = (long long)cdiv(oblocks * t, cus) * cdiv((int)(units ? units : 1), t);
    if (best < 0 || cost < best) { best = cost; s = t; }
  }
  a->per_split = units ? cdiv((int)units, s) : 0;
  a->splits = units ? cdiv((int)units, a->per_split) : 1;
  return 0;
}

}  // namespace

#ifdef WGRAD_TIMELINE
extern "C" __attribute__((visibility("default"))) int ssad_wdbg_read(void* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wdbg), sizeof(g_wdbg));
}
#endif

bool ssad_wino_wgrad_eligible(int Cout, int Cin) {
  const char* e = getenv("SSAD_WGRAD_ENGINE");
  if (e && e[0] == 'd') return false;            // "direct"
  if (e && e[0] == 'w') return true;             // "winograd"
  return Cout >= 32 && Cin >= 64;
}

size_t ssad_wino_wgrad_workspace_bytes(const ssad_conv_level* lv, int n_levels, int Cout, int Cin) {
  GArgs a;
  if (plan(lv, n_levels, Cout, Cin, &a)) return 0;
  return sizeof(float) * (size_t)a.splits * 16 * a.mblocks * GM * a.cblocks * GC;
}

int ssad_wino_wgrad_launch(const ssad_conv_level* lv, int n_levels, float* dW, int Cout, int Cin,
                           int accumulate, void* workspace, size_t workspace_bytes,
                           hipStream_t stream) {
  GArgs a;
  const int rc = plan(lv, n_levels, Cout, Cin, &a);
  if (rc) return rc;
  const size_t need = sizeof(float) * (size_t)a.splits * 16 * a.mblocks * GM * a.cblocks * GC;
  if (!workspace || workspace_bytes < need) return SSAD_E_WORKSPACE;
  a.slabs = (float*)workspace;
  if (a.total_units == 0) {
    if (!accumulate) (void)hipMemsetAsync(dW, 0, sizeof(float) * (size_t)Cout * Cin * 9, stream);
    return (int)hipGetLastError();
  }
  {
    // runs of one whole split per XCD when that divides evenly, else the largest common run length
    const int pairs = a.mblocks * a.cblocks, total = pairs * a.splits;
    int g = 1;
    if ((total & 7) == 0) {
      int x = total >> 3, y = pairs;
      while (y) { const int t = x % y; x = y; y = t; }       // gcd(total / 8, pairs)
      g = x;
    }
    a.xcd_group = g;
    hipLaunchKernelGGL(wino_wgrad_kernel, dim3(total), dim3(kBlock), 0, stream, a);
  }
  hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(cdiv(Cin, 64), Cout), dim3(256), 0,
                     stream, (const float*)a.slabs, a.splits, a.mblocks * GM, a.cblocks * GC, Cout,
                     Cin, dW, accumulate);
  return (int)hipGetLastError();
}
