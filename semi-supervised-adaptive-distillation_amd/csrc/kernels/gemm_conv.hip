// gemm_conv.hip -- pointwise (1x1) convolution of an NCHW fp32 tensor as an exact-fp32 MFMA
// GEMM with the ResNet bottleneck's tail in its epilogue, its data gradient and its filter
// gradient.  This is the backbone's dominant non-3x3 work (row f1): the bottleneck's 1x1
// layers, the projection shortcuts and FPN's lateral convolutions
// (detectron/lib/modeling/ResNet.py:221-283, FPN.py:116-250; reference algorithm
// caffe2/operators/conv_op_impl.h:126-173 -- for a 1x1 kernel im2col is the identity and the
// layer is Y[n] = W . X[n]).
//
//   forward        Y[n][m][p]  = act( sum_k Wt[k][m] X[n][k][p] + bias[m] + R[n][m][p] )
//   data gradient  dX[n][c][p] (+)= mask( sum_m W[m][c] dY[n][m][p] )      (same kernel: the
//                  filter in its natural [M][C] layout IS the [K][M'] operand of this product)
//   filter grad.   dW[m][c]    = sum_{n,p} dY[n][m][p] X[n][c][p]          (gemm_nt below)
//
// Design for MI355X (gfx950):
//  * v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles per instruction per SIMD); a wave owns a
//    64 x 64 output block = 2 x 2 MFMA tiles (64 accumulator registers), a workgroup of four
//    waves a 128 x 128 tile (64 x 128 for 64-wide outputs), three workgroups per CU.
//  * The batch is flattened into the GEMM's column dimension (q = n * P + p): a 20 x 28 map
//    (P = 560) does not waste the last 128-column tile of every image and 16 images of it
//    fill exactly 70 tiles.
//  * Operands travel HBM -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`: no staging
//    registers, no ds_write), in K chunks of 16 through a three-deep ring; the loads of two
//    chunks stay in flight across the one barrier per chunk (raw s_barrier + counted vmcnt).
//    Both tiles are [k][128] rows in LDS, so an MFMA operand is one ds_read of 32 consecutive
//    floats per half-wave (conflict free) and one ds_read2_b32 serves both row tiles.
//  * Tile order is XCD-aware: consecutive tiles (same columns, next 128 output channels) run
//    on the same XCD, so the activation tile is fetched from HBM once and re-read from L2.
//  * Bias, residual, ReLU, the ReluGradient mask and accumulation happen in registers: the
//    pre-activation tensor is never written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "conv_internal.h"
#include "ssad_kernels.h"

namespace {

using ssad_dev::uniform_rsrc;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int kThreads = 256;
constexpr unsigned kOob = 0x80000000u;     // buffer offset past any descriptor: loads 0, stores dropped
constexpr int BN = 128;                    // columns (flattened pixels) per tile
constexpr int BK = 16;                     // reduction chunk
constexpr int NSTAGE = 3;

struct GemmArgs {
  const float* a;        // [K][lda]
  const float* x;        // [N][K][P]
  float* y;              // [N][M][P]
  const float* bias;     // [M] or null
  const float* res;      // [N][M][P] or null
  const float* mask;     // [N][M][P] or null
  int lda, N, K, P, M, flags;
  int mtiles, ctiles;    // tiles along M and along the flattened columns
  long long Q;           // N * P
  // implicit-GEMM convolution (gemm_conv_nn_kernel<BM, true>): x is the image [N][C][H][W], the
  // columns are the output pixels (P = OH * OW), row k = (c, ky, kx) of the reduction is gathered
  int C, H, W, OW, ksize, stride, pad;
  // split-K (a k x k layer on a small map is a few tiles with a very long reduction: P6 is 36 tiles of K = 18 432):
  // workgroup b owns tile b % tiles and the chunks [split * per_split, ...) with split = b / tiles; with splits > 1
  // it writes its raw partial tile to part[split][N][M][P] and splitk_reduce_kernel applies the epilogue terms
  int splits, per_split;
  float* part;
};

// the b32 buffer builtins move 32-bit INTEGERS: floats go through a bit cast, not a conversion
__device__ __forceinline__ float ldf(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
}
__device__ __forceinline__ void stf(float v, __amdgpu_buffer_rsrc_t rs, unsigned off) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, off, 0, 0);
}

// bijective XCD-aware remap: workgroup b runs on XCD b % 8 (observed dispatch order; speed
// only), so give XCD x the contiguous range of tile indices [start_x, start_x + count_x)
__device__ __forceinline__ int xcd_remap(int b, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = b & 7, k = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// IMPLICIT = true: the B operand is not a stored [K][P] matrix but the im2col view of an image,
// gathered by the DMA itself (4 bytes per lane: a stride-2 window is not contiguous): the 7x7/2 stem
// without its 1.35 GB column buffer (conv_op_impl.h:126-173 builds that buffer; same sums, same order
// within a k-chunk).
template <int BM, bool IMPLICIT = false>
__global__ __launch_bounds__(kThreads, 3) void gemm_conv_nn_kernel(const GemmArgs g) {
  constexpr int TI = BM / 64;                  // 32-row MFMA tiles per wave along M (wave = BM/2 rows)
  constexpr int A_STAGE = BK * BM;             // floats
  constexpr int B_STAGE = BK * BN;
  constexpr int STAGE = A_STAGE + B_STAGE;
  constexpr int A_ROW_LANES = BM / 4;          // lanes that cover one A row with 16 B each
  constexpr int A_ROWS_PER_INSTR = 64 / A_ROW_LANES;          // 2 (BM = 128) or 4 (BM = 64)
  constexpr int A_INSTR = BK / A_ROWS_PER_INSTR;              // wave-instructions per stage: 8 or 4
  constexpr int B_INSTR = BK / 2;                             // 8
  constexpr int A_PER_WAVE = A_INSTR / 4;                     // 2 | 1
  constexpr int B_PER_WAVE = IMPLICIT ? BK * (BN / 64) / 4 : B_INSTR / 4;      // 8 (64 columns each) | 2
  constexpr int LOADS = A_PER_WAVE + B_PER_WAVE;              // per wave per stage
  __shared__ __attribute__((aligned(16))) float lds[NSTAGE * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int j = lane & 31, h = lane >> 5;
  const int ntiles = g.mtiles * g.ctiles;
  const int split = g.splits > 1 ? (int)blockIdx.x / ntiles : 0;
  const int tile = xcd_remap((int)blockIdx.x - split * ntiles, ntiles);
  const int mt = tile % g.mtiles, ct = tile / g.mtiles;
  const int m0 = mt * BM;
  const long long q0 = (long long)ct * BN;
  const int K = g.K, P = g.P;

  const __amdgpu_buffer_rsrc_t ars = uniform_rsrc(g.a, (unsigned)((long long)K * g.lda * 4));
  const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(
      g.x, IMPLICIT ? (unsigned)((long long)g.N * g.C * g.H * g.W * 4) : (unsigned)((long long)g.N * K * P * 4));

  // ---- per-lane DMA sources (fixed for the tile; the chunk moves them by a uniform offset) ----
  // A: wave-instruction `ia` (0..A_INSTR) covers rows ia * A_ROWS_PER_INSTR + lane / A_ROW_LANES
  const int a_row = lane / A_ROW_LANES, a_col = (lane % A_ROW_LANES) * 4;
  unsigned a_voff = (unsigned)((a_row * g.lda + m0 + a_col) * 4);
  if (m0 + a_col >= g.lda) a_voff = kOob;
  // B: wave-instruction `ib` covers rows 2 * ib + (lane >> 5), columns (lane & 31) * 4 .. + 3
  const long long qb = q0 + (lane & 31) * 4;
  unsigned b_voff = kOob;
  if (qb < g.Q) {
    const int n = (int)(qb / P), p = (int)(qb - (long long)n * P);
    b_voff = (unsigned)((((long long)n * K + h) * P + p) * 4);
  }
  // implicit mode: this lane's two columns (lane, lane + 64): offset of the window's top-left input pixel and the
  // taps that stay inside the image as bit masks (bit ky of ymask, bit kx of xmask; both 0 for a column past Q)
  int ioff[2] = {0, 0};
  unsigned ymask[2] = {0, 0}, xmask[2] = {0, 0};
  // ... and this wave's four reduction rows of the NEXT chunk to fetch, as (c, ky, kx): wave-uniform, advanced by
  // BK rows per chunk with carries instead of two divisions per row and chunk (the divisions were ~500 scalar
  // instructions per chunk in front of the MFMAs: P6's forward pass ran at 56 TF/s)
  int s_c[4] = {0, 0, 0, 0}, s_ky[4] = {0, 0, 0, 0}, s_kx[4] = {0, 0, 0, 0};
  int d_c = 0, d_ky = 0, d_kx = 0;
  if (IMPLICIT) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const long long qc = q0 + hf * 64 + lane;
      const bool ok = qc < g.Q;
      const int n = ok ? (int)(qc / P) : 0, pp = ok ? (int)(qc - (long long)n * P) : 0;
      const int iy0 = (pp / g.OW) * g.stride - g.pad, ix0 = (pp % g.OW) * g.stride - g.pad;
      ioff[hf] = ((n * g.C * g.H + iy0) * g.W + ix0) * 4;       // may be negative at the border
      if (ok)
        for (int t = 0; t < g.ksize; ++t) {
          ymask[hf] |= (unsigned)((unsigned)(iy0 + t) < (unsigned)g.H) << t;
          xmask[hf] |= (unsigned)((unsigned)(ix0 + t) < (unsigned)g.W) << t;
        }
    }
    const int kk2 = g.ksize * g.ksize;
    const int row0 = (g.splits > 1 ? split * g.per_split : 0) * BK + wave * 4;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int row = row0 + v, c = row / kk2, r2 = row - c * kk2;
      s_c[v] = c; s_ky[v] = r2 / g.ksize; s_kx[v] = r2 - s_ky[v] * g.ksize;
    }
    d_c = BK / kk2;
    const int d_r = BK - d_c * kk2;
    d_ky = d_r / g.ksize; d_kx = d_r - d_ky * g.ksize;
  }
  auto issue = [&](int chunk, int buf) {
    const int k0 = chunk * BK;
    float* base = lds + buf * STAGE;
    // (rows past K are masked per lane only in the chunk that contains K: a per-lane select is a VALU
    // instruction, and VALU cycles are cycles the fp32 MFMAs of the SIMD do not get -- tools/coissue_probe.hip)
    const bool tail = k0 + BK > K;
#pragma unroll
    for (int u = 0; u < A_PER_WAVE; ++u) {
      const int ia = wave * A_PER_WAVE + u;
      const int row = k0 + ia * A_ROWS_PER_INSTR;                  // first row of this instruction
      unsigned vo = a_voff;
      if (tail) vo = (row + a_row < K) ? a_voff : kOob;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (lds_ptr)(base + ia * 256), 16, vo, row * g.lda * 4, 0, 0);
    }
    if (IMPLICIT) {
      // (issue() is called for consecutive chunks, so the (c, ky, kx) state above is always this chunk's)
#pragma unroll
      for (int u = 0; u < B_PER_WAVE; ++u) {
        const int ib = wave * B_PER_WAVE + u;                        // (row in chunk, 64-column half)
        const int rr = ib >> 1, hf = ib & 1, v = u >> 1;
        const int soff = ((s_c[v] * g.H + s_ky[v]) * g.W + s_kx[v]) * 4;                  // scalar
        const unsigned in = (ymask[hf] >> s_ky[v]) & (xmask[hf] >> s_kx[v]) & 1u;
        unsigned vo = in ? (unsigned)(ioff[hf] + soff) : kOob;
        if (tail) vo = (k0 + rr < K) ? vo : kOob;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr)(base + A_STAGE + rr * BN + hf * 64), 4, vo, 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {                                   // + BK rows, with carries
        s_kx[v] += d_kx;
        if (s_kx[v] >= g.ksize) { s_kx[v] -= g.ksize; ++s_ky[v]; }
        s_ky[v] += d_ky;
        if (s_ky[v] >= g.ksize) { s_ky[v] -= g.ksize; ++s_c[v]; }
        s_c[v] += d_c;
      }
    } else {
#pragma unroll
      for (int u = 0; u < B_PER_WAVE; ++u) {
        const int ib = wave * B_PER_WAVE + u;
        const int row = k0 + ib * 2;
        unsigned vo = b_voff;
        if (tail) vo = (row + h < K) ? b_voff : kOob;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr)(base + A_STAGE + ib * 256), 16, vo, row * P * 4, 0, 0);
      }
    }
  };

  f32x16 acc[TI][2];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.0f;

  int chunk0 = 0, chunks = (K + BK - 1) / BK;
  if (g.splits > 1) {
    chunk0 = split * g.per_split;
    chunks = chunks < chunk0 + g.per_split ? chunks : chunk0 + g.per_split;
  }
  issue(chunk0, 0);
  if (chunk0 + 1 < chunks) issue(chunk0 + 1, 1);
  const int a_rd = wm * (BM / 2) + j;          // + k * BM (+ 32 for the second row tile)
  const int b_rd = A_STAGE + wn * 64 + j;      // + k * BN (+ 32)
  int buf = 0;
  for (int c = chunk0; c < chunks; ++c) {
    // this wave's loads of chunk c have landed (those of chunk c + 1 may still fly) ...
    if (c + 1 < chunks) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // ... and after the barrier everybody's have; every wave is also done with chunk c - 1,
    // whose buffer the next DMA overwrites
    __builtin_amdgcn_s_barrier();
    if (c + 2 < chunks) issue(c + 2, buf == 0 ? 2 : buf - 1);
    // Operand reads: ONE base register per operand (lane part + stage), every (k, tile) displacement an
    // immediate offset of a ds_read_b32.  Read through volatile pointers: hipcc otherwise pairs the two tiles
    // of a step into ds_read2_b32, whose 8-bit offsets do not reach the next k row (512 B), and pays for it
    // with two v_add_u32 per step -- VALU issue that the MFMAs of the SIMD wait for.
    typedef const volatile __attribute__((address_space(3))) float* lds_cvf;
    const lds_cvf sa = (lds_cvf)(lds + buf * STAGE + a_rd + h * BM);
    const lds_cvf sb = (lds_cvf)(lds + buf * STAGE + b_rd + h * BN);
    // operands of step ks + 1 are requested before the MFMAs of step ks are issued
    float av[2][TI], bv[2][2];
#pragma unroll
    for (int i = 0; i < TI; ++i) av[0][i] = sa[i * 32];
#pragma unroll
    for (int t = 0; t < 2; ++t) bv[0][t] = sb[t * 32];
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < BK / 2) {
        const int k = 2 * (ks + 1);
#pragma unroll
        for (int i = 0; i < TI; ++i) av[nxt][i] = sa[k * BM + i * 32];
#pragma unroll
        for (int t = 0; t < 2; ++t) bv[nxt][t] = sb[k * BN + t * 32];
      }
      __builtin_amdgcn_sched_barrier(0);      // keep the reads ahead of this step's MFMAs
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][t], acc[i][t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++buf == NSTAGE) buf = 0;
  }

  // ---- epilogue: bias, residual, ReLU, ReluGradient mask, accumulate -- in registers ----------
  const bool partial = g.splits > 1;          // raw partial sums: the reduce kernel owns every epilogue term
  const bool relu = !partial && (g.flags & SSAD_GEMM_RELU), accum = !partial && (g.flags & SSAD_GEMM_ACCUMULATE);
  const float* const gbias = partial ? nullptr : g.bias;
  const float* const gres = partial ? nullptr : g.res;
  const float* const gmask = partial ? nullptr : g.mask;
  const long long ybytes = (long long)g.N * g.M * P * 4;
  float* const yout = partial ? g.part + (long long)split * ((long long)g.N * g.M * P) : g.y;
  const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(yout, (unsigned)ybytes);
  const __amdgpu_buffer_rsrc_t rrs = uniform_rsrc(gres ? gres : yout, (unsigned)ybytes);
  const __amdgpu_buffer_rsrc_t krs = uniform_rsrc(gmask ? gmask : yout, (unsigned)ybytes);
  // Every optional term is applied to a whole 32 x 32 tile (16 values per lane) under ONE
  // wave-uniform branch, so its 16 loads are issued back to back and waited for once (a
  // per-element "load or not" makes hipcc branch and drain vmcnt per element).
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    float bvv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bvv[r] = 0.0f;
    if (gbias) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        bvv[r] = gbias[m < g.M ? m : g.M - 1];
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const long long q = q0 + wn * 64 + t * 32 + j;
      long long col = -1;                         // element offset of (n, m = 0, p), or -1 outside
      if (q < g.Q) {
        const int n = (int)(q / P), p = (int)(q - (long long)n * P);
        col = (long long)n * g.M * P + p;
      }
      unsigned off[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        off[r] = (col >= 0 && m < g.M) ? (unsigned)((col + (long long)m * P) * 4) : kOob;
      }
      f32x16 v = acc[i][t];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] += bvv[r];
      if (gres) {
        float tmp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) tmp[r] = ldf(rrs, off[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += tmp[r];
      }
      if (relu) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.0f);
      }
      if (gmask) {
        float tmp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) tmp[r] = ldf(krs, off[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = tmp[r] > 0.0f ? v[r] : 0.0f;
      }
      if (accum) {
        float tmp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) tmp[r] = ldf(yrs, off[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += tmp[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) stf(v[r], yrs, off[r]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Filter gradient: dW[m][c] = sum over q = (n, p) of dY[n][m][p] X[n][c][p].  Both operands have
// the reduction index contiguous in memory, so both LDS tiles are [row][16] blocks read with
// ds_read_b128 (four reduction steps per read); the reduction order inside a chunk is permuted
// identically for both operands, which a dot product does not care about.  The DMA writes LDS
// lane-linearly, so the bank swizzle is applied on the SOURCE side (which 16-byte piece of the
// row a lane fetches) and undone by the reader: piece c of row r lives in slot c ^ ((r >> 2) & 3).
// Split over the columns: gridDim.y workgroups each reduce a contiguous range of 16-column
// chunks into their own [M][C] slab; a second kernel adds the slabs in a fixed order.
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* x;     // [N][C][P]
  const float* dy;    // [N][M][P]
  float* part;        // [splits][M][C]
  int N, C, P, M;
  int mtiles, ctiles;
  int chunks;         // total 16-column chunks = N * P / 16
  int per_split;      // chunks per split
  int splits;         // splits that own >= 1 chunk
  int xcd_group;      // consecutive (split, tile) work items per XCD run (1 = round robin)
};

__global__ __launch_bounds__(kThreads, 3) void gemm_conv_nt_kernel(const WgradArgs g) {
  constexpr int TSTAGE = 128 * BK;              // floats per operand per stage
  constexpr int STAGE = 2 * TSTAGE;
  __shared__ __attribute__((aligned(16))) float lds[NSTAGE * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int j = lane & 31, h = lane >> 5;
  // work item v = (split, tile), split-major; an XCD (workgroups b, b + 8, ...) takes runs of xcd_group
  // consecutive items: the tiles of one split read the same column range of X and dY
  int v = (int)blockIdx.x;
  {
    const int G = (int)gridDim.x, xg = g.xcd_group;
    if (xg > 1 && (G & 7) == 0 && ((G >> 3) % xg) == 0) {
      const int r = v >> 3, x = v & 7;
      v = (r / xg) * (8 * xg) + x * xg + (r % xg);
    }
  }
  const int ntiles = g.mtiles * g.ctiles;
  const int split = v / ntiles, tile = v - split * ntiles;
  const int mt = tile % g.mtiles, ct = tile / g.mtiles;
  const int m0 = mt * 128, c0 = ct * 128;
  const int P = g.P;
  const int ch0 = split * g.per_split;
  const int ch1 = min(ch0 + g.per_split, g.chunks);
  const int nch = ch1 - ch0;

  const __amdgpu_buffer_rsrc_t yrs = uniform_rsrc(g.dy, (unsigned)((long long)g.N * g.M * P * 4));
  const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(g.x, (unsigned)((long long)g.N * g.C * P * 4));

  // a wave-instruction covers 16 rows x 64 B: lane -> row (lane >> 2), slot (lane & 3), which
  // holds source piece slot ^ ((row >> 2) & 3)
  const int d_row = lane >> 2, d_slot = lane & 3;
  // The lane's part of a source offset (its row of the tile, its 16-byte piece) is fixed for the whole kernel;
  // what moves with the chunk -- image n, first column p -- is wave-uniform and goes into the instruction's SCALAR
  // offset.  (Round 2 rebuilt the full 64-bit per-lane offset for every DMA: ~40 VALU instructions per chunk
  // beside 32 MFMAs, and VALU issue is time the fp32 MFMAs of the SIMD do not get, tools/coissue_probe.hip.)
  unsigned vy0[2], vx0[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = (wave * 2 + u) * 16 + d_row;
    const int piece = d_slot ^ ((row >> 2) & 3);
    vy0[u] = (m0 + row < g.M) ? (unsigned)((((long long)(m0 + row)) * P + piece * 4) * 4) : kOob;
    vx0[u] = (c0 + row < g.C) ? (unsigned)((((long long)(c0 + row)) * P + piece * 4) * 4) : kOob;
  }
  auto issue = [&](int chunk, int buf) {
    const long long q = (long long)(ch0 + chunk) * BK;            // first column of the chunk
    const int n = (int)(q / P), p = (int)(q - (long long)n * P);  // 16 | P: a chunk stays in one image
    const int sy = __builtin_amdgcn_readfirstlane((int)((((long long)n * g.M) * P + p) * 4));
    const int sx = __builtin_amdgcn_readfirstlane((int)((((long long)n * g.C) * P + p) * 4));
    float* base = lds + buf * STAGE;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int inst = wave * 2 + u;                               // 8 instructions per operand
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yrs, (lds_ptr)(base + inst * 256), 16, vy0[u], sy, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr)(base + TSTAGE + inst * 256), 16, vx0[u], sx, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.0f;

  if (nch > 0) issue(0, 0);
  if (nch > 1) issue(1, 1);
  int buf = 0;
  for (int c = 0; c < nch; ++c) {
    if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (c + 2 < nch) issue(c + 2, buf == 0 ? 2 : buf - 1);
    const float* s = lds + buf * STAGE;
    // lane (j, h) reads the two pieces 2 * u + h of its row: reduction steps (u, e), e = 0..3
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float4 av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + i * 32 + j;
        av[i] = *reinterpret_cast<const float4*>(s + row * BK + (((2 * u + h) ^ ((row >> 2) & 3)) << 2));
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = wn * 64 + t * 32 + j;
        bv[t] = *reinterpret_cast<const float4*>(s + TSTAGE + row * BK + (((2 * u + h) ^ ((row >> 2) & 3)) << 2));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[t][e], acc[i][t], 0, 0, 0);
    }
    if (++buf == NSTAGE) buf = 0;
  }

  float* slab = g.part + (long long)split * g.M * g.C;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int c = c0 + wn * 64 + t * 32 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < g.M && c < g.C) slab[(long long)m * g.C + c] = acc[i][t][r];
      }
    }
}

// dW (+)= sum of the slabs in index order (deterministic); optionally also the transposed copy
// Wt-shaped gradient is not needed: SGD runs on the natural layout.
__global__ __launch_bounds__(kThreads) void gemm_conv_wgrad_reduce_kernel(const float* __restrict__ part,
                                                                          int splits, long long n,
                                                                          int accumulate,
                                                                          float* __restrict__ dw) {
  const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  // eight independent chains (fixed order): with two, a thread's `splits` loads (up to 3 x CUs / tiles) went out
  // two at a time and the launch was a chain of load latencies (59 us average in the step)
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 7 < splits; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += part[(long long)(k + u) * n + i];
  }
  for (; k < splits; ++k) a[k & 7] += part[(long long)k * n + i];
  const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  dw[i] = accumulate ? dw[i] + s : s;
}

// Wt[k][m] = W[m][k] (the forward's A operand), rows padded to `ldm` with zeros
__global__ __launch_bounds__(kThreads) void transpose_filter_kernel(const float* __restrict__ w, int M, int K,
                                                                    int ldm, float* __restrict__ wt) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int m = m0 + r, k = k0 + tx;
    tile[r][tx] = (m < M && k < K) ? w[(long long)m * K + k] : 0.0f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, m = m0 + tx;
    if (k < K && m < ldm) wt[(long long)k * ldm + m] = tile[tx][r];
  }
}

// every entry of a table in one launch: blockIdx.y = entry, blockIdx.x = 32 x 32 tile of that entry (workgroups past
// an entry's last tile leave at once)
struct TransposeTable {
  ssad_transpose_entry e[SSAD_MAX_TRANSPOSE_ENTRIES];
};
__global__ __launch_bounds__(kThreads) void transpose_filter_multi_kernel(const TransposeTable t) {
  const ssad_transpose_entry e = t.e[blockIdx.y];
  const int kt = (e.K + 31) / 32, mt = (e.ldm + 31) / 32;
  if ((int)blockIdx.x >= kt * mt) return;
  __shared__ float tile[32][33];
  const int k0 = ((int)blockIdx.x % kt) * 32, m0 = ((int)blockIdx.x / kt) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int m = m0 + r, k = k0 + tx;
    tile[r][tx] = (m < e.M && k < e.K) ? e.w[(long long)m * e.K + k] : 0.0f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, m = m0 + tx;
    if (k < e.K && m < e.ldm) e.wt[(long long)k * e.ldm + m] = tile[tx][r];
  }
}

// y[n][c][oy][ox] = x[n][c][oy * s][ox * s] (the input of a strided pointwise convolution) and
// its gradient (dx zero except at the sampled positions)
__global__ __launch_bounds__(kThreads) void subsample_kernel(const float* __restrict__ x, long long planes,
                                                             int H, int W, int s, int OH, int OW,
                                                             float* __restrict__ y) {
  const long long total = planes * OH * OW;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
    const int ox = (int)(i % OW);
    const long long t = i / OW;
    const int oy = (int)(t % OH);
    const long long pl = t / OH;
    y[i] = x[(pl * H + (long long)oy * s) * W + (long long)ox * s];
  }
}

__global__ __launch_bounds__(kThreads) void subsample_grad_kernel(const float* __restrict__ dy, long long planes,
                                                                  int H, int W, int s, int OH, int OW,
                                                                  int accumulate, float* __restrict__ dx) {
  const long long total = planes * H * W;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
    const int xw = (int)(i % W);
    const long long t = i / W;
    const int yh = (int)(t % H);
    const long long pl = t / H;
    float v = 0.0f;
    if (yh % s == 0 && xw % s == 0 && yh / s < OH && xw / s < OW)
      v = dy[(pl * OH + yh / s) * OW + xw / s];
    dx[i] = accumulate ? dx[i] + v : v;
  }
}

// Stride 2 on a map whose width is a multiple of 4 (every stride-2 block of the backbones at 600 / 500 px): a thread
// turns ONE 16-byte load of an even input row into one 8-byte store (elements 0 and 2 of the four); a 64-lane group
// walks an output row, four rows per workgroup; the row index is decomposed once per thread with 32-bit arithmetic.
// (The general kernel above pays two 64-bit divisions per element: 0.68 ms for res3.0's input at bs 16 where the
// bytes need 0.1 ms.)
__global__ __launch_bounds__(kThreads) void subsample2_kernel(const float* __restrict__ x, unsigned rows, int H, int W,
                                                              int OH, int OW, float* __restrict__ y) {
  const unsigned r = blockIdx.x * 4 + (threadIdx.x >> 6);          // output row: plane * OH + oy
  if (r >= rows) return;
  const unsigned pl = r / (unsigned)OH, oy = r - pl * (unsigned)OH;
  const float4* src = reinterpret_cast<const float4*>(x + ((size_t)pl * H + 2 * oy) * W);
  float2* dst = reinterpret_cast<float2*>(y + (size_t)r * OW);
  for (int j = threadIdx.x & 63; j < (OW >> 1); j += 64) {
    const float4 v = src[j];
    dst[j] = make_float2(v.x, v.z);
  }
}

// its gradient: a thread writes 16 bytes of dx -- (dy0, 0, dy1, 0) on an even row, zeros on an odd one
__global__ __launch_bounds__(kThreads) void subsample2_grad_kernel(const float* __restrict__ dy, unsigned rows, int H,
                                                                   int W, int OH, int OW, int accumulate,
                                                                   float* __restrict__ dx) {
  const unsigned r = blockIdx.x * 4 + (threadIdx.x >> 6);          // input row: plane * H + yh
  if (r >= rows) return;
  const unsigned pl = r / (unsigned)H, yh = r - pl * (unsigned)H;
  const bool live = !(yh & 1) && (yh >> 1) < (unsigned)OH;
  const float2* src = reinterpret_cast<const float2*>(dy + ((size_t)pl * OH + (yh >> 1)) * OW);
  float4* dst = reinterpret_cast<float4*>(dx + (size_t)r * W);
  for (int j = threadIdx.x & 63; j < (W >> 2); j += 64) {
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (live) { const float2 d = src[j]; v.x = d.x; v.z = d.y; }
    if (accumulate) { const float4 o = dst[j]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    dst[j] = v;
  }
}

// split-K epilogue of the implicit GEMM: y = epilogue(sum_s part[s]) in split order (deterministic); the terms of
// gemm_conv_nn_kernel's epilogue in the same order (bias, residual, ReLU, mask, accumulate)
__global__ __launch_bounds__(kThreads) void splitk_reduce_kernel(const float* __restrict__ part, int splits, long long n,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ res,
                                                                 const float* __restrict__ mask, int flags, int M, int P,
                                                                 float* __restrict__ y) {
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
    float v = 0.0f;
    for (int s = 0; s < splits; ++s) v += part[(long long)s * n + i];
    if (bias) v += bias[(int)((i / P) % M)];
    if (res) v += res[i];
    if (flags & SSAD_GEMM_RELU) v = fmaxf(v, 0.0f);
    if (mask) v = mask[i] > 0.0f ? v : 0.0f;
    if (flags & SSAD_GEMM_ACCUMULATE) v += y[i];
    y[i] = v;
  }
}

int pick_splits(int tiles, int chunks) {
  const int cus = ssad_cu_count();
  // about three workgroups per CU, at least 8 chunks (128 columns) each
  int s = (3 * cus + tiles - 1) / tiles;
  const int cap = chunks / 8 > 0 ? chunks / 8 : 1;
  if (s > cap) s = cap;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  return s;
}

}  // namespace

extern "C" {

int ssad_conv1x1_gemm(const ssad_gemm_conv* d, ssad_stream_t stream) {
  if (!d || !d->a || !d->x || !d->y || d->N < 0 || d->K < 1 || d->P < 1 || d->M < 1) return SSAD_E_BADARG;
  if (d->lda < d->M || (d->lda & 3) || (d->P & 3)) return SSAD_E_BADARG;        // 16-byte DMA pieces
  if (((uintptr_t)d->a | (uintptr_t)d->x) & 15) return SSAD_E_BADARG;
  if ((d->flags & SSAD_GEMM_ACCUMULATE) && (d->bias || d->residual)) return SSAD_E_BADARG;
  const long long Q = (long long)d->N * d->P;
  if (Q == 0) return 0;
  const long long big = (long long)d->N * (d->K > d->M ? d->K : d->M) * d->P * 4;
  if (big >= (1LL << 31) || (long long)d->K * d->lda * 4 >= (1LL << 31)) return SSAD_E_BADARG;
  GemmArgs g;
  g.a = d->a; g.x = d->x; g.y = d->y; g.bias = d->bias; g.res = d->residual; g.mask = d->mask;
  g.lda = d->lda; g.N = d->N; g.K = d->K; g.P = d->P; g.M = d->M; g.flags = d->flags;
  g.Q = Q;
  g.ctiles = (int)((Q + BN - 1) / BN);
  g.C = g.H = g.W = g.OW = g.ksize = g.stride = g.pad = 0;
  g.splits = 1; g.per_split = 0; g.part = nullptr;
  hipStream_t s = (hipStream_t)stream;
  // 64-row tiles for 64-wide outputs, and wherever 128-row tiles would leave the chip with
  // fewer than two workgroups per CU (res5 at bs 16 is 70 column tiles: 280 tiles of 128 rows
  // put two workgroups on 24 CUs and one on the rest, i.e. the launch takes twice its share)
  const int cus = ssad_cu_count();
  // (measured: the 64-row tile runs within a few % of the 128-row one on large problems and
  // quantises better on mid-sized ones, so it is the default below ~6 tiles of 128 rows per CU)
  const bool small = (long long)((d->M + 127) / 128) * g.ctiles < 6LL * cus;
  if (d->M <= 64 || small) {
    g.mtiles = (d->M + 63) / 64;
    hipLaunchKernelGGL(gemm_conv_nn_kernel<64>, dim3(g.mtiles * g.ctiles), dim3(kThreads), 0, s, g);
  } else {
    g.mtiles = (d->M + 127) / 128;
    hipLaunchKernelGGL(gemm_conv_nn_kernel<128>, dim3(g.mtiles * g.ctiles), dim3(kThreads), 0, s, g);
  }
  return (int)hipGetLastError();
}

// split-K plan of the implicit GEMM: (splits, chunks per split).  A launch wants about four workgroups per CU (the
// 4-byte gather is latency-bound: P6 forward 0.333 ms with two per CU, 0.306 with four); a split is worth its slab
// traffic only with >= 16 chunks (256 reduction rows) of its own.
static void implicit_split_plan(int M, long long Q, int K, int* splits, int* per_split) {
  const int mt = M <= 64 ? (M + 63) / 64 : (M + 127) / 128;
  const long long tiles = (long long)mt * ((Q + BN - 1) / BN);
  const int chunks = (K + BK - 1) / BK;
  static const int force = [] { const char* e = getenv("SSAD_IMPLICIT_SPLITS"); return e ? atoi(e) : 0; }();
  long long s = (4LL * ssad_cu_count() + tiles - 1) / tiles;
  if (s > chunks / 16) s = chunks / 16;
  if (force > 0) s = force;
  if (s > chunks) s = chunks;
  if (s < 1) s = 1;
  const int per = (int)((chunks + s - 1) / s);
  *per_split = per;
  *splits = (chunks + per - 1) / per;           // every split owns >= 1 chunk
}

size_t ssad_conv_implicit_gemm_workspace_bytes(int N, int M, int C, int H, int W, int kernel, int stride, int pad) {
  if (N < 1 || M < 1 || C < 1 || kernel < 1 || stride < 1 || pad < 0 || H + 2 * pad < kernel || W + 2 * pad < kernel)
    return 0;
  const int OH = (H + 2 * pad - kernel) / stride + 1, OW = (W + 2 * pad - kernel) / stride + 1;
  int splits, per;
  implicit_split_plan(M, (long long)N * OH * OW, C * kernel * kernel, &splits, &per);
  return splits > 1 ? (size_t)splits * N * M * OH * OW * sizeof(float) : 0;
}

int ssad_conv_implicit_gemm_ws(const ssad_gemm_conv* d, int C, int H, int W, int kernel, int stride, int pad,
                               void* workspace, size_t workspace_bytes, ssad_stream_t stream) {
  if (!d || !d->a || !d->x || !d->y || d->N < 0 || d->M < 1 || C < 1 || H < 1 || W < 1) return SSAD_E_BADARG;
  if (kernel < 1 || stride < 1 || pad < 0 || H + 2 * pad < kernel || W + 2 * pad < kernel) return SSAD_E_BADARG;
  if (kernel > 32) return SSAD_E_BADARG;                  // the taps inside the image are one 32-bit mask per axis
  const int OH = (H + 2 * pad - kernel) / stride + 1, OW = (W + 2 * pad - kernel) / stride + 1;
  const int K = C * kernel * kernel;
  if (d->K != K || d->P != OH * OW) return SSAD_E_BADARG;
  // (no constraint on P: the im2col view is gathered 4 bytes per lane and the epilogue stores
  // element-wise -- odd output maps such as P7's 5 x 7 run here too)
  if (d->lda < d->M || (d->lda & 3)) return SSAD_E_BADARG;                       // 16-byte filter pieces
  if ((uintptr_t)d->a & 15) return SSAD_E_BADARG;
  if ((d->flags & SSAD_GEMM_ACCUMULATE) && (d->bias || d->residual)) return SSAD_E_BADARG;
  const long long Q = (long long)d->N * d->P;
  if (Q == 0) return 0;
  if ((long long)d->N * C * H * W * 4 >= (1LL << 31) || (long long)d->N * d->M * d->P * 4 >= (1LL << 31) ||
      (long long)K * d->lda * 4 >= (1LL << 31))
    return SSAD_E_BADARG;
  {
    // the ResNet stem has its own kernel (stem.hip: raw patch staged once in LDS instead of a 12x im2col gather)
    if (C == 3 && kernel == 7 && stride == 2 && pad == 3 && d->M == 64 && !d->bias && !d->residual &&
        !d->mask && d->flags == 0)
      return ssad_stem7x7s2_launch(d->x, d->a, d->lda, d->N, H, W, d->y, (hipStream_t)stream);
  }
  GemmArgs g;
  g.a = d->a; g.x = d->x; g.y = d->y; g.bias = d->bias; g.res = d->residual; g.mask = d->mask;
  g.lda = d->lda; g.N = d->N; g.K = K; g.P = d->P; g.M = d->M; g.flags = d->flags;
  g.Q = Q;
  g.ctiles = (int)((Q + BN - 1) / BN);
  g.C = C; g.H = H; g.W = W; g.OW = OW; g.ksize = kernel; g.stride = stride; g.pad = pad;
  g.splits = 1; g.per_split = 0; g.part = nullptr;
  if (workspace) {
    // no workspace = no split (the stem and the operator surface's forward pass: ssad_conv_implicit_gemm)
    implicit_split_plan(d->M, Q, K, &g.splits, &g.per_split);
    if (g.splits > 1) {
      if (workspace_bytes < (size_t)g.splits * d->N * d->M * d->P * sizeof(float)) return SSAD_E_WORKSPACE;
      if ((uintptr_t)workspace & 15) return SSAD_E_BADARG;
      g.part = (float*)workspace;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  if (d->M <= 64) {
    g.mtiles = (d->M + 63) / 64;
    hipLaunchKernelGGL((gemm_conv_nn_kernel<64, true>), dim3(g.mtiles * g.ctiles * g.splits), dim3(kThreads), 0, s, g);
  } else {
    g.mtiles = (d->M + 127) / 128;
    hipLaunchKernelGGL((gemm_conv_nn_kernel<128, true>), dim3(g.mtiles * g.ctiles * g.splits), dim3(kThreads), 0, s, g);
  }
  if (g.splits > 1) {
    const long long n = (long long)d->N * d->M * d->P;
    long long b = (n + kThreads - 1) / kThreads;
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)b), dim3(kThreads), 0, s, (const float*)g.part, g.splits,
                       n, d->bias, d->residual, d->mask, d->flags, d->M, d->P, d->y);
  }
  return (int)hipGetLastError();
}

int ssad_conv_implicit_gemm(const ssad_gemm_conv* d, int C, int H, int W, int kernel, int stride, int pad,
                            ssad_stream_t stream) {
  return ssad_conv_implicit_gemm_ws(d, C, H, W, kernel, stride, pad, nullptr, 0, stream);
}

int ssad_transpose_filter(const float* w, int M, int K, int ldm, float* wt, ssad_stream_t stream) {
  if (!w || !wt || M < 1 || K < 1 || ldm < M) return SSAD_E_BADARG;
  hipLaunchKernelGGL(transpose_filter_kernel, dim3((K + 31) / 32, (ldm + 31) / 32), dim3(kThreads), 0,
                     (hipStream_t)stream, w, M, K, ldm, wt);
  return (int)hipGetLastError();
}

int ssad_transpose_filters(const ssad_transpose_entry* entries, int n_entries, ssad_stream_t stream) {
  if (n_entries < 0 || (n_entries > 0 && !entries)) return SSAD_E_BADARG;
  if (n_entries == 0) return 0;
  for (int i = 0; i < n_entries; ++i)
    if (!entries[i].w || !entries[i].wt || entries[i].M < 1 || entries[i].K < 1 || entries[i].ldm < entries[i].M)
      return SSAD_E_BADARG;
  for (int base = 0; base < n_entries; base += SSAD_MAX_TRANSPOSE_ENTRIES) {
    const int cnt = n_entries - base < SSAD_MAX_TRANSPOSE_ENTRIES ? n_entries - base : SSAD_MAX_TRANSPOSE_ENTRIES;
    TransposeTable t;
    long long most = 0;
    for (int i = 0; i < SSAD_MAX_TRANSPOSE_ENTRIES; ++i) {
      t.e[i] = i < cnt ? entries[base + i] : ssad_transpose_entry{nullptr, nullptr, 0, 0, 0, 0};
      if (i < cnt) {
        const long long tiles = (long long)((t.e[i].K + 31) / 32) * ((t.e[i].ldm + 31) / 32);
        if (tiles > most) most = tiles;
      }
    }
    if (most >= (1LL << 31)) return SSAD_E_BADARG;
    hipLaunchKernelGGL(transpose_filter_multi_kernel, dim3((unsigned)most, (unsigned)cnt), dim3(kThreads), 0,
                       (hipStream_t)stream, t);
  }
  return (int)hipGetLastError();
}

size_t ssad_conv1x1_wgrad_workspace_bytes(int N, int C, int P, int M) {
  if (N < 1 || C < 1 || P < 1 || M < 1) return 0;
  const int tiles = ((M + 127) / 128) * ((C + 127) / 128);
  const int chunks = (int)(((long long)N * P) / BK);
  return (size_t)pick_splits(tiles, chunks) * (size_t)M * (size_t)C * sizeof(float);
}

int ssad_conv1x1_wgrad(const float* x, const float* dy, int N, int C, int P, int M, float* dw,
                       int accumulate, void* workspace, size_t workspace_bytes, ssad_stream_t stream) {
  if (!x || !dy || !dw || N < 1 || C < 1 || P < 1 || M < 1) return SSAD_E_BADARG;
  if ((P % BK) || (((uintptr_t)x | (uintptr_t)dy) & 15)) return SSAD_E_BADARG;
  if ((long long)N * (C > M ? C : M) * P * 4 >= (1LL << 31)) return SSAD_E_BADARG;
  if (!workspace || workspace_bytes < ssad_conv1x1_wgrad_workspace_bytes(N, C, P, M)) return SSAD_E_WORKSPACE;
  WgradArgs g;
  g.x = x; g.dy = dy; g.part = (float*)workspace;
  g.N = N; g.C = C; g.P = P; g.M = M;
  g.mtiles = (M + 127) / 128; g.ctiles = (C + 127) / 128;
  g.chunks = (int)(((long long)N * P) / BK);
  const int splits = pick_splits(g.mtiles * g.ctiles, g.chunks);
  g.per_split = (g.chunks + splits - 1) / splits;
  const int used = (g.chunks + g.per_split - 1) / g.per_split;        // splits that own >= 1 chunk
  hipStream_t s = (hipStream_t)stream;
  g.splits = used;
  g.xcd_group = 1;      // round robin: runs of operand-sharing tiles per XCD measured worse
  hipLaunchKernelGGL(gemm_conv_nt_kernel, dim3(g.mtiles * g.ctiles * used), dim3(kThreads), 0, s, g);
  const long long n = (long long)M * C;
  hipLaunchKernelGGL(gemm_conv_wgrad_reduce_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, s, (const float*)g.part, used, n, accumulate, dw);
  return (int)hipGetLastError();
}

int ssad_subsample(const float* x, int N, int C, int H, int W, int stride, float* y, ssad_stream_t stream) {
  if (!x || !y || N < 0 || C < 1 || H < 1 || W < 1 || stride < 1) return SSAD_E_BADARG;
  const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  const long long total = (long long)N * C * OH * OW;
  if (total == 0) return 0;
  const long long orows = (long long)N * C * OH;
  if (stride == 2 && (W & 3) == 0 && orows < (1LL << 31) && !(((uintptr_t)x | (uintptr_t)y) & 15)) {
    hipLaunchKernelGGL(subsample2_kernel, dim3((unsigned)((orows + 3) / 4)), dim3(kThreads), 0, (hipStream_t)stream, x,
                       (unsigned)orows, H, W, OH, OW, y);
    return (int)hipGetLastError();
  }
  long long b = (total + kThreads - 1) / kThreads;
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(subsample_kernel, dim3((unsigned)b), dim3(kThreads), 0, (hipStream_t)stream, x,
                     (long long)N * C, H, W, stride, OH, OW, y);
  return (int)hipGetLastError();
}

int ssad_subsample_grad(const float* dy, int N, int C, int H, int W, int stride, int accumulate, float* dx,
                        ssad_stream_t stream) {
  if (!dy || !dx || N < 0 || C < 1 || H < 1 || W < 1 || stride < 1) return SSAD_E_BADARG;
  const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  const long long total = (long long)N * C * H * W;
  if (total == 0) return 0;
  const long long irows = (long long)N * C * H;
  if (stride == 2 && (W & 3) == 0 && irows < (1LL << 31) && !(((uintptr_t)dy | (uintptr_t)dx) & 15)) {
    hipLaunchKernelGGL(subsample2_grad_kernel, dim3((unsigned)((irows + 3) / 4)), dim3(kThreads), 0,
                       (hipStream_t)stream, dy, (unsigned)irows, H, W, OH, OW, accumulate, dx);
    return (int)hipGetLastError();
  }
  long long b = (total + kThreads - 1) / kThreads;
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(subsample_grad_kernel, dim3((unsigned)b), dim3(kThreads), 0, (hipStream_t)stream, dy,
                     (long long)N * C, H, W, stride, OH, OW, accumulate, dx);
  return (int)hipGetLastError();
}

}  // extern "C"
