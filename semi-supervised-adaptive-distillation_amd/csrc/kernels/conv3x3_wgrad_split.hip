// conv3x3_wgrad_split.hip -- the 3x3 filter gradient on the fp16 matrix pipes by operand splitting.
//
//   dW[m][c][ky][kx] (+)= sum over levels, images, pixels of dY[n][m][y][x] * X[n][c][y + ky - 1][x + kx - 1]
//   (caffe2/operators/conv_op_cudnn.cc:1011-1058, the backward-filter call; conv_op_impl.h:450-575 is the CPU form)
//
// Direct form: a GEMM with the pixels as the reduction index, one 32 x 32 accumulator block per tap.  Both operands are
// scaled by a power of two taken from the tensor's measured |max| and split on the fly into hi + lo fp16 (22 bits);
// hi.hi + lo.hi + hi.lo go through three v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator, the scales are divided out
// in the reduction.  27 fp16 products per output pixel at 16x the fp32 MFMA rate = 1.7 fp32-equivalents, where the
// F(3x3, 2x2) engine of conv3x3_wgrad_winograd.hip spends 4.
//
// Layout of the work.  The reduction index of one MFMA is 16 consecutive pixels of a row (lanes 0-31 carry pixels 0-7,
// lanes 32-63 pixels 8-15), so every image is cut into strips 16 pixels wide and a CHUNK is 4 rows of one strip.  A
// workgroup (4 waves, one per SIMD) owns a 64 x 64 (output x input channel) block of dW for one share of the chunks; a
// wave owns 32 x 32 of it = 9 accumulator blocks = 144 registers (a 64 x 32 wave block needs 288 accumulators against
// 256 AGPRs, and hipcc then spills accumulator blocks inside the loop).  Chunk c + 2's dY (64 channels x 4 x 16) and
// X (64 channels x 6 x 18) are fetched as fp32 into registers while chunk c multiplies, then split and written to the
// other LDS stage while chunk c + 1 multiplies: no packed copy of the tensors exists in memory.  The X
// operand of the taps kx = 0 and kx = 2 is the kx = 1 operand shifted by one fp16: four v_alignbit_b32 on the
// aligned 16-byte LDS read plus one neighbouring word, instead of a misaligned read or shifted copies.
//
// Partial blocks go to a slab per share ([share][tap][Mp][Cp]); wsplit_reduce_kernel adds them in a fixed order.
#include <mutex>
#include <type_traits>

#include "split_common.h"

using namespace ssad_split;

// Debug switches (throw-away builds, tools/dbg/r6_wsplit_ablate.sh): 1 no fetch inside the loop, 2 no split / LDS write
// inside the loop, 4 one product instead of three, 8 no MFMA at all, 16 no barrier.  Results are wrong under any of them.
#ifndef WSPLIT_ABLATE
#define WSPLIT_ABLATE 0
#endif

namespace {

constexpr int CO_T = 64, CI_T = 64;         // workgroup block of dW
constexpr int RW = 4, XR = RW + 2;          // rows of dY / of X per chunk
constexpr int SW = 16;                      // strip width = the MFMA's reduction depth
// LDS records: one per channel, the 16-byte groups (row, half strip) side by side, padded so that 16 consecutive
// channels land on 16 different 4-bank sets.  X's record also holds the halo words (row, side).
constexpr int A_REC = RW * 2 * 16 + 16;             // 144
constexpr int B_HALO = XR * 2 * 16;                 // 192: byte offset of the halo words in a record
constexpr int B_REC = B_HALO + XR * 2 * 4;          // 240
constexpr int A_PLANE = CO_T * A_REC, B_PLANE = CI_T * B_REC;
constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;    // hi + lo of both operands: 67 584 bytes
constexpr int LDS_BYTES = 2 * STAGE;

struct WLevel {
  const float* x;
  const float* dy;
  int N, H, W;
  int strips, yblocks, chunk_start;
  unsigned x_bytes, dy_bytes;
  int vec;                                  // rows start on 16-byte boundaries: 16-byte loads
};
struct WArgs {
  WLevel lv[SSAD_MAX_LEVELS];
  int n_levels, M, C;
  int mtiles, ctiles, shares, per_share, total;
  int xcd_runs;                             // shares per XCD when the grid is laid out share-major per XCD, else 0
  float* slabs;
  const unsigned* amax;                     // [0] = |max| of X over the levels, [1] = of dY
};

// ---- |max| of X (word 0) and dY (word 1) over all levels ---------------------------------------------------------
struct StatTable {
  const float* p[2 * SSAD_MAX_LEVELS];
  long long n[2 * SSAD_MAX_LEVELS];
  int word[2 * SSAD_MAX_LEVELS];
  int block_start[2 * SSAD_MAX_LEVELS + 1];
  int count;
  unsigned* amax;
};
__global__ __launch_bounds__(kThreads) void wsplit_absmax_kernel(const StatTable t) {
  int k = 0;
  for (int j = 1; j < t.count; ++j) k += (int)blockIdx.x >= t.block_start[j];
  const int nb = t.block_start[k + 1] - t.block_start[k], b = (int)blockIdx.x - t.block_start[k];
  const float* x = t.p[k];
  const long long n = t.n[k], n4 = n >> 2;
  unsigned m = 0;
  const uint4* x4 = reinterpret_cast<const uint4*>(x);
#pragma unroll 4
  for (long long i = (long long)b * kThreads + threadIdx.x; i < n4; i += (long long)nb * kThreads) {
    const uint4 v = x4[i];
    const unsigned a0 = v.x & 0x7fffffffu, a1 = v.y & 0x7fffffffu, a2 = v.z & 0x7fffffffu, a3 = v.w & 0x7fffffffu;
    const unsigned p = a0 > a1 ? a0 : a1, q = a2 > a3 ? a2 : a3;
    const unsigned r = p > q ? p : q;
    m = m > r ? m : r;
  }
  if (b == 0)
    for (long long i = n4 * 4 + threadIdx.x; i < n; i += kThreads) {
      const unsigned a = __float_as_uint(x[i]) & 0x7fffffffu;
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
    if (m) atomicMax(t.amax + t.word[k], m);
  }
}

// ---- the main kernel -----------------------------------------------------------------------------------------
struct Chunk {
  int l, n, y0, x0;
};

__device__ __forceinline__ Chunk decode(const WArgs& a, int q) {
  int l = 0;
  for (int j = 1; j < a.n_levels; ++j) l += q >= a.lv[j].chunk_start;
  l = __builtin_amdgcn_readfirstlane(l);
  const WLevel& L = a.lv[l];
  int r = q - L.chunk_start;
  const int strip = r % L.strips;
  r /= L.strips;
  const int yb = r % L.yblocks;
  Chunk c;
  c.l = l;
  c.n = __builtin_amdgcn_readfirstlane(r / L.yblocks);
  c.y0 = __builtin_amdgcn_readfirstlane(yb * RW);
  c.x0 = __builtin_amdgcn_readfirstlane(strip * SW);
  return c;
}

// What one thread fetches for a chunk: 2 groups (8 pixels) of dY, 3 groups of X and 3 halo pixels of X.
//   dY: thread -> (half strip g = t & 1, row r = (t >> 1) & 3, channel (t >> 3) + 32 i), i = 0..1
//   X:  thread -> (half strip g = t & 1, row 2 i + ((t >> 1) & 1), channel t >> 2),       i = 0..2
// Groups outside the tensor are sent to an offset the descriptor rejects (they read 0) and are zeroed again when the
// group is split (a group that straddles the end of a row is cut there).
struct Fetch {
  float dy[2][8];
  float x[3][8];
  float h[3];
};

template <bool VEC>
__device__ __forceinline__ void load8(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, int soff, float (&v)[8]) {
  if (VEC) {
    const f32x4 p = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, soff, 0));
    const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, soff + 16, 0));
    v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
    v[4] = q[0]; v[5] = q[1]; v[6] = q[2]; v[7] = q[3];
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, soff + 4 * e, 0));
  }
}

template <bool VEC>
__device__ __forceinline__ void fetch_chunk(const WArgs& a, const Chunk& c, int m0, int c0, int t, Fetch& f) {
  const WLevel& L = a.lv[c.l];
  const int H = L.H, W = L.W;
  const __amdgpu_buffer_rsrc_t drs = uniform_rsrc(L.dy, L.dy_bytes);
  const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(L.x, L.x_bytes);
  const int g = t & 1;
  {
    const int r = (t >> 1) & 3, m = m0 + (t >> 3), y = c.y0 + r, x = c.x0 + 8 * g;
    const unsigned off = (unsigned)((((c.n * a.M + m) * H + y) * W + x) * 4);
    const unsigned stride = (unsigned)(32 * H * W * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      load8<VEC>(drs, (y < H && x < W && m + 32 * i < a.M) ? off + i * stride : kOob, 0, f.dy[i]);
  }
  {
    // (per-i vector offsets here: row -1 of channel 0 would be a negative vector offset, which the range check rejects
    // whatever the scalar offset adds)
    const int ch = c0 + (t >> 2), x = c.x0 + 8 * g;
    const int hx = g ? c.x0 + SW : c.x0 - 1;           // the halo pixel: left of the strip (g = 0) or right of it
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int y = c.y0 - 1 + 2 * i + ((t >> 1) & 1);
      const bool ok = y >= 0 && y < H;
      const int row = ((c.n * a.C + ch) * H + y) * W;
      load8<VEC>(xrs, ok && x < W ? (unsigned)((row + x) * 4) : kOob, 0, f.x[i]);
      f.h[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
          xrs, ok && hx >= 0 && hx < W ? (unsigned)((row + hx) * 4) : kOob, 0, 0));
    }
  }
}

// Geometry of the chunk whose fetch is in the registers (all wave-uniform).
struct Geo {
  int W, H, y0, x0;
};

// Split and write to LDS the dY groups [d0, d1) and the X groups (+ halo pixels) [x0, x1) of a fetched chunk.
template <int D0, int D1, int X0, int X1>
__device__ __forceinline__ void stage_part(const Geo& G, int mleft, int cleft, int t, const Fetch& f, char* stage,
                                           float sx, float sdy) {
  const int g = t & 1;
  const int nv = G.W - (G.x0 + 8 * g);                // valid pixels of the group (<= 0: none)
  if (D0 < D1) {
    const int r = (t >> 1) & 3, co = t >> 3;
    const int nvr = G.y0 + r < G.H ? nv : 0;
    char* p0 = stage + co * A_REC + (2 * r + g) * 16;
#pragma unroll
    for (int i = D0; i < D1; ++i) {
      const int n = co + 32 * i < mleft ? nvr : 0;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = e < n ? f.dy[i][e] : 0.0f;
      half8 hi, lo;
      split8(v, sdy, hi, lo);
      char* p = p0 + i * 32 * A_REC;
      *reinterpret_cast<half8*>(p) = hi;
      *reinterpret_cast<half8*>(p + A_PLANE) = lo;
    }
  }
  if (X0 < X1) {
    const int rlo = (t >> 1) & 1, ci = t >> 2;
    const int hx = g ? G.x0 + SW : G.x0 - 1;
    const bool chok = ci < cleft;
    char* p0 = stage + 2 * A_PLANE + ci * B_REC + (2 * rlo + g) * 16;
    char* q0 = stage + 2 * A_PLANE + ci * B_REC + B_HALO + (2 * rlo + g) * 4;
#pragma unroll
    for (int i = X0; i < X1; ++i) {
      const int y = G.y0 - 1 + 2 * i + rlo;
      const bool rowok = chok && y >= 0 && y < G.H;
      const int n = rowok ? nv : 0;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = e < n ? f.x[i][e] : 0.0f;
      half8 hi, lo;
      split8(v, sx, hi, lo);
      char* p = p0 + i * 64;
      *reinterpret_cast<half8*>(p) = hi;
      *reinterpret_cast<half8*>(p + B_PLANE) = lo;
      // halo word: the left pixel sits in the HIGH half (it is shifted in from below), the right one in the LOW half
      const float hs = (rowok && hx >= 0 && hx < G.W) ? f.h[i] * sx : 0.0f;
      const _Float16 hh = (_Float16)hs;
      const _Float16 hl = (_Float16)(hs - (float)hh);
      const unsigned bh = (unsigned)__builtin_bit_cast(unsigned short, hh), bl = (unsigned)__builtin_bit_cast(unsigned short, hl);
      char* q = q0 + i * 16;
      *reinterpret_cast<unsigned*>(q) = g ? bh : bh << 16;
      *reinterpret_cast<unsigned*>(q + B_PLANE) = g ? bl : bl << 16;
    }
  }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kThreads, 1) void wsplit_kernel(const WArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wco = wave & 1, wci = wave >> 1;
  const int g = lane >> 5, ln = lane & 31;

  // block -> (share, tile): the tiles of one share read the same pixels, so they sit on one XCD (blocks go to the
  // XCDs round-robin)
  const int tiles = a.mtiles * a.ctiles;
  int share, tile;
  if (a.xcd_runs) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    share = xcd * a.xcd_runs + j / tiles;
    tile = j % tiles;
  } else {
    share = blockIdx.x / tiles;
    tile = blockIdx.x % tiles;
  }
  const int mt = tile / a.ctiles, ct = tile % a.ctiles;
  const int m0 = mt * CO_T, c0 = ct * CI_T;
  const int q_begin = share * a.per_share;
  const int q_end = q_begin + a.per_share < a.total ? q_begin + a.per_share : a.total;

  const int mleft = a.M - m0, cleft = a.C - c0;       // channels of the tile that exist
  const float sx = pow2f(15 - split_exponent(a.amax[0]));
  const float sdy = pow2f(15 - split_exponent(a.amax[1]));

  float16v acc[3][3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ky][kx][e] = 0.0f;

  // per-lane LDS offsets of the operands inside a stage
  const int a_off = (wco * 32 + ln) * A_REC + g * 16;                         // + r * 32 (+ A_PLANE)
  const int b_off = 2 * A_PLANE + (wci * 32 + ln) * B_REC;                    // + rr * 32 + g * 16 (+ B_PLANE)
  const int prev_base = g ? 12 : B_HALO, prev_step = g ? 32 : 8;              // the word left of the group
  const int next_base = g ? B_HALO + 4 : 16, next_step = g ? 8 : 32;          // the word right of it

  // A REGION = one row of the chunk (16 pixels) x one filter row ky: 9 MFMAs.  The three products of an accumulator
  // block are issued three blocks apart, so no MFMA waits for the one before it.  The LDS reads of a region are issued
  // one region ahead (raw words; the shifted operands are made where they are used), and a sixth of the next
  // chunk's split runs in the shadow of each of the first six regions' MFMAs; sched_barriers keep the compiler from
  // merging regions (it then hoists every read of the chunk and spills).
  struct BRaw { u32x4 m[2]; unsigned pv[2], nx[2]; };
  struct ARaw { half8 hi, lo; };
  auto load_b = [&](const char* stage, int rr) {
    BRaw w;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const char* rec = stage + b_off + pl * B_PLANE;
      w.m[pl] = *reinterpret_cast<const u32x4*>(rec + rr * 32 + g * 16);
      w.pv[pl] = *reinterpret_cast<const unsigned*>(rec + prev_base + rr * prev_step);
      w.nx[pl] = *reinterpret_cast<const unsigned*>(rec + next_base + rr * next_step);
    }
    return w;
  };
  auto load_a = [&](const char* stage, int r) {
    ARaw w;
    const char* p = stage + a_off + r * 32;
    w.hi = *reinterpret_cast<const half8*>(p);
    w.lo = *reinterpret_cast<const half8*>(p + A_PLANE);
    return w;
  };
  auto mul = [&](auto KY, const ARaw& A, const BRaw& w) {
    constexpr int ky = decltype(KY)::value;
    half8 b[2][3];                                    // [hi / lo][kx]
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const u32x4 m = w.m[pl];
      u32x4 lft, rgt;
      lft[0] = __builtin_amdgcn_alignbit(m[0], w.pv[pl], 16);
      lft[1] = __builtin_amdgcn_alignbit(m[1], m[0], 16);
      lft[2] = __builtin_amdgcn_alignbit(m[2], m[1], 16);
      lft[3] = __builtin_amdgcn_alignbit(m[3], m[2], 16);
      rgt[0] = lft[1];
      rgt[1] = lft[2];
      rgt[2] = lft[3];
      rgt[3] = __builtin_amdgcn_alignbit(w.nx[pl], m[3], 16);
      b[pl][0] = __builtin_bit_cast(half8, lft);
      b[pl][1] = __builtin_bit_cast(half8, m);
      b[pl][2] = __builtin_bit_cast(half8, rgt);
    }
#pragma unroll
    for (int pr = 0; pr < ((WSPLIT_ABLATE & 8) ? 0 : (WSPLIT_ABLATE & 4) ? 1 : 3); ++pr)   // hi.hi, lo.hi, hi.lo
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        acc[ky][kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 1 ? A.lo : A.hi, b[pr == 2][kx], acc[ky][kx], 0, 0, 0);
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;

  // Schedule of one chunk q (stage s): rows 0 and 1 multiply while chunk q+1 is split into stage s^1; then chunk q+2
  // is fetched into the registers just freed; rows 2 and 3; barrier.  The fetch has two rows and a barrier to land.
  // Past the end the fetch repeats the last chunk and the split writes a stage nobody reads: no branches.
  Fetch f;
  Chunk c1{0, 0, 0, 0};
  if (q_begin < q_end) {
    const Chunk c = decode(a, q_begin);
    if (a.lv[c.l].vec) fetch_chunk<true>(a, c, m0, c0, t, f); else fetch_chunk<false>(a, c, m0, c0, t, f);
    const Geo G0{a.lv[c.l].W, a.lv[c.l].H, c.y0, c.x0};
    stage_part<0, 2, 0, 3>(G0, mleft, cleft, t, f, lds, sx, sdy);
    c1 = decode(a, q_begin + 1 < q_end ? q_begin + 1 : q_end - 1);
    if (a.lv[c1.l].vec) fetch_chunk<true>(a, c1, m0, c0, t, f); else fetch_chunk<false>(a, c1, m0, c0, t, f);
  }
  __syncthreads();

#define SB() __builtin_amdgcn_sched_barrier(0)
  for (int q = q_begin; q < q_end; ++q) {
    const char* stage = lds + ((q - q_begin) & 1) * STAGE;
    char* other = lds + (((q - q_begin) & 1) ^ 1) * STAGE;
    const Geo G{a.lv[c1.l].W, a.lv[c1.l].H, c1.y0, c1.x0};
    ARaw A = load_a(stage, 0);
    BRaw b0 = load_b(stage, 0), b1;
    SB();
    // row 0
    b1 = load_b(stage, 1); mul(K0{}, A, b0); if (!(WSPLIT_ABLATE & 2)) stage_part<0, 1, 0, 0>(G, mleft, cleft, t, f, other, sx, sdy); SB();
    b0 = load_b(stage, 2); mul(K1{}, A, b1); if (!(WSPLIT_ABLATE & 2)) stage_part<1, 2, 0, 0>(G, mleft, cleft, t, f, other, sx, sdy); SB();
    b1 = load_b(stage, 1); ARaw A1 = load_a(stage, 1);
    mul(K2{}, A, b0); if (!(WSPLIT_ABLATE & 2)) stage_part<0, 0, 0, 1>(G, mleft, cleft, t, f, other, sx, sdy); SB();
    // row 1
    b0 = load_b(stage, 2); mul(K0{}, A1, b1); if (!(WSPLIT_ABLATE & 2)) stage_part<0, 0, 1, 2>(G, mleft, cleft, t, f, other, sx, sdy); SB();
    b1 = load_b(stage, 3); mul(K1{}, A1, b0); if (!(WSPLIT_ABLATE & 2)) stage_part<0, 0, 2, 3>(G, mleft, cleft, t, f, other, sx, sdy); SB();
    b0 = load_b(stage, 2); A = load_a(stage, 2);
    mul(K2{}, A1, b1); SB();
    // the registers of the fetch are free: chunk q + 2
    c1 = decode(a, q + 2 < q_end ? q + 2 : q_end - 1);
    if (!(WSPLIT_ABLATE & 1)) {
      if (a.lv[c1.l].vec) fetch_chunk<true>(a, c1, m0, c0, t, f); else fetch_chunk<false>(a, c1, m0, c0, t, f);
    }
    SB();
    // row 2
    b1 = load_b(stage, 3); mul(K0{}, A, b0); SB();
    b0 = load_b(stage, 4); mul(K1{}, A, b1); SB();
    b1 = load_b(stage, 3); A1 = load_a(stage, 3); mul(K2{}, A, b0); SB();
    // row 3
    b0 = load_b(stage, 4); mul(K0{}, A1, b1); SB();
    b1 = load_b(stage, 5); mul(K1{}, A1, b0); SB();
    mul(K2{}, A1, b1);
    if (!(WSPLIT_ABLATE & 16)) __syncthreads();
  }
#undef SB

  // ---- partial block -> slab [share][tap][Mp][Cp] ----
  const int Mp = a.mtiles * CO_T, Cp = a.ctiles * CI_T;
  float* slab = a.slabs + (long long)share * 9 * Mp * Cp;
  const int c = c0 + wci * 32 + ln;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wco * 32 + 8 * (e >> 2) + 4 * g + (e & 3);
        slab[((long long)(ky * 3 + kx) * Mp + m) * Cp + c] = acc[ky][kx][e];
      }
}

// dW[m][c][tap] (+)= 2^(ex - 15) 2^(edy - 15) sum over shares (fixed order) of slab[share][tap][m][c]
__global__ __launch_bounds__(256) void wsplit_reduce_kernel(const float* __restrict__ slabs, int shares, int Mp, int Cp,
                                                            int M, int C, const unsigned* __restrict__ amax,
                                                            float* __restrict__ dW, int accumulate) {
  __shared__ float o[64 * 9];
  const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;           // 4 parts x 64 channels; part p sums taps p, p+4, p+8
  const int c0 = blockIdx.x * 64, m = blockIdx.y;
  const float ux = pow2f(split_exponent(amax[0]) - 15), udy = pow2f(split_exponent(amax[1]) - 15);
  const long long tap_stride = (long long)Mp * Cp, share_stride = 9 * tap_stride;
  for (int tap = part; tap < 9; tap += 4) {
    const float* p = slabs + tap * tap_stride + (long long)m * Cp + c0 + cl;
    float s = 0.0f;
    if (c0 + cl < Cp)
      for (int sh = 0; sh < shares; ++sh) s += p[sh * share_stride];
    o[cl * 9 + tap] = (s * ux) * udy;
  }
  __syncthreads();
  const int nvalid = (C - c0 < 64 ? C - c0 : 64) * 9;
  float* dst = dW + ((long long)m * C + c0) * 9;
  for (int e = threadIdx.x; e < nvalid; e += 256) {
    if (accumulate) dst[e] += o[e]; else dst[e] = o[e];
  }
}

constexpr size_t kHeader = 256;             // the two |max| words, padded

int plan(const ssad_conv_level* lv, int n_levels, int Cout, int Cin, WArgs* a) {
  if (n_levels < 1 || n_levels > SSAD_MAX_LEVELS || Cout <= 0 || Cin <= 0) return SSAD_E_BADARG;
  a->n_levels = n_levels;
  a->M = Cout; a->C = Cin;
  a->mtiles = (Cout + CO_T - 1) / CO_T;
  a->ctiles = (Cin + CI_T - 1) / CI_T;
  long long chunks = 0;
  for (int l = 0; l < SSAD_MAX_LEVELS; ++l) {
    WLevel& L = a->lv[l];
    L = WLevel{};
    if (l >= n_levels) { L.chunk_start = (int)chunks; continue; }
    L.x = lv[l].x; L.dy = lv[l].aux;
    L.N = lv[l].N; L.H = lv[l].H; L.W = lv[l].W;
    if (L.N < 0 || L.H < 0 || L.W < 0) return SSAD_E_BADARG;
    const long long xb = 4LL * L.N * Cin * L.H * L.W, db = 4LL * L.N * Cout * L.H * L.W;
    if (xb >= (1LL << 31) || db >= (1LL << 31)) return SSAD_E_BADARG;      // byte offsets are 32-bit, 2^31 = "outside"
    L.x_bytes = (unsigned)xb; L.dy_bytes = (unsigned)db;
    L.strips = (L.W + SW - 1) / SW; L.yblocks = (L.H + RW - 1) / RW;
    L.chunk_start = (int)chunks;
    L.vec = (L.W % 4 == 0) && (((uintptr_t)L.x | (uintptr_t)L.dy) & 15) == 0;
    if (L.N && L.H && L.W && (!L.x || !L.dy)) return SSAD_E_BADARG;
    chunks += (long long)L.N * L.strips * L.yblocks;
    if (chunks >= (1LL << 30)) return SSAD_E_BADARG;
  }
  a->total = (int)chunks;
  const int cus = ssad_cu_count(), tiles = a->mtiles * a->ctiles;
  // one workgroup per CU: as many shares as fit in one round, at least 4 chunks each; a multiple of 8 when there
  // are that many, so that the tiles of a share can be kept on one XCD
  int s = cus / tiles;
  if (s < 1) s = 1;
  if (s >= 8) s &= ~7;
  while (s > 1 && chunks / s < 4) --s;
  a->per_share = chunks ? (int)((chunks + s - 1) / s) : 0;
  a->shares = chunks ? (int)((chunks + a->per_share - 1) / a->per_share) : 1;
  a->xcd_runs = (a->shares % 8 == 0) ? a->shares / 8 : 0;
  return 0;
}

}  // namespace

size_t ssad_split_wgrad_workspace_bytes(const ssad_conv_level* lv, int n_levels, int Cout, int Cin) {
  WArgs a;
  if (plan(lv, n_levels, Cout, Cin, &a)) return 0;
  return kHeader + sizeof(float) * (size_t)a.shares * 9 * a.mtiles * CO_T * a.ctiles * CI_T;
}

int ssad_split_wgrad_launch(const ssad_conv_level* lv, int n_levels, float* dW, int Cout, int Cin, int accumulate,
                            void* workspace, size_t workspace_bytes, hipStream_t stream) {
  WArgs a;
  const int rc = plan(lv, n_levels, Cout, Cin, &a);
  if (rc) return rc;
  const size_t need = kHeader + sizeof(float) * (size_t)a.shares * 9 * a.mtiles * CO_T * a.ctiles * CI_T;
  if (!workspace || workspace_bytes < need) return SSAD_E_WORKSPACE;
  if (a.total == 0) {
    if (!accumulate) (void)hipMemsetAsync(dW, 0, sizeof(float) * (size_t)Cout * Cin * 9, stream);
    return (int)hipGetLastError();
  }
  unsigned* amax = (unsigned*)workspace;
  a.amax = amax;
  a.slabs = (float*)((char*)workspace + kHeader);
  (void)hipMemsetAsync(amax, 0, 8, stream);
  {
    StatTable st{};
    int blocks = 0, k = 0;
    for (int w = 0; w < 2; ++w)
      for (int l = 0; l < n_levels; ++l) {
        const long long n = (long long)a.lv[l].N * (w ? Cout : Cin) * a.lv[l].H * a.lv[l].W;
        if (n == 0) continue;
        st.p[k] = w ? a.lv[l].dy : a.lv[l].x;
        st.n[k] = n;
        st.word[k] = w;
        st.block_start[k] = blocks;
        long long nb = (n + 8 * 4 * kThreads - 1) / (8 * 4 * kThreads);
        blocks += (int)(nb > 512 ? 512 : nb);
        ++k;
      }
    st.block_start[k] = blocks;
    st.count = k;
    st.amax = amax;
    hipLaunchKernelGGL(wsplit_absmax_kernel, dim3(blocks), dim3(kThreads), 0, stream, st);
  }
  static std::once_flag lds_once;           // > 64 KiB of dynamic LDS needs the opt-in, once per process
  std::call_once(lds_once, [&] {
    (void)hipFuncSetAttribute((const void*)wsplit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  });
  const int tiles = a.mtiles * a.ctiles;
  hipLaunchKernelGGL(wsplit_kernel, dim3(tiles * a.shares), dim3(kThreads), LDS_BYTES, stream, a);
  hipLaunchKernelGGL(wsplit_reduce_kernel, dim3((Cin + 63) / 64, Cout), dim3(256), 0, stream, (const float*)a.slabs,
                     a.shares, a.mtiles * CO_T, a.ctiles * CI_T, Cout, Cin, (const unsigned*)amax, dW, accumulate);
  return (int)hipGetLastError();
}
