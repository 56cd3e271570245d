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
// workgroup (8 waves, two per SIMD: one wave's fetch, split and waits run under the other's MFMAs) owns a 128 x 64
// (output x input channel) block of dW for one share of the chunks; a wave owns 32 x 32 of it = 9 accumulator blocks =
// 144 registers (a 64 x 32 wave block needs 288 accumulators against 256 AGPRs, and hipcc then spills accumulator
// blocks inside the loop).  Chunk c + 2's dY (128 channels x 4 x 16) and X (64 channels x 6 x 18) are fetched as fp32
// into registers while chunk c multiplies, then split and written to the
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

constexpr int CO_T = 128, CI_T = 64;        // workgroup block of dW
constexpr int kWG = 512;                    // 8 waves, two per SIMD
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
  // |max| words: the largest of xa[0 .. na) is X's, of da[0 .. na) dY's (one word each when this call measured them,
  // one per level when the caller handed over words measured elsewhere)
  const unsigned* xa;
  const unsigned* da;
  int na;
};

__device__ __forceinline__ unsigned max_word(const unsigned* w, int n) {
  unsigned m = w[0];
  for (int i = 1; i < n; ++i) m = m > w[i] ? m : w[i];
  return m;
}

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

// Chunk q -> coordinates (divisions: once per workgroup) and chunk -> the next one (scalar compares).  Chunks are
// numbered strip-fastest, then row block, image, level.
struct Cursor {
  int l, n, yb, strip;
};
__device__ __forceinline__ Cursor decode(const WArgs& a, int q) {
  int l = 0;
#pragma unroll
  for (int j = 1; j < SSAD_MAX_LEVELS; ++j) l += (j < a.n_levels && q >= a.lv[j].chunk_start);
  l = __builtin_amdgcn_readfirstlane(l);
  const WLevel& L = a.lv[l];
  int r = q - L.chunk_start;
  Cursor c;
  c.l = l;
  c.strip = __builtin_amdgcn_readfirstlane(r % L.strips);
  r /= L.strips;
  c.yb = __builtin_amdgcn_readfirstlane(r % L.yblocks);
  c.n = __builtin_amdgcn_readfirstlane(r / L.yblocks);
  return c;
}
__device__ __forceinline__ void advance(const WArgs& a, Cursor& c) {
  if (++c.strip < a.lv[c.l].strips) return;
  c.strip = 0;
  if (++c.yb < a.lv[c.l].yblocks) return;
  c.yb = 0;
  if (++c.n < a.lv[c.l].N) return;
  c.n = 0;
  do ++c.l; while (c.l < a.n_levels - 1 && a.lv[c.l + 1].chunk_start == a.lv[c.l].chunk_start);   // skip empty levels
}
__device__ __forceinline__ Chunk chunk_of(const Cursor& c) { return Chunk{c.l, c.n, c.yb * RW, c.strip * SW}; }

// What one thread (of 512) fetches for a chunk: 2 groups (8 pixels) of dY, 3 quads (4 pixels) of X and 2 halo pixels of X.
//   dY:   thread -> (half strip g = t & 1, row r = (t >> 1) & 3, channel (t >> 3) + 64 i),            i = 0..1
//   X:    thread -> (quad t & 3 of the strip, row 2 i + ((t >> 2) & 1), channel t >> 3),              i = 0..2
//   halo: thread -> (side t & 1: left / right of the strip, row 4 i + ((t >> 1) & 3), channel t >> 3), i = 0..1 (rows < 6)
// Anything outside the tensor is sent to an offset the descriptor rejects and reads 0; a group that straddles the end
// of a row is cut when it is split (MASKED: only the last strip of a level whose width is not a multiple of 16).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

struct Fetch {
  float dy[2][8];
  f32x4 x[3];
  float h[2];
};

template <bool VEC>
__device__ __forceinline__ void load8(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float (&v)[8]) {
  if (VEC) {
    const f32x4 p = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
    const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 16, 0));
    v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
    v[4] = q[0]; v[5] = q[1]; v[6] = q[2]; v[7] = q[3];
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 4 * e, 0));
  }
}

template <bool VEC>
__device__ __forceinline__ void fetch_chunk(const WArgs& a, const Chunk& c, int m0, int c0, int t, Fetch& f) {
  const WLevel& L = a.lv[c.l];
  const int H = L.H, W = L.W;
  const __amdgpu_buffer_rsrc_t drs = uniform_rsrc(L.dy, L.dy_bytes);
  const __amdgpu_buffer_rsrc_t xrs = uniform_rsrc(L.x, L.x_bytes);
  {
    const int g = t & 1, r = (t >> 1) & 3, m = m0 + (t >> 3), y = c.y0 + r, x = c.x0 + 8 * g;
    const unsigned off = (unsigned)((((c.n * a.M + m) * H + y) * W + x) * 4);
    const unsigned stride = (unsigned)(64 * H * W * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      load8<VEC>(drs, (y < H && x < W && m + 64 * i < a.M) ? off + i * stride : kOob, f.dy[i]);
  }
  {
    const int ch = c0 + (t >> 3);
    const int chrow = (c.n * a.C + ch) * H;
    const bool chok = ch < a.C;
    const int x = c.x0 + 4 * (t & 3);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int y = c.y0 - 1 + 2 * i + ((t >> 2) & 1);
      const unsigned off = (chok && y >= 0 && y < H && x < W) ? (unsigned)(((chrow + y) * W + x) * 4) : kOob;
      if (VEC) {
        f.x[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          f.x[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, off, 4 * e, 0));
      }
    }
    const int hx = (t & 1) ? c.x0 + SW : c.x0 - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rr = 4 * i + ((t >> 1) & 3), y = c.y0 - 1 + rr;
      const bool ok = chok && rr < XR && y >= 0 && y < H && hx >= 0 && hx < W;
      f.h[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
          xrs, ok ? (unsigned)(((chrow + y) * W + hx) * 4) : kOob, 0, 0));
    }
  }
}

// Split and write to LDS part of a fetched chunk: the dY groups [D0, D1), the X quads [X0, X1), the halo pixels
// [H0, H1).  nv = the strip's valid pixels (MASKED only).
template <bool MASKED, int D0, int D1, int X0, int X1, int H0, int H1>
__device__ __forceinline__ void stage_part(int nv, int t, const Fetch& f, char* stage, float sx, float sdy) {
  if (D0 < D1) {
    const int g = t & 1, r = (t >> 1) & 3, co = t >> 3;
    char* p0 = stage + co * A_REC + (2 * r + g) * 16;
#pragma unroll
    for (int i = D0; i < D1; ++i) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (!MASKED || 8 * g + e < nv) ? f.dy[i][e] : 0.0f;
      half8 hi, lo;
      split8(v, sdy, hi, lo);
      char* p = p0 + i * 64 * A_REC;
      *reinterpret_cast<half8*>(p) = hi;
      *reinterpret_cast<half8*>(p + A_PLANE) = lo;
    }
  }
  if (X0 < X1) {
    const int q4 = t & 3, rlo = (t >> 2) & 1, ci = t >> 3;
    char* p0 = stage + 2 * A_PLANE + ci * B_REC + rlo * 32 + q4 * 8;
#pragma unroll
    for (int i = X0; i < X1; ++i) {
      half4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xs = ((!MASKED || 4 * q4 + e < nv) ? f.x[i][e] : 0.0f) * sx;
        const _Float16 h = (_Float16)xs;
        hi[e] = h;
        lo[e] = (_Float16)(xs - (float)h);
      }
      char* p = p0 + i * 64;
      *reinterpret_cast<half4*>(p) = hi;
      *reinterpret_cast<half4*>(p + B_PLANE) = lo;
    }
  }
  if (H0 < H1) {
    const int side = t & 1, ci = t >> 3;
    // halo word: the left pixel sits in the HIGH half (it is shifted in from below), the right one in the LOW half
    char* q0 = stage + 2 * A_PLANE + ci * B_REC + B_HALO + (2 * ((t >> 1) & 3) + side) * 4;
#pragma unroll
    for (int i = H0; i < H1; ++i) {
      const float hs = f.h[i] * sx;
      const _Float16 hh = (_Float16)hs;
      const _Float16 hl = (_Float16)(hs - (float)hh);
      const unsigned bh = (unsigned)__builtin_bit_cast(unsigned short, hh), bl = (unsigned)__builtin_bit_cast(unsigned short, hl);
      if (i == 0 || ((t >> 1) & 3) < XR - 4) {
        char* q = q0 + i * 32;
        *reinterpret_cast<unsigned*>(q) = side ? bh : bh << 16;
        *reinterpret_cast<unsigned*>(q + B_PLANE) = side ? bl : bl << 16;
      }
    }
  }
}

__global__ __launch_bounds__(kWG, 1) void wsplit_kernel(const WArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wco = wave & 3, wci = wave >> 2;
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

  const float sx = pow2f(15 - split_exponent(max_word(a.xa, a.na)));
  const float sdy = pow2f(15 - split_exponent(max_word(a.da, a.na)));

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

  // One row of X (16 + 2 pixels, 32 channels of the wave) is read and shifted ONCE and multiplies every (dY row r,
  // filter row ky) pair with r + ky = that row: up to 3 x 9 MFMAs.  The three products of an accumulator block are
  // issued three blocks apart, so no MFMA waits for the one before it.
  struct BOps { half8 b[2][3]; };                     // [hi / lo][kx]
  struct ARow { half8 hi, lo; };
  auto load_b = [&](const char* stage, int rr) {
    BOps w;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const char* rec = stage + b_off + pl * B_PLANE;
      const u32x4 m = *reinterpret_cast<const u32x4*>(rec + rr * 32 + g * 16);
      const unsigned pv = *reinterpret_cast<const unsigned*>(rec + prev_base + rr * prev_step);
      const unsigned nx = *reinterpret_cast<const unsigned*>(rec + next_base + rr * next_step);
      u32x4 lft, rgt;
      lft[0] = __builtin_amdgcn_alignbit(m[0], pv, 16);
      lft[1] = __builtin_amdgcn_alignbit(m[1], m[0], 16);
      lft[2] = __builtin_amdgcn_alignbit(m[2], m[1], 16);
      lft[3] = __builtin_amdgcn_alignbit(m[3], m[2], 16);
      rgt[0] = lft[1];
      rgt[1] = lft[2];
      rgt[2] = lft[3];
      rgt[3] = __builtin_amdgcn_alignbit(nx, m[3], 16);
      w.b[pl][0] = __builtin_bit_cast(half8, lft);
      w.b[pl][1] = __builtin_bit_cast(half8, m);
      w.b[pl][2] = __builtin_bit_cast(half8, rgt);
    }
    return w;
  };
  auto load_a = [&](const char* stage, int r) {
    ARow w;
    const char* p = stage + a_off + r * 32;
    w.hi = *reinterpret_cast<const half8*>(p);
    w.lo = *reinterpret_cast<const half8*>(p + A_PLANE);
    return w;
  };
  auto mul = [&](auto KY, const ARow& A, const BOps& w) {
    constexpr int ky = decltype(KY)::value;
#pragma unroll
    for (int pr = 0; pr < ((WSPLIT_ABLATE & 8) ? 0 : (WSPLIT_ABLATE & 4) ? 1 : 3); ++pr)   // hi.hi, lo.hi, hi.lo
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        acc[ky][kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 1 ? A.lo : A.hi, w.b[pr == 2][kx], acc[ky][kx], 0, 0, 0);
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;

  // Schedule of one chunk q (stage s): X rows 0-2 multiply while chunk q+1 is split into stage s^1; then chunk q+2
  // is fetched into the registers just freed; X rows 3-5; barrier.  Two waves share a SIMD, so one wave's split,
  // address arithmetic and waits run under the other's MFMAs.  Past the end the fetch repeats the last chunk and
  // the split writes a stage nobody reads: no branches.
  Fetch f;
  Chunk c1{0, 0, 0, 0};
  Cursor cur{0, 0, 0, 0};
  if (q_begin < q_end) {
    cur = decode(a, q_begin);
    const Chunk c = chunk_of(cur);
    if (a.lv[c.l].vec) fetch_chunk<true>(a, c, m0, c0, t, f); else fetch_chunk<false>(a, c, m0, c0, t, f);
    stage_part<true, 0, 2, 0, 3, 0, 2>(a.lv[c.l].W - c.x0, t, f, lds, sx, sdy);
    if (q_begin + 1 < q_end) advance(a, cur);
    c1 = chunk_of(cur);
    if (a.lv[c1.l].vec) fetch_chunk<true>(a, c1, m0, c0, t, f); else fetch_chunk<false>(a, c1, m0, c0, t, f);
  }
  __syncthreads();

#define SB() __builtin_amdgcn_sched_barrier(0)
#define STAGE_PART(...)                                                              \
  if (!(WSPLIT_ABLATE & 2)) {                                                        \
    if (nv >= SW) stage_part<false, __VA_ARGS__>(nv, t, f, other, sx, sdy);          \
    else stage_part<true, __VA_ARGS__>(nv, t, f, other, sx, sdy);                    \
  }
  for (int q = q_begin; q < q_end; ++q) {
    const char* stage = lds + ((q - q_begin) & 1) * STAGE;
    char* other = lds + (((q - q_begin) & 1) ^ 1) * STAGE;
    const int nv = __builtin_amdgcn_readfirstlane(a.lv[c1.l].W - c1.x0);
    SB();
    ARow A0 = load_a(stage, 0);
    BOps B = load_b(stage, 0);
    mul(K0{}, A0, B);
    STAGE_PART(0, 1, 0, 0, 0, 0)
    SB();
    ARow A1 = load_a(stage, 1);
    B = load_b(stage, 1);
    mul(K0{}, A1, B); mul(K1{}, A0, B);
    STAGE_PART(1, 2, 0, 0, 0, 0)
    SB();
    ARow A2 = load_a(stage, 2);
    B = load_b(stage, 2);
    mul(K0{}, A2, B); mul(K1{}, A1, B); mul(K2{}, A0, B);
    STAGE_PART(0, 0, 0, 3, 0, 2)
    SB();
    // the registers of the fetch are free: chunk q + 2
    if (q + 2 < q_end) advance(a, cur);
    c1 = chunk_of(cur);
    if (!(WSPLIT_ABLATE & 1)) {
      if (a.lv[c1.l].vec) fetch_chunk<true>(a, c1, m0, c0, t, f); else fetch_chunk<false>(a, c1, m0, c0, t, f);
    }
    SB();
    ARow A3 = load_a(stage, 3);
    B = load_b(stage, 3);
    mul(K0{}, A3, B); mul(K1{}, A2, B); mul(K2{}, A1, B);
    SB();
    B = load_b(stage, 4);
    mul(K1{}, A3, B); mul(K2{}, A2, B);
    SB();
    B = load_b(stage, 5);
    mul(K2{}, A3, B);
    if (!(WSPLIT_ABLATE & 16)) __syncthreads();
  }
#undef STAGE_PART
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
                                                            int M, int C, const unsigned* __restrict__ xa,
                                                            const unsigned* __restrict__ da, int na,
                                                            float* __restrict__ dW, int accumulate) {
  __shared__ float o[64 * 9];
  const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;           // 4 parts x 64 channels; part p sums taps p, p+4, p+8
  const int c0 = blockIdx.x * 64, m = blockIdx.y;
  const float ux = pow2f(split_exponent(max_word(xa, na)) - 15), udy = pow2f(split_exponent(max_word(da, na)) - 15);
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
                            void* workspace, size_t workspace_bytes, const unsigned* x_amax, const unsigned* dy_amax,
                            hipStream_t stream) {
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
  a.slabs = (float*)((char*)workspace + kHeader);
  if (x_amax && dy_amax) {                  // measured by the caller: word l = level l (levels without pixels hold 0)
    a.xa = x_amax; a.da = dy_amax; a.na = n_levels;
  } else {
  a.xa = amax; a.da = amax + 1; a.na = 1;
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
  }
  static std::once_flag lds_once;           // > 64 KiB of dynamic LDS needs the opt-in, once per process
  std::call_once(lds_once, [&] {
    (void)hipFuncSetAttribute((const void*)wsplit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  });
  const int tiles = a.mtiles * a.ctiles;
  hipLaunchKernelGGL(wsplit_kernel, dim3(tiles * a.shares), dim3(kWG), LDS_BYTES, stream, a);
  hipLaunchKernelGGL(wsplit_reduce_kernel, dim3((Cin + 63) / 64, Cout), dim3(256), 0, stream, (const float*)a.slabs,
                     a.shares, a.mtiles * CO_T, a.ctiles * CI_T, Cout, Cin, a.xa, a.da, a.na, dW, accumulate);
  return (int)hipGetLastError();
}
