// conv3x3.hip -- RetinaNet subnet convolutions (3x3, stride 1, pad 1, NCHW,
// fp32) on the gfx950 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// Replaces the cuDNN calls of caffe2/operators/conv_op_cudnn.cc:567-617
// (forward + bias) and :1011-1058 (backward bias / filter / data); the
// arithmetic definition is the reference's default engine,
// caffe2/operators/conv_op_impl.h:121-194 and :450-575.
//
// Design (MI355X-first, not an im2col+GEMM translation):
//
//  * Implicit GEMM with NO im2col buffer.  A workgroup owns a 2-D patch of
//    output pixels (rows x 16 columns) of one image of one FPN level and a
//    block of output channels.  The input patch with its 1-pixel halo is
//    staged ONCE per 8-channel chunk into LDS (zero-filled outside the
//    image), and the nine filter taps are nine shifted reads of that patch:
//    every B-operand read is `ds_read_b32 base + immediate`, no per-tap
//    address arithmetic, no boundary masks in the inner loop.
//  * LDS row pitch is 48 floats (= 16 mod 32), so the two 16-pixel rows of a
//    32-pixel MFMA tile fall on disjoint banks; the k and k+1 channel of an
//    MFMA (lanes 0-31 / 32-63) are separate half-wave accesses.
//  * The filter is repacked once per step into the exact lane order of the
//    MFMA A operand (ssad_conv_pack_filter): a wave's filter traffic is a
//    linear stream of 16-byte-per-lane loads (one float4 = the 8 channels of
//    one tap), prefetched two taps ahead straight into VGPRs -- the filter
//    never occupies LDS and each weight is fetched once per workgroup.
//  * fp32 MFMA issues one instruction per 64 cycles per SIMD; with two waves
//    per SIMD the matrix pipe stays fed while the other wave reads LDS.
//  * All FPN levels that share a filter (the five pyramid levels of a
//    RetinaNet tower layer) go into ONE launch through a level table, so the
//    small P5..P7 maps fill the tail of the P3/P4 wave instead of costing
//    separate under-filled launches.
//  * Bias, ReLU (forward) and the ReLU-gradient mask (data gradient) are
//    epilogue options; the data gradient is the same kernel run with the
//    flipped/transposed packed filter.
//  * Weight gradient: the reduction dimension is pixels.  A workgroup owns
//    128 output x 64 input channels x 9 taps (144 accumulator VGPRs per wave)
//    and a contiguous share of all (level, image, patch) pixel patches;
//    partial slabs are combined by a fixed-order reduce kernel (deterministic,
//    also performs the sum over the levels that share the filter,
//    caffe2/python/core.py:706-741).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>

#include "conv_internal.h"
#include "ssad_kernels.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 8;        // input channels per LDS chunk
constexpr int TW = 16;       // patch width in pixels
constexpr int PITCH = 48;    // LDS floats per patch row (18 used)
constexpr int kBlock = 512;  // 8 waves

__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

using ssad_dev::uniform_rsrc;

// ---------------------------------------------------------------------------
// Filter packing
// ---------------------------------------------------------------------------
// packed[mtile][chunk][tap][lane][cp], lane = kk*32 + i:
//   value = Wsrc[out = mtile*32 + i][in = chunk*8 + 2*cp + kk][tap]
// i.e. exactly the A operand (lane l holds A[i = l&31][k = l>>5]) of the MFMA
// for k-pair cp of that tap, so one 16-byte load per lane feeds 4 MFMAs.

__global__ void pack_filter_kernel(
    const float* __restrict__ w, int Cout, int Cin, float* __restrict__ pf,
    float* __restrict__ pd) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // forward: outputs = Cout, inputs = Cin
  {
    const int mtiles = cdiv(Cout, 32), chunks = cdiv(Cin, KC);
    const long long total = (long long)mtiles * chunks * 9 * 256;
    if (pf && tid < total + 512) {
      float v = 0.0f;
      if (tid < total) {
        const int cp = tid & 3, lane = (tid >> 2) & 63;
        long long r = tid >> 8;
        const int tap = r % 9; r /= 9;
        const int chunk = r % chunks; const int mt = r / chunks;
        const int m = mt * 32 + (lane & 31), c = chunk * KC + 2 * cp + (lane >> 5);
        if (m < Cout && c < Cin) v = w[((long long)m * Cin + c) * 9 + tap];
      }
      pf[tid] = v;
    }
  }
  // data gradient: outputs = Cin, inputs = Cout, taps flipped
  {
    const int mtiles = cdiv(Cin, 32), chunks = cdiv(Cout, KC);
    const long long total = (long long)mtiles * chunks * 9 * 256;
    if (pd && tid < total + 512) {
      float v = 0.0f;
      if (tid < total) {
        const int cp = tid & 3, lane = (tid >> 2) & 63;
        long long r = tid >> 8;
        const int tap = r % 9; r /= 9;
        const int chunk = r % chunks; const int mt = r / chunks;
        const int co = mt * 32 + (lane & 31);            // output of dgrad = fwd input ch
        const int ci = chunk * KC + 2 * cp + (lane >> 5); // input of dgrad = fwd output ch
        if (co < Cin && ci < Cout) v = w[((long long)ci * Cin + co) * 9 + (8 - tap)];
      }
      pd[tid] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// Forward / data-gradient kernel
// ---------------------------------------------------------------------------

struct FwdLevel {
  const float* x;
  float* y;
  const float* aux;
  const float* packed;
  const float* bias;
  int N, H, W;
  int tiles_x, tiles_y;   // patches per image
  int block_start;        // first blockIdx.x of this level
};

struct FwdArgs {
  FwdLevel lv[SSAD_MAX_CONV_PROBLEMS];
  int n_levels;
  const float* packed;
  const float* bias;
  int M, K;        // output / input channels
  int chunks;      // ceil(K / 8)
  int flags;
};

// WM x WP waves; each wave: 32 output channels x PT pixel tiles (2 rows x 16).
template <int WM, int WP, int PT, bool TAP_FENCE>
__global__ __launch_bounds__(kBlock, 2) void conv3x3_kernel(const FwdArgs args) {
  static_assert(WM * WP * 64 == kBlock, "8 waves");
  constexpr int PR = 2 * PT * WP;            // patch rows
  constexpr int CS = (PR + 2) * PITCH;       // LDS floats per channel
  constexpr int BUF = KC * CS;               // floats per buffer
  constexpr int STAGE = KC * (PR + 2) * (TW + 2);
  constexpr int SITER = cdiv(STAGE, kBlock);
  __shared__ float lds[2 * BUF];

  // ---- which patch -------------------------------------------------------
  int l = 0;
#pragma unroll
  for (int i = 1; i < SSAD_MAX_CONV_PROBLEMS; ++i)
    if (i < args.n_levels && (int)blockIdx.x >= args.lv[i].block_start) l = i;
  const FwdLevel& L = args.lv[l];
  const int H = L.H, W = L.W;
  int pid = blockIdx.x - L.block_start;
  const int per_img = L.tiles_x * L.tiles_y;
  const int n = pid / per_img;
  pid -= n * per_img;
  const int ty = pid / L.tiles_x;
  const int tx = pid - ty * L.tiles_x;
  const int y0 = ty * PR, x0 = tx * TW;
  const int K = args.K, M = args.M;
  const int HW = H * W;
  const float* xin = L.x + (long long)n * K * HW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wp = wave / WM;
  const int mtile = blockIdx.y * WM + wm;
  const int mtiles = cdiv(M, 32);
  const bool active = mtile < mtiles;         // wave-uniform

  // ---- staging map (fixed per thread) -------------------------------------
  // element e -> (channel c, patch row r, patch col q).  The global side is a
  // raw buffer load: descriptor = this image's K x H x W block (channels past
  // K read as 0), per-lane byte offset pushed out of range outside the image,
  // chunk step through the scalar offset.
  constexpr unsigned kOOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t xrsrc = uniform_rsrc(xin, K * HW * 4);
  int s_lds[SITER];
  unsigned s_voff[SITER];
#pragma unroll
  for (int it = 0; it < SITER; ++it) {
    const int e = tid + it * kBlock;
    const int c = e / ((PR + 2) * (TW + 2));
    const int rem = e - c * ((PR + 2) * (TW + 2));
    const int r = rem / (TW + 2);
    const int q = rem - r * (TW + 2);
    const int gy = y0 - 1 + r, gx = x0 - 1 + q;
    const bool ok = (e < STAGE) && gy >= 0 && gy < H && gx >= 0 && gx < W;
    s_lds[it] = (e < STAGE) ? c * CS + r * PITCH + q : -1;
    s_voff[it] = ok ? (unsigned)((c * HW + gy * W + gx) * 4) : kOOB;
  }
  const int chunk_bytes = KC * HW * 4;

  // ---- operand addressing ---------------------------------------------------
  const int kk = lane >> 5;                  // which channel of the k-pair
  const int prow = (lane >> 4) & 1;          // row inside the 2-row pixel tile
  const int pcol = lane & 15;
  // B base: channel kk, row (wp*PT*2 + prow), col pcol   (+ tap/ch/tile imms)
  const float* bbase = lds + kk * CS + (wp * PT * 2 + prow) * PITCH + pcol;
  const float4* astream = reinterpret_cast<const float4*>(L.packed) +
                          ((long long)(active ? mtile : 0) * args.chunks) * 9 * 64 + lane;

  f32x16 acc[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  float sreg[SITER];
  auto stage_load = [&](int ch) {
    const int soff = __builtin_amdgcn_readfirstlane(ch * chunk_bytes);
#pragma unroll
    for (int it = 0; it < SITER; ++it)
      sreg[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
          xrsrc, s_voff[it], soff, 0));
  };
  auto stage_store = [&](float* buf) {
#pragma unroll
    for (int it = 0; it < SITER; ++it)
      if (s_lds[it] >= 0) buf[s_lds[it]] = sreg[it];
  };

  // ---- prologue: stage chunk 0 ----------------------------------------------
  stage_load(0);
  stage_store(lds);
  __syncthreads();

  const int chunks = args.chunks;
  if (active) {
    float4 a0 = astream[0];
    float4 a1 = astream[64];
    for (int ch = 0; ch < chunks; ++ch) {
      const float* buf = bbase + (ch & 1) * BUF;
      const bool more = ch + 1 < chunks;
      if (more) stage_load(ch + 1);          // lands while we compute
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        // prefetch the A operands two taps ahead (linear stream)
        const float4 a2 = astream[(long long)(ch * 9 + tap + 2) * 64];
        const int r = tap / 3, s = tap % 3;
        const float av[4] = {a0.x, a0.y, a0.z, a0.w};
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
#pragma unroll
          for (int t = 0; t < PT; ++t) {
            const float b = buf[(2 * cp) * CS + (2 * t + r) * PITCH + s];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cp], b, acc[t], 0, 0, 0);
          }
        }
        a0 = a1;
        a1 = a2;
        // keep each tap's {A wait, 16 LDS reads, 4*PT MFMAs} together
        if constexpr (TAP_FENCE) __builtin_amdgcn_sched_barrier(0);
      }
      if (more) stage_store(lds + ((ch + 1) & 1) * BUF);
      __syncthreads();
    }
  } else {
    // a wave past the last output-channel tile only helps staging
    for (int ch = 0; ch < chunks; ++ch) {
      const bool more = ch + 1 < chunks;
      if (more) {
        stage_load(ch + 1);
        stage_store(lds + ((ch + 1) & 1) * BUF);
      }
      __syncthreads();
    }
    return;
  }

  // ---- epilogue ----------------------------------------------------------------
  const int flags = args.flags;
  float* yout = L.y + (long long)n * M * HW;
  const float* aux = (flags & SSAD_CONV_MASK_AUX) ? L.aux + (long long)n * M * HW : nullptr;
  const int mbase = mtile * 32 + 4 * kk;
  // this lane's 16 output channels: bias read once, before any store
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = mbase + (r & 3) + 8 * (r >> 2);
    bv[r] = (L.bias && m < M) ? L.bias[m] : 0.0f;
  }
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int py = y0 + (wp * PT + t) * 2 + prow;
    const int px = x0 + pcol;
    const bool pin = py < H && px < W;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mbase + (r & 3) + 8 * (r >> 2);
      if (pin && m < M) {
        float v = acc[t][r] + bv[r];
        if (flags & SSAD_CONV_RELU) v = v > 0.0f ? v : 0.0f;
        if (flags & SSAD_CONV_SIGMOID) v = 1.0f / (1.0f + expf(-v));
        const int o = m * HW + py * W + px;
        if (aux) v = aux[o] > 0.0f ? v : 0.0f;
        yout[o] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Weight gradient
// ---------------------------------------------------------------------------

constexpr int WG_MT = 4;      // m-tiles (of 32) per workgroup
constexpr int WG_CT = 2;      // c-tiles (of 32) per workgroup
constexpr int WG_PR = 4;      // patch rows -> 64 pixels = 32 k-steps
constexpr int WG_DS = 65;     // dY LDS stride per output channel (odd)
constexpr int WG_XS = 109;    // X  LDS stride per input channel (6*18 = 108, odd+1)
constexpr int WG_DYF = WG_MT * 32 * WG_DS;
constexpr int WG_XF = WG_CT * 32 * WG_XS;
constexpr int WG_BUF = WG_DYF + WG_XF;

struct WgLevel {
  const float* x;
  const float* dy;
  int N, H, W;
  int tiles_x, tiles_y;
  int patch_start;        // first global patch index of this level
};

struct WgArgs {
  WgLevel lv[SSAD_MAX_LEVELS];
  int n_levels;
  int M, K;               // Cout, Cin
  int mblocks, cblocks;   // ceil(M/128), ceil(K/64)
  int total_patches;
  int splits;
  float* slabs;           // [split][mtile_pad][ctile_pad][9][16][64]
};

__global__ __launch_bounds__(kBlock, 2) void conv3x3_wgrad_kernel(const WgArgs args) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wmt = wave % WG_MT, wct = wave / WG_MT;
  const int split = blockIdx.x;
  const int mb = blockIdx.y, cb = blockIdx.z;
  const int M = args.M, K = args.K;
  const int m_base = mb * (WG_MT * 32), c_base = cb * (WG_CT * 32);

  // patches [p_begin, p_end) of the flattened (level, image, ty, tx) list
  const int P = args.total_patches;
  const int p_begin = (int)((long long)P * split / args.splits);
  const int p_end = (int)((long long)P * (split + 1) / args.splits);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  const int kk = lane >> 5, li = lane & 31;
  const float* a_base = wlds + (wmt * 32 + li) * WG_DS + kk;
  const float* b_base = wlds + WG_DYF + (wct * 32 + li) * WG_XS + kk;

  // Staging maps, chosen so that every per-thread quantity is either fixed
  // for the whole kernel or advances by a wave-uniform stride per iteration:
  //   dY: thread -> pixel (tid & 63) of the 4x16 patch, channels (tid>>6) + 8*it
  //   X : thread -> halo position (tid & 127) of the 6x18 patch (108 used),
  //       channels (tid>>7) + 4*it
  constexpr int DY_IT = WG_MT * 32 / 8;                        // 16
  constexpr int X_IT = WG_CT * 32 / 4;                         // 16
  constexpr int X_POS = (WG_PR + 2) * (TW + 2);                // 108
  float dreg[DY_IT], xreg[X_IT];
  const int d_px = tid & 63, d_m0 = tid >> 6;
  const int d_r = d_px >> 4, d_c = d_px & 15;
  const int x_pos = tid & 127, x_c0 = tid >> 7;
  const int x_r = x_pos / (TW + 2), x_q = x_pos - x_r * (TW + 2);
  const bool x_used = x_pos < X_POS;

  // Patch being prefetched.  Staging loads are raw buffer loads: the
  // descriptor (wave-uniform, SGPRs) bounds the channel range, the per-lane
  // byte offset is ONE VGPR that is pushed out of range for pixels outside the
  // image, and the per-iteration channel step goes through the scalar offset.
  // Out-of-range reads return 0, so there is no select after the load and all
  // 16 loads of a half-patch are in flight together.
  constexpr unsigned kOOB = 0x80000000u;
  __amdgpu_buffer_rsrc_t q_dy_rsrc, q_x_rsrc;
  unsigned q_dy_voff = kOOB, q_x_voff = kOOB;
  int q_HW4 = 0;
  auto locate = [&](int p) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < SSAD_MAX_LEVELS; ++i)
      if (i < args.n_levels && p >= args.lv[i].patch_start) l = i;
    const WgLevel& L = args.lv[l];
    int q = p - L.patch_start;
    const int per_img = L.tiles_x * L.tiles_y;
    const int n = q / per_img;
    q -= n * per_img;
    const int ty = q / L.tiles_x, tx = q - ty * L.tiles_x;
    const int y0 = ty * WG_PR, x0 = tx * TW;
    const int H = L.H, W = L.W;
    const int HW = H * W;
    q_HW4 = HW * 4;
    const int m_left = M - m_base < WG_MT * 32 ? M - m_base : WG_MT * 32;
    const int c_left = K - c_base < WG_CT * 32 ? K - c_base : WG_CT * 32;
    float* dyp = const_cast<float*>(L.dy) + ((long long)n * M + m_base) * HW;
    float* xp = const_cast<float*>(L.x) + ((long long)n * K + c_base) * HW;
    q_dy_rsrc = uniform_rsrc(dyp, (m_left > 0 ? m_left : 0) * q_HW4);
    q_x_rsrc = uniform_rsrc(xp, (c_left > 0 ? c_left : 0) * q_HW4);
    {
      const int gy = y0 + d_r, gx = x0 + d_c;
      const bool pin = gy < H && gx < W;
      q_dy_voff = pin ? (unsigned)((d_m0 * HW + gy * W + gx) * 4) : kOOB;
    }
    {
      const int gy = y0 - 1 + x_r, gx = x0 - 1 + x_q;
      const bool pin = x_used && gy >= 0 && gy < H && gx >= 0 && gx < W;
      q_x_voff = pin ? (unsigned)((x_c0 * HW + gy * W + gx) * 4) : kOOB;
    }
  };
  auto load_dy = [&]() {
#pragma unroll
    for (int it = 0; it < DY_IT; ++it)
      dreg[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
          q_dy_rsrc, q_dy_voff, __builtin_amdgcn_readfirstlane(it * 8 * q_HW4), 0));
  };
  auto load_x = [&]() {
#pragma unroll
    for (int it = 0; it < X_IT; ++it)
      xreg[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
          q_x_rsrc, q_x_voff, __builtin_amdgcn_readfirstlane(it * 4 * q_HW4), 0));
  };
  auto store_dy = [&](float* buf) {
#pragma unroll
    for (int it = 0; it < DY_IT; ++it) buf[(d_m0 + it * 8) * WG_DS + d_px] = dreg[it];
  };
  auto store_x = [&](float* buf) {
    if (x_used) {
#pragma unroll
      for (int it = 0; it < X_IT; ++it)
        buf[WG_DYF + (x_c0 + it * 4) * WG_XS + x_pos] = xreg[it];
    }
  };
  auto compute = [&](const float* ab, const float* bb, int t_begin) {
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) {
      const int t = t_begin + tt;
      const float a = ab[2 * t];
      const int prow = t >> 3, pcol = 2 * (t & 7);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int r = tap / 3, s = tap % 3;
        const float b = bb[(prow + r) * (TW + 2) + pcol + s];
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tap], 0, 0, 0);
      }
    }
  };

  if (p_begin < p_end) {
    locate(p_begin);
    load_dy();
    load_x();
    store_dy(wlds);
    store_x(wlds);
  }
  __syncthreads();

  // Per patch: 32 k-steps (2 pixels each) x 9 taps.  The next patch is staged
  // in two halves (dY during k-steps 0-15, X during 16-31) so that at most 16
  // staging registers are live beside the 144 accumulators.
  for (int p = p_begin; p < p_end; ++p) {
    const int cur = (p - p_begin) & 1;
    const bool more = p + 1 < p_end;
    float* nxt = wlds + (cur ^ 1) * WG_BUF;
    const float* ab = a_base + cur * WG_BUF;
    const float* bb = b_base + cur * WG_BUF;
    if (more) { locate(p + 1); load_dy(); }
    compute(ab, bb, 0);
    if (more) { store_dy(nxt); load_x(); }
    compute(ab, bb, 16);
    if (more) store_x(nxt);
    __syncthreads();
  }

  // partial slab: [split][mtile][ctile][tap][reg][lane]
  const int mtp = args.mblocks * WG_MT, ctp = args.cblocks * WG_CT;
  const int mt = mb * WG_MT + wmt, ct = cb * WG_CT + wct;
  float* out = args.slabs + ((((long long)split * mtp + mt) * ctp + ct) * 9) * 1024 + lane;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(tap * 16 + r) * 64] = acc[tap][r];
}

// dW[m][c][tap] (+)= sum_split slab[...]; one thread per slab element.
__global__ void wgrad_reduce_kernel(
    const float* __restrict__ slabs, int splits, int mtp, int ctp, int M, int K,
    float* __restrict__ dW, int accumulate) {
  const long long per_split = (long long)mtp * ctp * 9 * 1024;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= per_split) return;
  const int lane = e & 63, reg = (e >> 6) & 15;
  long long r = e >> 10;
  const int tap = r % 9; r /= 9;
  const int ct = r % ctp; const int mt = r / ctp;
  const int i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
  const int m = mt * 32 + i, c = ct * 32 + (lane & 31);
  if (m >= M || c >= K) return;
  float s = 0.0f;
  for (int sp = 0; sp < splits; ++sp) s += slabs[sp * per_split + e];
  float* o = dW + ((long long)m * K + c) * 9 + tap;
  *o = accumulate ? *o + s : s;
}

// db[m] (+)= sum over levels, images, pixels of dY.  HBM-streaming: workgroup
// (m, part) sums the (level, image) rows r with r % kBiasParts == part using
// 16-byte loads; partials are combined in a fixed order (deterministic).
constexpr int kBiasParts = 8;
struct BiasArgs {
  const float* dy[SSAD_MAX_LEVELS];
  int N[SSAD_MAX_LEVELS];
  int HW[SSAD_MAX_LEVELS];
  int n_levels;
  int M;
};

__global__ __launch_bounds__(256) void bias_grad_kernel(
    const BiasArgs args, double* __restrict__ partials) {
  const int m = blockIdx.x, part = blockIdx.y;
  double acc = 0.0;
  int row = 0;
  for (int l = 0; l < args.n_levels; ++l) {
    const int HW = args.HW[l];
    for (int n = 0; n < args.N[l]; ++n, ++row) {
      if (row % kBiasParts != part) continue;
      const float* p = args.dy[l] + ((long long)n * args.M + m) * HW;
      float a = 0.0f;
      if ((((uintptr_t)p) & 15) == 0) {
        const int n4 = HW >> 2;
        const float4* p4 = reinterpret_cast<const float4*>(p);
        int i = threadIdx.x;
        for (; i + 768 < n4; i += 1024) {
          const float4 v0 = p4[i], v1 = p4[i + 256], v2 = p4[i + 512], v3 = p4[i + 768];
          a += (v0.x + v0.y + v0.z + v0.w) + (v1.x + v1.y + v1.z + v1.w) +
               (v2.x + v2.y + v2.z + v2.w) + (v3.x + v3.y + v3.z + v3.w);
        }
        for (; i < n4; i += 256) { const float4 v = p4[i]; a += v.x + v.y + v.z + v.w; }
        for (int j = n4 * 4 + threadIdx.x; j < HW; j += 256) a += p[j];
      } else {
        for (int i = threadIdx.x; i < HW; i += 256) a += p[i];
      }
      acc += (double)a;
    }
  }
  __shared__ double ws[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[m * kBiasParts + part] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ void bias_grad_finalize_kernel(const double* __restrict__ partials, int M,
                                          float* __restrict__ db, int accumulate) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < kBiasParts; ++k) t += partials[m * kBiasParts + k];
  db[m] = accumulate ? db[m] + (float)t : (float)t;
}

void launch_bias_grad(const ssad_conv_level* levels_host, int n_levels, int Cout, float* db, int accumulate,
                      double* partials, hipStream_t s) {
  BiasArgs b;
  b.n_levels = n_levels; b.M = Cout;
  for (int l = 0; l < SSAD_MAX_LEVELS; ++l) {
    b.dy[l] = l < n_levels ? levels_host[l].aux : nullptr;
    b.N[l] = l < n_levels ? levels_host[l].N : 0;
    b.HW[l] = l < n_levels ? levels_host[l].H * levels_host[l].W : 0;
  }
  hipLaunchKernelGGL(bias_grad_kernel, dim3(Cout, kBiasParts), dim3(256), 0, s, b, partials);
  hipLaunchKernelGGL(bias_grad_finalize_kernel, dim3((Cout + 255) / 256), dim3(256), 0, s,
                     (const double*)partials, Cout, db, accumulate);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

template <int WM, int WP, int PT, bool TAP_FENCE>
int launch_fwd(const ssad_conv_level* lv, int n_levels, const float* packed,
               const float* bias, int M, int K, int flags, hipStream_t s) {
  constexpr int PR = 2 * PT * WP;
  FwdArgs a;
  a.n_levels = n_levels;
  a.packed = packed; a.bias = bias; a.M = M; a.K = K; a.chunks = cdiv(K, KC);
  a.flags = flags;
  long long blocks = 0;
  for (int l = 0; l < n_levels; ++l) {
    FwdLevel& L = a.lv[l];
    L.x = lv[l].x; L.y = lv[l].y; L.aux = lv[l].aux;
    L.packed = lv[l].packed ? lv[l].packed : packed;
    L.bias = lv[l].packed ? lv[l].bias : bias;
    if (!L.packed) return SSAD_E_BADARG;
    L.N = lv[l].N; L.H = lv[l].H; L.W = lv[l].W;
    if (L.N < 0 || L.H < 0 || L.W < 0) return SSAD_E_BADARG;
    if ((long long)L.H * L.W * (K > M ? K : M) >= (1LL << 29)) return SSAD_E_BADARG;
    if ((flags & SSAD_CONV_MASK_AUX) && !L.aux) return SSAD_E_BADARG;
    L.tiles_x = cdiv(L.W, TW); L.tiles_y = cdiv(L.H, PR);
    L.block_start = (int)blocks;
    blocks += (long long)L.N * L.tiles_x * L.tiles_y;
    if (blocks >= (1LL << 31)) return SSAD_E_BADARG;
  }
  for (int l = n_levels; l < SSAD_MAX_CONV_PROBLEMS; ++l) a.lv[l] = FwdLevel{};
  if (blocks == 0) return 0;
  const int gy = cdiv(cdiv(M, 32), WM);
  hipLaunchKernelGGL((conv3x3_kernel<WM, WP, PT, TAP_FENCE>), dim3((unsigned)blocks, gy),
                     dim3(kBlock), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

size_t ssad_conv_packed_filter_floats(int M, int K) {
  return (size_t)cdiv(M, 32) * cdiv(K, KC) * 9 * 256 + 512;
}

int ssad_conv_pack_filter(const float* w, int Cout, int Cin, float* packed_fwd,
                          float* packed_dgrad, ssad_stream_t stream) {
  if (Cout <= 0 || Cin <= 0 || !w) return SSAD_E_BADARG;
  const size_t nf = ssad_conv_packed_filter_floats(Cout, Cin);
  const size_t nd = ssad_conv_packed_filter_floats(Cin, Cout);
  const size_t n = nf > nd ? nf : nd;
  hipLaunchKernelGGL(pack_filter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, Cout, Cin, packed_fwd, packed_dgrad);
  return (int)hipGetLastError();
}

int ssad_conv3x3_forward(const ssad_conv_level* levels_host, int n_levels,
                         const float* packed, const float* bias, int Cout, int Cin,
                         int flags, ssad_stream_t stream) {
  if (n_levels < 1 || n_levels > SSAD_MAX_CONV_PROBLEMS || Cout <= 0 || Cin <= 0)
    return SSAD_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (Cout <= 64)
    return launch_fwd<2, 4, 2, true>(levels_host, n_levels, packed, bias, Cout, Cin, flags, s);
  // patch = 8 rows (PT = 4) or 4 rows (PT = 2): pick the one whose workgroup
  // count quantises better onto the 256 CUs (all workgroups cost the same).
  int use_pt2 = 0;
  {
    long long t4 = 0, t2 = 0;
    for (int l = 0; l < n_levels; ++l) {
      const long long tx = cdiv(levels_host[l].W, TW);
      t4 += (long long)levels_host[l].N * tx * cdiv(levels_host[l].H, 8);
      t2 += (long long)levels_host[l].N * tx * cdiv(levels_host[l].H, 4);
    }
    const int gy = cdiv(cdiv(Cout, 32), 8);
    auto makespan = [&](long long tiles, double unit) {
      const long long blocks = tiles * gy;
      return (double)((blocks + 255) / 256) * unit;
    };
    use_pt2 = makespan(t2, 0.52) < makespan(t4, 1.0);
  }
  // (TAP_FENCE on: 5-10 % faster than free scheduling, round 1; the unfenced instantiations are gone)
  if (use_pt2) return launch_fwd<8, 1, 2, true>(levels_host, n_levels, packed, bias, Cout, Cin, flags, s);
  return launch_fwd<8, 1, 4, true>(levels_host, n_levels, packed, bias, Cout, Cin, flags, s);
}

static int wgrad_plan(const ssad_conv_level* lv, int n_levels, int Cout, int Cin, WgArgs* a) {
  if (n_levels < 1 || n_levels > SSAD_MAX_LEVELS || Cout <= 0 || Cin <= 0) return SSAD_E_BADARG;
  a->n_levels = n_levels;
  a->M = Cout; a->K = Cin;
  a->mblocks = cdiv(Cout, WG_MT * 32);
  a->cblocks = cdiv(Cin, WG_CT * 32);
  long long patches = 0;
  for (int l = 0; l < n_levels; ++l) {
    WgLevel& L = a->lv[l];
    L.x = lv[l].x; L.dy = lv[l].aux;
    L.N = lv[l].N; L.H = lv[l].H; L.W = lv[l].W;
    if (L.N < 0 || L.H < 0 || L.W < 0) return SSAD_E_BADARG;
    if ((long long)L.H * L.W * (Cin > Cout ? Cin : Cout) >= (1LL << 29)) return SSAD_E_BADARG;
    L.tiles_x = cdiv(L.W, TW); L.tiles_y = cdiv(L.H, WG_PR);
    L.patch_start = (int)patches;
    patches += (long long)L.N * L.tiles_x * L.tiles_y;
    if (patches >= (1LL << 31)) return SSAD_E_BADARG;
  }
  for (int l = n_levels; l < SSAD_MAX_LEVELS; ++l) a->lv[l] = WgLevel{};
  a->total_patches = (int)patches;
  const int ob = a->mblocks * a->cblocks;
  int splits = 256 / ob;                       // one workgroup per CU
  if (splits < 1) splits = 1;
  if (splits > patches) splits = (int)(patches > 0 ? patches : 1);
  a->splits = splits;
  return 0;
}

size_t ssad_conv3x3_wgrad_workspace_bytes(const ssad_conv_level* levels_host, int n_levels,
                                          int Cout, int Cin) {
  WgArgs a;
  if (wgrad_plan(levels_host, n_levels, Cout, Cin, &a)) return 0;
  size_t slab = sizeof(float) * (size_t)a.splits * a.mblocks * WG_MT * a.cblocks * WG_CT * 9 * 1024;
  // sized for either engine, so the choice can change between calls
  const size_t wino = ssad_wino_wgrad_workspace_bytes(levels_host, n_levels, Cout, Cin);
  if (wino > slab) slab = wino;
  return slab + sizeof(double) * (size_t)Cout * kBiasParts;
}

int ssad_conv3x3_wgrad(const ssad_conv_level* levels_host, int n_levels, float* dW, float* db,
                       int Cout, int Cin, int accumulate, void* workspace,
                       size_t workspace_bytes, ssad_stream_t stream) {
  WgArgs a;
  const int rc = wgrad_plan(levels_host, n_levels, Cout, Cin, &a);
  if (rc) return rc;
  for (int l = 0; l < n_levels; ++l)
    if (!levels_host[l].aux && levels_host[l].N * levels_host[l].H * levels_host[l].W > 0)
      return SSAD_E_BADARG;
  const int mtp = a.mblocks * WG_MT, ctp = a.cblocks * WG_CT;
  size_t slab_bytes = sizeof(float) * (size_t)a.splits * mtp * ctp * 9 * 1024;
  const size_t wino_bytes = ssad_wino_wgrad_workspace_bytes(levels_host, n_levels, Cout, Cin);
  if (wino_bytes > slab_bytes) slab_bytes = wino_bytes;
  const size_t need = slab_bytes + sizeof(double) * (size_t)Cout * kBiasParts;
  if (!workspace || workspace_bytes < need) return SSAD_E_WORKSPACE;
  a.slabs = (float*)workspace;
  hipStream_t s = (hipStream_t)stream;
  if (ssad_wino_wgrad_eligible(Cout, Cin)) {
    const int wrc = ssad_wino_wgrad_launch(levels_host, n_levels, dW, Cout, Cin, accumulate,
                                           workspace, slab_bytes, s);
    if (wrc) return wrc;
  } else {
    const size_t lds_bytes = sizeof(float) * 2 * WG_BUF;
    static std::once_flag lds_once;    // > 64 KiB of dynamic LDS needs the opt-in, once per process
    std::call_once(lds_once, [&] {
      (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    });
    hipLaunchKernelGGL(conv3x3_wgrad_kernel, dim3(a.splits, a.mblocks, a.cblocks), dim3(kBlock),
                       lds_bytes, s, a);
    const long long per_split = (long long)mtp * ctp * 9 * 1024;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((per_split + 255) / 256)), dim3(256),
                       0, s, (const float*)a.slabs, a.splits, mtp, ctp, Cout, Cin, dW, accumulate);
  }
  if (db) launch_bias_grad(levels_host, n_levels, Cout, db, accumulate, (double*)((char*)workspace + slab_bytes), s);
  return (int)hipGetLastError();
}

/* The same contract on the split-operand engine (conv3x3_wgrad_split.hip). */
size_t ssad_conv3x3_wgrad_split_workspace_bytes(const ssad_conv_level* levels_host, int n_levels, int Cout, int Cin) {
  const size_t slab = ssad_split_wgrad_workspace_bytes(levels_host, n_levels, Cout, Cin);
  if (!slab) return 0;
  return ((slab + 255) & ~(size_t)255) + sizeof(double) * (size_t)Cout * kBiasParts;
}

int ssad_conv3x3_wgrad_split(const ssad_conv_level* levels_host, int n_levels, float* dW, float* db, int Cout, int Cin,
                             int accumulate, void* workspace, size_t workspace_bytes, ssad_stream_t stream) {
  return ssad_conv3x3_wgrad_split_amax(levels_host, n_levels, dW, db, Cout, Cin, accumulate, workspace, workspace_bytes,
                                       nullptr, nullptr, stream);
}

int ssad_conv3x3_wgrad_split_amax(const ssad_conv_level* levels_host, int n_levels, float* dW, float* db, int Cout,
                                  int Cin, int accumulate, void* workspace, size_t workspace_bytes,
                                  const unsigned* x_amax, const unsigned* dy_amax, ssad_stream_t stream) {
  if (!levels_host || !dW || (x_amax == nullptr) != (dy_amax == nullptr)) return SSAD_E_BADARG;
  const size_t slab = ssad_split_wgrad_workspace_bytes(levels_host, n_levels, Cout, Cin);
  if (!slab) return SSAD_E_BADARG;
  const size_t slab_bytes = (slab + 255) & ~(size_t)255;
  if (!workspace || workspace_bytes < slab_bytes + sizeof(double) * (size_t)Cout * kBiasParts) return SSAD_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int rc = ssad_split_wgrad_launch(levels_host, n_levels, dW, Cout, Cin, accumulate, workspace, slab_bytes,
                                         x_amax, dy_amax, s);
  if (rc) return rc;
  if (db) launch_bias_grad(levels_host, n_levels, Cout, db, accumulate, (double*)((char*)workspace + slab_bytes), s);
  return (int)hipGetLastError();
}

}  // extern "C"
