// conv3x3_winograd24.hip -- Winograd F(2x4, 3x3) forward / data-gradient engine (round 5).
//
// Same operator contract as conv3x3_winograd.hip (3x3, stride 1, pad 1, NCHW fp32;
// caffe2/operators/conv_op_cudnn.cc:567-617 forward, :1040-1058 data gradient), same persistent kernel skeleton --
// LDS-DMA staging of the raw 10 x 18 patch (or two 10 x 10 sub-patches), filter operands through a hand-counted
// register ring, input transform threaded through the MFMA steps -- but the tile is 2 rows x 4 columns:
//
//     Y = A2^T [ (G2 g G4^T) (.) (B2^T d B4) ] A4        F(2,3) down the rows, F(4,3) along them
//
// 24 products per 8 outputs = 3 multiplies per output instead of F(2x2)'s 4 (direct: 9).  The 8 x 16-pixel patch
// of the F(2x2) kernel is 4 x 4 such tiles = ONE 16-tile MFMA column group, so a wave holds 16 channels x 16 tiles
// x 24 products = 96 accumulator VGPRs and issues 96 `v_mfma_f32_16x16x4_f32` per 16-channel chunk where the
// F(2x2) kernel issues 128; everything around the MFMAs (patch DMA, transform cost per chunk, barrier) stays what
// it was.  Measured on the F(2x2) kernel with a quarter of its MFMAs compiled out (WINO_ABLATE 64): -19 % per
// launch; this kernel: see DESIGN.md 3.10.
//
// Accuracy: F(4,3)'s transforms multiply by 4, 5, 8 where F(2,3) only adds.  Against a float64 convolution:
// 1.5-2.2e-6 of the output scale, F(2x2) 0.8-1.9e-6 on the same inputs (tools/dbg/r5_f24_check.py); through the
// whole subnets step (ten layers forward, ten backward) every gradient tensor within 9.3e-6 relative L2 of the
// oracle where F(2x2) is within 5.1e-6 (tools/dbg/r5_chain_flips.py, seeds without a ReLU-mask flip) -- an order
// inside the 1e-4 parity bar either way (F(4x4) in both directions measured 4-8e-6 per layer and was not built).
// Who uses it: the frozen teacher (round 5, first: model_builder.py:373-411 builds it in test mode) and, since the
// end of round 5, the trained networks' forward pass and data gradient too (the same kernel on the flipped /
// transposed pack with the fused ReluGradient mask); the filter gradient keeps its F(3x3, 2x2) engine.  Layers with
// fewer than 128 outputs stay on the F(2x2) kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "conv_internal.h"
#include "ssad_kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// W24_ABLATE (debug builds only, wrong results): 1 no input transform, 2 no B-operand LDS reads in the step loop,
// 4 no filter-operand ring reloads / waits, 8 no end-of-chunk barrier, 16 no patch DMA, 32 no epilogue stores
// 64 patch DMA issued but every lane out of range (no memory traffic), 128 filter ring loads issued out of range,
// 512 every OTHER filter ring load out of range (round 6: the filter traffic of an A-operand-reuse-2 design, patch
// traffic unchanged -- the best case of that design on this skeleton)
// (tools/dbg/w24_ablate.sh: what a chunk's time is made of)
#ifndef W24_ABLATE
#define W24_ABLATE 0
#endif

constexpr int KC = 16;                   // input channels per chunk
constexpr int KS = KC / 4;               // MFMA k-steps per chunk
constexpr int XQ = 6;                    // column positions of the transformed tile (F(4,3): 6)
constexpr int XI = 4 * XQ;               // products per tile: 4 row positions x 6 column positions
constexpr int STEPS = KS * XQ;           // A-stream float4 per lane per chunk = MFMA steps of 4
constexpr int PR = 8, PC = 16;           // output patch (4 x 4 tiles of 2 x 4 pixels)
constexpr int SP = 8;                    // sub-patch edge (4 x 2 tiles)
constexpr int kBlock = 512;
constexpr int BM = 128;
constexpr int ZP = 40;                   // raw row pitch (two channels side by side, 20 columns each)
constexpr int ZCP = (PR + 2) * ZP;       // floats per channel pair
constexpr int ZRAW = (KC / 2) * ZCP;     // 3200 floats = 50 wave-loads
constexpr int ZL = 7;                    // wave-loads per wave per chunk
constexpr int ZRAWP = 8 * ZL * 64;       // padded raw buffer
constexpr int NRAW = 3;
constexpr int ZNT = 256;                 // work items per workgroup in the LDS list
constexpr int AD = 8;                    // filter operand ring depth (steps)
constexpr int VP = 16;                   // V row: the 16 tiles of one (product, channel)
constexpr int VBUF = XI * KC * VP;       // transformed floats per buffer (24 KB)
constexpr unsigned kOOBOff = 0x80000000u;
constexpr int kNoSub = 1 << 24;
static_assert(STEPS == 24 && AD == 8 && ZL == 7, "the counted waits below are written for these");

__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
using ssad_dev::uniform_rsrc;

// U = G2 g G4^T (4 x 6), in MFMA A-operand order:
//   packed[mt][chunk][ks][xq][lane][xr], lane = k * 16 + i:  U[a = xr][b = xq][out = mt*16 + i][in = chunk*KC + ks*4 + k]
__device__ __forceinline__ float wino24_u(const float* g, int a, int b) {
  // rows of G2 g (3 columns each)
  float r[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
    r[j] = a == 0 ? g0 : a == 1 ? 0.5f * (g0 + g1 + g2) : a == 2 ? 0.5f * (g0 - g1 + g2) : g2;
  }
  // ... times G4^T: (1/4, 0, 0), (-1/6)(1, 1, 1), (-1/6)(1, -1, 1), (1/24, 1/12, 1/6), (1/24, -1/12, 1/6), (0, 0, 1)
  switch (b) {
    case 0: return 0.25f * r[0];
    case 1: return (-1.0f / 6.0f) * (r[0] + r[1] + r[2]);
    case 2: return (-1.0f / 6.0f) * (r[0] - r[1] + r[2]);
    case 3: return (1.0f / 24.0f) * r[0] + (1.0f / 12.0f) * r[1] + (1.0f / 6.0f) * r[2];
    case 4: return (1.0f / 24.0f) * r[0] - (1.0f / 12.0f) * r[1] + (1.0f / 6.0f) * r[2];
    default: return r[2];
  }
}

struct PackTable {
  ssad_pack_entry e[SSAD_MAX_PACK_ENTRIES];
};
__global__ void wino24_pack_multi_kernel(const PackTable t) {
  const ssad_pack_entry& e = t.e[blockIdx.y];
  // z = 0: forward filter (M = Cout outputs, K = Cin inputs); z = 1: the data gradient's filter, flipped and
  // transposed (M = Cin, K = Cout: dX = conv(dY, W'), W'[ci][co][k] = W[co][ci][8 - k])
  const bool dg = blockIdx.z == 1;
  float* dst = dg ? e.packed_dgrad : e.packed_fwd;
  if (!dst) return;
  const int M = dg ? e.Cin : e.Cout, K = dg ? e.Cout : e.Cin;
  const int mtiles = cdiv(M, 16), chunks = cdiv(K, KC);
  const long long total = (long long)mtiles * chunks * STEPS * 256;
  for (long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x; tid < total + 1024;
       tid += (long long)gridDim.x * blockDim.x) {
    float v = 0.0f;
    if (tid < total) {
      const int xr = tid & 3, lane = (tid >> 2) & 63;
      long long r = tid >> 8;
      const int xq = (int)(r % XQ); r /= XQ;
      const int ks = (int)(r % KS); r /= KS;
      const int chunk = (int)(r % chunks), mt = (int)(r / chunks);
      const int out = mt * 16 + (lane & 15), in = chunk * KC + ks * 4 + (lane >> 4);
      if (out < M && in < K) {
        float g[9];
        const float* w = dg ? e.w + ((long long)in * e.Cin + out) * 9 : e.w + ((long long)out * e.Cin + in) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) g[k] = w[dg ? 8 - k : k];
        v = wino24_u(g, xr, xq);
      }
    }
    dst[tid] = v;
  }
}

struct WLevel {
  const float* x;
  float* y;
  const float* aux;                      // SSAD_CONV_MASK_AUX: the forward output whose sign masks y (fused ReluGradient)
  const float* packed;
  const float* bias;
  int N, H, W;
  int tiles_x, tiles_y, block_start;     // 8 x 16 patches
  int sub_x, sub_y, pair_start;          // pairs of 8 x 8 sub-patches
};
struct WArgs {
  WLevel lv[SSAD_MAX_CONV_PROBLEMS];
  int n_levels;
  int M, K, chunks, flags;
  int patches, mblocks, items;
  int xcd_group;
};
struct WTile {
  int l, mb;
  int n[2], y0[2], x0[2];
};

// See wino_conv_z_kernel (conv3x3_winograd.hip) for the skeleton and how it got there; what differs is marked F24.
template <bool PAIRS>
__global__ __launch_bounds__(kBlock, 1) void wino24_conv_kernel(const WArgs args) {
  __shared__ float raw[NRAW * ZRAWP];
  __shared__ float vbuf[2 * VBUF];
  __shared__ int lv_start[32], lv_tx[32], lv_per[32];
  __shared__ int trec[ZNT * 8];
  __shared__ unsigned zvoff[8 * ZL * 64];

  const int K = args.K, M = args.M;
  const int chunks = args.chunks;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = args.items;
  const int G = (int)gridDim.x;
  int slot = (int)blockIdx.x;
  {
    const int xg = args.xcd_group;
    if (xg > 1 && (G & 7) == 0 && ((G >> 3) % xg) == 0) {
      const int r = slot >> 3, x = slot & 7;
      slot = (r / xg) * (8 * xg) + x * xg + (r % xg);
    }
  }
  const int my_n = total > slot ? (total - slot + G - 1) / G : 0;
  const int S = my_n * chunks;

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
    const int t = slot + i * G;
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
      r[4 + 3 * h] = PAIRS ? sx * SP : sx * PC + h * SP;
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

  // ---- staging (as wino_conv_z_kernel) ----
  ssad_dev::rsrc_words xrs = ssad_dev::uniform_rsrc_words(args.lv[0].x, 0);
  int chunk_bytes = 0;
  int ld_tile = 0, ld_ch = 0, ld_buf = 0;
  unsigned* myvoff = zvoff + wave * (ZL * 64) + lane;
  const unsigned raw_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)raw;
  bool dma_real = true;
  int dma_soff = 0;
  unsigned dma_dst = 0;
  auto dma_begin = [&](bool real) {
    dma_real = real;
    if (real && ld_ch == 0) {
      const WTile Tt = get_tile(ld_tile);
      const WLevel& L = args.lv[Tt.l];
      const int H = L.H, W = L.W, HW = H * W;
      xrs = ssad_dev::uniform_rsrc_words(L.x, (unsigned)((long long)L.N * K * HW * 4));
      chunk_bytes = KC * HW * 4;
#pragma unroll 1
      for (int j = 0; j < ZL; ++j) {
        const int e = (wave + 8 * j) * 64 + lane;
        const int p = e / ZCP, rem = e - p * ZCP;
        const int r = rem / ZP, cq = rem - r * ZP;
        const int hi = cq >= ZP / 2 ? 1 : 0;
        const int q = cq - hi * (ZP / 2);
        const int sb = PAIRS && q >= SP + 2 ? 1 : 0;
        const int gy = (sb ? Tt.y0[1] : Tt.y0[0]) - 1 + r, gx = (sb ? Tt.x0[1] : Tt.x0[0]) - 1 + q - sb * (SP + 2);
        const int n = sb ? Tt.n[1] : Tt.n[0];
        const bool ok = (e < ZRAW) & (PAIRS || q < PC + 2) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
        myvoff[j * 64] = ok ? (unsigned)(((n * K + 2 * p + hi) * HW + gy * W + gx) * 4) : kOOBOff;
      }
    }
    dma_soff = __builtin_amdgcn_readfirstlane(real ? ld_ch * chunk_bytes : 0);
    dma_dst = __builtin_amdgcn_readfirstlane(raw_lds + (unsigned)(ld_buf * ZRAWP + wave * 64) * 4u);
  };
  auto dma_offset = [&](int j) { return (dma_real && !(W24_ABLATE & 64)) ? myvoff[j * 64] : kOOBOff; };
  auto dma_issue = [&](int j, unsigned vo) { ssad_dev::lds_dma<4>(xrs, dma_dst + j * 2048, vo, dma_soff); };
  auto dma_end = [&]() {
    if (dma_real) {
      if (++ld_ch == chunks) { ld_ch = 0; ++ld_tile; }
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

  // ---- F24 transform: work item = (tile of 16, channel, row position a); wave w owns a = w & 3 for channels
  //      rnd * 8 + (w >> 2) * 4 + (lane >> 4).  Row a of B2^T d is dA + sg dB (as F(2x2)); along the row the six
  //      columns of the tile's window go through B4^T:
  //        o0 = 4 t0 - 5 t2 + t4     o1 = (t4 - 4 t2) + (t3 - 4 t1)    o2 = (t4 - 4 t2) - (t3 - 4 t1)
  //        o5 = 4 t1 - 5 t3 + t5     o3 = (t4 - t2) + 2 (t3 - t1)      o4 = (t4 - t2) - 2 (t3 - t1)
  const int t_a = wave & 3;
  const int t_ra = t_a == 0 ? 0 : t_a == 2 ? 2 : 1;
  const int t_rb = t_a == 0 ? 2 : t_a == 1 ? 2 : t_a == 2 ? 1 : 3;
  const float t_sg = t_a == 1 ? 1.0f : -1.0f;
  const int t_tile = lane & 15, t_c = (wave >> 2) * 4 + (lane >> 4);
  // tile t of the work item: PAIRS: sub-patch t >> 3, tile row (t & 7) >> 1, tile column t & 1; else row t >> 2, column t & 3
  const int t_src = (t_c >> 1) * ZCP + (t_c & 1) * (ZP / 2) +
      (PAIRS ? (2 * ((t_tile & 7) >> 1)) * ZP + 4 * (t_tile & 1) + (t_tile >> 3) * (SP + 2)
             : (2 * (t_tile >> 2)) * ZP + 4 * (t_tile & 3));
  const int t_dst = (t_a * KC + t_c) * VP + t_tile;          // + b * 4 KC VP: V row = (4 b + a) KC + channel
  const int t_oa = t_src + t_ra * ZP, t_ob = t_src + t_rb * ZP;
  auto xf_load = [&](const float* rb0, int rnd, float2 (&d)[6]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      d[k] = *reinterpret_cast<const float2*>(rb0 + t_oa + rnd * 4 * ZCP + 2 * k);
      d[3 + k] = *reinterpret_cast<const float2*>(rb0 + t_ob + rnd * 4 * ZCP + 2 * k);
    }
  };
  auto xf_store = [&](float* vb0, int rnd, const float2 (&d)[6]) {
    const float t0 = fmaf(t_sg, d[3].x, d[0].x), t1 = fmaf(t_sg, d[3].y, d[0].y);
    const float t2 = fmaf(t_sg, d[4].x, d[1].x), t3 = fmaf(t_sg, d[4].y, d[1].y);
    const float t4 = fmaf(t_sg, d[5].x, d[2].x), t5 = fmaf(t_sg, d[5].y, d[2].y);
    const float p = fmaf(-4.0f, t2, t4), q = fmaf(-4.0f, t1, t3);
    const float r = t4 - t2, s = t3 - t1;
    float* o = vb0 + t_dst + rnd * 8 * VP;
    o[0 * 4 * KC * VP] = fmaf(4.0f, t0, fmaf(-5.0f, t2, t4));
    o[1 * 4 * KC * VP] = p + q;
    o[2 * 4 * KC * VP] = p - q;
    o[3 * 4 * KC * VP] = fmaf(2.0f, s, r);
    o[4 * 4 * KC * VP] = fmaf(-2.0f, s, r);
    o[5 * 4 * KC * VP] = fmaf(4.0f, t1, fmaf(-5.0f, t3, t5));
  };
  auto transform = [&](const float* rb0, float* vb0) {
#pragma unroll
    for (int rnd = 0; rnd < KC / 8; ++rnd) {
      float2 d[6];
      xf_load(rb0, rnd, d);
      xf_store(vb0, rnd, d);
    }
  };

  // ---- compute side ----
  const int kq = lane >> 4, jn = lane & 15;
  const float* bbase = vbuf + kq * VP + jn;
  const int mtiles = cdiv(M, 16);
  const int stream_bytes = (mtiles * chunks * STEPS * 256 + 1024) * 4;
  const unsigned a_voff = lane * 16;
  auto stream_off = [&](const WTile& Tt) {
    int mt = Tt.mb * (BM / 16) + wave;
    if (mt >= mtiles) mt = 0;
    return __builtin_amdgcn_readfirstlane(mt * chunks * STEPS * 1024);
  };
  auto a_load_at = [&](f32x4& dst, unsigned voff, const ssad_dev::rsrc_words& rs, int soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff));
  };
  auto a_load = [&](f32x4& dst, const ssad_dev::rsrc_words& rs, int soff) { a_load_at(dst, a_voff, rs, soff); };
  auto ring_landed = [&](f32x4 (&r)[AD]) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 :: "memory");
  };

  // ---- prologue ----
  WTile T = get_tile(0);
  ssad_dev::rsrc_words arsrc = ssad_dev::uniform_rsrc_words(args.lv[T.l].packed, (unsigned)stream_bytes);
  int abase = stream_off(T);
  f32x4 ar[AD];
#pragma unroll
  for (int k = 0; k < AD; ++k) a_load(ar[k], arsrc, abase + k * 1024);
  dma_next(true);
  if (S > 1) dma_next(true);
  if (S > 2) dma_next(true);
  ring_landed(ar);
  __syncthreads();
  transform(raw, vbuf);
  __syncthreads();

  int s = 0;
  int rbuf = 1;
  WTile Tn = T;
  ssad_dev::rsrc_words nrsrc = arsrc;
  int nbase = abase;
  const int look = chunks > 1 ? 1 : 0;
  for (int i = 0; i < my_n; ++i) {
    const int mt = T.mb * (BM / 16) + wave;
    const bool active = mt < mtiles;
    f32x4 acc[XI];
#pragma unroll
    for (int x = 0; x < XI; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < chunks; ++ch, ++s) {
      auto side_dma = [&]() { dma_next(s + 3 < S); };
      auto side_look = [&]() {
        if (ch == look && i + 1 < my_n) {
          Tn = get_tile(i + 1);
          nrsrc = ssad_dev::uniform_rsrc_words(args.lv[Tn.l].packed, (unsigned)stream_bytes);
          nbase = stream_off(Tn);
        }
      };
      const bool xf = s + 1 < S;
      const float* xsrc = raw + rbuf * ZRAWP;
      float* xdst = vbuf + ((s + 1) & 1) * VBUF;
      const bool last = ch == chunks - 1;
      const unsigned tail_voff = last ? kOOBOff : a_voff;
      const unsigned tail_voff_ab = (W24_ABLATE & 128) ? kOOBOff : tail_voff;
      if (active) {
        const float* vb = bbase + (s & 1) * VBUF;
        float bc[4];
#pragma unroll
        for (int xr = 0; xr < 4; ++xr) bc[xr] = vb[(xr * KC) * VP];
        float2 xd[6];
        unsigned dma_vo = kOOBOff;
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {       // step = ks * 6 + xq
          const int xq = step % XQ;
          const int nks = (step + 1) / XQ, nxq = (step + 1) % XQ;
          if (!(W24_ABLATE & 4)) {
            // ring slot of this step: 7 younger ring loads + the DMA instructions issued since it was requested
            // (conv3x3_winograd.hip: R(j) is requested at step j - 8, D_k at the end of step k <= 6)
            const int younger = step < AD ? (step < ZL ? step : ZL) : (ZL + AD - step > 0 ? ZL + AD - step : 0);
            switch (younger) {
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
            acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[xr], bc[xr], acc[xi], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(W24_ABLATE & 2) && step < STEPS - 1) bc[xr] = vb[((nxq * 4 + xr) * KC + nks * 4) * VP];
            __builtin_amdgcn_sched_barrier(0);
          }
          if (!(W24_ABLATE & 4))
            a_load_at(ar[step & (AD - 1)], (step + AD >= STEPS || (W24_ABLATE & 128) || ((W24_ABLATE & 512) && (step & 1))) ? ((W24_ABLATE & 512) && (step & 1) ? kOOBOff : tail_voff_ab) : a_voff, arsrc,
                      abase + (ch * STEPS + step + AD) * 1024);
          // F24 transform of chunk s + 1: two rounds of 8 channels
          if (!(W24_ABLATE & 1)) {
            if (xf && step == 2) xf_load(xsrc, 0, xd);
            if (xf && step == 8) xf_store(xdst, 0, xd);
            if (xf && step == 14) xf_load(xsrc, 1, xd);
            if (xf && step == 20) xf_store(xdst, 1, xd);
          }
          if (!(W24_ABLATE & 16)) {
            if (step == 0) { dma_begin(s + 3 < S); dma_vo = dma_offset(0); }
            if (step < ZL) dma_issue(step, dma_vo);
            if (step + 1 < ZL) dma_vo = dma_offset(step + 1);
            if (step == ZL - 1) dma_end();
          }
          if (step == ZL) side_look();
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        side_dma();
        side_look();
        if (xf) transform(xsrc, xdst);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (++rbuf == NRAW) rbuf = 0;
      // end of chunk: this wave's LDS traffic done, the DMA of the previous chunk landed (younger than its last
      // instruction: that chunk's STEPS - ZL ring loads and this chunk's STEPS + ZL instructions)
      if (W24_ABLATE & (4 | 16)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((STEPS - ZL) + STEPS + ZL) : "memory");
      if (!(W24_ABLATE & 8)) __builtin_amdgcn_s_barrier();
    }
    // the next item's first operands fly during the epilogue (every wave: one definition of the ring, see
    // conv3x3_winograd.hip "IN-FLIGHT RING REGISTERS AND THE COMPILER")
#pragma unroll
    for (int k = 0; k < AD; ++k) a_load(ar[k], nrsrc, nbase + k * 1024);
    if (active && !(W24_ABLATE & 32)) {
      const WLevel& L = args.lv[T.l];
      const int H = L.H, W = L.W, HW = H * W;
      const int flags = args.flags;
      const bool relu = flags & SSAD_CONV_RELU, sigm = flags & SSAD_CONV_SIGMOID;
      const bool masked = flags & SSAD_CONV_MASK_AUX;
      // this lane's tile: 2 rows x 4 columns at (py, px) of image sn
      const int sub = PAIRS ? (jn >> 3) : 0;
      const int sy0 = sub ? T.y0[1] : T.y0[0], sx0 = sub ? T.x0[1] : T.x0[0], sn = sub ? T.n[1] : T.n[0];
      const int py = PAIRS ? sy0 + 2 * ((jn & 7) >> 1) : T.y0[0] + 2 * (jn >> 2);
      const int px = PAIRS ? sx0 + 4 * (jn & 1) : T.x0[0] + 4 * (jn & 3);
      const __amdgpu_buffer_rsrc_t yrsrc = uniform_rsrc(L.y, (unsigned)((long long)L.N * M * HW * 4));
      const __amdgpu_buffer_rsrc_t krsrc = uniform_rsrc(masked ? L.aux : L.y, (unsigned)((long long)L.N * M * HW * 4));
      const bool whole = px + 3 < W && !sigm && mt * 16 + 16 <= M;      // four pixels inside: one 16-byte store per row
      // Everything the epilogue reads from memory is requested FIRST -- the four biases and, for the data gradient,
      // the eight 16-byte mask rows of this lane -- and consumed after the inverse transform: with the load beside
      // each store the wave waited out a memory round trip eight times per item.
      float bv[4];
      unsigned vos[4][2];
      f32x4 kvs[4][2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mt * 16 + kq * 4 + r;
        bv[r] = (L.bias && m < M) ? L.bias[m] : 0.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int yy = py + a;
          vos[r][a] = (whole && yy < H) ? (unsigned)((((long long)sn * M + m) * HW + yy * W + px) * 4) : kOOBOff;
          if (masked)
            kvs[r][a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(krsrc, vos[r][a], 0, 0));
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mt * 16 + kq * 4 + r;
        const float bias = bv[r];
        float t[2][XQ];
#pragma unroll
        for (int b = 0; b < XQ; ++b) {
          const float m0 = acc[b * 4 + 0][r], m1 = acc[b * 4 + 1][r], m2 = acc[b * 4 + 2][r], m3 = acc[b * 4 + 3][r];
          t[0][b] = m0 + m1 + m2;
          t[1][b] = m1 - m2 - m3;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const float e = t[a][1] - t[a][2], f = t[a][1] + t[a][2];
          const float g = t[a][3] - t[a][4], h = t[a][3] + t[a][4];
          f32x4 o;
          o[0] = t[a][0] + f + h + bias;
          o[1] = fmaf(2.0f, g, e) + bias;
          o[2] = fmaf(4.0f, h, f) + bias;
          o[3] = fmaf(8.0f, g, e) + t[a][5] + bias;
          if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.0f ? o[k] : 0.0f;
          }
          const int yy = py + a;
          if (whole) {
            const unsigned vo = vos[r][a];
            if (masked) {
              const f32x4 kv = kvs[r][a];
#pragma unroll
              for (int k = 0; k < 4; ++k) o[k] = kv[k] > 0.0f ? o[k] : 0.0f;
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o),
                                                   yrsrc, vo, 0, 0);
            // keep the data registers untouched while the store reads them (conv3x3_winograd.hip, split tail)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 3" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
          } else if (m < M && yy < H) {
            float* yout = L.y + ((long long)sn * M + m) * HW + yy * W;
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (px + k < W) {
                float v = o[k];
                if (sigm) v = 1.0f / (1.0f + expf(-v));
                if (masked) v = L.aux[((long long)sn * M + m) * HW + yy * W + px + k] > 0.0f ? v : 0.0f;
                yout[px + k] = v;
              }
          }
        }
      }
    }
    ring_landed(ar);
    T = Tn;
    arsrc = nrsrc;
    abase = nbase;
  }
}

// same rule as the F(2x2) kernel's launcher: sub-patch pairs where they save >= 4 % of the computed pixels
bool level_wants_pairs(int H, int W) {
  static const int geom = [] { const char* e = getenv("SSAD_WINO_PAIRS"); return (e && *e) ? atoi(e) : -1; }();
  if (geom >= 0) return geom != 0;
  const long long by_patch = (long long)cdiv(W, PC) * cdiv(H, PR) * 128;
  const long long by_sub = (long long)cdiv(W, SP) * cdiv(H, SP) * 64;
  return by_sub * 100 <= by_patch * 96;
}

}  // namespace

extern "C" {

size_t ssad_conv_wino24_filter_floats(int M, int K) {
  return (size_t)cdiv(M, 16) * cdiv(K, KC) * STEPS * 256 + 1024;
}

int ssad_conv_wino24_pack_filters(const ssad_pack_entry* entries_host, int n_entries, ssad_stream_t stream) {
  if (n_entries < 0 || (n_entries > 0 && !entries_host)) return SSAD_E_BADARG;
  for (int base = 0; base < n_entries; base += SSAD_MAX_PACK_ENTRIES) {
    const int cnt = n_entries - base < SSAD_MAX_PACK_ENTRIES ? n_entries - base : SSAD_MAX_PACK_ENTRIES;
    PackTable t;
    size_t nmax = 0;
    bool any_dgrad = false;
    for (int i = 0; i < cnt; ++i) {
      const ssad_pack_entry& e = entries_host[base + i];
      if (e.Cout <= 0 || e.Cin <= 0 || !e.w || (!e.packed_fwd && !e.packed_dgrad)) return SSAD_E_BADARG;
      t.e[i] = e;
      const size_t nf = ssad_conv_wino24_filter_floats(e.Cout, e.Cin), nd = ssad_conv_wino24_filter_floats(e.Cin, e.Cout);
      nmax = nf > nmax ? nf : nmax;
      if (e.packed_dgrad) { nmax = nd > nmax ? nd : nmax; any_dgrad = true; }
    }
    for (int i = cnt; i < SSAD_MAX_PACK_ENTRIES; ++i) t.e[i] = ssad_pack_entry{};
    size_t bx = (nmax + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(wino24_pack_multi_kernel, dim3((unsigned)bx, (unsigned)cnt, any_dgrad ? 2u : 1u), dim3(256), 0,
                       (hipStream_t)stream, t);
  }
  return (int)hipGetLastError();
}

int ssad_conv3x3_forward_wino24(const ssad_conv_level* lv, int n_levels, const float* packed, const float* bias,
                                int Cout, int Cin, int flags, ssad_stream_t stream) {
  if (n_levels < 1 || n_levels > SSAD_MAX_CONV_PROBLEMS || Cout <= 0 || Cin <= 0) return SSAD_E_BADARG;
  if ((flags & SSAD_CONV_MASK_AUX) && (flags & SSAD_CONV_SIGMOID)) return SSAD_E_BADARG;
  for (int l = 0; l < n_levels; ++l) {
    if (!(lv[l].packed ? lv[l].packed : packed)) return SSAD_E_BADARG;
    if ((flags & SSAD_CONV_MASK_AUX) && !lv[l].aux) return SSAD_E_BADARG;
    if (lv[l].N < 0 || lv[l].H < 0 || lv[l].W < 0) return SSAD_E_BADARG;
    if ((long long)lv[l].N * lv[l].H * lv[l].W * (Cin > Cout ? Cin : Cout) >= (1LL << 29)) return SSAD_E_BADARG;
  }
  for (int pass = 0; pass < 2; ++pass) {
    const bool use_pairs = pass == 1;
    WArgs a;
    a.M = Cout; a.K = Cin; a.chunks = cdiv(Cin, KC); a.flags = flags;
    a.mblocks = cdiv(Cout, BM);
    int nl = 0;
    long long blocks = 0, pairs = 0;
    for (int l = 0; l < n_levels; ++l) {
      if ((long long)lv[l].N * lv[l].H * lv[l].W == 0) continue;
      if (level_wants_pairs(lv[l].H, lv[l].W) != use_pairs) continue;
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
    const int cus = ssad_cu_count();
    a.patches = (int)(use_pairs ? pairs : blocks);
    const long long total = (long long)a.patches * a.mblocks;
    if (total >= (1LL << 31)) return SSAD_E_BADARG;
    a.items = (int)total;
    long long grid = total < cus ? total : cus;
    if (grid * ZNT < total) grid = (total + ZNT - 1) / ZNT;
    a.xcd_group = 8;
    if (use_pairs) hipLaunchKernelGGL((wino24_conv_kernel<true>), dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((wino24_conv_kernel<false>), dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

}  // extern "C"
