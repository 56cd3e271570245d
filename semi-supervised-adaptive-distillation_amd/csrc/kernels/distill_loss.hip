// distill_loss.hip -- SigmoidAdaptiveDistillLoss (+Gradient) and PowSum for
// gfx950 (MI355X).
//
// What the reference does (caffe2/modules/detectron/
// sigmoid_adaptive_distillation_loss_op.cu:108-141): an elementwise kernel
// writes a full-size `losses_` temp, a single 128-thread block sums it
// (caffe2/utils/math_gpu.cu:1023-1058) and a third launch scales one float.
// The gradient op (.cu:144-171) writes dX and then re-reads/re-writes all of
// it to multiply by `scale`.  PowSum (pow_sum_op.cu:26-43) repeats the
// temp + one-block-sum pattern once per FPN level.
//
// What this file does instead: every FPN level of a step is handled by ONE
// streaming launch.  HBM traffic is the algorithmic minimum -- logits +
// teacher probabilities read once with 16-byte loads, labels read once per
// (image, anchor, position) and reused across the 80 classes from cache,
// dX written once with /Np and *scale folded in.  The sum is a wavefront
// (64-lane) shuffle reduction in double, one partial per workgroup, followed
// by a fixed-order finalize launch: results are deterministic and closer to
// the exact sum than the reference's 128-lane fp32 order.
//
// Work decomposition: logits are N x (A*C) x H x W.  For a fixed (image n,
// anchor a) the C class planes of H*W floats are contiguous ("slab") and share
// ONE H*W label plane.  A work item is (slab, block of positions): each thread
// owns 4 consecutive positions (one 16-byte load per plane), reads their labels
// once into registers and walks the C class planes, so label bytes leave HBM
// once per (image, anchor, position) -- the algorithmic minimum -- and no
// per-element index arithmetic remains in the loop.  For small maps the 256
// threads split into position lanes x class lanes.

#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "ssad_kernels.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 8192;   // partial slots in the workspace

struct LevelArgs {
  const float* x;
  const float* q;
  const int32_t* g;
  float* out;
  int hw;            // H*W
  int slab;          // C*H*W floats per (n, a)
  int n_slabs;       // N*A
  int chunks;        // position blocks per slab
  int items;         // n_slabs * chunks
  int block_start;   // first blockIdx.x of this level
  int blocks;        // blocks assigned to this level
  int vec4;          // 1: HW % 4 == 0 and 16-byte aligned pointers
  int pl_shift;      // log2(position lanes); class lanes = 256 >> pl_shift
  int classes;       // C
  int cgroups;       // class groups per position block (work-item granularity)
  int cper;          // classes per group
};

struct LaunchArgs {
  LevelArgs lv[SSAD_MAX_LEVELS];
  int n_levels;
  float gamma, alpha, beta, scale;
  int ignored;
};

// ---- math helpers ---------------------------------------------------------

template <bool FAST> __device__ __forceinline__ float exp_f(float v) {
  if constexpr (FAST) return __expf(v); else return expf(v);
}
template <bool FAST> __device__ __forceinline__ float log_f(float v) {
  if constexpr (FAST) return __logf(v); else return logf(v);
}

// AT^gamma and AT^(gamma-1); gamma == 2 and gamma == 1 are multiplies.
template <int GAMMA_MODE>
__device__ __forceinline__ void pow_pair(float at, float gamma, float& pg, float& pgm1) {
  if constexpr (GAMMA_MODE == 2) { pg = at * at; pgm1 = at; }
  else if constexpr (GAMMA_MODE == 1) { pg = at; pgm1 = 1.0f; }
  else { pg = powf(at, gamma); pgm1 = powf(at, gamma - 1.0f); }
}

constexpr float kLogFltMin = -87.33654475f;  // logf(FLT_MIN)

// Shared front end of the forward and gradient formulas
// (.cu:56-61 and .cu:88-95).  Returns the pieces both need.
struct Pieces {
  float p;        // student probability sigma(x)
  float logp;     // log(max(FLT_MIN, p))
  float log1mp;   // -max(x,0) - log(1 + e^-|x|)
  float at;       // adaptive target 1 - exp(-DL)
  float edl;      // exp(-DL)
};

template <bool FAST, bool NEED_P, bool BETA0>
__device__ __forceinline__ Pieces front(float x, float q, float beta) {
  Pieces r;
  const float ax = fabsf(x);
  const float e = exp_f<FAST>(-ax);            // e^-|x|  in (0, 1]
  const float onepe = 1.0f + e;
  const float sp = log_f<FAST>(onepe);         // softplus(-|x|)
  const float xpos = fmaxf(x, 0.0f);
  // binary entropy of the teacher: NaN outside (0,1), incl. 0*log(0), and it
  // propagates even when beta == 0 (.cu:58-59).
  float ent;
  if constexpr (!BETA0) {
    ent = beta * (q * log_f<false>(q) + (1.0f - q) * log_f<false>(1.0f - q));
  } else {
    ent = (q > 0.0f && q < 1.0f) ? 0.0f : __builtin_nanf("");   // a select, no branch
  }
  // DL = -x*(q - [x>=0]) + softplus(-|x|) + ent
  const float dl = (xpos - x * q) + sp + ent;
  r.edl = exp_f<FAST>(-dl);
  r.at = 1.0f - r.edl;
  r.log1mp = -xpos - sp;
  // log p = min(x,0) - softplus(-|x|), clamped where p underflows FLT_MIN
  r.logp = fmaxf(fminf(x, 0.0f) - sp, kLogFltMin);
  if constexpr (NEED_P) {
    const float inv = __frcp_rn(onepe);
    r.p = (x >= 0.0f) ? inv : e * inv;
  } else {
    r.p = 0.0f;
  }
  return r;
}

template <bool FAST, int GAMMA_MODE, bool BETA0>
__device__ __forceinline__ float loss_elem(
    float x, float q, bool keep, float gamma, float beta, float w_pos, float w_neg) {
  const Pieces f = front<FAST, false, BETA0>(x, q, beta);
  float pg, pgm1;
  pow_pair<GAMMA_MODE>(f.at, gamma, pg, pgm1);
  const float ce = q * f.logp * w_pos + (1.0f - q) * f.log1mp * w_neg;
  const float v = -pg * ce;
  return v * (keep ? 1.0f : 0.0f);   // multiply: NaN survives an ignored label
}

template <bool FAST, int GAMMA_MODE, bool BETA0>
__device__ __forceinline__ float grad_elem(
    float x, float q, bool keep, float gamma, float alpha, float beta, float mult) {
  const Pieces f = front<FAST, true, BETA0>(x, q, beta);
  float pg, pgm1;
  pow_pair<GAMMA_MODE>(f.at, gamma, pg, pgm1);
  const float S = alpha * q * f.logp + (1.0f - alpha) * (1.0f - q) * f.log1mp;
  const float diff = q - f.p;
  const float t1 = -diff * gamma * pgm1 * f.edl * S;
  const float t2 = pg * (alpha * diff - (1.0f - 2.0f * alpha) * (1.0f - q) * f.p);
  const float g = -(t1 + t2) * mult;   // mult = dloss * scale / Np
  return g * (keep ? 1.0f : 0.0f);
}

// ---- reductions -----------------------------------------------------------

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Sum over the 256-thread workgroup; result valid in thread 0.
__device__ __forceinline__ double block_sum(double v) {
  __shared__ double wsum[kThreads / 64];
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) wsum[wid] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < kThreads / 64; ++i) t += wsum[i];
  }
  return t;
}

__device__ __forceinline__ int find_level(const LaunchArgs& a, int bid) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < SSAD_MAX_LEVELS; ++i)
    if (i < a.n_levels && bid >= a.lv[i].block_start) l = i;
  return l;
}

// ---- forward ---------------------------------------------------------------

// Streamed-once data: non-temporal 16-byte accesses (env SSAD_LOSS_NT=0 turns
// them into plain accesses for A/B runs).
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef SSAD_LOSS_NT
#define SSAD_LOSS_NT 1
#endif
__device__ __forceinline__ float4 ld4(const float* p) {
#if SSAD_LOSS_NT
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void st4(float* p, const float4& r) {
#if SSAD_LOSS_NT
  f32x4 v; v.x = r.x; v.y = r.y; v.z = r.z; v.w = r.w;
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
  *reinterpret_cast<float4*>(p) = r;
#endif
}

// Shared traversal: calls f(x, q, t, d) for every element of the work items of
// this workgroup (x logit, q teacher probability or 0 when !HAS_Q, t label of
// the element's (image, anchor, position), d class index).  MODE 0: sum f's
// result; MODE 1: store it to L.out; MODE 2: f returns {value to store, two
// values to sum} (fused losses + gradient).
struct Acc2 { float a, b; };
struct Fused { float store, a, b; };

template <int MODE, bool HAS_Q, class F>
__device__ __forceinline__ Acc2 traverse(const LevelArgs& L, int lb, F f) {
  Acc2 acc{0.0f, 0.0f};
  const int pl = 1 << L.pl_shift;
  const int cl = kThreads >> L.pl_shift;
  const int pi = threadIdx.x & (pl - 1);
  const int ci = threadIdx.x >> L.pl_shift;
  const int hw = L.hw;
  auto fold = [&](float& dst, float x, float q, int t, int d) {
    if constexpr (MODE == 2) {
      const Fused r = f(x, q, t, d);
      dst = r.store; acc.a += r.a; acc.b += r.b;
    } else {
      const float r = f(x, q, t, d);
      if constexpr (MODE == 1) dst = r; else acc.a += r;
    }
  };
  for (int item = lb; item < L.items; item += L.blocks) {
    const int sc = item / L.cgroups;
    const int cg = item - sc * L.cgroups;
    const int slab = sc / L.chunks;
    const int chunk = sc - slab * L.chunks;
    const int c_begin = cg * L.cper + ci;
    const int c_end = (cg + 1) * L.cper < L.classes ? (cg + 1) * L.cper : L.classes;
    const float* __restrict__ xs = L.x + (size_t)slab * L.slab;
    const float* __restrict__ qs = HAS_Q ? L.q + (size_t)slab * L.slab : nullptr;
    float* __restrict__ ds = MODE != 0 ? L.out + (size_t)slab * L.slab : nullptr;
    const int32_t* __restrict__ gs = L.g + (size_t)slab * hw;
    if (L.vec4) {
      const int pos = (chunk * pl + pi) * 4;
      if (pos < hw) {
        const int4 gv = *reinterpret_cast<const int4*>(gs + pos);
        auto one = [&](int c, const float4& xv, const float4& qv) {
          float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
          fold(r.x, xv.x, qv.x, gv.x, c); fold(r.y, xv.y, qv.y, gv.y, c);
          fold(r.z, xv.z, qv.z, gv.z, c); fold(r.w, xv.w, qv.w, gv.w, c);
          if constexpr (MODE != 0) st4(ds + c * hw + pos, r);
        };
        int c = c_begin;
        // four class planes per step: all 16-byte loads are issued before the
        // first use, so each wave keeps 8 KiB in flight
        for (; c + 3 * cl < c_end; c += 4 * cl) {
          float4 xv[4], qv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int o = (c + u * cl) * hw + pos;
            xv[u] = ld4(xs + o);
            qv[u] = HAS_Q ? ld4(qs + o) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) one(c + u * cl, xv[u], qv[u]);
        }
        for (; c < c_end; c += cl) {
          const int o = c * hw + pos;
          one(c, ld4(xs + o), HAS_Q ? ld4(qs + o) : make_float4(0.f, 0.f, 0.f, 0.f));
        }
      }
    } else {
      const int pos = chunk * pl + pi;
      if (pos < hw) {
        const int t = gs[pos];
#pragma unroll 4
        for (int c = c_begin; c < c_end; c += cl) {
          const int o = c * hw + pos;
          float r = 0.0f;
          fold(r, xs[o], HAS_Q ? qs[o] : 0.0f, t, c);
          if constexpr (MODE != 0) ds[o] = r;
        }
      }
    }
  }
  return acc;
}

// ---- SigmoidFocalLoss element formulas (sigmoid_focal_loss_op.cu:33-66, 74-105)

struct FocalPieces { float p, omp, logp, log1mp; };

template <bool FAST>
__device__ __forceinline__ FocalPieces focal_front(float x) {
  FocalPieces r;
  const float e = exp_f<FAST>(-fabsf(x));
  const float onepe = 1.0f + e;
  const float sp = log_f<FAST>(onepe);
  const float inv = __frcp_rn(onepe);
  r.p = (x >= 0.0f) ? inv : e * inv;
  r.omp = 1.0f - r.p;                      // as the reference: 1 - (rounded p)
  r.logp = fmaxf(fminf(x, 0.0f) - sp, kLogFltMin);
  r.log1mp = -fmaxf(x, 0.0f) - sp;
  return r;
}

template <bool FAST, int GAMMA_MODE>
__device__ __forceinline__ float focal_loss_elem(float x, int t, int d, float gamma, float zp, float zn) {
  const FocalPieces f = focal_front<FAST>(x);
  float a, b, unused;
  pow_pair<GAMMA_MODE>(f.omp, gamma, a, unused);
  pow_pair<GAMMA_MODE>(f.p, gamma, b, unused);
  const float c1 = (t == d + 1) ? 1.0f : 0.0f;
  const float c2 = (t != -1 && t != d + 1) ? 1.0f : 0.0f;
  return -c1 * (a * f.logp) * zp - c2 * (b * f.log1mp) * zn;
}

template <bool FAST, int GAMMA_MODE>
__device__ __forceinline__ float focal_grad_elem(float x, int t, int d, float gamma, float zp,
                                                 float zn, float mult) {
  const FocalPieces f = focal_front<FAST>(x);
  float a, b, unused;
  pow_pair<GAMMA_MODE>(f.omp, gamma, a, unused);
  pow_pair<GAMMA_MODE>(f.p, gamma, b, unused);
  const float term1 = a * (f.omp - f.p * gamma * f.logp);
  const float term2 = b * (f.log1mp * f.omp * gamma - f.p);
  const float c1 = (t == d + 1) ? 1.0f : 0.0f;
  const float c2 = (t != -1 && t != d + 1) ? 1.0f : 0.0f;
  return (-c1 * zp * term1 - c2 * zn * term2) * mult;      // mult = dloss * scale
}

template <bool FAST, int GAMMA_MODE, bool BETA0>
__global__ __launch_bounds__(kThreads) void distill_fwd_kernel(
    const LaunchArgs args, const float* __restrict__ normalizer,
    double* __restrict__ partials) {
  const LevelArgs& L = args.lv[find_level(args, blockIdx.x)];
  const int lb = blockIdx.x - L.block_start;
  const float np = fmaxf(normalizer[0], 1.0f);
  const float w_pos = args.alpha / np;
  const float w_neg = (1.0f - args.alpha) / np;
  const float gamma = args.gamma, beta = args.beta;
  const int ignored = args.ignored;
  const Acc2 acc = traverse<0, true>(L, lb, [&](float x, float q, int t, int) {
    return loss_elem<FAST, GAMMA_MODE, BETA0>(x, q, t != ignored, gamma, beta, w_pos, w_neg);
  });
  const double t = block_sum((double)acc.a);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// One workgroup per level: fixed-order sum of that level's partials, then
// the float multiply by scale (math::Scale on one element, .cu:137-138).
__global__ __launch_bounds__(kThreads) void distill_finalize_kernel(
    const LaunchArgs args, const double* __restrict__ partials) {
  const LevelArgs& L = args.lv[blockIdx.x];
  double v = 0.0;
  for (int i = threadIdx.x; i < L.blocks; i += kThreads) v += partials[L.block_start + i];
  const double t = block_sum(v);
  if (threadIdx.x == 0) L.out[0] = (float)t * args.scale;
}

// ---- backward --------------------------------------------------------------

template <bool FAST, int GAMMA_MODE, bool BETA0>
__global__ __launch_bounds__(kThreads) void distill_bwd_kernel(
    const LaunchArgs args, const float* __restrict__ normalizer,
    const float* __restrict__ dloss, int dloss_stride) {
  const int level = find_level(args, blockIdx.x);
  const LevelArgs& L = args.lv[level];
  const int lb = blockIdx.x - L.block_start;
  const float np = fmaxf(normalizer[0], 1.0f);
  const float mult = dloss[(size_t)level * dloss_stride] * args.scale / np;
  const float gamma = args.gamma, alpha = args.alpha, beta = args.beta;
  const int ignored = args.ignored;
  traverse<1, true>(L, lb, [&](float x, float q, int t, int) {
    return grad_elem<FAST, GAMMA_MODE, BETA0>(x, q, t != ignored, gamma, alpha, beta, mult);
  });
}

// ---- SigmoidFocalLoss kernels --------------------------------------------------

struct FocalScalars { float gamma, alpha, scale; };

template <bool FAST, int GAMMA_MODE>
__global__ __launch_bounds__(kThreads) void focal_fwd_kernel(
    const LaunchArgs args, const FocalScalars fs, const float* __restrict__ fg_num,
    double* __restrict__ partials) {
  const LevelArgs& L = args.lv[find_level(args, blockIdx.x)];
  const int lb = blockIdx.x - L.block_start;
  const float np = fmaxf(fg_num[0], 1.0f);
  const float zp = fs.alpha / np, zn = (1.0f - fs.alpha) / np, gamma = fs.gamma;
  const Acc2 acc = traverse<0, false>(L, lb, [&](float x, float, int t, int d) {
    return focal_loss_elem<FAST, GAMMA_MODE>(x, t, d, gamma, zp, zn);
  });
  const double t = block_sum((double)acc.a);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

template <bool FAST, int GAMMA_MODE>
__global__ __launch_bounds__(kThreads) void focal_bwd_kernel(
    const LaunchArgs args, const FocalScalars fs, const float* __restrict__ fg_num,
    const float* __restrict__ dloss, int dloss_stride) {
  const int level = find_level(args, blockIdx.x);
  const LevelArgs& L = args.lv[level];
  const int lb = blockIdx.x - L.block_start;
  const float np = fmaxf(fg_num[0], 1.0f);
  const float zp = fs.alpha / np, zn = (1.0f - fs.alpha) / np, gamma = fs.gamma;
  const float mult = dloss[(size_t)level * dloss_stride] * fs.scale;
  traverse<1, false>(L, lb, [&](float x, float, int t, int d) {
    return focal_grad_elem<FAST, GAMMA_MODE>(x, t, d, gamma, zp, zn, mult);
  });
}

// Both classification losses of the student in ONE pass over the logits:
// distillation loss sum, focal loss sum and dX = d(distill)/dx + d(focal)/dx
// (the reference runs two forward kernels, two gradient kernels, two Scale
// passes and an autograd Sum over the same N x 720 x H x W tensor).
template <bool FAST, int GAMMA_MODE, bool BETA0>
__global__ __launch_bounds__(kThreads) void cls_losses_fused_kernel(
    const LaunchArgs args, const FocalScalars fs, const float* __restrict__ normalizer,
    const float* __restrict__ fg_num, double* __restrict__ partials, int focal_offset) {
  const LevelArgs& L = args.lv[find_level(args, blockIdx.x)];
  const int lb = blockIdx.x - L.block_start;
  const float np = fmaxf(normalizer[0], 1.0f);
  const float w_pos = args.alpha / np, w_neg = (1.0f - args.alpha) / np;
  const float d_mult = args.scale / np;                  // dloss = 1 (utils/blob.py:166-172)
  const float nf = fmaxf(fg_num[0], 1.0f);
  const float zp = fs.alpha / nf, zn = (1.0f - fs.alpha) / nf;
  const float gamma = args.gamma, alpha = args.alpha, beta = args.beta, fgamma = fs.gamma;
  const float f_mult = fs.scale;
  const int ignored = args.ignored;
  const Acc2 acc = traverse<2, true>(L, lb, [&](float x, float q, int t, int d) {
    const bool keep = t != ignored;
    Fused r;
    r.a = loss_elem<FAST, GAMMA_MODE, BETA0>(x, q, keep, gamma, beta, w_pos, w_neg);
    r.b = focal_loss_elem<FAST, 2>(x, t, d, fgamma, zp, zn);
    r.store = grad_elem<FAST, GAMMA_MODE, BETA0>(x, q, keep, gamma, alpha, beta, d_mult) +
              focal_grad_elem<FAST, 2>(x, t, d, fgamma, zp, zn, f_mult);
    return r;
  });
  const double ta = block_sum((double)acc.a);
  __syncthreads();
  const double tb = block_sum((double)acc.b);
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = ta;
    partials[focal_offset + blockIdx.x] = tb;
  }
}

// finalize for the fused kernel: level l -> out_a[l], out_b[l]
__global__ __launch_bounds__(kThreads) void fused_finalize_kernel(
    const LaunchArgs args, const double* __restrict__ partials, int focal_offset,
    float* __restrict__ out_a, float* __restrict__ out_b, float scale_a, float scale_b) {
  const LevelArgs& L = args.lv[blockIdx.x];
  double va = 0.0, vb = 0.0;
  for (int i = threadIdx.x; i < L.blocks; i += kThreads) {
    va += partials[L.block_start + i];
    vb += partials[focal_offset + L.block_start + i];
  }
  const double ta = block_sum(va);
  __syncthreads();
  const double tb = block_sum(vb);
  if (threadIdx.x == 0) {
    out_a[blockIdx.x] = (float)ta * scale_a;
    out_b[blockIdx.x] = (float)tb * scale_b;
  }
}

// ---- SelectSmoothL1Loss (select_smooth_l1_loss_op.cu:23-86) ---------------------
// M foreground boxes x 4 coordinates, gathered through the location list: two
// dependent global round trips per element, so the pass is latency-bound and
// wants many workgroups (64K boxes on the finest level of a 16-image batch took
// 480 us in one workgroup).  All FPN levels run in ONE launch (blockIdx.y = level);
// per-workgroup double partials go to a caller-provided workspace (nothing is
// allocated here, so the step can be captured in a graph) and are summed in index
// order by one workgroup per level: the result does not depend on timing.
constexpr int kSl1Blocks = 512;     // partial slots per level

struct Sl1Args {
  ssad_smooth_l1_level lv[SSAD_MAX_LEVELS];
  int n_levels;
};

// element e = 4*i + j of a level's list -> index into Y_hat, or -1 when the entry is to be
// skipped.  The reference's labelling can list anchors of the full anchor field that lie
// outside the (cropped) prediction map (roi_data/retinanet.py:278-293) and would read out of
// bounds there; any entry outside [0,N) x [0,D) x [0,H) x [0,W) contributes nothing here.
__device__ __forceinline__ long long sl1_index(const ssad_smooth_l1_level& L, int e) {
  const int i = e >> 2, j = e & 3;
  const int n = (int)L.L[i * 4], c = (int)L.L[i * 4 + 1], y = (int)L.L[i * 4 + 2], x = (int)L.L[i * 4 + 3];
  if (n < 0 || n >= L.N || c < 0 || c + j >= L.D || y < 0 || y >= L.H || x < 0 || x >= L.W) return -1;
  return ((long long)n * L.D + c + j) * L.H * L.W + (long long)y * L.W + x;
}

__global__ __launch_bounds__(kThreads) void smooth_l1_fwd_kernel(
    const Sl1Args args, const float* __restrict__ S, float beta, double* __restrict__ partials) {
  const ssad_smooth_l1_level& L = args.lv[blockIdx.y];
  const double s = (double)fmaxf(S[0], 1.0f);
  double acc = 0.0;
  for (int e = blockIdx.x * kThreads + threadIdx.x; e < L.M * 4; e += gridDim.x * kThreads) {
    const long long ind = sl1_index(L, e);
    if (ind < 0) continue;
    const float val = L.Y_hat[ind] - L.Y[e];
    const float a = fabsf(val);
    acc += (double)(a < beta ? (float)((0.5 * (double)val * (double)val / (double)beta) / s)
                             : (float)(((double)a - 0.5 * (double)beta) / s));
  }
  const double t = block_sum(acc);
  if (threadIdx.x == 0) partials[blockIdx.y * kSl1Blocks + blockIdx.x] = t;
}

__global__ __launch_bounds__(kThreads) void smooth_l1_finalize_kernel(
    const Sl1Args args, const double* __restrict__ partials, int n, float scale) {
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += kThreads) v += partials[blockIdx.x * kSl1Blocks + i];
  const double t = block_sum(v);
  if (threadIdx.x == 0) args.lv[blockIdx.x].loss[0] = (float)t * scale;
}

// zero fill of every level's dY_hat (the reference's math::Set, .cu:143-145) in one launch
__global__ __launch_bounds__(kThreads) void smooth_l1_zero_kernel(const Sl1Args args) {
  const ssad_smooth_l1_level& L = args.lv[blockIdx.y];
  const long long n = (long long)L.N * L.D * L.H * L.W;
  float* __restrict__ o = L.dY_hat;
  const long long n4 = (((uintptr_t)o & 15) == 0) ? (n >> 2) : 0;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (long long)gridDim.x * kThreads)
    reinterpret_cast<float4*>(o)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long i = n4 * 4 + (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads)
    o[i] = 0.0f;
}

// scatters the M*4 non-zero entries into the zero-filled dY_hat
__global__ __launch_bounds__(kThreads) void smooth_l1_bwd_kernel(
    const Sl1Args args, const float* __restrict__ S, const float* __restrict__ dloss, float beta,
    float scale) {
  const ssad_smooth_l1_level& L = args.lv[blockIdx.y];
  const float s = fmaxf(S[0], 1.0f);
  const float nd = scale * dloss[0];
  for (int e = blockIdx.x * kThreads + threadIdx.x; e < L.M * 4; e += gridDim.x * kThreads) {
    const long long ind = sl1_index(L, e);
    if (ind < 0) continue;
    const float val = L.Y_hat[ind] - L.Y[e];
    const float a = fabsf(val);
    const float sign = (float)((0.0f < val) - (val < 0.0f));
    L.dY_hat[ind] = a < beta ? nd * val / beta / s : nd * sign / s;
  }
}

// ---- PowSum ----------------------------------------------------------------

struct PowArgs {
  const float* ptr[SSAD_MAX_POWSUM_INPUTS];
  long long n[SSAD_MAX_POWSUM_INPUTS];
  int block_start[SSAD_MAX_POWSUM_INPUTS];
  int blocks[SSAD_MAX_POWSUM_INPUTS];
  int n_inputs;
  float power;
};

// x^p.  Positive normal x (teacher probabilities) take the exp2/log2 path;
// everything else goes through powf for the exact special-case behaviour.
template <bool FAST>
__device__ __forceinline__ float pow_elem(float x, float p) {
  if constexpr (FAST) {
    if (x >= FLT_MIN && x < 3.0e38f) return __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(x));
  }
  return powf(x, p);
}

template <bool FAST>
__global__ __launch_bounds__(kThreads) void pow_sum_kernel(
    const PowArgs args, double* __restrict__ partials) {
  int j = 0;
#pragma unroll
  for (int i = 1; i < SSAD_MAX_POWSUM_INPUTS; ++i)
    if (i < args.n_inputs && (int)blockIdx.x >= args.block_start[i]) j = i;
  const float* x = args.ptr[j];
  const long long n = args.n[j];
  const int lb = blockIdx.x - args.block_start[j];
  const int nb = args.blocks[j];
  const float p = args.power;
  float acc = 0.0f;
  double dacc = 0.0;
  const bool aligned = ((uintptr_t)x & 15) == 0;
  const long long n4 = aligned ? (n >> 2) : 0;
  int folds = 0;
  const long long step = (long long)nb * kThreads;
  long long i4 = (long long)lb * kThreads + threadIdx.x;
  auto add4 = [&](const float4& v) {
    acc += (pow_elem<FAST>(v.x, p) + pow_elem<FAST>(v.y, p)) +
           (pow_elem<FAST>(v.z, p) + pow_elem<FAST>(v.w, p));
  };
  for (; i4 + 3 * step < n4; i4 += 4 * step) {      // four 16-byte loads in flight
    const float4 v0 = ld4(x + 4 * i4), v1 = ld4(x + 4 * (i4 + step)),
                 v2 = ld4(x + 4 * (i4 + 2 * step)), v3 = ld4(x + 4 * (i4 + 3 * step));
    add4(v0); add4(v1); add4(v2); add4(v3);
    if (++folds == 16) { dacc += (double)acc; acc = 0.0f; folds = 0; }
  }
  for (; i4 < n4; i4 += step) add4(ld4(x + 4 * i4));
  for (long long i = n4 * 4 + (long long)lb * kThreads + threadIdx.x; i < n; i += (long long)nb * kThreads)
    acc += pow_elem<FAST>(x[i], p);
  dacc += (double)acc;
  const double t = block_sum(dacc);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

__global__ __launch_bounds__(kThreads) void pow_sum_finalize_kernel(
    const double* __restrict__ partials, int n, float* __restrict__ out, int accumulate) {
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += kThreads) v += partials[i];
  const double t = block_sum(v);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.0f) + (float)t;
}

// ---- host side ---------------------------------------------------------------

int gamma_mode(float g) { return g == 2.0f ? 2 : (g == 1.0f ? 1 : 0); }

bool accurate_math() {
  static const bool v = [] {
    const char* e = getenv("SSAD_ACCURATE_MATH");
    return e && e[0] == '1';
  }();
  return v;
}

int build_args(const ssad_distill_level* lv, int n_levels,
               const ssad_distill_params* P, LaunchArgs* out, int* total_blocks,
               bool out_is_tensor = false) {
  if (n_levels < 1 || n_levels > SSAD_MAX_LEVELS || !P) return SSAD_E_BADARG;
  if (P->num_classes <= 0 || !(P->scale >= 0.0f)) return SSAD_E_BADARG;
  LaunchArgs& a = *out;
  a.n_levels = n_levels;
  a.gamma = P->gamma; a.alpha = P->alpha; a.beta = P->beta; a.scale = P->scale;
  a.ignored = P->ignored_label;
  long long total_items = 0;
  for (int l = 0; l < n_levels; ++l) {
    const ssad_distill_level& s = lv[l];
    if (s.N < 0 || s.D < 0 || s.H < 0 || s.W < 0) return SSAD_E_BADARG;
    if (s.D % P->num_classes != 0) return SSAD_E_BADARG;
    const long long hw = (long long)s.H * s.W;
    const long long slab = hw * P->num_classes;
    const long long n_slabs = (long long)s.N * (s.D / P->num_classes);
    if (slab >= (1LL << 31) || n_slabs >= (1LL << 31) || hw * 4 >= (1LL << 31)) return SSAD_E_BADARG;
    LevelArgs& L = a.lv[l];
    L.x = s.logits; L.q = s.teacher_prob; L.g = s.labels; L.out = s.out;
    L.hw = (int)hw; L.slab = (int)slab; L.n_slabs = (int)n_slabs;
    // `out` is a full-size tensor (16-byte stores) only for the gradient
    const uintptr_t al = (uintptr_t)s.logits | (uintptr_t)s.teacher_prob |
                         (uintptr_t)s.labels | (out_is_tensor ? (uintptr_t)s.out : 0);
    L.vec4 = (hw % 4 == 0) && ((al & 15) == 0);
    L.classes = P->num_classes;
    // position lanes: smallest power of two covering the plane, at most 256
    const long long units = L.vec4 ? hw / 4 : hw;      // 16-byte (or scalar) columns
    int shift = 0;
    while ((1LL << shift) < units && shift < 8) ++shift;
    L.pl_shift = shift;
    const long long chunks = units > 0 ? (units + (1LL << shift) - 1) >> shift : 0;
    // class groups: >= 4 iterations per thread, about C/4 classes per item
    const int cl = kThreads >> shift;
    static const int cg_want = [] { const char* e = getenv("SSAD_LOSS_CGROUPS"); return e ? atoi(e) : 4; }();
    int cper = (P->num_classes + cg_want - 1) / cg_want;
    if (cper < 4 * cl) cper = 4 * cl;
    cper = (cper + cl - 1) / cl * cl;
    L.cper = cper;
    L.cgroups = (P->num_classes + cper - 1) / cper;
    const long long items = chunks * n_slabs * L.cgroups;
    if (items >= (1LL << 31)) return SSAD_E_BADARG;
    L.chunks = (int)(chunks > 0 ? chunks : 1);
    L.items = (int)items;
    total_items += items;
  }
  // distribute at most kMaxBlocks blocks proportionally to the work
  // tuning override, clamped: the workspace holds kMaxBlocks partial slots (the fused kernel's
  // focal partials start at slot kMaxBlocks) and every level owns at least one block
  static const int max_blocks_env = [] { const char* e = getenv("SSAD_LOSS_MAXBLOCKS"); return e ? atoi(e) : kMaxBlocks; }();
  const int lo = 2 * n_levels + 1;
  const int max_blocks = max_blocks_env > kMaxBlocks ? kMaxBlocks : (max_blocks_env < lo ? lo : max_blocks_env);
  int start = 0;
  for (int l = 0; l < n_levels; ++l) {
    LevelArgs& L = a.lv[l];
    long long b = L.items;
    if (total_items > max_blocks - n_levels) {
      b = (long long)L.items * (max_blocks - n_levels) / total_items;
    }
    if (b < 1) b = 1;   // every level owns >= 1 block so its output is written
    if (b > L.items && L.items > 0) b = L.items;
    L.block_start = start;
    L.blocks = (int)b;
    start += (int)b;
  }
  *total_blocks = start;
  return 0;
}

#define LAUNCH_BY_MODE3(KERNEL, FAST_, b0, gm, ...)                                   \
  do {                                                                                \
    if (b0) {                                                                         \
      if (gm == 2) hipLaunchKernelGGL((KERNEL<FAST_, 2, true>), __VA_ARGS__);         \
      else if (gm == 1) hipLaunchKernelGGL((KERNEL<FAST_, 1, true>), __VA_ARGS__);    \
      else hipLaunchKernelGGL((KERNEL<FAST_, 0, true>), __VA_ARGS__);                 \
    } else {                                                                          \
      if (gm == 2) hipLaunchKernelGGL((KERNEL<FAST_, 2, false>), __VA_ARGS__);        \
      else if (gm == 1) hipLaunchKernelGGL((KERNEL<FAST_, 1, false>), __VA_ARGS__);   \
      else hipLaunchKernelGGL((KERNEL<FAST_, 0, false>), __VA_ARGS__);                \
    }                                                                                 \
  } while (0)

#define LAUNCH_BY_MODE(KERNEL, fast, b0, gm, ...)                                     \
  do {                                                                                \
    if (fast) LAUNCH_BY_MODE3(KERNEL, true, b0, gm, __VA_ARGS__);                     \
    else LAUNCH_BY_MODE3(KERNEL, false, b0, gm, __VA_ARGS__);                         \
  } while (0)

}  // namespace

extern "C" {

size_t ssad_distill_loss_workspace_bytes(int n_levels) {
  (void)n_levels;
  return sizeof(double) * kMaxBlocks;
}

int ssad_distill_loss_forward(
    const ssad_distill_level* levels_host, int n_levels, const float* normalizer,
    const ssad_distill_params* params_host, void* workspace,
    size_t workspace_bytes, ssad_stream_t stream) {
  LaunchArgs a;
  int blocks = 0;
  const int rc = build_args(levels_host, n_levels, params_host, &a, &blocks);
  if (rc) return rc;
  if (!workspace || workspace_bytes < sizeof(double) * (size_t)blocks) return SSAD_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  double* partials = (double*)workspace;
  const bool fast = !accurate_math();
  const int gm = gamma_mode(a.gamma);
  LAUNCH_BY_MODE(distill_fwd_kernel, fast, a.beta == 0.0f, gm, dim3(blocks), dim3(kThreads), 0, s,
                 a, normalizer, partials);
  hipLaunchKernelGGL(distill_finalize_kernel, dim3(n_levels), dim3(kThreads), 0, s,
                     a, (const double*)partials);
  return (int)hipGetLastError();
}

int ssad_distill_loss_backward(
    const ssad_distill_level* levels_host, int n_levels, const float* normalizer,
    const float* dloss, int dloss_stride, const ssad_distill_params* params_host,
    ssad_stream_t stream) {
  LaunchArgs a;
  int blocks = 0;
  const int rc = build_args(levels_host, n_levels, params_host, &a, &blocks, true);
  if (rc) return rc;
  long long total = 0;
  for (int l = 0; l < n_levels; ++l) total += a.lv[l].items;
  if (total == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const bool fast = !accurate_math();
  const int gm = gamma_mode(a.gamma);
  LAUNCH_BY_MODE(distill_bwd_kernel, fast, a.beta == 0.0f, gm, dim3(blocks), dim3(kThreads), 0, s,
                 a, normalizer, dloss, dloss_stride);
  return (int)hipGetLastError();
}

static int focal_levels(const ssad_distill_level* lv, int n_levels, const ssad_focal_params* F,
                        LaunchArgs* a, int* blocks, bool out_is_tensor) {
  if (!F) return SSAD_E_BADARG;
  ssad_distill_params P{F->gamma, F->alpha, 0.0f, F->num_classes, -1, F->scale};
  return build_args(lv, n_levels, &P, a, blocks, out_is_tensor);
}

int ssad_focal_loss_forward(const ssad_distill_level* levels_host, int n_levels,
                            const float* fg_num, const ssad_focal_params* params_host,
                            void* workspace, size_t workspace_bytes, ssad_stream_t stream) {
  LaunchArgs a;
  int blocks = 0;
  const int rc = focal_levels(levels_host, n_levels, params_host, &a, &blocks, false);
  if (rc) return rc;
  if (!workspace || workspace_bytes < sizeof(double) * (size_t)blocks) return SSAD_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const FocalScalars fs{params_host->gamma, params_host->alpha, params_host->scale};
  const int gm = gamma_mode(fs.gamma);
  double* partials = (double*)workspace;
#define FOCAL_LAUNCH(K, ...)                                                              \
  do {                                                                                    \
    if (!accurate_math()) {                                                               \
      if (gm == 2) hipLaunchKernelGGL((K<true, 2>), __VA_ARGS__);                         \
      else if (gm == 1) hipLaunchKernelGGL((K<true, 1>), __VA_ARGS__);                    \
      else hipLaunchKernelGGL((K<true, 0>), __VA_ARGS__);                                 \
    } else {                                                                              \
      if (gm == 2) hipLaunchKernelGGL((K<false, 2>), __VA_ARGS__);                        \
      else if (gm == 1) hipLaunchKernelGGL((K<false, 1>), __VA_ARGS__);                   \
      else hipLaunchKernelGGL((K<false, 0>), __VA_ARGS__);                                \
    }                                                                                     \
  } while (0)
  FOCAL_LAUNCH(focal_fwd_kernel, dim3(blocks), dim3(kThreads), 0, s, a, fs, fg_num, partials);
  hipLaunchKernelGGL(distill_finalize_kernel, dim3(n_levels), dim3(kThreads), 0, s, a,
                     (const double*)partials);
  return (int)hipGetLastError();
}

int ssad_focal_loss_backward(const ssad_distill_level* levels_host, int n_levels,
                             const float* fg_num, const float* dloss, int dloss_stride,
                             const ssad_focal_params* params_host, ssad_stream_t stream) {
  LaunchArgs a;
  int blocks = 0;
  const int rc = focal_levels(levels_host, n_levels, params_host, &a, &blocks, true);
  if (rc) return rc;
  long long total = 0;
  for (int l = 0; l < n_levels; ++l) total += a.lv[l].items;
  if (total == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const FocalScalars fs{params_host->gamma, params_host->alpha, params_host->scale};
  const int gm = gamma_mode(fs.gamma);
  FOCAL_LAUNCH(focal_bwd_kernel, dim3(blocks), dim3(kThreads), 0, s, a, fs, fg_num, dloss,
               dloss_stride);
  return (int)hipGetLastError();
}

size_t ssad_cls_losses_fused_workspace_bytes(int n_levels) {
  (void)n_levels;
  return 2 * sizeof(double) * kMaxBlocks;
}

int ssad_cls_losses_fused(const ssad_distill_level* levels_host, int n_levels,
                          const float* normalizer, const float* fg_num,
                          const ssad_distill_params* distill_host,
                          const ssad_focal_params* focal_host, float* distill_losses,
                          float* focal_losses, void* workspace, size_t workspace_bytes,
                          ssad_stream_t stream) {
  if (!focal_host || !distill_host || focal_host->gamma != 2.0f ||
      focal_host->num_classes != distill_host->num_classes)
    return SSAD_E_BADARG;     // the fused kernel specialises the focal gamma = 2 of RetinaNet
  LaunchArgs a;
  int blocks = 0;
  const int rc = build_args(levels_host, n_levels, distill_host, &a, &blocks, true);
  if (rc) return rc;
  if (!workspace || workspace_bytes < 2 * sizeof(double) * kMaxBlocks) return SSAD_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  double* partials = (double*)workspace;
  const FocalScalars fs{focal_host->gamma, focal_host->alpha, focal_host->scale};
  const bool fast = !accurate_math();
  const int gm = gamma_mode(a.gamma);
  LAUNCH_BY_MODE(cls_losses_fused_kernel, fast, a.beta == 0.0f, gm, dim3(blocks), dim3(kThreads),
                 0, s, a, fs, normalizer, fg_num, partials, kMaxBlocks);
  hipLaunchKernelGGL(fused_finalize_kernel, dim3(n_levels), dim3(kThreads), 0, s, a,
                     (const double*)partials, kMaxBlocks, distill_losses, focal_losses, a.scale,
                     fs.scale);
  return (int)hipGetLastError();
}

size_t ssad_select_smooth_l1_workspace_bytes(int n_levels) {
  return sizeof(double) * kSl1Blocks * (size_t)(n_levels > 0 ? n_levels : 1);
}

static int sl1_args(const ssad_smooth_l1_level* lv, int n_levels, float beta, float scale, Sl1Args* a,
                    int* max_m, long long* max_n) {
  if (n_levels < 1 || n_levels > SSAD_MAX_LEVELS || !lv || !(beta > 0.0f) || !(scale >= 0.0f))
    return SSAD_E_BADARG;
  a->n_levels = n_levels;
  *max_m = 0; *max_n = 0;
  for (int l = 0; l < n_levels; ++l) {
    const ssad_smooth_l1_level& s = lv[l];
    if (s.N < 0 || s.D < 0 || s.H < 0 || s.W < 0 || s.M < 0 || s.M >= (1 << 29)) return SSAD_E_BADARG;
    a->lv[l] = s;
    if (s.M > *max_m) *max_m = s.M;
    const long long n = (long long)s.N * s.D * s.H * s.W;
    if (n > *max_n) *max_n = n;
  }
  return 0;
}

int ssad_select_smooth_l1_levels(const ssad_smooth_l1_level* levels_host, int n_levels, const float* S,
                                 const float* dloss, float beta, float scale, int want_forward,
                                 void* workspace, size_t workspace_bytes, ssad_stream_t stream) {
  Sl1Args a;
  int max_m = 0;
  long long max_n = 0;
  const int rc = sl1_args(levels_host, n_levels, beta, scale, &a, &max_m, &max_n);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  int grid = (max_m * 4 + kThreads - 1) / kThreads;
  if (grid < 1) grid = 1;
  if (grid > kSl1Blocks) grid = kSl1Blocks;
  if (want_forward) {
    if (!workspace || workspace_bytes < ssad_select_smooth_l1_workspace_bytes(n_levels)) return SSAD_E_WORKSPACE;
    for (int l = 0; l < n_levels; ++l) if (!a.lv[l].loss) return SSAD_E_BADARG;
    double* partials = (double*)workspace;
    hipLaunchKernelGGL(smooth_l1_fwd_kernel, dim3(grid, n_levels), dim3(kThreads), 0, st, a, S, beta, partials);
    hipLaunchKernelGGL(smooth_l1_finalize_kernel, dim3(n_levels), dim3(kThreads), 0, st, a,
                       (const double*)partials, grid, scale);
  }
  if (dloss) {
    for (int l = 0; l < n_levels; ++l) if (!a.lv[l].dY_hat) return SSAD_E_BADARG;
    if (max_n > 0) {
      long long zb = (max_n / 4 + kThreads - 1) / kThreads;
      if (zb < 1) zb = 1;
      if (zb > 2048) zb = 2048;
      hipLaunchKernelGGL(smooth_l1_zero_kernel, dim3((int)zb, n_levels), dim3(kThreads), 0, st, a);
    }
    if (max_m > 0)
      hipLaunchKernelGGL(smooth_l1_bwd_kernel, dim3(grid > 256 ? 256 : grid, n_levels), dim3(kThreads), 0, st,
                         a, S, dloss, beta, scale);
  }
  return (int)hipGetLastError();
}

int ssad_select_smooth_l1_forward(const float* Y_hat, const float* Y, const float* L,
                                  const float* S, int N, int D, int H, int W, int M, float beta,
                                  float scale, float* loss, void* workspace, size_t workspace_bytes,
                                  ssad_stream_t stream) {
  const ssad_smooth_l1_level lv{Y_hat, Y, L, loss, nullptr, N, D, H, W, M};
  return ssad_select_smooth_l1_levels(&lv, 1, S, nullptr, beta, scale, 1, workspace, workspace_bytes, stream);
}

int ssad_select_smooth_l1_backward(const float* Y_hat, const float* Y, const float* L,
                                   const float* S, const float* dloss, int N, int D, int H, int W,
                                   int M, float beta, float scale, float* dY_hat,
                                   ssad_stream_t stream) {
  if (N < 0 || D < 0 || H < 0 || W < 0 || M < 0 || !(beta > 0.0f) || !(scale >= 0.0f) || !dloss)
    return SSAD_E_BADARG;
  if (M == 0) return 0;
  // dY_hat zero-filled by the caller (the operator does it with its own math::Set)
  Sl1Args a;
  a.n_levels = 1;
  a.lv[0] = ssad_smooth_l1_level{Y_hat, Y, L, nullptr, dY_hat, N, D, H, W, M};
  const int grid = (M * 4 + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(smooth_l1_bwd_kernel, dim3(grid > 256 ? 256 : grid, 1), dim3(kThreads), 0,
                     (hipStream_t)stream, a, S, dloss, beta, scale);
  return (int)hipGetLastError();
}

size_t ssad_pow_sum_workspace_bytes(int n_inputs) {
  (void)n_inputs;
  return sizeof(double) * kMaxBlocks;
}

int ssad_pow_sum(
    const float* const* inputs_host, const int64_t* sizes_host, int n_inputs,
    float power, float* out, void* workspace, size_t workspace_bytes,
    ssad_stream_t stream) {
  if (n_inputs < 1 || !out) return SSAD_E_BADARG;
  if (!workspace || workspace_bytes < sizeof(double) * kMaxBlocks) return SSAD_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  double* partials = (double*)workspace;
  const bool fast = !accurate_math();
  // groups of SSAD_MAX_POWSUM_INPUTS inputs per launch; later groups add on
  for (int g0 = 0; g0 < n_inputs; g0 += SSAD_MAX_POWSUM_INPUTS) {
    PowArgs a;
    const int cnt = (n_inputs - g0 < SSAD_MAX_POWSUM_INPUTS) ? n_inputs - g0 : SSAD_MAX_POWSUM_INPUTS;
    a.n_inputs = cnt;
    a.power = power;
    long long total = 0;
    for (int j = 0; j < cnt; ++j) {
      if (sizes_host[g0 + j] < 0) return SSAD_E_BADARG;
      total += sizes_host[g0 + j];
    }
    int start = 0;
    const int budget = 2048;  // 8 workgroups per CU
    for (int j = 0; j < cnt; ++j) {
      const long long n = sizes_host[g0 + j];
      a.ptr[j] = inputs_host[g0 + j];
      a.n[j] = n;
      long long want = (n + (long long)kThreads * 16 - 1) / ((long long)kThreads * 16);
      long long share = total > 0 ? (n * budget + total - 1) / total : 1;
      long long b = want < share ? want : share;
      if (b < 1) b = 1;
      a.block_start[j] = start;
      a.blocks[j] = (int)b;
      start += (int)b;
    }
    for (int j = cnt; j < SSAD_MAX_POWSUM_INPUTS; ++j) {
      a.ptr[j] = nullptr; a.n[j] = 0; a.block_start[j] = start; a.blocks[j] = 0;
    }
    if (fast) hipLaunchKernelGGL(pow_sum_kernel<true>, dim3(start), dim3(kThreads), 0, s, a, partials);
    else hipLaunchKernelGGL(pow_sum_kernel<false>, dim3(start), dim3(kThreads), 0, s, a, partials);
    hipLaunchKernelGGL(pow_sum_finalize_kernel, dim3(1), dim3(kThreads), 0, s,
                       (const double*)partials, start, out, g0 > 0 ? 1 : 0);
  }
  return (int)hipGetLastError();
}

}  // extern "C"
