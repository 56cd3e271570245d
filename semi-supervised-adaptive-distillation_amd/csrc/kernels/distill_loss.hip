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
  int group_start;   // first arrival-group counter of this level (in-launch finalize)
};

struct LaunchArgs {
  LevelArgs lv[SSAD_MAX_LEVELS];
  int n_levels;
  float gamma, alpha, beta, scale;
  int ignored;
  int fences;          // ticket memory ordering (arrive_last)
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

// ---- in-launch finalize ------------------------------------------------------
// The last workgroup of a level to arrive sums that level's partials in index order (the
// order the separate finalize launch used: results are bit-identical and independent of which
// workgroup happens to be last) -- one launch instead of two per loss.  Cross-XCD visibility
// (per-XCD L2s are not coherent): partials are published with 8-byte agent-scope atomic stores
// (write-through), the store is waited for, then the arrival counter is bumped with an agent-scope
// atomic; the reducer reads the partials with agent-scope atomic loads.  Arrival is two-level
// (groups of kTicketGroup workgroups, then one counter per level) so that no single address sees
// thousands of same-address atomics.  Counters live in the caller's workspace, must be ZERO before
// the first launch (the workspace contract in ssad_kernels.h) and are left zero by every launch.
constexpr int kTicketGroup = 32;
constexpr int kTicketStride = 16;                         // ints: one 64-byte line per group counter
constexpr int kTicketGroups = kMaxBlocks / kTicketGroup + SSAD_MAX_LEVELS;
constexpr size_t kTicketBytes = sizeof(unsigned) * ((size_t)kTicketGroups * kTicketStride + 64);

struct Tickets {
  unsigned* group;      // [kTicketGroups][kTicketStride]
  unsigned* level;      // [SSAD_MAX_LEVELS] (padded)
};
__host__ __device__ inline Tickets tickets_at(void* base) {
  Tickets t;
  t.group = (unsigned*)base;
  t.level = t.group + (size_t)kTicketGroups * kTicketStride;
  return t;
}

__device__ __forceinline__ void publish(double* slot, double v) {
  __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double peek(const double* slot) {
  return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Thread 0 only, after its publish() calls.  lb = workgroup index within the level, nb = the
// level's workgroup count, g0 = index of the level's first group counter.  True for exactly one
// workgroup of the level: the last to arrive.
// Memory ordering, `fences`:
//   0  relaxed agent-scope atomics + s_waitcnt vmcnt(0): the partials are agent-scope atomic (write-through) stores
//      and gfx9 counts stores in vmcnt, so they have left the CU before the arrival is visible; the reducer reads
//      them with agent-scope atomic loads behind a control dependence + barrier.  Correct on gfx950 by the ISA's
//      counter semantics, not by the C++ memory model.
//   1  the arrival is a RELEASE read-modify-write (the chain of RMWs on one counter is a release sequence) and the
//      reducing workgroup issues an ACQUIRE fence before its peeks: what the memory model asks for.
//   2  every arrival ACQ_REL.
// Measured on MI355X, config 3's fused classification-loss launch (2.2 G logits, 2048 workgroups): mode 0 = 0.308 ms,
// mode 1 = 0.798 ms, mode 2 = 0.828 ms (PowSum 0.105 / 0.195 / 0.203): the agent-scope RELEASE alone -- an L2
// write-back (buffer_wbl2) per arrival under the other workgroups' streaming traffic -- costs 2.6x, so the default
// stays 0 (ticket_fences()); modes 1 / 2 are kept selectable (SSAD_TICKET_FENCES) and tested for equal bits.
__device__ __forceinline__ unsigned ticket_add(unsigned* p, int fences) {
  if (fences == 2) return __hip_atomic_fetch_add(p, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  if (fences == 1) return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool arrive_last(const Tickets& t, int level, int lb, int nb, int g0, int fences) {
  if (fences == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the published partials have left
  const int g = lb / kTicketGroup;
  const int gsize = (nb - g * kTicketGroup) < kTicketGroup ? (nb - g * kTicketGroup) : kTicketGroup;
  unsigned* gc = t.group + (size_t)(g0 + g) * kTicketStride;
  if (ticket_add(gc, fences) != (unsigned)gsize - 1u) return false;
  __hip_atomic_store(gc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int ngroups = (nb + kTicketGroup - 1) / kTicketGroup;
  if (ticket_add(t.level + level, fences) != (unsigned)ngroups - 1u) return false;
  __hip_atomic_store(t.level + level, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// Every thread of the reducing workgroup, after the __syncthreads() that broadcast arrive_last():
// the peeks below must not be satisfied by anything fetched before the last arrival was seen.
__device__ __forceinline__ void reducer_acquire(int fences) {
  if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
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

// ---- packed (<2 x float>) element math of the hot configuration ------------------------------
// gamma = 2, beta = 0, fast math, 16-byte aligned planes: see the note at cls_losses_fused_kernel.
typedef float v2f __attribute__((ext_vector_type(2)));

struct FusedK {              // wave-uniform constants of the packed path
  float w_pos, w_neg;        // alpha / Np, (1 - alpha) / Np          (distillation)
  float two_np;              // gamma * Np = 2 Np  (S = -Np * ce)
  float alpha, c12a;         // alpha, 1 - 2 alpha
  float neg_dmult;           // -(scale / Np)        (dloss = 1)
  float zp, zn;              // alpha_f / Nf, (1 - alpha_f) / Nf      (focal)
  float g1c, g2c;            // -zp * scale_f, zn * scale_f
};

struct PairOut { v2f store, a, b; };

// One pair of logits of one class plane.  keepf = 1/0 per position (label != ignored_label),
// notign = 1/0 (label != -1: the focal loss's own ignore value), tm1 = label - 1 (foreground class
// index); d = class index of the plane.
// WHAT: 0 = distillation loss only (r.a), 1 = distillation gradient only (r.store), 2 = everything.
template <int WHAT>
__device__ __forceinline__ PairOut fused_pair(v2f x, v2f q, v2f keepf, v2f notign, int tm1x, int tm1y,
                                              int d, const FusedK& K) {
  constexpr float kL2E = 1.4426950408889634f, kLN2 = 0.6931471805599453f;
  constexpr float kNegLogFltMin = 87.33654475f;
  v2f e, l2, inv, xpos;
  e.x = __builtin_amdgcn_exp2f(fabsf(x.x) * -kL2E);
  e.y = __builtin_amdgcn_exp2f(fabsf(x.y) * -kL2E);
  const v2f onepe = e + 1.0f;
  l2.x = __builtin_amdgcn_logf(onepe.x);  l2.y = __builtin_amdgcn_logf(onepe.y);
  inv.x = __builtin_amdgcn_rcpf(onepe.x); inv.y = __builtin_amdgcn_rcpf(onepe.y);
  const v2f sp = l2 * kLN2;                       // log(1 + e^-|x|)
  xpos.x = fmaxf(x.x, 0.0f); xpos.y = fmaxf(x.y, 0.0f);
  const v2f B = sp + xpos;                        // -log(1 - p)
  v2f A = sp + (xpos - x);                        // -log p  (xpos - x = max(-x, 0) exactly)
  A.x = fminf(A.x, kNegLogFltMin); A.y = fminf(A.y, kNegLogFltMin);   // log(max(FLT_MIN, p))
  const v2f einv = e * inv;
  v2f p;
  p.x = x.x >= 0.0f ? inv.x : einv.x;
  p.y = x.y >= 0.0f ? inv.y : einv.y;
  const v2f omq = 1.0f - q;
  const v2f qq = q * omq;                         // > 0 exactly when q is in (0, 1)
  v2f mult;                                       // NaN for a teacher probability outside (0, 1):
  mult.x = qq.x > 0.0f ? keepf.x : __builtin_nanf("");   // 0 * log 0 of .cu:58-59, also when the
  mult.y = qq.y > 0.0f ? keepf.y : __builtin_nanf("");   // anchor is ignored
  const v2f dl = B - x * q;
  v2f edl;
  edl.x = __builtin_amdgcn_exp2f(dl.x * -kL2E);
  edl.y = __builtin_amdgcn_exp2f(dl.y * -kL2E);
  const v2f at = 1.0f - edl;
  const v2f pg = at * at;
  const v2f ceN = (q * A) * K.w_pos + (omq * B) * K.w_neg;            // -ce / Np
  PairOut r;
  r.a = (pg * ceN) * mult;
  const v2f diff = q - p;
  const v2f t1 = ((diff * at) * edl) * (ceN * K.two_np);
  const v2f t2 = pg * (diff * K.alpha - (omq * p) * K.c12a);
  const v2f gd = (t1 + t2) * (mult * K.neg_dmult);
  if constexpr (WHAT != 2) {
    r.b = v2f{0.0f, 0.0f};
    r.store = gd;
    return r;
  }
  // SigmoidFocalLoss (sigmoid_focal_loss_op.cu:33-66, 74-105), gamma = 2
  const v2f omp = 1.0f - p;
  const v2f af = omp * omp, bf = p * p;
  const v2f l1 = (af * A) * K.zp;
  const v2f nz = notign * K.zn, ng = notign * K.g2c;
  const v2f l2f = (bf * B) * nz;
  const v2f g1 = (af * (omp + (p * A) * 2.0f)) * K.g1c;
  const v2f g2 = (bf * ((B * omp) * 2.0f + p)) * ng;
  const bool cx = tm1x == d, cy = tm1y == d;
  r.b.x = cx ? l1.x : l2f.x;
  r.b.y = cy ? l1.y : l2f.y;
  v2f fg;
  fg.x = cx ? g1.x : g2.x;
  fg.y = cy ? g1.y : g2.y;
  r.store = gd + fg;
  return r;
}

// The packed traversal: work decomposition of traverse() for the 16-byte aligned case, pairs
// of positions, four class planes in flight per thread.
template <int WHAT>
__device__ __forceinline__ Acc2 traverse_packed(const LevelArgs& L, int lb, const FusedK& K, int ignored) {
  const int pl = 1 << L.pl_shift, cl = kThreads >> L.pl_shift;
  const int pi = threadIdx.x & (pl - 1), ci = threadIdx.x >> L.pl_shift;
  const int hw = L.hw;
  v2f sa = {0.0f, 0.0f}, sb = {0.0f, 0.0f};
  for (int item = lb; item < L.items; item += L.blocks) {
    const int sc = item / L.cgroups;
    const int cg = item - sc * L.cgroups;
    const int slab = sc / L.chunks;
    const int chunk = sc - slab * L.chunks;
    const int c_begin = cg * L.cper + ci;
    const int c_end = (cg + 1) * L.cper < L.classes ? (cg + 1) * L.cper : L.classes;
    const int pos = (chunk * pl + pi) * 4;
    if (pos >= hw) continue;
    const float* __restrict__ xs = L.x + (size_t)slab * L.slab + pos;
    const float* __restrict__ qs = L.q + (size_t)slab * L.slab + pos;
    float* __restrict__ ds = WHAT != 0 ? L.out + (size_t)slab * L.slab + pos : nullptr;
    const int4 gv = *reinterpret_cast<const int4*>(L.g + (size_t)slab * hw + pos);
    // everything that depends only on the label, once per 4 positions
    const v2f k01 = {gv.x != ignored ? 1.0f : 0.0f, gv.y != ignored ? 1.0f : 0.0f};
    const v2f k23 = {gv.z != ignored ? 1.0f : 0.0f, gv.w != ignored ? 1.0f : 0.0f};
    const v2f n01 = {gv.x != -1 ? 1.0f : 0.0f, gv.y != -1 ? 1.0f : 0.0f};
    const v2f n23 = {gv.z != -1 ? 1.0f : 0.0f, gv.w != -1 ? 1.0f : 0.0f};
    const int t0 = gv.x - 1, t1 = gv.y - 1, t2 = gv.z - 1, t3 = gv.w - 1;
    auto plane = [&](int c, const float4& xv, const float4& qv) {
      const PairOut r0 = fused_pair<WHAT>(v2f{xv.x, xv.y}, v2f{qv.x, qv.y}, k01, n01, t0, t1, c, K);
      const PairOut r1 = fused_pair<WHAT>(v2f{xv.z, xv.w}, v2f{qv.z, qv.w}, k23, n23, t2, t3, c, K);
      if constexpr (WHAT != 1) sa += r0.a + r1.a;
      if constexpr (WHAT == 2) sb += r0.b + r1.b;
      if constexpr (WHAT != 0)
        st4(ds + c * hw, make_float4(r0.store.x, r0.store.y, r1.store.x, r1.store.y));
    };
    int c = c_begin;
    for (; c + 3 * cl < c_end; c += 4 * cl) {
      float4 xv[4], qv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xv[u] = ld4(xs + (c + u * cl) * hw);
        qv[u] = ld4(qs + (c + u * cl) * hw);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) plane(c + u * cl, xv[u], qv[u]);
    }
    for (; c < c_end; c += cl) plane(c, ld4(xs + c * hw), ld4(qs + c * hw));
  }
  return Acc2{sa.x + sa.y, sb.x + sb.y};
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
  Acc2 acc;
  if (FAST && GAMMA_MODE == 2 && BETA0 && L.vec4) {
    const FusedK K{w_pos, w_neg, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    acc = traverse_packed<0>(L, lb, K, ignored);
  } else {
    acc = traverse<0, true>(L, lb, [&](float x, float q, int t, int) {
      return loss_elem<FAST, GAMMA_MODE, BETA0>(x, q, t != ignored, gamma, beta, w_pos, w_neg);
    });
  }
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
  if (FAST && GAMMA_MODE == 2 && BETA0 && L.vec4) {
    const FusedK K{alpha / np, (1.0f - alpha) / np, 2.0f * np, alpha, 1.0f - 2.0f * alpha, -mult,
                   0.0f, 0.0f, 0.0f, 0.0f};
    traverse_packed<1>(L, lb, K, ignored);
    return;
  }
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
//
// Round 3: the pass was VALU-bound, not HBM-bound -- 78 vector instructions + 4 transcendentals
// per logit (ISA count) = 0.32 ms of issue time for 137.5 M logits against 0.26 ms of streaming.
// The hot path (gamma = 2 for both losses, beta = 0, fast math, 16-byte aligned planes) now works
// on PAIRS of positions in <2 x float> so that adds / multiplies / fmas issue as v_pk_*_f32 (two
// results per lane per instruction), shares every sub-expression of the four formulas
// (S = -Np * ce; -log p and -log(1-p) kept positive; the correctly rounded division behind
// __frcp_rn replaced by v_rcp_f32; raw v_exp_f32 / v_log_f32 without denormal range fix-ups: their
// arguments are in [1, 2] resp. <= 0), hoists everything that depends only on the label out of
// the class loop, and reduces both sums in the last-arriving workgroup (no finalize launch).
template <bool FAST, int GAMMA_MODE, bool BETA0>
__global__ __launch_bounds__(kThreads) void cls_losses_fused_kernel(
    const LaunchArgs args, const FocalScalars fs, const float* __restrict__ normalizer,
    const float* __restrict__ fg_num, double* __restrict__ partials, int focal_offset,
    float* __restrict__ out_a, float* __restrict__ out_b) {
  const int level = find_level(args, blockIdx.x);
  const LevelArgs& L = args.lv[level];
  const int lb = blockIdx.x - L.block_start;
  const float np = fmaxf(normalizer[0], 1.0f);
  const float w_pos = args.alpha / np, w_neg = (1.0f - args.alpha) / np;
  const float d_mult = args.scale / np;                  // dloss = 1 (utils/blob.py:166-172)
  const float nf = fmaxf(fg_num[0], 1.0f);
  const float zp = fs.alpha / nf, zn = (1.0f - fs.alpha) / nf;
  const float gamma = args.gamma, alpha = args.alpha, beta = args.beta, fgamma = fs.gamma;
  const float f_mult = fs.scale;
  const int ignored = args.ignored;
  Acc2 acc{0.0f, 0.0f};
  if (FAST && GAMMA_MODE == 2 && BETA0 && L.vec4) {
    const FusedK K{w_pos, w_neg, 2.0f * np, alpha, 1.0f - 2.0f * alpha, -d_mult, zp, zn, -zp * f_mult,
                   zn * f_mult};
    acc = traverse_packed<2>(L, lb, K, ignored);
  } else {
    acc = traverse<2, true>(L, lb, [&](float x, float q, int t, int d) {
      const bool keep = t != ignored;
      Fused r;
      r.a = loss_elem<FAST, GAMMA_MODE, BETA0>(x, q, keep, gamma, beta, w_pos, w_neg);
      r.b = focal_loss_elem<FAST, 2>(x, t, d, fgamma, zp, zn);
      r.store = grad_elem<FAST, GAMMA_MODE, BETA0>(x, q, keep, gamma, alpha, beta, d_mult) +
                focal_grad_elem<FAST, 2>(x, t, d, fgamma, zp, zn, f_mult);
      return r;
    });
  }
  double ta = block_sum((double)acc.a);
  __syncthreads();
  double tb = block_sum((double)acc.b);
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    publish(partials + blockIdx.x, ta);
    publish(partials + focal_offset + blockIdx.x, tb);
    s_last = arrive_last(tickets_at(partials + 2 * (size_t)focal_offset), level, lb, L.blocks,
                         L.group_start, args.fences) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  reducer_acquire(args.fences);
  // this level's last workgroup: fixed-order sum of the level's partials, then the float
  // multiply by scale (math::Scale on one element, .cu:137-138)
  double va = 0.0, vb = 0.0;
  for (int i = threadIdx.x; i < L.blocks; i += kThreads) {
    va += peek(partials + L.block_start + i);
    vb += peek(partials + focal_offset + L.block_start + i);
  }
  ta = block_sum(va);
  __syncthreads();
  tb = block_sum(vb);
  if (threadIdx.x == 0) {
    out_a[level] = (float)ta * args.scale;
    out_b[level] = (float)tb * fs.scale;
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
  int fences;
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
    const PowArgs args, double* __restrict__ partials, float* __restrict__ out, int accumulate) {
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
  double t = block_sum(dacc);
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    publish(partials + blockIdx.x, t);
    s_last = arrive_last(tickets_at(partials + kMaxBlocks), 0, blockIdx.x, gridDim.x, 0, args.fences) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  reducer_acquire(args.fences);
  // last workgroup to arrive: the partials in index order (deterministic), one launch
  double v = 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += kThreads) v += peek(partials + i);
  t = block_sum(v);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.0f) + (float)t;
}

// ---- host side ---------------------------------------------------------------

int gamma_mode(float g) { return g == 2.0f ? 2 : (g == 1.0f ? 1 : 0); }

// SSAD_TICKET_FENCES: memory ordering of the in-launch finalize (see arrive_last).  Default 0: same-call A/B on
// MI355X (tools/dbg/r4_fence_ab.sh, bench.py --workload heads, two rounds): fused classification losses
// 0.308 / 0.798 / 0.828 ms for modes 0 / 1 / 2, PowSum 0.105 / 0.195 / 0.203 ms -- already the release half
// (buffer_wbl2 at every one of 2048 arrivals) costs 2.6x on an HBM-bound streaming kernel.
int ticket_fences() {
  static const int v = [] { const char* e = getenv("SSAD_TICKET_FENCES"); return e ? atoi(e) : 0; }();
  return v < 0 ? 0 : (v > 2 ? 2 : v);
}

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
  a.fences = ticket_fences();
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
    constexpr int cg_want = 4;        // (sweep: tools/loss_sweep.py, round 2)
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
  constexpr int max_blocks_env = kMaxBlocks;
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
    L.group_start = l == 0 ? 0 : a.lv[l - 1].group_start +
                                  (a.lv[l - 1].blocks + kTicketGroup - 1) / kTicketGroup;
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
  return 2 * sizeof(double) * kMaxBlocks + kTicketBytes;
}

static int cls_losses_fused_impl(const ssad_distill_level* levels_host, int n_levels,
                                 const float* normalizer, const float* fg_num,
                                 const ssad_distill_params* distill_host,
                                 const ssad_focal_params* focal_host, float* distill_losses,
                                 float* focal_losses, void* workspace, size_t workspace_bytes,
                                 ssad_stream_t stream, bool zero_tickets) {
  if (!focal_host || !distill_host || focal_host->gamma != 2.0f ||
      focal_host->num_classes != distill_host->num_classes)
    return SSAD_E_BADARG;     // the fused kernel specialises the focal gamma = 2 of RetinaNet
  LaunchArgs a;
  int blocks = 0;
  const int rc = build_args(levels_host, n_levels, distill_host, &a, &blocks, true);
  if (rc) return rc;
  if (!workspace || workspace_bytes < ssad_cls_losses_fused_workspace_bytes(n_levels)) return SSAD_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  double* partials = (double*)workspace;
  // arrival counters: zeroed here unless the caller vouches for them (a few KB; also heals a buffer
  // whose counters an aborted launch left non-zero)
  if (zero_tickets &&
      hipMemsetAsync(partials + 2 * (size_t)kMaxBlocks, 0, kTicketBytes, s) != hipSuccess)
    return (int)hipGetLastError();
  const FocalScalars fs{focal_host->gamma, focal_host->alpha, focal_host->scale};
  const bool fast = !accurate_math();
  const int gm = gamma_mode(a.gamma);
  // one launch: the sums are finished by each level's last-arriving workgroup
  LAUNCH_BY_MODE(cls_losses_fused_kernel, fast, a.beta == 0.0f, gm, dim3(blocks), dim3(kThreads),
                 0, s, a, fs, normalizer, fg_num, partials, kMaxBlocks, distill_losses, focal_losses);
  return (int)hipGetLastError();
}

int ssad_cls_losses_fused(const ssad_distill_level* levels_host, int n_levels,
                          const float* normalizer, const float* fg_num,
                          const ssad_distill_params* distill_host,
                          const ssad_focal_params* focal_host, float* distill_losses,
                          float* focal_losses, void* workspace, size_t workspace_bytes,
                          ssad_stream_t stream) {
  return cls_losses_fused_impl(levels_host, n_levels, normalizer, fg_num, distill_host, focal_host,
                               distill_losses, focal_losses, workspace, workspace_bytes, stream, true);
}

int ssad_cls_losses_fused_prezeroed(const ssad_distill_level* levels_host, int n_levels,
                                    const float* normalizer, const float* fg_num,
                                    const ssad_distill_params* distill_host,
                                    const ssad_focal_params* focal_host, float* distill_losses,
                                    float* focal_losses, void* workspace, size_t workspace_bytes,
                                    ssad_stream_t stream) {
  return cls_losses_fused_impl(levels_host, n_levels, normalizer, fg_num, distill_host, focal_host,
                               distill_losses, focal_losses, workspace, workspace_bytes, stream, false);
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
  return sizeof(double) * kMaxBlocks + kTicketBytes;
}

static int pow_sum_impl(
    const float* const* inputs_host, const int64_t* sizes_host, int n_inputs,
    float power, float* out, void* workspace, size_t workspace_bytes,
    ssad_stream_t stream, bool zero_tickets) {
  if (n_inputs < 1 || !out) return SSAD_E_BADARG;
  if (!workspace || workspace_bytes < ssad_pow_sum_workspace_bytes(n_inputs)) return SSAD_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  double* partials = (double*)workspace;
  if (zero_tickets && hipMemsetAsync(partials + kMaxBlocks, 0, kTicketBytes, s) != hipSuccess)
    return (int)hipGetLastError();
  const bool fast = !accurate_math();
  // groups of SSAD_MAX_POWSUM_INPUTS inputs per launch; later groups add on
  for (int g0 = 0; g0 < n_inputs; g0 += SSAD_MAX_POWSUM_INPUTS) {
    PowArgs a;
    const int cnt = (n_inputs - g0 < SSAD_MAX_POWSUM_INPUTS) ? n_inputs - g0 : SSAD_MAX_POWSUM_INPUTS;
    a.n_inputs = cnt;
    a.power = power;
    a.fences = ticket_fences();
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
    const int accumulate = g0 > 0 ? 1 : 0;
    if (fast) hipLaunchKernelGGL(pow_sum_kernel<true>, dim3(start), dim3(kThreads), 0, s, a, partials, out,
                                 accumulate);
    else hipLaunchKernelGGL(pow_sum_kernel<false>, dim3(start), dim3(kThreads), 0, s, a, partials, out,
                            accumulate);
  }
  return (int)hipGetLastError();
}

int ssad_pow_sum(
    const float* const* inputs_host, const int64_t* sizes_host, int n_inputs,
    float power, float* out, void* workspace, size_t workspace_bytes,
    ssad_stream_t stream) {
  return pow_sum_impl(inputs_host, sizes_host, n_inputs, power, out, workspace, workspace_bytes, stream, true);
}

int ssad_pow_sum_prezeroed(
    const float* const* inputs_host, const int64_t* sizes_host, int n_inputs,
    float power, float* out, void* workspace, size_t workspace_bytes,
    ssad_stream_t stream) {
  return pow_sum_impl(inputs_host, sizes_host, n_inputs, power, out, workspace, workspace_bytes, stream, false);
}

}  // extern "C"
