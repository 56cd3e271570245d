// anchor_labels.hip -- RetinaNet anchor labelling on the device (row f4): the
// classification labels, the foreground location list and the box regression
// targets that the reference builds in numpy inside its data-loader threads
// (detectron/lib/roi_data/retinanet.py:97-306) for
//   retnet_cls_labels_fpn{l}        int32  N x A x h x w
//   retnet_roi_fg_bbox_locs_fpn{l}  float  M x 4   [image, 4*anchor, y, x]
//   retnet_roi_bbox_targets_fpn{l}  float  M x 4
//   retnet_fg_num, retnet_bg_num
//
// Semantics kept exactly (they decide which anchors train):
//  * anchors = float32(float64 cell anchor + shift), all fields of all levels
//    concatenated (data_utils.py:52-103, retinanet.py:75-94);
//  * IoU with the +1 convention in exactly the mixed float/double arithmetic the
//    cython source compiles to (utils/cython_bbox.pyx:56-73) -- explicit *_rn
//    intrinsics, no FMA contraction, because labels depend on exact equality of
//    IoU values;
//  * label = class of the arg-max gt (first maximum) for every anchor that
//    attains some gt's best IoU (all ties) or has IoU >= 0.5; then every anchor
//    with best IoU < 0.4 becomes background 0 (overriding the tie rule); the rest -1
//    (retinanet.py:214-243).  fg_num counts the foreground BEFORE that override,
//    bg_num = sum over images of (num_bg + 1) * (C - 1) + num_fg * (C - 2) (:300-304);
//  * the fg list is taken from the WHOLE field of anchors (the reference indexes
//    the un-cropped label map, :278-293), ordered image, anchor, y, x; only the
//    label blob is cropped to h x w = int(blob size / stride).
// Deviation: an image without ground truth is an assertion failure in the
// reference (:118-119); here all its anchors become background.
//
// Five small launches: per-anchor best IoU + per-gt best IoU (atomicMax on the
// float bits), label assignment + cropped label maps + per-image counts, then an
// order-preserving stream compaction of the foreground (block counts, scan,
// scatter with wave-ballot prefixes).

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ssad_kernels.h"

namespace {

constexpr int kT = 256;
constexpr int kCB = 1024;          // compaction block

struct ALArgs {
  const double* cells;             // [levels][A][4]
  const float* gt_boxes;           // [N][Gmax][4]
  const int* gt_classes;           // [N][Gmax]
  const int* gt_counts;            // [N]
  int levels, A, k_min, N, Gmax;
  int fs[SSAD_MAX_LEVELS];         // field size per level
  int start[SSAD_MAX_LEVELS + 1];  // first anchor index of each level (per image)
  int h[SSAD_MAX_LEVELS], w[SSAD_MAX_LEVELS];
  int T;                           // anchors per image
  float pos_thr, neg_thr;
  int num_classes;
  // workspace
  float* a_max;                    // [N][T]
  int* a_arg;                      // [N][T]
  unsigned* g_max;                 // [N][Gmax]  (float bits; IoU >= 0 so bit order = value order)
  signed char* label;              // [N][T] final label (-1, 0, 1..)  (classes < 128)
  int* img_counts;                 // [N][2] num_fg, num_bg
  int* blk_counts;                 // [levels][max_blocks]
  int* blk_offsets;
  int max_blocks;
  // outputs
  int* labels_out[SSAD_MAX_LEVELS];
  float* locs_out[SSAD_MAX_LEVELS];
  float* targets_out[SSAD_MAX_LEVELS];
  int capacity;
  int* counts_out;                 // [levels]
  float* fg_bg_out;                // [2]
};

struct Anchor { float x1, y1, x2, y2; int level, a, y, x; };

__device__ __forceinline__ Anchor anchor_of(const ALArgs& p, int t) {
  Anchor r;
  int l = 0;
#pragma unroll
  for (int i = 1; i < SSAD_MAX_LEVELS; ++i)
    if (i < p.levels && t >= p.start[i]) l = i;
  const int fs = p.fs[l];
  int rem = t - p.start[l];
  r.level = l;
  r.a = rem / (fs * fs);
  rem -= r.a * fs * fs;
  r.y = rem / fs;
  r.x = rem - r.y * fs;
  const double stride = (double)(1 << (p.k_min + l));
  const double* c = p.cells + ((long long)l * p.A + r.a) * 4;
  const double sx = (double)r.x * stride, sy = (double)r.y * stride;
  r.x1 = (float)(c[0] + sx); r.y1 = (float)(c[1] + sy);
  r.x2 = (float)(c[2] + sx); r.y2 = (float)(c[3] + sy);
  return r;
}

// cython_bbox.pyx:56-73 in the arithmetic the cython source compiles to (checked bit
// for bit against the compiled reference through the oracle): coordinate differences
// are float, but Cython emits the literal 1 next to a C float as the double 1.0, so
// "+ 1", both area products and the union sum are double and round to float only on
// assignment to the float locals; iw * ih and the final division are float.
__device__ __forceinline__ float iou(const Anchor& b, const float* q) {
  const float box_area = (float)__dmul_rn((double)__fsub_rn(q[2], q[0]) + 1.0,
                                          (double)__fsub_rn(q[3], q[1]) + 1.0);
  const float iw = (float)((double)__fsub_rn(fminf(b.x2, q[2]), fmaxf(b.x1, q[0])) + 1.0);
  if (!(iw > 0.0f)) return 0.0f;
  const float ih = (float)((double)__fsub_rn(fminf(b.y2, q[3]), fmaxf(b.y1, q[1])) + 1.0);
  if (!(ih > 0.0f)) return 0.0f;
  const double area = __dmul_rn((double)__fsub_rn(b.x2, b.x1) + 1.0,
                                (double)__fsub_rn(b.y2, b.y1) + 1.0);
  const float inter = __fmul_rn(iw, ih);
  const float ua = (float)__dsub_rn(__dadd_rn(area, (double)box_area), (double)inter);
  return __fdiv_rn(inter, ua);
}

__global__ __launch_bounds__(kT) void al_iou_max_kernel(const ALArgs p) {
  const int n = blockIdx.y;
  const int t = blockIdx.x * kT + threadIdx.x;
  if (t >= p.T) return;
  const Anchor b = anchor_of(p, t);
  const int G = min(max(p.gt_counts[n], 0), p.Gmax);   // a count beyond the padded width would read past gt_boxes
  const float* gts = p.gt_boxes + (long long)n * p.Gmax * 4;
  float best = 0.0f;
  int arg = 0;
  for (int g = 0; g < G; ++g) {
    const float ov = iou(b, gts + g * 4);
    if (ov > best) { best = ov; arg = g; }      // first maximum, as numpy argmax
    if (ov > 0.0f) atomicMax(p.g_max + (long long)n * p.Gmax + g, __float_as_uint(ov));
  }
  p.a_max[(long long)n * p.T + t] = best;
  p.a_arg[(long long)n * p.T + t] = arg;
}

__global__ __launch_bounds__(kT) void al_assign_kernel(const ALArgs p) {
  __shared__ int s_fg, s_bg;
  if (threadIdx.x == 0) { s_fg = 0; s_bg = 0; }
  __syncthreads();
  const int n = blockIdx.y;
  const int t = blockIdx.x * kT + threadIdx.x;
  if (t < p.T) {
    const Anchor b = anchor_of(p, t);
    const int G = min(max(p.gt_counts[n], 0), p.Gmax);
    const float* gts = p.gt_boxes + (long long)n * p.Gmax * 4;
    const float best = p.a_max[(long long)n * p.T + t];
    const int arg = p.a_arg[(long long)n * p.T + t];
    int label = -1;
    if (G > 0) {
      bool tie = false;
      for (int g = 0; g < G; ++g) {
        const float ov = iou(b, gts + g * 4);
        tie = tie || (__float_as_uint(ov) == p.g_max[(long long)n * p.Gmax + g]);
      }
      const int cls = p.gt_classes[(long long)n * p.Gmax + arg];
      if (tie) label = cls;
      if (best >= p.pos_thr) label = cls;
    }
    const bool fg = label >= 1;
    const bool bg = G > 0 ? best < p.neg_thr : true;
    if (bg) label = 0;
    if (fg) atomicAdd(&s_fg, 1);
    if (bg) atomicAdd(&s_bg, 1);
    p.label[(long long)n * p.T + t] = (signed char)label;
    if (b.y < p.h[b.level] && b.x < p.w[b.level])
      p.labels_out[b.level][(((long long)n * p.A + b.a) * p.h[b.level] + b.y) * p.w[b.level] + b.x] =
          label;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_fg) atomicAdd(p.img_counts + n * 2, s_fg);
    if (s_bg) atomicAdd(p.img_counts + n * 2 + 1, s_bg);
  }
}

// flat index of the fg list within a level: ((n * A + a) * fs + y) * fs + x
__device__ __forceinline__ bool fg_flag(const ALArgs& p, int l, long long f, int* n_out, int* t_out) {
  const int fs = p.fs[l];
  const long long per_img = (long long)p.A * fs * fs;
  if (f >= per_img * p.N) return false;
  const int n = (int)(f / per_img);
  const int t = p.start[l] + (int)(f - (long long)n * per_img);
  *n_out = n; *t_out = t;
  return p.label[(long long)n * p.T + t] > 0;
}

__global__ __launch_bounds__(kCB) void al_count_kernel(const ALArgs p) {
  __shared__ int s;
  if (threadIdx.x == 0) s = 0;
  __syncthreads();
  const int l = blockIdx.y;
  int n, t;
  const bool f = fg_flag(p, l, (long long)blockIdx.x * kCB + threadIdx.x, &n, &t);
  const unsigned long long m = __ballot(f);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s, __popcll(m));
  __syncthreads();
  if (threadIdx.x == 0) p.blk_counts[l * p.max_blocks + blockIdx.x] = s;
}

__global__ __launch_bounds__(kCB) void al_scan_kernel(const ALArgs p) {
  // one workgroup per level: exclusive scan of the block counts (serial over chunks of 1024)
  __shared__ int buf[kCB];
  __shared__ int carry;
  const int l = blockIdx.x;
  const int fs = p.fs[l];
  const long long total = (long long)p.N * p.A * fs * fs;
  const int blocks = (int)((total + kCB - 1) / kCB);
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < blocks; base += kCB) {
    const int i = base + threadIdx.x;
    const int v = i < blocks ? p.blk_counts[l * p.max_blocks + i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < kCB; off <<= 1) {          // Hillis-Steele inclusive scan
      const int add = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
      __syncthreads();
      buf[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < blocks) p.blk_offsets[l * p.max_blocks + i] = carry + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == kCB - 1) carry += buf[kCB - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.counts_out[l] = carry;
}

__global__ __launch_bounds__(kCB) void al_scatter_kernel(const ALArgs p) {
  __shared__ int wsum[kCB / 64];
  const int l = blockIdx.y;
  int n = 0, t = 0;
  const bool f = fg_flag(p, l, (long long)blockIdx.x * kCB + threadIdx.x, &n, &t);
  const unsigned long long m = __ballot(f);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wsum[wv] = __popcll(m);
  __syncthreads();
  int before = 0;
  for (int i = 0; i < wv; ++i) before += wsum[i];
  if (!f) return;
  const int pos = p.blk_offsets[l * p.max_blocks + blockIdx.x] + before +
                  __popcll(m & ((1ull << lane) - 1ull));
  if (pos >= p.capacity) return;
  const Anchor b = anchor_of(p, t);
  float* loc = p.locs_out[l] + (long long)pos * 4;
  loc[0] = (float)n; loc[1] = (float)(4 * b.a); loc[2] = (float)b.y; loc[3] = (float)b.x;
  // boxes.py:193-224 with weights (1,1,1,1), float32
  const float* q = p.gt_boxes + ((long long)n * p.Gmax + p.a_arg[(long long)n * p.T + t]) * 4;
  const float ew = __fadd_rn(__fsub_rn(b.x2, b.x1), 1.0f), eh = __fadd_rn(__fsub_rn(b.y2, b.y1), 1.0f);
  const float ex = __fadd_rn(b.x1, __fmul_rn(0.5f, ew)), ey = __fadd_rn(b.y1, __fmul_rn(0.5f, eh));
  const float gw = __fadd_rn(__fsub_rn(q[2], q[0]), 1.0f), gh = __fadd_rn(__fsub_rn(q[3], q[1]), 1.0f);
  const float gx = __fadd_rn(q[0], __fmul_rn(0.5f, gw)), gy = __fadd_rn(q[1], __fmul_rn(0.5f, gh));
  float* tg = p.targets_out[l] + (long long)pos * 4;
  tg[0] = __fdiv_rn(__fsub_rn(gx, ex), ew);
  tg[1] = __fdiv_rn(__fsub_rn(gy, ey), eh);
  tg[2] = logf(__fdiv_rn(gw, ew));
  tg[3] = logf(__fdiv_rn(gh, eh));
}

__global__ void al_finalize_kernel(const ALArgs p) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float fg = 0.0f;       // retinanet.py:157: float32 accumulation
  double bg = 0.0;       // :158 / :300-304: float64, cast at the end
  for (int n = 0; n < p.N; ++n) {
    const int nf = p.img_counts[n * 2], nb = p.img_counts[n * 2 + 1];
    fg += (float)nf;
    bg += ((double)nb + 1.0) * (double)(p.num_classes - 1) + (double)(float)nf * (double)(p.num_classes - 2);
  }
  p.fg_bg_out[0] = fg;
  p.fg_bg_out[1] = (float)bg;
}

int plan(ALArgs* a, int levels, int A, int k_min, const int* fs, int N, int Gmax) {
  if (levels < 1 || levels > SSAD_MAX_LEVELS || A < 1 || N < 0 || Gmax < 0) return SSAD_E_BADARG;
  long long T = 0;
  long long maxb = 1;
  for (int l = 0; l < levels; ++l) {
    if (fs[l] < 1) return SSAD_E_BADARG;
    a->fs[l] = fs[l];
    a->start[l] = (int)T;
    T += (long long)A * fs[l] * fs[l];
    const long long b = ((long long)N * A * fs[l] * fs[l] + kCB - 1) / kCB;
    if (b > maxb) maxb = b;
  }
  if (T >= (1LL << 30) || maxb >= (1LL << 30)) return SSAD_E_BADARG;
  for (int l = levels; l <= SSAD_MAX_LEVELS; ++l) a->start[l] = (int)T;
  a->levels = levels; a->A = A; a->k_min = k_min; a->N = N; a->Gmax = Gmax;
  a->T = (int)T;
  a->max_blocks = (int)maxb;
  return 0;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

size_t ssad_retinanet_anchor_labels_workspace_bytes(int levels, int A, int k_min,
                                                    const int* field_sizes_host, int N,
                                                    int Gmax) {
  ALArgs a;
  if (plan(&a, levels, A, k_min, field_sizes_host, N, Gmax)) return 0;
  const size_t NT = (size_t)N * a.T;
  return align256(NT * 4) + align256(NT * 4) + align256((size_t)N * (Gmax > 0 ? Gmax : 1) * 4) +
         align256(NT) + align256((size_t)N * 2 * 4 + 4) +
         2 * align256((size_t)levels * a.max_blocks * 4);
}

int ssad_retinanet_anchor_labels(
    const double* cell_anchors, int levels, int A, int k_min, const int* field_sizes_host,
    const int* crop_h_host, const int* crop_w_host, const float* gt_boxes, const int* gt_classes,
    const int* gt_counts, int N, int Gmax, int num_classes, float positive_overlap,
    float negative_overlap, int* const* labels_out_host, float* const* locs_out_host,
    float* const* targets_out_host, int capacity, int* counts_out, float* fg_bg_out,
    void* workspace, size_t workspace_bytes, ssad_stream_t stream) {
  ALArgs a;
  const int rc = plan(&a, levels, A, k_min, field_sizes_host, N, Gmax);
  if (rc) return rc;
  if (!cell_anchors || !counts_out || !fg_bg_out || capacity < 0 || num_classes < 2 ||
      num_classes > 128)
    return SSAD_E_BADARG;
  if (N > 0 && (!gt_counts || (Gmax > 0 && (!gt_boxes || !gt_classes)))) return SSAD_E_BADARG;
  const size_t need = ssad_retinanet_anchor_labels_workspace_bytes(levels, A, k_min,
                                                                   field_sizes_host, N, Gmax);
  if (!workspace || workspace_bytes < need) return SSAD_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  a.cells = cell_anchors; a.gt_boxes = gt_boxes; a.gt_classes = gt_classes; a.gt_counts = gt_counts;
  a.pos_thr = positive_overlap; a.neg_thr = negative_overlap; a.num_classes = num_classes;
  for (int l = 0; l < SSAD_MAX_LEVELS; ++l) {
    a.h[l] = l < levels ? crop_h_host[l] : 0;
    a.w[l] = l < levels ? crop_w_host[l] : 0;
    a.labels_out[l] = l < levels ? labels_out_host[l] : nullptr;
    a.locs_out[l] = l < levels ? locs_out_host[l] : nullptr;
    a.targets_out[l] = l < levels ? targets_out_host[l] : nullptr;
    if (l < levels && (a.h[l] < 0 || a.w[l] < 0 || a.h[l] > a.fs[l] || a.w[l] > a.fs[l]))
      return SSAD_E_BADARG;
    if (l >= levels) a.fs[l] = 1;
  }
  a.capacity = capacity; a.counts_out = counts_out; a.fg_bg_out = fg_bg_out;
  char* w = (char*)workspace;
  const size_t NT = (size_t)N * a.T;
  a.a_max = (float*)w; w += align256(NT * 4);
  a.a_arg = (int*)w; w += align256(NT * 4);
  a.g_max = (unsigned*)w;
  const size_t gbytes = align256((size_t)N * (Gmax > 0 ? Gmax : 1) * 4);
  w += gbytes;
  a.label = (signed char*)w; w += align256(NT);
  a.img_counts = (int*)w;
  const size_t cbytes = align256((size_t)N * 2 * 4 + 4);
  w += cbytes;
  a.blk_counts = (int*)w; w += align256((size_t)levels * a.max_blocks * 4);
  a.blk_offsets = (int*)w;
  (void)hipMemsetAsync(a.g_max, 0, gbytes, s);
  (void)hipMemsetAsync(a.img_counts, 0, cbytes, s);
  if (N > 0) {
    const dim3 grid((a.T + kT - 1) / kT, N);
    hipLaunchKernelGGL(al_iou_max_kernel, grid, dim3(kT), 0, s, a);
    hipLaunchKernelGGL(al_assign_kernel, grid, dim3(kT), 0, s, a);
    hipLaunchKernelGGL(al_count_kernel, dim3(a.max_blocks, levels), dim3(kCB), 0, s, a);
  } else {
    (void)hipMemsetAsync(a.blk_counts, 0, (size_t)levels * a.max_blocks * 4, s);
  }
  hipLaunchKernelGGL(al_scan_kernel, dim3(levels), dim3(kCB), 0, s, a);
  if (N > 0) hipLaunchKernelGGL(al_scatter_kernel, dim3(a.max_blocks, levels), dim3(kCB), 0, s, a);
  hipLaunchKernelGGL(al_finalize_kernel, dim3(1), dim3(64), 0, s, a);
  return (int)hipGetLastError();
}

}  // extern "C"
