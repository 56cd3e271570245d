// detect.hip -- RetinaNet inference post-processing on the device (row f4, second
// half): what detectron/lib/core/test_retinanet.py:108-206 does in numpy for one
// image -- per level: scores above the threshold, the pre_nms_topn best, anchor
// decode (utils/boxes.py:150-190), rescale and clip (:132-147); then per-class
// greedy NMS (utils/cython_nms.pyx:37-92, suppression at IoU >= thresh, +1 box
// convention) and the dets_per_im best survivors, sorted by score.
//
// Pipeline (all on the caller's stream, no host round trip):
//   1. one 64-bit key per (level, anchor, class, y, x):
//        [63:61] level order | [60:29] score bits (0 when not a candidate) | [28:0] ~index
//      and, per level, the pre_nms_topn LARGEST keys: a radix select (eight 8-bit passes: histogram of the
//      digit among the keys that match the prefix found so far, then the digit in which the k-th largest
//      lies) gives the k-th largest key exactly (keys are unique), a compaction collects the keys >= it, a
//      rank sort orders those <= topn keys;
//   2. decode the <= levels * pre_nms_topn survivors to boxes;
//   3. order them by (class, score descending) -- rank sort;
//   4. 64 x 64 suppression bit-matrix for same-class pairs, then one workgroup per
//      class walks its segment in score order (the serial part of greedy NMS);
//   5. order the survivors by score (rank sort), emit the first dets_per_im.
// Score ties are broken by element index (the reference's argpartition / argsort
// leave them unspecified).  Every kernel is this file's: rounds 1-5 sorted with rocPRIM's radix sort through
// hipCUB (the whole 8.6 M-key array of an image for the top 1000 per level); the rank sort is O(n^2) comparisons
// on n = levels x topn <= 65 535 keys, exact and stable, a few microseconds at the reference's n = 5000.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ssad_kernels.h"

namespace {

constexpr int kT = 256;

struct DArgs {
  const float* prob[SSAD_MAX_LEVELS];     // [1][A*C][H][W]
  const float* delta[SSAD_MAX_LEVELS];    // [1][A*4][H][W]
  const double* cells;                    // [levels][A][4]
  int H[SSAD_MAX_LEVELS], W[SSAD_MAX_LEVELS];
  long long estart[SSAD_MAX_LEVELS + 1];  // first element of each level in the key array
  int levels, A, C, k_min;
  float th, th_last, nms_thresh, scale, xform_clip;
  int topn, dets_per_im, im_h, im_w;
  int n;                                  // levels * topn candidate slots
};

__device__ __forceinline__ int level_of(const DArgs& p, long long e) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < SSAD_MAX_LEVELS; ++i)
    if (i < p.levels && e >= p.estart[i]) l = i;
  return l;
}

__global__ __launch_bounds__(kT) void det_keys_kernel(const DArgs p, unsigned long long* keys) {
  const long long total = p.estart[p.levels];
  for (long long e = (long long)blockIdx.x * kT + threadIdx.x; e < total;
       e += (long long)gridDim.x * kT) {
    const int l = level_of(p, e);
    const long long idx = e - p.estart[l];
    const float s = p.prob[l][idx];
    const float th = l == p.levels - 1 ? p.th_last : p.th;
    const unsigned long long sb = s > th ? (unsigned long long)__float_as_uint(s) : 0ull;
    // descending sort: level 0 first, higher score first, lower index first
    keys[e] = ((unsigned long long)(7 - l) << 61) | (sb << 29) |
              (unsigned long long)((~(unsigned)idx) & 0x1fffffffu);
  }
}

// ---- top-k per level: radix select -------------------------------------------------------------------------------
struct SelState {
  unsigned long long prefix[SSAD_MAX_LEVELS];   // the bits of the k-th largest key found so far
  int want[SSAD_MAX_LEVELS];                    // its rank among the keys that match the prefix (1 = largest)
  int k[SSAD_MAX_LEVELS];                       // min(topn, elements of the level)
  int count[SSAD_MAX_LEVELS];                   // compaction cursor
};

__global__ void det_select_init_kernel(const DArgs p, SelState* st, unsigned* hist, unsigned long long* sel) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < SSAD_MAX_LEVELS) {
    const long long E = t < p.levels ? p.estart[t + 1] - p.estart[t] : 0;
    const int k = (int)(E < p.topn ? E : p.topn);
    st->prefix[t] = 0; st->want[t] = k; st->k[t] = k; st->count[t] = 0;
  }
  if (t < SSAD_MAX_LEVELS * 256) hist[t] = 0;
  for (int i = t; i < p.n; i += gridDim.x * blockDim.x) sel[i] = 0;   // slots past a level's k: key 0 = "not a candidate"
}

// histogram of digit (key >> shift) & 255 over the keys of level blockIdx.y whose bits above the digit equal the prefix
__global__ __launch_bounds__(kT) void det_select_hist_kernel(const DArgs p, const unsigned long long* keys,
                                                             const SelState* st, unsigned* hist, int shift) {
  __shared__ unsigned h[256];
  const int l = blockIdx.y;
  h[threadIdx.x] = 0;
  __syncthreads();
  const unsigned long long prefix = st->prefix[l];
  const long long lo = p.estart[l], hi = p.estart[l + 1];
  for (long long e = lo + (long long)blockIdx.x * kT + threadIdx.x; e < hi; e += (long long)gridDim.x * kT) {
    const unsigned long long key = keys[e];
    const bool match = shift >= 56 || ((key ^ prefix) >> (shift + 8)) == 0;
    if (match) atomicAdd(&h[(unsigned)(key >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[l * 256 + threadIdx.x], h[threadIdx.x]);
}

// the digit in which the wanted key lies; one thread per level
__global__ void det_select_pick_kernel(int levels, SelState* st, unsigned* hist, int shift) {
  const int l = threadIdx.x;
  if (l >= levels) return;
  int want = st->want[l];
  if (st->k[l] > 0) {
    for (int d = 255; d >= 0; --d) {
      const int c = (int)hist[l * 256 + d];
      if (want <= c) { st->prefix[l] |= (unsigned long long)d << shift; break; }
      want -= c;
    }
    st->want[l] = want;
  }
  for (int d = 0; d < 256; ++d) hist[l * 256 + d] = 0;
}

// the keys >= the k-th largest, in arrival order (exactly k of them: keys are unique)
__global__ __launch_bounds__(kT) void det_select_compact_kernel(const DArgs p, const unsigned long long* keys,
                                                                SelState* st, unsigned long long* sel) {
  const int l = blockIdx.y;
  if (st->k[l] == 0) return;
  const unsigned long long kth = st->prefix[l];
  const long long lo = p.estart[l], hi = p.estart[l + 1];
  for (long long e = lo + (long long)blockIdx.x * kT + threadIdx.x; e < hi; e += (long long)gridDim.x * kT) {
    const unsigned long long key = keys[e];
    if (key >= kth) {
      const int slot = atomicAdd(&st->count[l], 1);
      if (slot < p.topn) sel[(long long)l * p.topn + slot] = key;
    }
  }
}

// Rank sort of `segs` segments of seg_len keys each (blockIdx.y = segment): out[rank] = key, rank = the number of keys
// that precede it (descending: larger first; ascending: smaller first; equal keys in index order -- stable).  vals
// (may be null) travel with their keys.  O(seg_len^2) comparisons through LDS tiles.
__global__ __launch_bounds__(kT) void det_rank_sort_kernel(const unsigned long long* in, const int* vals_in, int seg_len,
                                                           int descending, unsigned long long* out, int* vals_out) {
  __shared__ unsigned long long tile[kT];
  const long long base = (long long)blockIdx.y * seg_len;
  const int i = blockIdx.x * kT + threadIdx.x;
  const unsigned long long mine = i < seg_len ? in[base + i] : 0;
  int rank = 0;
  for (int j0 = 0; j0 < seg_len; j0 += kT) {
    const int j = j0 + threadIdx.x;
    tile[threadIdx.x] = j < seg_len ? in[base + j] : 0;
    __syncthreads();
    const int m = seg_len - j0 < kT ? seg_len - j0 : kT;
    for (int u = 0; u < m; ++u) {
      const unsigned long long o = tile[u];
      const bool before = descending ? (o > mine) : (o < mine);
      rank += (before || (o == mine && j0 + u < i)) ? 1 : 0;
    }
    __syncthreads();
  }
  if (i < seg_len) {
    out[base + rank] = mine;
    if (vals_in) vals_out[base + rank] = vals_in[base + i];
  }
}

// slot = l * topn + r: the r-th best element of level l
__global__ __launch_bounds__(kT) void det_decode_kernel(const DArgs p,
                                                        const unsigned long long* keys,
                                                        float* boxes, float* scores,
                                                        unsigned long long* ckeys, int* cvals) {
  const int slot = blockIdx.x * kT + threadIdx.x;
  if (slot >= p.n) return;
  const int l = slot / p.topn;
  const unsigned long long key = keys[slot];       // the level's r-th largest key (0 past its last element)
  const unsigned sb = (unsigned)((key >> 29) & 0xffffffffull);
  cvals[slot] = slot;
  if (sb == 0) {                                   // not a candidate
    ckeys[slot] = ~0ull;                           // sorts last
    scores[slot] = 0.0f;
    return;
  }
  const int idx = (int)((~(unsigned)key) & 0x1fffffffu);
  const int H = p.H[l], W = p.W[l], HW = H * W;
  const int x = idx % W, y = (idx / W) % H;
  const int ac = idx / HW;                         // a * C + cls
  const int a = ac / p.C, cls = ac - a * p.C;
  const float stride = (float)(1 << (p.k_min + l));
  const double* c = p.cells + ((long long)l * p.A + a) * 4;
  // boxes = float32([x, y, x, y]) * stride; boxes += cell_anchor (float64 add, float32 store)
  const float bx = __fmul_rn((float)x, stride), by = __fmul_rn((float)y, stride);
  const float b0 = (float)((double)bx + c[0]), b1 = (float)((double)by + c[1]);
  const float b2 = (float)((double)bx + c[2]), b3 = (float)((double)by + c[3]);
  const float* d = p.delta[l] + (long long)a * 4 * HW + (long long)y * W + x;
  const float dx = d[0], dy = d[HW], dw = fminf(d[2 * HW], p.xform_clip),
              dh = fminf(d[3 * HW], p.xform_clip);
  const float w = __fadd_rn(__fsub_rn(b2, b0), 1.0f), h = __fadd_rn(__fsub_rn(b3, b1), 1.0f);
  const float cx = __fadd_rn(b0, __fmul_rn(0.5f, w)), cy = __fadd_rn(b1, __fmul_rn(0.5f, h));
  const float pcx = __fadd_rn(__fmul_rn(dx, w), cx), pcy = __fadd_rn(__fmul_rn(dy, h), cy);
  const float pw = __fmul_rn(expf(dw), w), ph = __fmul_rn(expf(dh), h);
  float o0 = __fsub_rn(pcx, __fmul_rn(0.5f, pw));
  float o1 = __fsub_rn(pcy, __fmul_rn(0.5f, ph));
  float o2 = __fsub_rn(__fadd_rn(pcx, __fmul_rn(0.5f, pw)), 1.0f);
  float o3 = __fsub_rn(__fadd_rn(pcy, __fmul_rn(0.5f, ph)), 1.0f);
  o0 = __fdiv_rn(o0, p.scale); o1 = __fdiv_rn(o1, p.scale);
  o2 = __fdiv_rn(o2, p.scale); o3 = __fdiv_rn(o3, p.scale);
  const float xm = (float)(p.im_w - 1), ym = (float)(p.im_h - 1);
  o0 = fmaxf(fminf(o0, xm), 0.0f); o1 = fmaxf(fminf(o1, ym), 0.0f);
  o2 = fmaxf(fminf(o2, xm), 0.0f); o3 = fmaxf(fminf(o3, ym), 0.0f);
  float* bo = boxes + (long long)slot * 4;
  bo[0] = o0; bo[1] = o1; bo[2] = o2; bo[3] = o3;
  scores[slot] = __uint_as_float(sb);
  // ascending sort: class, then score descending, then slot
  ckeys[slot] = ((unsigned long long)cls << 48) | ((unsigned long long)(~sb) << 16) |
                (unsigned long long)(slot & 0xffff);
}

// gather boxes / scores / classes into class-sorted order
__global__ __launch_bounds__(kT) void det_gather_kernel(int n, const unsigned long long* skeys,
                                                        const int* svals, const float* boxes,
                                                        const float* scores, float* sboxes,
                                                        float* sscores, int* scls) {
  const int i = blockIdx.x * kT + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = skeys[i];
  const int slot = svals[i];
  scls[i] = k == ~0ull ? -1 : (int)(k >> 48);
  sscores[i] = scores[slot];
#pragma unroll
  for (int j = 0; j < 4; ++j) sboxes[(long long)i * 4 + j] = boxes[(long long)slot * 4 + j];
}

// cython_nms.pyx:72-79, float32
__device__ __forceinline__ bool suppresses(const float* a, float aarea, const float* b, float barea,
                                           float thresh) {
  const float xx1 = fmaxf(a[0], b[0]), yy1 = fmaxf(a[1], b[1]);
  const float xx2 = fminf(a[2], b[2]), yy2 = fminf(a[3], b[3]);
  const float w = fmaxf(0.0f, __fadd_rn(__fsub_rn(xx2, xx1), 1.0f));
  const float h = fmaxf(0.0f, __fadd_rn(__fsub_rn(yy2, yy1), 1.0f));
  const float inter = __fmul_rn(w, h);
  const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(aarea, barea), inter));
  return ovr >= thresh;
}

__device__ __forceinline__ float box_area(const float* b) {
  return __fmul_rn(__fadd_rn(__fsub_rn(b[2], b[0]), 1.0f), __fadd_rn(__fsub_rn(b[3], b[1]), 1.0f));
}

// mask[i][bj] bit t: box i suppresses box bj*64+t (same class, later in score order)
__global__ __launch_bounds__(64) void det_nms_mask_kernel(int n, int words, const float* sboxes,
                                                          const int* scls, float thresh,
                                                          unsigned long long* mask) {
  __shared__ float jb[64][4];
  __shared__ int jc[64];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj < bi) return;
  const int t = threadIdx.x;
  const int j = bj * 64 + t;
  if (j < n) {
#pragma unroll
    for (int q = 0; q < 4; ++q) jb[t][q] = sboxes[(long long)j * 4 + q];
    jc[t] = scls[j];
  } else {
    jc[t] = -2;
  }
  __syncthreads();
  const int i = bi * 64 + t;
  if (i >= n) return;
  const int ci = scls[i];
  unsigned long long bits = 0;
  if (ci >= 0) {
    float ib[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ib[q] = sboxes[(long long)i * 4 + q];
    const float ia = box_area(ib);
    for (int u = 0; u < 64; ++u) {
      const int jj = bj * 64 + u;
      if (jj > i && jc[u] == ci && suppresses(ib, ia, jb[u], box_area(jb[u]), thresh))
        bits |= 1ull << u;
    }
  }
  mask[(long long)i * words + bj] = bits;
}

// one workgroup per class: greedy walk of the class segment in score order
__global__ __launch_bounds__(kT) void det_nms_scan_kernel(int n, int words, int C, const int* scls,
                                                          const float* sscores,
                                                          const unsigned long long* mask,
                                                          unsigned long long* fkeys) {
  extern __shared__ unsigned long long remv[];
  __shared__ int seg[2];
  const int cls = blockIdx.x;
  for (int w = threadIdx.x; w < words; w += kT) remv[w] = 0;
  if (threadIdx.x == 0) {
    // segment of this class in the class-sorted arrays (binary searches)
    int lo = 0, hi = n;
    while (lo < hi) { const int m = (lo + hi) >> 1; const int c = scls[m]; if (c >= 0 && c < cls) lo = m + 1; else hi = m; }
    seg[0] = lo;
    hi = n;
    while (lo < hi) { const int m = (lo + hi) >> 1; const int c = scls[m]; if (c >= 0 && c <= cls) lo = m + 1; else hi = m; }
    seg[1] = lo;
  }
  __syncthreads();
  const int s = seg[0], e = seg[1];
  for (int i = s; i < e; ++i) {
    const bool removed = (remv[i >> 6] >> (i & 63)) & 1ull;      // uniform read
    __syncthreads();
    if (!removed) {
      for (int w = threadIdx.x + (i >> 6); w < words; w += kT) remv[w] |= mask[(long long)i * words + w];
      if (threadIdx.x == 0)   // descending sort key: score, then earlier position first
        fkeys[i] = ((unsigned long long)__float_as_uint(sscores[i]) << 32) |
                   (unsigned long long)(~(unsigned)i);
    } else if (threadIdx.x == 0) {
      fkeys[i] = 0;
    }
    __syncthreads();
  }
  (void)C;
}

__global__ __launch_bounds__(kT) void det_emit_kernel(int n, int dets_per_im,
                                                      const unsigned long long* fsorted,
                                                      const float* sboxes, const int* scls,
                                                      float* out, int* count) {
  const int r = blockIdx.x * kT + threadIdx.x;
  if (r == 0) {
    int c = 0;
    while (c < dets_per_im && c < n && fsorted[c] != 0) ++c;
    *count = c;
  }
  if (r >= dets_per_im || r >= n) return;
  const unsigned long long k = fsorted[r];
  float* o = out + (long long)r * 6;
  if (k == 0) { for (int j = 0; j < 6; ++j) o[j] = 0.0f; return; }
  const int i = (int)(~(unsigned)k);
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = sboxes[(long long)i * 4 + j];
  o[4] = __uint_as_float((unsigned)(k >> 32));
  o[5] = (float)(scls[i] + 1);                  // class ids are 1-based in the output
}

size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct Plan {
  long long total;
  int n, words;
  size_t sort1, sort2, sort3;
};

int make_plan(int levels, int A, int C, const int* H, const int* W, int topn, Plan* p,
              long long* estart) {
  if (levels < 1 || levels > SSAD_MAX_LEVELS || A < 1 || C < 1 || C > 0x7fff || topn < 1)
    return SSAD_E_BADARG;
  long long t = 0;
  for (int l = 0; l < levels; ++l) {
    if (H[l] < 0 || W[l] < 0) return SSAD_E_BADARG;
    const long long e = (long long)A * C * H[l] * W[l];
    if (e >= (1LL << 29)) return SSAD_E_BADARG;      // 29 index bits in the key
    estart[l] = t;
    t += e;
  }
  for (int l = levels; l <= SSAD_MAX_LEVELS; ++l) estart[l] = t;
  if ((long long)levels * topn > 0xffff) return SSAD_E_BADARG;   // 16 slot bits in the class key
  p->total = t;
  p->n = levels * topn;
  p->words = (p->n + 63) / 64;
  // (sort1: the selection's state + histograms; the former radix-sort scratch sizes stay in the plan as zero)
  p->sort1 = al(sizeof(SelState)) + al(SSAD_MAX_LEVELS * 256 * sizeof(unsigned));
  p->sort2 = p->sort3 = 0;
  return 0;
}

}  // namespace

extern "C" {

size_t ssad_retinanet_detect_workspace_bytes(int levels, int A, int C, const int* H_host,
                                             const int* W_host, int pre_nms_topn) {
  Plan p;
  long long es[SSAD_MAX_LEVELS + 1];
  if (make_plan(levels, A, C, H_host, W_host, pre_nms_topn, &p, es)) return 0;
  size_t tmp = p.sort1 > p.sort2 ? p.sort1 : p.sort2;
  if (p.sort3 > tmp) tmp = p.sort3;
  const size_t n = (size_t)p.n;
  const size_t ks = (size_t)p.total > n ? (size_t)p.total : n;       // keys_s holds the levels' top-k (n slots) too
  return al((size_t)p.total * 8) + al(ks * 8) + al(tmp) + al(n * 16) + al(n * 4) + 2 * al(n * 8) +
         2 * al(n * 4) + al(n * 16) + al(n * 4) + al(n * 4) + al(n * (size_t)p.words * 8) +
         2 * al(n * 8);
}

int ssad_retinanet_detect(
    const float* const* cls_prob_host, const float* const* box_pred_host,
    const double* cell_anchors, int levels, int A, int C, int k_min, const int* H_host,
    const int* W_host, float inference_th, int pre_nms_topn, float nms_thresh, int dets_per_im,
    float im_scale, int im_height, int im_width, float bbox_xform_clip, float* dets_out,
    int* count_out, void* workspace, size_t workspace_bytes, ssad_stream_t stream) {
  Plan pl;
  DArgs a;
  const int rc = make_plan(levels, A, C, H_host, W_host, pre_nms_topn, &pl, a.estart);
  if (rc) return rc;
  if (!cls_prob_host || !box_pred_host || !cell_anchors || !dets_out || !count_out ||
      dets_per_im < 1 || dets_per_im > pl.n || !(im_scale > 0.0f))
    return SSAD_E_BADARG;
  const size_t need = ssad_retinanet_detect_workspace_bytes(levels, A, C, H_host, W_host,
                                                            pre_nms_topn);
  if (!workspace || workspace_bytes < need) return SSAD_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  for (int l = 0; l < SSAD_MAX_LEVELS; ++l) {
    a.prob[l] = l < levels ? cls_prob_host[l] : nullptr;
    a.delta[l] = l < levels ? box_pred_host[l] : nullptr;
    a.H[l] = l < levels ? H_host[l] : 0;
    a.W[l] = l < levels ? W_host[l] : 0;
  }
  a.cells = cell_anchors;
  a.levels = levels; a.A = A; a.C = C; a.k_min = k_min;
  a.th = inference_th; a.th_last = 0.0f;        // test_retinanet.py:136: level k_max uses 0.0
  a.nms_thresh = nms_thresh; a.scale = im_scale; a.xform_clip = bbox_xform_clip;
  a.topn = pre_nms_topn; a.dets_per_im = dets_per_im; a.im_h = im_height; a.im_w = im_width;
  a.n = pl.n;
  const size_t n = (size_t)pl.n;
  char* w = (char*)workspace;
  auto take = [&](size_t b) { char* r = w; w += al(b); return r; };
  unsigned long long* keys = (unsigned long long*)take((size_t)pl.total * 8);
  unsigned long long* keys_s = (unsigned long long*)take(((size_t)pl.total > n ? (size_t)pl.total : n) * 8);
  size_t tmp_bytes = pl.sort1 > pl.sort2 ? pl.sort1 : pl.sort2;
  if (pl.sort3 > tmp_bytes) tmp_bytes = pl.sort3;
  void* tmp = take(tmp_bytes);
  float* boxes = (float*)take(n * 16);
  float* scores = (float*)take(n * 4);
  unsigned long long* ckeys = (unsigned long long*)take(n * 8);
  unsigned long long* ckeys_s = (unsigned long long*)take(n * 8);
  int* cvals = (int*)take(n * 4);
  int* cvals_s = (int*)take(n * 4);
  float* sboxes = (float*)take(n * 16);
  float* sscores = (float*)take(n * 4);
  int* scls = (int*)take(n * 4);
  unsigned long long* mask = (unsigned long long*)take(n * (size_t)pl.words * 8);
  unsigned long long* fkeys = (unsigned long long*)take(n * 8);
  unsigned long long* fkeys_s = (unsigned long long*)take(n * 8);

  if (pl.total > 0) {
    long long blocks = (pl.total + kT - 1) / kT;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(det_keys_kernel, dim3((unsigned)blocks), dim3(kT), 0, s, a, keys);
  }
  // top-k per level: keys_s[l * topn + r] = the level's r-th largest key
  {
    SelState* st = (SelState*)tmp;
    unsigned* hist = (unsigned*)((char*)tmp + al(sizeof(SelState)));
    unsigned long long* sel = fkeys;             // scratch until the NMS scan claims it (stream order)
    hipLaunchKernelGGL(det_select_init_kernel, dim3(8), dim3(kT), 0, s, a, st, hist, sel);
    if (pl.total > 0) {
      long long per = (pl.total / levels + kT - 1) / kT;
      if (per > 512) per = 512;
      if (per < 1) per = 1;
      for (int shift = 56; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(det_select_hist_kernel, dim3((unsigned)per, (unsigned)levels), dim3(kT), 0, s, a, keys, st, hist, shift);
        hipLaunchKernelGGL(det_select_pick_kernel, dim3(1), dim3(64), 0, s, levels, st, hist, shift);
      }
      hipLaunchKernelGGL(det_select_compact_kernel, dim3((unsigned)per, (unsigned)levels), dim3(kT), 0, s, a, keys, st, sel);
    }
    hipLaunchKernelGGL(det_rank_sort_kernel, dim3((unsigned)((pre_nms_topn + kT - 1) / kT), (unsigned)levels), dim3(kT), 0, s,
                       sel, (const int*)nullptr, pre_nms_topn, 1, keys_s, (int*)nullptr);
  }
  const int nb = (pl.n + kT - 1) / kT;
  hipLaunchKernelGGL(det_decode_kernel, dim3(nb), dim3(kT), 0, s, a, keys_s, boxes, scores, ckeys,
                     cvals);
  hipLaunchKernelGGL(det_rank_sort_kernel, dim3((unsigned)nb, 1), dim3(kT), 0, s, ckeys, cvals, pl.n, 0, ckeys_s, cvals_s);
  hipLaunchKernelGGL(det_gather_kernel, dim3(nb), dim3(kT), 0, s, pl.n, ckeys_s, cvals_s, boxes,
                     scores, sboxes, sscores, scls);
  hipLaunchKernelGGL(det_nms_mask_kernel, dim3(pl.words, pl.words), dim3(64), 0, s, pl.n, pl.words,
                     sboxes, scls, nms_thresh, mask);
  (void)hipMemsetAsync(fkeys, 0, n * 8, s);
  hipLaunchKernelGGL(det_nms_scan_kernel, dim3(C), dim3(kT), (size_t)pl.words * 8, s, pl.n, pl.words,
                     C, scls, sscores, mask, fkeys);
  hipLaunchKernelGGL(det_rank_sort_kernel, dim3((unsigned)nb, 1), dim3(kT), 0, s, fkeys, (const int*)nullptr, pl.n, 1, fkeys_s,
                     (int*)nullptr);
  hipLaunchKernelGGL(det_emit_kernel, dim3((dets_per_im + kT - 1) / kT), dim3(kT), 0, s, pl.n,
                     dets_per_im, fkeys_s, sboxes, scls, dets_out, count_out);
  return (int)hipGetLastError();
}

}  // extern "C"
