// The native step driver (include/ssad_program.h): walks an array of op records and calls the
// raw launchers of ssad_kernels.h -- the counterpart of Caffe2's SimpleNet::Run
// (caffe2/caffe2/core/net_simple.cc), which walks a vector of operators on the net's stream.
// With a ssad_timing every op is bracketed by HIP events recorded on the launch stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "ssad_program.h"

struct ssad_timing {
  std::vector<hipEvent_t> pool;     // events, reused across resets
  size_t used = 0;
  struct Rec { int klass; double work; size_t e0, e1; };
  std::vector<Rec> recs;
  std::vector<int> only;            // empty: every op is bracketed; else only ops of these classes
  bool wants(int klass) const {
    if (only.empty()) return true;
    for (int k : only)
      if (k == klass) return true;
    return false;
  }
  hipEvent_t get() {
    if (used == pool.size()) {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      pool.push_back(e);
    }
    return pool[used++];
  }
};

namespace {

// auxiliary streams of the executor (one set per device) and a ring of sync events.  An event may
// be re-recorded once every consumer has been enqueued behind it; the ring is far longer than the
// number of fork / join points of a step, and hipStreamWaitEvent captures the record that was
// current when it was called.
struct AuxStreams {
  hipStream_t s[SSAD_MAX_AUX_STREAMS] = {nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> ring;
  size_t next = 0;
  hipEvent_t event() {
    if (ring.size() < 1024) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
      ring.push_back(e);
      return e;
    }
    hipEvent_t e = ring[next];
    next = (next + 1) % ring.size();
    return e;
  }
};

// One set per device, created on first use.  The map is shared by every thread that runs a
// program (one per stream in the step), hence the lock; a set whose streams could not all be
// created is destroyed again, not leaked or half-registered.
AuxStreams* aux_streams(hipError_t* why) {
  static std::map<int, AuxStreams*> per_device;
  static std::mutex lock;
  int dev = 0;
  hipError_t err = hipGetDevice(&dev);
  if (err != hipSuccess) { *why = err; return nullptr; }
  std::lock_guard<std::mutex> guard(lock);
  auto it = per_device.find(dev);
  if (it != per_device.end()) return it->second;
  AuxStreams* a = new AuxStreams();
  for (int k = 0; k < SSAD_MAX_AUX_STREAMS; ++k) {
    // (Round 6 measured the step with these streams confined to n compute units, hipExtStreamCreateWithCUMask:
    // 86.0 ms free-for-all, 90.3 at 192 CUs, 95.3 at 128, 109.5 at 96 -- profiles/r06_experiments.md.  The filter
    // gradients do not lose to sharing, they live off it; the switch is gone again.)
    err = hipStreamCreateWithFlags(&a->s[k], hipStreamNonBlocking);
    if (err != hipSuccess) {
      for (int j = 0; j < k; ++j) (void)hipStreamDestroy(a->s[j]);
      delete a;
      *why = err;
      return nullptr;
    }
  }
  per_device[dev] = a;
  return a;
}

int run_op(const ssad_op& o, ssad_stream_t s) {
  const void* const* p = o.p;
  const int32_t* i = o.i;
  const float* f = o.f;
  switch (o.code) {
    case SSAD_OP_WINO_PACK_FILTERS:
      return (i[1] == 3 ? ssad_conv_split_pack_filters : i[1] == 2 ? ssad_conv_wino24_pack_filters
                                                                   : ssad_conv_wino_pack_filters)((const ssad_pack_entry*)p[0], i[0], s);
    case SSAD_OP_PACK_FILTER:
      return ssad_conv_pack_filter((const float*)p[0], i[0], i[1], (float*)p[1], (float*)p[2], s);
    case SSAD_OP_CONV3X3:
      if (i[4] == 3)
        return ssad_conv3x3_forward_split((const ssad_conv_level*)p[0], i[0], (const float*)p[1], (const float*)p[2], i[1],
                                          i[2], i[3], (void*)p[3], (size_t)o.l[0], (const unsigned*)p[4], (unsigned*)p[5], s);
      return (i[4] == 2 ? ssad_conv3x3_forward_wino24 : i[4] ? ssad_conv3x3_forward_wino : ssad_conv3x3_forward)(
          (const ssad_conv_level*)p[0], i[0], (const float*)p[1], (const float*)p[2], i[1], i[2], i[3], s);
    case SSAD_OP_CONV3X3_WGRAD:
      if (i[4] == 1)
        return ssad_conv3x3_wgrad_split_amax((const ssad_conv_level*)p[0], i[0], (float*)p[1], (float*)p[2], i[1], i[2],
                                             i[3], (void*)p[3], (size_t)o.l[0], (const unsigned*)p[4],
                                             (const unsigned*)p[5], s);
      return ssad_conv3x3_wgrad((const ssad_conv_level*)p[0], i[0], (float*)p[1], (float*)p[2], i[1], i[2],
                                i[3], (void*)p[3], (size_t)o.l[0], s);
    case SSAD_OP_POW_SUM:
      return ssad_pow_sum_prezeroed((const float* const*)p[0], (const int64_t*)p[1], i[0], f[0], (float*)p[2],
                          (void*)p[3], (size_t)o.l[0], s);
    case SSAD_OP_CLS_LOSSES_FUSED:
      return ssad_cls_losses_fused_prezeroed((const ssad_distill_level*)p[0], i[0], (const float*)p[1],
                                   (const float*)p[2], (const ssad_distill_params*)p[3],
                                   (const ssad_focal_params*)p[4], (float*)p[5], (float*)p[6],
                                   (void*)p[7], (size_t)o.l[0], s);
    case SSAD_OP_DISTILL_FWD:
      return ssad_distill_loss_forward((const ssad_distill_level*)p[0], i[0], (const float*)p[1],
                                       (const ssad_distill_params*)p[2], (void*)p[3], (size_t)o.l[0], s);
    case SSAD_OP_DISTILL_BWD:
      return ssad_distill_loss_backward((const ssad_distill_level*)p[0], i[0], (const float*)p[1],
                                        (const float*)p[2], i[1], (const ssad_distill_params*)p[3], s);
    case SSAD_OP_FOCAL_FWD:
      return ssad_focal_loss_forward((const ssad_distill_level*)p[0], i[0], (const float*)p[1],
                                     (const ssad_focal_params*)p[2], (void*)p[3], (size_t)o.l[0], s);
    case SSAD_OP_FOCAL_BWD:
      return ssad_focal_loss_backward((const ssad_distill_level*)p[0], i[0], (const float*)p[1],
                                      (const float*)p[2], i[1], (const ssad_focal_params*)p[3], s);
    case SSAD_OP_SMOOTH_L1:
      return ssad_select_smooth_l1_levels((const ssad_smooth_l1_level*)p[0], i[0], (const float*)p[1],
                                          (const float*)p[2], f[0], f[1], i[1], (void*)p[3],
                                          (size_t)o.l[0], s);
    case SSAD_OP_SGD_FLAT:
      return ssad_momentum_sgd_flat((float*)p[0], (float*)p[1], (float*)p[2], (const float*)p[3], f[0], f[1],
                                    (const ssad_sgd_segment*)p[4], i[0], (const int*)p[5], s);
    case SSAD_OP_FILL:
      return ssad_fill((float*)p[0], f[0], o.l[0], s);
    case SSAD_OP_SCALE:
      return ssad_scale((const float*)p[0], (float*)p[1], f[0], o.l[0], s);
    case SSAD_OP_SUM_N:
      return ssad_sum_n((const float* const*)p[0], i[0], (float*)p[1], o.l[0], s);
    case SSAD_OP_CHECK_FINITE:
      return ssad_check_finite((const float*)p[0], o.l[0], (int*)p[1], s);
    case SSAD_OP_LOSS_SCALE_UPDATE:
      return ssad_loss_scale_update((float*)p[0], (int*)p[1], f[0], f[1], i[0], f[2], f[3], s);
    case SSAD_OP_F16_PACK_ACT:
      return ssad_f16_pack_activations_dyn((const float*)p[0], i[0], i[1], i[2], i[3], f[0],
                                           (const float*)p[1], (void*)p[2], s);
    case SSAD_OP_F16_UNPACK_ACT:
      return ssad_f16_unpack_activations_dyn(p[0], i[0], i[1], i[2], i[3], f[0], (const float*)p[1],
                                             (float*)p[2], s);
    case SSAD_OP_F16_PACK_FILTER:
      return ssad_f16_pack_filter((const float*)p[0], i[0], i[1], (void*)p[1], (void*)p[2], s);
    case SSAD_OP_F16_CONV3X3:
      return ssad_conv3x3_forward_f16_levels((const ssad_f16_level*)p[0], i[0], p[1], (const float*)p[2],
                                             i[1], i[2], i[3], s);
    case SSAD_OP_F16_WGRAD:
      return ssad_conv3x3_wgrad_f16_levels_dyn((const ssad_f16_wgrad_level*)p[0], i[0], i[1], i[2], i[3],
                                               f[0], (const float*)p[1], (float*)p[2], (float*)p[3],
                                               (void*)p[4], (size_t)o.l[0], s);
    case SSAD_OP_AFFINE_CHANNEL:
      return ssad_affine_channel((const float*)p[0], (const float*)p[1], (const float*)p[2],
                                 (const float*)p[3], (float*)p[4], i[0], i[1], i[2], i[3], s);
    case SSAD_OP_UPSAMPLE:
      return ssad_upsample_nearest((const float*)p[0], (const float*)p[1], (float*)p[2], i[0], i[1], i[2],
                                   i[3], i[4], s);
    case SSAD_OP_UPSAMPLE_GRAD:
      return ssad_upsample_nearest_grad((const float*)p[0], (float*)p[1], i[0], i[1], i[2], i[3], i[4], s);
    case SSAD_OP_STEM_POOL:
      return ssad_max_pool3x3s2_bias_relu((const float*)p[0], (const float*)p[1], i[0], i[1], i[2], i[3],
                                          i[4], (float*)p[2], s);
    case SSAD_OP_RELU_GRAD_ROWSUM:
      return ssad_relu_grad_rowsum((const float*)p[0], (const float*)p[1], (float*)p[2], (float*)p[3], i[0],
                                   i[1], i[2], s);
    case SSAD_OP_RELU_GRAD:
      return ssad_relu_grad((const float*)p[0], (const float*)p[1], (float*)p[2], o.l[0], s);
    case SSAD_OP_GEMM_CONV:
      return ssad_conv1x1_gemm((const ssad_gemm_conv*)p[0], s);
    case SSAD_OP_GEMM_CONV_SPLIT:
      return ssad_conv1x1_gemm_split_amax((const ssad_gemm_conv*)p[0], (const float*)p[2], (const unsigned*)p[3],
                                          (unsigned*)p[4], (void*)p[1], (size_t)o.l[0], s);
    case SSAD_OP_SPLIT_ABSMAX_LEVELS:
      return ssad_split_absmax_levels((const ssad_conv_level*)p[0], i[0], i[1], i[2], (unsigned*)p[1], s);
    case SSAD_OP_SPLIT_ABSMAX:
      return ssad_split_absmax((const float*)p[0], (long long)o.l[0], (unsigned*)p[1], s);
    case SSAD_OP_GEMM_SPLIT_PACK:
      return ssad_gemm_split_pack_filters((const ssad_gemm_pack_entry*)p[0], i[0], s);
    case SSAD_OP_CONV1X1_WGRAD:
      if (i[5] == 1)
        return ssad_conv1x1_wgrad_split_amax((const float*)p[0], (const float*)p[1], i[0], i[1], i[2], i[3],
                                             (float*)p[2], i[4], (void*)p[3], (size_t)o.l[0], (const unsigned*)p[4],
                                             (const unsigned*)p[5], s);
      return ssad_conv1x1_wgrad((const float*)p[0], (const float*)p[1], i[0], i[1], i[2], i[3], (float*)p[2],
                                i[4], (void*)p[3], (size_t)o.l[0], s);
    case SSAD_OP_TRANSPOSE_FILTER:
      return ssad_transpose_filter((const float*)p[0], i[0], i[1], i[2], (float*)p[1], s);
    case SSAD_OP_SUBSAMPLE:
      return ssad_subsample((const float*)p[0], i[0], i[1], i[2], i[3], i[4], (float*)p[1], s);
    case SSAD_OP_SUBSAMPLE_GRAD:
      return ssad_subsample_grad((const float*)p[0], i[0], i[1], i[2], i[3], i[4], i[5], (float*)p[1], s);
    case SSAD_OP_RELU:
      return ssad_relu((const float*)p[0], (float*)p[1], o.l[0], s);
    case SSAD_OP_IM2COL_BATCHED:
      return ssad_im2col_batched((const float*)p[0], i[0], i[1], i[2], i[3], i[4], i[5], i[6], (float*)p[1], s);
    case SSAD_OP_GROUPED_CONV3X3:
      return ssad_grouped_conv3x3_forward((const float*)p[0], (const float*)p[1], (const float*)p[2], i[0], i[1],
                                          i[2], i[3], i[4], i[5], i[6], (float*)p[3], s);
    case SSAD_OP_CONV_IMPLICIT:
      return ssad_conv_implicit_gemm((const ssad_gemm_conv*)p[0], i[0], i[1], i[2], i[3], i[4], i[5], s);
    case SSAD_OP_GROUPED_PACK:
      return ssad_grouped_conv3x3_pack_filter((const float*)p[0], i[0], i[1], (float*)p[1], s);
    case SSAD_OP_CHANNEL_SUM:
      return ssad_channel_sum((const float*)p[0], i[0], i[1], i[2], (float*)p[1], i[3], s);
    case SSAD_OP_PW_F16:
      return ssad_conv1x1_f16((const ssad_pw_f16*)p[0], s);
    case SSAD_OP_PW_F16_PACK:
      return ssad_pw_f16_pack_filter((const float*)p[0], i[0], i[1], (void*)p[1], (void*)p[2], s);
    case SSAD_OP_PW_F16_WGRAD:
      return ssad_conv1x1_wgrad_f16(p[0], p[1], i[0], i[1], i[2], i[3], i[4], i[5], f[0], (const float*)p[2],
                                    (float*)p[3], (float*)p[4], (void*)p[5], (size_t)o.l[0], s);
    case SSAD_OP_F16_EW:
      return ssad_f16_elementwise(i[0], p[0], p[1], (void*)p[2], i[1], i[2], i[3], i[4], i[5], i[6], s);
    case SSAD_OP_STEM_POOL_F16:
      return ssad_stem_pool_f16((const float*)p[0], (const float*)p[1], i[0], i[1], i[2], i[3], (void*)p[2], s);
    case SSAD_OP_GROUPED_F16:
      return ssad_grouped_conv3x3_f16(p[0], p[1], (const float*)p[2], i[0], i[1], i[2], i[3], i[4], i[5],
                                      (void*)p[3], s);
    case SSAD_OP_GROUPED_F16_PACK:
      return ssad_grouped_conv3x3_f16_pack_filter((const float*)p[0], i[0], i[1], (void*)p[1], s);
    case SSAD_OP_TRANSPOSE_FILTERS:
      return ssad_transpose_filters((const ssad_transpose_entry*)p[0], i[0], s);
    case SSAD_OP_F16_PACK_FILTERS:
      return ssad_f16_pack_filters((const ssad_f16_pack_entry*)p[0], i[0], s);
    case SSAD_OP_CONV_IMPLICIT_WS:
      return ssad_conv_implicit_gemm_ws((const ssad_gemm_conv*)p[0], i[0], i[1], i[2], i[3], i[4], i[5], (void*)p[1],
                                        (size_t)o.l[0], s);
    case SSAD_OP_CONV_KXK_WGRAD:
      return ssad_conv_kxk_wgrad((const float*)p[0], (const float*)p[1], i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7],
                                 (float*)p[2], (int)o.l[1], (void*)p[3], (size_t)o.l[0], s);
    case SSAD_OP_CONV_KXK_DGRAD:
      return ssad_conv_kxk_dgrad((const float*)p[0], (const float*)p[1], i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7],
                                 (float*)p[2], (const float*)p[4], (int)o.l[1], (void*)p[3], (size_t)o.l[0], s);
    default:
      return SSAD_E_BADARG;
  }
}

}  // namespace

extern "C" {

ssad_timing* ssad_timing_create(void) { return new ssad_timing(); }

void ssad_timing_destroy(ssad_timing* t) {
  if (!t) return;
  for (hipEvent_t e : t->pool) (void)hipEventDestroy(e);
  delete t;
}

int ssad_timing_select(ssad_timing* t, const int* klasses, int n) {
  if (!t || n < 0 || (n > 0 && !klasses)) return SSAD_E_BADARG;
  t->only.assign(klasses, klasses + n);
  return 0;
}

void ssad_timing_reset(ssad_timing* t) {
  if (!t) return;
  t->used = 0;
  t->recs.clear();
}

int ssad_timing_collect(ssad_timing* t, ssad_timing_class* out, int max_out) {
  if (!t) return SSAD_E_BADARG;
  std::map<int, ssad_timing_class> acc;
  for (const auto& r : t->recs) {
    float ms = 0.0f;
    const hipError_t err = hipEventElapsedTime(&ms, t->pool[r.e0], t->pool[r.e1]);
    if (err != hipSuccess) return -(int)err;
    ssad_timing_class& c = acc[r.klass];
    c.klass = r.klass;
    c.launches += 1;
    c.ms += (double)ms;
    c.work += r.work;
  }
  int n = 0;
  for (const auto& kv : acc) {
    if (out && n < max_out) out[n] = kv.second;
    ++n;
  }
  return n;
}

int ssad_program_run(const ssad_op* ops, int n_ops, ssad_stream_t stream, ssad_timing* timing,
                     int* failed_index) {
  if (n_ops < 0 || (n_ops > 0 && !ops)) return SSAD_E_BADARG;
  hipStream_t main_s = (hipStream_t)stream;
  AuxStreams* aux = nullptr;              // created on first use, per device
  bool dirty[SSAD_MAX_AUX_STREAMS + 1] = {false, false, false, false};
  // per stream: the event that closed the previous op (consecutive ops of a stream share an event)
  size_t prev_idx[SSAD_MAX_AUX_STREAMS + 1];
  bool have_prev[SSAD_MAX_AUX_STREAMS + 1] = {false, false, false, false};
  auto stream_of = [&](int k) -> hipStream_t { return k == 0 ? main_s : aux->s[k - 1]; };
  auto join = [&](int k) -> int {
    hipEvent_t e = aux->event();
    if (!e) return (int)hipErrorOutOfMemory;
    hipError_t err = hipEventRecord(e, aux->s[k - 1]);
    if (err == hipSuccess) err = hipStreamWaitEvent(main_s, e, 0);
    dirty[k] = false;
    have_prev[0] = false;                  // the main stream's next op starts after the wait
    return (int)err;
  };
  // Error path: whatever was already enqueued on an auxiliary stream is joined first, so that work
  // the caller enqueues on the main stream afterwards cannot race with it; the first error wins.
  auto fail = [&](int k, int rc) {
    if (failed_index) *failed_index = k;
    if (aux)
      for (int j = 1; j <= SSAD_MAX_AUX_STREAMS; ++j)
        if (dirty[j]) (void)join(j);
    return rc;
  };
  for (int k = 0; k < n_ops; ++k) {
    const ssad_op& o = ops[k];
    const int sid = (o.code == SSAD_OP_FORK || o.code == SSAD_OP_JOIN) ? o.i[0] : o.stream;
    if (sid < 0 || sid > SSAD_MAX_AUX_STREAMS) return fail(k, SSAD_E_BADARG);
    if (sid > 0 && !aux) {
      hipError_t why = hipSuccess;
      aux = aux_streams(&why);
      if (!aux) return fail(k, why != hipSuccess ? (int)why : SSAD_E_BADARG);
    }
    if (o.code == SSAD_OP_FORK) {
      if (sid == 0) return fail(k, SSAD_E_BADARG);
      hipEvent_t e = aux->event();
      if (!e) return fail(k, (int)hipErrorOutOfMemory);
      hipError_t err = hipEventRecord(e, main_s);
      if (err == hipSuccess) err = hipStreamWaitEvent(aux->s[sid - 1], e, 0);
      if (err != hipSuccess) return fail(k, (int)err);
      dirty[sid] = true;
      have_prev[sid] = false;
      continue;
    }
    if (o.code == SSAD_OP_JOIN) {
      if (sid == 0) return fail(k, SSAD_E_BADARG);
      if (dirty[sid]) {
        const int rc = join(sid);
        if (rc) return fail(k, rc);
      }
      continue;
    }
    hipStream_t hs = stream_of(sid);
    if (sid > 0) dirty[sid] = true;
    const bool timed = timing && timing->wants(o.klass);
    if (timing && !timed) have_prev[sid] = false;      // the next timed op of this stream needs its own start
    if (timed && !have_prev[sid]) {
      hipEvent_t e0 = timing->get();
      if (!e0) return fail(k, (int)hipErrorOutOfMemory);
      const hipError_t e = hipEventRecord(e0, hs);
      if (e != hipSuccess) return fail(k, (int)e);
      prev_idx[sid] = timing->used - 1;
      have_prev[sid] = true;
    }
    const int rc = run_op(o, (ssad_stream_t)hs);
    if (rc != 0) return fail(k, rc);
    if (timed) {
      hipEvent_t e1 = timing->get();
      if (!e1) return fail(k, (int)hipErrorOutOfMemory);
      const hipError_t e = hipEventRecord(e1, hs);
      if (e != hipSuccess) return fail(k, (int)e);
      timing->recs.push_back({o.klass, o.work, prev_idx[sid], timing->used - 1});
      prev_idx[sid] = timing->used - 1;
    }
  }
  for (int k = 1; k <= SSAD_MAX_AUX_STREAMS; ++k)
    if (dirty[k]) {
      const int rc = join(k);
      if (rc) return fail(n_ops - 1, rc);
    }
  return 0;
}

}  // extern "C"
