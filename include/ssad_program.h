/*
 * ssad_program.h -- the native step driver: a training iteration as a flat PROGRAM of kernel
 * launches that ONE C-ABI call enqueues on a HIP stream.
 *
 * The reference's analogue is Caffe2's C++ net executor (caffe2/caffe2/core/net_simple.cc: a
 * vector of operators run in order on the net's stream; net_dag.cc:181-270 for the DAG
 * form): Python builds the net once, C++ runs it every iteration.  Here the host (Python,
 * as in the reference) builds an array of `ssad_op` records once -- every pointer, shape
 * and scalar of every launch of the iteration, over buffers allocated once -- and each
 * iteration is `ssad_program_run(ops, n, stream, timing)`: no interpreter, no allocation,
 * no per-launch FFI crossing between the kernels, so the launching thread stays far ahead
 * of the GPU.  A data-parallel step is a few program SEGMENTS with the RCCL all-reduce of a
 * gradient bucket issued between them.  Because a segment only enqueues launches on
 * `stream`, it can also be captured into a hipGraph by the caller.
 *
 * An op record maps 1:1 onto a raw launcher of ssad_kernels.h (same arguments, same error
 * codes); the executor adds nothing to the arithmetic.
 *
 * Timing: with a `ssad_timing` the executor brackets every op with HIP events ON THE LAUNCH
 * STREAM and, after the caller has synchronised, reports per timing class (`klass`, chosen
 * by the program builder: one class per kernel family) the number of launches, their
 * summed duration and their summed algorithmic work -- bench.py's rooflines come from here.
 */
#ifndef SSAD_PROGRAM_H_
#define SSAD_PROGRAM_H_

#include <stddef.h>
#include <stdint.h>

#include "ssad_kernels.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Argument slots per op code: i[] ints, f[] floats, l[] 64-bit ints (sizes), p[] pointers
 * (device pointers, or pointers to HOST descriptor arrays that must outlive the program:
 * names ending in _host). */
enum ssad_opcode {
  /* ssad_conv_wino_pack_filters(p0 = entries_host, i0 = n); i1 == 2: ssad_conv_wino24_pack_filters (the F(2x4)
   * engine: forward and / or data-gradient packs); i1 == 3: ssad_conv_split_pack_filters (the split-operand engine) */
  SSAD_OP_WINO_PACK_FILTERS = 1,
  /* ssad_conv_pack_filter(p0 = w, i0 = Cout, i1 = Cin, p1 = packed_fwd, p2 = packed_dgrad) */
  SSAD_OP_PACK_FILTER = 2,
  /* ssad_conv3x3_forward[_wino](p0 = levels_host, i0 = n, p1 = packed, p2 = bias, i1 = Cout,
   * i2 = Cin, i3 = flags); i4: 0 the direct engine, 1 Winograd F(2x2, 3x3), 2 Winograd F(2x4, 3x3)
   * (ssad_conv3x3_forward_wino24), 3 the split-operand engine (ssad_conv3x3_forward_split: p3 = workspace, l0 =
   * workspace bytes, p4 = amax_in or NULL, p5 = amax_out or NULL) */
  SSAD_OP_CONV3X3 = 3,
  /* ssad_conv3x3_wgrad(p0 = levels_host, i0 = n, p1 = dW, p2 = db, i1 = Cout, i2 = Cin,
   * i3 = accumulate, p3 = workspace, l0 = workspace_bytes); i4 == 1: ssad_conv3x3_wgrad_split_amax, same arguments +
   * p4 = X's |max| words, p5 = dY's (both NULL: measured by the call) */
  SSAD_OP_CONV3X3_WGRAD = 4,
  /* ssad_pow_sum(p0 = inputs_host, p1 = sizes_host, i0 = n, f0 = power, p2 = out,
   * p3 = workspace, l0 = workspace_bytes) */
  SSAD_OP_POW_SUM = 5,
  /* ssad_cls_losses_fused(p0 = levels_host, i0 = n, p1 = normalizer, p2 = fg_num,
   * p3 = distill_params_host, p4 = focal_params_host, p5 = distill_losses, p6 = focal_losses,
   * p7 = workspace, l0 = workspace_bytes) */
  SSAD_OP_CLS_LOSSES_FUSED = 6,
  /* ssad_distill_loss_forward(p0 = levels_host, i0 = n, p1 = normalizer, p2 = params_host,
   * p3 = workspace, l0 = workspace_bytes) */
  SSAD_OP_DISTILL_FWD = 7,
  /* ssad_distill_loss_backward(p0 = levels_host, i0 = n, p1 = normalizer, p2 = dloss,
   * i1 = dloss_stride, p3 = params_host) */
  SSAD_OP_DISTILL_BWD = 8,
  /* ssad_focal_loss_forward(p0 = levels_host, i0 = n, p1 = fg_num, p2 = params_host,
   * p3 = workspace, l0 = workspace_bytes) */
  SSAD_OP_FOCAL_FWD = 9,
  /* ssad_focal_loss_backward(p0 = levels_host, i0 = n, p1 = fg_num, p2 = dloss,
   * i1 = dloss_stride, p3 = params_host) */
  SSAD_OP_FOCAL_BWD = 10,
  /* ssad_select_smooth_l1_levels(p0 = levels_host, i0 = n, p1 = S, p2 = dloss, f0 = beta,
   * f1 = scale, i1 = want_forward, p3 = workspace, l0 = workspace_bytes) */
  SSAD_OP_SMOOTH_L1 = 11,
  /* ssad_momentum_sgd_flat(p0 = w, p1 = g, p2 = m, p3 = lr, f0 = momentum, f1 = weight_decay,
   * p4 = segments_host, i0 = n_segments, p5 = skip_flag) */
  SSAD_OP_SGD_FLAT = 12,
  /* ssad_fill(p0 = y, f0 = value, l0 = n) */
  SSAD_OP_FILL = 13,
  /* ssad_scale(p0 = x, p1 = y, f0 = alpha, l0 = n) */
  SSAD_OP_SCALE = 14,
  /* ssad_sum_n(p0 = inputs_host, i0 = n_inputs, p1 = out, l0 = n) */
  SSAD_OP_SUM_N = 15,
  /* ssad_check_finite(p0 = x, l0 = n, p1 = flag) */
  SSAD_OP_CHECK_FINITE = 16,
  /* ssad_loss_scale_update(p0 = state, p1 = counters, f0 = growth, f1 = backoff,
   * i0 = growth_interval, f2 = min_scale, f3 = max_scale) */
  SSAD_OP_LOSS_SCALE_UPDATE = 17,

  /* fp16 storage path */
  /* ssad_f16_pack_activations_dyn(p0 = x, i0..i3 = N, C, H, W, f0 = scale, p1 = scale_dev,
   * p2 = x_blocked) */
  SSAD_OP_F16_PACK_ACT = 32,
  /* ssad_f16_unpack_activations_dyn(p0 = x_blocked, i0..i3 = N, C, H, W, f0 = scale,
   * p1 = scale_dev, p2 = x) */
  SSAD_OP_F16_UNPACK_ACT = 33,
  /* ssad_f16_pack_filter(p0 = w, i0 = M, i1 = C, p1 = packed_fwd, p2 = packed_dgrad) */
  SSAD_OP_F16_PACK_FILTER = 34,
  /* ssad_conv3x3_forward_f16_levels(p0 = levels_host, i0 = n, p1 = packed, p2 = bias, i1 = C,
   * i2 = M, i3 = flags) */
  SSAD_OP_F16_CONV3X3 = 35,
  /* ssad_conv3x3_wgrad_f16_levels_dyn(p0 = levels_host, i0 = n, i1 = C, i2 = M, i3 = accumulate,
   * f0 = scale, p1 = scale_dev, p2 = dw, p3 = db, p4 = workspace, l0 = workspace_bytes) */
  SSAD_OP_F16_WGRAD = 36,

  /* backbone (row f1) */
  /* ssad_affine_channel(p0 = x, p1 = scale, p2 = bias, p3 = residual, p4 = y, i0 = N, i1 = C,
   * i2 = HW, i3 = relu) */
  SSAD_OP_AFFINE_CHANNEL = 48,
  /* ssad_upsample_nearest(p0 = x, p1 = addend, p2 = y, i0..i3 = N, C, H, W, i4 = scale) */
  SSAD_OP_UPSAMPLE = 49,
  /* ssad_upsample_nearest_grad(p0 = dy, p1 = dx, i0..i3 = N, C, H, W, i4 = scale) */
  SSAD_OP_UPSAMPLE_GRAD = 50,
  /* ssad_max_pool3x3s2_bias_relu(p0 = x, p1 = bias, i0..i3 = N, C, H, W, i4 = relu, p2 = y) */
  SSAD_OP_STEM_POOL = 51,
  /* ssad_relu_grad_rowsum(p0 = y, p1 = dy, p2 = dx, p3 = rowsum, i0 = N, i1 = C, i2 = HW) */
  SSAD_OP_RELU_GRAD_ROWSUM = 52,
  /* ssad_relu_grad(p0 = y, p1 = dy, p2 = dx, l0 = n) */
  SSAD_OP_RELU_GRAD = 53,
  /* ssad_conv1x1_gemm(p0 = desc_host) */
  SSAD_OP_GEMM_CONV = 54,
  /* ssad_channel_sum(p0 = dy, i0 = N, i1 = C, i2 = HW, p1 = out, i3 = accumulate) */
  SSAD_OP_CHANNEL_SUM = 55,
  /* ssad_conv1x1_wgrad(p0 = x, p1 = dy, i0..i3 = N, C, P, M, p2 = dw, i4 = accumulate,
   * p3 = workspace, l0 = workspace_bytes); i5 == 1: ssad_conv1x1_wgrad_split_amax, same arguments + p4 = x's |max|
   * word, p5 = dy's (both NULL: measured by the call) */
  SSAD_OP_CONV1X1_WGRAD = 56,
  /* ssad_transpose_filter(p0 = w, i0 = M, i1 = K, i2 = ldm, p1 = wt) */
  SSAD_OP_TRANSPOSE_FILTER = 57,
  /* ssad_subsample(p0 = x, i0..i3 = N, C, H, W, i4 = stride, p1 = y) */
  SSAD_OP_SUBSAMPLE = 58,
  /* ssad_subsample_grad(p0 = dy, i0..i3 = N, C, H, W, i4 = stride, i5 = accumulate, p1 = dx) */
  SSAD_OP_SUBSAMPLE_GRAD = 59,
  /* ssad_relu(p0 = x, p1 = y, l0 = n) */
  SSAD_OP_RELU = 60,
  /* Concurrency inside a program (HIP streams + events, no kernel): independent launches -- the
   * filter gradients of a layer beside the data-gradient chain that continues below it -- run on an
   * auxiliary stream so that one kernel's last partial round of workgroups is filled by another's.
   *   FORK(i0 = k): aux stream k waits for everything enqueued so far on the main stream;
   *   JOIN(i0 = k): the main stream waits for everything enqueued so far on aux stream k.
   * ssad_program_run joins every auxiliary stream it used before it returns, so a program segment
   * is always complete on the main stream when the next one (or an RCCL all-reduce) is enqueued. */
  SSAD_OP_FORK = 62,
  SSAD_OP_JOIN = 63,
  /* ssad_im2col_batched(p0 = x, i0..i3 = N, C, H, W, i4 = kernel, i5 = stride, i6 = pad, p1 = col) */
  SSAD_OP_IM2COL_BATCHED = 61,
  /* ssad_grouped_conv3x3_forward(p0 = x, p1 = packed filter, p2 = bias, i0..i3 = N, C, H, W, i4 = group,
   * i5 = stride, i6 = relu, p3 = y) */
  SSAD_OP_GROUPED_CONV3X3 = 64,
  /* ssad_grouped_conv3x3_pack_filter(p0 = w, i0 = C, i1 = group, p1 = packed) */
  SSAD_OP_GROUPED_PACK = 65,
  /* ssad_conv_implicit_gemm(p0 = ssad_gemm_conv*, i0..i5 = C, H, W, kernel, stride, pad) */
  SSAD_OP_CONV_IMPLICIT = 66,
  /* the backbones in fp16 storage (ssad_kernels.h: gemm_f16.hip, grouped_f16.hip) */
  /* p0 = const ssad_pw_f16* (host) */
  SSAD_OP_PW_F16 = 67,
  /* i0 = M, i1 = C; p0 = w, p1 = packed_fwd, p2 = packed_dgrad */
  SSAD_OP_PW_F16_PACK = 68,
  /* i0..i5 = N, C, H, W, M, accumulate; f0 = scale; l0 = workspace bytes; p0 = x, p1 = dy, p2 = scale_dev,
     p3 = dw, p4 = db, p5 = workspace */
  SSAD_OP_PW_F16_WGRAD = 69,
  /* i0 = mode, i1..i4 = N, C, H, W (of y), i5 = stride, i6 = accumulate; p0 = a, p1 = b, p2 = y */
  SSAD_OP_F16_EW = 70,
  /* i0..i3 = N, C, H, W (of z); p0 = z, p1 = bias, p2 = y */
  SSAD_OP_STEM_POOL_F16 = 71,
  /* i0..i5 = N, C, H, W, group, relu; p0 = x, p1 = packed, p2 = bias, p3 = y */
  SSAD_OP_GROUPED_F16 = 72,
  /* i0 = C, i1 = group; p0 = w, p1 = packed */
  SSAD_OP_GROUPED_F16_PACK = 73,
  /* k x k / strided convolutions at their own size (FPN's P6 / P7: 3x3, stride 2), ssad_kernels.h */
  /* ssad_conv_implicit_gemm_ws(p0 = ssad_gemm_conv*, i0..i5 = C, H, W, kernel, stride, pad, p1 = workspace,
   * l0 = workspace bytes): forward with split-K */
  SSAD_OP_CONV_IMPLICIT_WS = 74,
  /* ssad_conv_kxk_wgrad(p0 = x, p1 = dy, i0..i3 = N, C, H, W, i4 = M, i5 = kernel, i6 = stride, i7 = pad, p2 = dw,
   * l1 = accumulate, p3 = workspace, l0 = workspace bytes) */
  SSAD_OP_CONV_KXK_WGRAD = 75,
  /* ssad_conv_kxk_dgrad(p0 = w, p1 = dy, i0..i7 as above, p2 = dx, p4 = mask or NULL, l1 = accumulate,
   * p3 = workspace, l0 = workspace bytes) */
  SSAD_OP_CONV_KXK_DGRAD = 76,
  /* ssad_transpose_filters(p0 = const ssad_transpose_entry* (host), i0 = n_entries) */
  SSAD_OP_TRANSPOSE_FILTERS = 77,
  /* p0 = const ssad_f16_pack_entry* (host table, kept alive by the caller), i0 = entries */
  SSAD_OP_F16_PACK_FILTERS = 78,
  /* ssad_conv1x1_gemm_split_amax(p0 = const ssad_gemm_conv* (host), p1 = workspace, l0 = workspace bytes, p2 = packed
   * filter or NULL, p3 = x's |max| word or NULL, p4 = word y's |max| is folded into, or NULL) */
  SSAD_OP_GEMM_CONV_SPLIT = 79,
  /* ssad_split_absmax_levels(p0 = levels_host, i0 = n, i1 = channels, i2 = field, p1 = words) */
  SSAD_OP_SPLIT_ABSMAX_LEVELS = 80,
  /* ssad_split_absmax(p0 = x, l0 = elements, p1 = word) */
  SSAD_OP_SPLIT_ABSMAX = 81,
  /* ssad_gemm_split_pack_filters(p0 = const ssad_gemm_pack_entry* (host), i0 = entries) */
  SSAD_OP_GEMM_SPLIT_PACK = 82
};

typedef struct ssad_op {
  int32_t code;        /* enum ssad_opcode */
  int32_t klass;       /* timing class chosen by the builder (>= 0) */
  int32_t stream;      /* 0 = the stream ssad_program_run was given; 1..SSAD_MAX_AUX_STREAMS = one
                          of the executor's auxiliary streams (see SSAD_OP_FORK / SSAD_OP_JOIN) */
  int32_t reserved;
  int32_t i[8];
  float f[4];
  int64_t l[2];
  const void* p[8];
  double work;         /* algorithmic work of this launch: FLOPs for MFMA-bound classes, bytes
                          for HBM-bound ones (SURVEY.md 8d per-unit figures x units) */
} ssad_op;

#define SSAD_MAX_AUX_STREAMS 3

typedef struct ssad_timing ssad_timing;     /* opaque: event pool + records */

typedef struct ssad_timing_class {
  int32_t klass;
  int32_t launches;
  double ms;           /* summed duration of the class's ops (HIP events on the launch stream) */
  double work;         /* summed ssad_op.work */
} ssad_timing_class;

SSAD_API ssad_timing* ssad_timing_create(void);
SSAD_API void ssad_timing_destroy(ssad_timing* t);
/* forget the records taken so far (events are kept for reuse) */
SSAD_API void ssad_timing_reset(ssad_timing* t);
/* Bracket only the ops of the listed timing classes (n = 0: every op again).  Events between all
 * ~500 ops of a step keep consecutive kernels from overlapping their tails (measured: 1 % of the
 * step; more when collectives are in flight), so a throughput run times only what it reports. */
SSAD_API int ssad_timing_select(ssad_timing* t, const int* klasses, int n);
/* The caller must have synchronised the stream(s).  Writes up to max_out classes (ascending
 * klass) and returns how many there are, or a negative hipError_t. */
SSAD_API int ssad_timing_collect(ssad_timing* t, ssad_timing_class* out, int max_out);

/* Enqueue ops[0..n_ops) on `stream` in order.  Returns 0, or the first failing launcher's code
 * with *failed_index (may be NULL) set to the op's index.  timing may be NULL. */
SSAD_API int ssad_program_run(const ssad_op* ops, int n_ops, ssad_stream_t stream, ssad_timing* timing,
                              int* failed_index);

#ifdef __cplusplus
}
#endif
#endif /* SSAD_PROGRAM_H_ */
