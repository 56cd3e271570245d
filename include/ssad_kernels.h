/*
 * ssad_kernels.h -- raw C-ABI launchers for the gfx950 hot-path kernels.
 *
 * This is layer (3) of the drop-in boundary: plain pointers,
 * sizes and a stream handle, no C++ / torch / Caffe2 types.  The operator
 * classes in csrc/ops/ (registered for HIPContext, see c2hip_capi.h) call
 * exactly these entry points from RunOnDevice(); a foreign host (the torch
 * bridge, a cgo/ctypes stub, the reference's own Operator<HIPContext> build)
 * can call them directly.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - every launcher is asynchronous on `stream` and returns 0 on success or
 *     the hipError_t of the failed call / a negative SSAD_E_* code;
 *   - tensors are dense NCHW fp32, labels int32, as the reference's
 *     operators require.
 *
 * Reference citations are relative to /root/reference/caffe2/.
 */
#ifndef SSAD_KERNELS_H_
#define SSAD_KERNELS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSAD_API __attribute__((visibility("default")))

#define SSAD_E_BADARG (-1)       /* inconsistent dims / unsupported geometry */
#define SSAD_E_WORKSPACE (-2)    /* workspace too small */
#define SSAD_MAX_LEVELS 8        /* FPN levels fused into one launch */
#define SSAD_MAX_CONV_PROBLEMS 24 /* (level, filter) problems per conv launch */
#define SSAD_MAX_POWSUM_INPUTS 8 /* inputs per PowSum launch (chunked above) */

typedef void* ssad_stream_t;

/* ---------------------------------------------------------------------- */
/* SigmoidAdaptiveDistillLoss                                              */
/* replaces modules/detectron/sigmoid_adaptive_distillation_loss_op.cu:    */
/*   28-67 + 108-141 (kernel, math::Sum, math::Scale) in ONE streaming     */
/*   pass + a fixed-order partial reduce (no full-size losses_ temp).      */
/* ---------------------------------------------------------------------- */

typedef struct {
  float gamma;         /* arg "gamma", default 1.0  (.h:33)  */
  float alpha;         /* arg "alpha", default 0.25 (.h:34)  */
  float beta;          /* arg "beta",  default 0    (.h:35)  */
  int num_classes;     /* arg "num_classes", default 80      */
  int ignored_label;   /* arg "ignored_label", default -1    */
  float scale;         /* arg "scale", default 1.0, >= 0     */
} ssad_distill_params;

/* One FPN level of logits N x D x H x W (D = A*num_classes), teacher probs
 * of the same shape, labels N x A x H x W. */
typedef struct {
  const float* logits;
  const float* teacher_prob;
  const int32_t* labels;
  float* out;          /* fwd: scalar loss;  bwd: dX (N x D x H x W) */
  int N, D, H, W;
} ssad_distill_level;

/* bytes of scratch the forward needs for `n_levels` levels */
SSAD_API size_t ssad_distill_loss_workspace_bytes(int n_levels);

/* loss[l] = scale * sum_i loss_i over level l, for all levels in one launch
 * (+ one tiny fixed-order finalize launch).  normalizer: device scalar
 * (input 3 of the op). */
SSAD_API int ssad_distill_loss_forward(
    const ssad_distill_level* levels_host, int n_levels,
    const float* normalizer, const ssad_distill_params* params_host,
    void* workspace, size_t workspace_bytes, ssad_stream_t stream);

/* dX = d(scale*loss)/d(logits) * dloss, /Np and *scale folded into the one
 * pass (.cu:69-105 + 161-168).  dloss[l]: device scalars, one per level
 * (dloss_stride = 0 to share one). */
SSAD_API int ssad_distill_loss_backward(
    const ssad_distill_level* levels_host, int n_levels,
    const float* normalizer, const float* dloss, int dloss_stride,
    const ssad_distill_params* params_host, ssad_stream_t stream);

/* ---------------------------------------------------------------------- */
/* The student's supervised losses                   */
/* SigmoidFocalLoss    modules/detectron/sigmoid_focal_loss_op.cu:26-172   */
/* SelectSmoothL1Loss  modules/detectron/select_smooth_l1_loss_op.cu:23-176*/
/* ---------------------------------------------------------------------- */

typedef struct {
  float gamma;        /* arg "gamma", default 1.0  */
  float alpha;        /* arg "alpha", default 0.25 */
  int num_classes;    /* arg "num_classes", default 80 */
  float scale;        /* arg "scale", default 1.0, >= 0 */
} ssad_focal_params;

/* Same level struct as the distillation loss; teacher_prob is unused (may be
 * NULL).  Labels: -1 ignore, 0 background, 1..num_classes foreground class.
 * fg_num: device scalar (input 2 of the op).  Workspace as
 * ssad_distill_loss_workspace_bytes. */
SSAD_API int ssad_focal_loss_forward(
    const ssad_distill_level* levels_host, int n_levels, const float* fg_num,
    const ssad_focal_params* params_host, void* workspace, size_t workspace_bytes,
    ssad_stream_t stream);
SSAD_API int ssad_focal_loss_backward(
    const ssad_distill_level* levels_host, int n_levels, const float* fg_num,
    const float* dloss, int dloss_stride, const ssad_focal_params* params_host,
    ssad_stream_t stream);

/* Fused classification losses for the training step: ONE pass over the
 * logits yields distill_losses[l], focal_losses[l] and levels[l].out = dX =
 * d(distill)/dx + d(focal)/dx with both loss gradients = 1.0 (what the
 * reference obtains from 2 forward ops, 2 gradient ops and an autograd Sum).
 * Requires focal gamma == 2 (RetinaNet).
 * ONE launch: each level's sums are finished by its last-arriving workgroup (arrival counters
 * in the workspace, which the launcher zeroes with a hipMemsetAsync of a few KB on `stream` before
 * the launch -- the caller's buffer needs no preparation, and a buffer whose counters an aborted
 * launch left non-zero heals).  Do not share a workspace between streams.
 * ssad_cls_losses_fused_prezeroed (also ssad_pow_sum_prezeroed): the same without the memset, for a
 * caller that zero-filled the buffer once after allocating it and only ever passes it to these two
 * entry points on one stream (every completed launch leaves the counters zero again); this is what
 * ssad_program_run uses for the workspaces it is given. */
SSAD_API size_t ssad_cls_losses_fused_workspace_bytes(int n_levels);
SSAD_API int ssad_cls_losses_fused(
    const ssad_distill_level* levels_host, int n_levels, const float* normalizer,
    const float* fg_num, const ssad_distill_params* distill_host,
    const ssad_focal_params* focal_host, float* distill_losses, float* focal_losses,
    void* workspace, size_t workspace_bytes, ssad_stream_t stream);
SSAD_API int ssad_cls_losses_fused_prezeroed(
    const ssad_distill_level* levels_host, int n_levels, const float* normalizer,
    const float* fg_num, const ssad_distill_params* distill_host,
    const ssad_focal_params* focal_host, float* distill_losses, float* focal_losses,
    void* workspace, size_t workspace_bytes, ssad_stream_t stream);

/* Y_hat N x D x H x W box predictions; Y M x 4 targets; L M x 4 float rows
 * (n, c, y, x) locating each foreground box; S device scalar (#fg).  List entries that fall
 * outside the prediction map contribute nothing (the reference reads out of bounds there).
 * workspace: ssad_select_smooth_l1_workspace_bytes(1) bytes, caller provided -- the launchers
 * allocate nothing, so a step built from them can be captured in a HIP graph. */
SSAD_API size_t ssad_select_smooth_l1_workspace_bytes(int n_levels);
SSAD_API int ssad_select_smooth_l1_forward(
    const float* Y_hat, const float* Y, const float* L, const float* S, int N, int D, int H,
    int W, int M, float beta, float scale, float* loss, void* workspace, size_t workspace_bytes,
    ssad_stream_t stream);
/* dY_hat must be zero-filled by the caller (the operator does it, as the
 * reference's math::Set); writes the M*4 non-zero entries. */
SSAD_API int ssad_select_smooth_l1_backward(
    const float* Y_hat, const float* Y, const float* L, const float* S, const float* dloss,
    int N, int D, int H, int W, int M, float beta, float scale, float* dY_hat,
    ssad_stream_t stream);
/* The training step's form: every FPN level in one launch each for the forward, its
 * fixed-order finalize, the zero fill of dY_hat and the gradient scatter (4 launches
 * instead of the reference's 5 x {kernel, Sum, Scale, Set, kernel, Scale}).
 * want_forward != 0: loss[l] written; dloss != NULL: dY_hat[l] = full gradient (zero filled
 * here).  M may be 0 for a level. */
typedef struct ssad_smooth_l1_level {
  const float* Y_hat;   /* N x D x H x W */
  const float* Y;       /* M x 4 */
  const float* L;       /* M x 4 */
  float* loss;          /* device scalar (forward) */
  float* dY_hat;        /* N x D x H x W (gradient), or NULL */
  int N, D, H, W, M;
} ssad_smooth_l1_level;
SSAD_API int ssad_select_smooth_l1_levels(
    const ssad_smooth_l1_level* levels_host, int n_levels, const float* S, const float* dloss,
    float beta, float scale, int want_forward, void* workspace, size_t workspace_bytes,
    ssad_stream_t stream);

/* ---------------------------------------------------------------------- */
/* PowSum  (modules/detectron/pow_sum_op.cu:26-43)                         */
/* ---------------------------------------------------------------------- */

SSAD_API size_t ssad_pow_sum_workspace_bytes(int n_inputs);

/* out[0] = sum_j sum_i powf(inputs[j][i], power); all inputs in one launch (the sum is finished by
 * the last-arriving workgroup; arrival counters as for ssad_cls_losses_fused: zeroed by the
 * launcher, or by the caller for the _prezeroed form). */
SSAD_API int ssad_pow_sum(
    const float* const* inputs_host, const int64_t* sizes_host, int n_inputs,
    float power, float* out, void* workspace, size_t workspace_bytes,
    ssad_stream_t stream);
SSAD_API int ssad_pow_sum_prezeroed(
    const float* const* inputs_host, const int64_t* sizes_host, int n_inputs,
    float power, float* out, void* workspace, size_t workspace_bytes,
    ssad_stream_t stream);

/* ---------------------------------------------------------------------- */
/* Elementwise ops on the path                                             */
/* ---------------------------------------------------------------------- */

/* Relu, in place allowed (caffe2/operators/relu_op.cu:22-27) */
SSAD_API int ssad_relu(const float* x, float* y, int64_t n, ssad_stream_t stream);
/* ReluGradient: dx = y > 0 ? dy : 0 (relu_op.cu:29-36), in place allowed */
SSAD_API int ssad_relu_grad(const float* y, const float* dy, float* dx,
                            int64_t n, ssad_stream_t stream);
/* Sigmoid (caffe2/operators/sigmoid_op.cu:25-29) */
SSAD_API int ssad_sigmoid(const float* x, float* y, int64_t n, ssad_stream_t stream);
/* out = sum_k in[k]  (the autograd Sum of shared-weight gradients,
 * caffe2/python/core.py:706-741); out may alias in[0] */
SSAD_API int ssad_sum_n(const float* const* inputs_host, int n_inputs,
                        float* out, int64_t n, ssad_stream_t stream);
/* y = alpha * x (math::Scale), in place allowed */
SSAD_API int ssad_scale(const float* x, float* y, float alpha, int64_t n,
                        ssad_stream_t stream);
/* out = sum_k w_k[0] * x_k (caffe2 WeightedSum; each w_k is a one-element
 * device blob); out may alias x_0; n_pairs <= 8 */
SSAD_API int ssad_weighted_sum(const float* const* xs_host, const float* const* ws_host,
                               int n_pairs, float* out, int64_t n, ssad_stream_t stream);
/* AffineChannel (caffe2/modules/detectron/affine_channel_op.cu:27-40) with the
 * residual Sum and Relu of a ResNet bottleneck folded in:
 * y[n][c][p] = act(x[n][c][p] * scale[c] + bias[c] + residual[n][c][p]);
 * scale, bias, residual may be NULL (1, 0, 0); relu != 0 clamps at 0; y may
 * alias x or residual.  NCHW with HW = H*W. */
SSAD_API int ssad_affine_channel(const float* x, const float* scale, const float* bias,
                                 const float* residual, float* y, int N, int C, int HW,
                                 int relu, ssad_stream_t stream);
/* UpsampleNearest (caffe2/modules/detectron/upsample_nearest_op.cu:62-151) of the FPN
 * top-down path; x: N x C x H x W -> y: N x C x (H*scale) x (W*scale), nearest
 * neighbour; addend (same shape as y, may be NULL, may alias y) folds the lateral Sum
 * of FPN.py:283-306 into the pass.  The gradient sums each scale x scale block. */
SSAD_API int ssad_upsample_nearest(const float* x, const float* addend, float* y, int N, int C,
                                   int H, int W, int scale, ssad_stream_t stream);
SSAD_API int ssad_upsample_nearest_grad(const float* dy, float* dx, int N, int C, int H, int W,
                                        int scale, ssad_stream_t stream);
/* y[i] = value (ConstantFill) */
SSAD_API int ssad_fill(float* y, float value, int64_t n, ssad_stream_t stream);
/* Fused parameter update (detectron/lib/modeling/optimizer.py:115-130 +
 * caffe2/sgd/momentum_sgd_op_gpu.cu:22-38): g' = is_bias ? 2g : g + wd*w;
 * m = lr*g' + mu*m; g = m; w -= m.  lr: device scalar. */
SSAD_API int ssad_momentum_sgd_update(
    float* w, float* g, float* m, const float* lr, float momentum,
    float weight_decay, int is_bias, int64_t n, ssad_stream_t stream);
/* The same update for a whole model in ONE launch: w, g, m are flat buffers holding every
 * parameter, `segments_host` lists the parameters (element offset, length, bias or weight).
 * skip_flag (device int, may be NULL): when non-zero at execution time nothing is updated
 * (the step is dropped: a mixed-precision gradient overflowed, see ssad_check_finite). */
#define SSAD_MAX_SGD_SEGMENTS 64
typedef struct ssad_sgd_segment {
  int64_t offset, n;
  int is_bias;
  /* optional per-row gradient factor (device pointer, n / row_len floats; NULL = none):
   * g' = row_scale[k / row_len] * g + wd * w.  A filter stored with a frozen AffineChannel
   * scale s folded in (W' = s W, detectron/lib/modeling/ResNet.py:270-283 + affine_channel_op.cc:
   * the scale is not a trained blob) follows the reference's update of W exactly when its
   * gradient rows are multiplied by s^2. */
  int row_len;
  const float* row_scale;
} ssad_sgd_segment;
SSAD_API int ssad_momentum_sgd_flat(
    float* w, float* g, float* m, const float* lr, float momentum, float weight_decay,
    const ssad_sgd_segment* segments_host, int n_segments, const int* skip_flag,
    ssad_stream_t stream);
/* flag[0] |= 1 when any of x[0..n) is Inf or NaN (flag is NOT cleared here) */
SSAD_API int ssad_check_finite(const float* x, int64_t n, int* flag, ssad_stream_t stream);
/* Dynamic loss scaling, entirely on the device (no host round trip in the step):
 * state = {scale, 1/scale} floats; counters = {overflow flag, good steps} ints.
 * overflow: scale *= backoff, good = 0; else ++good and, at `growth_interval`, scale *= growth.
 * scale stays within [min_scale, max_scale]; the flag is cleared for the next step. */
SSAD_API int ssad_loss_scale_update(float* state, int* counters, float growth, float backoff,
                                    int growth_interval, float min_scale, float max_scale,
                                    ssad_stream_t stream);

/* ---------------------------------------------------------------------- */
/* Conv 3x3 / stride 1 / pad 1, NCHW fp32, exact-fp32 MFMA                 */
/* replaces caffe2/operators/conv_op_cudnn.cc:567-617 (fwd) and            */
/* 1011-1058 (bwd bias / filter / data)                                    */
/* ---------------------------------------------------------------------- */

/* One FPN level sharing the same filter: X is N x Cin x H x W,
 * Y is N x Cout x H x W. */
typedef struct {
  const float* x;      /* fwd: input;   dgrad: dY;        wgrad: X        */
  float* y;            /* fwd: output;  dgrad: dX;        wgrad: unused   */
  const float* aux;    /* dgrad: forward output to mask by (ReluGradient
                          fused) or NULL;  wgrad: dY                       */
  int N, H, W;
  /* forward / dgrad only: this problem's own packed filter and bias, or NULL
   * to use the launch-wide ones.  Lets independent convolutions of equal
   * (Cout, Cin) -- the cls and bbox tower layer of the same depth, teacher and
   * student -- share ONE launch, which fills the last wave of workgroups. */
  const float* packed;
  const float* bias;
} ssad_conv_level;

/* floats in a packed filter for (M outputs, K input channels) */
SSAD_API size_t ssad_conv_packed_filter_floats(int M, int K);

/* Repack W[Cout][Cin][3][3] into the MFMA A-operand stream used by the
 * forward kernel (packed_fwd, Cout x Cin) and/or by the data-gradient kernel
 * (packed_dgrad: flipped + transposed, Cin x Cout).  Either may be NULL. */
SSAD_API int ssad_conv_pack_filter(
    const float* w, int Cout, int Cin, float* packed_fwd, float* packed_dgrad,
    ssad_stream_t stream);

#define SSAD_CONV_RELU 1      /* y = max(y, 0) in the epilogue            */
#define SSAD_CONV_MASK_AUX 2  /* y = aux > 0 ? y : 0 (fused ReluGradient) */
#define SSAD_CONV_SIGMOID 4   /* y = 1/(1+exp(-y)) (teacher cls_pred -> prob,
                                 caffe2/operators/sigmoid_op.cu:25-29 fused) */
#define SSAD_CONV_SPLIT_TAIL 8 /* Winograd F(2x2) engine only, a scheduling hint: split this launch's partial round
                                 even behind full rounds (ssad_conv_wino_split_tail setting 2 for one launch) -- for
                                 launches that have the chip to themselves, e.g. the subnets' forward pass */

/* y = conv3x3(x, packed) (+ bias) for every level in one launch
 * (n_levels <= SSAD_MAX_CONV_PROBLEMS).
 * Used for the forward (packed_fwd, Cout outputs, Cin inputs) and for the
 * data gradient (packed_dgrad, outputs = Cin, inputs = Cout, bias NULL). */
SSAD_API int ssad_conv3x3_forward(
    const ssad_conv_level* levels_host, int n_levels, const float* packed,
    const float* bias, int Cout, int Cin, int flags, ssad_stream_t stream);

/* Winograd F(2x2,3x3) engine for the same forward / data-gradient contract
 * (2.25x fewer multiplies; fp32 accuracy ~1e-6 relative).  The filter is
 * packed by its own routine; levels / flags as ssad_conv3x3_forward.  Meant
 * for Cout >= 128; the direct kernel serves narrow outputs. */
SSAD_API size_t ssad_conv_wino_filter_floats(int M, int K);
SSAD_API int ssad_conv_wino_pack_filter(
    const float* w, int Cout, int Cin, float* packed_fwd, float* packed_dgrad,
    ssad_stream_t stream);
SSAD_API int ssad_conv3x3_forward_wino(
    const ssad_conv_level* levels_host, int n_levels, const float* packed,
    const float* bias, int Cout, int Cin, int flags, ssad_stream_t stream);
/* Kernel launches one ssad_conv3x3_forward_wino call makes for these level shapes: 1, or 2 when some maps
 * are staged as 8 x 16-pixel patches and others as pairs of 8 x 8 sub-patches (host-side query, no device
 * work; profiling tools attribute hardware counters to calls by launch order). */
SSAD_API int ssad_conv3x3_forward_wino_launches(const ssad_conv_level* levels_host, int n_levels);
/* ... for these channel counts and flags, the split-tail launches included (see ssad_conv_wino_split_tail) */
SSAD_API int ssad_conv3x3_forward_wino_launches_for(const ssad_conv_level* levels_host, int n_levels, int Cout,
                                                    int Cin, int flags);
/* The split tail of the persistent Winograd kernel (round 5): the items of the last, partial round of the grid are
 * cut along the reduction into 2 / 4 / 8 units each, one per workgroup; the last-arriving unit of an item adds the
 * partial results in unit order (deterministic) and applies bias / ReLU / mask.  Setting 1 (default, or
 * SSAD_WINO_SPLIT_TAIL): only launches WITHOUT a full round are split (FPN's small levels, small batches: P6 at
 * bs 16 -36 %, BASELINE config 2's step -3.6 %); setting 2: also the partial round behind full rounds, as a second
 * launch (res4 256 -> 256 at 40 x 56 x 16: 2.19 rounds took 3, -11 % isolated; res5 -16 %; the single-stream step
 * -1.0 ms -- but +1.1 ms in the overlapped step, whose other streams already fill those tails, hence not the
 * default); 0: off.  on < 0 only queries; returns the previous setting.  Results with and without it agree to fp32
 * round-off (another summation order over the input channels), each is reproducible bit for bit. */
SSAD_API int ssad_conv_wino_split_tail(int on);
/* ssad_conv_wino_pack_filter for a whole table of filters in one launch (the training step
 * repacks every filter after each update: 20 student filters x {forward, data gradient}). */
#define SSAD_MAX_PACK_ENTRIES 32   /* per launch; longer tables are chunked */
typedef struct ssad_pack_entry {
  const float* w;          /* [Cout][Cin][3][3] */
  int Cout, Cin;
  float* packed_fwd;       /* or NULL */
  float* packed_dgrad;     /* or NULL */
} ssad_pack_entry;
SSAD_API int ssad_conv_wino_pack_filters(const ssad_pack_entry* entries_host, int n_entries,
                                         ssad_stream_t stream);

/* Winograd F(2x4, 3x3) engine (round 5; conv3x3_winograd24.hip): 3 multiplies per output where F(2x2) does 4; fp32
 * error 1.5-2.2e-6 of the output scale against a float64 convolution where F(2x2) has 0.8-1.9e-6.  First the frozen
 * teacher's engine (model_builder.py:373-411 builds the teacher in test mode), then the trained networks' forward
 * pass and data gradient as well.  Same level / flag contract as ssad_conv3x3_forward_wino (bias,
 * SSAD_CONV_RELU, SSAD_CONV_SIGMOID, SSAD_CONV_MASK_AUX with the data-gradient pack), its own filter pack (entries
 * with packed_fwd and / or packed_dgrad).  Meant for >= 128 outputs. */
SSAD_API size_t ssad_conv_wino24_filter_floats(int M, int K);
SSAD_API int ssad_conv_wino24_pack_filters(const ssad_pack_entry* entries_host, int n_entries, ssad_stream_t stream);
SSAD_API int ssad_conv3x3_forward_wino24(
    const ssad_conv_level* levels_host, int n_levels, const float* packed,
    const float* bias, int Cout, int Cin, int flags, ssad_stream_t stream);

/* fp32 convolution on the 2.5 PFLOP/s fp16 matrix pipes by operand splitting (round 6, conv3x3_split.hip).
 * gfx950's fp32 MFMA runs at 1/16 of the fp16 rate and there is no xf32 mode (MI355X_MICROARCH.md).  Each fp32
 * operand is written as  x * s = hi + lo  with hi = fp16(x s), lo = fp16(x s - hi): 22 significant bits, s = a power
 * of two per tensor chosen from the tensor's measured |max| so that hi never overflows; the direct-form product is
 * three v_mfma_f32_32x32x16_f16 per operand pair (hi hi + lo hi + hi lo, the 2^-22 lo lo term dropped) accumulated
 * in fp32, and the exact power-of-two scales are divided out in the epilogue.  Same operator contract as
 * ssad_conv3x3_forward_wino24 (conv_op_cudnn.cc:567-617 / :1040-1058: NCHW fp32 in and out, bias, SSAD_CONV_RELU,
 * SSAD_CONV_SIGMOID, SSAD_CONV_MASK_AUX with the data-gradient pack); error against a float64 convolution
 * ~3e-7 of the output scale (Winograd F(2x4) fp32: ~2e-6).  Elements below 2^-29 of a tensor's |max| keep fewer
 * than 22 bits (absolute error <= 2^-40 |max|); a tensor holding Inf / NaN is passed through unscaled.
 *   workspace: the split, channel-blocked copy of every level's input + one |max| word per level
 *   (ssad_conv3x3_split_workspace_bytes); the call = |max| pass + split pass + convolution, on `stream`.
 *   amax_in  (device, n_levels words, or NULL): the inputs' |max| as float bit patterns, when a producer already
 *            measured them -- the |max| pass is skipped;
 *   amax_out (device, n_levels words the caller has zeroed, or NULL): the kernel folds the |max| of each level's
 *            OUTPUT into word l (atomicMax), ready to be the next layer's amax_in. */
SSAD_API size_t ssad_conv_split_filter_floats(int M, int K);
SSAD_API int ssad_conv_split_pack_filters(const ssad_pack_entry* entries_host, int n_entries, ssad_stream_t stream);
SSAD_API size_t ssad_conv3x3_split_workspace_bytes(const ssad_conv_level* levels_host, int n_levels, int Cin);
SSAD_API int ssad_conv3x3_forward_split(
    const ssad_conv_level* levels_host, int n_levels, const float* packed,
    const float* bias, int Cout, int Cin, int flags, void* workspace, size_t workspace_bytes,
    const unsigned* amax_in, unsigned* amax_out, ssad_stream_t stream);

SSAD_API size_t ssad_conv3x3_wgrad_workspace_bytes(
    const ssad_conv_level* levels_host, int n_levels, int Cout, int Cin);

/* dW[Cout][Cin][3][3] = sum over levels, images, pixels (overwrites, beta=0
 * like conv_op_cudnn.cc:1037; accumulate != 0 adds to dW instead) and
 * db[Cout] = sum dY (db may be NULL).  Deterministic split-K reduction. */
SSAD_API int ssad_conv3x3_wgrad(
    const ssad_conv_level* levels_host, int n_levels, float* dW, float* db,
    int Cout, int Cin, int accumulate, void* workspace, size_t workspace_bytes,
    ssad_stream_t stream);

/* The same contract (dW, db, accumulate, deterministic reduction) on the split-operand engine: dY and X are scaled by a
 * power of two from their measured |max| and split on the fly into hi + lo fp16; hi.hi + lo.hi + hi.lo run on
 * v_mfma_f32_32x32x16_f16 into fp32 accumulators (22-bit products, direct form: no Winograd transform error).
 * The call = |max| pass + main kernel + slab reduction (+ the bias gradient) on `stream`. */
SSAD_API size_t ssad_conv3x3_wgrad_split_workspace_bytes(
    const ssad_conv_level* levels_host, int n_levels, int Cout, int Cin);
SSAD_API int ssad_conv3x3_wgrad_split(
    const ssad_conv_level* levels_host, int n_levels, float* dW, float* db,
    int Cout, int Cin, int accumulate, void* workspace, size_t workspace_bytes,
    ssad_stream_t stream);
/* ... with the |max| words of X and dY handed over (device, n_levels words each: word l = level l, 0 for a level
 * without pixels) by whoever measured them -- an earlier split-engine call on the same tensors or
 * ssad_split_absmax_levels: the call's own |max| pass and its memset are skipped.  Both or neither. */
SSAD_API int ssad_conv3x3_wgrad_split_amax(
    const ssad_conv_level* levels_host, int n_levels, float* dW, float* db,
    int Cout, int Cin, int accumulate, void* workspace, size_t workspace_bytes,
    const unsigned* x_amax, const unsigned* dy_amax, ssad_stream_t stream);

/* The split engines' |max| pass on its own: the |max| (as float bit patterns) of every tensor of a level table, folded
 * into caller-owned words with atomicMax -- word k = problem k; the caller zeroes the words (a program zeroes its whole
 * table of words with ONE fill per step).  field 0: levels_host[k].x, 1: .aux; channels = their channel count.
 * A tensor measured once serves every split-engine call that reads it (forward, data gradient, filter gradient). */
SSAD_API int ssad_split_absmax_levels(const ssad_conv_level* levels_host, int n, int channels, int field,
                                      unsigned* words, ssad_stream_t stream);
SSAD_API int ssad_split_absmax(const float* x, long long n, unsigned* word, ssad_stream_t stream);

/* ---------------------------------------------------------------------- */
/* RetinaNet anchor labelling on the device (row f4)                       */
/* ---------------------------------------------------------------------- */

/* Replaces the numpy labelling of detectron/lib/roi_data/retinanet.py:97-306
 * (+ data_utils.py:52-103, utils/cython_bbox.pyx:31-74, utils/boxes.py:193-224).
 *   cell_anchors     device double [levels][A][4]: the A anchors of one cell per level
 *                    (modeling/generate_anchors.py), octave-major then aspect ratio
 *   field_sizes_host anchors are laid on a field_size x field_size grid per level
 *                    (data_utils.py:71-75); crop_h/w_host: int(blob size / stride)
 *   gt_boxes         device float [N][Gmax][4] (scaled), gt_classes int [N][Gmax]
 *                    (1..num_classes-1), gt_counts int [N] valid entries per image
 * Outputs per level l (host arrays of device pointers):
 *   labels_out[l]    int32 [N][A][crop_h][crop_w]       retnet_cls_labels_fpn
 *   locs_out[l]      float [capacity][4] = image, 4*anchor, y, x   retnet_roi_fg_bbox_locs_fpn
 *   targets_out[l]   float [capacity][4]                retnet_roi_bbox_targets_fpn
 *   counts_out       device int [levels]: entries produced (M); entries beyond
 *                    `capacity` are dropped -- compare on the host
 *   fg_bg_out        device float [2] = retnet_fg_num, retnet_bg_num
 * The list order (image, anchor, y, x) and every tie rule follow the reference. */
SSAD_API size_t ssad_retinanet_anchor_labels_workspace_bytes(
    int levels, int A, int k_min, const int* field_sizes_host, int N, int Gmax);
SSAD_API int ssad_retinanet_anchor_labels(
    const double* cell_anchors, int levels, int A, int k_min, const int* field_sizes_host,
    const int* crop_h_host, const int* crop_w_host, const float* gt_boxes,
    const int* gt_classes, const int* gt_counts, int N, int Gmax, int num_classes,
    float positive_overlap, float negative_overlap, int* const* labels_out_host,
    float* const* locs_out_host, float* const* targets_out_host, int capacity,
    int* counts_out, float* fg_bg_out, void* workspace, size_t workspace_bytes,
    ssad_stream_t stream);

/* ---------------------------------------------------------------------- */
/* Default convolution engine helpers and MaxPool (backbone, row f1)        */
/* ---------------------------------------------------------------------- */

/* output extent of a convolution / pooling axis (conv_pool_op_base.h:520-560, no
 * legacy padding): floor((in + pad_a + pad_b - (dilation*(kernel-1)+1)) / stride) + 1;
 * -1 when the window does not fit */
SSAD_API int ssad_conv_out_size(int in, int kernel, int dilation, int pad_a, int pad_b, int stride);
/* math::Im2col / Col2im, NCHW (caffe2/utils/math_gpu.cu): one image x[C][H][W] <->
 * col[C*kh*kw][OH*OW]; Col2im is written in gather form (deterministic) */
SSAD_API int ssad_im2col(const float* x, int C, int H, int W, int kh, int kw, int dil_h, int dil_w,
                         int pad_t, int pad_l, int pad_b, int pad_r, int stride_h, int stride_w,
                         float* col, ssad_stream_t stream);
/* Im2col of a whole batch, square kernel / stride / pad: col[n][C*k*k][OH*OW] */
SSAD_API int ssad_im2col_batched(const float* x, int N, int C, int H, int W, int kernel, int stride,
                                 int pad, float* col, ssad_stream_t stream);
SSAD_API int ssad_col2im(const float* col, int C, int H, int W, int kh, int kw, int dil_h,
                         int dil_w, int pad_t, int pad_l, int pad_b, int pad_r, int stride_h,
                         int stride_w, float* x, ssad_stream_t stream);
/* out[c] (+)= sum_{n,p} dy[n][c][p] (the bias gradient, conv_op_impl.h:470-486) */
SSAD_API int ssad_channel_sum(const float* dy, int N, int C, int HW, float* out, int accumulate,
                              ssad_stream_t stream);
/* ReluGradient (caffe2/operators/relu_op.cu:44-53) of an NCHW tensor fused with the plane sums
 * a per-channel bias gradient needs: dx = y > 0 ? dy : 0 (y NULL: dx = dy, and dx may then be
 * NULL = sums only) and rowsum[n][c] = sum over H*W of dx; db[c] = sum_n rowsum[n][c]. */
SSAD_API int ssad_relu_grad_rowsum(const float* y, const float* dy, float* dx, float* rowsum, int N,
                                   int C, int HW, ssad_stream_t stream);
/* Pointwise convolution with the bottleneck tail in its epilogue (detectron/lib/modeling/
 * ResNet.py:176-197, :223-283: Conv(kernel=1) -> AffineChannel, folded into w / bias -> Sum with
 * the shortcut -> Relu):  y[n][m][p] = act(sum_c w[m][c] x[n][c][p] + bias[m] (+ residual[n][m][p])).
 * NCHW fp32, exact fp32 MFMA.  C % 32 == 0, M % 128 == 0, P % 4 == 0 (else SSAD_E_BADARG: use
 * the default engine).  bias / residual may be NULL. */
SSAD_API int ssad_conv1x1_bias_act(const float* x, const float* w, const float* bias,
                                   const float* residual, float* y, int N, int C, int P, int M,
                                   int relu, ssad_stream_t stream);
/* The same with the input channels split over two tensors, x [N][C1][P] and x2 [N][C2][P]
 * (C1 + C2 = 128, w [M][C1 + C2]): a bottleneck's last layer and its projection shortcut
 * (ResNet.py:199-213) as ONE product, y = act([W3 | Wproj] . [y2 ; x] + bias) -- the
 * projection's output is never written. */
SSAD_API int ssad_conv1x1_bias_act2(const float* x, int C1, const float* x2, int C2, const float* w,
                                    const float* bias, const float* residual, float* y, int N, int P,
                                    int M, int relu, ssad_stream_t stream);
/* Pointwise (1x1) convolution as an exact-fp32 MFMA GEMM over the whole batch (row f1: the
 * bottleneck's 1x1 layers, projection shortcuts, FPN laterals; detectron/lib/modeling/
 * ResNet.py:221-283, FPN.py:116-250; algorithm caffe2/operators/conv_op_impl.h:126-173 with the
 * identity im2col of a 1x1 kernel):
 *     y[n][m][p] = act( sum_k a[k][m] x[n][k][p] + bias[m] + residual[n][m][p] )
 * forward:        a = W^T, [K = Cin][lda >= M = Cout] from ssad_transpose_filter (zero padded);
 * data gradient:  a = W itself ([K = Cout][lda = M = Cin]), x = dY, y = dX, no bias.
 * mask (may be NULL): y = mask[n][m][p] > 0 ? y : 0 -- the ReluGradient of the layer below fused
 * into the data gradient.  SSAD_GEMM_ACCUMULATE: y += result (a shortcut's gradient meeting the
 * main branch's; bias / residual must be NULL then).  Requires P % 4 == 0, lda % 4 == 0 and
 * 16-byte aligned a / x; tensors below 2 GiB.  Stride-2 pointwise layers run on the output of
 * ssad_subsample (a strided pointwise convolution is the pointwise convolution of the subsampled
 * map). */
#define SSAD_GEMM_RELU 1
#define SSAD_GEMM_ACCUMULATE 8
typedef struct ssad_gemm_conv {
  const float* a;          /* [K][lda] */
  const float* x;          /* [N][K][P] */
  float* y;                /* [N][M][P] */
  const float* bias;       /* [M] or NULL */
  const float* residual;   /* [N][M][P] or NULL */
  const float* mask;       /* [N][M][P] or NULL */
  int lda, N, K, P, M, flags;
} ssad_gemm_conv;
SSAD_API int ssad_conv1x1_gemm(const ssad_gemm_conv* desc_host, ssad_stream_t stream);
/* The same contract on the split-operand engine (round 6, gemm_split.hip): x and a as hi + lo fp16 under per-tensor
 * power-of-two scales from their measured |max|, three fp16 MFMAs per operand pair, fp32 accumulation (see
 * ssad_conv3x3_forward_split for the arithmetic and its limits).  x is split INSIDE the GEMM kernel (no packed copy of
 * it exists); a is split into the workspace.  Any K, M, P; tensors below 2 GiB (SSAD_E_BADARG otherwise: use
 * ssad_conv1x1_gemm).  workspace: ssad_conv1x1_gemm_split_workspace_bytes(desc) bytes; the call = |max| pass + split of
 * a + the GEMM on `stream`.  Pays on the compute-bound layers (K >= 256, M >= 256: res4, res5, the laterals); the
 * HBM-bound ones stay on ssad_conv1x1_gemm. */
SSAD_API size_t ssad_conv1x1_gemm_split_workspace_bytes(const ssad_gemm_conv* desc_host);
SSAD_API int ssad_conv1x1_gemm_split(const ssad_gemm_conv* desc_host, void* workspace, size_t workspace_bytes,
                                     ssad_stream_t stream);
/* ... with what the caller already has handed over, either or both: packed_a = the filter split by
 * ssad_gemm_split_pack_filters (desc->a is then not read), x_amax = x's |max| word (device; ssad_split_absmax or an
 * earlier call's).  With both the call is the GEMM kernel alone.  y_amax_out (device word the caller has zeroed, or
 * NULL): the kernel folds the |max| of what it stores to y into it (atomicMax, one per workgroup) -- the next
 * layer's x_amax. */
SSAD_API int ssad_conv1x1_gemm_split_amax(const ssad_gemm_conv* desc_host, const float* packed_a, const unsigned* x_amax,
                                          unsigned* y_amax_out, void* workspace, size_t workspace_bytes,
                                          ssad_stream_t stream);
/* Split a table of filters a[K][lda] (as ssad_gemm_conv.a) into the engine's operand order in three launches: dst
 * holds ssad_gemm_split_filter_floats(K, M) floats (a header with the filter's |max| + the hi and lo planes). */
typedef struct ssad_gemm_pack_entry {
  const float* a;
  float* dst;
  int lda, K, M;
} ssad_gemm_pack_entry;
SSAD_API size_t ssad_gemm_split_filter_floats(int K, int M);
SSAD_API int ssad_gemm_split_pack_filters(const ssad_gemm_pack_entry* entries_host, int n_entries, ssad_stream_t stream);
/* Convolution of any kernel / stride / pad (group 1) as an IMPLICIT GEMM, forward: the same kernel as
 * ssad_conv1x1_gemm with the im2col view of the image gathered by the DMA itself -- no column buffer
 * (caffe2/operators/conv_op_impl.h:126-173 materialises one per image; the 7x7/2 stem's is 1.35 GB at
 * bs 16).  d->x = image [N][C][H][W], d->a = transposed filter [C*kernel*kernel][lda] (ssad_transpose_filter
 * of w [M][C][k][k]), d->K = C*kernel*kernel, d->P = OH*OW (a multiple of 4), d->y = [N][M][OH][OW];
 * bias / residual / mask / flags as in ssad_conv1x1_gemm. */
SSAD_API int ssad_conv_implicit_gemm(const ssad_gemm_conv* d, int C, int H, int W, int kernel, int stride, int pad,
                                     ssad_stream_t stream);
/* The same with split-K through the caller's workspace: a k x k layer on a small map is a handful of tiles with a
 * very long reduction (FPN's P6, 3x3 / stride 2, 2048 -> 256 on 10 x 14 x 16 images: 36 tiles, K = 18 432).  The
 * reduction is cut into `splits` ranges of whole 16-row chunks, one workgroup per (tile, range); partial tiles go
 * to workspace[split][N][M][P] and a second launch adds them in split order (deterministic) and applies bias /
 * residual / ReLU / mask / accumulate.  workspace == NULL or a plan of one split = ssad_conv_implicit_gemm.
 * (detectron/lib/modeling/FPN.py:193-224 builds P6 / P7 with stride 2; conv_op_impl.h:126-173.) */
SSAD_API size_t ssad_conv_implicit_gemm_workspace_bytes(int N, int M, int C, int H, int W, int kernel, int stride,
                                                        int pad);
SSAD_API int ssad_conv_implicit_gemm_ws(const ssad_gemm_conv* d, int C, int H, int W, int kernel, int stride, int pad,
                                        void* workspace, size_t workspace_bytes, ssad_stream_t stream);
/* Gradients of a k x k / strided convolution (group 1) at the layer's own size (conv_op_impl.h:358-577: im2col,
 * dW += dY col^T, dcol = W^T dY, col2im -- there per image, here with the batch flattened into the GEMM's column
 * index so that P6 / P7's 140 / 35 pixels per image fill tiles): conv_strided.hip.
 *   wgrad: dw[M][C][k][k] (+)= sum_{n,oy,ox} dy[n][m][oy][ox] x[n][c][oy*s+ky-pad][ox*s+kx-pad]
 *   dgrad: dx[N][C][H][W] (+)= mask > 0 ? sum_{m,ky,kx} w[m][c][ky][kx] dy[n][m][(iy+pad-ky)/s][(ix+pad-kx)/s] : 0
 *          (mask: optional [N][C][H][W], the ReluGradient of the layer below; w in its natural layout, C*k*k % 4 == 0)
 * Every buffer involved must stay below 2 GiB (the column buffer is C*k*k x N*OH*OW floats). */
SSAD_API size_t ssad_conv_kxk_wgrad_workspace_bytes(int N, int C, int H, int W, int M, int kernel, int stride,
                                                    int pad);
SSAD_API int ssad_conv_kxk_wgrad(const float* x, const float* dy, int N, int C, int H, int W, int M, int kernel,
                                 int stride, int pad, float* dw, int accumulate, void* workspace,
                                 size_t workspace_bytes, ssad_stream_t stream);
SSAD_API size_t ssad_conv_kxk_dgrad_workspace_bytes(int N, int C, int H, int W, int M, int kernel, int stride,
                                                    int pad);
SSAD_API int ssad_conv_kxk_dgrad(const float* w, const float* dy, int N, int C, int H, int W, int M, int kernel,
                                 int stride, int pad, float* dx, const float* mask, int accumulate, void* workspace,
                                 size_t workspace_bytes, ssad_stream_t stream);
/* wt[k][m] = w[m][k], rows padded with zeros to ldm >= M (ldm % 4 == 0) */
SSAD_API int ssad_transpose_filter(const float* w, int M, int K, int ldm, float* wt, ssad_stream_t stream);
/* ssad_transpose_filter for a whole table of filters in ONE launch (the training step re-transposes every trainable
 * pointwise filter after each update: 34 launches of 5 us for the R-50 student, a serial chain that waits for a
 * free CU slot 34 times when another stream's persistent kernels hold the chip). */
#define SSAD_MAX_TRANSPOSE_ENTRIES 64   /* per launch; longer tables are chunked */
typedef struct ssad_transpose_entry {
  const float* w;          /* [M][K] */
  float* wt;               /* [K][ldm] */
  int M, K, ldm;
  int reserved;
} ssad_transpose_entry;
SSAD_API int ssad_transpose_filters(const ssad_transpose_entry* entries_host, int n_entries, ssad_stream_t stream);
/* dw[m][c] (+)= sum_{n,p} dy[n][m][p] x[n][c][p]  (conv_op_impl.h:451-500 for a 1x1 kernel);
 * deterministic split reduction through the caller's workspace.  P % 16 == 0. */
SSAD_API size_t ssad_conv1x1_wgrad_workspace_bytes(int N, int C, int P, int M);
SSAD_API int ssad_conv1x1_wgrad(const float* x, const float* dy, int N, int C, int P, int M, float* dw,
                                int accumulate, void* workspace, size_t workspace_bytes,
                                ssad_stream_t stream);
/* The same contract on the split-operand engine (gemm_split.hip, wpoint_split_kernel): dy and x scaled by a power
 * of two from their measured |max| and split on the fly into hi + lo fp16, three v_mfma_f32_32x32x16_f16 per
 * operand pair, fp32 accumulation.  P % 8 == 0. */
SSAD_API size_t ssad_conv1x1_wgrad_split_workspace_bytes(int N, int C, int P, int M);
SSAD_API int ssad_conv1x1_wgrad_split(const float* x, const float* dy, int N, int C, int P, int M, float* dw,
                                      int accumulate, void* workspace, size_t workspace_bytes,
                                      ssad_stream_t stream);
/* ... with x's and dy's |max| words handed over (device, one word each; both or neither) */
SSAD_API int ssad_conv1x1_wgrad_split_amax(const float* x, const float* dy, int N, int C, int P, int M, float* dw,
                                           int accumulate, void* workspace, size_t workspace_bytes,
                                           const unsigned* x_amax, const unsigned* dy_amax, ssad_stream_t stream);
/* y[n][c][oy][ox] = x[n][c][oy*stride][ox*stride], OH = (H-1)/stride + 1; and its gradient
 * dx (+)= scatter(dy) (zero off the sampled grid) */
SSAD_API int ssad_subsample(const float* x, int N, int C, int H, int W, int stride, float* y,
                            ssad_stream_t stream);
SSAD_API int ssad_subsample_grad(const float* dy, int N, int C, int H, int W, int stride, int accumulate,
                                 float* dx, ssad_stream_t stream);
/* Grouped 3x3 convolution, pad 1, stride 1 or 2, forward (ResNeXt's cardinality-64 layer:
 * detectron/lib/modeling/ResNet.py:247-258; conv_op_impl.h:93-98,126-173), NCHW fp32:
 * x [N][C][H][W], filter [C][C/group][3][3] packed once into MFMA operand order, bias [C] or NULL
 * -> y [N][C][OH][OW], OH = (H-1)/stride + 1.  Channels per group 4, 8, 16 or 32
 * (SSAD_E_BADARG / -1 otherwise: use the default engine). */
SSAD_API long long ssad_grouped_conv3x3_filter_floats(int C, int group);
SSAD_API int ssad_grouped_conv3x3_pack_filter(const float* w, int C, int group, float* packed,
                                              ssad_stream_t stream);
SSAD_API int ssad_grouped_conv3x3_forward(const float* x, const float* packed, const float* bias, int N, int C,
                                          int H, int W, int group, int stride, int relu, float* y,
                                          ssad_stream_t stream);
/* MaxPool / MaxPoolGradient, NCHW (caffe2/operators/pool_op.cu): windows are clipped to
 * the image; every input equal to its window's maximum receives the gradient */
SSAD_API int ssad_max_pool_forward(const float* x, int N, int C, int H, int W, int kh, int kw,
                                   int stride_h, int stride_w, int pad_t, int pad_l, int pad_b,
                                   int pad_r, float* y, ssad_stream_t stream);
/* The ResNet stem's pool (detectron/lib/modeling/ResNet.py:166-168: kernel 3, stride 2,
 * pad 1) with the AffineChannel bias and ReLU that precede it folded in:
 * y = relu?(max(window of x) + bias[c]) -- equal to pooling relu(x + bias[c]) because both
 * are monotonic.  bias may be NULL. */
SSAD_API int ssad_max_pool3x3s2_bias_relu(const float* x, const float* bias, int N, int C, int H,
                                          int W, int relu, float* y, ssad_stream_t stream);
SSAD_API int ssad_max_pool_backward(const float* x, const float* y, const float* dy, int N, int C,
                                    int H, int W, int kh, int kw, int stride_h, int stride_w,
                                    int pad_t, int pad_l, int pad_b, int pad_r, float* dx,
                                    ssad_stream_t stream);

/* RetinaNet inference post-processing for ONE image (detectron/lib/core/
 * test_retinanet.py:108-206): per level the candidates with score > inference_th
 * (0 on the coarsest level), the pre_nms_topn best of them, anchor decode
 * (utils/boxes.py:150-190) / rescale by 1/im_scale / clip to the image
 * (:132-147), per-class greedy NMS (utils/cython_nms.pyx:37-92), and the
 * dets_per_im best survivors sorted by score.
 *   cls_prob_host[l]  device float [1][A*C][H_l][W_l] (sigmoid scores)
 *   box_pred_host[l]  device float [1][A*4][H_l][W_l]
 *   cell_anchors      device double [levels][A][4]
 *   dets_out          device float [dets_per_im][6] = x1, y1, x2, y2, score, class (1-based)
 *   count_out         device int: rows produced
 * Score ties are ordered by element index (unspecified in the reference). */
SSAD_API size_t ssad_retinanet_detect_workspace_bytes(
    int levels, int A, int C, const int* H_host, const int* W_host, int pre_nms_topn);
SSAD_API int ssad_retinanet_detect(
    const float* const* cls_prob_host, const float* const* box_pred_host,
    const double* cell_anchors, int levels, int A, int C, int k_min, const int* H_host,
    const int* W_host, float inference_th, int pre_nms_topn, float nms_thresh, int dets_per_im,
    float im_scale, int im_height, int im_width, float bbox_xform_clip, float* dets_out,
    int* count_out, void* workspace, size_t workspace_bytes, ssad_stream_t stream);

/* ---------------------------------------------------------------------- */
/* fp16 storage / fp32 accumulation (BASELINE config 5's precision)        */
/* ---------------------------------------------------------------------- */
/* The reference's fp16 route is CudnnConvOp<float16> with fp32 math
 * (caffe2/operators/conv_op_cudnn.cc:631-636).  Activations between the subnet layers
 * are channel-blocked fp16, Xb[n][ceil(C/8)][H][W][8] (a tail block is zero padded), so
 * one 16-byte load is a v_mfma_f32_32x32x16_f16 operand for every filter tap. */
/* x_blocked = fp16(x_nchw * scale) / x_nchw = float(x_blocked) * scale: the scale carries the
 * loss scaling of a mixed-precision backward pass (gradients of the logits are ~1e-6) */
SSAD_API int ssad_f16_pack_activations(const float* x_nchw, int N, int C, int H, int W, float scale,
                                       void* x_blocked, ssad_stream_t stream);
SSAD_API int ssad_f16_unpack_activations(const void* x_blocked, int N, int C, int H, int W, float scale,
                                         float* x_nchw, ssad_stream_t stream);
/* The same with a further factor read from device memory at execution time (scale_dev[0], or
 * NULL): the DYNAMIC loss scale of ssad_loss_scale_update and its reciprocal never visit the host */
SSAD_API int ssad_f16_pack_activations_dyn(const float* x_nchw, int N, int C, int H, int W, float scale,
                                           const float* scale_dev, void* x_blocked, ssad_stream_t stream);
SSAD_API int ssad_f16_unpack_activations_dyn(const void* x_blocked, int N, int C, int H, int W, float scale,
                                             const float* scale_dev, float* x_nchw, ssad_stream_t stream);
/* The same layout change for float16 NCHW blobs (the operator surface's Conv on
 * TensorProto::FLOAT16 tensors), and plain element casts (round to nearest even) */
SSAD_API int ssad_f16_block_activations(const void* x_nchw_f16, int N, int C, int H, int W,
                                        void* x_blocked, ssad_stream_t stream);
SSAD_API int ssad_f16_unblock_activations(const void* x_blocked, int N, int C, int H, int W,
                                          void* x_nchw_f16, ssad_stream_t stream);
SSAD_API int ssad_cast_f16_to_f32(const void* in_f16, float* out, long long n, ssad_stream_t stream);
SSAD_API int ssad_cast_f32_to_f16(const float* in, void* out_f16, long long n, ssad_stream_t stream);
/* halves to allocate for either packed form of an [M][C][3][3] filter */
SSAD_API size_t ssad_f16_filter_halves(int M, int C);
/* w [M][C][3][3] fp32 -> packed_fwd [9][ceil(C/8)][M][8] and / or packed_dgrad
 * [9][ceil(M/8)][C][8] (flipped taps, roles of M and C exchanged); either may be NULL */
SSAD_API int ssad_f16_pack_filter(const float* w, int M, int C, void* packed_fwd,
                                  void* packed_dgrad, ssad_stream_t stream);
#define SSAD_F16_OUT_NCHW_F32 16   /* prediction layers: write y as NCHW fp32 for the loss kernels */
/* y = conv3x3(x_blocked, packed) (+ bias[M], fp32) (ReLU with SSAD_CONV_RELU, sigmoid with
 * SSAD_CONV_SIGMOID, only together with SSAD_F16_OUT_NCHW_F32); y is blocked fp16
 * [N][M/8][H][W][8] (M % 8 == 0) or, with SSAD_F16_OUT_NCHW_F32, float [N][M][H][W].  Each level's
 * tensors are addressed through 32-bit buffer offsets: at most 2 GiB per tensor (SSAD_E_BADARG beyond).  Data gradient: the same call with packed_dgrad,
 * C and M exchanged, bias NULL, and with SSAD_CONV_MASK_AUX the fused ReluGradient
 * y = aux > 0 ? y : 0 (aux blocked fp16 like y, else NULL). */
SSAD_API int ssad_conv3x3_forward_f16(const void* x_blocked, const void* packed, const float* bias,
                                      const void* aux, int N, int C, int H, int W, int M, int flags,
                                      void* y, ssad_stream_t stream);
/* The same for every FPN level sharing the filter -- and for several such problems of equal
 * (C, M), see the per-entry packed / bias -- in ONE launch (n_levels <= SSAD_MAX_F16_LEVELS);
 * aux per level, all NULL or (with SSAD_CONV_MASK_AUX) all set. */
#define SSAD_MAX_F16_LEVELS 24
typedef struct ssad_f16_level {
  const void* x;      /* blocked fp16 [N][ceil(C/8)][H][W][8] */
  void* y;            /* blocked fp16 [N][M/8][H][W][8] or float [N][M][H][W] */
  const void* aux;    /* blocked fp16 like y, or NULL */
  int N, H, W;
  /* optional per-entry filter / bias (NULL = the call's): independent convolutions of one
   * (C, M) -- teacher and student, cls and bbox tower layers of equal depth -- in one launch */
  const void* packed;
  const float* bias;
} ssad_f16_level;
SSAD_API int ssad_conv3x3_forward_f16_levels(const ssad_f16_level* levels_host, int n_levels,
                                             const void* packed, const float* bias, int C, int M,
                                             int flags, ssad_stream_t stream);

/* Filter and bias gradient from channel-blocked fp16 x [N][ceil(C/8)][H][W][8] and dy
 * [N][ceil(M/8)][H][W][8] (conv_op_impl.h:451-510): dw [M][C][3][3] and db [M] in fp32,
 * overwritten, or added to when accumulate != 0 (the five FPN levels share one filter:
 * caffe2/python/core.py:706-741 sums their gradients); both are multiplied by `scale` (the
 * inverse loss scale) on the way out.  db may be NULL.  Deterministic. */
SSAD_API size_t ssad_conv3x3_wgrad_f16_workspace_bytes(int N, int C, int H, int W, int M);
SSAD_API int ssad_conv3x3_wgrad_f16(const void* x_blocked, const void* dy_blocked, int N, int C, int H,
                                    int W, int M, int accumulate, float scale, float* dw, float* db,
                                    void* workspace, size_t workspace_bytes, ssad_stream_t stream);
/* The same summed over every FPN level sharing the filter, in one launch */
typedef struct ssad_f16_wgrad_level {
  const void* x;      /* blocked fp16 [N][ceil(C/8)][H][W][8] */
  const void* dy;     /* blocked fp16 [N][ceil(M/8)][H][W][8] */
  int N, H, W;
} ssad_f16_wgrad_level;
SSAD_API size_t ssad_conv3x3_wgrad_f16_levels_workspace_bytes(const ssad_f16_wgrad_level* levels_host,
                                                              int n_levels, int C, int M);
SSAD_API int ssad_conv3x3_wgrad_f16_levels(const ssad_f16_wgrad_level* levels_host, int n_levels, int C,
                                           int M, int accumulate, float scale, float* dw, float* db,
                                           void* workspace, size_t workspace_bytes,
                                           ssad_stream_t stream);
/* ... times scale_dev[0] read on the device (the reciprocal of the dynamic loss scale), or NULL */
SSAD_API int ssad_conv3x3_wgrad_f16_levels_dyn(const ssad_f16_wgrad_level* levels_host, int n_levels,
                                               int C, int M, int accumulate, float scale,
                                               const float* scale_dev, float* dw, float* db,
                                               void* workspace, size_t workspace_bytes,
                                               ssad_stream_t stream);

/* ---- the backbones in the same precision (BASELINE config 5; gemm_f16.hip, grouped_f16.hip) ---- */
/* Pointwise (1x1) convolution on channel-blocked fp16 tensors, fp32 accumulation:
 *   y[n][m][oy][ox] = act( sum_c W[m][c] x[n][c][stride oy][stride ox] + bias[m] + residual )
 * x [N][ceil(C/8)][Hi][Wi][8], y [N][M/8][Ho][Wo][8] (M % 8 == 0); w = packed_fwd of
 * ssad_pw_f16_pack_filter.  residual (or NULL): blocked fp16 like y -- or, with
 * SSAD_PW_F16_RES_UPSAMPLE2, the [Ho/2][Wo/2] map whose 2x nearest upsampling is added (FPN's
 * top-down Sum, FPN.py:283-306).  SSAD_CONV_RELU: ReLU.  mask (or NULL): blocked fp16 like y,
 * y = mask > 0 ? y : 0 (ReluGradient of the layer below, for the data gradient -- the same call
 * with packed_dgrad, C and M exchanged).  Hi / Wi = 0: Ho * stride, Wo * stride.
 * Reference: CudnnConvOp<float16>, fp32 math (conv_op_cudnn.cc:631-636), algorithm
 * conv_op_impl.h:126-173. */
#define SSAD_PW_F16_RES_UPSAMPLE2 32
typedef struct ssad_pw_f16 {
  const void* x;
  const void* w;
  const float* bias;
  const void* residual;
  const void* mask;
  void* y;
  int N, C, M, Ho, Wo, Hi, Wi, stride, flags;
} ssad_pw_f16;
SSAD_API int ssad_conv1x1_f16(const ssad_pw_f16* desc_host, ssad_stream_t stream);
/* w [M][C] fp32 -> packed_fwd [ceil(C/8)][M][8] and / or packed_dgrad [ceil(M/8)][C][8] fp16 */
SSAD_API size_t ssad_pw_f16_filter_halves(int M, int C);
SSAD_API int ssad_pw_f16_pack_filter(const float* w, int M, int C, void* packed_fwd, void* packed_dgrad,
                                     ssad_stream_t stream);
/* Filter (and optionally bias) gradient of a pointwise layer from blocked fp16 x and dy of one
 * map size: dw [M][C], db [M] fp32, times scale (times scale_dev[0] if not NULL).  Deterministic. */
SSAD_API size_t ssad_conv1x1_wgrad_f16_workspace_bytes(int N, int C, int H, int W, int M);
SSAD_API int ssad_conv1x1_wgrad_f16(const void* x_blocked, const void* dy_blocked, int N, int C, int H, int W,
                                    int M, int accumulate, float scale, const float* scale_dev, float* dw,
                                    float* db, void* workspace, size_t workspace_bytes, ssad_stream_t stream);
/* Elementwise passes on blocked fp16 tensors; N, C, H, W describe y.
 *   0 subsample       y[y][x] = a[stride y][stride x]                     (a: H stride x W stride)
 *   1 subsample grad  y[y][x] (+)= both multiples of stride ? a[y/stride][x/stride] : 0
 *   2 upsample grad   y = sum of a's 2 x 2 blocks (a: 2H x 2W) (+ b)    upsample_nearest_op.cu:62-151
 *   3 sum             y = a + b
 *   4 relu            y = max(a, 0)
 *   5 relu grad       y = a > 0 ? b : 0                                  relu_op.cu:44-53 */
SSAD_API int ssad_f16_elementwise(int mode, const void* a, const void* b, void* y, int N, int C, int H, int W,
                                  int stride, int accumulate, ssad_stream_t stream);
/* Every fp16 filter of a network packed in ONE launch (the step repacks each trained filter after its
 * update: config 5's R-101 student has 66 pointwise and 45 3x3 layers, i.e. 111 launches of 5-13 us that
 * ran as a serial chain at the head of every step).  taps = 1: the pointwise layout of
 * ssad_pw_f16_pack_filter; taps = 9: the 3x3 layout of ssad_f16_pack_filter.  wf / wd may each be NULL. */
#define SSAD_MAX_F16_PACK_ENTRIES 64   /* per launch; longer tables are chunked */
typedef struct ssad_f16_pack_entry {
  const float* w;          /* [M][C] or [M][C][3][3] fp32 */
  void* wf;                /* forward pack (or NULL) */
  void* wd;                /* data-gradient pack (or NULL) */
  int M, C, taps, reserved;
} ssad_f16_pack_entry;
SSAD_API int ssad_f16_pack_filters(const ssad_f16_pack_entry* entries_host, int n_entries, ssad_stream_t stream);
/* The strided pointwise layer's view of its input at any map size (odd maps too: the output is
 * [(Hi - 1) / stride + 1][(Wi - 1) / stride + 1], conv_pool_op_base.h:45-194 with kernel 1, pad 0):
 * y[y][x] = a[stride y][stride x] on blocked fp16. */
SSAD_API int ssad_f16_subsample(const void* a_blocked, int N, int C, int Hi, int Wi, int stride, void* y_blocked,
                                ssad_stream_t stream);
/* bias + ReLU + 3x3 / stride 2 / pad 1 max pool of the stem's fp32 NCHW output z [N][C][H][W],
 * written blocked fp16 [N][C/8][ceil(H/2)][ceil(W/2)][8] */
SSAD_API int ssad_stem_pool_f16(const float* z, const float* bias, int N, int C, int H, int W, void* y_blocked,
                                ssad_stream_t stream);
/* ResNeXt's grouped 3x3 (stride 1, pad 1; ResNet.py:247-258) on blocked fp16, forward:
 * w [C][C/group][3][3] -> packed (ssad_grouped_conv3x3_f16_filter_halves halves); C % 64 == 0,
 * C / group in {4, 8, 16, 32}; y = relu?(conv + bias) */
SSAD_API size_t ssad_grouped_conv3x3_f16_filter_halves(int C, int group);
SSAD_API int ssad_grouped_conv3x3_f16_pack_filter(const float* w, int C, int group, void* packed,
                                                  ssad_stream_t stream);
SSAD_API int ssad_grouped_conv3x3_f16(const void* x_blocked, const void* packed, const float* bias, int N, int C,
                                      int H, int W, int group, int relu, void* y_blocked, ssad_stream_t stream);

/* Plain fp32 GEMM on the matrix cores (exact fp32 v_mfma_f32_32x32x2_f32): row-major
 * C[M x N] = alpha op(A) op(B) + beta C for `batch` problems `stride_*` elements apart -- math::Gemm /
 * math::GemmStridedBatched (caffe2/utils/math_gpu.cu:33-80) of the default convolution engine's im2col
 * route (conv_op_impl.h:126-173, :451-560).  Any sizes / leading dimensions; beta == 0 does not read C. */
SSAD_API int ssad_gemm_f32(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* A, int lda,
                           long long stride_a, const float* B, int ldb, long long stride_b, float beta, float* C,
                           int ldc, long long stride_c, int batch, ssad_stream_t stream);

/* ---------------------------------------------------------------------- */
/* Introspection                                                           */
/* ---------------------------------------------------------------------- */
SSAD_API const char* ssad_kernels_arch(void);   /* "gfx950" */
/* ABI version of this header.  History: 2 -> 3 (round 4): ssad_sgd_segment grew row_len / row_scale
 * (24 -> 32 bytes); ssad_pow_sum / ssad_cls_losses_fused need the larger workspace their
 * *_workspace_bytes report and zero their own arrival counters; *_prezeroed entry points added.
 * A binding built against version 2 must refuse to run against this library. */
#define SSAD_KERNELS_ABI_VERSION 3
SSAD_API int ssad_kernels_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SSAD_KERNELS_H_ */
