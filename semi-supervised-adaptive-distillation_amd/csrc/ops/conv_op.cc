#include "ops/conv_op.h"

#include "c2/blas.h"
#include "ssad_kernels.h"

namespace caffe2 {

ConvGeometry ParseConvGeometry(const OperatorBase& op) { return ParseConvGeometryFrom(op); }
ConvGeometry ParseConvGeometry(const OperatorDef& def) { return ParseConvGeometryFrom(DefArgs(def)); }

bool IsSubnetGeometry(const ConvGeometry& g) {
  return g.order == "NCHW" && g.group == 1 && g.kernel == vector<int>{3, 3} &&
         g.stride == vector<int>{1, 1} && g.dilation == vector<int>{1, 1} &&
         g.pads == vector<int>{1, 1, 1, 1};
}

// A pointwise layer the fp16 GEMM kernel takes (gemm_f16.hip): 1x1, no padding, no dilation, one
// group, equal strides 1 or 2 -- the bottleneck 1x1s, projection shortcuts and FPN laterals.
bool IsPointwiseF16Geometry(const ConvGeometry& g) {
  return g.order == "NCHW" && g.group == 1 && g.kernel == vector<int>{1, 1} &&
         g.dilation == vector<int>{1, 1} && g.pads == vector<int>{0, 0, 0, 0} &&
         g.stride[0] == g.stride[1] && (g.stride[0] == 1 || g.stride[0] == 2);
}

bool IsDefaultEngineGeometry(const ConvGeometry& g) {
  return g.order == "NCHW" && g.group >= 1 && g.kernel.size() == 2;
}

// Algorithm choice, like cuDNN's internal one (conv_op_cudnn.cc:541-558):
// Winograd F(2x2,3x3) for outputs >= 32 channels wide, the direct kernel
// otherwise; arg hip_algo = "direct" | "winograd" overrides.
bool UseWinograd(const string& algo, int out_channels) {
  if (algo == "direct") return false;
  if (algo == "winograd") return true;
  // ("winograd24" names the F(2x4, 3x3) engine the net lowering assigns; wherever it does not apply -- narrow
  // layers, the filter gradient -- the choice is the automatic one)
  return out_channels >= 32;
}

namespace {
enum { INPUT = 0, FILTER = 1, BIAS = 2, OUTPUT_GRAD = 2 };
enum { FILTER_GRAD = 0, BIAS_OR_INPUT_GRAD = 1, INPUT_GRAD = 2 };
}  // namespace

template <>
bool ConvOp<float, HIPContext>::RunDefaultEngine();
template <>
bool ConvGradientOp<float, HIPContext>::RunDefaultEngine();
template <>
bool ConvOp<float, HIPContext>::RunFloat16();
template <>
bool ConvOp<float, HIPContext>::RunFloat16Pointwise();
template <>
bool ConvGradientOp<float, HIPContext>::RunFloat16Pointwise();
template <>
bool ConvGradientOp<float, HIPContext>::RunFloat16();

namespace {
size_t BlockedHalves(int N, int C, int H, int W) {
  return (size_t)N * (size_t)((C + 7) / 8) * 8 * (size_t)H * (size_t)W;
}
}  // namespace

template <>
bool ConvOp<float, HIPContext>::RunOnDevice() {
  auto& X = Input(INPUT);
  auto& filter = Input(FILTER);
  auto* Y = Output(0);
  CAFFE_ENFORCE(!fuse_sigmoid_ || (!X.IsType<float16>() && IsSubnetGeometry(geom_)),
                "Conv(fuse_sigmoid): only the fp32 3x3 / stride 1 / pad 1 engines carry the Sigmoid epilogue");
  if (X.IsType<float16>()) return RunFloat16();
  CAFFE_ENFORCE_EQ(X.ndim(), 4);
  CAFFE_ENFORCE_EQ(X.ndim(), filter.ndim());
  const int N = X.dim32(0), C = X.dim32(1), H = X.dim32(2), W = X.dim32(3);
  const int M = filter.dim32(0);
  CAFFE_ENFORCE(C == filter.dim32(1) * geom_.group,
                "Convolution op: input channels does not match: # of input channels ", C,
                " is not equal to kernel channels * group:", filter.dim32(1), "*", geom_.group);
  CAFFE_ENFORCE(filter.dim32(2) == geom_.kernel[0] && filter.dim32(3) == geom_.kernel[1]);
  const float* bias = nullptr;
  if (InputSize() == 3) {
    auto& b = Input(BIAS);
    CAFFE_ENFORCE(b.ndim() == 1);
    CAFFE_ENFORCE(b.dim32(0) == M);
    bias = b.data<float>();
  }
  if (!IsSubnetGeometry(geom_)) return RunDefaultEngine();
  Y->Resize(N, M, H, W);   // 3x3 / s1 / p1 keeps the spatial size

  hipStream_t s = context_.hip_stream();
  ssad_conv_level lv{X.data<float>(), Y->mutable_data<float>(), nullptr, N, H, W, nullptr, nullptr};
  const int flags = (fuse_relu_ ? SSAD_CONV_RELU : 0) | (fuse_sigmoid_ ? SSAD_CONV_SIGMOID : 0);
  int rc;
  // the packed filter is rebuilt only when the filter blob was written since (ops/filter_pack_cache.h)
  // "split" / "winograd24" are assigned by the net lowering (net_lowering.cc): the split-operand engine from 256
  // outputs up, F(2x4) from 128
  const bool sp = algo_ == "split" && M >= 256;
  const bool f24 = (algo_ == "winograd24" || algo_ == "split") && M >= 128 && !sp;
  const bool wino = UseWinograd((algo_ == "winograd24" || algo_ == "split") ? string("auto") : algo_, M);
  const auto kind = sp ? FilterPackCache::SPLIT_FWD : f24 ? FilterPackCache::WINO24_FWD
                       : wino ? FilterPackCache::WINO_FWD : FilterPackCache::DIRECT_FWD;
  const long long before = pack_cache_.packs_issued();
  pack_cache_.Want(filter, kind);
  pack_cache_.Flush(s);
  g_filter_packs_issued += pack_cache_.packs_issued() - before;
  const float* packed = pack_cache_.Packed(filter, kind);
  rc = sp ? split_.Run(&lv, 1, packed, bias, M, C, flags, s)
       : f24 ? ssad_conv3x3_forward_wino24(&lv, 1, packed, bias, M, C, flags, s)
       : wino ? ssad_conv3x3_forward_wino(&lv, 1, packed, bias, M, C, flags, s)
              : ssad_conv3x3_forward(&lv, 1, packed, bias, M, C, flags, s);
  CAFFE_ENFORCE_EQ(rc, 0, "Conv launch failed");
  ++g_conv_launch_calls;
  return true;
}

// A pointwise layer the GEMM kernels take: 1x1, no padding, no dilation, equal strides, one group,
// 16-byte pixel rows, 16-byte aligned channel rows (C % 4), tensors below 2 GiB.
// SSAD_CONV1X1_ENGINE=blas keeps the im2col + general-GEMM route (kernels/gemm_general.hip).
static bool PointwiseGemmEligible(const ConvGeometry& g, int N, int C, int M, int P) {
  static const bool off = [] { const char* e = getenv("SSAD_CONV1X1_ENGINE"); return e && std::string(e) == "blas"; }();
  if (off) return false;
  if (g.kernel[0] != 1 || g.kernel[1] != 1 || g.group != 1) return false;
  if (g.pads != vector<int>{0, 0, 0, 0} || g.dilation != vector<int>{1, 1}) return false;
  if (g.stride[0] != g.stride[1] || g.stride[0] < 1) return false;
  if ((P & 3) || (C & 3) || N < 1) return false;
  const long long big = (long long)N * (C > M ? C : M) * P * 4;
  return big < (1LL << 31) && (long long)C * ((M + 3) / 4 * 4) * 4 < (1LL << 31);
}

// A k x k layer (any stride / pad, one group, square kernel, no dilation) the implicit-GEMM kernel
// takes: any output map size, tensors below 2 GiB.  SSAD_CONV1X1_ENGINE=blas disables it too.
static bool ImplicitGemmEligible(const ConvGeometry& g, int N, int C, int H, int W, int M, int P) {
  static const bool off = [] {
    const char* e = getenv("SSAD_CONV1X1_ENGINE");
    return e && std::string(e) == "blas";
  }();
  if (off || g.group != 1 || g.kernel[0] != g.kernel[1] || g.dilation != vector<int>{1, 1}) return false;
  if (g.stride[0] != g.stride[1] || g.stride[0] < 1) return false;
  if (g.pads[0] != g.pads[1] || g.pads[0] != g.pads[2] || g.pads[0] != g.pads[3]) return false;
  if (N < 1) return false;
  const long long K = (long long)C * g.kernel[0] * g.kernel[0];
  return (long long)N * C * H * W * 4 < (1LL << 31) && (long long)N * M * P * 4 < (1LL << 31) &&
         K * ((M + 3) / 4 * 4) * 4 < (1LL << 31);
}

// Default engine, forward (conv_op_impl.h:31-202): per image im2col -> col[C*kh*kw][OH*OW],
// Y[n] = filter[M][C*kh*kw] . col, then the bias; a 1x1 / stride 1 / pad 0 layer skips the
// im2col (its col buffer IS the image).
template <>
bool ConvOp<float, HIPContext>::RunDefaultEngine() {
  auto& X = Input(INPUT);
  auto& filter = Input(FILTER);
  auto* Y = Output(0);
  const int N = X.dim32(0), C = X.dim32(1), H = X.dim32(2), W = X.dim32(3);
  const int M = filter.dim32(0);
  const int kh = geom_.kernel[0], kw = geom_.kernel[1];
  const int OH = ssad_conv_out_size(H, kh, geom_.dilation[0], geom_.pads[0], geom_.pads[2], geom_.stride[0]);
  const int OW = ssad_conv_out_size(W, kw, geom_.dilation[1], geom_.pads[1], geom_.pads[3], geom_.stride[1]);
  CAFFE_ENFORCE(OH > 0 && OW > 0, "Conv: the kernel does not fit the padded input");
  Y->Resize(N, M, OH, OW);
  hipStream_t s = context_.hip_stream();
  // groups (conv_op_impl.h:93-98,126-173; gradient :451-500,524-560): group g maps input channels [g*C/G, (g+1)*C/G) to
  // output channels [g*M/G, (g+1)*M/G); with the col rows ordered (c, kh, kw) the groups are
  // G independent GEMMs over contiguous row blocks -> one strided-batched call per image
  const int G = geom_.group;
  CAFFE_ENFORCE(M % G == 0, "the number of output channels is not divisible by group");
  const int K = C * kh * kw, P = OH * OW, Kg = K / G, Mg = M / G;
  const bool pointwise = kh == 1 && kw == 1 && geom_.stride == vector<int>{1, 1} &&
                         geom_.pads == vector<int>{0, 0, 0, 0};
  // Pointwise layers (the bottleneck 1x1s, projection shortcuts, FPN laterals; stride 1 or the
  // stride-2 first blocks): this repo's fp32-MFMA GEMM over the whole batch with the bias (and a
  // fused Relu) in its epilogue (kernels/gemm_conv.hip) instead of N per-image GEMM calls + a bias pass.
  // A strided pointwise convolution is the pointwise convolution of the subsampled map.
  if (PointwiseGemmEligible(geom_, N, C, M, P)) {
    const int st = geom_.stride[0];
    const float* xin = X.data<float>();
    if (st > 1) {
      col_buffer_.Resize((TIndex)N * C * P);
      CAFFE_ENFORCE_EQ(ssad_subsample(xin, N, C, H, W, st, col_buffer_.mutable_data<float>(), s), 0);
      xin = col_buffer_.data<float>();
    }
    const int ldm = (M + 3) / 4 * 4;
    packed_filter_.Resize((TIndex)C * ldm);
    float* wt = packed_filter_.mutable_data<float>();
    CAFFE_ENFORCE_EQ(ssad_transpose_filter(filter.data<float>(), M, C, ldm, wt, s), 0);
    const ssad_gemm_conv d{wt, xin, Y->mutable_data<float>(),
                           InputSize() == 3 ? Input(BIAS).data<float>() : nullptr, nullptr, nullptr,
                           ldm, N, C, P, M, fuse_relu_ ? SSAD_GEMM_RELU : 0};
    CAFFE_ENFORCE_EQ(ssad_conv1x1_gemm(&d, s), 0, "pointwise Conv launch failed");
    return true;
  }
  // k x k (7x7/2 stem, 3x3/2 of P6 / P7 and of the first bottleneck blocks under STRIDE_1X1 = False):
  // the same GEMM kernel with the im2col view gathered by its DMA -- no column buffer, whole batch,
  // bias (and a fused Relu) in the epilogue.
  if (!pointwise && ImplicitGemmEligible(geom_, N, C, H, W, M, P)) {
    const int ldm = (M + 3) / 4 * 4;
    packed_filter_.Resize((TIndex)K * ldm);
    float* wt = packed_filter_.mutable_data<float>();
    CAFFE_ENFORCE_EQ(ssad_transpose_filter(filter.data<float>(), M, K, ldm, wt, s), 0);
    const ssad_gemm_conv d{wt, X.data<float>(), Y->mutable_data<float>(),
                           InputSize() == 3 ? Input(BIAS).data<float>() : nullptr, nullptr, nullptr,
                           ldm, N, K, P, M, fuse_relu_ ? SSAD_GEMM_RELU : 0};
    CAFFE_ENFORCE_EQ(ssad_conv_implicit_gemm(&d, C, H, W, kh, geom_.stride[0], geom_.pads[0], s), 0,
                     "implicit-GEMM Conv launch failed");
    return true;
  }
  if (!pointwise) col_buffer_.Resize((TIndex)K * P);
  const float* Wd = filter.data<float>();
  for (int n = 0; n < N; ++n) {
    const float* xn = X.data<float>() + (size_t)n * C * H * W;
    const float* col = xn;
    if (!pointwise) {
      float* cb = col_buffer_.mutable_data<float>();
      CAFFE_ENFORCE_EQ(ssad_im2col(xn, C, H, W, kh, kw, geom_.dilation[0], geom_.dilation[1],
                                   geom_.pads[0], geom_.pads[1], geom_.pads[2], geom_.pads[3],
                                   geom_.stride[0], geom_.stride[1], cb, s), 0);
      col = cb;
    }
    float* yn = Y->mutable_data<float>() + (size_t)n * M * P;
    if (G == 1)
      GemmRowMajor(s, false, false, M, P, K, 1.0f, Wd, K, col, P, 0.0f, yn, P);
    else
      GemmRowMajorStridedBatched(s, false, false, Mg, P, Kg, 1.0f, Wd, Kg, (long long)Mg * Kg, col, P,
                                 (long long)Kg * P, 0.0f, yn, P, (long long)Mg * P, G);
  }
  if (InputSize() == 3 || fuse_relu_) {
    const float* bias = InputSize() == 3 ? Input(BIAS).data<float>() : nullptr;
    CAFFE_ENFORCE_EQ(ssad_affine_channel(Y->data<float>(), nullptr, bias, nullptr,
                                         Y->mutable_data<float>(), N, M, P, fuse_relu_, s), 0);
  }
  return true;
}

// Default engine, backward (conv_op_impl.h:358-577): dfilter = sum_n dY[n] . col[n]^T,
// dbias = channel sums of dY, dX[n] = col2im(filter^T . dY[n]).
template <>
bool ConvGradientOp<float, HIPContext>::RunDefaultEngine() {
  auto& X = Input(INPUT);
  auto& filter = Input(FILTER);
  auto& dY = Input(OUTPUT_GRAD);
  auto* dfilter = Output(FILTER_GRAD);
  const int N = X.dim32(0), C = X.dim32(1), H = X.dim32(2), W = X.dim32(3);
  const int M = filter.dim32(0);
  const int kh = geom_.kernel[0], kw = geom_.kernel[1];
  const int OH = ssad_conv_out_size(H, kh, geom_.dilation[0], geom_.pads[0], geom_.pads[2], geom_.stride[0]);
  const int OW = ssad_conv_out_size(W, kw, geom_.dilation[1], geom_.pads[1], geom_.pads[3], geom_.stride[1]);
  CAFFE_ENFORCE(dY.dim32(0) == N && dY.dim32(1) == M && dY.dim32(2) == OH && dY.dim32(3) == OW,
                "output gradient shape does not match the convolution output");
  CAFFE_ENFORCE(!relu_grad_on_input_, "relu_grad_on_input is an extension of the 3x3 engine");
  dfilter->ResizeLike(filter);
  hipStream_t s = context_.hip_stream();
  const int G = geom_.group;
  CAFFE_ENFORCE(M % G == 0, "the number of output channels is not divisible by group");
  const int K = C * kh * kw, P = OH * OW, Kg = K / G, Mg = M / G;
  const bool pointwise = kh == 1 && kw == 1 && geom_.stride == vector<int>{1, 1} &&
                         geom_.pads == vector<int>{0, 0, 0, 0};
  const bool want_dx = OutputSize() == 3 || (no_bias_ && OutputSize() == 2);
  Tensor<HIPContext>* dX = want_dx ? Output(no_bias_ ? BIAS_OR_INPUT_GRAD : INPUT_GRAD) : nullptr;
  if (dX) dX->ResizeLike(X);
  if (PointwiseGemmEligible(geom_, N, C, M, P) && P % 16 == 0) {
    // dfilter = sum_{n,p} dY X^T (split reduction), dX = filter^T . dY on the GEMM kernels
    const int st = geom_.stride[0];
    const float* xin = X.data<float>();
    if (st > 1) {
      col_buffer_.Resize((TIndex)N * C * P);
      CAFFE_ENFORCE_EQ(ssad_subsample(xin, N, C, H, W, st, col_buffer_.mutable_data<float>(), s), 0);
      xin = col_buffer_.data<float>();
    }
    const size_t wsb = ssad_conv1x1_wgrad_workspace_bytes(N, C, P, M);
    workspace_.Resize((TIndex)wsb);
    CAFFE_ENFORCE_EQ(ssad_conv1x1_wgrad(xin, dY.data<float>(), N, C, P, M, dfilter->mutable_data<float>(), 0,
                                        workspace_.mutable_data<uint8_t>(), wsb, s), 0);
    if (dX) {
      float* dxs = dX->mutable_data<float>();
      if (st > 1) dxs = col_buffer_.mutable_data<float>();     // X's subsampled copy is no longer needed
      const ssad_gemm_conv d{filter.data<float>(), dY.data<float>(), dxs, nullptr, nullptr, nullptr,
                             C, N, M, P, C, 0};
      CAFFE_ENFORCE_EQ(ssad_conv1x1_gemm(&d, s), 0, "pointwise ConvGradient launch failed");
      if (st > 1)
        CAFFE_ENFORCE_EQ(ssad_subsample_grad(dxs, N, C, H, W, st, 0, dX->mutable_data<float>(), s), 0);
    }
    if (!no_bias_) {
      auto* dbias = Output(BIAS_OR_INPUT_GRAD);
      dbias->Resize(M);
      CAFFE_ENFORCE_EQ(ssad_channel_sum(dY.data<float>(), N, M, P, dbias->mutable_data<float>(), 0, s), 0);
    }
    return true;
  }
  {
    // k x k / strided layers of group 1 (FPN's P6 / P7, the stem): the whole batch at once -- im2col with the batch
    // flattened into the GEMM's column index, filter gradient on the nt GEMM kernel (deterministic split
    // reduction), data gradient = nn GEMM + gather-form col2im (kernels/conv_strided.hip).  The per-image loop
    // below (conv_op_impl.h:451-560 as written) stays for grouped / dilated / rectangular / asymmetric geometries;
    // (rounds 3-5 could force the per-image path with an environment switch; retired)
    constexpr bool batched = true;
    const bool square = kh == kw && geom_.stride[0] == geom_.stride[1] && geom_.dilation == vector<int>{1, 1} &&
                        geom_.pads[0] == geom_.pads[1] && geom_.pads[0] == geom_.pads[2] &&
                        geom_.pads[0] == geom_.pads[3];
    const bool dx_ok = !dX || (K % 4 == 0 && ((uintptr_t)filter.data<float>() & 15) == 0);
    // The column buffer of a whole batch can be large (7x7 stem, 16 images of 600x1000: 1.4 GB + 0.6 GB of
    // transposed dY, against 88 MB per image in the loop below): the batch is processed in image groups whose
    // workspace stays under a budget (SSAD_CONVGRAD_WS_BYTES, default 256 MiB); the filter gradient accumulates
    // over the groups in order (deterministic).  A layer whose single image exceeds the budget takes the loop.
    const size_t budget = [] {
      const char* e = getenv("SSAD_CONVGRAD_WS_BYTES");
      const long long b = e ? atoll(e) : 0;
      return b > 0 ? (size_t)b : (size_t)256 << 20;
    }();
    auto need = [&](int n) -> size_t {
      const size_t a = ssad_conv_kxk_wgrad_workspace_bytes(n, C, H, W, M, kh, geom_.stride[0], geom_.pads[0]);
      if (a == 0) return 0;
      const size_t b = dX ? ssad_conv_kxk_dgrad_workspace_bytes(n, C, H, W, M, kh, geom_.stride[0], geom_.pads[0]) : 0;
      return a > b ? a : b;
    };
    int grp = 0;
    if (batched && G == 1 && square && !pointwise && dx_ok && need(1) > 0 && need(1) <= budget) {
      grp = 1;
      while (grp < N) {                      // the largest group under the budget (need() is monotone in n)
        const int nxt = grp * 2 < N ? grp * 2 : N;
        const size_t nb = need(nxt);
        if (nb == 0 || nb > budget) break;
        grp = nxt;
      }
    }
    if (grp > 0) {
      workspace_.Resize((TIndex)need(grp));
      for (int n0 = 0; n0 < N; n0 += grp) {
        const int n = (N - n0 < grp) ? N - n0 : grp;
        const float* xg = X.data<float>() + (size_t)n0 * C * H * W;
        const float* dyg = dY.data<float>() + (size_t)n0 * M * P;
        const size_t wsw = ssad_conv_kxk_wgrad_workspace_bytes(n, C, H, W, M, kh, geom_.stride[0], geom_.pads[0]);
        CAFFE_ENFORCE_EQ(ssad_conv_kxk_wgrad(xg, dyg, n, C, H, W, M, kh, geom_.stride[0], geom_.pads[0],
                                             dfilter->mutable_data<float>(), n0 > 0 ? 1 : 0,
                                             workspace_.mutable_data<uint8_t>(), wsw, s), 0,
                         "ConvGradient (filter, batched im2col) launch failed");
        if (dX) {
          const size_t wsd = ssad_conv_kxk_dgrad_workspace_bytes(n, C, H, W, M, kh, geom_.stride[0], geom_.pads[0]);
          CAFFE_ENFORCE_EQ(ssad_conv_kxk_dgrad(filter.data<float>(), dyg, n, C, H, W, M, kh, geom_.stride[0],
                                               geom_.pads[0], dX->mutable_data<float>() + (size_t)n0 * C * H * W,
                                               nullptr, 0, workspace_.mutable_data<uint8_t>(), wsd, s), 0,
                           "ConvGradient (input, batched col2im) launch failed");
        }
      }
      if (!no_bias_) {
        auto* dbias = Output(BIAS_OR_INPUT_GRAD);
        dbias->Resize(M);
        CAFFE_ENFORCE_EQ(ssad_channel_sum(dY.data<float>(), N, M, P, dbias->mutable_data<float>(), 0, s), 0);
      }
      return true;
    }
  }
  if (!pointwise) col_buffer_.Resize((TIndex)K * P);
  for (int n = 0; n < N; ++n) {
    const float* xn = X.data<float>() + (size_t)n * C * H * W;
    const float* dyn = dY.data<float>() + (size_t)n * M * P;
    const float* col = xn;
    if (!pointwise) {
      float* cb = col_buffer_.mutable_data<float>();
      CAFFE_ENFORCE_EQ(ssad_im2col(xn, C, H, W, kh, kw, geom_.dilation[0], geom_.dilation[1],
                                   geom_.pads[0], geom_.pads[1], geom_.pads[2], geom_.pads[3],
                                   geom_.stride[0], geom_.stride[1], cb, s), 0);
      col = cb;
    }
    // dfilter[M][K] (+)= dY[n][M][P] . col[K][P]^T
    if (G == 1)
      GemmRowMajor(s, false, true, M, K, P, 1.0f, dyn, P, col, P, n == 0 ? 0.0f : 1.0f,
                   dfilter->mutable_data<float>(), K);
    else   // per group: dfilter[g][Mg][Kg] (+)= dY[n][g][Mg][P] . col[g][Kg][P]^T
      GemmRowMajorStridedBatched(s, false, true, Mg, Kg, P, 1.0f, dyn, P, (long long)Mg * P, col, P,
                                 (long long)Kg * P, n == 0 ? 0.0f : 1.0f,
                                 dfilter->mutable_data<float>(), Kg, (long long)Mg * Kg, G);
    if (dX) {
      float* dxn = dX->mutable_data<float>() + (size_t)n * C * H * W;
      float* dcol = pointwise ? dxn : col_buffer_.mutable_data<float>();
      // dcol[K][P] = filter[M][K]^T . dY[n][M][P]
      if (G == 1)
        GemmRowMajor(s, true, false, K, P, M, 1.0f, filter.data<float>(), K, dyn, P, 0.0f, dcol, P);
      else   // per group: dcol[g][Kg][P] = filter[g][Mg][Kg]^T . dY[n][g][Mg][P]
        GemmRowMajorStridedBatched(s, true, false, Kg, P, Mg, 1.0f, filter.data<float>(), Kg,
                                   (long long)Mg * Kg, dyn, P, (long long)Mg * P, 0.0f, dcol, P,
                                   (long long)Kg * P, G);
      if (!pointwise)
        CAFFE_ENFORCE_EQ(ssad_col2im(dcol, C, H, W, kh, kw, geom_.dilation[0], geom_.dilation[1],
                                     geom_.pads[0], geom_.pads[1], geom_.pads[2], geom_.pads[3],
                                     geom_.stride[0], geom_.stride[1], dxn, s), 0);
    }
  }
  if (!no_bias_) {
    auto* dbias = Output(BIAS_OR_INPUT_GRAD);
    dbias->Resize(M);
    CAFFE_ENFORCE_EQ(ssad_channel_sum(dY.data<float>(), N, M, P, dbias->mutable_data<float>(), 0, s), 0);
  }
  return true;
}

template <>
bool ConvGradientOp<float, HIPContext>::RunOnDevice() {
  auto& X = Input(INPUT);
  auto& filter = Input(FILTER);
  auto& dY = Input(OUTPUT_GRAD);
  auto* dfilter = Output(FILTER_GRAD);
  if (X.IsType<float16>()) return RunFloat16();
  CAFFE_ENFORCE_EQ(X.ndim(), 4);
  CAFFE_ENFORCE_EQ(X.ndim(), filter.ndim());
  const int N = X.dim32(0), C = X.dim32(1), H = X.dim32(2), W = X.dim32(3);
  const int M = filter.dim32(0);
  CAFFE_ENFORCE(filter.dim32(1) * geom_.group == C);
  CAFFE_ENFORCE(filter.dim32(2) == geom_.kernel[0] && filter.dim32(3) == geom_.kernel[1]);
  CAFFE_ENFORCE_EQ(dY.ndim(), 4);
  if (!IsSubnetGeometry(geom_)) return RunDefaultEngine();
  CAFFE_ENFORCE(dY.dim32(0) == N && dY.dim32(1) == M && dY.dim32(2) == H && dY.dim32(3) == W,
                "output gradient shape does not match the convolution output");
  dfilter->ResizeLike(filter);
  float* db = nullptr;
  if (!no_bias_) {
    auto* dbias = Output(BIAS_OR_INPUT_GRAD);
    dbias->Resize(M);
    db = dbias->mutable_data<float>();
  }
  hipStream_t s = context_.hip_stream();

  // filter (+ bias) gradient: overwrite, beta = 0 (conv_op_cudnn.cc:1037)
  ssad_conv_level wl{X.data<float>(), nullptr, dY.data<float>(), N, H, W, nullptr, nullptr};
  // hip_algo = "split": the >= 128-wide filter gradients on the split-operand engine (conv3x3_wgrad_split.hip)
  const bool wsplit = algo_ == "split" && M >= 128 && C >= 64;
  const size_t wsb = wsplit ? ssad_conv3x3_wgrad_split_workspace_bytes(&wl, 1, M, C)
                            : ssad_conv3x3_wgrad_workspace_bytes(&wl, 1, M, C);
  workspace_.Resize((TIndex)wsb);
  int rc = (wsplit ? ssad_conv3x3_wgrad_split : ssad_conv3x3_wgrad)(&wl, 1, dfilter->mutable_data<float>(), db, M, C, 0,
                                                                     workspace_.mutable_data<uint8_t>(), wsb, s);
  CAFFE_ENFORCE_EQ(rc, 0, "ConvGradient (filter) launch failed");
  ++g_conv_launch_calls;

  if (OutputSize() == 3 || (no_bias_ && OutputSize() == 2)) {
    auto* dX = Output(no_bias_ ? BIAS_OR_INPUT_GRAD : INPUT_GRAD);
    dX->ResizeLike(X);
    ssad_conv_level dl{dY.data<float>(), dX->mutable_data<float>(),
                       relu_grad_on_input_ ? X.data<float>() : nullptr, N, H, W, nullptr, nullptr};
    const int flags = relu_grad_on_input_ ? SSAD_CONV_MASK_AUX : 0;
    const bool sp = algo_ == "split" && C >= 256;            // trained nets (net_lowering.cc)
    const bool f24 = (algo_ == "winograd24" || algo_ == "split") && C >= 128 && !sp;
    const bool wino = UseWinograd(algo_, C);      // the data gradient has C output channels
    const auto kind = sp ? FilterPackCache::SPLIT_DGRAD : f24 ? FilterPackCache::WINO24_DGRAD
                         : wino ? FilterPackCache::WINO_DGRAD : FilterPackCache::DIRECT_DGRAD;
    const long long before = pack_cache_.packs_issued();
    pack_cache_.Want(filter, kind);
    pack_cache_.Flush(s);
    g_filter_packs_issued += pack_cache_.packs_issued() - before;
    const float* packed = pack_cache_.Packed(filter, kind);
    rc = sp ? split_.Run(&dl, 1, packed, nullptr, C, M, flags, s)
         : f24 ? ssad_conv3x3_forward_wino24(&dl, 1, packed, nullptr, C, M, flags, s)
         : wino ? ssad_conv3x3_forward_wino(&dl, 1, packed, nullptr, C, M, flags, s)
                : ssad_conv3x3_forward(&dl, 1, packed, nullptr, C, M, flags, s);
    CAFFE_ENFORCE_EQ(rc, 0, "ConvGradient (data) launch failed");
    ++g_conv_launch_calls;
  }
  return true;
}

// float16 blobs, forward: X, filter, (bias) and Y in fp16, accumulation in fp32.
template <>
bool ConvOp<float, HIPContext>::RunFloat16() {
  auto& X = Input(INPUT);
  auto& filter = Input(FILTER);
  auto* Y = Output(0);
  if (IsPointwiseF16Geometry(geom_)) return RunFloat16Pointwise();
  CAFFE_ENFORCE(IsSubnetGeometry(geom_),
                "float16 Conv: the HIP engines implement kernel 3 / stride 1 / pad 1 and kernel 1 / stride 1|2 / "
                "pad 0 (group 1, NCHW)");
  CAFFE_ENFORCE(filter.IsType<float16>(), "float16 Conv: the filter must be float16 too");
  CAFFE_ENFORCE_EQ(X.ndim(), 4);
  CAFFE_ENFORCE_EQ(filter.ndim(), 4);
  const int N = X.dim32(0), C = X.dim32(1), H = X.dim32(2), W = X.dim32(3);
  const int M = filter.dim32(0);
  CAFFE_ENFORCE(C == filter.dim32(1) && filter.dim32(2) == 3 && filter.dim32(3) == 3);
  hipStream_t s = context_.hip_stream();
  // filter: fp16 [M][C][3][3] -> fp32 -> packed fp16; bias -> fp32
  auto& wf32 = f16_scratch_[0];
  wf32.Resize((TIndex)filter.size());
  CAFFE_ENFORCE_EQ(ssad_cast_f16_to_f32(filter.raw_data(), wf32.mutable_data<float>(), filter.size(), s), 0);
  packed_filter_.Resize((TIndex)((ssad_f16_filter_halves(M, C) + 1) / 2));
  float* packed = packed_filter_.mutable_data<float>();
  CAFFE_ENFORCE_EQ(ssad_f16_pack_filter(wf32.data<float>(), M, C, packed, nullptr, s), 0);
  const float* bias = nullptr;
  if (InputSize() == 3) {
    auto& b = Input(BIAS);
    CAFFE_ENFORCE(b.IsType<float16>() && b.ndim() == 1 && b.dim32(0) == M);
    auto& b32 = f16_scratch_[1];
    b32.Resize(M);
    CAFFE_ENFORCE_EQ(ssad_cast_f16_to_f32(b.raw_data(), b32.mutable_data<float>(), M, s), 0);
    bias = b32.data<float>();
  }
  // activations: NCHW fp16 -> channel-blocked -> convolution -> NCHW fp16
  auto& xb = f16_scratch_[2];
  xb.Resize((TIndex)((BlockedHalves(N, C, H, W) + 1) / 2));
  CAFFE_ENFORCE_EQ(ssad_f16_block_activations(X.raw_data(), N, C, H, W, xb.mutable_data<float>(), s), 0);
  Y->Resize(N, M, H, W);
  void* y = Y->raw_mutable_data(TypeMeta::Make<float16>());
  const int flags = fuse_relu_ ? SSAD_CONV_RELU : 0;
  if (M % 8 == 0) {
    auto& yb = f16_scratch_[3];
    yb.Resize((TIndex)((BlockedHalves(N, M, H, W) + 1) / 2));
    CAFFE_ENFORCE_EQ(ssad_conv3x3_forward_f16(xb.data<float>(), packed, bias, nullptr, N, C, H, W, M, flags,
                                              yb.mutable_data<float>(), s), 0, "float16 Conv launch failed");
    CAFFE_ENFORCE_EQ(ssad_f16_unblock_activations(yb.data<float>(), N, M, H, W, y, s), 0);
  } else {   // an output width that is not a multiple of 8 (bbox_pred: 36) leaves through fp32
    auto& y32 = f16_scratch_[3];
    y32.Resize((TIndex)Y->size());
    CAFFE_ENFORCE_EQ(ssad_conv3x3_forward_f16(xb.data<float>(), packed, bias, nullptr, N, C, H, W, M,
                                              flags | SSAD_F16_OUT_NCHW_F32, y32.mutable_data<float>(), s),
                     0, "float16 Conv launch failed");
    CAFFE_ENFORCE_EQ(ssad_cast_f32_to_f16(y32.data<float>(), y, Y->size(), s), 0);
  }
  return true;
}

// float16 blobs, gradient: X, filter, dY in fp16 -> dfilter, dbias, dX in fp16; sums in fp32.
template <>
bool ConvGradientOp<float, HIPContext>::RunFloat16() {
  auto& X = Input(INPUT);
  auto& filter = Input(FILTER);
  auto& dY = Input(OUTPUT_GRAD);
  auto* dfilter = Output(FILTER_GRAD);
  if (IsPointwiseF16Geometry(geom_)) return RunFloat16Pointwise();
  CAFFE_ENFORCE(IsSubnetGeometry(geom_),
                "float16 ConvGradient: the HIP engines implement kernel 3 / stride 1 / pad 1 and kernel 1 / "
                "stride 1|2 / pad 0 (group 1, NCHW)");
  CAFFE_ENFORCE(filter.IsType<float16>() && dY.IsType<float16>(),
                "float16 ConvGradient: filter and output gradient must be float16 too");
  CAFFE_ENFORCE(!relu_grad_on_input_, "relu_grad_on_input is an extension of the fp32 3x3 engine");
  const int N = X.dim32(0), C = X.dim32(1), H = X.dim32(2), W = X.dim32(3);
  const int M = filter.dim32(0);
  CAFFE_ENFORCE(C == filter.dim32(1) && filter.dim32(2) == 3 && filter.dim32(3) == 3);
  CAFFE_ENFORCE(dY.ndim() == 4 && dY.dim32(0) == N && dY.dim32(1) == M && dY.dim32(2) == H &&
                    dY.dim32(3) == W,
                "output gradient shape does not match the convolution output");
  hipStream_t s = context_.hip_stream();
  auto& xb = f16_scratch_[0];
  auto& dyb = f16_scratch_[1];
  xb.Resize((TIndex)((BlockedHalves(N, C, H, W) + 1) / 2));
  dyb.Resize((TIndex)((BlockedHalves(N, M, H, W) + 1) / 2));
  CAFFE_ENFORCE_EQ(ssad_f16_block_activations(X.raw_data(), N, C, H, W, xb.mutable_data<float>(), s), 0);
  CAFFE_ENFORCE_EQ(ssad_f16_block_activations(dY.raw_data(), N, M, H, W, dyb.mutable_data<float>(), s), 0);
  // filter / bias gradient in fp32, stored as fp16 (conv_op_cudnn.cc:1037: overwrite)
  auto& dw32 = f16_scratch_[2];
  auto& db32 = f16_scratch_[3];
  dw32.Resize((TIndex)filter.size());
  db32.Resize(M);
  const size_t wsb = ssad_conv3x3_wgrad_f16_workspace_bytes(N, C, H, W, M);
  workspace_.Resize((TIndex)wsb);
  CAFFE_ENFORCE_EQ(ssad_conv3x3_wgrad_f16(xb.data<float>(), dyb.data<float>(), N, C, H, W, M, 0, 1.0f,
                                          dw32.mutable_data<float>(),
                                          no_bias_ ? nullptr : db32.mutable_data<float>(),
                                          workspace_.mutable_data<uint8_t>(), wsb, s),
                   0, "float16 ConvGradient (filter) launch failed");
  dfilter->ResizeLike(filter);
  CAFFE_ENFORCE_EQ(ssad_cast_f32_to_f16(dw32.data<float>(), dfilter->raw_mutable_data(TypeMeta::Make<float16>()),
                                        filter.size(), s), 0);
  if (!no_bias_) {
    auto* dbias = Output(BIAS_OR_INPUT_GRAD);
    dbias->Resize(M);
    CAFFE_ENFORCE_EQ(ssad_cast_f32_to_f16(db32.data<float>(), dbias->raw_mutable_data(TypeMeta::Make<float16>()),
                                          M, s), 0);
  }
  if (OutputSize() == 3 || (no_bias_ && OutputSize() == 2)) {
    auto* dX = Output(no_bias_ ? BIAS_OR_INPUT_GRAD : INPUT_GRAD);
    dX->ResizeLike(X);
    auto& wf32 = f16_scratch_[4];
    wf32.Resize((TIndex)filter.size());
    CAFFE_ENFORCE_EQ(ssad_cast_f16_to_f32(filter.raw_data(), wf32.mutable_data<float>(), filter.size(), s), 0);
    packed_filter_.Resize((TIndex)((ssad_f16_filter_halves(M, C) + 1) / 2));
    float* packed = packed_filter_.mutable_data<float>();
    CAFFE_ENFORCE_EQ(ssad_f16_pack_filter(wf32.data<float>(), M, C, nullptr, packed, s), 0);
    void* dx = dX->raw_mutable_data(TypeMeta::Make<float16>());
    auto& out = f16_scratch_[5];
    if (C % 8 == 0) {
      out.Resize((TIndex)((BlockedHalves(N, C, H, W) + 1) / 2));
      CAFFE_ENFORCE_EQ(ssad_conv3x3_forward_f16(dyb.data<float>(), packed, nullptr, nullptr, N, M, H, W, C, 0,
                                                out.mutable_data<float>(), s), 0,
                       "float16 ConvGradient (data) launch failed");
      CAFFE_ENFORCE_EQ(ssad_f16_unblock_activations(out.data<float>(), N, C, H, W, dx, s), 0);
    } else {
      out.Resize((TIndex)X.size());
      CAFFE_ENFORCE_EQ(ssad_conv3x3_forward_f16(dyb.data<float>(), packed, nullptr, nullptr, N, M, H, W, C,
                                                SSAD_F16_OUT_NCHW_F32, out.mutable_data<float>(), s), 0,
                       "float16 ConvGradient (data) launch failed");
      CAFFE_ENFORCE_EQ(ssad_cast_f32_to_f16(out.data<float>(), dx, X.size(), s), 0);
    }
  }
  return true;
}

// float16 blobs, pointwise layer (the backbones' 1x1 convolutions under CudnnConvOp<float16>,
// conv_op_cudnn.cc:631-636): NCHW fp16 -> channel-blocked -> pw_f16_kernel -> NCHW fp16.
template <>
bool ConvOp<float, HIPContext>::RunFloat16Pointwise() {
  auto& X = Input(INPUT);
  auto& filter = Input(FILTER);
  auto* Y = Output(0);
  CAFFE_ENFORCE(filter.IsType<float16>(), "float16 Conv: the filter must be float16 too");
  CAFFE_ENFORCE_EQ(X.ndim(), 4);
  CAFFE_ENFORCE_EQ(filter.ndim(), 4);
  const int N = X.dim32(0), C = X.dim32(1), H = X.dim32(2), W = X.dim32(3);
  const int M = filter.dim32(0), st = geom_.stride[0];
  CAFFE_ENFORCE(C == filter.dim32(1) && filter.dim32(2) == 1 && filter.dim32(3) == 1);
  CAFFE_ENFORCE(M % 8 == 0, "float16 pointwise Conv: the output width must be a multiple of 8 channels");
  const int OH = (H - 1) / st + 1, OW = (W - 1) / st + 1;
  hipStream_t s = context_.hip_stream();
  auto& wf32 = f16_scratch_[0];
  wf32.Resize((TIndex)filter.size());
  CAFFE_ENFORCE_EQ(ssad_cast_f16_to_f32(filter.raw_data(), wf32.mutable_data<float>(), filter.size(), s), 0);
  packed_filter_.Resize((TIndex)((ssad_pw_f16_filter_halves(M, C) + 1) / 2));
  float* packed = packed_filter_.mutable_data<float>();
  CAFFE_ENFORCE_EQ(ssad_pw_f16_pack_filter(wf32.data<float>(), M, C, packed, nullptr, s), 0);
  const float* bias = nullptr;
  if (InputSize() == 3) {
    auto& b = Input(BIAS);
    CAFFE_ENFORCE(b.IsType<float16>() && b.ndim() == 1 && b.dim32(0) == M);
    auto& b32 = f16_scratch_[1];
    b32.Resize(M);
    CAFFE_ENFORCE_EQ(ssad_cast_f16_to_f32(b.raw_data(), b32.mutable_data<float>(), M, s), 0);
    bias = b32.data<float>();
  }
  auto& xb = f16_scratch_[2];
  auto& yb = f16_scratch_[3];
  xb.Resize((TIndex)((BlockedHalves(N, C, H, W) + 1) / 2));
  yb.Resize((TIndex)((BlockedHalves(N, M, OH, OW) + 1) / 2));
  CAFFE_ENFORCE_EQ(ssad_f16_block_activations(X.raw_data(), N, C, H, W, xb.mutable_data<float>(), s), 0);
  const ssad_pw_f16 d{xb.data<float>(), packed, bias, nullptr, nullptr, yb.mutable_data<float>(),
                      N, C, M, OH, OW, H, W, st, fuse_relu_ ? SSAD_CONV_RELU : 0};
  CAFFE_ENFORCE_EQ(ssad_conv1x1_f16(&d, s), 0, "float16 pointwise Conv launch failed");
  Y->Resize(N, M, OH, OW);
  CAFFE_ENFORCE_EQ(ssad_f16_unblock_activations(yb.data<float>(), N, M, OH, OW,
                                                Y->raw_mutable_data(TypeMeta::Make<float16>()), s), 0);
  return true;
}

// float16 blobs, pointwise layer, gradient (conv_op_impl.h:451-560 for a 1x1 kernel): dfilter / dbias
// summed in fp32 and stored as fp16, dX = W^T dY; a stride-2 layer is the stride-1 layer on the
// subsampled input, its dX scattered back to the even positions.
template <>
bool ConvGradientOp<float, HIPContext>::RunFloat16Pointwise() {
  auto& X = Input(INPUT);
  auto& filter = Input(FILTER);
  auto& dY = Input(OUTPUT_GRAD);
  auto* dfilter = Output(FILTER_GRAD);
  CAFFE_ENFORCE(filter.IsType<float16>() && dY.IsType<float16>(),
                "float16 ConvGradient: filter and output gradient must be float16 too");
  CAFFE_ENFORCE(!relu_grad_on_input_, "relu_grad_on_input is an extension of the fp32 3x3 engine");
  const int N = X.dim32(0), C = X.dim32(1), H = X.dim32(2), W = X.dim32(3);
  const int M = filter.dim32(0), st = geom_.stride[0];
  CAFFE_ENFORCE(C == filter.dim32(1) && filter.dim32(2) == 1 && filter.dim32(3) == 1);
  const int OH = (H - 1) / st + 1, OW = (W - 1) / st + 1;
  CAFFE_ENFORCE(dY.ndim() == 4 && dY.dim32(0) == N && dY.dim32(1) == M && dY.dim32(2) == OH && dY.dim32(3) == OW,
                "output gradient shape does not match the convolution output");
  hipStream_t s = context_.hip_stream();
  auto& xb = f16_scratch_[0];
  auto& dyb = f16_scratch_[1];
  xb.Resize((TIndex)((BlockedHalves(N, C, H, W) + 1) / 2));
  dyb.Resize((TIndex)((BlockedHalves(N, M, OH, OW) + 1) / 2));
  CAFFE_ENFORCE_EQ(ssad_f16_block_activations(X.raw_data(), N, C, H, W, xb.mutable_data<float>(), s), 0);
  CAFFE_ENFORCE_EQ(ssad_f16_block_activations(dY.raw_data(), N, M, OH, OW, dyb.mutable_data<float>(), s), 0);
  const float* xs = xb.data<float>();
  if (st > 1) {                                 // the strided layer's view of its input (odd maps included)
    auto& sub = f16_scratch_[6];
    sub.Resize((TIndex)((BlockedHalves(N, C, OH, OW) + 1) / 2));
    CAFFE_ENFORCE_EQ(ssad_f16_subsample(xb.data<float>(), N, C, H, W, st, sub.mutable_data<float>(), s), 0);
    xs = sub.data<float>();
  }
  auto& dw32 = f16_scratch_[2];
  auto& db32 = f16_scratch_[3];
  dw32.Resize((TIndex)filter.size());
  db32.Resize(M);
  const size_t wsb = ssad_conv1x1_wgrad_f16_workspace_bytes(N, C, OH, OW, M);
  workspace_.Resize((TIndex)wsb);
  CAFFE_ENFORCE_EQ(ssad_conv1x1_wgrad_f16(xs, dyb.data<float>(), N, C, OH, OW, M, 0, 1.0f, nullptr,
                                          dw32.mutable_data<float>(), no_bias_ ? nullptr : db32.mutable_data<float>(),
                                          workspace_.mutable_data<uint8_t>(), wsb, s),
                   0, "float16 pointwise ConvGradient (filter) launch failed");
  dfilter->ResizeLike(filter);
  CAFFE_ENFORCE_EQ(ssad_cast_f32_to_f16(dw32.data<float>(), dfilter->raw_mutable_data(TypeMeta::Make<float16>()),
                                        filter.size(), s), 0);
  if (!no_bias_) {
    auto* dbias = Output(BIAS_OR_INPUT_GRAD);
    dbias->Resize(M);
    CAFFE_ENFORCE_EQ(ssad_cast_f32_to_f16(db32.data<float>(), dbias->raw_mutable_data(TypeMeta::Make<float16>()),
                                          M, s), 0);
  }
  if (OutputSize() == 3 || (no_bias_ && OutputSize() == 2)) {
    CAFFE_ENFORCE(C % 8 == 0, "float16 pointwise ConvGradient: the input width must be a multiple of 8 channels");
    auto* dX = Output(no_bias_ ? BIAS_OR_INPUT_GRAD : INPUT_GRAD);
    dX->ResizeLike(X);
    auto& wf32 = f16_scratch_[4];
    wf32.Resize((TIndex)filter.size());
    CAFFE_ENFORCE_EQ(ssad_cast_f16_to_f32(filter.raw_data(), wf32.mutable_data<float>(), filter.size(), s), 0);
    packed_filter_.Resize((TIndex)((ssad_pw_f16_filter_halves(M, C) + 1) / 2));
    float* packed = packed_filter_.mutable_data<float>();
    CAFFE_ENFORCE_EQ(ssad_pw_f16_pack_filter(wf32.data<float>(), M, C, nullptr, packed, s), 0);
    auto& dxs = f16_scratch_[5];
    dxs.Resize((TIndex)((BlockedHalves(N, C, OH, OW) + 1) / 2));
    const ssad_pw_f16 d{dyb.data<float>(), packed, nullptr, nullptr, nullptr, dxs.mutable_data<float>(),
                        N, M, C, OH, OW, OH, OW, 1, 0};
    CAFFE_ENFORCE_EQ(ssad_conv1x1_f16(&d, s), 0, "float16 pointwise ConvGradient (data) launch failed");
    const float* full = dxs.data<float>();
    if (st > 1) {
      auto& up = f16_scratch_[7];
      up.Resize((TIndex)((BlockedHalves(N, C, H, W) + 1) / 2));
      CAFFE_ENFORCE_EQ(ssad_f16_elementwise(1, dxs.data<float>(), nullptr, up.mutable_data<float>(), N, C, H, W, st,
                                            0, s), 0);
      full = up.data<float>();
    }
    CAFFE_ENFORCE_EQ(ssad_f16_unblock_activations(full, N, C, H, W, dX->raw_mutable_data(TypeMeta::Make<float16>()),
                                                  s), 0);
  }
  return true;
}

REGISTER_HIP_OPERATOR(Conv, ConvOp<float, HIPContext>);
REGISTER_HIP_OPERATOR(ConvGradient, ConvGradientOp<float, HIPContext>);

OPERATOR_SCHEMA(Conv)
    .NumInputs(2, 3)
    .NumOutputs(1)
    .SetDoc("2-D convolution (cross-correlation) Y = W * X + b.")
    .Input(0, "X", "Input data blob, NCHW.")
    .Input(1, "filter", "The filter blob, M x C x kH x kW.")
    .Input(2, "bias", "The 1D bias blob of length M.")
    .Output(0, "Y", "Output data blob.");
OPERATOR_SCHEMA(ConvGradient).NumInputs(2, 3).NumOutputs(1, 3);

// caffe2/operators/conv_gradient_op.cc:35-77
class GetConvGradient : public GradientMakerBase {
  using GradientMakerBase::GradientMakerBase;
  vector<OperatorDef> GetGradientDefs() override {
    CAFFE_ENFORCE(def_.input.size() == 3 || def_.input.size() == 2);
    vector<Argument> args = def_.arg;
    // the fusion extension of the forward op does not transfer to the gradient
    for (auto it = args.begin(); it != args.end();)
      it = (it->name == "fuse_relu") ? args.erase(it) : it + 1;
    if (def_.input.size() == 3) {
      return SingleGradientDef("ConvGradient", "", vector<string>{I(0), I(1), GO(0)},
                               vector<string>{GI(1), GI(2), GI(0)}, args);
    }
    args.push_back(MakeArgument("no_bias", 1));
    return SingleGradientDef("ConvGradient", "", vector<string>{I(0), I(1), GO(0)},
                             vector<string>{GI(1), GI(0)}, args);
  }
};
REGISTER_GRADIENT(Conv, GetConvGradient);

}  // namespace caffe2
