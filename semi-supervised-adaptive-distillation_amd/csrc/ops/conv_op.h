// Conv / ConvGradient for HIPContext.
//
// Argument handling follows ConvPoolOpBase (caffe2/operators/
// conv_pool_op_base.h:45-194): `kernel` | `kernel_h`+`kernel_w` | `kernels`,
// likewise stride(s), pad(s) / pad_t..pad_r, dilation(s), `group`, `order`,
// `legacy_pad`.  Inputs/outputs follow conv_op.h:29-91 and
// conv_gradient_op.cc:35-77: Conv [X, filter, (bias)] -> [Y];
// ConvGradient [X, filter, dY] -> [dfilter, dbias, (dX)] or, with
// no_bias=1, [dfilter, (dX)].
//
// Two engines, chosen per definition like the reference's engine fall-through
// (caffe2/core/operator.cc:116-200): the geometry the RetinaNet subnets and the
// ResNet bottlenecks use (3x3, stride 1, pad 1, dilation 1, group 1, NCHW) runs on
// the matrix-core kernels of this repo; every other 2-D NCHW geometry (the
// backbone's 1x1, strided and 7x7 layers, and `group` > 1 -- ResNeXt's grouped 3x3 --
// as one strided-batched GEMM per image) runs on the DEFAULT engine of
// conv_op_impl.h:31-202 / :358-577 -- im2col + GEMM per image, with the GEMMs on this
// repo's general fp32-MFMA kernel (kernels/gemm_general.hip; no vendor BLAS).  NHWC and non-2-D convolutions raise UnsupportedOperatorFeature at
// construction (caffe2/core/operator.h:765-782).  The 3x3 engines' packed filter is
// cached per operator and rebuilt only when the filter blob was written since the last Run
// (ops/filter_pack_cache.h: Tensor::version()) -- round 5; rounds 1-4 repacked on every Run.
//
// float16 blobs (TensorProto::FLOAT16) dispatch like CudnnConvOp's DoRunWithType<float16, ...>
// (conv_op_cudnn.cc:631-636, :1115-1124): fp16 storage for X, filter, bias, Y and all
// gradients, fp32 math -- on the fp16 matrix-core kernels of kernels/conv3x3_f16.hip, for the
// 3x3 / stride 1 / pad 1 geometry.
//
// Two optional arguments exist for graph-level fusion and are NOT emitted by
// the reference graph builder (absent = reference behaviour):
//   Conv          fuse_relu=1             Y = max(conv, 0)  (Conv + in-place Relu)
//   Conv          fuse_sigmoid=1          Y = 1 / (1 + exp(-conv))  (Conv + Sigmoid; the 3x3 fp32 engines)
//   ConvGradient  relu_grad_on_input=1    dX masked by X > 0 (= ReluGradient of
//                                         the in-place Relu that produced X)
//   both          hip_algo="direct"|"winograd"   pin the algorithm (default: by width)
#ifndef C2HIP_CONV_OP_H_
#define C2HIP_CONV_OP_H_

#include <atomic>

#include "c2/operator.h"
#include "ops/filter_pack_cache.h"

namespace caffe2 {

// process-wide counters behind c2hip_counter() (tests: the pack cache and the net lowering at work)
extern std::atomic<long long> g_filter_packs_issued;
extern std::atomic<long long> g_conv_launch_calls;

struct ConvGeometry {
  vector<int> kernel, stride, pads, dilation;
  int group = 1;
  string order = "NCHW";
};

// The argument accessors of OperatorBase over a bare definition: lets the net lowering
// (ops/net_lowering.cc) read a convolution's geometry before any operator exists.
class DefArgs {
 public:
  explicit DefArgs(const OperatorDef& def) {
    for (const Argument& a : def.arg) args_[a.name] = &a;
  }
  bool HasArgument(const string& n) const { return args_.count(n) != 0; }
  template <typename T> T GetSingleArgument(const string& n, const T& dflt) const;
  template <typename T> vector<T> GetRepeatedArgument(const string& n, const vector<T>& dflt = {}) const;

 private:
  std::map<string, const Argument*> args_;
};
template <> inline int DefArgs::GetSingleArgument<int>(const string& n, const int& d) const {
  auto it = args_.find(n);
  if (it == args_.end()) return d;
  CAFFE_ENFORCE(it->second->has_i, "Argument ", n, " does not have an integer value");
  return (int)it->second->i;
}
template <> inline string DefArgs::GetSingleArgument<string>(const string& n, const string& d) const {
  auto it = args_.find(n);
  if (it == args_.end()) return d;
  CAFFE_ENFORCE(it->second->has_s, "Argument ", n, " does not have a string value");
  return it->second->s;
}
template <> inline vector<int> DefArgs::GetRepeatedArgument<int>(const string& n, const vector<int>& d) const {
  auto it = args_.find(n);
  if (it == args_.end()) return d;
  return vector<int>(it->second->ints.begin(), it->second->ints.end());
}

// caffe2/operators/conv_pool_op_base.h:45-194, restated.
template <class ArgSource>
ConvGeometry ParseConvGeometryFrom(const ArgSource& op) {
  ConvGeometry g;
  g.kernel = op.template GetRepeatedArgument<int>("kernels");
  g.stride = op.template GetRepeatedArgument<int>("strides");
  g.pads = op.template GetRepeatedArgument<int>("pads");
  g.dilation = op.template GetRepeatedArgument<int>("dilations");
  g.group = op.template GetSingleArgument<int>("group", 1);
  g.order = op.template GetSingleArgument<string>("order", "NCHW");
  const int legacy_pad = op.template GetSingleArgument<int>("legacy_pad", 0);   // NOTSET
  CAFFE_ENFORCE(legacy_pad == 0 || legacy_pad == 3,
                "legacy padding VALID/SAME is not supported by the HIP conv operators");

  auto pair_arg = [&](const char* single, const char* h, const char* w, vector<int>* out) {
    if (op.HasArgument(single)) {
      out->assign(2, op.template GetSingleArgument<int>(single, 0));
    } else if (op.HasArgument(h) && op.HasArgument(w)) {
      out->push_back(op.template GetSingleArgument<int>(h, 0));
      out->push_back(op.template GetSingleArgument<int>(w, 0));
    }
  };
  pair_arg("kernel", "kernel_h", "kernel_w", &g.kernel);
  pair_arg("stride", "stride_h", "stride_w", &g.stride);
  pair_arg("dilation", "dilation_h", "dilation_w", &g.dilation);
  if (op.HasArgument("pad")) {
    g.pads.assign(4, op.template GetSingleArgument<int>("pad", 0));
  } else if (op.HasArgument("pad_t") && op.HasArgument("pad_l") && op.HasArgument("pad_b") &&
             op.HasArgument("pad_r")) {
    g.pads = {op.template GetSingleArgument<int>("pad_t", 0), op.template GetSingleArgument<int>("pad_l", 0),
              op.template GetSingleArgument<int>("pad_b", 0), op.template GetSingleArgument<int>("pad_r", 0)};
  }
  if (g.kernel.empty()) g.kernel.assign(2, 0);
  if (g.stride.empty()) g.stride.assign(g.kernel.size(), 1);
  if (g.pads.empty()) g.pads.assign(g.kernel.size() * 2, 0);
  if (g.dilation.empty()) g.dilation.assign(g.kernel.size(), 1);
  CAFFE_ENFORCE_EQ(g.stride.size(), g.kernel.size());
  CAFFE_ENFORCE_EQ(g.dilation.size(), g.kernel.size());
  CAFFE_ENFORCE_EQ(g.pads.size(), 2 * g.kernel.size());
  for (size_t d = 0; d < g.kernel.size(); ++d) {
    CAFFE_ENFORCE_GE(g.pads[d], 0);
    CAFFE_ENFORCE_GE(g.pads[g.kernel.size() + d], 0);
    CAFFE_ENFORCE(g.kernel[d],
                  "If you are doing convolution or pooling, you will need to set explicitly "
                  "the kernel size.");
    CAFFE_ENFORCE_GE(g.dilation[d], 0);
    CAFFE_ENFORCE_GE(g.stride[d], 0);
  }
  CAFFE_ENFORCE(g.order == "NCHW" || g.order == "NHWC", "Unknown storage order: ", g.order);
  return g;
}


// Parses and validates the convolution arguments exactly as the base class of
// the reference does; shared by Conv and ConvGradient (and, over a bare definition, by the net lowering).
ConvGeometry ParseConvGeometry(const OperatorBase& op);
ConvGeometry ParseConvGeometry(const OperatorDef& def);
bool IsSubnetGeometry(const ConvGeometry& g);   // 3x3 / s1 / p1 / d1 / g1 / NCHW
bool IsDefaultEngineGeometry(const ConvGeometry& g);   // any 2-D NCHW geometry (group >= 1)
bool UseWinograd(const string& algo, int out_channels);

template <typename T, class Context>
class ConvOp final : public Operator<Context> {
 public:
  USE_OPERATOR_CONTEXT_FUNCTIONS;
  ConvOp(const OperatorDef& def, Workspace* ws)
      : Operator<Context>(def, ws),
        geom_(ParseConvGeometry(*this)),
        fuse_relu_(OperatorBase::GetSingleArgument<int>("fuse_relu", 0)),
        fuse_sigmoid_(OperatorBase::GetSingleArgument<int>("fuse_sigmoid", 0)),
        algo_(OperatorBase::GetSingleArgument<string>("hip_algo", "auto")) {
    if (!IsDefaultEngineGeometry(geom_))
      throw UnsupportedOperatorFeature("HIP Conv engines implement order=NCHW, 2-D only");
  }
  bool RunOnDevice() override;

 private:
  bool RunDefaultEngine();
  bool RunFloat16();
  ConvGeometry geom_;
  int fuse_relu_;
  int fuse_sigmoid_;                    // Y = sigmoid(conv) (set by the net lowering: Conv -> Sigmoid; 3x3 fp32 path only)
  string algo_;
  Tensor<Context> packed_filter_;
  FilterPackCache pack_cache_;
  SplitEngine split_;
  Tensor<Context> col_buffer_;
  Tensor<Context> f16_scratch_[4];
  bool RunFloat16Pointwise();
};

template <typename T, class Context>
class ConvGradientOp final : public Operator<Context> {
 public:
  USE_OPERATOR_CONTEXT_FUNCTIONS;
  ConvGradientOp(const OperatorDef& def, Workspace* ws)
      : Operator<Context>(def, ws),
        geom_(ParseConvGeometry(*this)),
        no_bias_(OperatorBase::GetSingleArgument<int>("no_bias", 0)),
        relu_grad_on_input_(OperatorBase::GetSingleArgument<int>("relu_grad_on_input", 0)),
        algo_(OperatorBase::GetSingleArgument<string>("hip_algo", "auto")) {
    CAFFE_ENFORCE(!(no_bias_ && OutputSize() == 3),
                  "If bias is not present, you should not have 3 grad output.");
    if (!IsDefaultEngineGeometry(geom_))
      throw UnsupportedOperatorFeature(
          "HIP ConvGradient engines implement order=NCHW, 2-D only");
  }
  bool RunOnDevice() override;

 private:
  ConvGeometry geom_;
  bool no_bias_;
  int relu_grad_on_input_;
  string algo_;
  bool RunDefaultEngine();
  bool RunFloat16();
  Tensor<Context> packed_filter_;
  FilterPackCache pack_cache_;
  SplitEngine split_;
  Tensor<Context> workspace_;
  Tensor<Context> col_buffer_;
  Tensor<Context> f16_scratch_[8];
  bool RunFloat16Pointwise();
};

}  // namespace caffe2
#endif
