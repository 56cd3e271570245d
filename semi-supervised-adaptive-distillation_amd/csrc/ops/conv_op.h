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
// The HIP engine implements the geometry the RetinaNet subnets use
// (3x3, stride 1, pad 1, dilation 1, group 1, NCHW) on the matrix cores; any
// other geometry raises UnsupportedOperatorFeature at construction, as an
// engine that cannot serve a definition does in the reference
// (caffe2/core/operator.h:765-782).
//
// Two optional arguments exist for graph-level fusion and are NOT emitted by
// the reference graph builder (absent = reference behaviour):
//   Conv          fuse_relu=1             Y = max(conv, 0)  (Conv + in-place Relu)
//   ConvGradient  relu_grad_on_input=1    dX masked by X > 0 (= ReluGradient of
//                                         the in-place Relu that produced X)
//   both          hip_algo="direct"|"winograd"   pin the algorithm (default: by width)
#ifndef C2HIP_CONV_OP_H_
#define C2HIP_CONV_OP_H_

#include "c2/operator.h"

namespace caffe2 {

struct ConvGeometry {
  vector<int> kernel, stride, pads, dilation;
  int group = 1;
  string order = "NCHW";
};

// Parses and validates the convolution arguments exactly as the base class of
// the reference does; shared by Conv and ConvGradient.
ConvGeometry ParseConvGeometry(const OperatorBase& op);
bool IsSubnetGeometry(const ConvGeometry& g);   // 3x3 / s1 / p1 / d1 / g1 / NCHW
bool UseWinograd(const string& algo, int out_channels);

template <typename T, class Context>
class ConvOp final : public Operator<Context> {
 public:
  USE_OPERATOR_CONTEXT_FUNCTIONS;
  ConvOp(const OperatorDef& def, Workspace* ws)
      : Operator<Context>(def, ws),
        geom_(ParseConvGeometry(*this)),
        fuse_relu_(OperatorBase::GetSingleArgument<int>("fuse_relu", 0)),
        algo_(OperatorBase::GetSingleArgument<string>("hip_algo", "auto")) {
    if (!IsSubnetGeometry(geom_))
      throw UnsupportedOperatorFeature(
          "HIP Conv engine implements kernel=3 stride=1 pad=1 dilation=1 group=1 NCHW only");
  }
  bool RunOnDevice() override;

 private:
  ConvGeometry geom_;
  int fuse_relu_;
  string algo_;
  Tensor<Context> packed_filter_;
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
    if (!IsSubnetGeometry(geom_))
      throw UnsupportedOperatorFeature(
          "HIP ConvGradient engine implements kernel=3 stride=1 pad=1 dilation=1 group=1 NCHW only");
  }
  bool RunOnDevice() override;

 private:
  ConvGeometry geom_;
  bool no_bias_;
  int relu_grad_on_input_;
  string algo_;
  Tensor<Context> packed_filter_;
  Tensor<Context> workspace_;
};

}  // namespace caffe2
#endif
