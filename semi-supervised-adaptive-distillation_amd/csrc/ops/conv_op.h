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
// construction (caffe2/core/operator.h:765-782).  The operator packs its filter on
// every RunOnDevice (the filter blob may have been updated in between, as under
// training); the fused step (head_pipeline) packs once per parameter update instead.
//
// float16 blobs (TensorProto::FLOAT16) dispatch like CudnnConvOp's DoRunWithType<float16, ...>
// (conv_op_cudnn.cc:631-636, :1115-1124): fp16 storage for X, filter, bias, Y and all
// gradients, fp32 math -- on the fp16 matrix-core kernels of kernels/conv3x3_f16.hip, for the
// 3x3 / stride 1 / pad 1 geometry.
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
  string algo_;
  Tensor<Context> packed_filter_;
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
  Tensor<Context> workspace_;
  Tensor<Context> col_buffer_;
  Tensor<Context> f16_scratch_[8];
  bool RunFloat16Pointwise();
};

}  // namespace caffe2
#endif
