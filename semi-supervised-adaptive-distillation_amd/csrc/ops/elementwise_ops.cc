// The small stock operators the hot path runs through, for HIPContext:
//   Relu / ReluGradient        caffe2/operators/relu_op.cu:38-66
//   Sigmoid                    caffe2/operators/sigmoid_op.cu:25-29
//   Sum                        caffe2/operators/utility_ops.h (SumOp), used by
//                              the autograd accumulation core.py:706-741
//   Scale                      caffe2/operators/scale_op.h
//   WeightedSum                caffe2/operators/utility_ops.h (WeightedSumOp)
//   ConstantFill (float)       caffe2/operators/filler_op.h
//   MomentumSGDUpdate          caffe2/sgd/momentum_sgd_op.h:97-131
#include "c2/operator.h"
#include "ops/conv_op.h"
#include "ssad_kernels.h"

namespace caffe2 {

#define LAUNCH_OK(call, what) CAFFE_ENFORCE_EQ((call), 0, what, " launch failed")

class ReluHIPOp final : public Operator<HIPContext> {
 public:
  using Operator<HIPContext>::Operator;
  bool RunOnDevice() override {
    auto& X = Input(0);
    auto* Y = Output(0);
    CAFFE_ENFORCE_GT(X.size(), 0);
    Y->ResizeLike(X);
    LAUNCH_OK(ssad_relu(X.data<float>(), Y->mutable_data<float>(), X.size(),
                        context_.hip_stream()), "Relu");
    return true;
  }
};

class ReluGradientHIPOp final : public Operator<HIPContext> {
 public:
  using Operator<HIPContext>::Operator;
  bool RunOnDevice() override {
    auto& Y = Input(0);
    auto& dY = Input(1);
    auto* dX = Output(0);
    CAFFE_ENFORCE_GT(Y.size(), 0);
    CAFFE_ENFORCE_EQ(dY.size(), Y.size());
    dX->ResizeLike(Y);
    LAUNCH_OK(ssad_relu_grad(Y.data<float>(), dY.data<float>(), dX->mutable_data<float>(),
                             Y.size(), context_.hip_stream()), "ReluGradient");
    return true;
  }
};

class SigmoidHIPOp final : public Operator<HIPContext> {
 public:
  using Operator<HIPContext>::Operator;
  bool RunOnDevice() override {
    auto& X = Input(0);
    auto* Y = Output(0);
    Y->ResizeLike(X);
    LAUNCH_OK(ssad_sigmoid(X.data<float>(), Y->mutable_data<float>(), X.size(),
                           context_.hip_stream()), "Sigmoid");
    return true;
  }
};

class SumHIPOp final : public Operator<HIPContext> {
 public:
  using Operator<HIPContext>::Operator;
  bool RunOnDevice() override {
    auto& X0 = Input(0);
    auto* Y = Output(0);
    vector<const float*> ptrs(InputSize());
    for (int i = 0; i < InputSize(); ++i) {
      CAFFE_ENFORCE(Input(i).dims() == X0.dims(), "Sum: input ", i, " has a different shape");
      ptrs[i] = Input(i).data<float>();
    }
    Y->ResizeLike(X0);
    LAUNCH_OK(ssad_sum_n(ptrs.data(), InputSize(), Y->mutable_data<float>(), X0.size(),
                         context_.hip_stream()), "Sum");
    return true;
  }
};

class ScaleHIPOp final : public Operator<HIPContext> {
 public:
  ScaleHIPOp(const OperatorDef& d, Workspace* ws)
      : Operator<HIPContext>(d, ws), scale_(GetSingleArgument<float>("scale", 1.0f)) {}
  bool RunOnDevice() override {
    auto& X = Input(0);
    auto* Y = Output(0);
    Y->ResizeLike(X);
    LAUNCH_OK(ssad_scale(X.data<float>(), Y->mutable_data<float>(), scale_, X.size(),
                         context_.hip_stream()), "Scale");
    return true;
  }
 private:
  float scale_;
};

// inputs X0, w0, X1, w1, ... ; output may alias X0
class WeightedSumHIPOp final : public Operator<HIPContext> {
 public:
  using Operator<HIPContext>::Operator;
  bool RunOnDevice() override {
    CAFFE_ENFORCE_EQ(InputSize() % 2, 0);
    const int pairs = InputSize() / 2;
    auto& X0 = Input(0);
    vector<const float*> xs(pairs), ws(pairs);
    for (int k = 0; k < pairs; ++k) {
      CAFFE_ENFORCE_EQ(Input(2 * k).size(), X0.size());
      CAFFE_ENFORCE_EQ(Input(2 * k + 1).size(), 1);
      xs[k] = Input(2 * k).data<float>();
      ws[k] = Input(2 * k + 1).data<float>();
    }
    auto* Y = Output(0);
    Y->ResizeLike(X0);
    LAUNCH_OK(ssad_weighted_sum(xs.data(), ws.data(), pairs, Y->mutable_data<float>(), X0.size(),
                                context_.hip_stream()), "WeightedSum");
    return true;
  }
};

// float only; shape from arg `shape` or from input 0 (as filler_op.h)
class ConstantFillHIPOp final : public Operator<HIPContext> {
 public:
  ConstantFillHIPOp(const OperatorDef& d, Workspace* ws)
      : Operator<HIPContext>(d, ws),
        value_(GetSingleArgument<float>("value", 0.0f)),
        shape_(GetRepeatedArgument<int64_t>("shape")) {}
  bool RunOnDevice() override {
    auto* Y = Output(0);
    if (InputSize() > 0) Y->ResizeLike(Input(0)); else Y->Resize(shape_);
    LAUNCH_OK(ssad_fill(Y->mutable_data<float>(), value_, Y->size(), context_.hip_stream()),
              "ConstantFill");
    return true;
  }
 private:
  float value_;
  vector<int64_t> shape_;
};

// inputs [grad, momentum, lr, param] -> outputs [grad, momentum, param] in place
class MomentumSGDUpdateHIPOp final : public Operator<HIPContext> {
 public:
  MomentumSGDUpdateHIPOp(const OperatorDef& d, Workspace* ws)
      : Operator<HIPContext>(d, ws),
        momentum_(GetSingleArgument<float>("momentum", 0.0f)),
        nesterov_(GetSingleArgument<int>("nesterov", 0)) {
    CAFFE_ENFORCE(!nesterov_, "nesterov momentum is not on the path and not implemented");
  }
  bool RunOnDevice() override {
    auto& g = Input(0);
    auto& m = Input(1);
    CAFFE_ENFORCE_EQ(Input(2).size(), 1);
    CAFFE_ENFORCE_EQ(g.size(), m.size());
    CAFFE_ENFORCE_EQ(Input(3).size(), g.size());
    Output(0)->ResizeLike(g);
    Output(1)->ResizeLike(m);
    Output(2)->ResizeLike(Input(3));
    float* gp = Output(0)->mutable_data<float>();
    float* mp = Output(1)->mutable_data<float>();
    float* wp = Output(2)->mutable_data<float>();
    CAFFE_ENFORCE(gp == g.data<float>() && mp == m.data<float>() && wp == Input(3).data<float>(),
                  "MomentumSGDUpdate runs in place: outputs must alias grad, momentum, param");
    LAUNCH_OK(ssad_momentum_sgd_update(wp, gp, mp, Input(2).data<float>(), momentum_, 0.0f, 0,
                                       g.size(), context_.hip_stream()), "MomentumSGDUpdate");
    return true;
  }
 private:
  float momentum_;
  int nesterov_;
};

// AffineChannel / AffineChannelGradient (caffe2/modules/detectron/affine_channel_op.{cc,cu}):
// Y = X * scale[c] + bias[c];  dX = dY * scale[c].  Frozen-BN replacement of the backbone.
class AffineChannelHIPOp final : public Operator<HIPContext> {
 public:
  using Operator<HIPContext>::Operator;
  bool RunOnDevice() override {
    auto& X = Input(0);
    auto& scale = Input(1);
    auto& bias = Input(2);
    auto* Y = Output(0);
    CAFFE_ENFORCE_EQ(X.ndim(), 4, "AffineChannel: X must be N x C x H x W");
    CAFFE_ENFORCE_EQ(scale.size(), X.dim32(1));
    CAFFE_ENFORCE_EQ(bias.size(), X.dim32(1));
    Y->ResizeLike(X);
    LAUNCH_OK(ssad_affine_channel(X.data<float>(), scale.data<float>(), bias.data<float>(), nullptr,
                                  Y->mutable_data<float>(), X.dim32(0), X.dim32(1),
                                  X.dim32(2) * X.dim32(3), 0, context_.hip_stream()),
              "AffineChannel");
    return true;
  }
};

class AffineChannelGradientHIPOp final : public Operator<HIPContext> {
 public:
  using Operator<HIPContext>::Operator;
  bool RunOnDevice() override {
    auto& scale = Input(0);
    auto& dY = Input(1);
    auto* dX = Output(0);
    CAFFE_ENFORCE_EQ(dY.ndim(), 4, "AffineChannelGradient: dY must be N x C x H x W");
    CAFFE_ENFORCE_EQ(scale.size(), dY.dim32(1));
    dX->ResizeLike(dY);
    LAUNCH_OK(ssad_affine_channel(dY.data<float>(), scale.data<float>(), nullptr, nullptr,
                                  dX->mutable_data<float>(), dY.dim32(0), dY.dim32(1),
                                  dY.dim32(2) * dY.dim32(3), 0, context_.hip_stream()),
              "AffineChannelGradient");
    return true;
  }
};

// UpsampleNearest / UpsampleNearestGradient (upsample_nearest_op.{h,cu}); arg scale (int, 2)
class UpsampleNearestHIPOp final : public Operator<HIPContext> {
 public:
  UpsampleNearestHIPOp(const OperatorDef& def, Workspace* ws)
      : Operator<HIPContext>(def, ws), scale_(GetSingleArgument<int>("scale", 2)) {
    CAFFE_ENFORCE_GE(scale_, 1);
  }
  bool RunOnDevice() override {
    auto& X = Input(0);
    auto* Y = Output(0);
    CAFFE_ENFORCE_EQ(X.ndim(), 4, "UpsampleNearest: X must be N x C x H x W");
    Y->Resize(X.dim32(0), X.dim32(1), X.dim32(2) * scale_, X.dim32(3) * scale_);
    LAUNCH_OK(ssad_upsample_nearest(X.data<float>(), nullptr, Y->mutable_data<float>(), X.dim32(0),
                                    X.dim32(1), X.dim32(2), X.dim32(3), scale_,
                                    context_.hip_stream()), "UpsampleNearest");
    return true;
  }
 private:
  int scale_;
};

class UpsampleNearestGradientHIPOp final : public Operator<HIPContext> {
 public:
  UpsampleNearestGradientHIPOp(const OperatorDef& def, Workspace* ws)
      : Operator<HIPContext>(def, ws), scale_(GetSingleArgument<int>("scale", 2)) {
    CAFFE_ENFORCE_GE(scale_, 1);
  }
  bool RunOnDevice() override {
    auto& X = Input(0);
    auto& dY = Input(1);
    auto* dX = Output(0);
    CAFFE_ENFORCE_EQ(dY.ndim(), 4);
    CAFFE_ENFORCE_EQ(dY.dim32(2), X.dim32(2) * scale_);
    CAFFE_ENFORCE_EQ(dY.dim32(3), X.dim32(3) * scale_);
    dX->ResizeLike(X);
    LAUNCH_OK(ssad_upsample_nearest_grad(dY.data<float>(), dX->mutable_data<float>(), X.dim32(0),
                                         X.dim32(1), X.dim32(2), X.dim32(3), scale_,
                                         context_.hip_stream()), "UpsampleNearestGradient");
    return true;
  }
 private:
  int scale_;
};

// MaxPool / MaxPoolGradient, NCHW (caffe2/operators/pool_op.{cc,cu}); arguments parsed as
// ConvPoolOpBase does (kernel / stride / pad ...).  Gradient inputs [X, Y, dY] -> dX.
class MaxPoolHIPOp final : public Operator<HIPContext> {
 public:
  MaxPoolHIPOp(const OperatorDef& def, Workspace* ws)
      : Operator<HIPContext>(def, ws), geom_(ParseConvGeometry(*this)) {
    if (geom_.order != "NCHW" || geom_.kernel.size() != 2 || geom_.dilation != vector<int>{1, 1})
      throw UnsupportedOperatorFeature("HIP MaxPool implements order=NCHW, 2-D, dilation 1");
  }
  bool RunOnDevice() override {
    auto& X = Input(0);
    auto* Y = Output(0);
    CAFFE_ENFORCE_EQ(X.ndim(), 4);
    const int H = X.dim32(2), W = X.dim32(3);
    const int OH = ssad_conv_out_size(H, geom_.kernel[0], 1, geom_.pads[0], geom_.pads[2], geom_.stride[0]);
    const int OW = ssad_conv_out_size(W, geom_.kernel[1], 1, geom_.pads[1], geom_.pads[3], geom_.stride[1]);
    CAFFE_ENFORCE(OH > 0 && OW > 0, "MaxPool: the window does not fit the padded input");
    Y->Resize(X.dim32(0), X.dim32(1), OH, OW);
    LAUNCH_OK(ssad_max_pool_forward(X.data<float>(), X.dim32(0), X.dim32(1), H, W, geom_.kernel[0],
                                    geom_.kernel[1], geom_.stride[0], geom_.stride[1], geom_.pads[0],
                                    geom_.pads[1], geom_.pads[2], geom_.pads[3],
                                    Y->mutable_data<float>(), context_.hip_stream()), "MaxPool");
    return true;
  }
 private:
  ConvGeometry geom_;
};

class MaxPoolGradientHIPOp final : public Operator<HIPContext> {
 public:
  MaxPoolGradientHIPOp(const OperatorDef& def, Workspace* ws)
      : Operator<HIPContext>(def, ws), geom_(ParseConvGeometry(*this)) {
    if (geom_.order != "NCHW" || geom_.kernel.size() != 2 || geom_.dilation != vector<int>{1, 1})
      throw UnsupportedOperatorFeature("HIP MaxPoolGradient implements order=NCHW, 2-D, dilation 1");
  }
  bool RunOnDevice() override {
    auto& X = Input(0);
    auto& Y = Input(1);
    auto& dY = Input(2);
    auto* dX = Output(0);
    CAFFE_ENFORCE(dY.dims() == Y.dims(), "MaxPoolGradient: dY and Y differ in shape");
    dX->ResizeLike(X);
    LAUNCH_OK(ssad_max_pool_backward(X.data<float>(), Y.data<float>(), dY.data<float>(), X.dim32(0),
                                     X.dim32(1), X.dim32(2), X.dim32(3), geom_.kernel[0],
                                     geom_.kernel[1], geom_.stride[0], geom_.stride[1], geom_.pads[0],
                                     geom_.pads[1], geom_.pads[2], geom_.pads[3],
                                     dX->mutable_data<float>(), context_.hip_stream()),
              "MaxPoolGradient");
    return true;
  }
 private:
  ConvGeometry geom_;
};

// StopGradient (caffe2/operators/stop_gradient.h): identity forward, usually in place;
// registered with NO gradient so the backward pass ends there (frozen res2).
class StopGradientHIPOp final : public Operator<HIPContext> {
 public:
  using Operator<HIPContext>::Operator;
  bool RunOnDevice() override {
    auto& X = Input(0);
    auto* Y = Output(0);
    if (Y != &X) {
      Y->ResizeLike(X);
      context_.Copy<float, HIPContext, HIPContext>(X.size(), X.data<float>(),
                                                   Y->mutable_data<float>());
    }
    return true;
  }
};

REGISTER_HIP_OPERATOR(StopGradient, StopGradientHIPOp);
REGISTER_HIP_OPERATOR(MaxPool, MaxPoolHIPOp);
REGISTER_HIP_OPERATOR(MaxPoolGradient, MaxPoolGradientHIPOp);
REGISTER_HIP_OPERATOR(UpsampleNearest, UpsampleNearestHIPOp);
REGISTER_HIP_OPERATOR(UpsampleNearestGradient, UpsampleNearestGradientHIPOp);
REGISTER_HIP_OPERATOR(AffineChannel, AffineChannelHIPOp);
REGISTER_HIP_OPERATOR(AffineChannelGradient, AffineChannelGradientHIPOp);
REGISTER_HIP_OPERATOR(Relu, ReluHIPOp);
REGISTER_HIP_OPERATOR(ReluGradient, ReluGradientHIPOp);
REGISTER_HIP_OPERATOR(Sigmoid, SigmoidHIPOp);
REGISTER_HIP_OPERATOR(Sum, SumHIPOp);
REGISTER_HIP_OPERATOR(Scale, ScaleHIPOp);
REGISTER_HIP_OPERATOR(WeightedSum, WeightedSumHIPOp);
REGISTER_HIP_OPERATOR(ConstantFill, ConstantFillHIPOp);
REGISTER_HIP_OPERATOR(MomentumSGDUpdate, MomentumSGDUpdateHIPOp);

OPERATOR_SCHEMA(StopGradient).NumInputs(1).NumOutputs(1).AllowInplace({{0, 0}});
OPERATOR_SCHEMA(MaxPool).NumInputs(1).NumOutputs(1);
OPERATOR_SCHEMA(MaxPoolGradient).NumInputs(3).NumOutputs(1);
OPERATOR_SCHEMA(UpsampleNearest).NumInputs(1).NumOutputs(1);
OPERATOR_SCHEMA(UpsampleNearestGradient).NumInputs(2).NumOutputs(1);
OPERATOR_SCHEMA(AffineChannel).NumInputs(3).NumOutputs(1).AllowInplace({{0, 0}});
OPERATOR_SCHEMA(AffineChannelGradient).NumInputs(2).NumOutputs(1).AllowInplace({{1, 0}});
OPERATOR_SCHEMA(Relu).NumInputs(1).NumOutputs(1).AllowInplace({{0, 0}});
OPERATOR_SCHEMA(ReluGradient).NumInputs(2).NumOutputs(1).AllowInplace({{1, 0}});
OPERATOR_SCHEMA(Sigmoid).NumInputs(1).NumOutputs(1).AllowInplace({{0, 0}});
OPERATOR_SCHEMA(Sum).NumInputs(1, INT_MAX).NumOutputs(1).AllowInplace({{0, 0}});
OPERATOR_SCHEMA(Scale).NumInputs(1).NumOutputs(1).AllowInplace({{0, 0}});
OPERATOR_SCHEMA(WeightedSum).NumInputs(2, INT_MAX).NumOutputs(1).AllowInplace({{0, 0}});
OPERATOR_SCHEMA(ConstantFill).NumInputs(0, 1).NumOutputs(1).AllowInplace({{0, 0}});
OPERATOR_SCHEMA(MomentumSGDUpdate).NumInputs(4).NumOutputs(3).AllowInplace({{0, 0}, {1, 1}, {3, 2}});

// Relu's gradient is taken w.r.t. its OUTPUT (relu_op.cc GetReluGradient)
class GetReluGradient : public GradientMakerBase {
  using GradientMakerBase::GradientMakerBase;
  vector<OperatorDef> GetGradientDefs() override {
    return SingleGradientDef("ReluGradient", "", vector<string>{O(0), GO(0)},
                             vector<string>{GI(0)}, vector<Argument>());
  }
};
REGISTER_GRADIENT(Relu, GetReluGradient);

// affine_channel_op.cc:66-76: only dX; scale and bias are frozen
class GetAffineChannelGradient : public GradientMakerBase {
  using GradientMakerBase::GradientMakerBase;
  vector<OperatorDef> GetGradientDefs() override {
    return SingleGradientDef("AffineChannelGradient", "", vector<string>{I(1), GO(0)},
                             vector<string>{GI(0)}, vector<Argument>());
  }
};
REGISTER_GRADIENT(AffineChannel, GetAffineChannelGradient);

// upsample_nearest_op.cc: gradient op takes (X, dY) -> dX and inherits the arguments
class GetUpsampleNearestGradient : public GradientMakerBase {
  using GradientMakerBase::GradientMakerBase;
  vector<OperatorDef> GetGradientDefs() override {
    return SingleGradientDef("UpsampleNearestGradient", "", vector<string>{I(0), GO(0)},
                             vector<string>{GI(0)});
  }
};
REGISTER_GRADIENT(UpsampleNearest, GetUpsampleNearestGradient);

// pool_gradient_op.cc GetPoolGradient: [X, Y, dY] -> dX, arguments inherited
class GetMaxPoolGradient : public GradientMakerBase {
  using GradientMakerBase::GradientMakerBase;
  vector<OperatorDef> GetGradientDefs() override {
    return SingleGradientDef("MaxPoolGradient", "", vector<string>{I(0), O(0), GO(0)},
                             vector<string>{GI(0)});
  }
};
REGISTER_GRADIENT(MaxPool, GetMaxPoolGradient);
// caffe2/operators/utility_ops.cc GetSumGradient: every input's gradient is the output
// gradient blob itself; no operator is emitted
class GetSumGradient : public GradientMakerBase {
  using GradientMakerBase::GradientMakerBase;
  vector<OperatorDef> GetGradientDefs() override {
    for (int i = 0; i < (int)def_.input.size(); ++i) SetDense(i, GO(0));
    return vector<OperatorDef>();
  }
};
REGISTER_GRADIENT(Sum, GetSumGradient);
NO_GRADIENT(PowSum);
NO_GRADIENT(StopGradient);
NO_GRADIENT(ConstantFill);

}  // namespace caffe2
