// SigmoidFocalLoss(+Gradient) and SelectSmoothL1Loss(+Gradient) for
// HIPContext -- the student's supervised RetinaNet losses (SURVEY.md 8f row
// f2), built by detectron/lib/modeling/retinanet_heads.py:259-307.
//
// Contracts (arguments, defaults, inputs/outputs, gradient makers) follow
// caffe2/modules/detectron/sigmoid_focal_loss_op.{h,cc} and
// select_smooth_l1_loss_op.{h,cc}; like the reference there is no CPU
// implementation (the CPU registrations raise "Not Implemented").
#include "c2/operator.h"
#include "ssad_kernels.h"

namespace caffe2 {

namespace {
void Launched(int rc, const char* what) { CAFFE_ENFORCE_EQ(rc, 0, what, " launch failed"); }
}  // namespace

// ---------------------------------------------------------------------------
// SigmoidFocalLoss
// ---------------------------------------------------------------------------
template <typename T, class Context>
class SigmoidFocalLossOp final : public Operator<Context> {
 public:
  SigmoidFocalLossOp(const OperatorDef& def, Workspace* ws)
      : Operator<Context>(def, ws),
        scale_(OperatorBase::GetSingleArgument<float>("scale", 1.f)),
        num_classes_(OperatorBase::GetSingleArgument<int>("num_classes", 80)),
        gamma_(OperatorBase::GetSingleArgument<float>("gamma", 1.f)),
        alpha_(OperatorBase::GetSingleArgument<float>("alpha", 0.25f)) {
    CAFFE_ENFORCE(scale_ >= 0);
  }
  USE_OPERATOR_CONTEXT_FUNCTIONS;
  bool RunOnDevice() override { CAFFE_NOT_IMPLEMENTED; }   // no CPU implementation

 protected:
  float scale_;
  int num_classes_;
  float gamma_;
  float alpha_;
  Tensor<Context> partials_;
};

template <typename T, class Context>
class SigmoidFocalLossGradientOp final : public Operator<Context> {
 public:
  SigmoidFocalLossGradientOp(const OperatorDef& def, Workspace* ws)
      : Operator<Context>(def, ws),
        scale_(OperatorBase::GetSingleArgument<float>("scale", 1.f)),
        num_classes_(OperatorBase::GetSingleArgument<int>("num_classes", 80)),
        gamma_(OperatorBase::GetSingleArgument<float>("gamma", 1.f)),
        alpha_(OperatorBase::GetSingleArgument<float>("alpha", 0.25f)) {
    CAFFE_ENFORCE(scale_ >= 0);
  }
  USE_OPERATOR_CONTEXT_FUNCTIONS;
  bool RunOnDevice() override { CAFFE_NOT_IMPLEMENTED; }

 protected:
  float scale_;
  int num_classes_;
  float gamma_;
  float alpha_;
};

namespace {
template <class Ctx>
ssad_distill_level FocalLevel(const Tensor<Ctx>& X, const Tensor<Ctx>& T, const Tensor<Ctx>& wp,
                              int num_classes) {
  CAFFE_ENFORCE_EQ(X.ndim(), 4, "logits must be N x (A*num_classes) x H x W");
  CAFFE_ENFORCE_GT(num_classes, 0);
  const int N = X.dim32(0), D = X.dim32(1), H = X.dim32(2), W = X.dim32(3);
  CAFFE_ENFORCE_EQ(D % num_classes, 0, "channel dim must be num_anchors * num_classes");
  CAFFE_ENFORCE_EQ(T.size(), (TIndex)N * (D / num_classes) * H * W,
                   "labels must be N x num_anchors x H x W");
  CAFFE_ENFORCE_GE(wp.size(), 1);
  return ssad_distill_level{X.template data<float>(), nullptr, T.template data<int>(), nullptr,
                            N, D, H, W};
}
}  // namespace

template <>
bool SigmoidFocalLossOp<float, HIPContext>::RunOnDevice() {
  auto& X = Input(0);
  auto& T = Input(1);
  auto& wp = Input(2);
  auto* avg_loss = Output(0);
  ssad_distill_level lv = FocalLevel(X, T, wp, num_classes_);
  avg_loss->Resize(vector<TIndex>());
  lv.out = avg_loss->mutable_data<float>();
  const size_t wsb = ssad_distill_loss_workspace_bytes(1);
  partials_.Resize((TIndex)wsb);
  const ssad_focal_params P{gamma_, alpha_, num_classes_, scale_};
  Launched(ssad_focal_loss_forward(&lv, 1, wp.data<float>(), &P, partials_.mutable_data<uint8_t>(),
                                   wsb, context_.hip_stream()), "SigmoidFocalLoss");
  return true;
}

template <>
bool SigmoidFocalLossGradientOp<float, HIPContext>::RunOnDevice() {
  auto& X = Input(0);
  auto& T = Input(1);
  auto& wp = Input(2);
  auto& d_avg_loss = Input(InputSize() - 1);
  auto* dX = Output(0);
  ssad_distill_level lv = FocalLevel(X, T, wp, num_classes_);
  CAFFE_ENFORCE_GE(d_avg_loss.size(), 1);
  dX->ResizeLike(X);
  lv.out = dX->mutable_data<float>();
  const ssad_focal_params P{gamma_, alpha_, num_classes_, scale_};
  Launched(ssad_focal_loss_backward(&lv, 1, wp.data<float>(), d_avg_loss.data<float>(), 0, &P,
                                    context_.hip_stream()), "SigmoidFocalLossGradient");
  return true;
}

// ---------------------------------------------------------------------------
// SelectSmoothL1Loss
// ---------------------------------------------------------------------------
template <typename T, class Context>
class SelectSmoothL1LossOp final : public Operator<Context> {
 public:
  SelectSmoothL1LossOp(const OperatorDef& def, Workspace* ws)
      : Operator<Context>(def, ws),
        beta_(OperatorBase::GetSingleArgument<float>("beta", 1.f)),
        scale_(OperatorBase::GetSingleArgument<float>("scale", 1.f)) {
    CAFFE_ENFORCE(beta_ > 0);
    CAFFE_ENFORCE(scale_ >= 0);
  }
  USE_OPERATOR_CONTEXT_FUNCTIONS;
  bool RunOnDevice() override { CAFFE_NOT_IMPLEMENTED; }

 protected:
  float beta_;    // transition point from L1 to L2 loss
  float scale_;   // scale the loss by scale_
  Tensor<Context> partials_;   // reduction scratch (the reference's buff_, select_smooth_l1_loss_op.h)
};

template <typename T, class Context>
class SelectSmoothL1LossGradientOp final : public Operator<Context> {
 public:
  SelectSmoothL1LossGradientOp(const OperatorDef& def, Workspace* ws)
      : Operator<Context>(def, ws),
        beta_(OperatorBase::GetSingleArgument<float>("beta", 1.f)),
        scale_(OperatorBase::GetSingleArgument<float>("scale", 1.f)) {
    CAFFE_ENFORCE(beta_ > 0);
    CAFFE_ENFORCE(scale_ >= 0);
  }
  USE_OPERATOR_CONTEXT_FUNCTIONS;
  bool RunOnDevice() override { CAFFE_NOT_IMPLEMENTED; }

 protected:
  float beta_;
  float scale_;
};

template <>
bool SelectSmoothL1LossOp<float, HIPContext>::RunOnDevice() {
  auto& Y_hat = Input(0);   // N x (A*4) x H x W box predictions
  auto& Y = Input(1);       // M x 4 targets
  auto& L = Input(2);       // M x 4 locations (n, c, y, x) as floats
  auto& S = Input(3);       // number of foreground boxes over all levels
  auto* avg_loss = Output(0);
  avg_loss->Resize(vector<TIndex>());
  float* out = avg_loss->mutable_data<float>();
  hipStream_t s = context_.hip_stream();
  if (Y.size() == 0) {      // .cu:101-105
    Launched(ssad_fill(out, 0.0f, 1, s), "SelectSmoothL1Loss");
    return true;
  }
  CAFFE_ENFORCE_EQ(Y_hat.ndim(), 4);
  CAFFE_ENFORCE_EQ(L.size(), Y.size(), "one (n, c, y, x) row per target row");
  const int M = (int)(Y.size() / 4);
  const size_t wsb = ssad_select_smooth_l1_workspace_bytes(1);
  partials_.Resize((TIndex)wsb);
  Launched(ssad_select_smooth_l1_forward(Y_hat.data<float>(), Y.data<float>(), L.data<float>(),
                                         S.data<float>(), Y_hat.dim32(0), Y_hat.dim32(1),
                                         Y_hat.dim32(2), Y_hat.dim32(3), M, beta_, scale_, out,
                                         partials_.mutable_data<uint8_t>(), wsb, s),
           "SelectSmoothL1Loss");
  return true;
}

template <>
bool SelectSmoothL1LossGradientOp<float, HIPContext>::RunOnDevice() {
  auto& Y_hat = Input(0);
  auto& Y = Input(1);
  auto& L = Input(2);
  auto& S = Input(3);
  auto& d_avg_loss = Input(4);
  auto* d_Y_hat = Output(0);
  d_Y_hat->ResizeLike(Y_hat);
  hipStream_t s = context_.hip_stream();
  Launched(ssad_fill(d_Y_hat->mutable_data<float>(), 0.0f, d_Y_hat->size(), s), "zero d_Y_hat");
  if (Y.size() == 0) return true;
  CAFFE_ENFORCE_EQ(Y_hat.ndim(), 4);
  CAFFE_ENFORCE_EQ(L.size(), Y.size());
  const int M = (int)(Y.size() / 4);
  Launched(ssad_select_smooth_l1_backward(
               Y_hat.data<float>(), Y.data<float>(), L.data<float>(), S.data<float>(),
               d_avg_loss.data<float>(), Y_hat.dim32(0), Y_hat.dim32(1), Y_hat.dim32(2),
               Y_hat.dim32(3), M, beta_, scale_, d_Y_hat->mutable_data<float>(), s),
           "SelectSmoothL1LossGradient");
  return true;
}

REGISTER_CPU_OPERATOR(SigmoidFocalLoss, SigmoidFocalLossOp<float, CPUContext>);
REGISTER_CPU_OPERATOR(SigmoidFocalLossGradient, SigmoidFocalLossGradientOp<float, CPUContext>);
REGISTER_CPU_OPERATOR(SelectSmoothL1Loss, SelectSmoothL1LossOp<float, CPUContext>);
REGISTER_CPU_OPERATOR(SelectSmoothL1LossGradient, SelectSmoothL1LossGradientOp<float, CPUContext>);
REGISTER_HIP_OPERATOR(SigmoidFocalLoss, SigmoidFocalLossOp<float, HIPContext>);
REGISTER_HIP_OPERATOR(SigmoidFocalLossGradient, SigmoidFocalLossGradientOp<float, HIPContext>);
REGISTER_HIP_OPERATOR(SelectSmoothL1Loss, SelectSmoothL1LossOp<float, HIPContext>);
REGISTER_HIP_OPERATOR(SelectSmoothL1LossGradient, SelectSmoothL1LossGradientOp<float, HIPContext>);

OPERATOR_SCHEMA(SigmoidFocalLoss)
    .NumInputs(3)
    .NumOutputs(1)
    .Arg("scale", "(float) default 1.0; multiply the loss by this scale factor.")
    .Arg("alpha", "(float) default 0.25; Focal Loss's alpha hyper-parameter.")
    .Arg("gamma", "(float) default 1.0; Focal Loss's gamma hyper-parameter.")
    .Arg("num_classes", "(int) default 80; number of classes (excluding background).")
    .Input(0, "logits", "4D tensor (N, A * num_classes, H, W).")
    .Input(1, "labels", "4D int32 tensor (N, A, H, W): -1 ignore, 0 background, k class k.")
    .Input(2, "normalizer", "Scalar; the loss is normalized by 1 / max(1, normalizer).")
    .Output(0, "loss", "Scalar loss.");
OPERATOR_SCHEMA(SigmoidFocalLossGradient).NumInputs(4).NumOutputs(1);
OPERATOR_SCHEMA(SelectSmoothL1Loss)
    .NumInputs(4)
    .NumOutputs(1)
    .Arg("beta", "(float) default 1.0; L2 to L1 transition point.")
    .Arg("scale", "(float) default 1.0; multiply the loss by this scale factor.")
    .Input(0, "Y_hat", "4D tensor of bounding box regression predictions (N, 4 * A, H, W).")
    .Input(1, "Y", "2D tensor of labels shape (M, 4) for 4 contiguous channels starting at each of the M locations selected by the locations input.")
    .Input(2, "locations", "2D tensor of shape (M, 4) that identifies M 'select' locations: (n, c, y, x).")
    .Input(3, "normalizer", "Scalar; the loss is divided by max(1, normalizer).")
    .Output(0, "loss", "Scalar loss.");
OPERATOR_SCHEMA(SelectSmoothL1LossGradient).NumInputs(5).NumOutputs(1);

class GetSigmoidFocalLossGradient : public GradientMakerBase {
  using GradientMakerBase::GradientMakerBase;
  vector<OperatorDef> GetGradientDefs() override {
    return SingleGradientDef("SigmoidFocalLossGradient", "",
                             vector<string>{I(0), I(1), I(2), GO(0)}, vector<string>{GI(0)});
  }
};
REGISTER_GRADIENT(SigmoidFocalLoss, GetSigmoidFocalLossGradient);

class GetSelectSmoothL1LossGradient : public GradientMakerBase {
  using GradientMakerBase::GradientMakerBase;
  vector<OperatorDef> GetGradientDefs() override {
    return SingleGradientDef("SelectSmoothL1LossGradient", "",
                             vector<string>{I(0), I(1), I(2), I(3), GO(0)}, vector<string>{GI(0)});
  }
};
REGISTER_GRADIENT(SelectSmoothL1Loss, GetSelectSmoothL1LossGradient);

}  // namespace caffe2
