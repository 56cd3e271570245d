#include "ops/sigmoid_adaptive_distillation_loss_op.h"

#include "ssad_kernels.h"

namespace caffe2 {

namespace {

// Shape contract shared by both ops.  The reference reads dim32(0..3) of the
// logits and indexes the other inputs without checking them; the checks here
// turn what would be an out-of-bounds device read into an EnforceNotMet.
template <class Ctx>
ssad_distill_level CheckAndDescribe(const Tensor<Ctx>& X, const Tensor<Ctx>& T,
                                    const Tensor<Ctx>& G, const Tensor<Ctx>& wp,
                                    int num_classes) {
  CAFFE_ENFORCE_EQ(X.ndim(), 4, "logits must be N x (A*num_classes) x H x W");
  CAFFE_ENFORCE_GT(num_classes, 0);
  const int N = X.dim32(0), D = X.dim32(1), H = X.dim32(2), W = X.dim32(3);
  CAFFE_ENFORCE_EQ(D % num_classes, 0, "channel dim must be num_anchors * num_classes");
  CAFFE_ENFORCE_EQ(T.size(), X.size(), "teacher probabilities must match the logits");
  CAFFE_ENFORCE_EQ(G.size(), (TIndex)N * (D / num_classes) * H * W,
                   "labels must be N x num_anchors x H x W");
  CAFFE_ENFORCE_GE(wp.size(), 1, "normalizer must hold one value");
  ssad_distill_level lv;
  lv.logits = X.template data<float>();
  lv.teacher_prob = T.template data<float>();
  lv.labels = G.template data<int>();
  lv.out = nullptr;
  lv.N = N; lv.D = D; lv.H = H; lv.W = W;
  return lv;
}

void EnforceLaunch(int rc, const char* what) {
  CAFFE_ENFORCE_EQ(rc, 0, what, " launch failed");
}

}  // namespace

template <>
bool SigmoidAdaptiveDistillLossOp<float, HIPContext>::RunOnDevice() {
  auto& X = Input(0);    // logits
  auto& T = Input(1);    // teacher probabilities
  auto& G = Input(2);    // labels (only gate the ignored anchors)
  auto& wp = Input(3);   // normalizer
  auto* avg_loss = Output(0);

  ssad_distill_level lv = CheckAndDescribe(X, T, G, wp, num_classes_);
  avg_loss->Resize(vector<TIndex>());
  lv.out = avg_loss->mutable_data<float>();

  const size_t ws_bytes = ssad_distill_loss_workspace_bytes(1);
  partials_.Resize((TIndex)ws_bytes);
  void* ws = partials_.mutable_data<uint8_t>();

  const ssad_distill_params P{gamma_, alpha_, beta_, num_classes_, ignored_label_, scale_};
  EnforceLaunch(ssad_distill_loss_forward(&lv, 1, wp.data<float>(), &P, ws, ws_bytes,
                                          context_.hip_stream()),
                "SigmoidAdaptiveDistillLoss");
  return true;
}

template <>
bool SigmoidAdaptiveDistillLossGradientOp<float, HIPContext>::RunOnDevice() {
  auto& X = Input(0);
  auto& T = Input(1);
  auto& G = Input(2);
  auto& wp = Input(3);
  auto& d_avg_loss = Input(InputSize() - 1);
  auto* dX = Output(0);

  ssad_distill_level lv = CheckAndDescribe(X, T, G, wp, num_classes_);
  CAFFE_ENFORCE_GE(d_avg_loss.size(), 1);
  dX->ResizeLike(X);
  lv.out = dX->mutable_data<float>();

  const ssad_distill_params P{gamma_, alpha_, beta_, num_classes_, ignored_label_, scale_};
  EnforceLaunch(ssad_distill_loss_backward(&lv, 1, wp.data<float>(), d_avg_loss.data<float>(), 0,
                                           &P, context_.hip_stream()),
                "SigmoidAdaptiveDistillLossGradient");
  return true;
}

REGISTER_CPU_OPERATOR(SigmoidAdaptiveDistillLoss, SigmoidAdaptiveDistillLossOp<float, CPUContext>);
REGISTER_CPU_OPERATOR(SigmoidAdaptiveDistillLossGradient,
                      SigmoidAdaptiveDistillLossGradientOp<float, CPUContext>);
REGISTER_HIP_OPERATOR(SigmoidAdaptiveDistillLoss, SigmoidAdaptiveDistillLossOp<float, HIPContext>);
REGISTER_HIP_OPERATOR(SigmoidAdaptiveDistillLossGradient,
                      SigmoidAdaptiveDistillLossGradientOp<float, HIPContext>);

OPERATOR_SCHEMA(SigmoidAdaptiveDistillLoss)
    .NumInputs(4)
    .NumOutputs(1)
    .SetDoc("Adaptive distillation focal loss between student logits and teacher "
            "probabilities, normalised by max(1, normalizer).")
    .Arg("scale", "(float) default 1.0; multiply the loss by this scale factor.")
    .Arg("alpha", "(float) default 0.25; weight of the positive term.")
    .Arg("gamma", "(float) default 1.0; exponent of the adaptive modulating factor.")
    .Arg("beta", "(float) default 0; weight of the teacher-entropy term in the divergence.")
    .Arg("num_classes", "(int) default 80; number of classes (excluding background).")
    .Arg("ignored_label", "(int) default -1; anchors with this label contribute zero.")
    .Input(0, "logits", "4D tensor (N, A * num_classes, H, W) of student logits.")
    .Input(1, "teacher_prob", "4D tensor of teacher sigmoid probabilities, same shape.")
    .Input(2, "labels", "4D int32 tensor (N, A, H, W); only ignored_label is inspected.")
    .Input(3, "normalizer", "Scalar; the loss is normalized by 1 / max(1, normalizer).")
    .Output(0, "loss", "Scalar loss.");

OPERATOR_SCHEMA(SigmoidAdaptiveDistillLossGradient)
    .NumInputs(5)
    .NumOutputs(1)
    .Input(0, "logits", "See SigmoidAdaptiveDistillLoss.")
    .Input(1, "teacher_prob", "See SigmoidAdaptiveDistillLoss.")
    .Input(2, "labels", "See SigmoidAdaptiveDistillLoss.")
    .Input(3, "normalizer", "See SigmoidAdaptiveDistillLoss.")
    .Input(4, "d_loss", "Gradient of forward output 0 (loss)")
    .Output(0, "d_logits", "Gradient of forward input 0 (logits)");

// Only the logits receive a gradient (reference .cc:99-112).
class GetSigmoidAdaptiveDistillLossGradient : public GradientMakerBase {
  using GradientMakerBase::GradientMakerBase;
  vector<OperatorDef> GetGradientDefs() override {
    return SingleGradientDef("SigmoidAdaptiveDistillLossGradient", "",
                             vector<string>{I(0), I(1), I(2), I(3), GO(0)},
                             vector<string>{GI(0)});
  }
};
REGISTER_GRADIENT(SigmoidAdaptiveDistillLoss, GetSigmoidAdaptiveDistillLossGradient);

}  // namespace caffe2
