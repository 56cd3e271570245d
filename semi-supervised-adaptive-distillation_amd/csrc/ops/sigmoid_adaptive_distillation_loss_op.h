// SigmoidAdaptiveDistillLoss / SigmoidAdaptiveDistillLossGradient.
//
// Same class shape, argument names, defaults and input/output contract as
// the reference (caffe2/modules/detectron/
// sigmoid_adaptive_distillation_loss_op.h:27-87, .cc:21-112):
//   inputs  0 logits  N x (A*num_classes) x H x W  float
//           1 teacher probabilities, same shape     float
//           2 labels  N x A x H x W                 int32 (ignored_label gates)
//           3 normalizer, scalar                    float
//          (4 d_loss, scalar, gradient op only)
//   output  0 scalar loss / d_logits
//   args    scale(1.0, >= 0) num_classes(80) gamma(1.0) alpha(0.25) beta(0)
//           ignored_label(-1)
// As in the reference there is no CPU implementation (RunOnDevice for a
// generic Context is CAFFE_NOT_IMPLEMENTED); the HIPContext specialisation
// replaces the CUDAContext one.
#ifndef C2HIP_SIGMOID_ADAPTIVE_DISTILL_LOSS_OP_H_
#define C2HIP_SIGMOID_ADAPTIVE_DISTILL_LOSS_OP_H_

#include "c2/operator.h"

namespace caffe2 {

template <typename T, class Context>
class SigmoidAdaptiveDistillLossOp final : public Operator<Context> {
 public:
  SigmoidAdaptiveDistillLossOp(const OperatorDef& operator_def, Workspace* ws)
      : Operator<Context>(operator_def, ws),
        scale_(OperatorBase::GetSingleArgument<float>("scale", 1.f)),
        num_classes_(OperatorBase::GetSingleArgument<int>("num_classes", 80)),
        gamma_(OperatorBase::GetSingleArgument<float>("gamma", 1.f)),
        alpha_(OperatorBase::GetSingleArgument<float>("alpha", 0.25f)),
        beta_(OperatorBase::GetSingleArgument<float>("beta", 0.f)),
        ignored_label_(OperatorBase::GetSingleArgument<int>("ignored_label", -1)) {
    CAFFE_ENFORCE(scale_ >= 0);
  }
  USE_OPERATOR_CONTEXT_FUNCTIONS;

  bool RunOnDevice() override {
    // No CPU implementation for now (as the reference)
    CAFFE_NOT_IMPLEMENTED;
  }

 protected:
  float scale_;
  int num_classes_;
  float gamma_;
  float alpha_;
  float beta_;
  int ignored_label_;
  Tensor<Context> partials_;   // reduction scratch (replaces the full-size losses_)
};

template <typename T, class Context>
class SigmoidAdaptiveDistillLossGradientOp final : public Operator<Context> {
 public:
  SigmoidAdaptiveDistillLossGradientOp(const OperatorDef& def, Workspace* ws)
      : Operator<Context>(def, ws),
        scale_(OperatorBase::GetSingleArgument<float>("scale", 1.f)),
        num_classes_(OperatorBase::GetSingleArgument<int>("num_classes", 80)),
        gamma_(OperatorBase::GetSingleArgument<float>("gamma", 1.f)),
        alpha_(OperatorBase::GetSingleArgument<float>("alpha", 0.25f)),
        beta_(OperatorBase::GetSingleArgument<float>("beta", 0.f)),
        ignored_label_(OperatorBase::GetSingleArgument<int>("ignored_label", -1)) {
    CAFFE_ENFORCE(scale_ >= 0);
  }
  USE_OPERATOR_CONTEXT_FUNCTIONS;

  bool RunOnDevice() override {
    CAFFE_NOT_IMPLEMENTED;
  }

 protected:
  float scale_;
  int num_classes_;
  float gamma_;
  float alpha_;
  float beta_;
  int ignored_label_;
};

}  // namespace caffe2
#endif
