// PowSum: out = sum over all inputs of sum_i in[i]^power (scalar).
// Class shape and argument of caffe2/modules/detectron/pow_sum_op.h:25-42;
// schema of pow_sum_op.cc:26-39 (1..inf inputs, 1 output, no gradient).
#ifndef C2HIP_POW_SUM_OP_H_
#define C2HIP_POW_SUM_OP_H_

#include "c2/operator.h"

namespace caffe2 {

template <typename T, class Context>
class PowSumOp final : public Operator<Context> {
 public:
  PowSumOp(const OperatorDef& operator_def, Workspace* ws)
      : Operator<Context>(operator_def, ws),
        power(OperatorBase::GetSingleArgument<float>("power", 1.0f)) {}
  USE_OPERATOR_CONTEXT_FUNCTIONS;

  bool RunOnDevice() override {
    // No CPU implementation for now (as the reference)
    CAFFE_NOT_IMPLEMENTED;
  }

 protected:
  float power;
  Tensor<Context> _buff;   // partial sums (the reference's full-size pow temp is gone)
};

}  // namespace caffe2
#endif
