#include "ops/pow_sum_op.h"

#include "ssad_kernels.h"

namespace caffe2 {

template <>
bool PowSumOp<float, HIPContext>::RunOnDevice() {
  auto* res = Output(0);
  res->Resize(vector<TIndex>());
  const int n = InputSize();
  vector<const float*> ptrs(n);
  vector<int64_t> sizes(n);
  for (int i = 0; i < n; ++i) {
    auto& in = Input(i);
    ptrs[i] = in.data<float>();
    sizes[i] = in.size();
  }
  const size_t ws_bytes = ssad_pow_sum_workspace_bytes(n);
  if ((size_t)_buff.size() != ws_bytes) {
    // the launcher keeps its arrival counters in the scratch: zero once per allocation
    _buff.Resize((TIndex)ws_bytes);
    CAFFE_ENFORCE_EQ((int)hipMemsetAsync(_buff.mutable_data<uint8_t>(), 0, ws_bytes, context_.hip_stream()), 0,
                     "PowSum scratch memset failed");
  }
  const int rc = ssad_pow_sum(ptrs.data(), sizes.data(), n, power, res->mutable_data<float>(),
                              _buff.mutable_data<uint8_t>(), ws_bytes, context_.hip_stream());
  CAFFE_ENFORCE_EQ(rc, 0, "PowSum launch failed");
  return true;
}

REGISTER_CPU_OPERATOR(PowSum, PowSumOp<float, CPUContext>);
REGISTER_HIP_OPERATOR(PowSum, PowSumOp<float, HIPContext>);

OPERATOR_SCHEMA(PowSum)
    .NumInputs(1, INT_MAX)
    .NumOutputs(1)
    .Arg("power", "(float) default 1.0; exponent applied to every element")
    .Output(0, "sum", "Scalar: sum over all inputs of the element-wise powers.");

}  // namespace caffe2
