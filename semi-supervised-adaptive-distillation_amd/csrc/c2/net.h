// net.h -- NetBase / CreateNet / the net lowering.
//
// The reference runs a training iteration as ONE call, `workspace.RunNet(model.net.Proto().name)`
// (detectron/tools/train_net.py:165-189), on a net object that Workspace::CreateNet built once from
// the NetDef (caffe2/core/workspace.cc CreateNet -> caffe2/core/net.cc:35-60 CreateNet ->
// REGISTER_NET(simple | dag): caffe2/core/net_simple.cc:30-84, net_dag.cc:181-337).  This is that
// object for HIPContext:
//
//   * construction instantiates the operators once (operator.cc:116-200 per op, with the net's
//     device_option copied to operators that have none, net_simple.cc:41-50);
//   * Run() enqueues every operator on the device stream in order and synchronises ONCE at the end
//     ("simple" and "dag" alike: one in-order stream gives every dependency the DAG executor
//     enforces; the reference's executors synchronise after every operator, operator.h:378 --
//     NetDef arg `hip_sync_every_op` = 1 or C2HIP_NET_SYNC_EVERY_OP=1 restores that, e.g. to find
//     the operator behind an asynchronous fault);
//   * before instantiation the operator list goes through LowerNet (net_lowering.cc), the
//     counterpart of Caffe2's graph transforms (caffe2/core/transform.h, and what cuDNN / the DAG
//     executor's chains do implicitly): Conv + in-place Relu -> Conv(fuse_relu), ReluGradient ->
//     ConvGradient(relu_grad_on_input), the convolutions that are ready together -- ConvShared's five
//     FPN levels of one filter (detector.py:449-482), the cls and bbox tower layer of equal depth --
//     -> one ConvGroup / ConvGradientGroup operator = one multi-problem launch, and the autograd Sum
//     of a shared filter's gradient pieces (core.py:706-741) -> the group's filter-gradient launch.
//     NetDef arg `hip_lowering` = 0 or C2HIP_NET_LOWERING=0 instantiates the list as written.
#ifndef C2HIP_NET_H_
#define C2HIP_NET_H_

#include <functional>

#include "c2/operator.h"

namespace caffe2 {

struct LoweringOptions {
  bool fuse_relu = true;        // Conv + Relu, ReluGradient + ConvGradient
  bool group_convs = true;      // ConvGroup / ConvGradientGroup (+ Sum absorption)
  bool frozen_f24 = true;       // nets without gradient operators: Conv on the F(2x4, 3x3) engine (hip_algo = winograd24)
  bool train_f24 = true;        // trained nets: Conv and ConvGradient's data gradient on it too
  bool split = true;            // ... and, of those, the >= 256-wide ones on the split-operand engine (hip_algo = split:
                                // conv3x3_split.hip; round 6) -- NetDef arg hip_split, environment C2HIP_NET_SPLIT
  // TensorProto::DataType id of a blob that exists already (parameters do when a net is created:
  // the reference runs param_init_net first), 0 if unknown.  The fused 3x3 paths are fp32-only.
  std::function<int(const string&)> blob_dtype;
  // blobs that must hold their real contents after a run although the net only uses them
  // internally (NetDef.external_output + arg hip_keep_blobs)
  std::set<string> keep;
};

struct LoweringReport {
  int ops_in = 0, ops_out = 0;
  int relu_fused = 0, relu_grad_fused = 0, sigmoid_fused = 0;
  int conv_groups = 0, conv_group_members = 0;
  int conv_grad_groups = 0, conv_grad_group_members = 0;
  int sums_absorbed = 0;
  int frozen_f24 = 0;           // Conv operators of an evaluated-only net sent to the F(2x4, 3x3) engine
  int train_f24 = 0;            // Conv / ConvGradient operators of a trained net sent to it
  int split = 0;                // of both: marked hip_algo = split (the operators apply the width limits)
  bool fell_back = false;       // the lowered list failed its own verification: list kept as written
  string ToString() const;
};

// The options a net asks for: NetDef args hip_frozen_f24 / hip_train_f24 / hip_keep_blobs + external_output, each
// switch overridable by its environment variable (C2HIP_NET_FUSE_RELU, _GROUP_CONVS, _FROZEN_F24,
// _TRAIN_F24); blob_dtype is left to the caller.
C2HIP_API LoweringOptions LoweringOptionsFor(const NetDef& def);

// Pure function of the definition (+ the dtype probe): no device, no workspace.
C2HIP_API vector<OperatorDef> LowerNet(const NetDef& def, const LoweringOptions& opt,
                                       LoweringReport* report);

class C2HIP_API NetBase {
 public:
  NetBase(const NetDef& def, Workspace* ws);
  virtual ~NetBase() {}
  virtual bool Run();
  const string& Name() const { return name_; }
  const vector<OperatorDef>& lowered_ops() const { return lowered_; }
  const LoweringReport& report() const { return report_; }
  const vector<string>& skipped_blobs() const { return skipped_; }

 protected:
  string name_;
  Workspace* ws_ = nullptr;
  vector<string> produced_, skipped_;     // outputs of the lowered list / outputs of the definition it dropped
  bool sync_every_op_ = false;
  vector<OperatorDef> lowered_;
  LoweringReport report_;
  vector<std::unique_ptr<OperatorBase>> operators_;
};

C2HIP_API std::unique_ptr<NetBase> CreateNet(const NetDef& def, Workspace* ws);

}  // namespace caffe2
#endif  // C2HIP_NET_H_
