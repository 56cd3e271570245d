// net.cc -- the net object of Workspace::CreateNet / RunNet for HIPContext (see net.h).
#include "c2/net.h"

#include <set>
#include <utility>
#include <vector>

#include <cstdlib>

namespace caffe2 {

namespace {

const Argument* FindArg(const NetDef& def, const char* name) {
  for (const Argument& a : def.arg)
    if (a.name == name) return &a;
  return nullptr;
}

bool EnvFlag(const char* name, bool dflt) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  return !(e[0] == '0' || e[0] == 'n' || e[0] == 'N' || e[0] == 'f' || e[0] == 'F');
}

int BlobDtype(const Workspace* ws, const string& name) {
  const Blob* b = ws->GetBlob(name);
  if (!b) return 0;
  if (b->IsType<TensorHIP>()) return (int)b->Get<TensorHIP>().meta().id;
  if (b->IsType<TensorCPU>()) return (int)b->Get<TensorCPU>().meta().id;
  return 0;
}

}  // namespace

string LoweringReport::ToString() const {
  return MakeString("ops ", ops_in, " -> ", ops_out, "; Relu fused ", relu_fused, ", Sigmoid fused ", sigmoid_fused, ", ReluGradient fused ",
                    relu_grad_fused, "; ConvGroup ", conv_groups, " (", conv_group_members,
                    " Conv); ConvGradientGroup ", conv_grad_groups, " (", conv_grad_group_members,
                    " ConvGradient); Sum absorbed ", sums_absorbed, "; F(2x4) Conv ", frozen_f24, " evaluated / ", train_f24, " trained (", split, " of them marked split)",
                    fell_back ? "; FELL BACK to the list as written" : "");
}

LoweringOptions LoweringOptionsFor(const NetDef& def) {
  LoweringOptions opt;
  opt.fuse_relu = EnvFlag("C2HIP_NET_FUSE_RELU", true);
  opt.group_convs = EnvFlag("C2HIP_NET_GROUP_CONVS", true);
  const Argument* f24 = FindArg(def, "hip_frozen_f24");
  opt.frozen_f24 = EnvFlag("C2HIP_NET_FROZEN_F24", !(f24 && f24->has_i && f24->i == 0));
  const Argument* t24 = FindArg(def, "hip_train_f24");
  opt.train_f24 = EnvFlag("C2HIP_NET_TRAIN_F24", !(t24 && t24->has_i && t24->i == 0));
  const Argument* sp = FindArg(def, "hip_split");
  opt.split = EnvFlag("C2HIP_NET_SPLIT", !(sp && sp->has_i && sp->i == 0));
  for (const string& s : def.external_output) opt.keep.insert(s);
  if (const Argument* k = FindArg(def, "hip_keep_blobs"))
    for (const string& s : k->strings) opt.keep.insert(s);
  return opt;
}

NetBase::NetBase(const NetDef& def, Workspace* ws) : name_(def.name), ws_(ws) {
  const Argument* sync = FindArg(def, "hip_sync_every_op");
  sync_every_op_ = EnvFlag("C2HIP_NET_SYNC_EVERY_OP", sync && sync->has_i && sync->i != 0);
  const Argument* low = FindArg(def, "hip_lowering");
  const bool lowering = EnvFlag("C2HIP_NET_LOWERING", !(low && low->has_i && low->i == 0));

  // the net's device option is the default of operators that carry none (net_simple.cc:41-50)
  NetDef scoped = def;
  if (def.has_device_option)
    for (OperatorDef& op : scoped.op)
      if (!op.has_device_option) {
        op.device_option = def.device_option;
        op.has_device_option = true;
      }
  if (lowering) {
    LoweringOptions opt = LoweringOptionsFor(def);
    opt.blob_dtype = [ws](const string& n) { return BlobDtype(ws, n); };
    lowered_ = LowerNet(scoped, opt, &report_);
  } else {
    lowered_ = scoped.op;
    report_.ops_in = report_.ops_out = (int)lowered_.size();
  }
  {
    std::set<string> made;
    for (const OperatorDef& op : lowered_) made.insert(op.output.begin(), op.output.end());
    produced_.assign(made.begin(), made.end());
    std::set<string> gone;
    for (const OperatorDef& op : scoped.op)
      for (const string& o : op.output)
        if (!made.count(o)) gone.insert(o);
    skipped_.assign(gone.begin(), gone.end());
  }
  operators_.reserve(lowered_.size());
  for (const OperatorDef& op : lowered_) operators_.push_back(CreateOperator(op, ws));
}

bool NetBase::Run() {
  // the last operator enqueued per device: net_simple.cc / net_dag.cc finish every operator, so a net that spans
  // devices (or whose caller moved one device's operators to its own stream) must not return with work in flight
  // on any of them
  std::vector<std::pair<int, OperatorBase*>> last;
  for (auto& op : operators_) {
    const bool ok = sync_every_op_ ? op->Run() : op->RunAsync();
    if (!ok) return false;
    if (!op->OnDeviceStream()) continue;
    const int key = op->DeviceKey();
    bool seen = false;
    for (auto& kv : last)
      if (kv.first == key) { kv.second = op.get(); seen = true; }
    if (!seen) last.emplace_back(key, op.get());
  }
  // one synchronisation per run and device: every operator of a device enqueues on that device's one stream
  // (context.cc: per-(gpu, stream id) pool stream, or the caller's via c2hip_set_stream)
  bool ok = true;
  if (!sync_every_op_)
    for (auto& kv : last) ok = kv.second->Finish() && ok;
  if (ws_) {
    ws_->MarkWritten(produced_);
    ws_->MarkSkipped(skipped_, name_);
  }
  return ok;
}

std::unique_ptr<NetBase> CreateNet(const NetDef& def, Workspace* ws) {
  // caffe2/core/net.cc:35-60 looks `type` up in the net registry; every executor type the
  // reference's configs name ("simple", "dag": detectron/lib/core/config.py MODEL.EXECUTION_TYPE)
  // maps to the in-order single-stream executor here.
  CAFFE_ENFORCE(def.type.empty() || def.type == "simple" || def.type == "dag" ||
                    def.type == "async_dag" || def.type == "async_simple",
                "net type '", def.type, "' is not available for HIPContext");
  return std::unique_ptr<NetBase>(new NetBase(def, ws));
}

// ---- Workspace's net functions (caffe2/core/workspace.cc:180-260) ----------------------------
Workspace::Workspace() {}
Workspace::~Workspace() {
  net_map_.clear();   // operators hold raw Blob*: destroy them before the blobs
  blobs_.clear();
}

NetBase* Workspace::CreateNet(const NetDef& def, bool overwrite) {
  CAFFE_ENFORCE(!def.name.empty(), "NetDef should have a name");
  auto it = net_map_.find(def.name);
  if (it != net_map_.end()) {
    CAFFE_ENFORCE(overwrite, "A net with name ", def.name,
                  " already exists; pass overwrite = true to replace it");
    // workspace.cc:196-201: delete first, so that the old net's operators release their buffers
    net_map_.erase(it);
  }
  std::unique_ptr<NetBase> net = caffe2::CreateNet(def, this);
  NetBase* raw = net.get();
  net_map_[def.name] = std::move(net);
  return raw;
}

NetBase* Workspace::GetNet(const string& name) {
  auto it = net_map_.find(name);
  return it == net_map_.end() ? nullptr : it->second.get();
}

void Workspace::DeleteNet(const string& name) { net_map_.erase(name); }

bool Workspace::RunNet(const string& name) {
  NetBase* net = GetNet(name);
  CAFFE_ENFORCE(net != nullptr, "Network ", name, " does not exist yet.");
  return net->Run();
}

vector<string> Workspace::Nets() const {
  vector<string> names;
  for (const auto& kv : net_map_) names.push_back(kv.first);
  return names;
}

}  // namespace caffe2
