#include "c2/operator.h"

#include <mutex>

namespace caffe2 {

// Inputs must exist; outputs are created (caffe2/core/operator.cc:44-66).
OperatorBase::OperatorBase(const OperatorDef& def, Workspace* ws) : def_(def) {
  for (const Argument& a : def_.arg) {
    CAFFE_ENFORCE(!args_.count(a.name), "Duplicated argument name [", a.name,
                  "] found in operator def: ", ProtoDebugString(def_));
    args_[a.name] = &a;
  }
  for (const string& name : def_.input) {
    const Blob* b = ws->GetBlob(name);
    CAFFE_ENFORCE(b != nullptr, "op ", def_.type, ": Encountered a non-existing input blob: ", name);
    inputs_.push_back(b);
  }
  for (const string& name : def_.output) outputs_.push_back(ws->CreateBlob(name));
}

void OperatorRegistry::Register(const string& key, OperatorCreator c) {
  if (creators_.count(key)) {
    fprintf(stderr, "c2hip: operator key %s registered twice\n", key.c_str());
    abort();
  }
  creators_[key] = std::move(c);
}

std::unique_ptr<OperatorBase> OperatorRegistry::Create(const string& key, const OperatorDef& d,
                                                       Workspace* ws) const {
  auto it = creators_.find(key);
  if (it == creators_.end()) return nullptr;
  return it->second(d, ws);
}

vector<string> OperatorRegistry::Keys() const {
  vector<string> k;
  for (const auto& kv : creators_) k.push_back(kv.first);
  return k;
}

OperatorRegistry* CPUOperatorRegistry() { static OperatorRegistry* r = new OperatorRegistry(); return r; }
OperatorRegistry* HIPOperatorRegistry() { static OperatorRegistry* r = new OperatorRegistry(); return r; }

OperatorRegistry* RegistryForDevice(int device_type) {
  if (device_type == CPU) return CPUOperatorRegistry();
  if (IsGPUDeviceType(device_type)) return HIPOperatorRegistry();
  CAFFE_THROW("Device type ", device_type, " not registered.");
}

namespace {
std::map<string, OpSchema>& schemas() { static auto* m = new std::map<string, OpSchema>(); return *m; }
std::map<string, GradientCreator>& grads() { static auto* m = new std::map<string, GradientCreator>(); return *m; }
}  // namespace

OpSchema& OpSchemaRegistry::NewSchema(const string& key) {
  if (schemas().count(key)) {
    fprintf(stderr, "c2hip: schema %s registered twice\n", key.c_str());
    abort();
  }
  return schemas()[key];
}

const OpSchema* OpSchemaRegistry::Schema(const string& key) {
  auto it = schemas().find(key);
  return it == schemas().end() ? nullptr : &it->second;
}

void OpSchema::Verify(const OperatorDef& def) const {
  const int ni = (int)def.input.size(), no = (int)def.output.size();
  CAFFE_ENFORCE(ni >= min_in_ && ni <= max_in_, "Input size ", ni, " not in range [min=", min_in_,
                ", max=", max_in_, "] for operator ", def.type);
  CAFFE_ENFORCE(no >= min_out_ && no <= max_out_, "Output size ", no, " not in range [min=",
                min_out_, ", max=", max_out_, "] for operator ", def.type);
  for (int i = 0; i < ni; ++i)
    for (int o = 0; o < no; ++o)
      if (def.input[i] == def.output[o]) {
        CAFFE_ENFORCE(any_inplace_ || inplace_.count({i, o}), "Input index ", i,
                      " and output idx ", o, " (", def.input[i],
                      ") are set to be in-place but this is actually not supported by op ",
                      def.type);
      }
}

// caffe2/core/operator.cc:116-200
std::unique_ptr<OperatorBase> CreateOperator(const OperatorDef& def, Workspace* ws) {
  const OpSchema* schema = OpSchemaRegistry::Schema(def.type);
  if (schema) schema->Verify(def);
  const int dev = def.has_device_option ? def.device_option.device_type : (int)CPU;
  OperatorRegistry* reg = RegistryForDevice(dev);
  vector<string> engines;
  if (!def.engine.empty()) {
    size_t start = 0;
    while (start <= def.engine.size()) {          // comma-separated preference list
      const size_t end = def.engine.find(',', start);
      const string e = def.engine.substr(start, end == string::npos ? string::npos : end - start);
      if (!e.empty()) engines.push_back(e);
      if (end == string::npos) break;
      start = end + 1;
    }
  }
  for (const string& e : engines) {
    const string key = def.type + "_ENGINE_" + e;
    if (!reg->Has(key)) continue;
    try {
      auto op = reg->Create(key, def, ws);
      if (op) return op;
    } catch (const UnsupportedOperatorFeature&) {
      // fall through to the next engine / the default implementation
    }
  }
  auto op = reg->Create(def.type, def, ws);
  CAFFE_ENFORCE(op != nullptr, "Cannot create operator of type '", def.type, "' on the device '",
                dev == CPU ? "CPU" : "HIP",
                "'. Verify that implementation for the corresponding device exist. Operator def: ",
                ProtoDebugString(def));
  return op;
}

vector<OperatorDef> GradientMakerBase::SingleGradientDef(
    const string& type, const string& name, const vector<string>& inputs,
    const vector<string>& outputs, const vector<Argument>& args) const {
  OperatorDef g;
  g.type = type;
  g.name = name;
  g.input = inputs;
  g.output = outputs;
  g.arg = args;
  g.is_gradient_op = true;
  return vector<OperatorDef>{g};
}

GradientOpsMeta GradientMakerBase::Get() {
  GradientOpsMeta meta;
  meta.ops_ = GetGradientDefs();
  // gradient ops run where the forward op runs
  for (OperatorDef& g : meta.ops_) {
    if (def_.has_device_option && !g.has_device_option) {
      g.device_option = def_.device_option;
      g.has_device_option = true;
    }
    if (g.engine.empty()) g.engine = def_.engine;
  }
  meta.g_input_ = g_input_;
  return meta;
}

void GradientRegistry::Register(const string& key, GradientCreator c) { grads()[key] = std::move(c); }
bool GradientRegistry::Has(const string& key) { return grads().count(key) != 0; }
std::unique_ptr<GradientMakerBase> GradientRegistry::Create(
    const string& key, const OperatorDef& def, const vector<GradientWrapper>& g) {
  auto it = grads().find(key);
  return it == grads().end() ? nullptr : it->second(def, g);
}

GradientOpsMeta GetGradientForOp(const OperatorDef& def, const vector<GradientWrapper>& g_output) {
  auto maker = GradientRegistry::Create(def.type, def, g_output);
  CAFFE_ENFORCE(maker != nullptr, "Gradient maker for operator ", def.type, " not implemented.");
  return maker->Get();
}

}  // namespace caffe2
