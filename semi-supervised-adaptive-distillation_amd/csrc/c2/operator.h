// operator.h -- OperatorBase / Operator<Context>, the per-device operator
// registries, OpSchema and gradient makers.
//
// Same contract as the reference (caffe2/core/operator.h:45-297 OperatorBase,
// :329-480 Operator<Context>, :652-732 registration macros;
// caffe2/core/operator_schema.h; caffe2/core/operator_gradient.h:63-130,
// :320-335): an operator class has a ctor (const OperatorDef&, Workspace*),
// reads its arguments with GetSingleArgument / GetRepeatedArgument, resolves
// blobs by name at construction and implements `bool RunOnDevice()`.
// REGISTER_HIP_OPERATOR is the MI355X counterpart of REGISTER_CUDA_OPERATOR.
#ifndef C2HIP_OPERATOR_H_
#define C2HIP_OPERATOR_H_

#include <climits>
#include <functional>
#include <map>
#include <memory>
#include <set>

#include "c2/common.h"
#include "c2/context.h"
#include "c2/proto.h"
#include "c2/tensor.h"
#include "c2/workspace.h"

namespace caffe2 {

// ---------------------------------------------------------------------------
// OperatorBase
// ---------------------------------------------------------------------------
class C2HIP_API OperatorBase {
 public:
  OperatorBase(const OperatorDef& def, Workspace* ws);
  virtual ~OperatorBase() {}

  bool HasArgument(const string& name) const { return args_.count(name) != 0; }
  template <typename T> T GetSingleArgument(const string& name, const T& dflt) const;
  template <typename T> bool HasSingleArgumentOfType(const string& name) const;
  template <typename T> vector<T> GetRepeatedArgument(const string& name,
                                                      const vector<T>& dflt = {}) const;

  template <typename T> const T& Input(int idx) const {
    CAFFE_ENFORCE(idx >= 0 && idx < (int)inputs_.size(), "input index ", idx, " out of range");
    return inputs_[idx]->template Get<T>();
  }
  template <typename T> T* Output(int idx) {
    CAFFE_ENFORCE(idx >= 0 && idx < (int)outputs_.size(), "output index ", idx, " out of range");
    return outputs_[idx]->template GetMutable<T>();
  }
  template <typename T> bool InputIsType(int idx) const { return inputs_.at(idx)->template IsType<T>(); }
  int InputSize() const { return (int)inputs_.size(); }
  int OutputSize() const { return (int)outputs_.size(); }
  const vector<const Blob*>& Inputs() const { return inputs_; }
  const vector<Blob*>& Outputs() const { return outputs_; }
  const OperatorDef& def() const { return def_; }
  const OperatorDef& debug_def() const { return def_; }

  // Run = reference semantics (device work finished on return);
  // RunAsync = enqueue only.
  virtual bool Run(int /*stream_id*/ = 0) { CAFFE_NOT_IMPLEMENTED; }
  virtual bool RunAsync(int stream_id = 0) { return Run(stream_id); }
  // what a net needs to synchronise once per run instead of once per operator: does the operator
  // enqueue on a device stream, and wait for + check that stream
  virtual bool OnDeviceStream() const { return false; }
  virtual bool Finish() { return true; }
  // which device's stream that is (-1: none): a net finishes the last operator of EVERY device it touched
  virtual int DeviceKey() const { return -1; }

 protected:
  OperatorDef def_;
  std::map<string, const Argument*> args_;
  vector<const Blob*> inputs_;
  vector<Blob*> outputs_;
};

// ---------------------------------------------------------------------------
// Operator<Context>
// ---------------------------------------------------------------------------
template <class Context>
class Operator : public OperatorBase {
 public:
  Operator(const OperatorDef& def, Workspace* ws)
      : OperatorBase(def, ws), context_(def.device_option) {
    // the constructor runs with the device already selected
    // (caffe2/core/operator.h:333-337)
    context_.SwitchToDevice(0);
  }
  ~Operator() override {}

  const Tensor<Context>& Input(int idx) const {
    return OperatorBase::template Input<Tensor<Context>>(idx);
  }
  Tensor<Context>* Output(int idx) {
    return OperatorBase::template Output<Tensor<Context>>(idx);
  }

  bool Run(int stream_id = 0) final { return RunImpl(stream_id, true); }
  bool RunAsync(int stream_id = 0) final { return RunImpl(stream_id, false); }
  bool OnDeviceStream() const final { return Context::device_type() != CPU; }
  bool Finish() final { return context_.FinishDeviceComputation(); }
  int DeviceKey() const final { return context_.device_key(); }
  virtual bool RunOnDevice() = 0;

 protected:
  bool RunImpl(int stream_id, bool finish) {
    try {
      context_.SwitchToDevice(stream_id);
      const bool ok = RunOnDevice();
      const bool done = finish ? context_.FinishDeviceComputation() : true;
      return ok && done;
    } catch (EnforceNotMet& err) {
      err.AppendMessage("Error from operator: \n" + ProtoDebugString(this->def_));
      throw;
    }
  }
  Context context_;
};

#define USE_OPERATOR_BASE_FUNCTIONS                 \
  using OperatorBase::HasArgument;                  \
  using OperatorBase::GetSingleArgument;            \
  using OperatorBase::HasSingleArgumentOfType;      \
  using OperatorBase::GetRepeatedArgument;          \
  using OperatorBase::InputSize;                    \
  using OperatorBase::OutputSize;                   \
  using OperatorBase::def

#define USE_OPERATOR_FUNCTIONS(context)             \
  USE_OPERATOR_BASE_FUNCTIONS;                      \
  using Operator<context>::context_;                \
  using Operator<context>::Input;                   \
  using Operator<context>::Output

#define USE_OPERATOR_CONTEXT_FUNCTIONS USE_OPERATOR_FUNCTIONS(Context)

// ---------------------------------------------------------------------------
// Registries (caffe2/core/registry.h:55-221)
// ---------------------------------------------------------------------------
using OperatorCreator =
    std::function<std::unique_ptr<OperatorBase>(const OperatorDef&, Workspace*)>;

class C2HIP_API OperatorRegistry {
 public:
  void Register(const string& key, OperatorCreator c);
  bool Has(const string& key) const { return creators_.count(key) != 0; }
  std::unique_ptr<OperatorBase> Create(const string& key, const OperatorDef& d, Workspace* ws) const;
  vector<string> Keys() const;

 private:
  std::map<string, OperatorCreator> creators_;
};

C2HIP_API OperatorRegistry* CPUOperatorRegistry();
C2HIP_API OperatorRegistry* HIPOperatorRegistry();
// CUDA and HIP device types both resolve to the HIP registry.
C2HIP_API OperatorRegistry* RegistryForDevice(int device_type);

struct OperatorRegisterer {
  OperatorRegisterer(OperatorRegistry* r, const string& key, OperatorCreator c) {
    r->Register(key, std::move(c));
  }
};

#define C2HIP_CONCAT_(a, b) a##b
#define C2HIP_CONCAT(a, b) C2HIP_CONCAT_(a, b)
#define C2HIP_ANON(prefix) C2HIP_CONCAT(prefix, __COUNTER__)

#define C2HIP_REGISTER_(registry, key, ...)                                           \
  static ::caffe2::OperatorRegisterer C2HIP_ANON(g_c2hip_op_reg_)(                    \
      registry, key,                                                                  \
      [](const ::caffe2::OperatorDef& d, ::caffe2::Workspace* ws)                     \
          -> std::unique_ptr<::caffe2::OperatorBase> {                                \
        return std::unique_ptr<::caffe2::OperatorBase>(new __VA_ARGS__(d, ws));       \
      })

#define REGISTER_CPU_OPERATOR(name, ...) \
  C2HIP_REGISTER_(::caffe2::CPUOperatorRegistry(), #name, __VA_ARGS__)
#define REGISTER_HIP_OPERATOR(name, ...) \
  C2HIP_REGISTER_(::caffe2::HIPOperatorRegistry(), #name, __VA_ARGS__)
// engine-specific key `Name_ENGINE_<engine>` (caffe2/core/operator.h:700-712)
#define REGISTER_HIP_OPERATOR_WITH_ENGINE(name, engine, ...) \
  C2HIP_REGISTER_(::caffe2::HIPOperatorRegistry(), #name "_ENGINE_" #engine, __VA_ARGS__)

// Schema verify -> engine keys -> plain key (caffe2/core/operator.cc:116-200).
C2HIP_API std::unique_ptr<OperatorBase> CreateOperator(const OperatorDef& def, Workspace* ws);

// ---------------------------------------------------------------------------
// OpSchema (caffe2/core/operator_schema.h)
// ---------------------------------------------------------------------------
class C2HIP_API OpSchema {
 public:
  OpSchema& NumInputs(int n) { min_in_ = max_in_ = n; return *this; }
  OpSchema& NumInputs(int lo, int hi) { min_in_ = lo; max_in_ = hi; return *this; }
  OpSchema& NumOutputs(int n) { min_out_ = max_out_ = n; return *this; }
  OpSchema& NumOutputs(int lo, int hi) { min_out_ = lo; max_out_ = hi; return *this; }
  OpSchema& AllowInplace(std::set<std::pair<int, int>> s) { inplace_ = std::move(s); any_inplace_ = false; return *this; }
  OpSchema& AllowInplaceAny() { any_inplace_ = true; return *this; }
  OpSchema& SetDoc(const string& d) { doc_ = d; return *this; }
  OpSchema& Arg(const string& n, const string& d) { arg_docs_.push_back({n, d}); return *this; }
  OpSchema& Input(int i, const string& n, const string& d) { in_docs_.push_back({MakeString(i, ":", n), d}); return *this; }
  OpSchema& Output(int i, const string& n, const string& d) { out_docs_.push_back({MakeString(i, ":", n), d}); return *this; }
  // throws EnforceNotMet with the reason when def violates the schema
  void Verify(const OperatorDef& def) const;
  int min_input() const { return min_in_; }
  int max_input() const { return max_in_; }
  int min_output() const { return min_out_; }
  int max_output() const { return max_out_; }
  const vector<std::pair<string, string>>& args() const { return arg_docs_; }

 private:
  int min_in_ = 0, max_in_ = INT_MAX, min_out_ = 0, max_out_ = INT_MAX;
  std::set<std::pair<int, int>> inplace_;
  bool any_inplace_ = false;
  string doc_;
  vector<std::pair<string, string>> arg_docs_, in_docs_, out_docs_;
};

class C2HIP_API OpSchemaRegistry {
 public:
  static OpSchema& NewSchema(const string& key);
  static const OpSchema* Schema(const string& key);
};

#define OPERATOR_SCHEMA(name) \
  static ::caffe2::OpSchema& C2HIP_ANON(g_c2hip_schema_) = ::caffe2::OpSchemaRegistry::NewSchema(#name)

// ---------------------------------------------------------------------------
// Gradient makers (caffe2/core/operator_gradient.h)
// ---------------------------------------------------------------------------
struct GradientWrapper {
  string dense_;
  bool IsDense() const { return !dense_.empty(); }
  bool IsEmpty() const { return dense_.empty(); }
};

struct GradientOpsMeta {
  vector<OperatorDef> ops_;
  vector<GradientWrapper> g_input_;
};

class C2HIP_API GradientMakerBase {
 public:
  GradientMakerBase(const OperatorDef& def, const vector<GradientWrapper>& g_output)
      : def_(def), g_output_(g_output), g_input_(def.input.size()) {}
  virtual ~GradientMakerBase() {}
  virtual vector<OperatorDef> GetGradientDefs() { CAFFE_NOT_IMPLEMENTED; }
  virtual GradientOpsMeta Get();

 protected:
  string I(int i) const { CAFFE_ENFORCE(i >= 0 && i < (int)def_.input.size()); return def_.input[i]; }
  string O(int i) const { CAFFE_ENFORCE(i >= 0 && i < (int)def_.output.size()); return def_.output[i]; }
  string GI(int i) {
    CAFFE_ENFORCE(i >= 0 && i < (int)g_input_.size());
    g_input_[i].dense_ = def_.input[i] + "_grad";
    return g_input_[i].dense_;
  }
  // the gradient of input i IS an existing blob (no op writes it), e.g. Sum passes its
  // output gradient to every input (operator_gradient.h SetDense)
  void SetDense(int i, const string& name) {
    CAFFE_ENFORCE(i >= 0 && i < (int)g_input_.size());
    g_input_[i].dense_ = name;
  }
  string GO(int i) const {
    CAFFE_ENFORCE(i >= 0 && i < (int)g_output_.size() && g_output_[i].IsDense(),
                  "Gradient of output ", i, " of ", def_.type, " is not provided");
    return g_output_[i].dense_;
  }
  bool GradOutProvided(int i) const { return i < (int)g_output_.size() && g_output_[i].IsDense(); }
  vector<OperatorDef> SingleGradientDef(const string& type, const string& name,
                                        const vector<string>& inputs,
                                        const vector<string>& outputs,
                                        const vector<Argument>& args) const;
  // default: the gradient op inherits the forward op's arguments
  vector<OperatorDef> SingleGradientDef(const string& type, const string& name,
                                        const vector<string>& inputs,
                                        const vector<string>& outputs) const {
    return SingleGradientDef(type, name, inputs, outputs, def_.arg);
  }
  const OperatorDef& def_;
  const vector<GradientWrapper>& g_output_;
  vector<GradientWrapper> g_input_;
};

struct NoGradient : public GradientMakerBase {
  using GradientMakerBase::GradientMakerBase;
  vector<OperatorDef> GetGradientDefs() override { return {}; }
};

using GradientCreator = std::function<std::unique_ptr<GradientMakerBase>(
    const OperatorDef&, const vector<GradientWrapper>&)>;

class C2HIP_API GradientRegistry {
 public:
  static void Register(const string& key, GradientCreator c);
  static bool Has(const string& key);
  static std::unique_ptr<GradientMakerBase> Create(const string& key, const OperatorDef& def,
                                                   const vector<GradientWrapper>& g_output);
};

struct GradientRegisterer {
  GradientRegisterer(const string& key, GradientCreator c) { GradientRegistry::Register(key, std::move(c)); }
};

#define REGISTER_GRADIENT(name, ...)                                                   \
  static ::caffe2::GradientRegisterer C2HIP_ANON(g_c2hip_grad_reg_)(                   \
      #name, [](const ::caffe2::OperatorDef& d,                                        \
                const ::caffe2::vector<::caffe2::GradientWrapper>& g)                  \
                 -> std::unique_ptr<::caffe2::GradientMakerBase> {                     \
        return std::unique_ptr<::caffe2::GradientMakerBase>(new __VA_ARGS__(d, g));    \
      })
#define NO_GRADIENT(name) REGISTER_GRADIENT(name, ::caffe2::NoGradient)

C2HIP_API GradientOpsMeta GetGradientForOp(const OperatorDef& def,
                                           const vector<GradientWrapper>& g_output);

// ---------------------------------------------------------------------------
// argument access
// ---------------------------------------------------------------------------
template <> inline float OperatorBase::GetSingleArgument<float>(const string& n, const float& d) const {
  auto it = args_.find(n);
  if (it == args_.end()) return d;
  const Argument& a = *it->second;
  CAFFE_ENFORCE(a.has_f || a.has_i, "Argument ", n, " does not have a numeric value");
  return a.has_f ? a.f : (float)a.i;
}
template <> inline int OperatorBase::GetSingleArgument<int>(const string& n, const int& d) const {
  auto it = args_.find(n);
  if (it == args_.end()) return d;
  CAFFE_ENFORCE(it->second->has_i, "Argument ", n, " does not have an integer value");
  return (int)it->second->i;
}
template <> inline int64_t OperatorBase::GetSingleArgument<int64_t>(const string& n, const int64_t& d) const {
  auto it = args_.find(n);
  if (it == args_.end()) return d;
  CAFFE_ENFORCE(it->second->has_i, "Argument ", n, " does not have an integer value");
  return it->second->i;
}
template <> inline bool OperatorBase::GetSingleArgument<bool>(const string& n, const bool& d) const {
  auto it = args_.find(n);
  if (it == args_.end()) return d;
  CAFFE_ENFORCE(it->second->has_i, "Argument ", n, " does not have an integer value");
  return it->second->i != 0;
}
template <> inline string OperatorBase::GetSingleArgument<string>(const string& n, const string& d) const {
  auto it = args_.find(n);
  if (it == args_.end()) return d;
  CAFFE_ENFORCE(it->second->has_s, "Argument ", n, " does not have a string value");
  return it->second->s;
}
template <> inline bool OperatorBase::HasSingleArgumentOfType<float>(const string& n) const {
  auto it = args_.find(n); return it != args_.end() && it->second->has_f;
}
template <> inline bool OperatorBase::HasSingleArgumentOfType<int>(const string& n) const {
  auto it = args_.find(n); return it != args_.end() && it->second->has_i;
}
template <> inline bool OperatorBase::HasSingleArgumentOfType<string>(const string& n) const {
  auto it = args_.find(n); return it != args_.end() && it->second->has_s;
}
template <> inline vector<int> OperatorBase::GetRepeatedArgument<int>(const string& n, const vector<int>& d) const {
  auto it = args_.find(n);
  if (it == args_.end()) return d;
  return vector<int>(it->second->ints.begin(), it->second->ints.end());
}
template <> inline vector<int64_t> OperatorBase::GetRepeatedArgument<int64_t>(const string& n, const vector<int64_t>& d) const {
  auto it = args_.find(n);
  return it == args_.end() ? d : it->second->ints;
}
template <> inline vector<float> OperatorBase::GetRepeatedArgument<float>(const string& n, const vector<float>& d) const {
  auto it = args_.find(n);
  return it == args_.end() ? d : it->second->floats;
}

}  // namespace caffe2
#endif  // C2HIP_OPERATOR_H_
