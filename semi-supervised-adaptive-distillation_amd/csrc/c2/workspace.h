// workspace.h -- Blob and Workspace (caffe2/core/blob.h:41-130,
// caffe2/core/workspace.h:63-300): a workspace owns named blobs, a blob owns
// one typed object (here a TensorCPU or a TensorHIP), operators hold raw
// Blob* resolved at construction (caffe2/core/operator.cc:44-66); it also owns the nets created
// in it (workspace.h:209-233; implementation in c2/net.cc).
#ifndef C2HIP_WORKSPACE_H_
#define C2HIP_WORKSPACE_H_

#include <map>
#include <memory>
#include <typeinfo>

#include "c2/common.h"
#include "c2/tensor.h"

namespace caffe2 {

class Blob {
 public:
  Blob() {}
  Blob(const Blob&) = delete;
  Blob& operator=(const Blob&) = delete;
  ~Blob() { Reset(); }

  template <class T> bool IsType() const { return type_ && *type_ == typeid(T); }
  bool empty() const { return ptr_ == nullptr; }
  const char* TypeName() const { return type_ ? type_->name() : "(empty)"; }

  template <class T>
  const T& Get() const {
    CAFFE_ENFORCE(IsType<T>(), "wrong type for the Blob instance. Blob contains ", TypeName(),
                  " while caller expects ", typeid(T).name());
    return *static_cast<const T*>(ptr_);
  }
  template <class T>
  T* GetMutable() {
    if (!IsType<T>()) {
      Reset();
      ptr_ = new T();
      type_ = &typeid(T);
      destroy_ = [](void* p) { delete static_cast<T*>(p); };
    }
    return static_cast<T*>(ptr_);
  }
  void Reset() {
    if (ptr_) destroy_(ptr_);
    ptr_ = nullptr;
    type_ = nullptr;
    destroy_ = nullptr;
  }

 private:
  void* ptr_ = nullptr;
  const std::type_info* type_ = nullptr;
  void (*destroy_)(void*) = nullptr;
};

class NetBase;
struct NetDef;

class C2HIP_API Workspace {
 public:
  Workspace();
  ~Workspace();   // out of line (net.cc): the nets hold operators that hold Blob*, so they go first
  Workspace(const Workspace&) = delete;
  Workspace& operator=(const Workspace&) = delete;

  // caffe2/core/workspace.h:209-233: the workspace owns its nets by name.  CreateNet instantiates
  // (and lowers, net.h) the definition once; RunNet runs the instantiated object.
  NetBase* CreateNet(const NetDef& def, bool overwrite = false);
  NetBase* GetNet(const string& name);
  void DeleteNet(const string& name);
  bool RunNet(const string& name);
  vector<string> Nets() const;

  Blob* CreateBlob(const string& name) {
    auto it = blobs_.find(name);
    if (it != blobs_.end()) return it->second.get();
    Blob* b = new Blob();
    blobs_[name].reset(b);
    return b;
  }
  bool HasBlob(const string& name) const { return blobs_.count(name) != 0; }
  Blob* GetBlob(const string& name) {
    auto it = blobs_.find(name);
    return it == blobs_.end() ? nullptr : it->second.get();
  }
  const Blob* GetBlob(const string& name) const {
    auto it = blobs_.find(name);
    return it == blobs_.end() ? nullptr : it->second.get();
  }
  bool RemoveBlob(const string& name) { stale_.erase(name); return blobs_.erase(name) != 0; }
  // Blobs a LOWERED net no longer produces (net_lowering.cc fuses them away: logits under fuse_sigmoid, the dX
  // feeding a fused ReluGradient, the _grad_autosplit pieces).  The reference could fetch any blob of a net;
  // here a fetch of such a blob must not silently return stale contents: a net marks what its last run skipped,
  // any later writer (another net, an operator run once, FeedBlob) clears the mark, FetchBlob refuses a marked
  // blob and names the remedy (NetDef.external_output / hip_keep_blobs).
  void MarkSkipped(const vector<string>& names, const string& net) { for (const string& n : names) stale_[n] = net; }
  void MarkWritten(const vector<string>& names) { for (const string& n : names) stale_.erase(n); }
  void MarkWritten(const string& name) { stale_.erase(name); }
  const string* SkippedBy(const string& name) const {
    auto it = stale_.find(name);
    return it == stale_.end() ? nullptr : &it->second;
  }
  vector<string> Blobs() const {
    vector<string> names;
    for (const auto& kv : blobs_) names.push_back(kv.first);
    return names;
  }

 private:
  std::map<string, std::unique_ptr<Blob>> blobs_;
  std::map<string, std::unique_ptr<NetBase>> net_map_;
  std::map<string, string> stale_;
};

}  // namespace caffe2
#endif  // C2HIP_WORKSPACE_H_
