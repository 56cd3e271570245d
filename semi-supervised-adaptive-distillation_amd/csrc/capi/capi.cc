#include "c2hip_capi.h"

#include <cstring>

#include "c2/net.h"
#include "c2/operator.h"
#include "ops/conv_op.h"

using namespace caffe2;

struct c2hip_workspace { Workspace ws; };
struct c2hip_operator { std::unique_ptr<OperatorBase> op; };

namespace {

thread_local std::string g_last_error;

template <class F>
int guarded(F&& f) {
  try {
    f();
    g_last_error.clear();
    return 0;
  } catch (const std::exception& e) {
    g_last_error = e.what();
  } catch (...) {
    g_last_error = "unknown C++ exception";
  }
  return 1;
}

size_t join_to(const vector<string>& items, char* buf, size_t buflen) {
  string s;
  for (size_t i = 0; i < items.size(); ++i) {
    if (i) s += '\n';
    s += items[i];
  }
  if (buf && buflen > 0) {
    const size_t n = s.size() < buflen - 1 ? s.size() : buflen - 1;
    memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return s.size() + 1;
}

vector<string> split_lines(const char* s) {
  vector<string> out;
  if (!s) return out;
  string cur;
  for (const char* p = s;; ++p) {
    if (*p == '\n' || *p == 0) {
      out.push_back(cur);
      cur.clear();
      if (*p == 0) break;
    } else {
      cur.push_back(*p);
    }
  }
  return out;
}

NetDef parse_net(const void* bytes, size_t n) {
  NetDef def;
  CAFFE_ENFORCE(ParseNetDef(bytes, n, &def), "Cannot parse the serialized NetDef");
  return def;
}

size_t pack_defs(const vector<OperatorDef>& ops, void* buf, size_t buflen, int* n_ops) {
  string packed;
  for (const OperatorDef& g : ops) {
    const string s = SerializeOperatorDef(g);
    const uint32_t len = (uint32_t)s.size();
    packed.append((const char*)&len, 4);
    packed += s;
  }
  if (n_ops) *n_ops = (int)ops.size();
  if (buf && buflen >= packed.size() && !packed.empty()) memcpy(buf, packed.data(), packed.size());
  return packed.size();
}

OperatorDef parse_def(const void* bytes, size_t n) {
  OperatorDef def;
  CAFFE_ENFORCE(ParseOperatorDef(bytes, n, &def), "Cannot parse the serialized OperatorDef");
  return def;
}

}  // namespace

extern "C" {

const char* c2hip_last_error(void) { return g_last_error.c_str(); }

c2hip_workspace* c2hip_workspace_create(void) { return new c2hip_workspace(); }
void c2hip_workspace_destroy(c2hip_workspace* ws) { delete ws; }
int c2hip_has_blob(c2hip_workspace* ws, const char* name) { return ws->ws.HasBlob(name) ? 1 : 0; }
int c2hip_remove_blob(c2hip_workspace* ws, const char* name) { return ws->ws.RemoveBlob(name) ? 0 : 1; }
size_t c2hip_blobs(c2hip_workspace* ws, char* buf, size_t buflen) {
  return join_to(ws->ws.Blobs(), buf, buflen);
}

// FetchBlob / blob info of a blob that a lowered net fused away (workspace.h, MarkSkipped)
static void ThrowIfSkipped(const Workspace& w, const char* name) {
  if (const string* net = w.SkippedBy(name))
    CAFFE_THROW("blob ", name, " was not produced by the last run of net ", *net,
                ": the lowering fused its producer away.  List it in NetDef.external_output or in the net "
                "argument hip_keep_blobs (or create the net with hip_lowering = 0) to keep it");
}

int c2hip_feed_blob(c2hip_workspace* ws, const char* name, const void* host_data,
                    const int64_t* dims, int ndim, int dtype, int device_type, int device_id) {
  return guarded([&] {
    CAFFE_ENFORCE(ndim >= 0 && ndim <= C2HIP_MAX_DIMS);
    const TypeMeta meta = TypeMeta::FromId(dtype);
    const vector<TIndex> d(dims, dims + ndim);
    Blob* blob = ws->ws.CreateBlob(name);
    ws->ws.MarkWritten(name);
    if (device_type == C2HIP_CPU) {
      TensorCPU* t = blob->GetMutable<TensorCPU>();
      t->Resize(d);
      void* dst = t->raw_mutable_data(meta);
      if (t->nbytes()) memcpy(dst, host_data, t->nbytes());
    } else {
      CAFFE_ENFORCE(IsGPUDeviceType(device_type), "unknown device type ", device_type);
      HIPContext ctx(device_id);
      ctx.SwitchToDevice(0);
      TensorHIP* t = blob->GetMutable<TensorHIP>();
      t->Resize(d);
      void* dst = t->raw_mutable_data(meta);
      if (t->nbytes()) {
        HIP_ENFORCE(hipMemcpyAsync(dst, host_data, t->nbytes(), hipMemcpyHostToDevice,
                                   ctx.hip_stream()));
        ctx.FinishDeviceComputation();
      }
    }
  });
}

int c2hip_blob_info(c2hip_workspace* ws, const char* name, int* dtype, int* device_type,
                    int* ndim, int64_t* dims) {
  return guarded([&] {
    const Blob* b = ws->ws.GetBlob(name);
    ThrowIfSkipped(ws->ws, name);
    CAFFE_ENFORCE(b != nullptr, "Can't find blob: ", name);
    auto fill = [&](const auto& t, int dev) {
      *dtype = (int)t.meta().id;
      *device_type = dev;
      *ndim = t.ndim();
      CAFFE_ENFORCE_LE(t.ndim(), C2HIP_MAX_DIMS);
      for (int i = 0; i < t.ndim(); ++i) dims[i] = t.dim(i);
    };
    if (b->IsType<TensorCPU>()) fill(b->Get<TensorCPU>(), C2HIP_CPU);
    else if (b->IsType<TensorHIP>()) fill(b->Get<TensorHIP>(), C2HIP_HIP);
    else CAFFE_THROW("blob ", name, " does not hold a tensor");
  });
}

int c2hip_fetch_blob(c2hip_workspace* ws, const char* name, void* host_out, size_t nbytes) {
  return guarded([&] {
    const Blob* b = ws->ws.GetBlob(name);
    ThrowIfSkipped(ws->ws, name);
    CAFFE_ENFORCE(b != nullptr, "Can't find blob: ", name);
    if (b->IsType<TensorCPU>()) {
      const TensorCPU& t = b->Get<TensorCPU>();
      CAFFE_ENFORCE_EQ(t.nbytes(), nbytes);
      if (nbytes) memcpy(host_out, t.raw_data(), nbytes);
    } else if (b->IsType<TensorHIP>()) {
      const TensorHIP& t = b->Get<TensorHIP>();
      CAFFE_ENFORCE_EQ(t.nbytes(), nbytes);
      // all pool / external streams of the device may hold pending writers
      HIP_ENFORCE(hipDeviceSynchronize());
      if (nbytes) HIP_ENFORCE(hipMemcpy(host_out, t.raw_data(), nbytes, hipMemcpyDeviceToHost));
    } else {
      CAFFE_THROW("blob ", name, " does not hold a tensor");
    }
  });
}

void* c2hip_blob_data_ptr(c2hip_workspace* ws, const char* name) {
  void* p = nullptr;
  guarded([&] {
    const Blob* b = ws->ws.GetBlob(name);
    CAFFE_ENFORCE(b != nullptr, "Can't find blob: ", name);
    CAFFE_ENFORCE(b->IsType<TensorHIP>(), "blob ", name, " is not a HIP tensor");
    // the caller gets a WRITABLE device pointer (a torch view may update a filter through it): count it as a
    // write, so that caches keyed on the tensor's write generation (filter_pack_cache.h) repack
    TensorHIP& t = const_cast<TensorHIP&>(b->Get<TensorHIP>());
    t.MarkWritten();
    p = const_cast<void*>(t.raw_data());
  });
  return p;
}

int c2hip_share_external(c2hip_workspace* ws, const char* name, void* device_ptr,
                         const int64_t* dims, int ndim, int dtype, int device_id) {
  (void)device_id;
  return guarded([&] {
    CAFFE_ENFORCE(ndim >= 0 && ndim <= C2HIP_MAX_DIMS);
    Blob* blob = ws->ws.CreateBlob(name);
    blob->Reset();
    TensorHIP* t = blob->GetMutable<TensorHIP>();
    t->Resize(vector<TIndex>(dims, dims + ndim));
    t->ShareExternalPointer(device_ptr, TypeMeta::FromId(dtype));
  });
}

int c2hip_run_operator_once(c2hip_workspace* ws, const void* def_bytes, size_t n) {
  return guarded([&] {
    const OperatorDef def = parse_def(def_bytes, n);
    auto op = CreateOperator(def, &ws->ws);
    CAFFE_ENFORCE(op->Run(), "Error when running operator ", def.type);
    ws->ws.MarkWritten(def.output);
  });
}

c2hip_operator* c2hip_create_operator(c2hip_workspace* ws, const void* def_bytes, size_t n) {
  c2hip_operator* h = nullptr;
  guarded([&] {
    const OperatorDef def = parse_def(def_bytes, n);
    auto op = CreateOperator(def, &ws->ws);
    h = new c2hip_operator{std::move(op)};
  });
  return h;
}

int c2hip_run_operator(c2hip_operator* op, int sync) {
  return guarded([&] {
    const bool ok = sync ? op->op->Run() : op->op->RunAsync();
    CAFFE_ENFORCE(ok, "Error when running operator ", op->op->def().type);
  });
}

void c2hip_destroy_operator(c2hip_operator* op) { delete op; }

int c2hip_create_net(c2hip_workspace* ws, const void* netdef_bytes, size_t n, int overwrite) {
  return guarded([&] { ws->ws.CreateNet(parse_net(netdef_bytes, n), overwrite != 0); });
}

int c2hip_run_net(c2hip_workspace* ws, const char* name, int num_iter) {
  return guarded([&] {
    for (int i = 0; i < num_iter; ++i)
      CAFFE_ENFORCE(ws->ws.RunNet(name), "Error running net ", name);
  });
}

int c2hip_run_net_once(c2hip_workspace* ws, const void* netdef_bytes, size_t n) {
  return guarded([&] {
    // pybind_state.cc run_net_once -> Workspace::RunNetOnce: a temporary net, run, destroyed
    std::unique_ptr<NetBase> net = CreateNet(parse_net(netdef_bytes, n), &ws->ws);
    CAFFE_ENFORCE(net->Run(), "Error running net ", net->Name());
  });
}

int c2hip_delete_net(c2hip_workspace* ws, const char* name) {
  return guarded([&] { ws->ws.DeleteNet(name); });
}

size_t c2hip_nets(c2hip_workspace* ws, char* buf, size_t buflen) {
  return join_to(ws->ws.Nets(), buf, buflen);
}

size_t c2hip_net_lowered_ops(c2hip_workspace* ws, const char* name, void* buf, size_t buflen, int* n_ops) {
  size_t need = 0;
  guarded([&] {
    NetBase* net = ws->ws.GetNet(name);
    CAFFE_ENFORCE(net != nullptr, "Network ", name, " does not exist yet.");
    need = pack_defs(net->lowered_ops(), buf, buflen, n_ops);
  });
  return need;
}

size_t c2hip_lower_net(const void* netdef_bytes, size_t n, void* buf, size_t buflen, int* n_ops,
                       char* report_buf, size_t report_buflen) {
  size_t need = 0;
  guarded([&] {
    LoweringReport rep;
    const NetDef def = parse_net(netdef_bytes, n);
    LoweringOptions opt = LoweringOptionsFor(def);           // no dtype probe: filters are taken to be fp32
    const vector<OperatorDef> ops = LowerNet(def, opt, &rep);
    need = pack_defs(ops, buf, buflen, n_ops);
    if (report_buf && report_buflen) {
      const string r = rep.ToString();
      const size_t m = r.size() < report_buflen - 1 ? r.size() : report_buflen - 1;
      memcpy(report_buf, r.data(), m);
      report_buf[m] = 0;
    }
  });
  return need;
}

long long c2hip_counter(const char* name) {
  const string n = name ? name : "";
  if (n == "filter_packs") return g_filter_packs_issued.load();
  if (n == "conv_launch_calls") return g_conv_launch_calls.load();
  return -1;
}

size_t c2hip_registered_operators(int device_type, char* buf, size_t buflen) {
  size_t need = 0;
  guarded([&] { need = join_to(RegistryForDevice(device_type)->Keys(), buf, buflen); });
  return need;
}

int c2hip_has_schema(const char* op_type, int* min_in, int* max_in, int* min_out, int* max_out) {
  const OpSchema* s = OpSchemaRegistry::Schema(op_type);
  if (!s) return 0;
  if (min_in) *min_in = s->min_input();
  if (max_in) *max_in = s->max_input();
  if (min_out) *min_out = s->min_output();
  if (max_out) *max_out = s->max_output();
  return 1;
}

int c2hip_get_gradient_defs(const void* def_bytes, size_t n, const char* g_output_names,
                            void* out_defs, size_t out_defs_cap, size_t* out_defs_len,
                            int* n_defs, char* out_g_inputs, size_t out_g_inputs_cap) {
  return guarded([&] {
    const OperatorDef def = parse_def(def_bytes, n);
    vector<GradientWrapper> g_out(def.output.size());
    const vector<string> names = split_lines(g_output_names);
    for (size_t i = 0; i < g_out.size() && i < names.size(); ++i) g_out[i].dense_ = names[i];
    const GradientOpsMeta meta = GetGradientForOp(def, g_out);
    string packed;
    for (const OperatorDef& g : meta.ops_) {
      const string s = SerializeOperatorDef(g);
      const uint32_t len = (uint32_t)s.size();
      packed.append((const char*)&len, 4);
      packed += s;
    }
    CAFFE_ENFORCE_LE(packed.size(), out_defs_cap, "gradient def buffer too small");
    if (!packed.empty()) memcpy(out_defs, packed.data(), packed.size());
    *out_defs_len = packed.size();
    *n_defs = (int)meta.ops_.size();
    vector<string> gi;
    for (const GradientWrapper& w : meta.g_input_) gi.push_back(w.dense_);
    CAFFE_ENFORCE_LE(join_to(gi, out_g_inputs, out_g_inputs_cap), out_g_inputs_cap);
  });
}

int c2hip_set_stream(int device_id, void* hip_stream, int enabled) {
  return guarded([&] { HIPContext::SetExternalStream(device_id, (hipStream_t)hip_stream, enabled != 0); });
}

int c2hip_device_synchronize(int device_id) {
  return guarded([&] {
    HIP_ENFORCE(hipSetDevice(device_id));
    HIP_ENFORCE(hipDeviceSynchronize());
  });
}

}  // extern "C"
