// NCCLAllreduce / NCCLBroadcast for HIPContext on RCCL -- the collectives the reference's data-parallel net
// contains as operators (detectron/lib/modeling/optimizer.py:72-92 `model.net.NCCLAllreduce(gradients, gradients)`;
// caffe2/caffe2/contrib/nccl/cuda_nccl_op_gpu.cc:25-50,68-118; cuda_nccl_gpu.cc:141-230).
//
// Process model.  The reference drives all GPUs from ONE process: an operator lists the N per-GPU blobs of a
// parameter gradient and nccl::NCCL<T>::AllReduce issues the N per-device calls inside a group (ncclCommInitAll).
// Here a replica is a process (one per GPU), so a rank's net lists ITS blob only: the operator
// with one input is the reference's single-GPU no-op (`if (InputSize() == 1) return true`, :74-75) until the
// process has been given a communicator (c2hip_comm_init: ncclCommInitRank with the id rank 0 created), and from
// then on is this rank's part of the all-reduce over the communicator -- sum, in place when input and output
// coincide, on the operator's own stream.  N > 1 inputs are refused: one process does not own several GPUs' blobs.
//
// RCCL is loaded with dlopen at the first use (librccl.so.1 of the ROCm installation): the kernels and the rest
// of the operator library keep running on a box without it, and `ldd` of the library stays free of it.
#include <dlfcn.h>

#include <mutex>

#include "c2/operator.h"
#include "c2hip_capi.h"

namespace caffe2 {
namespace {

typedef struct ncclComm* ncclComm_t;
struct UniqueId { char internal[128]; };          // NCCL_UNIQUE_ID_BYTES (rccl.h:40-43)
enum { kNcclSum = 0, kNcclFloat16 = 6, kNcclFloat32 = 7 };   // rccl.h:448,465-466

struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string load_error;     // why the library or a symbol is missing, captured once
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
      const char* e = dlerror();          // dlerror() clears itself: read it once, keep the text
      r.load_error += std::string(r.load_error.empty() ? "" : "; ") + (e ? e : name);
    }
    if (!r.lib) return;
    r.load_error.clear();
    auto sym = [&](const char* n) {
      void* p = dlsym(r.lib, n);
      if (!p) {
        const char* e = dlerror();
        r.load_error += std::string(r.load_error.empty() ? "" : "; ") + (e ? e : n);
      }
      return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  });
  CAFFE_ENFORCE(r.lib && r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.Broadcast,
                "RCCL (librccl.so.1) could not be loaded: ", r.load_error.empty() ? "missing symbol" : r.load_error);
  return r;
}

#define RCCL_ENFORCE(call)                                                                         \
  do {                                                                                             \
    const int rc_ = (call);                                                                        \
    CAFFE_ENFORCE(rc_ == 0, "RCCL error ", rc_, ": ",                                             \
                  rccl().GetErrorString ? rccl().GetErrorString(rc_) : "?", " in " #call);         \
  } while (0)

struct Comm {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0, device = -1;
};
Comm& comm() {
  static Comm c;
  return c;
}
std::mutex& comm_mutex() {
  static std::mutex m;
  return m;
}

int nccl_type(const Tensor<HIPContext>& t) {
  if (t.IsType<float>()) return kNcclFloat32;
  if (t.IsType<float16>()) return kNcclFloat16;
  CAFFE_THROW("NCCL operators take float or float16 tensors (cuda_nccl_op_gpu.cc:77-85)");
}

class NCCLAllreduceHIPOp final : public Operator<HIPContext> {
 public:
  using Operator<HIPContext>::Operator;
  bool RunOnDevice() override {
    CAFFE_ENFORCE(InputSize() == 1 && OutputSize() == 1,
                  "NCCLAllreduce: one process drives one GPU here; a rank's net lists its own blob only (the "
                  "reference's N-blob form belongs to its one-process-all-GPUs model)");
    auto& X = Input(0);
    auto* Y = Output(0);
    std::lock_guard<std::mutex> g(comm_mutex());
    Comm& c = comm();
    if (!c.comm) {                                   // no communicator: the reference's single-GPU case
      if (Y != &X) Y->CopyFrom(X, &context_);
      return true;
    }
    const int dt = nccl_type(X);
    Y->ResizeLike(X);
    void* dst = dt == kNcclFloat32 ? (void*)Y->mutable_data<float>() : Y->raw_mutable_data(TypeMeta::Make<float16>());
    RCCL_ENFORCE(rccl().AllReduce(X.raw_data(), dst, (size_t)X.size(), dt, kNcclSum, c.comm, context_.hip_stream()));
    return true;
  }
};

class NCCLBroadcastHIPOp final : public Operator<HIPContext> {
 public:
  using Operator<HIPContext>::Operator;
  bool RunOnDevice() override {
    CAFFE_ENFORCE(InputSize() == 1 && OutputSize() == 1, "NCCLBroadcast: one blob per rank (see NCCLAllreduce)");
    auto& X = Input(0);
    auto* Y = Output(0);
    const int root = GetSingleArgument<int>("root", 0);
    std::lock_guard<std::mutex> g(comm_mutex());
    Comm& c = comm();
    if (!c.comm) {
      if (Y != &X) Y->CopyFrom(X, &context_);
      return true;
    }
    CAFFE_ENFORCE(root >= 0 && root < c.world, "NCCLBroadcast: root ", root, " of ", c.world, " ranks");
    const int dt = nccl_type(X);
    Y->ResizeLike(X);
    void* dst = dt == kNcclFloat32 ? (void*)Y->mutable_data<float>() : Y->raw_mutable_data(TypeMeta::Make<float16>());
    RCCL_ENFORCE(rccl().Broadcast(X.raw_data(), dst, (size_t)X.size(), dt, root, c.comm, context_.hip_stream()));
    return true;
  }
};

}  // namespace

REGISTER_HIP_OPERATOR(NCCLAllreduce, NCCLAllreduceHIPOp);
REGISTER_HIP_OPERATOR(NCCLBroadcast, NCCLBroadcastHIPOp);
// cuda_nccl_op_gpu.cc:201-260: N inputs, N outputs, in place allowed for matching positions
OPERATOR_SCHEMA(NCCLAllreduce).NumInputs(1, INT_MAX).NumOutputs(1, INT_MAX).AllowInplaceAny();
OPERATOR_SCHEMA(NCCLBroadcast).NumInputs(1, INT_MAX).NumOutputs(1, INT_MAX).AllowInplaceAny();

}  // namespace caffe2

// ---- C-ABI: the communicator of this process (include/c2hip_capi.h) -------------------------------------------
using namespace caffe2;

namespace {
thread_local std::string g_comm_error;
template <class F>
int comm_guard(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_comm_error = e.what();
  } catch (...) {
    g_comm_error = "unknown C++ exception";
  }
  return 1;
}
}  // namespace

extern "C" {

const char* c2hip_comm_last_error(void) { return g_comm_error.c_str(); }

int c2hip_comm_unique_id(void* id_out, size_t nbytes) {
  return comm_guard([&] {
    CAFFE_ENFORCE(id_out && nbytes >= sizeof(UniqueId), "c2hip_comm_unique_id: the id is 128 bytes");
    UniqueId id;
    RCCL_ENFORCE(rccl().GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
  });
}

int c2hip_comm_init(const void* id, size_t nbytes, int world, int rank, int device_id) {
  return comm_guard([&] {
    CAFFE_ENFORCE(id && nbytes >= sizeof(UniqueId) && world >= 1 && rank >= 0 && rank < world && device_id >= 0,
                  "c2hip_comm_init: bad arguments");
    std::lock_guard<std::mutex> g(comm_mutex());
    Comm& c = comm();
    CAFFE_ENFORCE(!c.comm, "c2hip_comm_init: this process already has a communicator (c2hip_comm_destroy first)");
    int previous = 0;
    HIP_ENFORCE(hipGetDevice(&previous));
    HIP_ENFORCE(hipSetDevice(device_id));
    UniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t nc = nullptr;
    try {
      RCCL_ENFORCE(rccl().CommInitRank(&nc, world, u, rank));
    } catch (...) {
      (void)hipSetDevice(previous);     // a failed init leaves the caller's device selection alone
      throw;
    }
    c.comm = nc; c.world = world; c.rank = rank; c.device = device_id;
  });
}

int c2hip_comm_world(void) {
  std::lock_guard<std::mutex> g(comm_mutex());
  return comm().comm ? comm().world : 0;
}

int c2hip_comm_destroy(void) {
  return comm_guard([&] {
    std::lock_guard<std::mutex> g(comm_mutex());
    Comm& c = comm();
    if (c.comm) {
      HIP_ENFORCE(hipSetDevice(c.device));
      HIP_ENFORCE(hipDeviceSynchronize());
      RCCL_ENFORCE(rccl().CommDestroy(c.comm));
    }
    c = Comm();
  });
}

}  // extern "C"
