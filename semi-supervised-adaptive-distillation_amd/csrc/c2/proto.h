// proto.h -- plain-struct mirror of the messages an operator / a net sees
// (caffe2/proto/caffe2.proto:97-106 Argument, :128-138 DeviceOption,
// :142-172 OperatorDef, :176-215 NetDef) plus a hand-written protobuf wire codec, so the C-ABI
// accepts exactly the bytes `op.SerializeToString()` produces in the
// reference's Python layer (caffe2/python/pybind_state.cc RunOperatorOnce).
// No protobuf dependency.
#ifndef C2HIP_PROTO_H_
#define C2HIP_PROTO_H_

#include "c2/common.h"

namespace caffe2 {

// caffe2.proto:112-116 has CPU/CUDA/MKLDNN/OPENGL only; HIP = 6 is the value
// later upstream Caffe2 assigned.  An OperatorDef that says CUDA is served by
// the HIP registry here (there is no CUDA on an MI355X), which is what lets
// the reference's graph builder run unchanged.
enum DeviceType { CPU = 0, CUDA = 1, MKLDNN = 2, OPENGL = 3, HIP = 6 };

struct Argument {
  string name;
  bool has_f = false, has_i = false, has_s = false;
  float f = 0.f;
  int64_t i = 0;
  string s;
  vector<float> floats;
  vector<int64_t> ints;
  vector<string> strings;
};

struct DeviceOption {
  int device_type = CPU;   // field 1
  int gpu_id = 0;          // field 2 (cuda_gpu_id); also field 6 (hip_gpu_id)
  uint32_t random_seed = 0;
  string node_name;
};

struct OperatorDef {
  vector<string> input;          // 1
  vector<string> output;         // 2
  string name;                   // 3
  string type;                   // 4
  vector<Argument> arg;          // 5
  DeviceOption device_option;    // 6
  bool has_device_option = false;
  string engine;                 // 7
  vector<string> control_input;  // 8
  bool is_gradient_op = false;   // 9
};

// caffe2.proto:176-215.  `type` names the executor ("simple", "dag", ...: caffe2/core/net.cc
// REGISTER_NET); `arg` carries executor options; external_input / external_output declare the
// blobs that must hold their real contents when a run returns.
struct NetDef {
  string name;                     // 1
  vector<OperatorDef> op;          // 2
  string type;                     // 3
  int num_workers = 0;             // 4 (deprecated upstream; detector.py:67 still sets it)
  DeviceOption device_option;      // 5
  bool has_device_option = false;
  vector<Argument> arg;            // 6
  vector<string> external_input;   // 7
  vector<string> external_output;  // 8
};

C2HIP_API bool ParseOperatorDef(const void* data, size_t n, OperatorDef* out);
C2HIP_API bool ParseNetDef(const void* data, size_t n, NetDef* out);
C2HIP_API string SerializeNetDef(const NetDef& def);
C2HIP_API string SerializeOperatorDef(const OperatorDef& def);
C2HIP_API string ProtoDebugString(const OperatorDef& def);

inline Argument MakeArgument(const string& name, float v) {
  Argument a; a.name = name; a.has_f = true; a.f = v; return a;
}
inline Argument MakeArgument(const string& name, int64_t v) {
  Argument a; a.name = name; a.has_i = true; a.i = v; return a;
}
inline Argument MakeArgument(const string& name, int v) { return MakeArgument(name, (int64_t)v); }
inline Argument MakeArgument(const string& name, const string& v) {
  Argument a; a.name = name; a.has_s = true; a.s = v; return a;
}

inline bool IsGPUDeviceType(int t) { return t == CUDA || t == HIP; }

}  // namespace caffe2
#endif  // C2HIP_PROTO_H_
