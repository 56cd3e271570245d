#include "c2/context.h"

#include <mutex>
#include <unordered_map>

namespace caffe2 {
namespace {

constexpr int kMaxGpus = 16;
constexpr int kMaxStreams = 8;

// Pool streams are created lazily, one per (gpu, stream_id), process-wide:
// a process drives one GPU, and operators of one net share stream 0 unless a
// scheduler assigns stream ids.
struct StreamPool {
  std::mutex mu;
  hipStream_t streams[kMaxGpus][kMaxStreams] = {};
  hipStream_t external[kMaxGpus] = {};
  bool external_on[kMaxGpus] = {};
};
StreamPool& pool() {
  static StreamPool p;
  return p;
}

}  // namespace

int HIPContext::CurrentDevice() {
  int d = 0;
  HIP_ENFORCE(hipGetDevice(&d));
  return d;
}

HIPContext::HIPContext(int gpu_id) : gpu_id_(gpu_id < 0 ? CurrentDevice() : gpu_id) {
  CAFFE_ENFORCE_LT(gpu_id_, kMaxGpus);
}

HIPContext::HIPContext(const DeviceOption& opt) : gpu_id_(opt.gpu_id) {
  CAFFE_ENFORCE(IsGPUDeviceType(opt.device_type), "HIPContext needs a GPU device option");
  CAFFE_ENFORCE_LT(gpu_id_, kMaxGpus);
}

void HIPContext::SwitchToDevice(int stream_id) {
  CAFFE_ENFORCE_LT(stream_id, kMaxStreams);
  stream_id_ = stream_id;
  HIP_ENFORCE(hipSetDevice(gpu_id_));
}

hipStream_t HIPContext::hip_stream() const {
  StreamPool& p = pool();
  std::lock_guard<std::mutex> lock(p.mu);
  if (p.external_on[gpu_id_]) return p.external[gpu_id_];
  hipStream_t& s = p.streams[gpu_id_][stream_id_];
  if (!s) {
    HIP_ENFORCE(hipSetDevice(gpu_id_));
    HIP_ENFORCE(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  }
  return s;
}

void HIPContext::SetExternalStream(int gpu_id, hipStream_t stream, bool enabled) {
  CAFFE_ENFORCE(gpu_id >= 0 && gpu_id < kMaxGpus);
  StreamPool& p = pool();
  std::lock_guard<std::mutex> lock(p.mu);
  p.external[gpu_id] = stream;
  p.external_on[gpu_id] = enabled;
}

bool HIPContext::FinishDeviceComputation() {
  HIP_ENFORCE(hipStreamSynchronize(hip_stream()));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) CAFFE_THROW("Encountered HIP error: ", hipGetErrorString(e));
  return true;
}

void* HIPContext::New(size_t nbytes) {
  void* p = nullptr;
  HIP_ENFORCE(hipMalloc(&p, nbytes ? nbytes : 1));
  return p;
}

void HIPContext::Delete(void* p) {
  if (p) (void)hipFree(p);
}

}  // namespace caffe2
