// context.h -- CPUContext and HIPContext.
//
// HIPContext is the MI355X counterpart of the reference's CUDAContext
// (caffe2/core/context_gpu.h:136-275): device selection, a per-(gpu,
// stream_id) HIP stream, FinishDeviceComputation() = stream sync + error
// check, static New/Delete allocator, CopyBytes.  Where CUDAContext exposes
// cuda_stream(), this exposes hip_stream().
//
// Differences by design:
//   * one process drives one GPU (process-per-GPU over RCCL replaces the
//     reference's single-process multi-GPU NCCL), so there are
//     no global cross-device mutexes;
//   * a host application that owns the stream (the torch bridge) can install
//     it with HIPContext::SetExternalStream so operators enqueue on the
//     caller's stream instead of a pool stream.
#ifndef C2HIP_CONTEXT_H_
#define C2HIP_CONTEXT_H_

#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>

#include "c2/common.h"
#include "c2/proto.h"

#define HIP_ENFORCE(expr)                                                         \
  do {                                                                            \
    const hipError_t c2_e_ = (expr);                                              \
    if (c2_e_ != hipSuccess)                                                      \
      CAFFE_THROW("HIP error ", (int)c2_e_, " (", hipGetErrorString(c2_e_),       \
                  ") in " #expr);                                                 \
  } while (0)

namespace caffe2 {

class CPUContext {
 public:
  CPUContext() {}
  explicit CPUContext(const DeviceOption& opt) {
    CAFFE_ENFORCE_EQ(opt.device_type, (int)CPU);
  }
  void SwitchToDevice(int /*stream_id*/ = 0) {}
  bool FinishDeviceComputation() { return true; }
  int device_key() const { return -1; }
  static void* New(size_t nbytes) {
    void* p = nullptr;
    if (posix_memalign(&p, 64, nbytes ? nbytes : 1) != 0) CAFFE_THROW("host allocation failed");
    return p;
  }
  static void Delete(void* p) { free(p); }
  template <class Src, class Dst>
  void CopyBytes(size_t n, const void* src, void* dst);
  template <typename T, class Src, class Dst>
  void Copy(size_t n, const T* src, T* dst) {
    CopyBytes<Src, Dst>(n * sizeof(T), src, dst);
  }
  static constexpr int device_type() { return CPU; }
};

class C2HIP_API HIPContext {
 public:
  explicit HIPContext(int gpu_id = -1);
  explicit HIPContext(const DeviceOption& opt);
  void SwitchToDevice(int stream_id = 0);
  bool FinishDeviceComputation();
  int hip_gpu_id() const { return gpu_id_; }
  int device_key() const { return gpu_id_; }
  hipStream_t hip_stream() const;
  static void* New(size_t nbytes);
  static void Delete(void* p);
  template <class Src, class Dst>
  void CopyBytes(size_t n, const void* src, void* dst) {
    if (n == 0) return;
    HIP_ENFORCE(hipMemcpyAsync(dst, src, n, hipMemcpyDefault, hip_stream()));
  }
  template <typename T, class Src, class Dst>
  void Copy(size_t n, const T* src, T* dst) {
    CopyBytes<Src, Dst>(n * sizeof(T), src, dst);
  }
  static constexpr int device_type() { return HIP; }

  // Operators of `gpu_id` enqueue on `stream` until it is reset with nullptr.
  static void SetExternalStream(int gpu_id, hipStream_t stream, bool enabled);
  static int CurrentDevice();

 private:
  int gpu_id_;
  int stream_id_ = 0;
};

template <>
inline void CPUContext::CopyBytes<CPUContext, CPUContext>(size_t n, const void* s, void* d) {
  if (n) memcpy(d, s, n);
}
template <>
inline void CPUContext::CopyBytes<HIPContext, CPUContext>(size_t n, const void* s, void* d) {
  if (n) HIP_ENFORCE(hipMemcpy(d, s, n, hipMemcpyDeviceToHost));
}
template <>
inline void CPUContext::CopyBytes<CPUContext, HIPContext>(size_t n, const void* s, void* d) {
  if (n) HIP_ENFORCE(hipMemcpy(d, s, n, hipMemcpyHostToDevice));
}

}  // namespace caffe2
#endif  // C2HIP_CONTEXT_H_
