// tensor.h -- Tensor<Context>, the subset of caffe2/core/tensor.h:109-772 the
// operators on the path use: dims()/dim32()/ndim()/size(), Resize /
// ResizeLike, lazily allocating mutable_data<T>(), type-checked data<T>(),
// raw_data(), IsType<T>(), ShareExternalPointer, CopyFrom.
#ifndef C2HIP_TENSOR_H_
#define C2HIP_TENSOR_H_

#include <atomic>
#include <initializer_list>
#include <memory>

#include "c2/common.h"
#include "c2/context.h"

namespace caffe2 {

// Numbering follows TensorProto::DataType (caffe2.proto:33-49).
enum class DataType : int {
  UNDEFINED = 0, FLOAT = 1, INT32 = 2, BYTE = 3, BOOL = 5, UINT8 = 6,
  INT8 = 7, INT64 = 10, FLOAT16 = 12, DOUBLE = 13
};

struct TypeMeta {
  DataType id = DataType::UNDEFINED;
  size_t itemsize = 0;
  template <typename T> static TypeMeta Make();
  static TypeMeta FromId(int id) {
    switch ((DataType)id) {
      case DataType::FLOAT: return {DataType::FLOAT, 4};
      case DataType::INT32: return {DataType::INT32, 4};
      case DataType::INT64: return {DataType::INT64, 8};
      case DataType::DOUBLE: return {DataType::DOUBLE, 8};
      case DataType::UINT8: return {DataType::UINT8, 1};
      case DataType::INT8: return {DataType::INT8, 1};
      case DataType::BOOL: return {DataType::BOOL, 1};
      case DataType::BYTE: return {DataType::BYTE, 1};
      case DataType::FLOAT16: return {DataType::FLOAT16, 2};
      default: CAFFE_THROW("unsupported data type id ", id);
    }
  }
  bool operator==(const TypeMeta& o) const { return id == o.id; }
  bool operator!=(const TypeMeta& o) const { return id != o.id; }
};
// storage tag of TensorProto::FLOAT16 blobs (caffe2/core/types.h: struct float16 { uint16_t x; })
struct float16 {
  uint16_t x;
};
template <> inline TypeMeta TypeMeta::Make<float>() { return {DataType::FLOAT, 4}; }
template <> inline TypeMeta TypeMeta::Make<float16>() { return {DataType::FLOAT16, 2}; }
template <> inline TypeMeta TypeMeta::Make<int>() { return {DataType::INT32, 4}; }
template <> inline TypeMeta TypeMeta::Make<int64_t>() { return {DataType::INT64, 8}; }
template <> inline TypeMeta TypeMeta::Make<double>() { return {DataType::DOUBLE, 8}; }
template <> inline TypeMeta TypeMeta::Make<uint8_t>() { return {DataType::UINT8, 1}; }
template <> inline TypeMeta TypeMeta::Make<int8_t>() { return {DataType::INT8, 1}; }
template <> inline TypeMeta TypeMeta::Make<bool>() { return {DataType::BOOL, 1}; }

inline uint64_t NextTensorUid() {
  static std::atomic<uint64_t> next{1};
  return next.fetch_add(1, std::memory_order_relaxed);
}

template <class Context>
class Tensor {
 public:
  Tensor() {}
  explicit Tensor(const vector<TIndex>& dims) { Resize(dims); }

  // -- shape -------------------------------------------------------------
  const vector<TIndex>& dims() const { return dims_; }
  int ndim() const { return (int)dims_.size(); }
  TIndex size() const { return size_; }
  TIndex dim(int i) const {
    CAFFE_ENFORCE(i >= 0 && i < ndim(), "dim index ", i, " out of range for ndim ", ndim());
    return dims_[i];
  }
  int dim32(int i) const {
    const TIndex d = dim(i);
    CAFFE_ENFORCE_LT(d, (TIndex)1 << 31);
    return (int)d;
  }
  size_t itemsize() const { return meta_.itemsize; }
  size_t nbytes() const { return (size_t)size_ * meta_.itemsize; }
  const TypeMeta& meta() const { return meta_; }
  template <typename T> bool IsType() const { return meta_ == TypeMeta::Make<T>(); }

  void Resize(const vector<TIndex>& dims) { SetDims(dims); }
  void Resize(const vector<int>& dims) { SetDims(vector<TIndex>(dims.begin(), dims.end())); }
  void Resize(std::initializer_list<TIndex> dims) { SetDims(vector<TIndex>(dims)); }
  template <typename... Ts>
  void Resize(TIndex d0, Ts... rest) { SetDims(vector<TIndex>{d0, (TIndex)rest...}); }
  void Resize() { SetDims(vector<TIndex>()); }
  template <class Other>
  void ResizeLike(const Tensor<Other>& o) { SetDims(o.dims()); }
  void Reshape(const vector<TIndex>& dims) {
    TIndex n = 1;
    for (TIndex d : dims) n *= d;
    CAFFE_ENFORCE_EQ(n, size_, "Reshape must keep the element count");
    dims_ = dims;
  }

  // -- data --------------------------------------------------------------
  template <typename T>
  T* mutable_data() {
    return static_cast<T*>(raw_mutable_data(TypeMeta::Make<T>()));
  }
  void* raw_mutable_data(const TypeMeta& meta) {
    ++version_;   // whoever asks for mutable data is about to write
    const size_t need = (size_t)size_ * meta.itemsize;
    if (data_ && meta_ == meta && capacity_ >= need) return data_.get();
    CAFFE_ENFORCE(!external_ || (meta_ == meta && capacity_ >= need),
                  "tensor over external memory cannot grow or change type");
    meta_ = meta;
    if (!data_ || capacity_ < need) {
      data_.reset(Context::New(need), Context::Delete);
      capacity_ = need;
    }
    return data_.get();
  }
  template <typename T>
  const T* data() const {
    CAFFE_ENFORCE(data_.get() || size_ == 0,
                  "The tensor is of non-zero shape, but its data is not allocated yet.");
    CAFFE_ENFORCE(IsType<T>() || size_ == 0, "Tensor type mismatch: holds data type ",
                  (int)meta_.id, ", requested ", (int)TypeMeta::Make<T>().id);
    return static_cast<const T*>(data_.get());
  }
  const void* raw_data() const {
    CAFFE_ENFORCE(data_.get() || size_ == 0, "tensor data is not allocated yet");
    return data_.get();
  }

  // Wrap memory owned elsewhere (caffe2/core/tensor.h ShareExternalPointer).
  void ShareExternalPointer(void* p, const TypeMeta& meta, size_t capacity_bytes = 0) {
    meta_ = meta;
    data_.reset(p, [](void*) {});
    capacity_ = capacity_bytes ? capacity_bytes : (size_t)size_ * meta.itemsize;
    external_ = true;
    ++version_;
  }

  // Write generation of the tensor: bumped by every raw_mutable_data / mutable_data<T> /
  // ShareExternalPointer, i.e. by every writer that goes through the tensor API (FeedBlob, an
  // operator's Output(i)->mutable_data<T>()).  Operators cache derived data (a packed filter)
  // under (pointer, version, dims).  Memory the tensor does not own can change behind its back
  // (a torch tensor wrapped by c2hip_share_external), so external() tensors are never cached.
  uint64_t version() const { return version_; }
  // a writer that does not go through mutable_data (the C-ABI hands out the device pointer) announces itself
  void MarkWritten() { ++version_; }
  bool external() const { return external_; }
  // process-unique identity of this tensor object (a cache keyed on the data pointer alone could
  // mistake a new tensor that landed on a freed address for the old one)
  uint64_t uid() const { return uid_; }

  template <class SrcContext, class Ctx>
  void CopyFrom(const Tensor<SrcContext>& src, Ctx* context) {
    SetDims(src.dims());
    if (src.meta().id == DataType::UNDEFINED) return;
    void* dst = raw_mutable_data(src.meta());
    context->template CopyBytes<SrcContext, Context>(src.nbytes(), src.raw_data(), dst);
  }

 private:
  void SetDims(const vector<TIndex>& dims) {
    TIndex n = 1;
    for (TIndex d : dims) {
      CAFFE_ENFORCE_GE(d, 0);
      n *= d;
    }
    dims_ = dims;
    size_ = n;
  }

  vector<TIndex> dims_;
  TIndex size_ = 0;   // a default tensor is empty with ndim 0... and size 0
  TypeMeta meta_;
  std::shared_ptr<void> data_;
  size_t capacity_ = 0;
  bool external_ = false;
  uint64_t version_ = 0;
  uint64_t uid_ = NextTensorUid();
};

using TensorCPU = Tensor<CPUContext>;
using TensorHIP = Tensor<HIPContext>;

}  // namespace caffe2
#endif  // C2HIP_TENSOR_H_
