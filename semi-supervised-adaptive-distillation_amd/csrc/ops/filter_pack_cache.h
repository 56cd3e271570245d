// filter_pack_cache.h -- packed-filter cache of the convolution operators.
//
// The 3x3 engines read the filter in MFMA operand order (kernels/conv3x3.hip, conv3x3_winograd.hip);
// the pack is a function of the filter blob alone.  The reference's counterpart is cuDNN's filter
// descriptor, rebuilt only when the shape changes (conv_op_cudnn.cc:356-361); here the packed copy is
// rebuilt only when the blob was WRITTEN since -- Tensor::version(), bumped by every mutable_data<T>() /
// FeedBlob (c2/tensor.h) -- so a frozen teacher packs once and a trained filter once per update, not
// once per Run and FPN level.  Tensors over memory the workspace does not own are never cached.
#ifndef C2HIP_FILTER_PACK_CACHE_H_
#define C2HIP_FILTER_PACK_CACHE_H_

#include <map>

#include "c2/operator.h"
#include "ssad_kernels.h"

namespace caffe2 {

class FilterPackCache {
 public:
  enum Kind { WINO_FWD = 0, WINO_DGRAD = 1, DIRECT_FWD = 2, DIRECT_DGRAD = 3, WINO24_FWD = 4, WINO24_DGRAD = 5,
              SPLIT_FWD = 6, SPLIT_DGRAD = 7 };     // 6 / 7: the split-operand engine's packs (conv3x3_split.hip)

  // Queue `filter` ([M][C][3][3], fp32) for layout `kind`; Flush() issues the packs that are stale
  // (one multi-filter launch for the Winograd layouts) on `stream`; Packed() is valid after it.
  void Want(const Tensor<HIPContext>& filter, Kind kind) {
    const Key key{filter.raw_data(), (int)kind};
    if (entries_.size() >= kMaxEntries && !entries_.count(key)) {
      // filters that were reallocated leave entries under their old addresses: drop everything that is not queued
      // (an operator packs a handful of filters; the next Want() of a live one repacks it once)
      for (auto it = entries_.begin(); it != entries_.end();) it = it->second.queued ? std::next(it) : entries_.erase(it);
    }
    Entry& e = entries_[key];
    const int M = filter.dim32(0), C = filter.dim32(1);
    const bool fresh = e.valid && !filter.external() && e.uid == filter.uid() && e.version == filter.version() &&
                       e.M == M && e.C == C;
    if (fresh || e.queued) return;
    e.M = M;
    e.C = C;
    e.version = filter.version();
    e.uid = filter.uid();
    const bool dgrad = kind == WINO_DGRAD || kind == DIRECT_DGRAD || kind == WINO24_DGRAD || kind == SPLIT_DGRAD;
    const bool wino = kind == WINO_FWD || kind == WINO_DGRAD;
    const int po = dgrad ? C : M, pi = dgrad ? M : C;       // the pack's (outputs, inputs)
    e.packed.Resize((TIndex)((kind == SPLIT_FWD || kind == SPLIT_DGRAD) ? ssad_conv_split_filter_floats(po, pi)
                             : (kind == WINO24_FWD || kind == WINO24_DGRAD) ? ssad_conv_wino24_filter_floats(po, pi)
                             : wino ? ssad_conv_wino_filter_floats(po, pi) : ssad_conv_packed_filter_floats(po, pi)));
    e.packed.mutable_data<float>();
    e.queued = true;
    e.src = filter.data<float>();
    queue_.push_back({filter.raw_data(), (int)kind});
  }

  void Flush(hipStream_t stream) {
    vector<ssad_pack_entry> wino, wino24, split;
    for (const Key& k : queue_) {
      Entry& e = entries_[k];
      float* p = e.packed.mutable_data<float>();
      switch ((Kind)k.second) {
        case WINO_FWD: wino.push_back({e.src, e.M, e.C, p, nullptr}); break;
        case WINO_DGRAD: wino.push_back({e.src, e.M, e.C, nullptr, p}); break;
        case WINO24_FWD: wino24.push_back({e.src, e.M, e.C, p, nullptr}); break;
        case WINO24_DGRAD: wino24.push_back({e.src, e.M, e.C, nullptr, p}); break;
        case SPLIT_FWD: split.push_back({e.src, e.M, e.C, p, nullptr}); break;
        case SPLIT_DGRAD: split.push_back({e.src, e.M, e.C, nullptr, p}); break;
        case DIRECT_FWD:
          CAFFE_ENFORCE_EQ(ssad_conv_pack_filter(e.src, e.M, e.C, p, nullptr, stream), 0);
          break;
        case DIRECT_DGRAD:
          CAFFE_ENFORCE_EQ(ssad_conv_pack_filter(e.src, e.M, e.C, nullptr, p, stream), 0);
          break;
      }
      e.queued = false;
      e.valid = true;
      ++packs_issued_;
    }
    if (!wino.empty())
      CAFFE_ENFORCE_EQ(ssad_conv_wino_pack_filters(wino.data(), (int)wino.size(), stream), 0,
                       "filter pack launch failed");
    if (!wino24.empty())
      CAFFE_ENFORCE_EQ(ssad_conv_wino24_pack_filters(wino24.data(), (int)wino24.size(), stream), 0,
                       "filter pack launch failed");
    if (!split.empty())
      CAFFE_ENFORCE_EQ(ssad_conv_split_pack_filters(split.data(), (int)split.size(), stream), 0,
                       "filter pack launch failed");
    queue_.clear();
  }

  const float* Packed(const Tensor<HIPContext>& filter, Kind kind) const {
    auto it = entries_.find({filter.raw_data(), (int)kind});
    CAFFE_ENFORCE(it != entries_.end() && it->second.valid, "filter was not packed");
    return it->second.packed.data<float>();
  }

  long long packs_issued() const { return packs_issued_; }

 private:
  using Key = std::pair<const void*, int>;
  struct Entry {
    Tensor<HIPContext> packed;
    const float* src = nullptr;
    uint64_t version = 0, uid = 0;
    int M = 0, C = 0;
    bool valid = false, queued = false;
  };
  static constexpr size_t kMaxEntries = 256;
  std::map<Key, Entry> entries_;
  vector<Key> queue_;
  long long packs_issued_ = 0;
};

// hip_algo = "split" (assigned by the net lowering): the split-operand engine for >= 128-wide problems.  Its
// workspace (the split copy of the launch's inputs) belongs to the operator and grows to the largest launch seen.
class SplitEngine {
 public:
  int Run(const ssad_conv_level* lv, int n, const float* packed, const float* bias, int M, int C, int flags,
          hipStream_t s) {
    const size_t need = ssad_conv3x3_split_workspace_bytes(lv, n, C);
    CAFFE_ENFORCE(need > 0, "split-operand engine: unsupported geometry");
    if ((size_t)ws_.size() < need) ws_.Resize((TIndex)need);
    return ssad_conv3x3_forward_split(lv, n, packed, bias, M, C, flags, ws_.mutable_data<uint8_t>(), need, nullptr,
                                      nullptr, s);
  }

 private:
  Tensor<HIPContext> ws_;
};

}  // namespace caffe2
#endif  // C2HIP_FILTER_PACK_CACHE_H_
