// ConvGroup / ConvGradientGroup for HIPContext -- the operators the net lowering (ops/net_lowering.cc)
// emits for convolutions that are ready together: ConvShared's five FPN levels of one filter
// (detectron/lib/modeling/detector.py:449-482, retinanet_heads.py:97-152), the cls and bbox tower layer
// of equal depth.  The reference issues one cuDNN call per (level, layer) (conv_op_cudnn.cc:567-617,
// :1011-1058) and sums the five filter-gradient pieces of a shared filter with a separate Sum operator
// (caffe2/python/core.py:706-741); here a group is ONE multi-problem launch per (Cout, Cin) class -- the
// small levels fill the tail of the large ones -- and the filter-gradient launch reduces over every level
// of its filter, so the pieces and their Sum never exist.
//
//   ConvGroup          inputs  [X_0, W_0, (b_0), X_1, W_1, (b_1), ...]      outputs [Y_0, Y_1, ...]
//   ConvGradientGroup  inputs  [X_0, W_0, dY_0, X_1, W_1, dY_1, ...]
//                      outputs [dW_f ...] [db_f ...] [dX_i ...]   f = distinct filters, arg filter_index[i]
//
// Arguments are the member operators' own (conv_pool_op_base.h:45-194 geometry, fuse_relu,
// relu_grad_on_input, no_bias, hip_algo); the geometry must be 3x3 / stride 1 / pad 1 and the blobs fp32
// (the lowering only groups those).  Packed filters come from the FilterPackCache.
#include <atomic>

#include "ops/conv_op.h"
#include "ops/filter_pack_cache.h"
#include "ssad_kernels.h"

namespace caffe2 {

std::atomic<long long> g_filter_packs_issued{0};   // c2hip_counter("filter_packs")
std::atomic<long long> g_conv_launch_calls{0};     // c2hip_counter("conv_launch_calls")

namespace {

struct Problem {
  const Tensor<HIPContext>* x = nullptr;
  const Tensor<HIPContext>* w = nullptr;
  const Tensor<HIPContext>* b = nullptr;     // forward: bias; gradient: dY
  Tensor<HIPContext>* y = nullptr;           // forward: Y; gradient: dX (or null)
  int N = 0, C = 0, H = 0, W = 0, M = 0;
};

void CheckSubnetProblem(const Problem& p) {
  CAFFE_ENFORCE(p.x->IsType<float>() && p.w->IsType<float>(),
                "ConvGroup / ConvGradientGroup are fp32 operators (float16 blobs run as single Conv operators)");
  CAFFE_ENFORCE_EQ(p.x->ndim(), 4);
  CAFFE_ENFORCE_EQ(p.w->ndim(), 4);
  CAFFE_ENFORCE(p.w->dim32(1) == p.C, "Convolution op: input channels does not match: # of input channels ", p.C,
                " is not equal to kernel channels:", p.w->dim32(1));
  CAFFE_ENFORCE(p.w->dim32(2) == 3 && p.w->dim32(3) == 3);
}

void Dims(Problem* p) {
  p->N = p->x->dim32(0); p->C = p->x->dim32(1); p->H = p->x->dim32(2); p->W = p->x->dim32(3);
  p->M = p->w->dim32(0);
}

// problems of equal (outputs, inputs) share launches; classes in order of first appearance
vector<vector<int>> Classes(const vector<Problem>& probs) {
  vector<vector<int>> classes;
  for (int i = 0; i < (int)probs.size(); ++i) {
    bool placed = false;
    for (auto& c : classes)
      if (probs[c[0]].M == probs[i].M && probs[c[0]].C == probs[i].C) { c.push_back(i); placed = true; break; }
    if (!placed) classes.push_back({i});
  }
  return classes;
}

}  // namespace

class ConvGroupOp final : public Operator<HIPContext> {
 public:
  ConvGroupOp(const OperatorDef& def, Workspace* ws)
      : Operator<HIPContext>(def, ws),
        geom_(ParseConvGeometry(*this)),
        fuse_relu_(GetSingleArgument<int>("fuse_relu", 0)),
        fuse_sigmoid_(GetSingleArgument<int>("fuse_sigmoid", 0)),
        algo_(GetSingleArgument<string>("hip_algo", "auto")) {
    CAFFE_ENFORCE(IsSubnetGeometry(geom_), "ConvGroup implements kernel 3 / stride 1 / pad 1 / group 1 / NCHW");
    CAFFE_ENFORCE(OutputSize() >= 1 && InputSize() % OutputSize() == 0, "ConvGroup: [X, W, (b)] per output");
    per_ = InputSize() / OutputSize();
    CAFFE_ENFORCE(per_ == 2 || per_ == 3, "ConvGroup: [X, W, (b)] per output");
  }

  bool RunOnDevice() override {
    const int k = OutputSize();
    vector<Problem> probs(k);
    for (int i = 0; i < k; ++i) {
      Problem& p = probs[i];
      p.x = &Input(per_ * i);
      p.w = &Input(per_ * i + 1);
      p.b = per_ == 3 ? &Input(per_ * i + 2) : nullptr;
      Dims(&p);
      CheckSubnetProblem(p);
      if (p.b) CAFFE_ENFORCE(p.b->ndim() == 1 && p.b->dim32(0) == p.M);
      p.y = Output(i);
      p.y->Resize(p.N, p.M, p.H, p.W);
    }
    hipStream_t s = context_.hip_stream();
    const long long before = cache_.packs_issued();
    // hip_algo = "winograd24" (set by the net lowering): the F(2x4, 3x3) engine for the problems it serves
    // (>= 128 outputs), "auto" for the rest
    auto kind_of = [&](const Problem& p) {
      // "split": the split-operand engine where it measured ahead (>= 256 outputs), F(2x4) for 128..255, F(2x2) below
      if (algo_ == "split")
        return p.M >= 256 ? FilterPackCache::SPLIT_FWD : p.M >= 128 ? FilterPackCache::WINO24_FWD : FilterPackCache::WINO_FWD;
      if (algo_ == "winograd24") return p.M >= 128 ? FilterPackCache::WINO24_FWD : FilterPackCache::WINO_FWD;
      return UseWinograd(algo_, p.M) ? FilterPackCache::WINO_FWD : FilterPackCache::DIRECT_FWD;
    };
    for (const Problem& p : probs) cache_.Want(*p.w, kind_of(p));
    cache_.Flush(s);
    g_filter_packs_issued += cache_.packs_issued() - before;
    const int flags = (fuse_relu_ ? SSAD_CONV_RELU : 0) | (fuse_sigmoid_ ? SSAD_CONV_SIGMOID : 0);
    for (const vector<int>& cls : Classes(probs)) {
      const Problem& p0 = probs[cls[0]];
      const auto kind = kind_of(p0);
      for (size_t at = 0; at < cls.size(); at += SSAD_MAX_CONV_PROBLEMS) {
        const int n = (int)std::min<size_t>(SSAD_MAX_CONV_PROBLEMS, cls.size() - at);
        ssad_conv_level lv[SSAD_MAX_CONV_PROBLEMS];
        for (int j = 0; j < n; ++j) {
          const Problem& p = probs[cls[at + j]];
          lv[j] = ssad_conv_level{p.x->data<float>(), p.y->mutable_data<float>(), nullptr, p.N, p.H, p.W,
                                  cache_.Packed(*p.w, kind), p.b ? p.b->data<float>() : nullptr};
        }
        // per-problem filter / bias; the launch-wide bias only says whether one is added at all
        const float* any_bias = lv[0].bias;
        const int rc = kind == FilterPackCache::SPLIT_FWD
                           ? split_.Run(lv, n, lv[0].packed, any_bias, p0.M, p0.C, flags, s)
                       : kind == FilterPackCache::WINO24_FWD
                           ? ssad_conv3x3_forward_wino24(lv, n, lv[0].packed, any_bias, p0.M, p0.C, flags, s)
                       : kind == FilterPackCache::WINO_FWD
                           ? ssad_conv3x3_forward_wino(lv, n, lv[0].packed, any_bias, p0.M, p0.C, flags, s)
                           : ssad_conv3x3_forward(lv, n, lv[0].packed, any_bias, p0.M, p0.C, flags, s);
        CAFFE_ENFORCE_EQ(rc, 0, "ConvGroup launch failed");
        ++g_conv_launch_calls;
      }
    }
    return true;
  }

 private:
  ConvGeometry geom_;
  int fuse_relu_;
  int fuse_sigmoid_;
  string algo_;
  int per_ = 3;
  FilterPackCache cache_;
  SplitEngine split_;
};

class ConvGradientGroupOp final : public Operator<HIPContext> {
 public:
  ConvGradientGroupOp(const OperatorDef& def, Workspace* ws)
      : Operator<HIPContext>(def, ws),
        geom_(ParseConvGeometry(*this)),
        no_bias_(GetSingleArgument<int>("no_bias", 0) != 0),
        relu_grad_on_input_(GetSingleArgument<int>("relu_grad_on_input", 0)),
        algo_(GetSingleArgument<string>("hip_algo", "auto")),
        filter_index_(GetRepeatedArgument<int>("filter_index")),
        n_filters_(GetSingleArgument<int>("n_filters", 0)) {
    CAFFE_ENFORCE(IsSubnetGeometry(geom_),
                  "ConvGradientGroup implements kernel 3 / stride 1 / pad 1 / group 1 / NCHW");
    CAFFE_ENFORCE(InputSize() % 3 == 0 && InputSize() >= 3, "ConvGradientGroup: [X, W, dY] per problem");
    k_ = InputSize() / 3;
    CAFFE_ENFORCE_EQ((int)filter_index_.size(), k_);
    CAFFE_ENFORCE(n_filters_ >= 1 && n_filters_ <= k_);
    const int fixed = n_filters_ * (no_bias_ ? 1 : 2);
    CAFFE_ENFORCE(OutputSize() == fixed || OutputSize() == fixed + k_,
                  "ConvGradientGroup: outputs are [dW_f ...] [db_f ...] and optionally one dX per problem");
    want_dx_ = OutputSize() == fixed + k_;
    for (int f : filter_index_) CAFFE_ENFORCE(f >= 0 && f < n_filters_);
  }

  bool RunOnDevice() override {
    vector<Problem> probs(k_);
    const int fixed = n_filters_ * (no_bias_ ? 1 : 2);
    for (int i = 0; i < k_; ++i) {
      Problem& p = probs[i];
      p.x = &Input(3 * i);
      p.w = &Input(3 * i + 1);
      p.b = &Input(3 * i + 2);                       // dY
      Dims(&p);
      CheckSubnetProblem(p);
      CAFFE_ENFORCE(p.b->ndim() == 4 && p.b->dim32(0) == p.N && p.b->dim32(1) == p.M && p.b->dim32(2) == p.H &&
                        p.b->dim32(3) == p.W,
                    "output gradient shape does not match the convolution output");
      if (want_dx_) {
        p.y = Output(fixed + i);
        p.y->ResizeLike(*p.x);
      }
    }
    hipStream_t s = context_.hip_stream();

    // filter (+ bias) gradients: one launch per filter over all of its problems -- the reduction over
    // levels IS the autograd Sum of core.py:706-741; overwrite, beta = 0 (conv_op_cudnn.cc:1037)
    for (int f = 0; f < n_filters_; ++f) {
      vector<int> mine;
      for (int i = 0; i < k_; ++i)
        if (filter_index_[i] == f) mine.push_back(i);
      CAFFE_ENFORCE(!mine.empty());
      const Problem& p0 = probs[mine[0]];
      for (int i : mine)
        CAFFE_ENFORCE(probs[i].w->raw_data() == p0.w->raw_data() && probs[i].M == p0.M && probs[i].C == p0.C,
                      "problems of one filter_index must read one filter blob");
      auto* dW = Output(f);
      dW->ResizeLike(*p0.w);
      float* db = nullptr;
      if (!no_bias_) {
        auto* dbt = Output(n_filters_ + f);
        dbt->Resize(p0.M);
        db = dbt->mutable_data<float>();
      }
      for (size_t at = 0; at < mine.size(); at += SSAD_MAX_LEVELS) {
        const int n = (int)std::min<size_t>(SSAD_MAX_LEVELS, mine.size() - at);
        ssad_conv_level lv[SSAD_MAX_LEVELS];
        for (int j = 0; j < n; ++j) {
          const Problem& p = probs[mine[at + j]];
          lv[j] = ssad_conv_level{p.x->data<float>(), nullptr, p.b->data<float>(), p.N, p.H, p.W, nullptr, nullptr};
        }
        // hip_algo = "split": the >= 128-wide filter gradients on the split-operand engine too (conv3x3_wgrad_split.hip;
        // same contract, same tolerance), like the native step's SSAD_SPLIT_CONV bit 32
        const bool wsplit = algo_ == "split" && p0.M >= 128 && p0.C >= 64;
        const size_t wsb = wsplit ? ssad_conv3x3_wgrad_split_workspace_bytes(lv, n, p0.M, p0.C)
                                  : ssad_conv3x3_wgrad_workspace_bytes(lv, n, p0.M, p0.C);
        if ((size_t)workspace_.size() < wsb) workspace_.Resize((TIndex)wsb);
        const int rc = (wsplit ? ssad_conv3x3_wgrad_split : ssad_conv3x3_wgrad)(
            lv, n, dW->mutable_data<float>(), db, p0.M, p0.C, at > 0 ? 1 : 0, workspace_.mutable_data<uint8_t>(),
            (size_t)workspace_.size(), s);
        CAFFE_ENFORCE_EQ(rc, 0, "ConvGradientGroup (filter) launch failed");
        ++g_conv_launch_calls;
      }
    }
    if (!want_dx_) return true;

    // data gradients: the forward kernel on the flipped / transposed pack, dX has C channels
    const long long before = cache_.packs_issued();
    auto dkind_of = [&](const Problem& p) {
      if (algo_ == "split" && p.C >= 256) return FilterPackCache::SPLIT_DGRAD;
      if (algo_ == "split" && p.C >= 128) return FilterPackCache::WINO24_DGRAD;
      if (algo_ == "winograd24" && p.C >= 128) return FilterPackCache::WINO24_DGRAD;
      return UseWinograd(algo_, p.C) ? FilterPackCache::WINO_DGRAD : FilterPackCache::DIRECT_DGRAD;
    };
    for (const Problem& p : probs) cache_.Want(*p.w, dkind_of(p));
    cache_.Flush(s);
    g_filter_packs_issued += cache_.packs_issued() - before;
    const int flags = relu_grad_on_input_ ? SSAD_CONV_MASK_AUX : 0;
    for (const vector<int>& cls : Classes(probs)) {
      const Problem& p0 = probs[cls[0]];
      const auto kind = dkind_of(p0);
      for (size_t at = 0; at < cls.size(); at += SSAD_MAX_CONV_PROBLEMS) {
        const int n = (int)std::min<size_t>(SSAD_MAX_CONV_PROBLEMS, cls.size() - at);
        ssad_conv_level lv[SSAD_MAX_CONV_PROBLEMS];
        for (int j = 0; j < n; ++j) {
          const Problem& p = probs[cls[at + j]];
          lv[j] = ssad_conv_level{p.b->data<float>(), p.y->mutable_data<float>(),
                                  relu_grad_on_input_ ? p.x->data<float>() : nullptr, p.N, p.H, p.W,
                                  cache_.Packed(*p.w, kind), nullptr};
        }
        const int rc = kind == FilterPackCache::SPLIT_DGRAD
                           ? split_.Run(lv, n, lv[0].packed, nullptr, p0.C, p0.M, flags, s)
                       : kind == FilterPackCache::WINO24_DGRAD
                           ? ssad_conv3x3_forward_wino24(lv, n, lv[0].packed, nullptr, p0.C, p0.M, flags, s)
                       : kind == FilterPackCache::WINO_DGRAD
                           ? ssad_conv3x3_forward_wino(lv, n, lv[0].packed, nullptr, p0.C, p0.M, flags, s)
                           : ssad_conv3x3_forward(lv, n, lv[0].packed, nullptr, p0.C, p0.M, flags, s);
        CAFFE_ENFORCE_EQ(rc, 0, "ConvGradientGroup (data) launch failed");
        ++g_conv_launch_calls;
      }
    }
    return true;
  }

 private:
  ConvGeometry geom_;
  bool no_bias_;
  int relu_grad_on_input_;
  string algo_;
  vector<int> filter_index_;
  int n_filters_;
  int k_ = 0;
  bool want_dx_ = false;
  Tensor<HIPContext> workspace_;
  FilterPackCache cache_;
  SplitEngine split_;
};

REGISTER_HIP_OPERATOR(ConvGroup, ConvGroupOp);
REGISTER_HIP_OPERATOR(ConvGradientGroup, ConvGradientGroupOp);

OPERATOR_SCHEMA(ConvGroup)
    .NumInputs(2, INT_MAX)
    .NumOutputs(1, INT_MAX)
    .SetDoc("Convolutions of equal arguments that are ready together, one multi-problem launch per "
            "(Cout, Cin) class; emitted by the net lowering, [X, W, (b)] per output.");
OPERATOR_SCHEMA(ConvGradientGroup)
    .NumInputs(3, INT_MAX)
    .NumOutputs(1, INT_MAX)
    .Arg("filter_index", "per problem: which of the n_filters distinct filters it reads")
    .Arg("n_filters", "number of distinct filters = number of dW (and db) outputs")
    .SetDoc("ConvGradients that are ready together; the filter gradient of a filter is reduced over all of "
            "its problems in one launch (the Sum of core.py:706-741 absorbed).");
NO_GRADIENT(ConvGroup);
NO_GRADIENT(ConvGradientGroup);

}  // namespace caffe2
