// net_lowering.cc -- LowerNet: the operator list of a NetDef rewritten for the MI355X kernels
// before the net instantiates it (c2/net.h).  Pure host logic over definitions.
//
// The reference gets the equivalent effects from three places: cuDNN's fused activation
// (conv_op_cudnn.cc), the DAG executor running independent chains side by side
// (net_dag.cc:181-270), and nothing at all for the per-level launches of a shared filter
// (detector.py:449-482 ConvShared: five cuDNN calls per layer).  Here they are one explicit
// transform, in the style of caffe2/core/transform.h passes:
//
//   F1  Conv -> Relu (in place, only reader)                =>  Conv(fuse_relu = 1)
//   F1s Conv -> Sigmoid (only reader, logits not kept)      =>  Conv(fuse_sigmoid = 1)
//   F2  ConvGradient -> ReluGradient on its dX, masked by the convolution's own input
//                                                            =>  ConvGradient(relu_grad_on_input = 1)
//   F3  Sum(pieces) where every piece is the filter (bias) gradient of a ConvGradient on ONE
//       filter -- the `_grad_autosplit_k` accumulation of caffe2/python/core.py:706-741 --
//       is absorbed: the ConvGradients become one tied unit whose filter-gradient launch sums
//       over all of them and writes the Sum's output
//   G   convolutions (3x3 / stride 1 / pad 1, fp32) of equal arguments that are READY TOGETHER
//       become one ConvGroup / ConvGradientGroup operator: one multi-problem launch per
//       (Cout, Cin) class -- the five FPN levels of a shared filter, the cls and bbox tower
//       layer of equal depth
//
// Blobs are tracked as VALUES (name, write generation); every rewrite happens on the value
// graph, the schedule honours read-after-write, write-after-read and write-after-write on the
// names, and the emitted list is replayed against the value graph before it is accepted --
// a list that fails the replay is dropped for the list as written (LoweringReport::fell_back).
#include <algorithm>
#include <map>
#include <set>

#include "c2/net.h"
#include "ops/conv_op.h"

namespace caffe2 {
namespace {

using Val = std::pair<string, int>;   // (blob name, write generation; 0 = as it enters the net)

struct Node {
  OperatorDef def;
  vector<Val> in, out;
  int orig = 0;
  bool removed = false;
  int tie = -1;
};

struct Tie {                 // ConvGradients of one filter whose gradient Sums are absorbed
  vector<int> members;       // node ids, original order
  Val dw, db;                // what the absorbed Sums wrote
  bool has_db = false;
};

const Argument* FindArg(const OperatorDef& d, const string& n) {
  for (const Argument& a : d.arg)
    if (a.name == n) return &a;
  return nullptr;
}

string ArgKey(const OperatorDef& d) {
  vector<string> parts;
  for (const Argument& a : d.arg) {
    std::ostringstream ss;
    ss << a.name << "=";
    if (a.has_f) ss << "f" << a.f;
    if (a.has_i) ss << "i" << a.i;
    if (a.has_s) ss << "s" << a.s;
    for (float f : a.floats) ss << ",f" << f;
    for (int64_t i : a.ints) ss << ",i" << i;
    for (const string& s : a.strings) ss << ",s" << s;
    parts.push_back(ss.str());
  }
  std::sort(parts.begin(), parts.end());
  std::ostringstream key;
  key << d.type << "|" << d.engine << "|" << (d.has_device_option ? d.device_option.device_type : -1) << ":"
      << d.device_option.gpu_id << "|" << d.input.size() << ">" << d.output.size();
  for (const string& p : parts) key << "|" << p;
  return key.str();
}

bool OnGpu(const OperatorDef& d) { return d.has_device_option && IsGPUDeviceType(d.device_option.device_type); }

struct Lowering {
  const LoweringOptions& opt;
  LoweringReport& rep;
  vector<Node> nodes;
  vector<Tie> ties;
  std::map<Val, int> producer;
  std::map<Val, vector<int>> readers;
  std::map<string, int> last_ver;

  Lowering(const LoweringOptions& o, LoweringReport& r) : opt(o), rep(r) {}

  void Build(const vector<OperatorDef>& ops) {
    nodes.resize(ops.size());
    for (size_t i = 0; i < ops.size(); ++i) {
      Node& n = nodes[i];
      n.def = ops[i];
      n.orig = (int)i;
      for (const string& s : n.def.input) n.in.push_back(Val(s, last_ver[s]));
      for (const string& s : n.def.output) n.out.push_back(Val(s, ++last_ver[s]));
    }
    Reindex();
  }

  void Reindex() {
    producer.clear();
    readers.clear();
    for (size_t i = 0; i < nodes.size(); ++i) {
      if (nodes[i].removed) continue;
      for (const Val& v : nodes[i].out) producer[v] = (int)i;
      for (const Val& v : nodes[i].in) readers[v].push_back((int)i);
    }
  }

  bool OnlyReader(const Val& v, int node) const {
    auto it = readers.find(v);
    if (it == readers.end()) return false;
    for (int r : it->second)
      if (r != node) return false;
    return !it->second.empty();
  }

  // a 3x3 / stride 1 / pad 1 / group 1 NCHW convolution on the GPU whose filter is an fp32 blob
  bool IsFusedPathConv(const Node& n) const {
    if (n.def.type != "Conv" && n.def.type != "ConvGradient") return false;
    if (!OnGpu(n.def) || n.in.size() < 2) return false;
    ConvGeometry g;
    try {
      g = ParseConvGeometry(n.def);
    } catch (const EnforceNotMet&) {
      return false;            // the operator's constructor will report it
    }
    if (!IsSubnetGeometry(g)) return false;
    if (opt.blob_dtype && opt.blob_dtype(n.in[1].first) != (int)DataType::FLOAT) return false;
    return true;
  }

  // ---- F1 / F2 ------------------------------------------------------------------------------
  void FuseRelu() {
    for (size_t j = 0; j < nodes.size(); ++j) {
      Node& r = nodes[j];
      if (r.removed || r.def.type != "Relu" || r.in.size() != 1 || r.out.size() != 1) continue;
      if (r.in[0].first != r.out[0].first) continue;                   // in place only
      auto pit = producer.find(r.in[0]);
      if (pit == producer.end()) continue;
      Node& c = nodes[pit->second];
      if (c.removed || c.def.type != "Conv" || !IsFusedPathConv(c)) continue;
      if (FindArg(c.def, "fuse_relu") || !OnlyReader(r.in[0], (int)j)) continue;
      if (!(c.def.device_option.gpu_id == r.def.device_option.gpu_id && OnGpu(r.def))) continue;
      c.def.arg.push_back(MakeArgument("fuse_relu", 1));
      c.out[0] = r.out[0];                     // the convolution now writes the activated value
      r.removed = true;
      ++rep.relu_fused;
      Reindex();
    }
    // F1s: Conv -> Sigmoid (retinanet_heads.py:153-163: the test-mode graph turns cls_pred into probabilities), the
    // logits read by nothing else and not asked for: Conv(fuse_sigmoid = 1) writes the probabilities, the logits
    // blob is not produced (sigmoid_op.cu:25-29 in the convolution's epilogue).  Not combined with fuse_relu.
    for (size_t j = 0; j < nodes.size(); ++j) {
      Node& r = nodes[j];
      if (r.removed || r.def.type != "Sigmoid" || r.in.size() != 1 || r.out.size() != 1) continue;
      auto pit = producer.find(r.in[0]);
      if (pit == producer.end()) continue;
      Node& c = nodes[pit->second];
      if (c.removed || c.def.type != "Conv" || !IsFusedPathConv(c)) continue;
      if (FindArg(c.def, "fuse_relu") || FindArg(c.def, "fuse_sigmoid") || !OnlyReader(r.in[0], (int)j)) continue;
      if (r.in[0].first != r.out[0].first && opt.keep.count(r.in[0].first)) continue;
      if (!(c.def.device_option.gpu_id == r.def.device_option.gpu_id && OnGpu(r.def))) continue;
      c.def.arg.push_back(MakeArgument("fuse_sigmoid", 1));
      c.out[0] = r.out[0];
      r.removed = true;
      ++rep.sigmoid_fused;
      Reindex();
    }
    for (size_t j = 0; j < nodes.size(); ++j) {
      Node& r = nodes[j];
      // ReluGradient [Y, dY] -> [dX]  (relu_op.cu:44-53: dX = Y > 0 ? dY : 0)
      if (r.removed || r.def.type != "ReluGradient" || r.in.size() != 2 || r.out.size() != 1) continue;
      auto pit = producer.find(r.in[1]);
      if (pit == producer.end()) continue;
      Node& c = nodes[pit->second];
      if (c.removed || c.def.type != "ConvGradient" || !IsFusedPathConv(c)) continue;
      if (FindArg(c.def, "relu_grad_on_input")) continue;
      const bool no_bias = FindArg(c.def, "no_bias") && FindArg(c.def, "no_bias")->i != 0;
      const size_t dx_slot = no_bias ? 1 : 2;
      if (c.out.size() != dx_slot + 1 || c.out[dx_slot] != r.in[1]) continue;   // dY must be the conv's dX
      if (c.in[0] != r.in[0]) continue;                                          // masked by the conv's own input
      if (!OnlyReader(r.in[1], (int)j) || opt.keep.count(r.in[1].first)) continue;
      c.def.arg.push_back(MakeArgument("relu_grad_on_input", 1));
      c.out[dx_slot] = r.out[0];
      r.removed = true;
      ++rep.relu_grad_fused;
      Reindex();
    }
  }

  // ---- F24: 3x3 convolutions on the F(2x4, 3x3) engine (conv3x3_winograd24.hip: 3 multiplies per output where
  //      F(2x2) does 4; fp32 error 1.5-2.2e-6 of the output scale against 0.8-1.9e-6).  A net without gradient
  //      operators is only evaluated (the teacher net, model_builder.py:373-411 in test mode; an inference net):
  //      option frozen_f24.  A trained net: option train_f24 -- Conv (>= 128 outputs) and the data gradient of
  //      ConvGradient (>= 128 input channels); the filter gradient keeps its own engine.  The operators apply the
  //      channel limits; hip_algo given by the graph wins.
  void MarkF24() {
    bool trained = false;
    for (const Node& n : nodes) {
      const string& t = n.def.type;
      // (StopGradient ends in "Gradient" and appears in evaluated-only graphs: it is not a gradient operator)
      if (n.def.is_gradient_op ||
          (t != "StopGradient" && t.size() > 8 && t.compare(t.size() - 8, 8, "Gradient") == 0) ||
          t == "MomentumSGDUpdate" || t == "WeightedSum")
        trained = true;
    }
    if (trained ? !opt.train_f24 : !opt.frozen_f24) return;
    for (Node& n : nodes) {
      if (n.removed || FindArg(n.def, "hip_algo") || !IsFusedPathConv(n)) continue;
      if (n.def.type == "ConvGradient" && !trained) continue;
      n.def.arg.push_back(MakeArgument("hip_algo", string(opt.split ? "split" : "winograd24")));
      ++(trained ? rep.train_f24 : rep.frozen_f24);
      rep.split += opt.split ? 1 : 0;
    }
  }

  // ---- F3 -----------------------------------------------------------------------------------
  // A Sum whose inputs are exactly the slot-`slot` outputs of ConvGradients that share one filter value.
  bool SumOfFilterGradients(const Node& s, size_t slot, vector<int>* members) const {
    if (s.removed || s.def.type != "Sum" || s.in.size() < 2 || s.out.size() != 1) return false;
    members->clear();
    Val filter;
    string key;
    for (const Val& piece : s.in) {
      auto pit = producer.find(piece);
      if (pit == producer.end()) return false;
      const Node& c = nodes[pit->second];
      if (c.removed || c.def.type != "ConvGradient" || c.tie >= 0 || !IsFusedPathConv(c)) return false;
      if (c.out.size() <= slot || c.out[slot] != piece) return false;
      if (!OnlyReader(piece, s.orig) || opt.keep.count(piece.first)) return false;
      if (members->empty()) {
        filter = c.in[1];
        key = ArgKey(c.def);
      } else if (c.in[1] != filter || ArgKey(c.def) != key) {
        return false;
      }
      if (std::find(members->begin(), members->end(), pit->second) != members->end()) return false;
      members->push_back(pit->second);
    }
    return true;
  }

  void AbsorbSums() {
    for (size_t i = 0; i < nodes.size(); ++i) {
      vector<int> mw;
      if (!SumOfFilterGradients(nodes[i], 0, &mw)) continue;
      const Node& first = nodes[mw[0]];
      const bool no_bias = FindArg(first.def, "no_bias") && FindArg(first.def, "no_bias")->i != 0;
      int sum_b = -1;
      if (!no_bias) {
        // the bias pieces of the same ConvGradients must meet in one Sum as well
        auto rit = readers.find(first.out[1]);
        if (rit == readers.end() || rit->second.size() != 1) continue;
        vector<int> mb;
        if (!SumOfFilterGradients(nodes[rit->second[0]], 1, &mb)) continue;
        vector<int> a = mw, b = mb;
        std::sort(a.begin(), a.end());
        std::sort(b.begin(), b.end());
        if (a != b) continue;
        sum_b = rit->second[0];
      }
      Tie t;
      t.members = mw;
      std::sort(t.members.begin(), t.members.end());
      t.dw = nodes[i].out[0];
      if (sum_b >= 0) {
        t.db = nodes[sum_b].out[0];
        t.has_db = true;
        nodes[sum_b].removed = true;
        ++rep.sums_absorbed;
      }
      nodes[i].removed = true;
      ++rep.sums_absorbed;
      for (int m : t.members) nodes[m].tie = (int)ties.size();
      ties.push_back(t);
      Reindex();
    }
  }

  // ---- scheduling units ---------------------------------------------------------------------
  struct Unit {
    vector<int> members;       // node ids
    std::set<Val> reads, writes;
    string key;                // non-empty: may share a launch with ready units of the same key
    int orig = 0;
    int tie = -1;
  };
  vector<Unit> units;

  void MakeUnits() {
    std::map<int, int> unit_of_tie;
    for (size_t i = 0; i < nodes.size(); ++i) {
      const Node& n = nodes[i];
      if (n.removed) continue;
      if (n.tie >= 0 && unit_of_tie.count(n.tie)) continue;
      Unit u;
      u.orig = n.orig;
      if (n.tie >= 0) {
        const Tie& t = ties[n.tie];
        unit_of_tie[n.tie] = (int)units.size();
        u.members = t.members;
        u.tie = n.tie;
        const bool no_bias = !t.has_db;
        for (int m : t.members) {
          for (const Val& v : nodes[m].in) u.reads.insert(v);
          const size_t dx_slot = no_bias ? 1 : 2;
          if (nodes[m].out.size() > dx_slot) u.writes.insert(nodes[m].out[dx_slot]);
        }
        u.writes.insert(t.dw);
        if (t.has_db) u.writes.insert(t.db);
      } else {
        u.members.push_back((int)i);
        for (const Val& v : n.in) u.reads.insert(v);
        for (const Val& v : n.out) u.writes.insert(v);
      }
      if (opt.group_convs && IsFusedPathConv(n)) u.key = ArgKey(n.def);
      units.push_back(u);
    }
  }

  bool Schedule(vector<vector<int>>* order) {
    const int U = (int)units.size();
    std::map<Val, int> prod;
    std::map<Val, vector<int>> rd;
    std::map<string, std::set<int>> versions;
    for (int u = 0; u < U; ++u) {
      for (const Val& v : units[u].writes) { prod[v] = u; versions[v.first].insert(v.second); }
      for (const Val& v : units[u].reads) { rd[v].push_back(u); versions[v.first].insert(v.second); }
    }
    vector<std::set<int>> deps(U);
    for (int u = 0; u < U; ++u) {
      for (const Val& v : units[u].reads) {
        auto it = prod.find(v);
        if (it != prod.end() && it->second != u) deps[u].insert(it->second);
      }
      for (const Val& w : units[u].writes)
        for (int older : versions[w.first]) {
          if (older >= w.second) break;
          const Val o(w.first, older);
          auto it = prod.find(o);
          if (it != prod.end() && it->second != u) deps[u].insert(it->second);     // write after write
          auto rt = rd.find(o);
          if (rt != rd.end())
            for (int r : rt->second)
              if (r != u) deps[u].insert(r);                                        // write after read
        }
    }
    vector<int> pending(U);
    vector<vector<int>> users(U);
    for (int u = 0; u < U; ++u) {
      pending[u] = (int)deps[u].size();
      for (int d : deps[u]) users[d].push_back(u);
    }
    std::set<std::pair<int, int>> ready;      // (original position, unit)
    for (int u = 0; u < U; ++u)
      if (!pending[u]) ready.insert({units[u].orig, u});
    int done = 0;
    auto retire = [&](int u) {
      ++done;
      for (int w : users[u])
        if (--pending[w] == 0) ready.insert({units[w].orig, w});
    };
    while (done < U) {
      if (ready.empty()) return false;                      // a cycle: cannot happen on a valid list
      int pick = -1;
      for (const auto& r : ready)
        if (units[r.second].key.empty()) { pick = r.second; break; }
      if (pick >= 0) {
        ready.erase({units[pick].orig, pick});
        order->push_back({pick});
        retire(pick);
        continue;
      }
      // only groupable units are ready: the first one and everything ready that can share its launch
      const string key = units[ready.begin()->second].key;
      vector<int> group;
      for (const auto& r : ready)
        if (units[r.second].key == key) group.push_back(r.second);
      for (int u : group) ready.erase({units[u].orig, u});
      order->push_back(group);
      for (int u : group) retire(u);
    }
    return true;
  }

  // ---- emission -----------------------------------------------------------------------------
  static void SetNames(OperatorDef* d, const vector<Val>& in, const vector<Val>& out) {
    d->input.clear();
    d->output.clear();
    for (const Val& v : in) d->input.push_back(v.first);
    for (const Val& v : out) d->output.push_back(v.first);
  }

  struct Emitted {
    OperatorDef def;
    vector<Val> in, out;
  };

  Emitted EmitSingle(int node) const {
    Emitted e;
    e.def = nodes[node].def;
    e.in = nodes[node].in;
    e.out = nodes[node].out;
    SetNames(&e.def, e.in, e.out);
    return e;
  }

  Emitted EmitConvGroup(const vector<int>& members) {
    Emitted e;
    e.def = nodes[members[0]].def;
    e.def.type = "ConvGroup";
    for (int m : members) {
      for (const Val& v : nodes[m].in) e.in.push_back(v);
      e.out.push_back(nodes[m].out[0]);
    }
    SetNames(&e.def, e.in, e.out);
    ++rep.conv_groups;
    rep.conv_group_members += (int)members.size();
    return e;
  }

  Emitted EmitConvGradientGroup(const vector<int>& group_units) {
    Emitted e;
    const Node& first = nodes[units[group_units[0]].members[0]];
    e.def = first.def;
    e.def.type = "ConvGradientGroup";
    const bool no_bias = FindArg(first.def, "no_bias") && FindArg(first.def, "no_bias")->i != 0;
    const size_t dx_slot = no_bias ? 1 : 2;
    vector<Val> dw, db, dx;
    Argument fidx;
    fidx.name = "filter_index";
    int n_members = 0;
    for (int u : group_units) {
      const Unit& unit = units[u];
      const int f = (int)dw.size();
      if (unit.tie >= 0) {
        dw.push_back(ties[unit.tie].dw);
        if (!no_bias) db.push_back(ties[unit.tie].db);
      } else {
        dw.push_back(nodes[unit.members[0]].out[0]);
        if (!no_bias) db.push_back(nodes[unit.members[0]].out[1]);
      }
      for (int m : unit.members) {
        for (const Val& v : nodes[m].in) e.in.push_back(v);
        if (nodes[m].out.size() > dx_slot) dx.push_back(nodes[m].out[dx_slot]);
        fidx.ints.push_back(f);
        ++n_members;
      }
    }
    e.out = dw;
    e.out.insert(e.out.end(), db.begin(), db.end());
    e.out.insert(e.out.end(), dx.begin(), dx.end());
    e.def.arg.push_back(fidx);
    e.def.arg.push_back(MakeArgument("n_filters", (int)dw.size()));
    SetNames(&e.def, e.in, e.out);
    ++rep.conv_grad_groups;
    rep.conv_grad_group_members += n_members;
    return e;
  }

  bool Emit(const vector<vector<int>>& order, vector<OperatorDef>* out) {
    vector<Emitted> list;
    for (const vector<int>& step : order) {
      const Unit& u0 = units[step[0]];
      const Node& n0 = nodes[u0.members[0]];
      const bool single = step.size() == 1 && u0.members.size() == 1 && u0.tie < 0;
      if (single) {
        list.push_back(EmitSingle(u0.members[0]));
      } else if (n0.def.type == "Conv") {
        vector<int> members;
        for (int u : step) members.push_back(units[u].members[0]);
        list.push_back(EmitConvGroup(members));
      } else {
        list.push_back(EmitConvGradientGroup(step));
      }
    }
    // replay: every read must find the value it was written against
    std::map<string, int> cur;
    for (const Emitted& e : list) {
      for (const Val& v : e.in) {
        auto it = cur.find(v.first);
        if ((it == cur.end() ? 0 : it->second) != v.second) return false;
      }
      for (const Val& v : e.out) cur[v.first] = v.second;
    }
    // and every name must end at the generation the list as written leaves it at (or at an
    // absorbed piece that nothing reads)
    for (const auto& kv : cur) {
      auto it = last_ver.find(kv.first);
      if (it == last_ver.end() || it->second != kv.second) return false;
    }
    for (const Emitted& e : list) out->push_back(e.def);
    return true;
  }
};

}  // namespace

vector<OperatorDef> LowerNet(const NetDef& def, const LoweringOptions& opt, LoweringReport* report) {
  LoweringReport local;
  LoweringReport& rep = report ? *report : local;
  rep = LoweringReport();
  rep.ops_in = (int)def.op.size();
  Lowering L(opt, rep);
  L.Build(def.op);
  if (opt.fuse_relu) L.FuseRelu();
  if (opt.frozen_f24 || opt.train_f24) L.MarkF24();
  if (opt.group_convs) L.AbsorbSums();
  L.MakeUnits();
  vector<vector<int>> order;
  vector<OperatorDef> out;
  if (!L.Schedule(&order) || !L.Emit(order, &out)) {
    rep = LoweringReport();
    rep.ops_in = rep.ops_out = (int)def.op.size();
    rep.fell_back = true;
    return def.op;
  }
  rep.ops_out = (int)out.size();
  return out;
}

}  // namespace caffe2
