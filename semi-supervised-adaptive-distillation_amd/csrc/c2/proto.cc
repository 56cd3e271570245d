// proto.cc -- protobuf wire format (varint / 32-bit / length-delimited) for
// OperatorDef, Argument and DeviceOption; field numbers from
// caffe2/proto/caffe2.proto:97-172.
#include "c2/proto.h"

#include <cstring>

namespace caffe2 {
namespace {

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  bool done() const { return p >= end; }
  uint64_t varint() {
    uint64_t v = 0;
    int shift = 0;
    while (p < end && shift < 64) {
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
      shift += 7;
    }
    ok = false;
    return 0;
  }
  uint32_t fixed32() {
    if (end - p < 4) { ok = false; return 0; }
    uint32_t v; memcpy(&v, p, 4); p += 4; return v;
  }
  Reader sub() {
    const uint64_t n = varint();
    if (!ok || (uint64_t)(end - p) < n) { ok = false; return Reader{p, p}; }
    Reader r{p, p + n};
    p += n;
    return r;
  }
  string bytes() { Reader r = sub(); return string((const char*)r.p, r.end - r.p); }
  void skip(int wire) {
    switch (wire) {
      case 0: varint(); break;
      case 1: if (end - p < 8) ok = false; else p += 8; break;
      case 2: sub(); break;
      case 5: fixed32(); break;
      default: ok = false;
    }
  }
};

float as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

bool parse_argument(Reader r, Argument* a) {
  while (r.ok && !r.done()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wire = (int)(key & 7);
    if (field == 1 && wire == 2) a->name = r.bytes();
    else if (field == 2 && wire == 5) { a->f = as_float(r.fixed32()); a->has_f = true; }
    else if (field == 3 && wire == 0) { a->i = (int64_t)r.varint(); a->has_i = true; }
    else if (field == 4 && wire == 2) { a->s = r.bytes(); a->has_s = true; }
    else if (field == 5 && wire == 5) a->floats.push_back(as_float(r.fixed32()));
    else if (field == 5 && wire == 2) {            // packed floats
      Reader s = r.sub();
      while (s.ok && !s.done()) a->floats.push_back(as_float(s.fixed32()));
      r.ok = r.ok && s.ok;
    } else if (field == 6 && wire == 0) a->ints.push_back((int64_t)r.varint());
    else if (field == 6 && wire == 2) {            // packed ints
      Reader s = r.sub();
      while (s.ok && !s.done()) a->ints.push_back((int64_t)s.varint());
      r.ok = r.ok && s.ok;
    } else if (field == 7 && wire == 2) a->strings.push_back(r.bytes());
    else r.skip(wire);                             // nets (8, 9) and unknowns
  }
  return r.ok;
}

bool parse_device(Reader r, DeviceOption* d) {
  while (r.ok && !r.done()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wire = (int)(key & 7);
    if (field == 1 && wire == 0) d->device_type = (int)r.varint();
    else if ((field == 2 || field == 6) && wire == 0) d->gpu_id = (int)r.varint();
    else if (field == 3 && wire == 0) d->random_seed = (uint32_t)r.varint();
    else if (field == 4 && wire == 2) d->node_name = r.bytes();
    else r.skip(wire);
  }
  return r.ok;
}

struct Writer {
  string out;
  void varint(uint64_t v) {
    while (v >= 0x80) { out.push_back((char)((v & 0x7f) | 0x80)); v >>= 7; }
    out.push_back((char)v);
  }
  void key(int field, int wire) { varint(((uint64_t)field << 3) | wire); }
  void bytes(int field, const string& s) { key(field, 2); varint(s.size()); out += s; }
  void f32(int field, float f) { key(field, 5); uint32_t u; memcpy(&u, &f, 4); out.append((const char*)&u, 4); }
  void i64(int field, int64_t v) { key(field, 0); varint((uint64_t)v); }
};

string write_argument(const Argument& a) {
  Writer w;
  if (!a.name.empty()) w.bytes(1, a.name);
  if (a.has_f) w.f32(2, a.f);
  if (a.has_i) w.i64(3, a.i);
  if (a.has_s) w.bytes(4, a.s);
  for (float f : a.floats) w.f32(5, f);
  for (int64_t i : a.ints) w.i64(6, i);
  for (const string& s : a.strings) w.bytes(7, s);
  return w.out;
}

string write_device(const DeviceOption& d) {
  Writer w;
  w.i64(1, d.device_type);
  if (IsGPUDeviceType(d.device_type)) w.i64(2, d.gpu_id);
  if (d.random_seed) w.i64(3, d.random_seed);
  if (!d.node_name.empty()) w.bytes(4, d.node_name);
  return w.out;
}

}  // namespace

static bool parse_operator(Reader r, OperatorDef* out);

bool ParseOperatorDef(const void* data, size_t n, OperatorDef* out) {
  return parse_operator(Reader{(const uint8_t*)data, (const uint8_t*)data + n}, out);
}

bool ParseNetDef(const void* data, size_t n, NetDef* out) {
  *out = NetDef();
  Reader r{(const uint8_t*)data, (const uint8_t*)data + n};
  while (r.ok && !r.done()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wire = (int)(key & 7);
    if (wire == 2 && field == 1) out->name = r.bytes();
    else if (wire == 2 && field == 2) {
      OperatorDef op;
      if (!parse_operator(r.sub(), &op)) return false;
      out->op.push_back(std::move(op));
    } else if (wire == 2 && field == 3) out->type = r.bytes();
    else if (wire == 0 && field == 4) out->num_workers = (int)r.varint();
    else if (wire == 2 && field == 5) {
      if (!parse_device(r.sub(), &out->device_option)) return false;
      out->has_device_option = true;
    } else if (wire == 2 && field == 6) {
      Argument a;
      if (!parse_argument(r.sub(), &a)) return false;
      out->arg.push_back(a);
    } else if (wire == 2 && field == 7) out->external_input.push_back(r.bytes());
    else if (wire == 2 && field == 8) out->external_output.push_back(r.bytes());
    else r.skip(wire);
  }
  return r.ok;
}

string SerializeNetDef(const NetDef& def) {
  Writer w;
  if (!def.name.empty()) w.bytes(1, def.name);
  for (const OperatorDef& op : def.op) w.bytes(2, SerializeOperatorDef(op));
  if (!def.type.empty()) w.bytes(3, def.type);
  if (def.num_workers) w.i64(4, def.num_workers);
  if (def.has_device_option) w.bytes(5, write_device(def.device_option));
  for (const Argument& a : def.arg) w.bytes(6, write_argument(a));
  for (const string& s : def.external_input) w.bytes(7, s);
  for (const string& s : def.external_output) w.bytes(8, s);
  return w.out;
}

static bool parse_operator(Reader r, OperatorDef* out) {
  *out = OperatorDef();
  while (r.ok && !r.done()) {
    const uint64_t key = r.varint();
    const int field = (int)(key >> 3), wire = (int)(key & 7);
    if (wire == 2 && field == 1) out->input.push_back(r.bytes());
    else if (wire == 2 && field == 2) out->output.push_back(r.bytes());
    else if (wire == 2 && field == 3) out->name = r.bytes();
    else if (wire == 2 && field == 4) out->type = r.bytes();
    else if (wire == 2 && field == 5) {
      Argument a;
      if (!parse_argument(r.sub(), &a)) return false;
      out->arg.push_back(a);
    } else if (wire == 2 && field == 6) {
      if (!parse_device(r.sub(), &out->device_option)) return false;
      out->has_device_option = true;
    } else if (wire == 2 && field == 7) out->engine = r.bytes();
    else if (wire == 2 && field == 8) out->control_input.push_back(r.bytes());
    else if (wire == 0 && field == 9) out->is_gradient_op = r.varint() != 0;
    else r.skip(wire);
  }
  return r.ok;
}

string SerializeOperatorDef(const OperatorDef& def) {
  Writer w;
  for (const string& s : def.input) w.bytes(1, s);
  for (const string& s : def.output) w.bytes(2, s);
  if (!def.name.empty()) w.bytes(3, def.name);
  if (!def.type.empty()) w.bytes(4, def.type);
  for (const Argument& a : def.arg) w.bytes(5, write_argument(a));
  if (def.has_device_option) w.bytes(6, write_device(def.device_option));
  if (!def.engine.empty()) w.bytes(7, def.engine);
  for (const string& s : def.control_input) w.bytes(8, s);
  if (def.is_gradient_op) w.i64(9, 1);
  return w.out;
}

string ProtoDebugString(const OperatorDef& def) {
  std::ostringstream ss;
  for (const string& s : def.input) ss << "input: \"" << s << "\" ";
  for (const string& s : def.output) ss << "output: \"" << s << "\" ";
  if (!def.name.empty()) ss << "name: \"" << def.name << "\" ";
  ss << "type: \"" << def.type << "\" ";
  for (const Argument& a : def.arg) {
    ss << "arg { name: \"" << a.name << "\"";
    if (a.has_f) ss << " f: " << a.f;
    if (a.has_i) ss << " i: " << a.i;
    if (a.has_s) ss << " s: \"" << a.s << "\"";
    for (float f : a.floats) ss << " floats: " << f;
    for (int64_t i : a.ints) ss << " ints: " << i;
    for (const string& s : a.strings) ss << " strings: \"" << s << "\"";
    ss << " } ";
  }
  if (def.has_device_option)
    ss << "device_option { device_type: " << def.device_option.device_type
       << " gpu_id: " << def.device_option.gpu_id << " } ";
  if (!def.engine.empty()) ss << "engine: \"" << def.engine << "\" ";
  return ss.str();
}

}  // namespace caffe2
