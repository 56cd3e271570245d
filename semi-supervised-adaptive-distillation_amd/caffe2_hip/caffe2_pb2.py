"""OperatorDef / Argument / DeviceOption / NetDef with the protobuf wire format of
caffe2/proto/caffe2.proto:97-215, hand-coded (no protoc in this image).
SerializeToString() / ParseFromString() interoperate with the C++ codec in
csrc/c2/proto.cc and with real caffe2_pb2 messages."""
import struct

CPU, CUDA, MKLDNN, OPENGL, HIP = 0, 1, 2, 3, 6


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _ld(field, payload):
    return _key(field, 2) + _varint(len(payload)) + payload


def _b(s):
    return s if isinstance(s, bytes) else str(s).encode("utf-8")


class _Reader(object):
    def __init__(self, data):
        self.d, self.p = memoryview(data), 0

    def done(self):
        return self.p >= len(self.d)

    def varint(self):
        v, shift = 0, 0
        while True:
            b = self.d[self.p]
            self.p += 1
            v |= (b & 0x7F) << shift
            if not b & 0x80:
                return v
            shift += 7

    def signed(self):
        v = self.varint()
        return v - (1 << 64) if v >= (1 << 63) else v

    def fixed32(self):
        v = struct.unpack_from("<f", self.d, self.p)[0]
        self.p += 4
        return v

    def bytes_(self):
        n = self.varint()
        b = bytes(self.d[self.p:self.p + n])
        self.p += n
        return b

    def skip(self, wire):
        if wire == 0:
            self.varint()
        elif wire == 1:
            self.p += 8
        elif wire == 2:
            self.bytes_()
        elif wire == 5:
            self.p += 4
        else:
            raise ValueError("bad wire type %d" % wire)


class Argument(object):
    def __init__(self, name=""):
        self.name = name
        self.f = self.i = self.s = None
        self.floats, self.ints, self.strings = [], [], []

    def HasField(self, n):
        return getattr(self, n) is not None

    def SerializeToString(self):
        out = _ld(1, _b(self.name)) if self.name else b""
        if self.f is not None:
            out += _key(2, 5) + struct.pack("<f", self.f)
        if self.i is not None:
            out += _key(3, 0) + _varint(int(self.i))
        if self.s is not None:
            out += _ld(4, _b(self.s))
        for v in self.floats:
            out += _key(5, 5) + struct.pack("<f", v)
        for v in self.ints:
            out += _key(6, 0) + _varint(int(v))
        for v in self.strings:
            out += _ld(7, _b(v))
        return out

    def ParseFromString(self, data):
        r = _Reader(data)
        while not r.done():
            k = r.varint()
            f, w = k >> 3, k & 7
            if (f, w) == (1, 2):
                self.name = r.bytes_().decode()
            elif (f, w) == (2, 5):
                self.f = r.fixed32()
            elif (f, w) == (3, 0):
                self.i = r.signed()
            elif (f, w) == (4, 2):
                self.s = r.bytes_()
            elif (f, w) == (5, 5):
                self.floats.append(r.fixed32())
            elif (f, w) == (6, 0):
                self.ints.append(r.signed())
            elif (f, w) == (7, 2):
                self.strings.append(r.bytes_())
            elif (f, w) == (5, 2):
                sub = _Reader(r.bytes_())
                while not sub.done():
                    self.floats.append(sub.fixed32())
            elif (f, w) == (6, 2):
                sub = _Reader(r.bytes_())
                while not sub.done():
                    self.ints.append(sub.signed())
            else:
                r.skip(w)
        return self

    def to_jsonable(self):
        d = {"name": self.name}
        for k in ("f", "i"):
            if getattr(self, k) is not None:
                d[k] = getattr(self, k)
        if self.s is not None:
            d["s"] = self.s.decode() if isinstance(self.s, bytes) else self.s
        for k in ("floats", "ints"):
            if getattr(self, k):
                d[k] = list(getattr(self, k))
        if self.strings:
            d["strings"] = [s.decode() if isinstance(s, bytes) else s for s in self.strings]
        return d


class DeviceOption(object):
    def __init__(self, device_type=CPU, cuda_gpu_id=0):
        self.device_type = device_type
        self.cuda_gpu_id = cuda_gpu_id   # doubles as the hip gpu id

    def SerializeToString(self):
        out = _key(1, 0) + _varint(self.device_type)
        if self.device_type in (CUDA, HIP):
            out += _key(2, 0) + _varint(self.cuda_gpu_id)
        return out

    def ParseFromString(self, data):
        r = _Reader(data)
        while not r.done():
            k = r.varint()
            f, w = k >> 3, k & 7
            if (f, w) == (1, 0):
                self.device_type = r.varint()
            elif f in (2, 6) and w == 0:
                self.cuda_gpu_id = r.varint()
            else:
                r.skip(w)
        return self

    def __eq__(self, o):
        return (isinstance(o, DeviceOption) and self.device_type == o.device_type
                and self.cuda_gpu_id == o.cuda_gpu_id)


class OperatorDef(object):
    def __init__(self):
        self.input, self.output = [], []
        self.name, self.type, self.engine = "", "", ""
        self.arg = []
        self.device_option = None
        self.control_input = []
        self.is_gradient_op = False

    def SerializeToString(self):
        out = b"".join(_ld(1, _b(s)) for s in self.input)
        out += b"".join(_ld(2, _b(s)) for s in self.output)
        if self.name:
            out += _ld(3, _b(self.name))
        if self.type:
            out += _ld(4, _b(self.type))
        out += b"".join(_ld(5, a.SerializeToString()) for a in self.arg)
        if self.device_option is not None:
            out += _ld(6, self.device_option.SerializeToString())
        if self.engine:
            out += _ld(7, _b(self.engine))
        out += b"".join(_ld(8, _b(s)) for s in self.control_input)
        if self.is_gradient_op:
            out += _key(9, 0) + _varint(1)
        return out

    def ParseFromString(self, data):
        r = _Reader(data)
        while not r.done():
            k = r.varint()
            f, w = k >> 3, k & 7
            if w == 2 and f == 1:
                self.input.append(r.bytes_().decode())
            elif w == 2 and f == 2:
                self.output.append(r.bytes_().decode())
            elif w == 2 and f == 3:
                self.name = r.bytes_().decode()
            elif w == 2 and f == 4:
                self.type = r.bytes_().decode()
            elif w == 2 and f == 5:
                self.arg.append(Argument().ParseFromString(r.bytes_()))
            elif w == 2 and f == 6:
                self.device_option = DeviceOption().ParseFromString(r.bytes_())
            elif w == 2 and f == 7:
                self.engine = r.bytes_().decode()
            elif w == 2 and f == 8:
                self.control_input.append(r.bytes_().decode())
            elif w == 0 and f == 9:
                self.is_gradient_op = bool(r.varint())
            else:
                r.skip(w)
        return self

    def to_jsonable(self):
        d = {"type": self.type, "input": list(self.input), "output": list(self.output),
             "arg": sorted((a.to_jsonable() for a in self.arg), key=lambda a: a["name"])}
        if self.engine:
            d["engine"] = self.engine
        if self.name:
            d["name"] = self.name
        return d


class NetDef(object):
    """caffe2.proto:176-215."""

    def __init__(self, name=""):
        self.name = name
        self.op = []
        self.type = ""
        self.num_workers = 0
        self.device_option = None
        self.arg = []
        self.external_input = []
        self.external_output = []

    def SerializeToString(self):
        out = _ld(1, _b(self.name)) if self.name else b""
        out += b"".join(_ld(2, o.SerializeToString()) for o in self.op)
        if self.type:
            out += _ld(3, _b(self.type))
        if self.num_workers:
            out += _key(4, 0) + _varint(int(self.num_workers))
        if self.device_option is not None:
            out += _ld(5, self.device_option.SerializeToString())
        out += b"".join(_ld(6, a.SerializeToString()) for a in self.arg)
        out += b"".join(_ld(7, _b(s)) for s in self.external_input)
        out += b"".join(_ld(8, _b(s)) for s in self.external_output)
        return out

    def ParseFromString(self, data):
        r = _Reader(data)
        while not r.done():
            k = r.varint()
            f, w = k >> 3, k & 7
            if w == 2 and f == 1:
                self.name = r.bytes_().decode()
            elif w == 2 and f == 2:
                self.op.append(OperatorDef().ParseFromString(r.bytes_()))
            elif w == 2 and f == 3:
                self.type = r.bytes_().decode()
            elif w == 0 and f == 4:
                self.num_workers = r.varint()
            elif w == 2 and f == 5:
                self.device_option = DeviceOption().ParseFromString(r.bytes_())
            elif w == 2 and f == 6:
                self.arg.append(Argument().ParseFromString(r.bytes_()))
            elif w == 2 and f == 7:
                self.external_input.append(r.bytes_().decode())
            elif w == 2 and f == 8:
                self.external_output.append(r.bytes_().decode())
            else:
                r.skip(w)
        return self
