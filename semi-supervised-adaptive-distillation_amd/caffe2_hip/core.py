"""caffe2.python.core, the slice the hot path uses (caffe2/python/core.py):
DeviceOption, DeviceScope/NameScope, CreateOperator, Net with dynamic
operator methods, GradientRegistry.GetGradientForOp and
Net.AddGradientOperators (incl. the `_grad_autosplit_k` + Sum accumulation of
core.py:706-741 for blobs with several consumers, e.g. the RetinaNet head
weights shared by five FPN levels)."""
import contextlib
import ctypes as C
import struct
import threading

import numpy as np

from . import _capi
from . import caffe2_pb2
from .caffe2_pb2 import Argument, OperatorDef

_tls = threading.local()
_REGISTERED_OPERATORS = set()


def DeviceOption(device_type, cuda_gpu_id=0):
    return caffe2_pb2.DeviceOption(device_type, cuda_gpu_id)


@contextlib.contextmanager
def DeviceScope(device_option):
    old = getattr(_tls, "device", None)
    _tls.device = device_option
    try:
        yield
    finally:
        _tls.device = old


def CurrentDeviceScope():
    return getattr(_tls, "device", None)


@contextlib.contextmanager
def NameScope(prefix):
    old = getattr(_tls, "namescope", "")
    _tls.namescope = old + prefix + ("/" if prefix and not prefix.endswith("/") else "")
    try:
        yield
    finally:
        _tls.namescope = old


def CurrentNameScope():
    return getattr(_tls, "namescope", "")


def ScopedName(name):
    return CurrentNameScope() + str(name)


def RefreshRegisteredOperators():
    L = _capi.load()
    _REGISTERED_OPERATORS.clear()
    for dev in (caffe2_pb2.CPU, caffe2_pb2.HIP):
        n = L.c2hip_registered_operators(dev, None, 0)
        buf = C.create_string_buffer(n)
        L.c2hip_registered_operators(dev, buf, n)
        for k in buf.value.decode().split("\n"):
            if k:
                _REGISTERED_OPERATORS.add(k.split("_ENGINE_")[0])


def IsOperator(op_type):
    if not _REGISTERED_OPERATORS:
        RefreshRegisteredOperators()
    return op_type in _REGISTERED_OPERATORS


def MakeArgument(key, value):
    """caffe2/python/utils.py MakeArgument."""
    a = Argument(key)
    if isinstance(value, np.ndarray):
        value = value.flatten().tolist()
    elif isinstance(value, np.generic):
        value = value.item()
    if isinstance(value, float):
        a.f = value
    elif isinstance(value, (bool, int)):
        a.i = int(value)
    elif isinstance(value, (str, bytes)):
        a.s = value if isinstance(value, bytes) else value.encode("utf-8")
    elif isinstance(value, (list, tuple)):
        if all(isinstance(v, (float, np.floating)) for v in value) and len(value):
            a.floats = [float(v) for v in value]
        elif all(isinstance(v, (bool, int, np.integer)) for v in value):
            a.ints = [int(v) for v in value]
        elif all(isinstance(v, (str, bytes)) for v in value):
            a.strings = [v if isinstance(v, bytes) else v.encode("utf-8") for v in value]
        else:
            raise ValueError("Unknown argument list type for %s" % key)
    else:
        raise ValueError("Unknown argument type: key=%s value=%r" % (key, value))
    return a


def _names(x):
    if isinstance(x, (str, bytes)) or not hasattr(x, "__iter__"):
        x = [x]
    return [str(v) for v in x]


def CreateOperator(operator_type, inputs, outputs, name="", control_input=None,
                   device_option=None, arg=None, engine=None, **kwargs):
    op = OperatorDef()
    op.type, op.name = operator_type, name
    op.input = _names(inputs)
    op.output = _names(outputs)
    if control_input:
        op.control_input = _names(control_input)
    if device_option is not None:
        op.device_option = device_option
    elif CurrentDeviceScope() is not None:
        op.device_option = CurrentDeviceScope()
    if engine is not None:
        op.engine = engine
    if arg is not None:
        op.arg.extend(arg)
    for k in sorted(kwargs):
        if kwargs[k] is not None:
            op.arg.append(MakeArgument(k, kwargs[k]))
    return op


class GradientRegistry(object):
    @classmethod
    def GetGradientForOp(cls, op, g_output):
        """Returns (gradient_ops, g_input) like core.GradientRegistry
        .GetGradientForOp: g_input[i] is the gradient blob name of op.input[i]
        or None."""
        L = _capi.load()
        ser = op.SerializeToString()
        names = "\n".join(g if g is not None else "" for g in g_output).encode()
        cap = 1 << 16
        out = C.create_string_buffer(cap)
        gin = C.create_string_buffer(cap)
        out_len, n_defs = C.c_size_t(0), C.c_int(0)
        _capi.check(L.c2hip_get_gradient_defs(ser, len(ser), names, out, cap, C.byref(out_len),
                                              C.byref(n_defs), gin, cap))
        raw, pos, ops = out.raw[:out_len.value], 0, []
        for _ in range(n_defs.value):
            (ln,) = struct.unpack_from("<I", raw, pos)
            ops.append(OperatorDef().ParseFromString(raw[pos + 4:pos + 4 + ln]))
            pos += 4 + ln
        g_input = [s if s else None for s in gin.value.decode().split("\n")]
        g_input += [None] * (len(op.input) - len(g_input))
        return ops, g_input[:len(op.input)]


class Net(object):
    """caffe2.python.core.Net: `net.<OpType>(inputs, outputs, **args)`; Proto() is the NetDef
    that workspace.CreateNet serializes (caffe2.proto:176-215)."""

    def __init__(self, name):
        self._net = caffe2_pb2.NetDef(name)

    def Name(self):
        return self._net.name

    def Proto(self):
        return self._net

    def __getattr__(self, op_type):
        if op_type.startswith("__"):
            raise AttributeError(op_type)
        if not IsOperator(op_type):
            raise AttributeError("Method " + op_type + " is not a registered operator.")
        return lambda *a, **kw: self._CreateAndAddToSelf(op_type, *a, **kw)

    def _CreateAndAddToSelf(self, op_type, inputs=None, outputs=None, **kwargs):
        inputs = [] if inputs is None else _names(inputs)
        if outputs is None:
            outputs = [ScopedName("%s_%d" % (op_type, len(self._net.op)))]
        outputs = _names(outputs)
        op = CreateOperator(op_type, inputs, outputs, **kwargs)
        self._net.op.append(op)
        return outputs[0] if len(outputs) == 1 else tuple(outputs)

    def AddGradientOperators(self, ys, skip=0):
        """ys: {blob: gradient blob} (already-computed loss gradients) or a
        list of blobs (a ConstantFill(1.0) `<blob>_autogen_grad` is added for
        each).  Appends the backward ops and returns {blob: gradient blob}."""
        if not isinstance(ys, dict):
            gen = {}
            for y in _names(ys):
                g = self.ConstantFill([y], [y + "_autogen_grad"], value=1.0)
                gen[y] = g
            ys = gen
        fwd = list(self._net.op[skip:])
        grad_ops, grad_map = GenerateBackward(fwd, {str(k): str(v) for k, v in ys.items()})
        self._net.op.extend(grad_ops)
        return grad_map


def GenerateBackward(fwd_ops, ys):
    """Reverse-mode graph generation over a straight-line op list.

    Blobs are tracked per write-version (an in-place Relu makes a new version
    of the same name).  A (blob, version) that receives several gradient
    pieces gets them renamed `<blob>_grad_autosplit_<k>` and summed into
    `<blob>_grad` by a Sum op placed right after the last piece
    (core.py:706-741)."""
    version = {}
    in_ver, out_ver = [], []
    for op in fwd_ops:
        in_ver.append([(b, version.get(b, 0)) for b in op.input])
        ov = []
        for b in op.output:
            version[b] = version.get(b, 0) + 1
            ov.append((b, version[b]))
        out_ver.append(ov)

    grads = {}      # (blob, ver) -> list of [grad_name, owning op-list index, slot]
    for y, g in ys.items():
        grads[(y, version.get(y, 0))] = [[g, None, None]]
    out_ops = []    # list of OperatorDef in emission order
    piece_refs = {}

    for idx in range(len(fwd_ops) - 1, -1, -1):
        op = fwd_ops[idx]
        g_out = []
        for key in out_ver[idx]:
            pieces = grads.get(key)
            if not pieces:
                g_out.append(None)
                continue
            if len(pieces) > 1:
                # rename the pieces and emit the accumulation now
                names = []
                for k, (gname, oi, slot) in enumerate(pieces):
                    new = "%s_grad_autosplit_%d" % (key[0], k)
                    if oi is not None:
                        out_ops[oi].output[slot] = new
                    else:
                        new = gname      # an aliased gradient (Sum's SetDense): read in place
                    names.append(new)
                total = key[0] + "_grad"
                out_ops.append(CreateOperator("Sum", names, [total],
                                              device_option=op.device_option))
                grads[key] = [[total, None, None]]
            g_out.append(grads[key][0][0])
        if all(g is None for g in g_out):
            continue
        gops, g_in = GradientRegistry.GetGradientForOp(op, g_out)
        base = len(out_ops)
        out_ops.extend(gops)
        for i, gname in enumerate(g_in):
            if gname is None:
                continue
            # find which emitted op/slot writes gname (last writer wins)
            owner = None
            for j in range(len(gops) - 1, -1, -1):
                if gname in gops[j].output:
                    owner = (base + j, gops[j].output.index(gname))
                    break
            key = in_ver[idx][i]
            entry = [gname, owner[0] if owner else None, owner[1] if owner else None]
            grads.setdefault(key, []).append(entry)
    # inputs of the graph that still hold several pieces (e.g. shared weights)
    for key, pieces in list(grads.items()):
        if len(pieces) > 1:
            names = []
            for k, (gname, oi, slot) in enumerate(pieces):
                new = "%s_grad_autosplit_%d" % (key[0], k)
                if oi is not None:
                    out_ops[oi].output[slot] = new
                else:
                    new = gname
                names.append(new)
            total = key[0] + "_grad"
            owners = [p[1] for p in pieces if p[1] is not None]
            dev = out_ops[owners[0]].device_option if owners else None
            out_ops.append(CreateOperator("Sum", names, [total], device_option=dev))
            grads[key] = [[total, None, None]]
    grad_map = {}
    for (blob, ver), pieces in grads.items():
        if pieces and (blob not in grad_map or ver >= version.get(blob, 0)):
            grad_map[blob] = pieces[0][0]
    return out_ops, grad_map
