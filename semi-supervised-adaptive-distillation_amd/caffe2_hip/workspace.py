"""caffe2.python.workspace, the slice the hot path uses
(caffe2/python/workspace.py + pybind_state.cc): one global workspace,
FeedBlob / FetchBlob / RunOperatorOnce / CreateNet / RunNet."""
import ctypes as C

import numpy as np

from . import _capi, caffe2_pb2, core

# TensorProto::DataType ids (caffe2.proto:33-49); float16 blobs carry fp16 storage for Conv
_DT = {np.dtype(np.float32): 1, np.dtype(np.int32): 2, np.dtype(np.int64): 10,
       np.dtype(np.float16): 12, np.dtype(np.float64): 13}
_NP = {v: k for k, v in _DT.items()}

_ws = None


def _handle():
    global _ws
    if _ws is None:
        _ws = _capi.load().c2hip_workspace_create()
    return _ws


def ResetWorkspace():
    global _ws
    _sync_twins.clear()
    _protos.clear()
    if _ws is not None:
        _capi.load().c2hip_workspace_destroy(_ws)
    _ws = None
    return True


def HasBlob(name):
    return bool(_capi.load().c2hip_has_blob(_handle(), str(name).encode()))


def Blobs():
    L = _capi.load()
    n = L.c2hip_blobs(_handle(), None, 0)
    buf = C.create_string_buffer(n)
    L.c2hip_blobs(_handle(), buf, n)
    return [s for s in buf.value.decode().split("\n") if s]


def FeedBlob(name, arr, device_option=None):
    """Copies a numpy array into blob `name` on the given (or scoped) device."""
    arr = np.ascontiguousarray(arr)
    if arr.dtype not in _DT:
        raise TypeError("FeedBlob: unsupported dtype %s" % arr.dtype)
    dev = device_option if device_option is not None else core.CurrentDeviceScope()
    dtype, gpu = (dev.device_type, dev.cuda_gpu_id) if dev is not None else (caffe2_pb2.CPU, 0)
    dims = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
    _capi.check(_capi.load().c2hip_feed_blob(
        _handle(), str(name).encode(), arr.ctypes.data_as(C.c_void_p), dims, arr.ndim,
        _DT[arr.dtype], dtype, gpu))
    return True


def ShareExternalTensor(name, device_ptr, shape, np_dtype=np.float32, gpu_id=0):
    """Wrap device memory owned by the caller (e.g. torch_tensor.data_ptr())."""
    dims = (C.c_int64 * max(len(shape), 1))(*shape)
    _capi.check(_capi.load().c2hip_share_external(
        _handle(), str(name).encode(), C.c_void_p(device_ptr), dims, len(shape),
        _DT[np.dtype(np_dtype)], gpu_id))


def _info(name):
    dt, dev, nd = C.c_int(), C.c_int(), C.c_int()
    dims = (C.c_int64 * 8)()
    _capi.check(_capi.load().c2hip_blob_info(_handle(), str(name).encode(), C.byref(dt),
                                             C.byref(dev), C.byref(nd), dims))
    return dt.value, dev.value, tuple(dims[i] for i in range(nd.value))


def FetchBlob(name):
    dt, _, shape = _info(name)
    out = np.empty(shape, _NP[dt])
    _capi.check(_capi.load().c2hip_fetch_blob(_handle(), str(name).encode(),
                                              out.ctypes.data_as(C.c_void_p), out.nbytes))
    return out


def BlobDataPtr(name):
    p = _capi.load().c2hip_blob_data_ptr(_handle(), str(name).encode())
    if not p:
        raise _capi.C2Error(_capi.load().c2hip_last_error().decode())
    return p


def RunOperatorOnce(op):
    ser = op.SerializeToString()
    _capi.check(_capi.load().c2hip_run_operator_once(_handle(), ser, len(ser)))
    return True


def RunOperatorsOnce(ops):
    for op in ops:
        RunOperatorOnce(op)
    return True


def _net_proto(net):
    return net.Proto() if hasattr(net, "Proto") else net


def CreateNet(net, overwrite=False):
    """workspace.CreateNet: the serialized NetDef goes to the library, which lowers the operator
    list (Conv + Relu fusion, one multi-level launch per shared filter: csrc/ops/net_lowering.cc)
    and instantiates the operators once."""
    proto = _net_proto(net)
    ser = proto.SerializeToString()
    _capi.check(_capi.load().c2hip_create_net(_handle(), ser, len(ser), int(bool(overwrite))))
    _protos[proto.name] = proto
    _sync_twins.pop(proto.name, None)
    return True


def RunNet(name, num_iter=1, sync_every_op=False):
    """workspace.RunNet (the training loop's one call per iteration, tools/train_net.py:173): the
    created net enqueues its operators in order and synchronises once per run.
    sync_every_op=True runs the net's operators one by one with a synchronisation after each --
    the reference executors' behaviour (caffe2/core/operator.h:378) -- through a twin of the net
    created with the NetDef argument hip_sync_every_op."""
    name = name.Proto().name if hasattr(name, "Proto") else str(name)
    if sync_every_op:
        if name not in _sync_twins:
            if name not in _protos:
                raise _capi.C2Error("Network %s does not exist yet." % name)
            CreateSyncTwin(_protos[name])
        name = _sync_twins[name]
    _capi.check(_capi.load().c2hip_run_net(_handle(), name.encode(), int(num_iter)))
    return True


_sync_twins = {}
_protos = {}


def CreateSyncTwin(net, lowering=False):
    """A second instantiation of `net` whose Run synchronises after every operator (and, by default,
    runs the operator list as written, without the lowering): the reference's execution model, for
    A/B checks against the lowered single-sync net."""
    proto = _net_proto(net)
    twin = caffe2_pb2.NetDef(proto.name + "__sync_every_op")
    twin.op = proto.op
    twin.type, twin.device_option = proto.type, proto.device_option
    twin.external_input, twin.external_output = proto.external_input, proto.external_output
    twin.arg = list(proto.arg) + [core.MakeArgument("hip_sync_every_op", 1),
                                  core.MakeArgument("hip_lowering", int(bool(lowering)))]
    ser = twin.SerializeToString()
    _capi.check(_capi.load().c2hip_create_net(_handle(), ser, len(ser), 1))
    _sync_twins[proto.name] = twin.name
    return twin.name


def RunNetOnce(net):
    ser = _net_proto(net).SerializeToString()
    _capi.check(_capi.load().c2hip_run_net_once(_handle(), ser, len(ser)))
    return True


def DeleteNet(name):
    name = name.Proto().name if hasattr(name, "Proto") else str(name)
    _capi.check(_capi.load().c2hip_delete_net(_handle(), name.encode()))
    _sync_twins.pop(name, None)
    _protos.pop(name, None)
    return True


def Nets():
    L = _capi.load()
    n = L.c2hip_nets(_handle(), None, 0)
    buf = C.create_string_buffer(n)
    L.c2hip_nets(_handle(), buf, n)
    return [s for s in buf.value.decode().split("\n") if s]


def _unpack_defs(raw, count):
    import struct
    pos, ops = 0, []
    for _ in range(count):
        (ln,) = struct.unpack_from("<I", raw, pos)
        ops.append(caffe2_pb2.OperatorDef().ParseFromString(raw[pos + 4:pos + 4 + ln]))
        pos += 4 + ln
    return ops


def LoweredOps(name):
    """The operator list a created net actually runs (after the lowering)."""
    name = name.Proto().name if hasattr(name, "Proto") else str(name)
    L, n_ops = _capi.load(), C.c_int(0)
    need = L.c2hip_net_lowered_ops(_handle(), name.encode(), None, 0, C.byref(n_ops))
    if need == 0 and n_ops.value == 0:
        _capi.check(1 if L.c2hip_last_error() else 0)
    buf = C.create_string_buffer(max(need, 1))
    L.c2hip_net_lowered_ops(_handle(), name.encode(), buf, need, C.byref(n_ops))
    return _unpack_defs(buf.raw[:need], n_ops.value)


def LowerNet(net):
    """The lowering alone (no workspace, no device): -> (operator list, one-line report)."""
    ser = _net_proto(net).SerializeToString()
    L, n_ops = _capi.load(), C.c_int(0)
    rep = C.create_string_buffer(1024)
    need = L.c2hip_lower_net(ser, len(ser), None, 0, C.byref(n_ops), rep, 1024)
    buf = C.create_string_buffer(max(need, 1))
    L.c2hip_lower_net(ser, len(ser), buf, need, C.byref(n_ops), rep, 1024)
    return _unpack_defs(buf.raw[:need], n_ops.value), rep.value.decode()


def Counter(name):
    return int(_capi.load().c2hip_counter(name.encode()))


def SetStream(gpu_id, hip_stream, enabled=True):
    _capi.check(_capi.load().c2hip_set_stream(gpu_id, C.c_void_p(hip_stream), int(enabled)))


# ---------------------------------------------------------------------------
# The data-parallel communicator of this process (include/c2hip_capi.h): what makes the registered
# NCCLAllreduce / NCCLBroadcast operators (optimizer.py:72-92) exchange over RCCL.  One process = one
# GPU: rank 0 calls CommUniqueId() and ships the 128 bytes to every rank (torch.distributed's store, a
# file, MPI: the launcher's business); every rank then calls CommInit.
# ---------------------------------------------------------------------------

def _comm_check(rc):
    if rc != 0:
        raise _capi.C2Error(_capi.load().c2hip_comm_last_error().decode("utf-8", "replace"))


def CommUniqueId():
    import ctypes as C
    buf = C.create_string_buffer(128)
    _comm_check(_capi.load().c2hip_comm_unique_id(buf, 128))
    return buf.raw


def CommInit(unique_id, world, rank, gpu_id=0):
    assert len(unique_id) == 128
    _comm_check(_capi.load().c2hip_comm_init(unique_id, 128, int(world), int(rank), int(gpu_id)))


def CommWorld():
    return int(_capi.load().c2hip_comm_world())


def CommDestroy():
    _comm_check(_capi.load().c2hip_comm_destroy())
