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
_nets = {}


def _handle():
    global _ws
    if _ws is None:
        _ws = _capi.load().c2hip_workspace_create()
    return _ws


def ResetWorkspace():
    global _ws
    for ops in _nets.values():
        for h in ops:
            _capi.load().c2hip_destroy_operator(h)
    _nets.clear()
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


def CreateNet(net, overwrite=False):
    proto = net.Proto() if hasattr(net, "Proto") else net
    if proto.name in _nets:
        if not overwrite:
            raise _capi.C2Error("net %s already exists" % proto.name)
        for h in _nets.pop(proto.name):
            _capi.load().c2hip_destroy_operator(h)
    handles = []
    for op in proto.op:
        ser = op.SerializeToString()
        h = _capi.load().c2hip_create_operator(_handle(), ser, len(ser))
        if not h:
            raise _capi.C2Error(_capi.load().c2hip_last_error().decode("utf-8", "replace"))
        handles.append(h)
    _nets[proto.name] = handles
    return True


def RunNet(name, num_iter=1, sync_every_op=False):
    """Runs the ops of a created net in order.  sync_every_op=True is the
    reference's behaviour (caffe2/core/operator.h:378 syncs the stream after
    every operator); the default enqueues the whole net and syncs once."""
    name = name.Proto().name if hasattr(name, "Proto") else str(name)
    L = _capi.load()
    for _ in range(num_iter):
        for h in _nets[name]:
            _capi.check(L.c2hip_run_operator(h, 1 if sync_every_op else 0))
        if not sync_every_op:
            _capi.check(L.c2hip_device_synchronize(0))
    return True


def RunNetOnce(net):
    CreateNet(net, overwrite=True)
    return RunNet(net)


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
