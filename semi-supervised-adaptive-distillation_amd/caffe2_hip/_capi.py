"""ctypes view of include/c2hip_capi.h."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libcaffe2_detectron_ops_hip.so")
_lib = None


class C2Error(RuntimeError):
    """An EnforceNotMet (or other C++ exception) raised inside the library."""


def load(path=None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise C2Error("%s not built (run __graft_entry__.build())" % p)
    L = C.CDLL(p)
    vp, cp, sz, i32 = C.c_void_p, C.c_char_p, C.c_size_t, C.c_int
    L.c2hip_last_error.restype = cp
    L.c2hip_workspace_create.restype = vp
    L.c2hip_workspace_destroy.argtypes = [vp]
    L.c2hip_has_blob.argtypes = [vp, cp]
    L.c2hip_remove_blob.argtypes = [vp, cp]
    L.c2hip_blobs.restype = sz
    L.c2hip_blobs.argtypes = [vp, cp, sz]
    L.c2hip_feed_blob.argtypes = [vp, cp, vp, C.POINTER(C.c_int64), i32, i32, i32, i32]
    L.c2hip_blob_info.argtypes = [vp, cp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32),
                                  C.POINTER(C.c_int64)]
    L.c2hip_fetch_blob.argtypes = [vp, cp, vp, sz]
    L.c2hip_blob_data_ptr.restype = vp
    L.c2hip_blob_data_ptr.argtypes = [vp, cp]
    L.c2hip_share_external.argtypes = [vp, cp, vp, C.POINTER(C.c_int64), i32, i32, i32]
    L.c2hip_run_operator_once.argtypes = [vp, cp, sz]
    L.c2hip_create_operator.restype = vp
    L.c2hip_create_operator.argtypes = [vp, cp, sz]
    L.c2hip_run_operator.argtypes = [vp, i32]
    L.c2hip_destroy_operator.argtypes = [vp]
    L.c2hip_create_net.argtypes = [vp, cp, sz, i32]
    L.c2hip_run_net.argtypes = [vp, cp, i32]
    L.c2hip_run_net_once.argtypes = [vp, cp, sz]
    L.c2hip_delete_net.argtypes = [vp, cp]
    L.c2hip_nets.restype = sz
    L.c2hip_nets.argtypes = [vp, cp, sz]
    L.c2hip_net_lowered_ops.restype = sz
    L.c2hip_net_lowered_ops.argtypes = [vp, cp, vp, sz, C.POINTER(i32)]
    L.c2hip_lower_net.restype = sz
    L.c2hip_lower_net.argtypes = [cp, sz, vp, sz, C.POINTER(i32), cp, sz]
    L.c2hip_counter.restype = C.c_longlong
    L.c2hip_counter.argtypes = [cp]
    L.c2hip_registered_operators.restype = sz
    L.c2hip_registered_operators.argtypes = [i32, cp, sz]
    L.c2hip_has_schema.argtypes = [cp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32),
                                   C.POINTER(i32)]
    L.c2hip_get_gradient_defs.argtypes = [cp, sz, cp, vp, sz, C.POINTER(sz), C.POINTER(i32),
                                          cp, sz]
    L.c2hip_set_stream.argtypes = [i32, vp, i32]
    L.c2hip_device_synchronize.argtypes = [i32]
    L.c2hip_comm_last_error.restype = C.c_char_p
    L.c2hip_comm_unique_id.argtypes = [vp, sz]
    L.c2hip_comm_init.argtypes = [vp, sz, i32, i32, i32]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise C2Error(load().c2hip_last_error().decode("utf-8", "replace"))
