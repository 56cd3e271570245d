"""ctypes bindings for the test-only CPU oracle (oracle/ssad_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never from the product package.

`load()` builds liboracle.so on demand (gcc).  `load_ref()` loads the
host-compiled reference kernel bodies (oracle/_ref/libref_kernels.so) when
they exist; they can only be built where /root/reference is present.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


class DistillParams(C.Structure):
    _fields_ = [
        ("gamma", C.c_float),
        ("alpha", C.c_float),
        ("beta", C.c_float),
        ("num_classes", C.c_int),
        ("ignored_label", C.c_int),
        ("scale", C.c_float),
    ]


class FocalParams(C.Structure):
    _fields_ = [("gamma", C.c_float), ("alpha", C.c_float), ("num_classes", C.c_int),
                ("scale", C.c_float)]


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "kernel_h", "kernel_w", "stride_h", "stride_w",
        "pad_t", "pad_l", "pad_b", "pad_r", "dilation_h", "dilation_w")]


def build(ref=False):
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.check_call(["make", "-s", "-C", _HERE] + targets)


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "ssad_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build()
    lib = C.CDLL(path)
    lib.oracle_sum128.restype = C.c_float
    lib.oracle_sum128.argtypes = [f32p, C.c_int64]
    lib.oracle_sum_f64.restype = C.c_double
    lib.oracle_sum_f64.argtypes = [f32p, C.c_int64]
    lib.oracle_distill_loss_forward.restype = None
    lib.oracle_distill_loss_forward.argtypes = [
        f32p, f32p, i32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int,
        C.POINTER(DistillParams), C.c_void_p, f64p]
    lib.oracle_distill_loss_backward.restype = None
    lib.oracle_distill_loss_backward.argtypes = [
        f32p, f32p, i32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int,
        C.POINTER(DistillParams), f32p]
    lib.oracle_focal_loss_forward.restype = None
    lib.oracle_focal_loss_forward.argtypes = [
        f32p, i32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(FocalParams),
        C.c_void_p, f64p]
    lib.oracle_focal_loss_backward.restype = None
    lib.oracle_focal_loss_backward.argtypes = [
        f32p, i32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(FocalParams), f32p]
    lib.oracle_select_smooth_l1_forward.restype = None
    lib.oracle_select_smooth_l1_forward.argtypes = [
        f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
        C.c_float, f64p]
    lib.oracle_select_smooth_l1_backward.restype = None
    lib.oracle_select_smooth_l1_backward.argtypes = [
        f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
        C.c_float, f32p]
    lib.oracle_pow_sum.restype = None
    lib.oracle_pow_sum.argtypes = [
        C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, C.c_float, f64p]
    for name in ("oracle_relu", "oracle_sigmoid"):
        getattr(lib, name).restype = None
        getattr(lib, name).argtypes = [f32p, f32p, C.c_int64]
    lib.oracle_relu_grad.restype = None
    lib.oracle_relu_grad.argtypes = [f32p, f32p, f32p, C.c_int64]
    lib.oracle_conv_out_dims.restype = None
    lib.oracle_conv_out_dims.argtypes = [
        C.c_int, C.c_int, C.POINTER(ConvGeom), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.oracle_conv_forward.restype = None
    lib.oracle_conv_forward.argtypes = [
        f32p, f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
        C.POINTER(ConvGeom), f32p]
    lib.oracle_conv_backward.restype = None
    lib.oracle_conv_backward.argtypes = [
        f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
        C.POINTER(ConvGeom), f32p, C.c_void_p, C.c_void_p]
    lib.oracle_sgd_update.restype = None
    lib.oracle_sgd_update.argtypes = [
        f32p, f32p, f32p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int]
    lib.oracle_num_threads.restype = C.c_int
    lib.oracle_set_num_threads.argtypes = [C.c_int]
    _LIB = lib
    return lib


def load_ref():
    """Host-compiled reference kernel bodies, or None when not built."""
    global _REF
    if _REF is not None:
        return _REF
    path = os.path.join(_HERE, "_ref", "libref_kernels.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    lib.ref_distill_loss_kernel.restype = None
    lib.ref_distill_loss_kernel.argtypes = [
        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, i32p, f32p,
        C.c_float, C.c_float, C.c_float, C.c_int, f32p]
    lib.ref_focal_loss_kernel.restype = None
    lib.ref_focal_loss_kernel.argtypes = [
        C.c_int, C.c_int, C.c_int, C.c_int, f32p, i32p, f32p, C.c_float, C.c_float, C.c_int, f32p]
    lib.ref_focal_grad_kernel.restype = None
    lib.ref_focal_grad_kernel.argtypes = [
        C.c_int, C.c_int, C.c_int, C.c_int, f32p, i32p, f32p, f32p, C.c_float, C.c_float,
        C.c_int, f32p]
    lib.ref_smoothl1_kernel.restype = None
    lib.ref_smoothl1_kernel.argtypes = [
        C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, f32p, f32p, C.c_float]
    lib.ref_smoothl1_grad_kernel.restype = None
    lib.ref_smoothl1_grad_kernel.argtypes = [
        C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, f32p, f32p, C.c_float, f32p,
        C.c_float]
    lib.ref_distill_grad_kernel.restype = None
    lib.ref_distill_grad_kernel.argtypes = [
        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, i32p, f32p,
        f32p, C.c_float, C.c_float, C.c_float, C.c_int, f32p]
    _REF = lib
    return lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _params(gamma, alpha, beta, num_classes, ignored_label, scale):
    return DistillParams(gamma, alpha, beta, num_classes, ignored_label, scale)


def set_num_threads(n):
    load().oracle_set_num_threads(int(n))


def num_threads():
    return int(load().oracle_num_threads())


def sum128(x):
    x = _c(x, np.float32).ravel()
    return float(load().oracle_sum128(x, x.size))


def distill_loss_forward(logits, teacher, labels, normalizer, *, gamma=1.0,
                         alpha=0.25, beta=0.0, num_classes=80,
                         ignored_label=-1, scale=1.0, want_elems=False):
    """Returns (loss_ref_order, loss_f64, per_element_or_None)."""
    logits = _c(logits, np.float32)
    teacher = _c(teacher, np.float32)
    labels = _c(labels, np.int32)
    wp = _c(np.asarray(normalizer).reshape(-1)[:1], np.float32)
    N, D, H, W = logits.shape
    P = _params(gamma, alpha, beta, num_classes, ignored_label, scale)
    out = np.zeros(2, np.float64)
    elems = np.empty(logits.shape, np.float32) if want_elems else None
    load().oracle_distill_loss_forward(
        logits, teacher, labels, wp, N, D, H, W, C.byref(P),
        elems.ctypes.data if want_elems else None, out)
    return np.float32(out[0]), float(out[1]), elems


def distill_loss_backward(logits, teacher, labels, normalizer, dloss=1.0, *,
                          gamma=1.0, alpha=0.25, beta=0.0, num_classes=80,
                          ignored_label=-1, scale=1.0):
    logits = _c(logits, np.float32)
    teacher = _c(teacher, np.float32)
    labels = _c(labels, np.int32)
    wp = _c(np.asarray(normalizer).reshape(-1)[:1], np.float32)
    go = _c(np.asarray(dloss).reshape(-1)[:1], np.float32)
    N, D, H, W = logits.shape
    P = _params(gamma, alpha, beta, num_classes, ignored_label, scale)
    dX = np.empty(logits.shape, np.float32)
    load().oracle_distill_loss_backward(
        logits, teacher, labels, wp, go, N, D, H, W, C.byref(P), dX)
    return dX


def focal_loss_forward(logits, labels, normalizer, *, gamma=1.0, alpha=0.25, num_classes=80,
                       scale=1.0, want_elems=False):
    """SigmoidFocalLoss: returns (loss_ref_order, loss_f64, per_element_or_None)."""
    logits = _c(logits, np.float32)
    labels = _c(labels, np.int32)
    wp = _c(np.asarray(normalizer).reshape(-1)[:1], np.float32)
    N, D, H, W = logits.shape
    P = FocalParams(gamma, alpha, num_classes, scale)
    out = np.zeros(2, np.float64)
    elems = np.empty(logits.shape, np.float32) if want_elems else None
    load().oracle_focal_loss_forward(logits, labels, wp, N, D, H, W, C.byref(P),
                                     elems.ctypes.data if want_elems else None, out)
    return np.float32(out[0]), float(out[1]), elems


def focal_loss_backward(logits, labels, normalizer, dloss=1.0, *, gamma=1.0, alpha=0.25,
                        num_classes=80, scale=1.0):
    logits = _c(logits, np.float32)
    labels = _c(labels, np.int32)
    wp = _c(np.asarray(normalizer).reshape(-1)[:1], np.float32)
    go = _c(np.asarray(dloss).reshape(-1)[:1], np.float32)
    N, D, H, W = logits.shape
    P = FocalParams(gamma, alpha, num_classes, scale)
    dX = np.empty(logits.shape, np.float32)
    load().oracle_focal_loss_backward(logits, labels, wp, go, N, D, H, W, C.byref(P), dX)
    return dX


def select_smooth_l1_forward(Y_hat, Y, L, S, *, beta=1.0, scale=1.0):
    """SelectSmoothL1Loss: returns (loss_ref_order, loss_f64)."""
    Y_hat = _c(Y_hat, np.float32)
    Y = _c(Y, np.float32).reshape(-1, 4)
    L = _c(L, np.float32).reshape(-1, 4)
    S = _c(np.asarray(S).reshape(-1)[:1], np.float32)
    N, D, H, W = Y_hat.shape
    out = np.zeros(2, np.float64)
    load().oracle_select_smooth_l1_forward(Y_hat, Y.ravel(), L.ravel(), S, N, D, H, W, Y.shape[0],
                                           beta, scale, out)
    return np.float32(out[0]), float(out[1])


def select_smooth_l1_backward(Y_hat, Y, L, S, dloss=1.0, *, beta=1.0, scale=1.0):
    Y_hat = _c(Y_hat, np.float32)
    Y = _c(Y, np.float32).reshape(-1, 4)
    L = _c(L, np.float32).reshape(-1, 4)
    S = _c(np.asarray(S).reshape(-1)[:1], np.float32)
    go = _c(np.asarray(dloss).reshape(-1)[:1], np.float32)
    N, D, H, W = Y_hat.shape
    out = np.empty(Y_hat.shape, np.float32)
    load().oracle_select_smooth_l1_backward(Y_hat, Y.ravel(), L.ravel(), S, go, N, D, H, W,
                                            Y.shape[0], beta, scale, out)
    return out


def pow_sum(inputs, power=1.0):
    """Returns (ref_order_f32, f64_sum)."""
    arrs = [_c(a, np.float32).ravel() for a in inputs]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    sizes = (C.c_int64 * len(arrs))(*[a.size for a in arrs])
    out = np.zeros(2, np.float64)
    load().oracle_pow_sum(ptrs, sizes, len(arrs), power, out)
    return np.float32(out[0]), float(out[1])


def relu(x):
    x = _c(x, np.float32)
    y = np.empty_like(x)
    load().oracle_relu(x.ravel(), y.ravel(), x.size)
    return y


def relu_grad(y, dy):
    y = _c(y, np.float32)
    dy = _c(dy, np.float32)
    dx = np.empty_like(y)
    load().oracle_relu_grad(y.ravel(), dy.ravel(), dx.ravel(), y.size)
    return dx


def sigmoid(x):
    x = _c(x, np.float32)
    y = np.empty_like(x)
    load().oracle_sigmoid(x.ravel(), y.ravel(), x.size)
    return y


def _geom(kernel=3, stride=1, pad=1, dilation=1):
    return ConvGeom(kernel, kernel, stride, stride, pad, pad, pad, pad,
                    dilation, dilation)


def conv_forward(X, Wt, bias=None, *, kernel=3, stride=1, pad=1, dilation=1):
    X = _c(X, np.float32)
    Wt = _c(Wt, np.float32)
    N, Cin, H, W = X.shape
    M = Wt.shape[0]
    g = _geom(kernel, stride, pad, dilation)
    ho, wo = C.c_int(), C.c_int()
    lib = load()
    lib.oracle_conv_out_dims(H, W, C.byref(g), C.byref(ho), C.byref(wo))
    Y = np.empty((N, M, ho.value, wo.value), np.float32)
    b = _c(bias, np.float32) if bias is not None else None
    lib.oracle_conv_forward(X, Wt, b.ctypes.data if b is not None else None,
                            N, Cin, H, W, M, C.byref(g), Y)
    return Y


def conv_backward(X, Wt, dY, *, kernel=3, stride=1, pad=1, dilation=1,
                  want_db=True, want_dx=True):
    """Returns (dW, db, dX) as ConvGradient does (conv_gradient_op.cc:35-77)."""
    X = _c(X, np.float32)
    Wt = _c(Wt, np.float32)
    dY = _c(dY, np.float32)
    N, Cin, H, W = X.shape
    M = Wt.shape[0]
    g = _geom(kernel, stride, pad, dilation)
    dW = np.empty_like(Wt)
    db = np.empty(M, np.float32) if want_db else None
    dX = np.empty_like(X) if want_dx else None
    load().oracle_conv_backward(
        X, Wt, dY, N, Cin, H, W, M, C.byref(g), dW,
        db.ctypes.data if want_db else None,
        dX.ctypes.data if want_dx else None)
    return dW, db, dX


def sgd_update(w, g, m, lr, momentum=0.9, weight_decay=1e-4, is_bias=False):
    """In place on copies; returns (w, g, m)."""
    w = _c(w, np.float32).copy()
    g = _c(g, np.float32).copy()
    m = _c(m, np.float32).copy()
    load().oracle_sgd_update(w.ravel(), g.ravel(), m.ravel(), w.size, lr,
                             momentum, weight_decay, int(is_bias))
    return w, g, m


# ---- host-compiled reference kernels (container only) ----------------------

def ref_distill_loss_elems(logits, teacher, labels, normalizer, *, gamma,
                           alpha, beta, num_classes, ignored_label):
    lib = load_ref()
    assert lib is not None, "oracle/_ref not built (needs /root/reference)"
    logits = _c(logits, np.float32)
    teacher = _c(teacher, np.float32)
    labels = _c(labels, np.int32)
    wp = _c(np.asarray(normalizer).reshape(-1)[:1], np.float32)
    N, D, H, W = logits.shape
    out = np.empty(logits.shape, np.float32)
    lib.ref_distill_loss_kernel(N, D, H, W, ignored_label, logits, teacher,
                                labels, wp, gamma, alpha, beta, num_classes, out)
    return out


def ref_distill_grad_elems(logits, teacher, labels, normalizer, dloss, *,
                           gamma, alpha, beta, num_classes, ignored_label):
    """dX straight out of the reference kernel (before the *scale pass)."""
    lib = load_ref()
    assert lib is not None, "oracle/_ref not built (needs /root/reference)"
    logits = _c(logits, np.float32)
    teacher = _c(teacher, np.float32)
    labels = _c(labels, np.int32)
    wp = _c(np.asarray(normalizer).reshape(-1)[:1], np.float32)
    go = _c(np.asarray(dloss).reshape(-1)[:1], np.float32)
    N, D, H, W = logits.shape
    out = np.empty(logits.shape, np.float32)
    lib.ref_distill_grad_kernel(N, D, H, W, ignored_label, logits, teacher,
                                labels, out, wp, gamma, alpha, beta,
                                num_classes, go)
    return out


def ref_focal_elems(logits, labels, normalizer, dloss, *, gamma, alpha, num_classes):
    """(per-element loss, dX before the *scale pass) from the reference kernels."""
    lib = load_ref()
    assert lib is not None, "oracle/_ref not built (needs /root/reference)"
    logits = _c(logits, np.float32)
    labels = _c(labels, np.int32)
    wp = _c(np.asarray(normalizer).reshape(-1)[:1], np.float32)
    go = _c(np.asarray(dloss).reshape(-1)[:1], np.float32)
    N, D, H, W = logits.shape
    le, dx = np.empty(logits.shape, np.float32), np.empty(logits.shape, np.float32)
    lib.ref_focal_loss_kernel(N, D, H, W, logits, labels, wp, gamma, alpha, num_classes, le)
    lib.ref_focal_grad_kernel(N, D, H, W, logits, labels, dx, wp, gamma, alpha, num_classes, go)
    return le, dx


def ref_smoothl1_elems(Y_hat, Y, L, S, dloss, *, beta, norm):
    """(zero-initialised loss buffer, d_Y_hat) from the reference kernels."""
    lib = load_ref()
    assert lib is not None, "oracle/_ref not built (needs /root/reference)"
    Y_hat = _c(Y_hat, np.float32)
    Y = _c(Y, np.float32).reshape(-1, 4)
    L = _c(L, np.float32).reshape(-1, 4)
    S = _c(np.asarray(S).reshape(-1)[:1], np.float32)
    go = _c(np.asarray(dloss).reshape(-1)[:1], np.float32)
    N, D, H, W = Y_hat.shape
    buf, dy = np.zeros(Y_hat.shape, np.float32), np.zeros(Y_hat.shape, np.float32)
    lib.ref_smoothl1_kernel(D, H, W, Y.shape[0], Y_hat, Y.ravel(), L.ravel(), buf, S, beta)
    lib.ref_smoothl1_grad_kernel(D, H, W, Y.shape[0], Y_hat, Y.ravel(), L.ravel(), dy, go, norm,
                                 S, beta)
    return buf, dy


# ---- the reference's own CPU Conv / ConvGradient operators (container build) ----------
# oracle/_ref/libref_conv.so = ConvOp<float, CPUContext> / ConvGradientOp<float, CPUContext>
# (caffe2/operators/conv_op_impl.h:31-202, :358-577 + utils/math_cpu.cc on the vendored Eigen)
# compiled from /root/reference by oracle/build_ref_conv.sh.  Used to PIN conv_forward /
# conv_backward above and to generate tests/golden/conv_ref.npz.

_REFCONV = None


def load_ref_conv():
    """The reference's compiled CPU convolution operators, or None when not built."""
    global _REFCONV
    if _REFCONV is not None:
        return _REFCONV
    path = os.path.join(_HERE, "_ref", "libref_conv.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    lib.ref_conv_forward.argtypes = [f32p, f32p, C.c_void_p] + [C.c_int] * 9 + [f32p]
    lib.ref_conv_backward.argtypes = [f32p, f32p, f32p] + [C.c_int] * 9 + [f32p, f32p, f32p]
    _REFCONV = lib
    return lib


def ref_conv_forward(X, Wt, bias=None, *, kernel=3, stride=1, pad=1, group=1):
    lib = load_ref_conv()
    X, Wt = _c(X, np.float32), _c(Wt, np.float32)
    N, Cin, H, W = X.shape
    M = Wt.shape[0]
    oh, ow = (H + 2 * pad - kernel) // stride + 1, (W + 2 * pad - kernel) // stride + 1
    Y = np.empty((N, M, oh, ow), np.float32)
    b = _c(bias, np.float32) if bias is not None else None
    rc = lib.ref_conv_forward(X, Wt, b.ctypes.data if b is not None else None, N, Cin, H, W, M,
                              kernel, pad, stride, group, Y)
    if rc != 0:
        raise RuntimeError("reference Conv failed (%d)" % rc)
    return Y


def ref_conv_backward(X, Wt, dY, *, kernel=3, stride=1, pad=1, group=1):
    """(dW, db, dX) of the reference's ConvGradient."""
    lib = load_ref_conv()
    X, Wt, dY = _c(X, np.float32), _c(Wt, np.float32), _c(dY, np.float32)
    N, Cin, H, W = X.shape
    M = Wt.shape[0]
    dW, db, dX = np.empty_like(Wt), np.empty(M, np.float32), np.empty_like(X)
    rc = lib.ref_conv_backward(X, Wt, dY, N, Cin, H, W, M, kernel, pad, stride, group, dW, db, dX)
    if rc != 0:
        raise RuntimeError("reference ConvGradient failed (%d)" % rc)
    return dW, db, dX
