"""ctypes binding of the raw kernel launchers (include/ssad_kernels.h).

torch is used here only as plumbing: device memory (torch tensors) and the
current HIP stream.  Every function launches hand-written gfx950 kernels from
libcaffe2_detectron_ops_hip.so; there is NO fallback -- if the library is
missing or a launch fails this raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libcaffe2_detectron_ops_hip.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)

MAX_LEVELS = 8
CONV_RELU = 1
CONV_MASK_AUX = 2
CONV_SIGMOID = 4
CONV_SPLIT_TAIL = 8


class KernelError(RuntimeError):
    pass


class DistillParams(C.Structure):
    _fields_ = [("gamma", C.c_float), ("alpha", C.c_float), ("beta", C.c_float),
                ("num_classes", C.c_int), ("ignored_label", C.c_int),
                ("scale", C.c_float)]


class FocalParams(C.Structure):
    _fields_ = [("gamma", C.c_float), ("alpha", C.c_float), ("num_classes", C.c_int),
                ("scale", C.c_float)]


class DistillLevel(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("teacher_prob", C.c_void_p),
                ("labels", C.c_void_p), ("out", C.c_void_p),
                ("N", C.c_int), ("D", C.c_int), ("H", C.c_int), ("W", C.c_int)]


class SmoothL1Level(C.Structure):
    _fields_ = [("Y_hat", C.c_void_p), ("Y", C.c_void_p), ("L", C.c_void_p), ("loss", C.c_void_p),
                ("dY_hat", C.c_void_p), ("N", C.c_int), ("D", C.c_int), ("H", C.c_int),
                ("W", C.c_int), ("M", C.c_int)]


class PackEntry(C.Structure):
    _fields_ = [("w", C.c_void_p), ("Cout", C.c_int), ("Cin", C.c_int), ("packed_fwd", C.c_void_p),
                ("packed_dgrad", C.c_void_p)]


class TransposeEntry(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wt", C.c_void_p), ("M", C.c_int), ("K", C.c_int), ("ldm", C.c_int),
                ("reserved", C.c_int)]


class F16PackEntry(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wf", C.c_void_p), ("wd", C.c_void_p), ("M", C.c_int), ("C", C.c_int),
                ("taps", C.c_int), ("reserved", C.c_int)]


class SgdSegment(C.Structure):
    _fields_ = [("offset", C.c_int64), ("n", C.c_int64), ("is_bias", C.c_int), ("row_len", C.c_int),
                ("row_scale", C.c_void_p)]


class GemmConv(C.Structure):
    _fields_ = [("a", C.c_void_p), ("x", C.c_void_p), ("y", C.c_void_p), ("bias", C.c_void_p),
                ("residual", C.c_void_p), ("mask", C.c_void_p), ("lda", C.c_int), ("N", C.c_int),
                ("K", C.c_int), ("P", C.c_int), ("M", C.c_int), ("flags", C.c_int)]


class PwF16(C.Structure):
    """ssad_pw_f16 (include/ssad_kernels.h): one pointwise convolution on blocked fp16 tensors."""
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
                ("mask", C.c_void_p), ("y", C.c_void_p), ("N", C.c_int), ("C", C.c_int), ("M", C.c_int),
                ("Ho", C.c_int), ("Wo", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int), ("stride", C.c_int),
                ("flags", C.c_int)]


class ConvLevel(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("aux", C.c_void_p),
                ("N", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("packed", C.c_void_p), ("bias", C.c_void_p)]


class GemmPackEntry(C.Structure):
    _fields_ = [("a", C.c_void_p), ("dst", C.c_void_p), ("lda", C.c_int), ("K", C.c_int), ("M", C.c_int)]


class F16WgradLevel(C.Structure):
    _fields_ = [("x", C.c_void_p), ("dy", C.c_void_p), ("N", C.c_int), ("H", C.c_int), ("W", C.c_int)]


class F16Level(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("aux", C.c_void_p),
                ("N", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("packed", C.c_void_p), ("bias", C.c_void_p)]


_lib = None


def lib():
    """The HIP extension; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KernelError(
            "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, sz, i32, i64, f32 = C.c_void_p, C.c_size_t, C.c_int, C.c_int64, C.c_float
    L.ssad_distill_loss_workspace_bytes.restype = sz
    L.ssad_distill_loss_workspace_bytes.argtypes = [i32]
    L.ssad_distill_loss_forward.argtypes = [
        C.POINTER(DistillLevel), i32, vp, C.POINTER(DistillParams), vp, sz, vp]
    L.ssad_distill_loss_backward.argtypes = [
        C.POINTER(DistillLevel), i32, vp, vp, i32, C.POINTER(DistillParams), vp]
    L.ssad_focal_loss_forward.argtypes = [
        C.POINTER(DistillLevel), i32, vp, C.POINTER(FocalParams), vp, sz, vp]
    L.ssad_focal_loss_backward.argtypes = [
        C.POINTER(DistillLevel), i32, vp, vp, i32, C.POINTER(FocalParams), vp]
    L.ssad_cls_losses_fused_workspace_bytes.restype = sz
    L.ssad_cls_losses_fused_workspace_bytes.argtypes = [i32]
    L.ssad_cls_losses_fused.argtypes = [
        C.POINTER(DistillLevel), i32, vp, vp, C.POINTER(DistillParams), C.POINTER(FocalParams),
        vp, vp, vp, sz, vp]
    L.ssad_cls_losses_fused_prezeroed.argtypes = L.ssad_cls_losses_fused.argtypes
    L.ssad_select_smooth_l1_workspace_bytes.restype = sz
    L.ssad_select_smooth_l1_workspace_bytes.argtypes = [i32]
    L.ssad_select_smooth_l1_forward.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, f32,
                                                vp, vp, sz, vp]
    L.ssad_select_smooth_l1_levels.argtypes = [C.POINTER(SmoothL1Level), i32, vp, vp, f32, f32, i32,
                                               vp, sz, vp]
    L.ssad_select_smooth_l1_backward.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32,
                                                 f32, vp, vp]
    L.ssad_fill.argtypes = [vp, f32, i64, vp]
    L.ssad_pow_sum_workspace_bytes.restype = sz
    L.ssad_pow_sum_workspace_bytes.argtypes = [i32]
    L.ssad_pow_sum.argtypes = [C.POINTER(vp), C.POINTER(i64), i32, f32, vp, vp, sz, vp]
    L.ssad_pow_sum_prezeroed.argtypes = L.ssad_pow_sum.argtypes
    L.ssad_relu.argtypes = [vp, vp, i64, vp]
    L.ssad_relu_grad.argtypes = [vp, vp, vp, i64, vp]
    L.ssad_sigmoid.argtypes = [vp, vp, i64, vp]
    L.ssad_sum_n.argtypes = [C.POINTER(vp), i32, vp, i64, vp]
    L.ssad_scale.argtypes = [vp, vp, f32, i64, vp]
    L.ssad_affine_channel.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.ssad_upsample_nearest.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.ssad_upsample_nearest_grad.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    L.ssad_max_pool3x3s2_bias_relu.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
    L.ssad_relu_grad_rowsum.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    L.ssad_conv1x1_bias_act.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.ssad_conv1x1_bias_act2.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.ssad_f16_pack_activations.argtypes = [vp, i32, i32, i32, i32, f32, vp, vp]
    L.ssad_f16_unpack_activations.argtypes = [vp, i32, i32, i32, i32, f32, vp, vp]
    L.ssad_f16_filter_halves.restype = sz
    L.ssad_f16_filter_halves.argtypes = [i32, i32]
    L.ssad_f16_pack_filter.argtypes = [vp, i32, i32, vp, vp, vp]
    L.ssad_conv3x3_forward_f16.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
    L.ssad_conv3x3_forward_f16_levels.argtypes = [C.POINTER(F16Level), i32, vp, vp, i32, i32, i32, vp]
    L.ssad_conv3x3_wgrad_f16_workspace_bytes.restype = sz
    L.ssad_conv3x3_wgrad_f16_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    L.ssad_conv3x3_wgrad_f16.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, sz, vp]
    L.ssad_conv3x3_wgrad_f16_levels_workspace_bytes.restype = sz
    L.ssad_conv3x3_wgrad_f16_levels_workspace_bytes.argtypes = [C.POINTER(F16WgradLevel), i32, i32, i32]
    L.ssad_conv3x3_wgrad_f16_levels.argtypes = [C.POINTER(F16WgradLevel), i32, i32, i32, i32, f32, vp, vp,
                                                vp, sz, vp]
    L.ssad_momentum_sgd_update.argtypes = [vp, vp, vp, vp, f32, f32, i32, i64, vp]
    L.ssad_momentum_sgd_flat.argtypes = [vp, vp, vp, vp, f32, f32, C.POINTER(SgdSegment), i32, vp, vp]
    L.ssad_check_finite.argtypes = [vp, i64, vp, vp]
    L.ssad_loss_scale_update.argtypes = [vp, vp, f32, f32, i32, f32, f32, vp]
    L.ssad_conv_wino_pack_filters.argtypes = [C.POINTER(PackEntry), i32, vp]
    L.ssad_conv_packed_filter_floats.restype = sz
    L.ssad_conv_packed_filter_floats.argtypes = [i32, i32]
    L.ssad_conv_pack_filter.argtypes = [vp, i32, i32, vp, vp, vp]
    L.ssad_conv3x3_forward.argtypes = [C.POINTER(ConvLevel), i32, vp, vp, i32, i32, i32, vp]
    L.ssad_conv_wino_filter_floats.restype = sz
    L.ssad_conv_wino_filter_floats.argtypes = [i32, i32]
    L.ssad_conv_wino_pack_filter.argtypes = [vp, i32, i32, vp, vp, vp]
    L.ssad_conv3x3_forward_wino.argtypes = [C.POINTER(ConvLevel), i32, vp, vp, i32, i32, i32, vp]
    L.ssad_conv3x3_forward_wino24.argtypes = [C.POINTER(ConvLevel), i32, vp, vp, i32, i32, i32, vp]
    L.ssad_conv_wino24_filter_floats.restype = sz
    L.ssad_conv_wino24_filter_floats.argtypes = [i32, i32]
    L.ssad_conv_wino24_pack_filters.argtypes = [C.POINTER(PackEntry), i32, vp]
    L.ssad_conv_split_filter_floats.restype = sz
    L.ssad_conv_split_filter_floats.argtypes = [i32, i32]
    L.ssad_conv_split_pack_filters.argtypes = [C.POINTER(PackEntry), i32, vp]
    L.ssad_conv3x3_split_workspace_bytes.restype = sz
    L.ssad_conv3x3_split_workspace_bytes.argtypes = [C.POINTER(ConvLevel), i32, i32]
    L.ssad_conv3x3_forward_split.argtypes = [C.POINTER(ConvLevel), i32, vp, vp, i32, i32, i32, vp, sz, vp, vp, vp]
    L.ssad_conv3x3_forward_wino_launches.argtypes = [C.POINTER(ConvLevel), i32]
    L.ssad_conv3x3_forward_wino_launches_for.argtypes = [C.POINTER(ConvLevel), i32, i32, i32, i32]
    L.ssad_conv_wino_split_tail.argtypes = [i32]
    L.ssad_conv3x3_wgrad_workspace_bytes.restype = sz
    L.ssad_conv3x3_wgrad_workspace_bytes.argtypes = [C.POINTER(ConvLevel), i32, i32, i32]
    L.ssad_conv3x3_wgrad.argtypes = [C.POINTER(ConvLevel), i32, vp, vp, i32, i32, i32, vp, sz, vp]
    L.ssad_conv3x3_wgrad_split_workspace_bytes.restype = sz
    L.ssad_conv3x3_wgrad_split_workspace_bytes.argtypes = [C.POINTER(ConvLevel), i32, i32, i32]
    L.ssad_conv3x3_wgrad_split.argtypes = [C.POINTER(ConvLevel), i32, vp, vp, i32, i32, i32, vp, sz, vp]
    L.ssad_conv1x1_gemm.argtypes = [C.POINTER(GemmConv), vp]
    L.ssad_conv1x1_gemm_split_workspace_bytes.restype = sz
    L.ssad_conv1x1_gemm_split_workspace_bytes.argtypes = [C.POINTER(GemmConv)]
    L.ssad_conv1x1_gemm_split.argtypes = [C.POINTER(GemmConv), vp, sz, vp]
    L.ssad_transpose_filter.argtypes = [vp, i32, i32, i32, vp, vp]
    L.ssad_conv1x1_wgrad_workspace_bytes.restype = sz
    L.ssad_conv1x1_wgrad_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.ssad_conv1x1_wgrad.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, vp, sz, vp]
    L.ssad_gemm_split_filter_floats.restype = sz
    L.ssad_gemm_split_filter_floats.argtypes = [i32, i32]
    L.ssad_gemm_split_pack_filters.argtypes = [C.POINTER(GemmPackEntry), i32, vp]
    L.ssad_conv1x1_gemm_split_amax.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    L.ssad_split_absmax.argtypes = [vp, C.c_longlong, vp, vp]
    L.ssad_split_absmax_levels.argtypes = [C.POINTER(ConvLevel), i32, i32, i32, vp, vp]
    L.ssad_conv3x3_wgrad_split_amax.argtypes = [C.POINTER(ConvLevel), i32, vp, vp, i32, i32, i32, vp, sz, vp, vp, vp]
    L.ssad_conv1x1_wgrad_split_amax.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, vp, sz, vp, vp, vp]
    L.ssad_conv1x1_wgrad_split_workspace_bytes.restype = sz
    L.ssad_conv1x1_wgrad_split_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.ssad_conv1x1_wgrad_split.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, vp, sz, vp]
    L.ssad_subsample.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
    L.ssad_subsample_grad.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, vp]
    L.ssad_conv_implicit_gemm.argtypes = [C.POINTER(GemmConv), i32, i32, i32, i32, i32, i32, vp]
    L.ssad_transpose_filters.argtypes = [C.POINTER(TransposeEntry), i32, vp]
    L.ssad_f16_pack_filters.argtypes = [C.POINTER(F16PackEntry), i32, vp]
    L.ssad_conv_implicit_gemm_workspace_bytes.restype = sz
    L.ssad_conv_implicit_gemm_workspace_bytes.argtypes = [i32] * 8
    L.ssad_conv_implicit_gemm_ws.argtypes = [C.POINTER(GemmConv), i32, i32, i32, i32, i32, i32, vp, sz, vp]
    L.ssad_conv_kxk_wgrad_workspace_bytes.restype = sz
    L.ssad_conv_kxk_wgrad_workspace_bytes.argtypes = [i32] * 8
    L.ssad_conv_kxk_wgrad.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, sz, vp]
    L.ssad_conv_kxk_dgrad_workspace_bytes.restype = sz
    L.ssad_conv_kxk_dgrad_workspace_bytes.argtypes = [i32] * 8
    L.ssad_conv_kxk_dgrad.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, sz, vp]
    L.ssad_grouped_conv3x3_forward.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    L.ssad_grouped_conv3x3_filter_floats.argtypes = [i32, i32]
    L.ssad_grouped_conv3x3_filter_floats.restype = C.c_longlong
    L.ssad_grouped_conv3x3_pack_filter.argtypes = [vp, i32, i32, vp, vp]
    L.ssad_conv1x1_f16.argtypes = [C.POINTER(PwF16), vp]
    L.ssad_pw_f16_filter_halves.restype = sz
    L.ssad_pw_f16_filter_halves.argtypes = [i32, i32]
    L.ssad_pw_f16_pack_filter.argtypes = [vp, i32, i32, vp, vp, vp]
    L.ssad_conv1x1_wgrad_f16_workspace_bytes.restype = sz
    L.ssad_conv1x1_wgrad_f16_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    L.ssad_conv1x1_wgrad_f16.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, sz, vp]
    L.ssad_f16_elementwise.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.ssad_f16_subsample.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
    L.ssad_stem_pool_f16.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    L.ssad_grouped_conv3x3_f16_filter_halves.restype = sz
    L.ssad_grouped_conv3x3_f16_filter_halves.argtypes = [i32, i32]
    L.ssad_grouped_conv3x3_f16_pack_filter.argtypes = [vp, i32, i32, vp, vp]
    L.ssad_grouped_conv3x3_f16.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
    L.ssad_kernels_arch.restype = C.c_char_p
    L.ssad_kernels_abi_version.restype = i32
    if L.ssad_kernels_abi_version() != ABI_VERSION:
        raise KernelError("%s has kernel ABI %d, this binding is written against %d (include/ssad_kernels.h): "
                          "rebuild csrc/" % (LIB_PATH, L.ssad_kernels_abi_version(), ABI_VERSION))
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        raise KernelError("%s failed with code %d" % (what, rc))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise KernelError("%s must be a contiguous float32 device tensor" % name)
    return t


_ws_cache = {}


ABI_VERSION = 3        # SSAD_KERNELS_ABI_VERSION of include/ssad_kernels.h


def _workspace(nbytes, tag):
    dev = torch.cuda.current_device()
    # one buffer per (device, stream, use): launches on different streams may overlap
    key = (dev, torch.cuda.current_stream().cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        # (zero-filled for tidiness only: the launchers that keep arrival counters here zero them)
        buf = torch.zeros(max(int(nbytes), 1), dtype=torch.uint8, device="cuda")
        _ws_cache[key] = buf
    return buf


# ---------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------

def _distill_levels(levels, outs):
    arr = (DistillLevel * len(levels))()
    for i, ((x, q, g), o) in enumerate(zip(levels, outs)):
        _f32c(x, "logits")
        if q is not None:
            _f32c(q, "teacher_prob")
        if g.dtype != torch.int32 or not g.is_contiguous() or not g.is_cuda:
            raise KernelError("labels must be a contiguous int32 device tensor")
        if x.dim() != 4 or (q is not None and q.shape != x.shape):
            raise KernelError("logits/teacher must be 4-D and equal-shaped")
        N, D, H, W = x.shape
        if g.dim() != 4 or g.shape[0] != N or tuple(g.shape[2:]) != (H, W) or g.shape[1] == 0 \
                or D % g.shape[1] != 0:
            raise KernelError("labels must be N x A x H x W with A dividing the logits' channels, "
                              "got %r for logits %r" % (tuple(g.shape), tuple(x.shape)))
        arr[i] = DistillLevel(x.data_ptr(), q.data_ptr() if q is not None else 0, g.data_ptr(),
                              o.data_ptr(), N, D, H, W)
    return arr


def distill_loss_forward(levels, normalizer, *, gamma=1.0, alpha=0.25, beta=0.0,
                         num_classes=80, ignored_label=-1, scale=1.0):
    """levels: list of (logits NxDxHxW, teacher_prob, labels NxAxHxW int32).
    Returns a float32 tensor [n_levels] of scalar losses."""
    L = lib()
    n = len(levels)
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    outs = [out[i:i + 1] for i in range(n)]
    arr = _distill_levels(levels, outs)
    P = DistillParams(gamma, alpha, beta, num_classes, ignored_label, scale)
    nb = L.ssad_distill_loss_workspace_bytes(n)
    ws = _workspace(nb, "distill")
    _check(L.ssad_distill_loss_forward(arr, n, _ptr(normalizer), C.byref(P),
                                       _ptr(ws), nb, _stream()), "distill_loss_forward")
    return out


def distill_loss_backward(levels, normalizer, dloss, *, gamma=1.0, alpha=0.25,
                          beta=0.0, num_classes=80, ignored_label=-1, scale=1.0,
                          out=None):
    """dloss: float32 device tensor with one value per level (or one shared).
    Returns the list of dX tensors."""
    L = lib()
    n = len(levels)
    outs = out if out is not None else [torch.empty_like(x) for (x, _, _) in levels]
    arr = _distill_levels(levels, outs)
    P = DistillParams(gamma, alpha, beta, num_classes, ignored_label, scale)
    stride = 1 if dloss.numel() >= n and n > 1 else 0
    _check(L.ssad_distill_loss_backward(arr, n, _ptr(normalizer), _ptr(dloss), stride,
                                        C.byref(P), _stream()), "distill_loss_backward")
    return outs


def focal_loss_forward(levels, fg_num, *, gamma=1.0, alpha=0.25, num_classes=80, scale=1.0):
    """SigmoidFocalLoss; levels: list of (logits, labels).  Returns [n_levels] losses."""
    L = lib()
    n = len(levels)
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    arr = _distill_levels([(x, None, g) for x, g in levels], [out[i:i + 1] for i in range(n)])
    P = FocalParams(gamma, alpha, num_classes, scale)
    nb = L.ssad_distill_loss_workspace_bytes(n)
    ws = _workspace(nb, "distill")
    _check(L.ssad_focal_loss_forward(arr, n, _ptr(fg_num), C.byref(P), _ptr(ws), nb, _stream()),
           "focal_loss_forward")
    return out


def focal_loss_backward(levels, fg_num, dloss, *, gamma=1.0, alpha=0.25, num_classes=80,
                        scale=1.0, out=None):
    L = lib()
    n = len(levels)
    outs = out if out is not None else [torch.empty_like(x) for x, _ in levels]
    arr = _distill_levels([(x, None, g) for x, g in levels], outs)
    P = FocalParams(gamma, alpha, num_classes, scale)
    stride = 1 if dloss.numel() >= n and n > 1 else 0
    _check(L.ssad_focal_loss_backward(arr, n, _ptr(fg_num), _ptr(dloss), stride, C.byref(P),
                                      _stream()), "focal_loss_backward")
    return outs


def cls_losses_fused(levels, normalizer, fg_num, distill_kw, focal_kw, out=None):
    """One pass over the logits: returns (distill_losses, focal_losses, dX list) where
    dX = d(distill)/dx + d(focal)/dx for loss gradients of 1.0."""
    L = lib()
    n = len(levels)
    dl = torch.empty(n, dtype=torch.float32, device="cuda")
    fl = torch.empty(n, dtype=torch.float32, device="cuda")
    outs = out if out is not None else [torch.empty_like(x) for x, _, _ in levels]
    arr = _distill_levels(levels, outs)
    DP = DistillParams(distill_kw["gamma"], distill_kw["alpha"], distill_kw.get("beta", 0.0),
                       distill_kw["num_classes"], distill_kw.get("ignored_label", -1),
                       distill_kw.get("scale", 1.0))
    FP = FocalParams(focal_kw["gamma"], focal_kw["alpha"], focal_kw["num_classes"],
                     focal_kw.get("scale", 1.0))
    nb = L.ssad_cls_losses_fused_workspace_bytes(n)
    ws = _workspace(nb, "clsfused")
    _check(L.ssad_cls_losses_fused(arr, n, _ptr(normalizer), _ptr(fg_num), C.byref(DP),
                                   C.byref(FP), _ptr(dl), _ptr(fl), _ptr(ws), nb, _stream()),
           "cls_losses_fused")
    return dl, fl, outs


def select_smooth_l1_forward(Y_hat, Y, Lc, S, *, beta=1.0, scale=1.0):
    """Y [M,4], Lc [M,4] float rows (n, c, y, x), S device scalar.  Returns a scalar tensor."""
    out = torch.zeros((), dtype=torch.float32, device="cuda")
    M = Y.shape[0] if Y.numel() else 0
    if M == 0:
        return out
    N, D, H, W = Y_hat.shape
    nb = lib().ssad_select_smooth_l1_workspace_bytes(1)
    ws = _workspace(nb, "smoothl1")
    _check(lib().ssad_select_smooth_l1_forward(
        _ptr(_f32c(Y_hat, "Y_hat")), _ptr(_f32c(Y, "Y")), _ptr(_f32c(Lc, "L")), _ptr(S), N, D, H,
        W, M, beta, scale, _ptr(out), _ptr(ws), nb, _stream()), "select_smooth_l1_forward")
    return out


def smooth_l1_levels_array(preds, targets, losses, d_preds):
    """ctypes level table of ssad_select_smooth_l1_levels: preds [N,D,H,W] per level, targets
    [(Y [M,4], L [M,4])], losses a float32 [n_levels] tensor (or None), d_preds per-level
    gradients (or None)."""
    n = len(preds)
    arr = (SmoothL1Level * n)()
    for i, (p, (Y, Lc)) in enumerate(zip(preds, targets)):
        _f32c(p, "Y_hat")
        M = Y.shape[0] if Y.numel() else 0
        if M:
            _f32c(Y, "Y"); _f32c(Lc, "L")
            if tuple(Lc.shape) != (M, 4) or tuple(Y.shape) != (M, 4):
                raise KernelError("Y and L must be [M, 4]")
        N, D, H, W = p.shape
        if d_preds is not None and d_preds[i].shape != p.shape:
            raise KernelError("gradient must have the prediction's shape")
        arr[i] = SmoothL1Level(p.data_ptr(), Y.data_ptr() if M else 0, Lc.data_ptr() if M else 0,
                               losses[i:i + 1].data_ptr() if losses is not None else 0,
                               _f32c(d_preds[i], "dY_hat").data_ptr() if d_preds is not None else 0,
                               N, D, H, W, M)
    return arr


def select_smooth_l1_levels(preds, targets, S, dloss=None, *, beta=1.0, scale=1.0, losses=None,
                            d_preds=None):
    """SelectSmoothL1Loss of every FPN level in one launch sequence: losses [n_levels] and, with
    dloss (device scalar), the full gradients d_preds (zero filled here)."""
    n = len(preds)
    if losses is None:
        losses = torch.empty(n, dtype=torch.float32, device="cuda")
    if dloss is not None and d_preds is None:
        d_preds = [torch.empty_like(p) for p in preds]
    arr = smooth_l1_levels_array(preds, targets, losses, d_preds if dloss is not None else None)
    nb = lib().ssad_select_smooth_l1_workspace_bytes(n)
    ws = _workspace(nb, "smoothl1")
    _check(lib().ssad_select_smooth_l1_levels(arr, n, _ptr(S), _ptr(dloss), beta, scale, 1, _ptr(ws), nb,
                                              _stream()), "select_smooth_l1_levels")
    return losses, d_preds


def select_smooth_l1_backward(Y_hat, Y, Lc, S, dloss, *, beta=1.0, scale=1.0, out=None):
    dy = out if out is not None else torch.empty_like(Y_hat)
    _check(lib().ssad_fill(_ptr(dy), 0.0, dy.numel(), _stream()), "fill")
    M = Y.shape[0] if Y.numel() else 0
    if M:
        N, D, H, W = Y_hat.shape
        _check(lib().ssad_select_smooth_l1_backward(
            _ptr(_f32c(Y_hat, "Y_hat")), _ptr(_f32c(Y, "Y")), _ptr(_f32c(Lc, "L")), _ptr(S),
            _ptr(dloss), N, D, H, W, M, beta, scale, _ptr(dy), _stream()),
            "select_smooth_l1_backward")
    return dy


def pow_sum(inputs, power=1.0):
    L = lib()
    n = len(inputs)
    for t in inputs:
        _f32c(t, "PowSum input")
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in inputs])
    sizes = (C.c_int64 * n)(*[t.numel() for t in inputs])
    out = torch.empty((), dtype=torch.float32, device="cuda")
    nb = L.ssad_pow_sum_workspace_bytes(n)
    ws = _workspace(nb, "powsum")
    _check(L.ssad_pow_sum(ptrs, sizes, n, power, _ptr(out), _ptr(ws), nb, _stream()),
           "pow_sum")
    return out


# ---------------------------------------------------------------------------
# elementwise
# ---------------------------------------------------------------------------

def relu_(x):
    _check(lib().ssad_relu(_ptr(x), _ptr(x), x.numel(), _stream()), "relu")
    return x


def relu(x):
    y = torch.empty_like(x)
    _check(lib().ssad_relu(_ptr(_f32c(x, "x")), _ptr(y), x.numel(), _stream()), "relu")
    return y


def relu_grad(y, dy, out=None):
    dx = out if out is not None else torch.empty_like(dy)
    _check(lib().ssad_relu_grad(_ptr(_f32c(y, "y")), _ptr(_f32c(dy, "dy")), _ptr(dx),
                                y.numel(), _stream()), "relu_grad")
    return dx


def conv1x1_bias_act(x, w, bias=None, residual=None, relu=True):
    """act(conv1x1(x, w) + bias[m] (+ residual)) in one pass (NCHW float32)."""
    _f32c(x, "x")
    N, Cc, H, W = x.shape
    M = w.shape[0]
    w2 = _f32c(w.reshape(M, Cc), "w")
    y = torch.empty((N, M, H, W), dtype=torch.float32, device="cuda")
    if bias is not None:
        _f32c(bias, "bias")
    if residual is not None:
        _f32c(residual, "residual")
    _check(lib().ssad_conv1x1_bias_act(_ptr(x), _ptr(w2), _ptr(bias), _ptr(residual), _ptr(y), N, Cc, H * W, M,
                                       int(relu), _stream()), "conv1x1_bias_act")
    return y


def conv1x1_bias_act2(x1, x2, w12, bias=None, relu=True):
    """act([W1 | W2] . [x1 ; x2] + bias): two pointwise convolutions into one output (a
    bottleneck's last layer + its projection shortcut); w12 is [M][C1 + C2]."""
    _f32c(x1, "x1"); _f32c(x2, "x2"); _f32c(w12, "w12")
    N, C1, H, W = x1.shape
    C2, M = x2.shape[1], w12.shape[0]
    assert x2.shape[0] == N and x2.shape[2:] == x1.shape[2:] and w12.shape[1] == C1 + C2
    y = torch.empty((N, M, H, W), dtype=torch.float32, device="cuda")
    if bias is not None:
        _f32c(bias, "bias")
    _check(lib().ssad_conv1x1_bias_act2(_ptr(x1), C1, _ptr(x2), C2, _ptr(w12), _ptr(bias), _ptr(None), _ptr(y),
                                        N, H * W, M, int(relu), _stream()), "conv1x1_bias_act2")
    return y


def relu_grad_rowsum(y, dy, want_dx=True):
    """(dx, rowsum[N][C]): dx = y > 0 ? dy : 0 (y None: dx = dy itself) and the plane sums of dx
    in the same pass; a bias gradient is rowsum.sum(0)."""
    _f32c(dy, "dy")
    N, Cc = dy.shape[0], dy.shape[1]
    hw = dy.numel() // max(N * Cc, 1)
    dx = torch.empty_like(dy) if (y is not None and want_dx) else None
    rs = torch.empty((N, Cc), dtype=torch.float32, device="cuda")
    _check(lib().ssad_relu_grad_rowsum(_ptr(_f32c(y, "y")) if y is not None else _ptr(None), _ptr(dy),
                                       _ptr(dx), _ptr(rs), N, Cc, hw, _stream()), "relu_grad_rowsum")
    return (dx if y is not None else dy), rs


def affine_channel_(x, bias, scale=None, residual=None, relu=False):
    """In place: x[n,c,:,:] = act(x * scale[c] + bias[c] + residual) (AffineChannel
    + the bottleneck's residual Sum + Relu in one pass over an NCHW tensor)."""
    _f32c(x, "x")
    N, Cc = x.shape[0], x.shape[1]
    hw = x.numel() // max(N * Cc, 1)
    if residual is not None:
        _f32c(residual, "residual")
    _check(lib().ssad_affine_channel(_ptr(x), _ptr(scale), _ptr(bias), _ptr(residual), _ptr(x),
                                     N, Cc, hw, int(relu), _stream()), "affine_channel")
    return x


def upsample_nearest(x, scale=2, addend=None, out=None):
    """Nearest-neighbour upsampling of an NCHW tensor, optionally + addend (FPN's
    lateral Sum folded in; `out` may be `addend` itself)."""
    _f32c(x, "x")
    N, Cc, H, W = x.shape
    y = out if out is not None else torch.empty((N, Cc, H * scale, W * scale), dtype=torch.float32,
                                                device="cuda")
    if addend is not None:
        _f32c(addend, "addend")
    _check(lib().ssad_upsample_nearest(_ptr(x), _ptr(addend), _ptr(y), N, Cc, H, W, scale,
                                       _stream()), "upsample_nearest")
    return y


def max_pool3x3s2_bias_relu(x, bias=None, relu=True):
    """relu(maxpool3x3/2,pad1(x) + bias[c]) in one pass (the ResNet stem's tail)."""
    _f32c(x, "x")
    N, Cc, H, W = x.shape
    y = torch.empty((N, Cc, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device="cuda")
    if bias is not None:
        _f32c(bias, "bias")
    _check(lib().ssad_max_pool3x3s2_bias_relu(_ptr(x), _ptr(bias), N, Cc, H, W, int(relu), _ptr(y),
                                              _stream()), "max_pool3x3s2_bias_relu")
    return y


def upsample_nearest_grad(dy, scale=2):
    _f32c(dy, "dy")
    N, Cc, OH, OW = dy.shape
    dx = torch.empty((N, Cc, OH // scale, OW // scale), dtype=torch.float32, device="cuda")
    _check(lib().ssad_upsample_nearest_grad(_ptr(dy), _ptr(dx), N, Cc, OH // scale, OW // scale,
                                            scale, _stream()), "upsample_nearest_grad")
    return dx


def sigmoid(x, out=None):
    y = out if out is not None else torch.empty_like(x)
    _check(lib().ssad_sigmoid(_ptr(_f32c(x, "x")), _ptr(y), x.numel(), _stream()), "sigmoid")
    return y


def sum_n(inputs, out=None):
    n = len(inputs)
    o = out if out is not None else torch.empty_like(inputs[0])
    ptrs = (C.c_void_p * n)(*[_f32c(t, "sum input").data_ptr() for t in inputs])
    _check(lib().ssad_sum_n(ptrs, n, _ptr(o), o.numel(), _stream()), "sum_n")
    return o


def scale_(x, alpha):
    _check(lib().ssad_scale(_ptr(x), _ptr(x), alpha, x.numel(), _stream()), "scale")
    return x


def momentum_sgd_update_(w, g, m, lr, momentum=0.9, weight_decay=1e-4, is_bias=False):
    """In place on w, g, m; lr is a float32 device scalar tensor."""
    _check(lib().ssad_momentum_sgd_update(
        _ptr(_f32c(w, "w")), _ptr(_f32c(g, "g")), _ptr(_f32c(m, "m")), _ptr(lr),
        momentum, weight_decay, int(is_bias), w.numel(), _stream()), "momentum_sgd")


# ---------------------------------------------------------------------------
# conv 3x3
# ---------------------------------------------------------------------------

def conv_pack_filter(w, want_fwd=True, want_dgrad=True):
    """w: Cout x Cin x 3 x 3.  Returns (packed_fwd, packed_dgrad)."""
    L = lib()
    _f32c(w, "filter")
    Cout, Cin, kh, kw = w.shape
    if (kh, kw) != (3, 3):
        raise KernelError("only 3x3 filters")
    pf = torch.empty(L.ssad_conv_packed_filter_floats(Cout, Cin), dtype=torch.float32,
                     device="cuda") if want_fwd else None
    pd = torch.empty(L.ssad_conv_packed_filter_floats(Cin, Cout), dtype=torch.float32,
                     device="cuda") if want_dgrad else None
    _check(L.ssad_conv_pack_filter(_ptr(w), Cout, Cin, _ptr(pf), _ptr(pd), _stream()),
           "conv_pack_filter")
    return pf, pd


def conv_wino_pack_filter(w, want_fwd=True, want_dgrad=True):
    """Winograd-domain filters U = G g G^T in MFMA operand order: (fwd, dgrad)."""
    L = lib()
    _f32c(w, "filter")
    Cout, Cin, kh, kw = w.shape
    if (kh, kw) != (3, 3):
        raise KernelError("only 3x3 filters")
    pf = torch.empty(L.ssad_conv_wino_filter_floats(Cout, Cin), dtype=torch.float32,
                     device="cuda") if want_fwd else None
    pd = torch.empty(L.ssad_conv_wino_filter_floats(Cin, Cout), dtype=torch.float32,
                     device="cuda") if want_dgrad else None
    _check(L.ssad_conv_wino_pack_filter(_ptr(w), Cout, Cin, _ptr(pf), _ptr(pd), _stream()),
           "conv_wino_pack_filter")
    return pf, pd


def conv_wino24_pack_filter(w, want_dgrad=False):
    """F(2x4, 3x3) filter pack U = G2 g G4^T; want_dgrad: -> (forward, data-gradient) packs (the latter of the
    flipped, transposed filter, Cin x Cout)."""
    L = lib()
    _f32c(w, "filter")
    Cout, Cin, kh, kw = w.shape
    if (kh, kw) != (3, 3):
        raise KernelError("only 3x3 filters")
    pf = torch.empty(L.ssad_conv_wino24_filter_floats(Cout, Cin), dtype=torch.float32, device="cuda")
    pd = torch.empty(L.ssad_conv_wino24_filter_floats(Cin, Cout), dtype=torch.float32,
                     device="cuda") if want_dgrad else None
    tab = (PackEntry * 1)(PackEntry(w.data_ptr(), Cout, Cin, pf.data_ptr(), pd.data_ptr() if want_dgrad else 0))
    _check(L.ssad_conv_wino24_pack_filters(tab, 1, _stream()), "conv_wino24_pack_filters")
    return (pf, pd) if want_dgrad else pf


def conv_split_pack_filter(w, want_dgrad=False):
    """Split-operand (3 x fp16 MFMA) filter pack: header with the filter's |max| + hi / lo fp16 planes; want_dgrad:
    -> (forward, data-gradient) packs (conv3x3_split.hip)."""
    L = lib()
    _f32c(w, "filter")
    Cout, Cin, kh, kw = w.shape
    if (kh, kw) != (3, 3):
        raise KernelError("only 3x3 filters")
    pf = torch.empty(L.ssad_conv_split_filter_floats(Cout, Cin), dtype=torch.float32, device="cuda")
    pd = torch.empty(L.ssad_conv_split_filter_floats(Cin, Cout), dtype=torch.float32,
                     device="cuda") if want_dgrad else None
    tab = (PackEntry * 1)(PackEntry(w.data_ptr(), Cout, Cin, pf.data_ptr(), pd.data_ptr() if want_dgrad else 0))
    _check(L.ssad_conv_split_pack_filters(tab, 1, _stream()), "conv_split_pack_filters")
    return (pf, pd) if want_dgrad else pf


def conv3x3_forward_split(xs, packed, bias, Cout, *, relu=False, sigmoid=False, out=None, mask_by=None, workspace=None,
                          amax_in=None, amax_out=None):
    """conv3x3_forward on the split-operand engine (fp32 operands as hi + lo fp16, three fp16 MFMAs per pair, fp32
    accumulation; xs: levels sharing the filter).  mask_by: the data-gradient form, as conv3x3_forward_wino24."""
    L = lib()
    for x in xs:
        _f32c(x, "conv input")
    Cin = xs[0].shape[1]
    ys = out if out is not None else [
        torch.empty((x.shape[0], Cout, x.shape[2], x.shape[3]), dtype=torch.float32, device="cuda") for x in xs]
    flags = (CONV_RELU if relu else 0) | (CONV_SIGMOID if sigmoid else 0) | (CONV_MASK_AUX if mask_by is not None else 0)
    arr = _conv_levels(xs, ys, mask_by)
    need = L.ssad_conv3x3_split_workspace_bytes(arr, len(xs), Cin)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
    _check(L.ssad_conv3x3_forward_split(arr, len(xs), _ptr(packed), _ptr(bias), Cout, Cin, flags, _ptr(workspace),
                                        workspace.numel(), _ptr(amax_in), _ptr(amax_out), _stream()),
           "conv3x3_forward_split")
    return ys


def conv3x3_forward_wino24(xs, packed, bias, Cout, *, relu=False, sigmoid=False, out=None, mask_by=None):
    """conv3x3_forward on the F(2x4, 3x3) engine (xs: levels sharing the filter).  mask_by: the data-gradient form --
    y = mask > 0 ? y : 0 (fused ReluGradient), packed = the data-gradient pack, Cout = the layer's input channels."""
    L = lib()
    for x in xs:
        _f32c(x, "conv input")
    Cin = xs[0].shape[1]
    ys = out if out is not None else [
        torch.empty((x.shape[0], Cout, x.shape[2], x.shape[3]), dtype=torch.float32, device="cuda") for x in xs]
    flags = (CONV_RELU if relu else 0) | (CONV_SIGMOID if sigmoid else 0) | (CONV_MASK_AUX if mask_by is not None else 0)
    arr = _conv_levels(xs, ys, mask_by)
    _check(L.ssad_conv3x3_forward_wino24(arr, len(xs), _ptr(packed), _ptr(bias), Cout, Cin, flags, _stream()),
           "conv3x3_forward_wino24")
    return ys


def _conv_levels(xs, ys, auxs, packs=None, biases=None):
    n = len(xs)
    arr = (ConvLevel * n)()
    for i in range(n):
        N, _, H, W = xs[i].shape
        arr[i] = ConvLevel(xs[i].data_ptr(),
                           ys[i].data_ptr() if ys is not None and ys[i] is not None else 0,
                           auxs[i].data_ptr() if auxs is not None and auxs[i] is not None else 0,
                           N, H, W,
                           packs[i].data_ptr() if packs is not None else 0,
                           biases[i].data_ptr() if biases is not None and biases[i] is not None else 0)
    return arr


def conv3x3_forward(xs, packed, bias, Cout, *, relu=False, sigmoid=False, mask_by=None,
                    out=None, wino=False):
    """xs: list of N x Cin x H x W tensors (FPN levels sharing the filter).
    Returns the list of N x Cout x H x W outputs (one launch for all levels)."""
    L = lib()
    for x in xs:
        _f32c(x, "conv input")
    Cin = xs[0].shape[1]
    ys = out if out is not None else [
        torch.empty((x.shape[0], Cout, x.shape[2], x.shape[3]), dtype=torch.float32,
                    device="cuda") for x in xs]
    flags = ((CONV_RELU if relu else 0) | (CONV_MASK_AUX if mask_by is not None else 0)
             | (CONV_SIGMOID if sigmoid else 0))
    arr = _conv_levels(xs, ys, mask_by)
    fn = L.ssad_conv3x3_forward_wino if wino else L.ssad_conv3x3_forward
    _check(fn(arr, len(xs), _ptr(packed), _ptr(bias), Cout, Cin, flags, _stream()),
           "conv3x3_forward")
    return ys


def conv3x3_forward_multi(problems, Cout, *, relu=False, sigmoid=False, wino=False):
    """Independent convolutions of equal (Cout, Cin) in ONE launch.
    problems: list of dicts {xs, packed, bias, out, mask_by(optional)} -- e.g. the
    cls-tower and bbox-tower layer of the same depth, for teacher and student."""
    L = lib()
    xs, ys, auxs, packs, biases = [], [], [], [], []
    for p in problems:
        n = len(p["xs"])
        xs += list(p["xs"]); ys += list(p["out"])
        auxs += list(p["mask_by"]) if p.get("mask_by") is not None else [None] * n
        packs += [p["packed"]] * n
        biases += [p.get("bias")] * n
    masked = [a is not None for a in auxs]
    if any(masked) and not all(masked):
        raise KernelError("either every problem of a launch is masked or none")
    Cin = xs[0].shape[1]
    flags = ((CONV_RELU if relu else 0) | (CONV_MASK_AUX if all(masked) and masked else 0)
             | (CONV_SIGMOID if sigmoid else 0))
    arr = _conv_levels(xs, ys, auxs, packs, biases)
    fn = L.ssad_conv3x3_forward_wino if wino else L.ssad_conv3x3_forward
    _check(fn(arr, len(xs), None, None, Cout, Cin, flags, _stream()), "conv3x3_forward_multi")
    return ys


def conv3x3_wgrad(xs, dys, Cout, *, want_db=True, accumulate=False, dW=None, db=None, split=False, x_amax=None,
                  dy_amax=None):
    """dW[Cout,Cin,3,3] and db[Cout] summed over all levels and images.  split: the split-operand engine
    (ssad_conv3x3_wgrad_split) instead of the exact-fp32 ones."""
    L = lib()
    Cin = xs[0].shape[1]
    for x, d in zip(xs, dys):
        _f32c(x, "x"); _f32c(d, "dy")
    arr = _conv_levels(xs, None, dys)
    size_fn = L.ssad_conv3x3_wgrad_split_workspace_bytes if split else L.ssad_conv3x3_wgrad_workspace_bytes
    fn = L.ssad_conv3x3_wgrad_split if split else L.ssad_conv3x3_wgrad
    nb = size_fn(arr, len(xs), Cout, Cin)
    ws = _workspace(nb, "wgrad")
    if dW is None:
        dW = torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device="cuda")
    if db is None and want_db:
        db = torch.empty(Cout, dtype=torch.float32, device="cuda")
    if split and x_amax is not None:
        _check(L.ssad_conv3x3_wgrad_split_amax(arr, len(xs), _ptr(dW), _ptr(db) if want_db else None, Cout, Cin,
                                               int(accumulate), _ptr(ws), nb, _ptr(x_amax), _ptr(dy_amax), _stream()),
               "conv3x3_wgrad_split_amax")
        return dW, db
    _check(fn(arr, len(xs), _ptr(dW), _ptr(db) if want_db else None, Cout,
              Cin, int(accumulate), _ptr(ws), nb, _stream()),
           "conv3x3_wgrad_split" if split else "conv3x3_wgrad")
    return dW, db


# ---- fp16 storage / fp32 accumulation (config 5's precision) ---------------------------

F16_OUT_NCHW_F32 = 16
PW_F16_RES_UPSAMPLE2 = 32
# ssad_f16_elementwise modes
EW_SUBSAMPLE, EW_SUBSAMPLE_GRAD, EW_UPSAMPLE_GRAD, EW_SUM2, EW_RELU, EW_RELU_GRAD = 0, 1, 2, 3, 4, 5


def f16_pack_activations(x, scale=1.0, out=None):
    """NCHW float32 -> channel-blocked float16 [N][ceil(C/8)][H][W][8] of x * scale (zero
    padded tail)."""
    _f32c(x, "x")
    N, Cc, H, W = x.shape
    xb = out if out is not None else torch.empty((N, (Cc + 7) // 8, H, W, 8), dtype=torch.float16,
                                                 device="cuda")
    _check(lib().ssad_f16_pack_activations(_ptr(x), N, Cc, H, W, float(scale), _ptr(xb), _stream()),
           "f16_pack_activations")
    return xb


def f16_unpack_activations(xb, channels, scale=1.0, out=None):
    """channel-blocked float16 -> NCHW float32 (times scale) with `channels` channels."""
    N, CB, H, W, _ = xb.shape
    assert xb.dtype == torch.float16 and xb.is_contiguous() and CB == (channels + 7) // 8
    x = out if out is not None else torch.empty((N, channels, H, W), dtype=torch.float32, device="cuda")
    _check(lib().ssad_f16_unpack_activations(_ptr(xb), N, channels, H, W, float(scale), _ptr(x), _stream()),
           "f16_unpack_activations")
    return x


def f16_pack_filter(w, fwd=True, dgrad=False):
    """[M][C][3][3] float32 -> (packed_fwd, packed_dgrad) float16 (None when not asked for)."""
    _f32c(w, "w")
    M, Cc = w.shape[0], w.shape[1]
    n = lib().ssad_f16_filter_halves(M, Cc)
    wf = torch.empty(n, dtype=torch.float16, device="cuda") if fwd else None
    wd = torch.empty(n, dtype=torch.float16, device="cuda") if dgrad else None
    _check(lib().ssad_f16_pack_filter(_ptr(w), M, Cc, _ptr(wf), _ptr(wd), _stream()), "f16_pack_filter")
    return wf, wd


def conv3x3_forward_f16(xb, packed, bias, Cin, Cout, *, relu=False, sigmoid=False, mask_by=None,
                        out_nchw_f32=False, out=None):
    """3x3 / stride 1 / pad 1 convolution of a channel-blocked fp16 tensor; fp32 accumulation.
    Output: blocked fp16 [N][Cout/8][H][W][8], or NCHW float32 for the prediction layers.
    mask_by (blocked fp16 like the output): y = mask_by > 0 ? y : 0 (fused ReluGradient)."""
    N, CB, H, W, _ = xb.shape
    assert xb.dtype == torch.float16 and xb.is_contiguous() and CB == (Cin + 7) // 8
    if out is not None:
        y = out
    elif out_nchw_f32:
        y = torch.empty((N, Cout, H, W), dtype=torch.float32, device="cuda")
    else:
        y = torch.empty((N, Cout // 8, H, W, 8), dtype=torch.float16, device="cuda")
    if bias is not None:
        _f32c(bias, "bias")
    if mask_by is not None:
        assert mask_by.dtype == torch.float16 and mask_by.is_contiguous() and mask_by.shape == y.shape
    flags = ((CONV_RELU if relu else 0) | (CONV_SIGMOID if sigmoid else 0)
             | (CONV_MASK_AUX if mask_by is not None else 0) | (F16_OUT_NCHW_F32 if out_nchw_f32 else 0))
    _check(lib().ssad_conv3x3_forward_f16(_ptr(xb), _ptr(packed), _ptr(bias), _ptr(mask_by), N, Cin, H, W,
                                          Cout, flags, _ptr(y), _stream()), "conv3x3_forward_f16")
    return y


def conv3x3_forward_f16_levels(xbs, packed, bias, Cin, Cout, outs, *, relu=False, sigmoid=False,
                               mask_bys=None, out_nchw_f32=False):
    """conv3x3_forward_f16 for every FPN level sharing the filter in one launch; `outs`
    (and `mask_bys`) are per-level lists."""
    n = len(xbs)
    arr = (F16Level * n)()
    for i, (xb, y) in enumerate(zip(xbs, outs)):
        N, CB, H, W, _ = xb.shape
        assert xb.dtype == torch.float16 and xb.is_contiguous() and CB == (Cin + 7) // 8
        assert y.is_contiguous() and y.dtype == (torch.float32 if out_nchw_f32 else torch.float16)
        arr[i].x, arr[i].y = xb.data_ptr(), y.data_ptr()
        arr[i].aux = mask_bys[i].data_ptr() if mask_bys is not None else None
        arr[i].N, arr[i].H, arr[i].W = N, H, W
    if bias is not None:
        _f32c(bias, "bias")
    flags = ((CONV_RELU if relu else 0) | (CONV_SIGMOID if sigmoid else 0)
             | (CONV_MASK_AUX if mask_bys is not None else 0) | (F16_OUT_NCHW_F32 if out_nchw_f32 else 0))
    _check(lib().ssad_conv3x3_forward_f16_levels(arr, n, _ptr(packed), _ptr(bias), Cin, Cout, flags,
                                                 _stream()), "conv3x3_forward_f16_levels")
    return outs


def conv3x3_forward_f16_multi(problems, Cin, Cout, *, relu=False, sigmoid=False, out_nchw_f32=False):
    """Independent fp16 convolutions of equal (Cin, Cout) in ONE launch.  problems: dicts
    {xs, packed, bias, out, mask_by(optional, all or none)} -- e.g. the cls- and bbox-tower layer
    of the same depth for teacher and student."""
    n = sum(len(p["xs"]) for p in problems)
    arr = (F16Level * n)()
    masked = [p.get("mask_by") is not None for p in problems]
    assert all(masked) or not any(masked)
    i = 0
    for p in problems:
        if p.get("bias") is not None:
            _f32c(p["bias"], "bias")
        for l, (xb, y) in enumerate(zip(p["xs"], p["out"])):
            N, CB, H, W, _ = xb.shape
            assert xb.dtype == torch.float16 and xb.is_contiguous() and CB == (Cin + 7) // 8
            assert y.is_contiguous() and y.dtype == (torch.float32 if out_nchw_f32 else torch.float16)
            arr[i].x, arr[i].y = xb.data_ptr(), y.data_ptr()
            arr[i].aux = p["mask_by"][l].data_ptr() if masked[0] else None
            arr[i].N, arr[i].H, arr[i].W = N, H, W
            arr[i].packed = p["packed"].data_ptr()
            arr[i].bias = p["bias"].data_ptr() if p.get("bias") is not None else None
            i += 1
    flags = ((CONV_RELU if relu else 0) | (CONV_SIGMOID if sigmoid else 0)
             | (CONV_MASK_AUX if masked[0] else 0) | (F16_OUT_NCHW_F32 if out_nchw_f32 else 0))
    _check(lib().ssad_conv3x3_forward_f16_levels(arr, n, None, None, Cin, Cout, flags, _stream()),
           "conv3x3_forward_f16_multi")


def conv3x3_wgrad_f16(xbs, dybs, Cin, Cout, *, scale=1.0, dW=None, db=None, bias_grad=True):
    """dW [Cout][Cin][3][3] and db [Cout] (float32, times scale) summed over the given levels;
    xbs / dybs are lists of channel-blocked fp16 tensors (one per FPN level sharing the filter)."""
    if dW is None:
        dW = torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device="cuda")
    if db is None and bias_grad:
        db = torch.empty((Cout,), dtype=torch.float32, device="cuda")
    L = lib()
    n = len(xbs)
    arr = (F16WgradLevel * n)()
    for i, (xb, dyb) in enumerate(zip(xbs, dybs)):
        N, CB, H, W, _ = xb.shape
        assert xb.dtype == torch.float16 and dyb.dtype == torch.float16
        assert xb.is_contiguous() and dyb.is_contiguous()
        assert CB == (Cin + 7) // 8 and dyb.shape == (N, (Cout + 7) // 8, H, W, 8)
        arr[i].x, arr[i].dy, arr[i].N, arr[i].H, arr[i].W = xb.data_ptr(), dyb.data_ptr(), N, H, W
    nbytes = L.ssad_conv3x3_wgrad_f16_levels_workspace_bytes(arr, n, Cin, Cout)
    ws = _workspace(nbytes, "wgrad_f16")
    _check(L.ssad_conv3x3_wgrad_f16_levels(arr, n, Cin, Cout, 0, float(scale), _ptr(dW), _ptr(db), _ptr(ws),
                                           nbytes, _stream()), "conv3x3_wgrad_f16_levels")
    return dW, db


# ---- pointwise convolution as an fp32-MFMA GEMM (row f1) -------------------------------------

GEMM_RELU = 1
GEMM_ACCUMULATE = 8


def transpose_filter(w, out=None):
    """[M][K](x1x1) -> W^T [K][ldm] with ldm = M rounded up to 4 (zero padded): the forward's A operand."""
    M, Kc = w.shape[0], w.shape[1]
    w2 = _f32c(w.reshape(M, Kc), "w")
    ldm = (M + 3) // 4 * 4
    wt = out if out is not None else torch.empty((Kc, ldm), dtype=torch.float32, device="cuda")
    _check(lib().ssad_transpose_filter(_ptr(w2), M, Kc, ldm, _ptr(wt), _stream()), "transpose_filter")
    return wt


def gemm_conv_desc(a, lda, x, y, K, M, bias=None, residual=None, mask=None, relu=False, accumulate=False):
    N = x.shape[0]
    P = x.numel() // max(N * K, 1)
    return GemmConv(a.data_ptr(), x.data_ptr(), y.data_ptr(), bias.data_ptr() if bias is not None else 0,
                    residual.data_ptr() if residual is not None else 0,
                    mask.data_ptr() if mask is not None else 0, lda, N, K, P, M,
                    (GEMM_RELU if relu else 0) | (GEMM_ACCUMULATE if accumulate else 0))


def _run_gemm(d, what, split):
    """ssad_conv1x1_gemm, or (split) the same descriptor on the split-operand engine of gemm_split.hip"""
    L = lib()
    if not split:
        _check(L.ssad_conv1x1_gemm(C.byref(d), _stream()), what)
        return
    need = L.ssad_conv1x1_gemm_split_workspace_bytes(C.byref(d))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
    _check(L.ssad_conv1x1_gemm_split(C.byref(d), _ptr(ws), ws.numel(), _stream()), what + " (split)")


def split_absmax(x, word=None):
    """|max| of one tensor as a float bit pattern in a device word (ssad_split_absmax); word: a zeroed int32[1]."""
    _f32c(x, "x")
    if word is None:
        word = torch.zeros(1, dtype=torch.int32, device="cuda")
    _check(lib().ssad_split_absmax(_ptr(x), x.numel(), _ptr(word), _stream()), "split_absmax")
    return word


def split_absmax_levels(xs, words=None):
    """|max| word per tensor of a list (one launch, ssad_split_absmax_levels)."""
    if words is None:
        words = torch.zeros(len(xs), dtype=torch.int32, device="cuda")
    arr = _conv_levels(xs, None, None)
    _check(lib().ssad_split_absmax_levels(arr, len(xs), xs[0].shape[1], 0, _ptr(words), _stream()), "split_absmax_levels")
    return words


def gemm_split_pack_filter(a, lda, Kc, M):
    """The split copy of a pointwise filter operand a[Kc][lda] (ssad_gemm_split_pack_filters, one entry)."""
    dst = torch.empty(lib().ssad_gemm_split_filter_floats(Kc, M), dtype=torch.float32, device="cuda")
    tab = (GemmPackEntry * 1)(GemmPackEntry(a.data_ptr(), dst.data_ptr(), lda, Kc, M))
    _check(lib().ssad_gemm_split_pack_filters(tab, 1, _stream()), "gemm_split_pack_filters")
    return dst


def conv1x1_forward_split_amax(x, wt, M, bias=None, residual=None, relu=False, packed_a=None, x_amax=None, y_amax=None):
    """conv1x1_forward on the split-operand engine with the filter split / x's |max| word handed over and (y_amax: a
    zeroed int32[1]) the |max| of the result folded into a word (ssad_conv1x1_gemm_split_amax)."""
    _f32c(x, "x"); _f32c(wt, "wt")
    N, Kc, H, W = x.shape
    y = torch.empty((N, M, H, W), dtype=torch.float32, device="cuda")
    d = gemm_conv_desc(wt, wt.shape[1], x, y, Kc, M, bias, residual, None, relu, False)
    L = lib()
    need = L.ssad_conv1x1_gemm_split_workspace_bytes(C.byref(d))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
    _check(L.ssad_conv1x1_gemm_split_amax(C.byref(d), _ptr(packed_a), _ptr(x_amax), _ptr(y_amax), _ptr(ws), ws.numel(),
                                          _stream()), "conv1x1_gemm_split_amax")
    return y


def conv1x1_forward(x, wt, M, bias=None, residual=None, relu=False, out=None, split=False):
    """act(conv1x1(x, W) + bias (+ residual)) with W^T from transpose_filter (x: N x K x H x W)."""
    _f32c(x, "x"); _f32c(wt, "wt")
    N, Kc, H, W = x.shape
    y = out if out is not None else torch.empty((N, M, H, W), dtype=torch.float32, device="cuda")
    d = gemm_conv_desc(wt, wt.shape[1], x, y, Kc, M, bias, residual, None, relu, False)
    _run_gemm(d, "conv1x1_gemm", split)
    return y


def conv1x1_dgrad(dy, w, mask=None, accumulate_into=None, split=False):
    """dX = W^T . dY (w: [M][C](x1x1) natural layout), optionally masked by mask > 0, optionally
    added onto `accumulate_into`."""
    _f32c(dy, "dy")
    N, M, H, W = dy.shape
    Cc = w.shape[1]
    w2 = _f32c(w.reshape(M, Cc), "w")
    dx = accumulate_into if accumulate_into is not None else torch.empty((N, Cc, H, W), dtype=torch.float32,
                                                                     device="cuda")
    d = gemm_conv_desc(w2, Cc, dy, dx, M, Cc, None, None, mask, False, accumulate_into is not None)
    _run_gemm(d, "conv1x1_gemm (dgrad)", split)
    return dx


def conv1x1_wgrad(x, dy, out=None, accumulate=False, split=False, x_amax=None, dy_amax=None):
    """dW [M][C] = sum_{n,p} dy[n][m][p] x[n][c][p].  split: the split-operand engine (ssad_conv1x1_wgrad_split)."""
    _f32c(x, "x"); _f32c(dy, "dy")
    N, Cc, H, W = x.shape
    M = dy.shape[1]
    dw = out if out is not None else torch.empty((M, Cc), dtype=torch.float32, device="cuda")
    L = lib()
    size_fn = L.ssad_conv1x1_wgrad_split_workspace_bytes if split else L.ssad_conv1x1_wgrad_workspace_bytes
    fn = L.ssad_conv1x1_wgrad_split if split else L.ssad_conv1x1_wgrad
    nb = size_fn(N, Cc, H * W, M)
    if split and not nb:
        raise KernelError("conv1x1_wgrad_split: unsupported geometry")
    ws = _workspace(nb, "wgrad1x1")
    if split and x_amax is not None:
        _check(L.ssad_conv1x1_wgrad_split_amax(_ptr(x), _ptr(dy), N, Cc, H * W, M, _ptr(dw), int(accumulate), _ptr(ws), nb,
                                               _ptr(x_amax), _ptr(dy_amax), _stream()), "conv1x1_wgrad_split_amax")
        return dw
    _check(fn(_ptr(x), _ptr(dy), N, Cc, H * W, M, _ptr(dw), int(accumulate), _ptr(ws), nb, _stream()),
           "conv1x1_wgrad_split" if split else "conv1x1_wgrad")
    return dw


def subsample(x, stride=2):
    _f32c(x, "x")
    N, Cc, H, W = x.shape
    y = torch.empty((N, Cc, (H - 1) // stride + 1, (W - 1) // stride + 1), dtype=torch.float32, device="cuda")
    _check(lib().ssad_subsample(_ptr(x), N, Cc, H, W, stride, _ptr(y), _stream()), "subsample")
    return y


def subsample_grad(dy, H, W, stride=2, accumulate_into=None):
    _f32c(dy, "dy")
    N, Cc = dy.shape[0], dy.shape[1]
    dx = accumulate_into if accumulate_into is not None else torch.empty((N, Cc, H, W), dtype=torch.float32,
                                                                     device="cuda")
    _check(lib().ssad_subsample_grad(_ptr(dy), N, Cc, H, W, stride, int(accumulate_into is not None), _ptr(dx),
                                     _stream()), "subsample_grad")
    return dx


def grouped_conv3x3_pack_filter(w, group):
    """[C][C/group][3][3] -> the grouped kernel's MFMA operand order."""
    _f32c(w, "w")
    Cc = w.shape[0]
    n = lib().ssad_grouped_conv3x3_filter_floats(Cc, group)
    if n < 0 or w.shape[1] * group != Cc:
        raise KernelError("grouped_conv3x3: %d channels in %d groups (4, 8, 16 or 32 per group)" % (Cc, group))
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    _check(lib().ssad_grouped_conv3x3_pack_filter(_ptr(w), Cc, group, _ptr(out), _stream()), "grouped pack")
    return out


def grouped_conv3x3_forward(x, w, bias=None, group=64, stride=1, relu=False, out=None, packed=None):
    """Grouped 3x3 / pad 1 convolution (ResNeXt): x [N,C,H,W], w [C, C/group, 3, 3] (or its pack)."""
    _f32c(x, "x")
    N, Cc, H, W = x.shape
    if packed is None:
        packed = grouped_conv3x3_pack_filter(w, group)
    oh, ow = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = out if out is not None else torch.empty((N, Cc, oh, ow), dtype=torch.float32, device="cuda")
    if bias is not None:
        _f32c(bias, "bias")
    _check(lib().ssad_grouped_conv3x3_forward(_ptr(x), _ptr(packed), _ptr(bias), N, Cc, H, W, group, stride,
                                              int(relu), _ptr(y), _stream()), "grouped_conv3x3_forward")
    return y


def conv_implicit_gemm(x, w, bias=None, stride=1, pad=0, relu=False, out=None, split_k=False):
    """k x k convolution (group 1) as an implicit GEMM: x [N,C,H,W], w [M,C,k,k] -> [N,M,OH,OW]."""
    _f32c(x, "x"); _f32c(w, "w")
    N, Cc, H, W = x.shape
    M, k = w.shape[0], w.shape[2]
    oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    wt = transpose_filter(w.reshape(M, Cc * k * k, 1, 1))
    y = out if out is not None else torch.empty((N, M, oh, ow), dtype=torch.float32, device="cuda")
    d = gemm_conv_desc(wt, wt.shape[1], x, y, Cc * k * k, M, bias, None, None, relu, False)
    d.P = oh * ow
    if split_k:
        nb = lib().ssad_conv_implicit_gemm_workspace_bytes(N, M, Cc, H, W, k, stride, pad)
        ws = _workspace(nb, "implicit_splitk")
        _check(lib().ssad_conv_implicit_gemm_ws(C.byref(d), Cc, H, W, k, stride, pad, _ptr(ws), nb, _stream()),
               "conv_implicit_gemm_ws")
        return y
    _check(lib().ssad_conv_implicit_gemm(C.byref(d), Cc, H, W, k, stride, pad, _stream()), "conv_implicit_gemm")
    return y


def conv_kxk_wgrad(x, dy, k, stride, pad, out=None, accumulate=False):
    """dW [M][C][k][k] of a k x k / strided convolution (conv_strided.hip): x [N,C,H,W], dy [N,M,OH,OW]."""
    _f32c(x, "x"); _f32c(dy, "dy")
    N, Cc, H, W = x.shape
    M = dy.shape[1]
    dw = out if out is not None else torch.empty((M, Cc, k, k), dtype=torch.float32, device="cuda")
    nb = lib().ssad_conv_kxk_wgrad_workspace_bytes(N, Cc, H, W, M, k, stride, pad)
    ws = _workspace(nb, "kxk_wgrad")
    _check(lib().ssad_conv_kxk_wgrad(_ptr(x), _ptr(dy), N, Cc, H, W, M, k, stride, pad, _ptr(dw), int(accumulate),
                                     _ptr(ws), nb, _stream()), "conv_kxk_wgrad")
    return dw


def conv_kxk_dgrad(w, dy, H, W, stride, pad, mask=None, out=None, accumulate=False):
    """dX [N][C][H][W] of a k x k / strided convolution: w [M,C,k,k], dy [N,M,OH,OW]; mask: ReluGradient of the
    layer below (dX = 0 where mask <= 0)."""
    _f32c(w, "w"); _f32c(dy, "dy")
    M, Cc, k = w.shape[0], w.shape[1], w.shape[2]
    N = dy.shape[0]
    dx = out if out is not None else torch.empty((N, Cc, H, W), dtype=torch.float32, device="cuda")
    nb = lib().ssad_conv_kxk_dgrad_workspace_bytes(N, Cc, H, W, M, k, stride, pad)
    ws = _workspace(nb, "kxk_dgrad")
    _check(lib().ssad_conv_kxk_dgrad(_ptr(w), _ptr(dy), N, Cc, H, W, M, k, stride, pad, _ptr(dx),
                                     _ptr(mask) if mask is not None else None, int(accumulate), _ptr(ws), nb,
                                     _stream()), "conv_kxk_dgrad")
    return dx
