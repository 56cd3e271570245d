"""Host side of the native step driver (include/ssad_program.h).

A `Program` is the flat list of kernel launches of one training iteration, built ONCE over
buffers that were allocated once; `run()` hands a slice of it to `ssad_program_run`, which
enqueues every launch from C++ in one call.  This is how the reference works too -- Python
builds a Caffe2 net once and the C++ executor runs it each iteration
(caffe2/caffe2/core/net_simple.cc) -- minus the per-operator dispatch.

torch appears only as the owner of device memory and of the HIP stream handle.
"""
import ctypes as C

import torch

from . import kernels as K

# op codes (enum ssad_opcode)
WINO_PACK_FILTERS, PACK_FILTER, CONV3X3, CONV3X3_WGRAD, POW_SUM, CLS_LOSSES_FUSED = 1, 2, 3, 4, 5, 6
DISTILL_FWD, DISTILL_BWD, FOCAL_FWD, FOCAL_BWD, SMOOTH_L1, SGD_FLAT = 7, 8, 9, 10, 11, 12
FILL, SCALE, SUM_N, CHECK_FINITE, LOSS_SCALE_UPDATE = 13, 14, 15, 16, 17
F16_PACK_ACT, F16_UNPACK_ACT, F16_PACK_FILTER, F16_CONV3X3, F16_WGRAD = 32, 33, 34, 35, 36
AFFINE_CHANNEL, UPSAMPLE, UPSAMPLE_GRAD, STEM_POOL, RELU_GRAD_ROWSUM, RELU_GRAD, CHANNEL_SUM = \
    48, 49, 50, 51, 52, 53, 55
GEMM_CONV, CONV1X1_WGRAD, TRANSPOSE_FILTER, SUBSAMPLE, SUBSAMPLE_GRAD, RELU, IM2COL_BATCHED = \
    54, 56, 57, 58, 59, 60, 61
FORK, JOIN = 62, 63
GROUPED_CONV3X3, GROUPED_PACK, CONV_IMPLICIT = 64, 65, 66
PW_F16, PW_F16_PACK, PW_F16_WGRAD, F16_EW, STEM_POOL_F16, GROUPED_F16, GROUPED_F16_PACK = 67, 68, 69, 70, 71, 72, 73
CONV_IMPLICIT_WS, CONV_KXK_WGRAD, CONV_KXK_DGRAD, TRANSPOSE_FILTERS, F16_PACK_FILTERS = 74, 75, 76, 77, 78
GEMM_CONV_SPLIT = 79
SPLIT_ABSMAX_LEVELS, SPLIT_ABSMAX, GEMM_SPLIT_PACK = 80, 81, 82

# timing classes: one per kernel family.  bound "mfma": work = direct-form FLOPs
# (2*9*Cout*Cin per output pixel; the Winograd engine executes 1/2.25 of them);
# bound "hbm": work = algorithmic bytes (SURVEY.md 8d per-element figures).
KLASS = {
    0: dict(name="other", bound=None),
    1: dict(name="filter pack (wino_pack_multi_kernel)", bound="hbm"),
    2: dict(name="subnet tower conv3x3 forward, 256-wide output (wino_conv_z_kernel)", bound="mfma", wino=True),
    16: dict(name="subnet conv3x3 data gradient, 256-wide output (wino_conv_z_kernel; runs beside the filter "
                  "gradients of the auxiliary stream, so its launches share the chip)", bound="mfma", wino=True),
    3: dict(name="cls_pred conv3x3 fwd, 720-wide output (wino_conv_z_kernel)", bound="mfma", wino=True),
    4: dict(name="bbox_pred conv3x3 fwd, 36-wide output (wino_conv_z_kernel)", bound="mfma", wino=True),
    5: dict(name="subnet conv3x3 filter gradient, tower layers (wino_wgrad_kernel + reduce + bias grad)",
            bound="mfma", wino=True),
    6: dict(name="cls_pred filter gradient (wino_wgrad_kernel + reduce + bias grad)", bound="mfma", wino=True),
    7: dict(name="bbox_pred filter gradient (wino_wgrad_kernel + reduce + bias grad)", bound="mfma", wino=True),
    8: dict(name="PowSum (pow_sum_kernel, one launch: sum finished by the last-arriving workgroup)", bound="hbm"),
    9: dict(name="fused classification losses fwd+bwd (cls_losses_fused_kernel, one launch)", bound="hbm"),
    10: dict(name="SelectSmoothL1Loss fwd+bwd (4 launches)", bound="hbm"),
    11: dict(name="momentum SGD, whole flat buffer (sgd_flat_kernel)", bound="hbm"),
    12: dict(name="SigmoidAdaptiveDistillLoss fwd (distill_fwd_kernel + finalize)", bound="hbm"),
    13: dict(name="SigmoidAdaptiveDistillLoss bwd (distill_bwd_kernel)", bound="hbm"),
    14: dict(name="SigmoidFocalLoss fwd (focal_fwd_kernel + finalize)", bound="hbm"),
    15: dict(name="SigmoidFocalLoss bwd (focal_bwd_kernel)", bound="hbm"),
    # the frozen teacher on the Winograd F(2x4, 3x3) engine: executed multiplies = direct-form / 3
    20: dict(name="teacher cls_pred conv3x3 fwd + sigmoid, 720-wide, F(2x4,3x3) (wino24_conv_kernel)", bound="mfma",
             wino=True, exec_div=3.0),
    21: dict(name="teacher tower conv3x3 fwd, F(2x4,3x3) (wino24_conv_kernel)", bound="mfma", wino=True, exec_div=3.0),
    # SSAD_STUDENT_F24: the trained subnets on the same engine (off by default)
    22: dict(name="cls_pred conv3x3 fwd, 720-wide, F(2x4,3x3) (wino24_conv_kernel)", bound="mfma", wino=True, exec_div=3.0),
    23: dict(name="subnet tower conv3x3 forward, F(2x4,3x3) (wino24_conv_kernel)", bound="mfma", wino=True, exec_div=3.0),
    # split-operand engine (conv3x3_split.hip): executes 3 x the direct-form flops, on the fp16 pipes (exec_div = 1/3:
    # executed = direct-form x 3); the call includes its split pass; the |max| words come from the pipeline's table (class 73)
    25: dict(name="teacher cls_pred conv3x3 fwd + sigmoid, split-operand engine (conv3x3_split_kernel + its split pass; |max| words from the table, class 73)",
             bound="mfma16", wino=True, exec_div=1.0 / 3.0),
    26: dict(name="cls_pred conv3x3 fwd, split-operand engine (conv3x3_split_kernel + its split pass; |max| words from the table, class 73)", bound="mfma16",
             wino=True, exec_div=1.0 / 3.0),
    27: dict(name="cls_pred data gradient, split-operand engine (conv3x3_split_kernel + its split pass; |max| words from the table, class 73)", bound="mfma16",
             wino=True, exec_div=1.0 / 3.0),
    28: dict(name="subnet tower conv3x3 forward, split-operand engine (conv3x3_split_kernel + its split pass; |max| words from the table, class 73)",
             bound="mfma16", wino=True, exec_div=1.0 / 3.0),
    29: dict(name="subnet tower data gradient, split-operand engine (conv3x3_split_kernel + its split pass; |max| words from the table, class 73)",
             bound="mfma16", wino=True, exec_div=1.0 / 3.0),
    24: dict(name="subnet conv3x3 data gradient, F(2x4,3x3) (wino24_conv_kernel)", bound="mfma", wino=True, exec_div=3.0),
    # direct (non-Winograd) engine
    18: dict(name="subnet conv3x3 fwd/dgrad, direct engine (conv3x3_kernel)", bound="mfma", wino=False),
    19: dict(name="subnet conv3x3 filter gradient, direct engine (conv3x3_wgrad_kernel + reduce)",
             bound="mfma", wino=False),
    # fp16 storage / fp32 accumulate (peak 2.5 PFLOP/s)
    32: dict(name="fp16 activation pack / unpack", bound="hbm"),
    33: dict(name="fp16 filter pack", bound="hbm"),
    34: dict(name="fp16 subnet conv3x3 fwd/dgrad, 256-wide (conv3x3_f16_kernel)", bound="mfma16"),
    35: dict(name="fp16 cls_pred conv3x3 fwd / dgrad (conv3x3_f16_kernel)", bound="mfma16"),
    36: dict(name="fp16 bbox_pred conv3x3 fwd / dgrad (conv3x3_f16_kernel)", bound="mfma16"),
    37: dict(name="fp16 filter gradient (conv3x3_wgrad_f16_kernel + bias grad + reduce)", bound="mfma16"),
    38: dict(name="gradient finiteness check + loss-scale update", bound="hbm"),
    # backbone (row f1)
    48: dict(name="backbone conv3x3 fwd/dgrad (wino_conv_z_kernel)", bound="mfma", wino=True),
    47: dict(name="frozen teacher backbone conv3x3 fwd, >= 128 wide, F(2x4,3x3) (wino24_conv_kernel)", bound="mfma",
             wino=True, exec_div=3.0),
    46: dict(name="backbone conv3x3 fwd/dgrad, >= 128 wide, F(2x4,3x3) (wino24_conv_kernel; SSAD_STUDENT_F24)",
             bound="mfma", wino=True, exec_div=3.0),
    49: dict(name="backbone conv3x3 filter gradient (wino_wgrad_kernel)", bound="mfma", wino=True),
    50: dict(name="backbone pointwise conv fwd / data gradient (gemm_conv_nn_kernel)", bound="mfma", wino=False),
    51: dict(name="backbone elementwise: subsample / scatter, ReluGradient + bias sums, upsample, pool, adds",
             bound="hbm"),
    52: dict(name="backbone pointwise conv filter gradient (gemm_conv_nt_kernel + reduce)", bound="mfma",
             wino=False),
    53: dict(name="stem 7x7/2 conv (implicit GEMM: gemm_conv_nn_kernel<64, true>)", bound="mfma", wino=False),
    54: dict(name="backbone filter packs (transpose / Winograd)", bound="hbm"),
    56: dict(name="grouped 3x3 conv, ResNeXt (grouped_conv3x3_kernel)", bound="mfma", wino=False),
    55: dict(name="backbone momentum SGD (sgd_flat_kernel)", bound="hbm"),
    66: dict(name="backbone conv3x3 fwd/dgrad, >= 256 wide, split-operand engine (conv3x3_split_kernel + passes; SSAD_SPLIT_CONV & 16)",
             bound="mfma16", wino=True, exec_div=1.0 / 3.0),
    67: dict(name="teacher backbone conv3x3 fwd, >= 256 wide, split-operand engine (conv3x3_split_kernel + passes)",
             bound="mfma16", wino=True, exec_div=1.0 / 3.0),
    68: dict(name="subnet conv3x3 filter gradient, tower layers, split-operand engine (wsplit_kernel + reduce + bias "
                  "grad; SSAD_SPLIT_CONV & 32)", bound="mfma16", wino=True, exec_div=1.0 / 3.0),
    69: dict(name="cls_pred filter gradient, split-operand engine (wsplit_kernel + reduce + bias grad)",
             bound="mfma16", wino=True, exec_div=1.0 / 3.0),
    70: dict(name="backbone conv3x3 filter gradient, >= 256 wide, split-operand engine (wsplit_kernel + reduce + bias grad; "
                  "SSAD_SPLIT_CONV & 64)", bound="mfma16", wino=True, exec_div=1.0 / 3.0),
    71: dict(name="backbone pointwise conv fwd / data gradient, K and M >= 256, split-operand GEMM (gemm_fly_kernel; "
                  "SSAD_SPLIT_CONV & 128)", bound="mfma16", wino=True, exec_div=1.0 / 3.0),
    72: dict(name="backbone pointwise conv filter gradient, C and M >= 256, split-operand engine (wpoint_split_kernel + "
                  "reduce; SSAD_SPLIT_CONV & 256)", bound="mfma16", wino=True, exec_div=1.0 / 3.0),
    73: dict(name="|max| passes of the split-operand engines (split_absmax_kernel: one per tensor and step, shared by the "
                  "forward, data-gradient and filter-gradient calls that read it)", bound="hbm"),
    74: dict(name="pointwise filter split for the split-operand GEMM (gsplit_*_kernel, all filters in 3 launches)",
             bound="hbm"),
    64: dict(name="P6 / P7 3x3 stride-2 conv fwd / data gradient at their own size (implicit GEMM with split-K; "
                  "flattened-batch GEMM + col2im)", bound="mfma", wino=False),
    65: dict(name="P6 / P7 3x3 stride-2 filter gradient (im2col + gemm_conv_nt_kernel + reduce)", bound="mfma",
             wino=False),
    # backbones in fp16 storage / fp32 accumulation (BASELINE config 5)
    57: dict(name="fp16 backbone pointwise conv fwd / data gradient (pw_f16_kernel)", bound="mfma16"),
    58: dict(name="fp16 backbone conv3x3 fwd / data gradient (conv3x3_f16_kernel)", bound="mfma16"),
    59: dict(name="fp16 backbone conv3x3 filter gradient (conv3x3_wgrad_f16_kernel<false> + reduce)", bound="mfma16"),
    60: dict(name="fp16 backbone pointwise filter gradient (conv3x3_wgrad_f16_kernel<true> + reduce)",
             bound="mfma16"),
    61: dict(name="fp16 grouped 3x3 conv, ResNeXt (grouped_f16_kernel)", bound="mfma16"),
    62: dict(name="fp16 backbone elementwise: subsample / scatter, ReluGradient, upsample gradient, sums, stem pool",
             bound="hbm"),
    63: dict(name="fp16 backbone filter packs", bound="hbm"),
}

PEAK = {"mfma": 157.3e12, "mfma16": 2.5e15, "hbm": 8.0e12}      # MI355X_MICROARCH.md chip table


class Op(C.Structure):
    _fields_ = [("code", C.c_int32), ("klass", C.c_int32), ("stream", C.c_int32), ("reserved", C.c_int32),
                ("i", C.c_int32 * 8), ("f", C.c_float * 4),
                ("l", C.c_int64 * 2), ("p", C.c_void_p * 8), ("work", C.c_double)]


class TimingClass(C.Structure):
    _fields_ = [("klass", C.c_int32), ("launches", C.c_int32), ("ms", C.c_double), ("work", C.c_double)]


_bound = False


def _lib():
    global _bound
    L = K.lib()
    if not _bound:
        L.ssad_timing_create.restype = C.c_void_p
        L.ssad_timing_destroy.argtypes = [C.c_void_p]
        L.ssad_timing_reset.argtypes = [C.c_void_p]
        L.ssad_timing_collect.argtypes = [C.c_void_p, C.POINTER(TimingClass), C.c_int]
        L.ssad_timing_select.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        L.ssad_program_run.argtypes = [C.POINTER(Op), C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        _bound = True
    return L


def _addr(v):
    """Device pointer of a tensor, address of a ctypes object, raw integer, or NULL."""
    if v is None:
        return None
    if isinstance(v, torch.Tensor):
        return v.data_ptr()
    if isinstance(v, int):
        return v
    if isinstance(v, C.c_void_p):
        return v.value
    return C.addressof(v)


class Timing(object):
    """Per-class launch timing taken by the executor with HIP events on the launch stream."""

    def __init__(self):
        self.handle = C.c_void_p(_lib().ssad_timing_create())

    def reset(self):
        _lib().ssad_timing_reset(self.handle)

    def select(self, klasses=()):
        """Time only the ops of these classes (empty: all)."""
        arr = (C.c_int * max(len(klasses), 1))(*klasses)
        rc = _lib().ssad_timing_select(self.handle, arr, len(klasses))
        if rc != 0:
            raise K.KernelError("ssad_timing_select failed with code %d" % rc)
        return self

    def collect(self):
        """After a stream/device synchronise: {klass: dict(launches, ms, work)}."""
        out = (TimingClass * 128)()
        n = _lib().ssad_timing_collect(self.handle, out, 128)
        if n < 0:
            raise K.KernelError("ssad_timing_collect failed with code %d" % n)
        return {out[k].klass: dict(launches=out[k].launches, ms=out[k].ms, work=out[k].work)
                for k in range(min(n, 128))}

    def __del__(self):
        try:
            if self.handle:
                _lib().ssad_timing_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class Program(object):
    def __init__(self):
        self.ops = []
        self.keep = []          # host tables / tensors the op records point into
        self.marks = {}
        self.arr = None
        self.default_stream = 0      # builders switch this around a block of ops meant for an aux stream

    def mark(self, name):
        self.marks[name] = len(self.ops)

    def fork(self, k=1):
        """Aux stream k waits for everything enqueued so far on the main stream."""
        return self.add(FORK, 0, i=(k,))

    def join(self, k=1):
        """The main stream waits for everything enqueued so far on aux stream k."""
        return self.add(JOIN, 0, i=(k,))

    def add(self, code, klass=0, i=(), f=(), l=(), p=(), work=0.0, keep=(), stream=None):
        o = Op()
        o.code, o.klass, o.work = code, klass, float(work)
        o.stream = self.default_stream if stream is None else int(stream)
        for k, v in enumerate(i):
            o.i[k] = int(v)
        for k, v in enumerate(f):
            o.f[k] = float(v)
        for k, v in enumerate(l):
            o.l[k] = int(v)
        for k, v in enumerate(p):
            o.p[k] = _addr(v)
        self.keep.extend([v for v in p if v is not None and not isinstance(v, int)])
        self.keep.extend(keep)
        self.ops.append(o)
        self.arr = None
        return len(self.ops) - 1

    def build(self):
        self.arr = (Op * max(len(self.ops), 1))(*self.ops)
        return self

    def set_ptr(self, op_index, slot, value):
        """Re-point one pointer slot of a built program (an input tensor that moved)."""
        self.ops[op_index].p[slot] = _addr(value)
        if self.arr is not None:
            self.arr[op_index].p[slot] = _addr(value)

    def run(self, lo=0, hi=None, timing=None, stream=None):
        """Enqueue ops[lo:hi] (indices or mark names) on `stream` (default: torch's current)."""
        if self.arr is None:
            self.build()
        lo = self.marks[lo] if isinstance(lo, str) else lo
        hi = len(self.ops) if hi is None else (self.marks[hi] if isinstance(hi, str) else hi)
        if hi <= lo:
            return
        st = C.c_void_p(stream if stream is not None else torch.cuda.current_stream().cuda_stream)
        failed = C.c_int(-1)
        first = C.cast(C.byref(self.arr, lo * C.sizeof(Op)), C.POINTER(Op))
        rc = _lib().ssad_program_run(first, hi - lo, st, timing.handle if timing is not None else None,
                                     C.byref(failed))
        if rc != 0:
            raise K.KernelError("program op %d (code %d) failed with code %d" % (
                lo + failed.value, self.ops[lo + failed.value].code if failed.value >= 0 else -1, rc))
