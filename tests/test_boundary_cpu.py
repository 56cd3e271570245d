"""Host-side checks that need no GPU: the C-ABI library loads and exports
every symbol include/*.h declares, the protobuf wire codec round-trips
between Python and C++, registry / schema / gradient makers behave like the
reference's, and the graph builder reproduces the op list captured from the
reference's retinanet_heads.py."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import ssad_amd  # noqa: F401
from ssad_amd.caffe2_hip import _capi, caffe2_pb2, core, dyndep, workspace
from ssad_amd.modeling import retinanet_heads as rh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    dyndep.InitOpsLibrary()
    return _capi.load()


def declared_symbols():
    names = []
    for h in ("ssad_kernels.h", "c2hip_capi.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"#define[^\n]*", "", text)
        names += re.findall(r"(?:SSAD_API|C2HIP_CAPI)\s+[^;(]*?\b(\w+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(lib):
    names = declared_symbols()
    assert len(names) >= 40, names
    raw = ctypes.CDLL(_capi.LIB_PATH)
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, "declared in include/ but not exported: %s" % missing
    assert raw.ssad_kernels_abi_version() == 3
    raw.ssad_kernels_arch.restype = ctypes.c_char_p
    assert raw.ssad_kernels_arch() == b"gfx950"


def test_registered_operators(lib):
    for op in ("SigmoidAdaptiveDistillLoss", "SigmoidAdaptiveDistillLossGradient", "PowSum",
               "Conv", "ConvGradient", "Relu", "ReluGradient", "Sigmoid", "Sum", "Scale",
               "WeightedSum", "ConstantFill", "MomentumSGDUpdate", "SigmoidFocalLoss",
               "SigmoidFocalLossGradient", "SelectSmoothL1Loss", "SelectSmoothL1LossGradient",
               "NCCLAllreduce", "NCCLBroadcast"):      # optimizer.py:72-92: the DP net's exchange is an operator
        assert core.IsOperator(op), op
    assert not core.IsOperator("NoSuchOp")
    mi, ma, mo, mx = (ctypes.c_int() for _ in range(4))
    assert lib.c2hip_has_schema(b"SigmoidAdaptiveDistillLoss", mi, ma, mo, mx)
    assert (mi.value, ma.value, mo.value, mx.value) == (4, 4, 1, 1)
    assert lib.c2hip_has_schema(b"SigmoidAdaptiveDistillLossGradient", mi, ma, mo, mx)
    assert (mi.value, ma.value) == (5, 5)
    assert lib.c2hip_has_schema(b"PowSum", mi, ma, mo, mx)
    assert mi.value == 1 and ma.value >= 1000 and mo.value == 1
    # no communicator in this process: the collectives are the reference's single-GPU no-ops
    from ssad_amd.caffe2_hip import workspace
    assert workspace.CommWorld() == 0
    workspace.CommDestroy()                  # idempotent


def test_proto_wire_roundtrip_python_and_cpp(lib):
    op = core.CreateOperator(
        "SigmoidAdaptiveDistillLoss", ["x", "t", "g", "n"], ["loss"], gamma=2.0, alpha=0.5,
        scale=0.125, beta=0.0, num_classes=80, ignored_label=-1,
        device_option=core.DeviceOption(caffe2_pb2.CUDA, 3), engine="CUDNN")
    back = caffe2_pb2.OperatorDef().ParseFromString(op.SerializeToString())
    assert back.to_jsonable() == op.to_jsonable()
    assert back.device_option == op.device_option
    # C++ parses the bytes, builds the gradient def, serializes it back
    gops, gin = core.GradientRegistry.GetGradientForOp(op, ["loss_grad"])
    assert len(gops) == 1 and gin == ["x_grad", None, None, None]   # logits only (.cc:99-112)
    g = gops[0]
    assert g.type == "SigmoidAdaptiveDistillLossGradient" and g.is_gradient_op
    assert g.input == ["x", "t", "g", "n", "loss_grad"] and g.output == ["x_grad"]
    args = {a.name: a for a in g.arg}
    assert args["ignored_label"].i == -1 and args["num_classes"].i == 80
    assert abs(args["scale"].f - 0.125) < 1e-9 and abs(args["gamma"].f - 2.0) < 1e-9
    assert g.device_option == op.device_option


def test_conv_and_relu_gradient_makers(lib):
    c = core.CreateOperator("Conv", ["X", "w", "b"], ["Y"], kernel=3, pad=1, stride=1,
                            order="NCHW", engine="CUDNN", exhaustive_search=False)
    gops, gin = core.GradientRegistry.GetGradientForOp(c, ["Y_grad"])
    assert gops[0].type == "ConvGradient" and gops[0].input == ["X", "w", "Y_grad"]
    assert gops[0].output == ["w_grad", "b_grad", "X_grad"] and gin == ["X_grad", "w_grad", "b_grad"]
    c2 = core.CreateOperator("Conv", ["X", "w"], ["Y"], kernel=3, pad=1, stride=1)
    gops, gin = core.GradientRegistry.GetGradientForOp(c2, ["Y_grad"])
    assert gops[0].output == ["w_grad", "X_grad"]
    assert any(a.name == "no_bias" and a.i == 1 for a in gops[0].arg)
    r = core.CreateOperator("Relu", ["Y"], ["Y"])
    gops, gin = core.GradientRegistry.GetGradientForOp(r, ["Y_grad"])
    assert gops[0].type == "ReluGradient" and gops[0].input == ["Y", "Y_grad"]
    with pytest.raises(_capi.C2Error):
        core.GradientRegistry.GetGradientForOp(core.CreateOperator("Sigmoid", ["a"], ["b"]), ["g"])


def test_errors_match_reference_behaviour(lib):
    workspace.ResetWorkspace()
    x = np.zeros((1, 6, 2, 2), np.float32)
    for n in ("x", "t"):
        workspace.FeedBlob(n, x)
    workspace.FeedBlob("g", np.zeros((1, 2, 2, 2), np.int32))
    workspace.FeedBlob("n", np.ones((), np.float32))
    # no CPU implementation, exactly like the reference (.h:42-45)
    op = core.CreateOperator("SigmoidAdaptiveDistillLoss", ["x", "t", "g", "n"], ["l"], num_classes=3)
    with pytest.raises(_capi.C2Error, match="Not Implemented"):
        workspace.RunOperatorOnce(op)
    with pytest.raises(_capi.C2Error, match="Not Implemented"):
        workspace.RunOperatorOnce(core.CreateOperator("PowSum", ["x"], ["s"], power=1.8))
    # scale < 0 is rejected in the constructor (.h:39)
    with pytest.raises(_capi.C2Error, match="scale_ >= 0"):
        workspace.RunOperatorOnce(core.CreateOperator(
            "SigmoidAdaptiveDistillLoss", ["x", "t", "g", "n"], ["l"], scale=-1.0))
    # schema: wrong input count
    with pytest.raises(_capi.C2Error, match="Input size"):
        workspace.RunOperatorOnce(core.CreateOperator("SigmoidAdaptiveDistillLoss", ["x", "t"], ["l"]))
    # missing input blob
    with pytest.raises(_capi.C2Error, match="non-existing input blob"):
        workspace.RunOperatorOnce(core.CreateOperator("PowSum", ["nope"], ["s"]))
    # unknown operator / no implementation for the device
    with pytest.raises(_capi.C2Error, match="Cannot create operator"):
        workspace.RunOperatorOnce(core.CreateOperator("Conv", ["x", "t"], ["y"], kernel=3))
    assert workspace.HasBlob("x") and "x" in workspace.Blobs()
    assert np.array_equal(workspace.FetchBlob("g"), np.zeros((1, 2, 2, 2), np.int32))
    workspace.ResetWorkspace()
    assert not workspace.HasBlob("x")


def golden_graph():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "head_graph_r50_distill.json")))


def op_signature(op):
    ja = op.to_jsonable()
    args = {a["name"]: a.get("f", a.get("i", a.get("s"))) for a in ja["arg"]}
    if op.engine:
        args["engine"] = op.engine
    return ja["type"], ja["input"], ja["output"], args


def same_args(mine, ref):
    if set(mine) != set(ref):
        return False
    for k, v in ref.items():
        if isinstance(v, str):
            if mine[k] != v:
                return False
        elif abs(float(mine[k]) - float(v)) > 1e-6 * max(1.0, abs(float(v))):
            return False
    return True


def build_student(train=True, prefix="", fuse=False):
    m = rh.HeadModel(rh.HeadConfig(fuse_relu=fuse), train=train)
    blobs = [prefix + "fpn_%d" % l for l in range(7, 2, -1)]
    rh.add_fpn_retinanet_outputs(m, blobs, 256, prefix)
    return m


def test_head_graph_matches_reference_builder(lib):
    """same ops, blob names, args as the list captured
    from the reference's retinanet_heads.py (tests/golden/make_head_graph.py)."""
    g = golden_graph()
    m = build_student()
    lg = {}
    lg.update(rh.add_fpn_retinanet_losses(m))
    lg.update(rh.add_distill_loss(m))
    ops = m.net.Proto().op
    assert len(ops) == len(g["student_ops"]) == 121
    for mine, ref in zip(ops, g["student_ops"]):
        t, i, o, a = op_signature(mine)
        assert (t, i, o) == (ref["type"], ref["input"], ref["output"])
        assert same_args(a, ref["args"]), (t, a, ref["args"])
    assert lg == g["loss_gradients"]
    assert m.losses == g["student_losses"] and m.metrics == g["student_metrics"]
    ref_params = {p["name"]: p for p in g["student_params"]}
    assert [p[0] for p in m.params] == [p["name"] for p in g["student_params"]]
    for name, shape, (filler, kw) in m.params:
        assert shape == ref_params[name]["shape"] and filler == ref_params[name]["init"][0]
        for k, v in ref_params[name]["init"][1].items():
            assert abs(kw[k] - v) < 1e-6
    # teacher (test mode): + Sigmoid per level under the teacher/ prefix
    t = build_student(train=False, prefix="teacher/")
    tops = t.net.Proto().op
    assert len(tops) == len(g["teacher_ops"])
    for mine, ref in zip(tops, g["teacher_ops"]):
        ty, i, o, a = op_signature(mine)
        assert (ty, i, o) == (ref["type"], ref["input"], ref["output"]) and same_args(a, ref["args"])


def test_backward_graph_shared_weight_accumulation(lib):
    """Gradient generation: 50 ConvGradient + 40 ReluGradient + 5 distill
    gradients; every shared head parameter gets five `_grad_autosplit_k`
    pieces summed by one Sum op (caffe2/python/core.py:706-741)."""
    m = build_student()
    loss_grads = rh.add_distill_loss(m)
    n_fwd = len(m.net.Proto().op)
    grad_map = m.net.AddGradientOperators(loss_grads)
    bwd = m.net.Proto().op[n_fwd:]
    hist = {}
    for op in bwd:
        hist[op.type] = hist.get(op.type, 0) + 1
    # only the cls subnet receives a gradient from the distillation loss
    assert hist["SigmoidAdaptiveDistillLossGradient"] == 5
    assert hist["ConvGradient"] == 25 and hist["ReluGradient"] == 20
    sums = [op for op in bwd if op.type == "Sum"]
    w = "retnet_cls_conv_n0_fpn3_w"
    s = [op for op in sums if op.output == [w + "_grad"]]
    assert len(s) == 1 and s[0].input == ["%s_grad_autosplit_%d" % (w, k) for k in range(5)]
    assert grad_map[w] == w + "_grad" and grad_map["retnet_cls_pred_fpn3_b"] == "retnet_cls_pred_fpn3_b_grad"
    assert len(sums) == 10   # 5 weights + 5 biases of the cls subnet
    # every autosplit piece is written by exactly one ConvGradient
    written = [o for op in bwd if op.type == "ConvGradient" for o in op.output]
    for k in range(5):
        assert written.count("%s_grad_autosplit_%d" % (w, k)) == 1
    assert grad_map["fpn_3"] == "fpn_3_grad"


def test_full_backward_graph_with_supervised_losses(lib):
    """All three loss families: 50 ConvGradient, 40 ReluGradient, the two
    gradients of every cls logit blob (focal + distill) summed by autograd."""
    m = build_student()
    lg = rh.add_fpn_retinanet_losses(m)
    lg.update(rh.add_distill_loss(m))
    n_fwd = len(m.net.Proto().op)
    grad_map = m.net.AddGradientOperators(lg)
    bwd = m.net.Proto().op[n_fwd:]
    hist = {}
    for op in bwd:
        hist[op.type] = hist.get(op.type, 0) + 1
    assert hist["ConvGradient"] == 50 and hist["ReluGradient"] == 40
    assert hist["SigmoidFocalLossGradient"] == 5 and hist["SelectSmoothL1LossGradient"] == 5
    assert hist["SigmoidAdaptiveDistillLossGradient"] == 5
    s = [op for op in bwd if op.type == "Sum" and op.output == ["retnet_cls_pred_fpn3_grad"]]
    assert len(s) == 1 and len(s[0].input) == 2
    # 20 shared parameters + 5 logits blobs + 5 fpn blobs (cls + bbox towers)
    assert hist["Sum"] == 30
    assert grad_map["fpn_7"] == "fpn_7_grad"


@pytest.mark.parametrize("capture,cfg_kw", [
    ("backbone_graph_r50_fpn.json", {}),
    # the ResNeXt-101-64x4d teacher body (configs/focal_distillation/
    # retinanet_X-101-64x4d-FPN_1x_teacher.yaml:3,19-24)
    ("backbone_graph_x101_64x4d_fpn.json",
     dict(block_counts=(3, 4, 23, 3), stride_1x1=False, num_groups=64, width_per_group=4)),
])
def test_backbone_graph_matches_reference_capture(capture, cfg_kw):
    """modeling/resnet_fpn.py emits, op for op and parameter for parameter, what the
    reference's ResNet.py / FPN.py builders emit (capture: tests/golden/make_backbone_graph.py)."""
    import json
    import os
    from ssad_amd.modeling import resnet_fpn as rf
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", capture)))
    model = rf.BodyModel(rf.BodyConfig(**cfg_kw))
    blobs, dim, scales = rf.add_fpn_resnet_conv5_body(model)
    assert [str(b) for b in blobs] == g["fpn_blobs"] and dim == g["fpn_dim"]
    assert scales == g["spatial_scales"]
    ops = model.net.Proto().op
    assert len(ops) == len(g["ops"])
    for mine, ref in zip(ops, g["ops"]):
        assert mine.type == ref["type"]
        assert list(mine.input) == ref["input"] and list(mine.output) == ref["output"], ref
        got = {}
        for a in mine.arg:
            if a.HasField("i"):
                got[a.name] = a.i
            elif a.HasField("s"):
                got[a.name] = a.s.decode() if isinstance(a.s, bytes) else a.s
            elif a.HasField("f"):
                got[a.name] = a.f
        want = {k: (int(v) if isinstance(v, bool) else v) for k, v in ref["args"].items()}
        engine = want.pop("engine", "")          # a field of OperatorDef, not an argument
        assert (mine.engine or "") == engine
        assert got == want, (ref, got)
    mine_params = [(n, s, [i[0], i[1]]) for n, s, i in model.params]
    assert mine_params == [(p["name"], p["shape"], p["init"]) for p in g["params"]]


def test_tuned_gemm_picks_file_is_well_formed():
    """tools/harness/tunableop_gfx950.csv (the per-shape rocBLAS / hipBLASLt picks bench.py loads, never
    searches): validator lines for this stack first, then one pick per GEMM key."""
    import csv
    path = os.path.join(ROOT, "tools", "harness", "tunableop_gfx950.csv")
    rows = list(csv.reader(open(path)))
    validators = {r[1]: r[2] for r in rows if r[0] == "Validator"}
    assert {"PT_VERSION", "ROCBLAS_VERSION", "HIPBLASLT_VERSION", "GCN_ARCH_NAME"} <= set(validators)
    assert validators["GCN_ARCH_NAME"].startswith("gfx950")
    picks = [r for r in rows if r[0] != "Validator"]
    assert len(picks) >= 40 and len({(r[0], r[1]) for r in picks}) == len(picks)
    for op, key, solution, ms in picks:
        assert op.startswith("Gemm") and solution.startswith("Gemm_") and float(ms) > 0


def test_kernel_entry_points_refuse_bad_arguments_without_a_device(lib):
    """The launchers validate their arguments before they touch the GPU: inconsistent calls
    return SSAD_E_BADARG (-1) / SSAD_E_WORKSPACE (-2) on a machine without a device too."""
    raw = ctypes.CDLL(_capi.LIB_PATH)
    vp, i32, f32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    one = ctypes.create_string_buffer(64)
    p = ctypes.cast(one, vp)
    raw.ssad_conv3x3_forward_f16.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
    # blocked output needs whole 8-channel blocks
    assert raw.ssad_conv3x3_forward_f16(p, p, None, None, 1, 32, 4, 4, 36, 0, p, None) == -1
    # the mask flag and the mask pointer go together
    assert raw.ssad_conv3x3_forward_f16(p, p, None, None, 1, 32, 4, 4, 32, 2, p, None) == -1
    assert raw.ssad_conv3x3_forward_f16(None, p, None, None, 1, 32, 4, 4, 32, 0, p, None) == -1
    raw.ssad_conv3x3_wgrad_f16.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, sz, vp]
    raw.ssad_conv3x3_wgrad_f16_workspace_bytes.restype = sz
    raw.ssad_conv3x3_wgrad_f16_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    need = raw.ssad_conv3x3_wgrad_f16_workspace_bytes(2, 256, 10, 14, 256)
    assert need >= 9 * 256 * 256 * 4
    assert raw.ssad_conv3x3_wgrad_f16(p, p, 2, 256, 10, 14, 256, 0, 1.0, p, None, p, need - 1, None) == -2
    assert raw.ssad_conv3x3_wgrad_f16(p, None, 2, 256, 10, 14, 256, 0, 1.0, p, None, p, need, None) == -1
    raw.ssad_f16_pack_activations.argtypes = [vp, i32, i32, i32, i32, f32, vp, vp]
    assert raw.ssad_f16_pack_activations(p, 1, 0, 4, 4, 1.0, p, None) == -1
    raw.ssad_max_pool3x3s2_bias_relu.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
    assert raw.ssad_max_pool3x3s2_bias_relu(None, None, 1, 4, 8, 8, 1, p, None) == -1
    raw.ssad_affine_channel.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    assert raw.ssad_affine_channel(p, None, None, None, p, 1, 0, 16, 1, None) == -1
    raw.ssad_conv_out_size.argtypes = [i32] * 6
    assert raw.ssad_conv_out_size(7, 3, 1, 1, 1, 2) == 4 and raw.ssad_conv_out_size(2, 7, 1, 0, 0, 1) == -1
    # round 3: the backbones' fp16 entry points
    from ssad_amd import kernels as K
    raw.ssad_conv1x1_f16.argtypes = [ctypes.POINTER(K.PwF16), vp]
    d = K.PwF16()
    d.x = d.w = d.y = p.value
    d.N, d.C, d.M, d.Ho, d.Wo, d.Hi, d.Wi, d.stride, d.flags = 1, 32, 36, 4, 4, 4, 4, 1, 0
    assert raw.ssad_conv1x1_f16(ctypes.byref(d), None) == -1            # whole 8-channel output blocks only
    d.M, d.stride = 32, 3
    assert raw.ssad_conv1x1_f16(ctypes.byref(d), None) == -1            # stride 1 or 2
    d.stride, d.Hi = 2, 6
    assert raw.ssad_conv1x1_f16(ctypes.byref(d), None) == -1            # (Ho - 1) * stride must lie inside the input
    d.stride, d.Hi, d.flags = 1, 4, K.PW_F16_RES_UPSAMPLE2
    assert raw.ssad_conv1x1_f16(ctypes.byref(d), None) == -1            # the upsampled residual needs a residual
    d.flags, d.N = 0, 0
    assert raw.ssad_conv1x1_f16(ctypes.byref(d), None) == 0             # an empty batch is a no-op
    raw.ssad_conv1x1_wgrad_f16_workspace_bytes.restype = sz
    raw.ssad_conv1x1_wgrad_f16_workspace_bytes.argtypes = [i32] * 5
    raw.ssad_conv1x1_wgrad_f16.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, sz, vp]
    need = raw.ssad_conv1x1_wgrad_f16_workspace_bytes(2, 256, 10, 14, 64)
    assert need >= 256 * 64 * 4
    assert raw.ssad_conv1x1_wgrad_f16(p, p, 2, 256, 10, 14, 64, 0, 1.0, None, p, None, p, need - 1, None) == -2
    assert raw.ssad_conv1x1_wgrad_f16(p, p, 2, 256, 10, 14, 64, 0, 1.0, None, None, None, p, need, None) == -1
    raw.ssad_grouped_conv3x3_f16.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
    assert raw.ssad_grouped_conv3x3_f16(p, p, None, 1, 256, 8, 8, 3, 1, p, None) == -1       # 256 % 3
    assert raw.ssad_grouped_conv3x3_f16(p, p, None, 1, 256, 8, 8, 2, 1, p, None) == -1       # 128-wide groups
    assert raw.ssad_grouped_conv3x3_f16(p, p, None, 1, 96, 8, 8, 24, 1, p, None) == -1       # C % 64
    raw.ssad_grouped_conv3x3_f16_filter_halves.restype = sz
    raw.ssad_grouped_conv3x3_f16_filter_halves.argtypes = [i32, i32]
    assert raw.ssad_grouped_conv3x3_f16_filter_halves(256, 64) == 16 * 5 * 64 * 8
    assert raw.ssad_grouped_conv3x3_f16_filter_halves(2048, 64) == 128 * 9 * 64 * 8
    raw.ssad_f16_elementwise.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    assert raw.ssad_f16_elementwise(9, p, None, p, 1, 8, 2, 2, 1, 0, None) == -1             # unknown mode
    assert raw.ssad_f16_elementwise(3, p, None, p, 1, 8, 2, 2, 1, 0, None) == -1             # a sum needs b
    assert raw.ssad_f16_elementwise(0, p, None, p, 1, 8, 2, 2, 0, 0, None) == -1             # stride >= 1
    raw.ssad_gemm_f32.argtypes = [i32, i32, i32, i32, i32, f32, vp, i32, ctypes.c_longlong, vp, i32, ctypes.c_longlong,
                                  f32, vp, i32, ctypes.c_longlong, i32, vp]
    assert raw.ssad_gemm_f32(0, 0, 4, 4, 4, 1.0, p, 0, 0, p, 4, 0, 0.0, p, 4, 0, 1, None) == -1   # lda < 1
    assert raw.ssad_gemm_f32(0, 0, 0, 4, 4, 1.0, None, 4, 0, None, 4, 0, 0.0, None, 4, 0, 1, None) == 0   # empty
    raw.ssad_momentum_sgd_flat.argtypes = [vp, vp, vp, vp, f32, f32, ctypes.POINTER(K.SgdSegment), i32, vp, vp]
    seg = (K.SgdSegment * 1)(K.SgdSegment(0, 8, 0, 0, p.value))
    assert raw.ssad_momentum_sgd_flat(p, p, p, p, 0.9, 1e-4, seg, 1, None, None) == -1      # row_scale without row_len


def test_split_engine_entry_points_refuse_bad_arguments_without_a_device(lib):
    """Round 6: the split-operand engines validate before they launch -- the two |max| words of a filter gradient go
    together, workspaces are checked against the size queries, tensors of 2 GiB and more are sent back to the exact
    engines, the filter table refuses an entry whose leading dimension is too small."""
    from ssad_amd import kernels as K
    raw = ctypes.CDLL(_capi.LIB_PATH)
    vp, i32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    one = ctypes.create_string_buffer(4096)
    p = ctypes.cast(one, vp)
    lv = (K.ConvLevel * 1)(K.ConvLevel(p.value, 0, p.value, 2, 10, 14, 0, 0))
    raw.ssad_conv3x3_wgrad_split_workspace_bytes.restype = sz
    raw.ssad_conv3x3_wgrad_split_workspace_bytes.argtypes = [ctypes.POINTER(K.ConvLevel), i32, i32, i32]
    need = raw.ssad_conv3x3_wgrad_split_workspace_bytes(lv, 1, 256, 256)
    assert need >= 9 * 256 * 256 * 4
    raw.ssad_conv3x3_wgrad_split_amax.argtypes = [ctypes.POINTER(K.ConvLevel), i32, vp, vp, i32, i32, i32, vp, sz, vp, vp, vp]
    assert raw.ssad_conv3x3_wgrad_split_amax(lv, 1, p, None, 256, 256, 0, p, need, p, None, None) == -1    # one word only
    assert raw.ssad_conv3x3_wgrad_split_amax(lv, 1, p, None, 256, 256, 0, p, need - 1, p, p, None) == -2
    assert raw.ssad_conv3x3_wgrad_split_amax(lv, 0, p, None, 256, 256, 0, p, need, p, p, None) == -1
    raw.ssad_conv1x1_wgrad_split_workspace_bytes.restype = sz
    raw.ssad_conv1x1_wgrad_split_workspace_bytes.argtypes = [i32] * 4
    raw.ssad_conv1x1_wgrad_split_amax.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, vp, sz, vp, vp, vp]
    need = raw.ssad_conv1x1_wgrad_split_workspace_bytes(2, 256, 160, 512)
    assert need >= 256 * 512 * 4
    assert raw.ssad_conv1x1_wgrad_split_workspace_bytes(2, 256, 12, 512) == 0                    # pixels % 8
    assert raw.ssad_conv1x1_wgrad_split_workspace_bytes(16, 1024, 160 * 224, 256) == 0           # 2.3 GB: exact engine
    assert raw.ssad_conv1x1_wgrad_split_amax(p, p, 2, 256, 160, 512, p, 0, p, need, None, p, None) == -1
    assert raw.ssad_conv1x1_wgrad_split_amax(p, p, 2, 256, 160, 512, p, 0, p, need - 1, p, p, None) == -2
    raw.ssad_gemm_split_filter_floats.restype = sz
    raw.ssad_gemm_split_filter_floats.argtypes = [i32, i32]
    assert raw.ssad_gemm_split_filter_floats(1024, 256) == 16 + 128 * 256 * 8
    assert raw.ssad_gemm_split_filter_floats(0, 256) == 0
    raw.ssad_gemm_split_pack_filters.argtypes = [ctypes.POINTER(K.GemmPackEntry), i32, vp]
    bad = (K.GemmPackEntry * 1)(K.GemmPackEntry(p.value, p.value, 100, 64, 128))               # lda < M
    assert raw.ssad_gemm_split_pack_filters(bad, 1, None) == -1
    raw.ssad_split_absmax.argtypes = [vp, ctypes.c_longlong, vp, vp]
    assert raw.ssad_split_absmax(p, 0, p, None) == -1 and raw.ssad_split_absmax(None, 16, p, None) == -1
    raw.ssad_split_absmax_levels.argtypes = [ctypes.POINTER(K.ConvLevel), i32, i32, i32, vp, vp]
    assert raw.ssad_split_absmax_levels(lv, 0, 256, 0, p, None) == -1
    assert raw.ssad_split_absmax_levels(lv, 1, 256, 0, None, None) == -1
    nox = (K.ConvLevel * 1)(K.ConvLevel(0, 0, p.value, 2, 10, 14, 0, 0))
    assert raw.ssad_split_absmax_levels(nox, 1, 256, 0, p, None) == -1                           # field 0 = x: missing


def test_winograd_launch_count_query_follows_the_level_shapes():
    """ssad_conv3x3_forward_wino_launches (host-side, no device work): one launch per staging geometry present.
    8 x 16 patches tile 80 x 112 exactly; 8 x 8 sub-patches save 12.5 % of the pixels on 40 x 56."""
    import ctypes as C
    from ssad_amd import kernels as K
    if os.environ.get("SSAD_WINO_PAIRS") or os.environ.get("SSAD_WINO_VARIANT"):
        pytest.skip("geometry forced through the environment")
    L = K.lib()

    def launches(shapes):
        arr = (K.ConvLevel * len(shapes))()
        for i, (n, h, w) in enumerate(shapes):
            arr[i].N, arr[i].H, arr[i].W = n, h, w
        return L.ssad_conv3x3_forward_wino_launches(arr, len(shapes))

    assert launches([(16, 80, 112)]) == 1
    assert launches([(16, 40, 56)]) == 1
    assert launches([(16, 80, 112), (16, 40, 56)]) == 2
    assert launches([(16, 80, 112), (16, 40, 56), (16, 20, 28), (16, 10, 14), (16, 5, 7)]) == 2
    assert launches([(0, 80, 112), (16, 40, 56)]) == 1          # an empty level launches nothing
    assert L.ssad_conv3x3_forward_wino_launches(None, 0) == 0


def test_strided_conv_entry_points_refuse_bad_arguments_without_a_device(lib):
    """Round 3's k x k / strided convolution entry points (FPN's P6 / P7 at their own size; conv_strided.hip,
    gemm_conv.hip) and the one-launch filter transposes: workspace sizing is host arithmetic, every argument error
    is found before the first HIP call."""
    raw = ctypes.CDLL(_capi.LIB_PATH)
    vp, i32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    buf = ctypes.create_string_buffer(256)
    p = ctypes.cast(buf, vp)
    for fn in ("ssad_conv_kxk_wgrad_workspace_bytes", "ssad_conv_kxk_dgrad_workspace_bytes",
               "ssad_conv_implicit_gemm_workspace_bytes"):
        getattr(raw, fn).restype = sz
        getattr(raw, fn).argtypes = [i32] * 8
    # P6 at the bench's size: N 16, C 2048, 20 x 28, M 256, 3x3 / 2 / 1 -> col buffer [18432][2240] + dyT + GEMM slabs
    need = raw.ssad_conv_kxk_wgrad_workspace_bytes(16, 2048, 20, 28, 256, 3, 2, 1)
    assert need >= 18432 * 2240 * 4 + 256 * 2240 * 4
    assert raw.ssad_conv_kxk_dgrad_workspace_bytes(16, 2048, 20, 28, 256, 3, 2, 1) >= 18432 * 2240 * 4
    # a column buffer of 2 GiB or more, a stride of 0, a kernel larger than the padded map: no plan
    assert raw.ssad_conv_kxk_wgrad_workspace_bytes(64, 2048, 40, 56, 256, 3, 1, 1) == 0
    assert raw.ssad_conv_kxk_wgrad_workspace_bytes(1, 8, 6, 6, 4, 3, 0, 1) == 0
    assert raw.ssad_conv_kxk_dgrad_workspace_bytes(1, 8, 2, 2, 4, 7, 1, 1) == 0
    # forward split-K plan: (N, M, C, H, W, kernel, stride, pad); a pointwise-sized reduction needs no slabs
    assert raw.ssad_conv_implicit_gemm_workspace_bytes(16, 256, 2048, 20, 28, 3, 2, 1) >= 2 * 16 * 256 * 140 * 4
    assert raw.ssad_conv_implicit_gemm_workspace_bytes(16, 64, 3, 640, 896, 7, 2, 3) == 0
    raw.ssad_conv_kxk_wgrad.argtypes = [vp, vp] + [i32] * 8 + [vp, i32, vp, sz, vp]
    raw.ssad_conv_kxk_dgrad.argtypes = [vp, vp] + [i32] * 8 + [vp, vp, i32, vp, sz, vp]
    assert raw.ssad_conv_kxk_wgrad(None, p, 1, 8, 6, 6, 4, 3, 2, 1, p, 0, p, 1 << 20, None) == -1
    assert raw.ssad_conv_kxk_wgrad(p, p, 1, 8, 6, 6, 4, 3, 2, 1, p, 0, None, 1 << 20, None) == -2
    assert raw.ssad_conv_kxk_wgrad(p, p, 1, 8, 6, 6, 4, 3, 2, 1, p, 0, p, 16, None) == -2
    assert raw.ssad_conv_kxk_dgrad(p, p, 1, 3, 6, 6, 4, 3, 2, 1, p, None, 0, p, 1 << 20, None) == -1   # C*k*k % 4
    assert raw.ssad_conv_kxk_dgrad(p, None, 1, 8, 6, 6, 4, 3, 2, 1, p, None, 0, p, 1 << 20, None) == -1

    class TransposeEntry(ctypes.Structure):
        _fields_ = [("w", vp), ("wt", vp), ("M", i32), ("K", i32), ("ldm", i32), ("reserved", i32)]
    raw.ssad_transpose_filters.argtypes = [ctypes.POINTER(TransposeEntry), i32, vp]
    tab = (TransposeEntry * 2)(TransposeEntry(ctypes.addressof(buf), ctypes.addressof(buf), 4, 4, 4, 0),
                               TransposeEntry(ctypes.addressof(buf), ctypes.addressof(buf), 8, 4, 4, 0))   # ldm < M
    assert raw.ssad_transpose_filters(tab, 2, None) == -1
    assert raw.ssad_transpose_filters(None, 1, None) == -1
    assert raw.ssad_transpose_filters(None, 0, None) == 0
