"""The net lowering (csrc/ops/net_lowering.cc) as pure host logic, no GPU: the NetDef wire codec, the
rewrite of the reference's captured head graph (detectron/lib/modeling/retinanet_heads.py:63-352 through
tests/golden/head_graph_r50_distill.json's builder) and the hazards the schedule must honour."""
import collections

import pytest

import ssad_amd  # noqa: F401
from ssad_amd.caffe2_hip import caffe2_pb2, core, dyndep, workspace
from ssad_amd.modeling import optimizer as opt
from ssad_amd.modeling import retinanet_heads as rh

GPU = core.DeviceOption(caffe2_pb2.HIP, 0)


@pytest.fixture(scope="module", autouse=True)
def lib():
    dyndep.InitOpsLibrary()


def head_nets(update=False):
    cfg = rh.HeadConfig(num_gpus=1)
    levels = list(cfg.levels())
    with core.DeviceScope(GPU):
        teacher = rh.HeadModel(cfg, train=False, name="teacher")
        rh.add_fpn_retinanet_outputs(teacher, ["teacher/fpn_%d" % l for l in reversed(levels)], 256, "teacher/")
        student = rh.HeadModel(cfg, train=True, name="student")
        rh.add_fpn_retinanet_outputs(student, ["fpn_%d" % l for l in reversed(levels)], 256)
        lg = rh.add_fpn_retinanet_losses(student)
        lg.update(rh.add_distill_loss(student))
        gm = student.net.AddGradientOperators(lg)
        if update:
            opt.add_allreduce_ops(student, gm)
            opt.add_param_update_ops(student, gm)
    return cfg, teacher, student, gm


def replay(ops):
    """What each blob holds after running `ops` in order, symbolically: a blob's value is the op that
    wrote it applied to the values it read.  Two op lists that leave the same symbolic value in a blob
    compute the same thing."""
    val = {}

    def get(n):
        return val.get(n, ("input", n))
    for op in ops:
        ins = tuple(get(n) for n in op.input)
        args = tuple(sorted((a.name, str(a.to_jsonable())) for a in op.arg))
        for k, o in enumerate(op.output):
            val[o] = (op.type, args, ins, k)
    return val


def test_netdef_wire_roundtrip():
    _, teacher, _, _ = head_nets()
    proto = teacher.net.Proto()
    proto.type, proto.num_workers = "dag", 4
    proto.external_input = ["teacher/fpn_3"]
    proto.external_output = ["teacher/retnet_cls_prob_fpn3"]
    proto.arg.append(core.MakeArgument("hip_lowering", 0))
    back = caffe2_pb2.NetDef().ParseFromString(proto.SerializeToString())
    assert back.name == "teacher" and back.type == "dag" and back.num_workers == 4
    assert back.external_input == proto.external_input and back.external_output == proto.external_output
    assert [o.to_jsonable() for o in back.op] == [o.to_jsonable() for o in proto.op]
    assert [a.to_jsonable() for a in back.arg] == [a.to_jsonable() for a in proto.arg]
    # the C++ codec reads the same bytes: lowering a net without convolutions returns it unchanged
    plain = core.Net("plain")
    with core.DeviceScope(GPU):
        plain.Relu(["x"], ["y"])
        plain.Sum(["y", "x"], ["z"])
    ops, report = workspace.LowerNet(plain)
    assert [o.to_jsonable() for o in ops] == [o.to_jsonable() for o in plain.Proto().op]
    assert "ops 2 -> 2" in report


def test_teacher_net_lowers_to_one_launch_per_depth():
    _, teacher, _, _ = head_nets()
    ops, report = workspace.LowerNet(teacher.net)
    hist = collections.Counter(o.type for o in ops)
    # 50 Conv + 40 Relu + 5 Sigmoid -> 4 tower depths (cls and bbox towers, five levels: 10 problems each)
    # + the cls prediction layer with the Sigmoid in its epilogue + the bbox prediction layer
    assert hist == {"ConvGroup": 6}, hist
    for g in [o for o in ops if o.type == "ConvGroup"][:4]:
        assert len(g.output) == 10 and len(g.input) == 30
        assert any(a.name == "fuse_relu" and a.i == 1 for a in g.arg)
        assert any(a.name == "engine" for a in g.arg) or g.engine == "CUDNN"
    preds = [o for o in ops if o.type == "ConvGroup"][4:]
    assert not any(a.name == "fuse_relu" for p in preds for a in p.arg)
    cls = [p for p in preds if any(a.name == "fuse_sigmoid" and a.i == 1 for a in p.arg)]
    box = [p for p in preds if not any(a.name == "fuse_sigmoid" for a in p.arg)]
    assert len(cls) == 1 and len(box) == 1
    # the probabilities are what the cls group writes; the logits blob is not produced
    assert sorted(cls[0].output) == sorted(["teacher/retnet_cls_prob_fpn%d" % l for l in range(3, 8)])
    assert sorted(box[0].output) == sorted(["teacher/retnet_bbox_pred_fpn%d" % l for l in range(3, 8)])
    assert "FELL BACK" not in report and "Relu fused 40" in report and "Sigmoid fused 5" in report
    # ... unless somebody asks for the logits
    teacher.net.Proto().external_output.append("teacher/retnet_cls_pred_fpn3")
    ops2, report2 = workspace.LowerNet(teacher.net)
    assert "Sigmoid fused 4" in report2 and collections.Counter(o.type for o in ops2)["Sigmoid"] == 1, report2
    teacher.net.Proto().external_output.pop()


def test_student_training_net_lowering():
    _, _, student, gm = head_nets(update=True)
    n_in = len(student.net.Proto().op)
    ops, report = workspace.LowerNet(student.net)
    hist = collections.Counter(o.type for o in ops)
    assert "FELL BACK" not in report, report
    assert hist["ConvGroup"] == 5 and hist["ConvGradientGroup"] == 5, hist
    assert hist["Conv"] == 0 and hist["ConvGradient"] == 0 and hist["Relu"] == 0 and hist["ReluGradient"] == 0
    # the 20 shared-filter gradient Sums are absorbed; the 5 logit-gradient Sums and the 5 fpn-gradient Sums stay
    assert hist["Sum"] == 10, hist
    assert hist["MomentumSGDUpdate"] == 20 and hist["NCCLAllreduce"] == 20
    assert "Sum absorbed 20" in report and "ReluGradient fused 40" in report
    assert len(ops) < n_in // 2
    # trained nets: forward and data gradient marked for the split-operand / F(2x4) engines (hip_algo = split: the
    # operators take the split engine from 256 channels up, F(2x4) from 128) unless the net says hip_train_f24 = 0
    assert "50 trained" not in report and " 100 trained (100 of them marked split)" in report, report   # 50 Conv + 50 ConvGradient
    for g in [o for o in ops if o.type in ("ConvGroup", "ConvGradientGroup")]:
        assert [a.s for a in g.arg if a.name == "hip_algo"] in (["split"], [b"split"]), g.type
    proto = student.net.Proto()
    proto.arg.append(core.MakeArgument("hip_train_f24", 0))
    ops0, report0 = workspace.LowerNet(student.net)
    assert " 0 trained" in report0 and not any(a.name == "hip_algo" for o in ops0 for a in o.arg), report0
    proto.arg.pop()
    groups = [o for o in ops if o.type == "ConvGradientGroup"]
    for g in groups:
        nf = [a for a in g.arg if a.name == "n_filters"][0].i
        fi = [a for a in g.arg if a.name == "filter_index"][0].ints
        assert nf == 2 and len(fi) == 10 and sorted(set(fi)) == [0, 1]
        assert len(g.input) == 30 and len(g.output) == 2 * nf + 10
        for w in g.output[:nf]:
            assert w.endswith("_w_grad"), w             # the Sum's output, not an autosplit piece
        for b in g.output[nf:2 * nf]:
            assert b.endswith("_b_grad"), b
    # every parameter's gradient blob is still produced, under the name the update ops read
    produced = {o for op in ops for o in op.output}
    for p, g in gm.items():
        if p.endswith("_w") or p.endswith("_b"):
            assert g in produced, (p, g)
    # order: a filter's update comes after every reader of the filter and after its gradient
    pos_update = {op.input[3]: i for i, op in enumerate(ops) if op.type == "MomentumSGDUpdate"}
    for i, op in enumerate(ops):
        if op.type in ("ConvGroup", "ConvGradientGroup"):
            for name in op.input:
                if name in pos_update:
                    assert i < pos_update[name], (op.type, name)


def test_lowered_list_computes_the_same_values():
    """Symbolic replay: the lowered list leaves, in every blob that survives, a value built from the
    same inputs -- checked structurally by expanding the group operators back into their members."""
    _, _, student, _ = head_nets(update=True)
    ops, _ = workspace.LowerNet(student.net)
    # expand the groups / fusions back into reference operators and compare with the original replay
    expanded = []
    for op in ops:
        if op.type == "ConvGroup":
            per = len(op.input) // len(op.output)
            relu = any(a.name == "fuse_relu" for a in op.arg)
            # (hip_algo picks the engine that computes the same convolution: not part of the value)
            args = [a for a in op.arg if a.name not in ("fuse_relu", "hip_algo")]
            for k, y in enumerate(op.output):
                c = caffe2_pb2.OperatorDef()
                c.type, c.input, c.output, c.arg = "Conv", op.input[per * k:per * k + per], [y], args
                expanded.append(c)
                if relu:
                    r = caffe2_pb2.OperatorDef()
                    r.type, r.input, r.output = "Relu", [y], [y]
                    expanded.append(r)
        else:
            expanded.append(op)
    # (the originals through the wire once, as the lowered list went: float arguments are fp32 there)
    wire = [caffe2_pb2.OperatorDef().ParseFromString(o.SerializeToString()) for o in student.net.Proto().op]
    want = replay(wire)
    got = replay(expanded)
    # forward values: every activation, prediction and loss blob
    for name, v in want.items():
        if v[0] in ("Conv", "Relu", "SigmoidFocalLoss", "SigmoidAdaptiveDistillLoss", "SelectSmoothL1Loss",
                    "PowSum"):
            assert got.get(name) == v, name


def test_write_after_read_and_keep():
    """(i) An in-place update of a filter must stay behind every convolution that reads the old value even
    when grouping pulls convolutions together; (ii) a blob named in external_output is never fused away."""
    net = core.Net("hazard")
    conv = dict(kernel=3, pad=1, stride=1, order="NCHW")
    with core.DeviceScope(GPU):
        net.Conv(["x0", "w", "b"], ["y0"], **conv)
        net.Scale(["w"], ["w"], scale=0.5)              # writes w between the two convolutions
        net.Conv(["x1", "w", "b"], ["y1"], **conv)
    ops, report = workspace.LowerNet(net)
    assert [o.type for o in ops] == ["Conv", "Scale", "Conv"], report       # not grouped across the write
    net2 = core.Net("keep")
    with core.DeviceScope(GPU):
        net2.Conv(["x0", "w", "b"], ["y0"], **conv)
        net2.Conv(["x1", "w", "b"], ["y1"], **conv)
        gm = net2.AddGradientOperators({"y0": "y0_grad", "y1": "y1_grad"})
    ops, report = workspace.LowerNet(net2)
    assert [o.type for o in ops] == ["ConvGroup", "ConvGradientGroup"], report
    assert "Sum absorbed 2" in report
    net2.Proto().external_output = ["w_grad_autosplit_0"]
    ops, report = workspace.LowerNet(net2)
    assert "Sum absorbed 0" in report and collections.Counter(o.type for o in ops)["Sum"] == 2
    assert gm["w"] == "w_grad"


def test_f24_engine_marking_of_evaluated_and_trained_nets():
    """A net without gradient operators (the teacher net, an inference net) is only evaluated: its 3x3 convolutions
    get hip_algo = "winograd24" (conv3x3_winograd24.hip; 3 multiplies per output, fp32 error ~2e-6 of the scale) --
    switch hip_frozen_f24.  A net that trains gets it on Conv and ConvGradient (the operators apply it to the forward
    pass and the data gradient of >= 128-wide layers) -- switch hip_train_f24, independent of the first; an explicit
    hip_algo in the graph is respected."""
    _, teacher, student, _ = head_nets(update=True)
    t_ops, t_rep = workspace.LowerNet(teacher.net)
    for g in [o for o in t_ops if o.type == "ConvGroup"]:
        assert [a.s for a in g.arg if a.name == "hip_algo"] == [b"split"], g.arg
    assert "F(2x4) Conv 50" in t_rep and "(50 of them marked split)" in t_rep
    # hip_split = 0: the same marks name the F(2x4) engine alone (round 5's behaviour)
    teacher.net.Proto().arg.append(core.MakeArgument("hip_split", 0))
    t_ops0, t_rep0 = workspace.LowerNet(teacher.net)
    for g in [o for o in t_ops0 if o.type == "ConvGroup"]:
        assert [a.s for a in g.arg if a.name == "hip_algo"] == [b"winograd24"], g.arg
    assert "(0 of them marked split)" in t_rep0
    teacher.net.Proto().arg.pop()
    s_ops, s_rep = workspace.LowerNet(student.net)
    assert "F(2x4) Conv 0 evaluated / 100 trained" in s_rep, s_rep
    student.net.Proto().arg.append(core.MakeArgument("hip_train_f24", 0))
    s_ops, s_rep = workspace.LowerNet(student.net)
    assert not any(a.name == "hip_algo" for o in s_ops for a in o.arg) and "F(2x4) Conv 0 evaluated / 0 trained" in s_rep
    teacher.net.Proto().arg.append(core.MakeArgument("hip_train_f24", 0))        # not the teacher's switch
    assert "F(2x4) Conv 50 evaluated" in workspace.LowerNet(teacher.net)[1]
    teacher.net.Proto().arg.append(core.MakeArgument("hip_frozen_f24", 0))
    assert "F(2x4) Conv 0 evaluated" in workspace.LowerNet(teacher.net)[1]
    net = core.Net("pinned")
    with core.DeviceScope(GPU):
        net.Conv(["x", "w", "b"], ["y"], kernel=3, pad=1, stride=1, order="NCHW", hip_algo="direct")
        net.Conv(["y", "w2", "b2"], ["z"], kernel=3, pad=1, stride=1, order="NCHW")
    ops, _ = workspace.LowerNet(net)
    algos = [[a.s for a in o.arg if a.name == "hip_algo"] for o in ops]
    assert algos == [[b"direct"], [b"split"]], algos
