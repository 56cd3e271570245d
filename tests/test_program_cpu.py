"""The native step driver's host side without a GPU: the program of one iteration is built
over CPU tensors (addresses only; nothing is launched), its structure is checked against the
reference graph's launch inventory, and the executor refuses to run it without a device."""
import ctypes
import os
import re

import pytest
import torch

import ssad_amd  # noqa: F401
from ssad_amd import kernels as K, program as PR
from ssad_amd.head_pipeline import DistillHeads, DistillHeadsF16
from ssad_amd.modeling.retinanet_heads import HeadConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(10, 14), (5, 7)]


def _codes(h, lo, hi):
    return [o.code for o in h.prog.ops[h.prog.marks[lo]:h.prog.marks[hi]]]


def _fwd_convs(h):
    """the convolution launches of the forward segment (which also holds the fill of the |max| table and the |max|
    pass over the bound inputs)"""
    m = h.prog.marks
    return [o for o in h.prog.ops[m["forward"]:m["losses"]] if o.code == PR.CONV3X3]


def test_program_header_symbols_are_exported():
    text = open(os.path.join(ROOT, "include", "ssad_program.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"SSAD_API\s+[^;(]*?\b(\w+)\s*\(", text)
    assert sorted(names) == ["ssad_program_run", "ssad_timing_collect", "ssad_timing_create",
                             "ssad_timing_destroy", "ssad_timing_reset", "ssad_timing_select"]
    raw = ctypes.CDLL(K.LIB_PATH)
    assert all(hasattr(raw, n) for n in names)
    # the Python mirror of ssad_op has the C layout: 4*4 + 8*4 + 4*4 + 2*8 + 8*8 + 8
    assert ctypes.sizeof(PR.Op) == 152 and PR.Op.p.offset == 80 and PR.Op.work.offset == 144


def test_distillation_step_program_structure():
    h = DistillHeads(HeadConfig(num_gpus=1), N=1, shapes=SHAPES, device="cpu")
    m = h.prog.marks
    assert list(m) == ["pack", "forward", "losses", "backward", "backward_late_done", "sgd", "sgd_update",
                       "ls_update", "end"]
    assert m["sgd"] == m["sgd_update"] and m["ls_update"] == m["end"]      # fp32: nothing around the update
    # 10 filters x {fwd, dgrad}: one launch per pack layout -- the split-operand engine (towers + cls_pred, forward
    # and data gradient), F(2x4) (bbox_pred's data gradient: 256 outputs) and F(2x2) (bbox_pred forward: 36 outputs)
    assert _codes(h, "pack", "forward") == [PR.WINO_PACK_FILTERS] * 3
    assert [(o.i[0], o.i[1]) for o in h.prog.ops[m["pack"]:m["forward"]]] == [(9, 3), (1, 2), (1, 0)]
    # 4 tower depths (teacher+student x cls+bbox in one launch each), teacher cls_pred (sigmoid),
    # student cls_pred, bbox_pred (student + teacher)
    # (in front: one fill of the |max| table, one |max| pass over the 8 bound inputs of the first depth; every later
    # launch is handed the words its producer's epilogue folded in)
    assert _codes(h, "forward", "losses") == [PR.FILL, PR.SPLIT_ABSMAX_LEVELS] + [PR.CONV3X3] * 7
    sp_f = [o for o in _fwd_convs(h) if o.i[4] == 3]
    assert all(o.p[4] and o.p[5] for o in sp_f) and sp_f[1].p[4] == sp_f[0].p[5]       # depth 1 reads depth 0's words
    assert [o.i[0] for o in _fwd_convs(h)] == [8, 8, 8, 8, 2, 2, 4]
    # engine (i[4]: 1 = F(2x2), 2 = F(2x4), 3 = split-operand) and timing class per launch
    assert [(o.i[4], o.klass) for o in _fwd_convs(h)] == \
        [(3, 28)] * 4 + [(3, 25), (3, 26), (1, 4)]
    assert {(o.i[4], o.klass) for o in h.prog.ops[m["backward"]:m["sgd"]] if o.code == PR.CONV3X3} == \
        {(2, 24), (3, 27), (3, 29)}
    # the split-engine launches share one workspace, sized for the largest (the four towers of one depth)
    sp = [o for o in h.prog.ops if o.code == PR.CONV3X3 and o.i[4] == 3]
    assert len(sp) == 11 and {o.p[3] for o in sp} == {h.split_ws.data_ptr()} and max(o.l[0] for o in sp) <= h.split_ws.numel()
    assert _codes(h, "losses", "backward") == [PR.POW_SUM, PR.CLS_LOSSES_FUSED, PR.SMOOTH_L1]
    bw = _codes(h, "backward", "sgd")
    assert bw.count(PR.CONV3X3_WGRAD) == 10 and bw.count(PR.CONV3X3) == 6
    # |max| passes of the backward segment: the loss gradient of the logits, bbox_pred's data gradient (its engine
    # does not fold), and the first tower data-gradient launch re-measuring its two inputs as one block
    assert bw.count(PR.SPLIT_ABSMAX_LEVELS) == 3
    wg = [o for o in h.prog.ops if o.code == PR.CONV3X3_WGRAD and o.i[4] == 1]
    assert len(wg) == 9 and all(o.p[4] and o.p[5] for o in wg)
    assert _codes(h, "sgd", "end") == [PR.SGD_FLAT]
    # the "late" bucket (predictions + upper tower half) is complete at the cut
    late = [o for o in h.prog.ops[m["backward"]:m["backward_late_done"]] if o.code == PR.CONV3X3_WGRAD]
    assert len(late) == 6
    # direct-form flops of SURVEY 8d: 2*9*Cout*Cin per output pixel
    px = sum(hh * ww for hh, ww in SHAPES)
    first = _fwd_convs(h)[0]
    assert first.work == 2.0 * 9 * 256 * 256 * px * 4 and first.klass == 28
    # every wgrad shares the one workspace, sized for the largest
    ws = {o.p[3] for o in h.prog.ops if o.code == PR.CONV3X3_WGRAD}
    assert ws == {h.wgrad_ws.data_ptr()}


def test_f24_switches_select_the_engine_per_launch(monkeypatch):
    """SSAD_STUDENT_F24 (bit mask) / SSAD_TEACHER_F24: 0 restores the F(2x2) engine and the round-4 timing classes;
    the bits move one family each."""
    monkeypatch.setenv("SSAD_SPLIT_CONV", "0")       # (the F(2x4) / F(2x2) switches alone; the split engine below)
    def engines(student, teacher):
        monkeypatch.setenv("SSAD_STUDENT_F24", str(student))
        monkeypatch.setenv("SSAD_TEACHER_F24", str(teacher))
        h = DistillHeads(HeadConfig(num_gpus=1), N=1, shapes=SHAPES, device="cpu")
        m = h.prog.marks
        fwd = [(o.i[4], o.klass) for o in _fwd_convs(h)]
        bwd = {(o.i[4], o.klass) for o in h.prog.ops[m["backward"]:m["sgd"]] if o.code == PR.CONV3X3}
        packs = [(o.i[0], o.i[1]) for o in h.prog.ops[m["pack"]:m["forward"]]]
        return fwd, bwd, packs
    fwd, bwd, packs = engines(0, 0)
    assert fwd == [(1, 2)] * 4 + [(1, 3), (1, 3), (1, 4)] and bwd == {(1, 16)} and packs == [(10, 0)]
    fwd, bwd, packs = engines(0, 1)                  # round 5, first half: the frozen teacher's cls_pred only
    assert fwd == [(1, 2)] * 4 + [(2, 20), (1, 3), (1, 4)] and bwd == {(1, 16)}
    fwd, bwd, packs = engines(1, 1)                  # + data gradients
    assert fwd[:4] == [(1, 2)] * 4 and bwd == {(2, 24)} and packs == [(10, 2), (10, 0)]
    fwd, bwd, packs = engines(3, 1)                  # + cls_pred forward
    assert fwd == [(1, 2)] * 4 + [(2, 20), (2, 22), (1, 4)]
    fwd, bwd, packs = engines(7, 1)                  # + towers (the teacher's join them: one launch per depth)
    assert fwd == [(2, 23)] * 4 + [(2, 20), (2, 22), (1, 4)]
    fwd, bwd, packs = engines(0, 2)                  # teacher towers alone: their own launch beside the student's
    assert [k for _, k in fwd[:8]] == [21, 2] * 4 and [e for e, _ in fwd[:8]] == [2, 1] * 4
    # the teacher pinned to F(2x2): the shared tower launch stays there, and the student's tower filters must be
    # packed for THAT engine (round-5 advisor finding: they were packed for F(2x4) and read by F(2x2))
    L = K.lib()
    for student in (7, 15):
        monkeypatch.setenv("SSAD_STUDENT_F24", str(student))
        monkeypatch.setenv("SSAD_TEACHER_F24", "0")
        h = DistillHeads(HeadConfig(num_gpus=1), N=1, shapes=SHAPES, device="cpu")
        m = h.prog.marks
        fwd = [(o.i[4], o.klass) for o in _fwd_convs(h)]
        assert fwd == [(1, 2)] * 4 + [(1, 3), (2, 22), (1, 4)]
        for t in ("cls", "bbox"):
            for name in h._layers(t)[:-1]:
                assert h.packed[name][0].numel() == L.ssad_conv_wino_filter_floats(256, 256), name
                assert h.packed[name][1].numel() == L.ssad_conv_wino24_filter_floats(256, 256), name   # dgrad: bit 1
        assert h.packed[h._layers("cls")[-1]][0].numel() == L.ssad_conv_wino24_filter_floats(720, 256)
    # SSAD_SPLIT_CONV: bit 1 = cls_pred forward (both networks), bit 2 = its data gradient
    monkeypatch.setenv("SSAD_STUDENT_F24", "15")
    monkeypatch.setenv("SSAD_TEACHER_F24", "1")
    for bits, want_f, want_b in ((1, [(3, 25), (3, 26)], {(2, 24)}), (2, [(2, 20), (2, 22)], {(2, 24), (3, 27)}),
                                 (4, [(2, 20), (2, 22)], {(2, 24)}), (8, [(2, 20), (2, 22)], {(2, 24), (3, 29)})):
        monkeypatch.setenv("SSAD_SPLIT_CONV", str(bits))
        h = DistillHeads(HeadConfig(num_gpus=1), N=1, shapes=SHAPES, device="cpu")
        m = h.prog.marks
        assert [(o.i[4], o.klass) for o in _fwd_convs(h)][4:6] == want_f
        assert {(o.i[4], o.klass) for o in h.prog.ops[m["backward"]:m["sgd"]] if o.code == PR.CONV3X3} == want_b
        cp = h._layers("cls")[-1]
        assert h.packed[cp][0].numel() == (L.ssad_conv_split_filter_floats if bits & 1 else L.ssad_conv_wino24_filter_floats)(720, 256)
        assert h.packed[cp][1].numel() == (L.ssad_conv_split_filter_floats if bits & 2 else L.ssad_conv_wino24_filter_floats)(256, 720)
        tw = h._layers("cls")[0]
        assert [(o.i[4], o.klass) for o in _fwd_convs(h)][:4] == [(3, 28) if bits & 4 else (2, 23)] * 4
        assert h.packed[tw][0].numel() == (L.ssad_conv_split_filter_floats if bits & 4 else L.ssad_conv_wino24_filter_floats)(256, 256)
        assert h.packed[tw][1].numel() == (L.ssad_conv_split_filter_floats if bits & 8 else L.ssad_conv_wino24_filter_floats)(256, 256)
        assert h.t_packed[tw].numel() == h.packed[tw][0].numel()          # the teacher's towers share the launch
    # bit 32: the >= 128-wide filter gradients (towers, cls_pred) on the split-operand engine with their own timing
    # classes; bbox_pred (36 outputs) keeps the F(3x3, 2x2) engine.  One workspace serves both engines.
    for bits, want in ((31, {(0, 5), (0, 6), (0, 7)}), (63, {(1, 68), (1, 69), (0, 7)})):
        monkeypatch.setenv("SSAD_SPLIT_CONV", str(bits))
        h = DistillHeads(HeadConfig(num_gpus=1), N=1, shapes=SHAPES, device="cpu")
        wg = [o for o in h.prog.ops if o.code == PR.CONV3X3_WGRAD]
        assert {(o.i[4], o.klass) for o in wg} == want and len(wg) == 10
        assert {o.p[3] for o in wg} == {h.wgrad_ws.data_ptr()} and max(o.l[0] for o in wg) <= h.wgrad_ws.numel()
    monkeypatch.delenv("SSAD_SPLIT_CONV")
    h = DistillHeads(HeadConfig(num_gpus=1), N=1, shapes=SHAPES, device="cpu")
    assert h.split_conv == 511
    monkeypatch.setenv("SSAD_SPLIT_CONV", "0")
    # without a teacher the student's towers are alone in their launch and follow bit 4
    monkeypatch.setenv("SSAD_STUDENT_F24", "7")
    h = DistillHeads(HeadConfig(num_gpus=1), N=1, shapes=SHAPES, device="cpu", distill=False)
    m = h.prog.marks
    assert [(o.i[4], o.klass) for o in _fwd_convs(h)][:4] == [(2, 23)] * 4
    assert h.packed[h._layers("cls")[0]][0].numel() == L.ssad_conv_wino24_filter_floats(256, 256)


def test_student_only_program_has_no_teacher_and_no_distillation():
    h = DistillHeads(HeadConfig(num_gpus=1), N=1, shapes=SHAPES, device="cpu", distill=False)
    assert h.teacher is None and not hasattr(h, "t_prob")
    assert [o.i[0] for o in _fwd_convs(h)] == [4, 4, 4, 4, 2, 2]
    assert _codes(h, "losses", "backward") == [PR.FOCAL_FWD, PR.FOCAL_BWD, PR.SMOOTH_L1]
    with pytest.raises(K.KernelError):
        h.step([torch.zeros(1, 256, a, b) for a, b in SHAPES], None, h.labels)   # no supervised inputs


def test_fp16_program_carries_the_dynamic_loss_scale():
    h = DistillHeadsF16(HeadConfig(num_gpus=1), N=1, shapes=SHAPES, device="cpu")
    assert _codes(h, "sgd", "end") == [PR.CHECK_FINITE, PR.SGD_FLAT, PR.LOSS_SCALE_UPDATE]
    sgd = h.prog.ops[h.prog.marks["sgd"] + 1]
    assert sgd.p[5] == h.ls_counters.data_ptr()                 # the update is skipped on overflow
    S, Sinv = h.ls_state.data_ptr(), h.ls_state.data_ptr() + 4
    bw = h.prog.ops[h.prog.marks["backward"]:h.prog.marks["sgd"]]
    assert {o.p[1] for o in bw if o.code == PR.F16_PACK_ACT} == {S}
    assert {o.p[1] for o in bw if o.code in (PR.F16_WGRAD, PR.F16_UNPACK_ACT)} == {Sinv}
    assert h.loss_scale == DistillHeadsF16.LOSS_SCALE


def test_inputs_are_rebound_not_copied():
    h = DistillHeads(HeadConfig(num_gpus=1), N=1, shapes=SHAPES, device="cpu")
    s = [torch.zeros(1, 256, a, b) for a, b in SHAPES]
    t = [torch.zeros(1, 256, a, b) for a, b in SHAPES]
    h._bind(student_fpn=s, teacher_fpn=t)
    first = ctypes.cast(_fwd_convs(h)[0].p[0], ctypes.POINTER(K.ConvLevel))
    # (the |max| pass over the bound inputs reads the same table, so it follows the rebinding)
    am = [o for o in h.prog.ops[h.prog.marks["forward"]:h.prog.marks["losses"]] if o.code == PR.SPLIT_ABSMAX_LEVELS]
    assert len(am) == 1 and am[0].p[0] == _fwd_convs(h)[0].p[0]
    # launch order of the first depth: teacher cls, student cls, teacher bbox, student bbox
    assert [first[k].x for k in range(8)] == [t[0].data_ptr(), t[1].data_ptr(), s[0].data_ptr(), s[1].data_ptr()] * 2
    with pytest.raises(K.KernelError):
        h._bind(student_fpn=[torch.zeros(1, 256, 3, 3)] * 2, teacher_fpn=t)


def test_running_without_a_device_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = DistillHeads(HeadConfig(num_gpus=1), N=1, shapes=SHAPES, device="cpu")
    with pytest.raises(K.KernelError):
        h.prog.run("pack", "forward", stream=0)


# ---------------------------------------------------------------------------
# backbone programs (built over CPU tensors: structure only)
# ---------------------------------------------------------------------------

def _count(prog, lo, hi):
    from collections import Counter
    return Counter(o.code for o in prog.ops[prog.marks[lo]:prog.marks[hi]])


def test_fp32_backbone_program_trains_no_body_bias_and_scales_folded_filters():
    """detectron/lib/modeling/ResNet.py:270-283 + affine_channel_op.cc: the body's convolutions have
    no bias and the frozen AffineChannel is never trained -- only FPN layers own bias parameters, and
    a filter with a folded scale s carries the s^2 row factor into the SGD table."""
    from ssad_amd.backbone_pipeline import NativeResNetFPN
    bb = NativeResNetFPN("r50", 1, (128, 128), "cpu", train=True)
    L = bb._layers
    assert all(l.gb is None for l in L.values() if l.train and l.affine)
    assert all(l.gb is not None for n, l in L.items() if n.startswith(("lat", "out", "p6", "p7")))
    assert not L["stem.0"].train and not L["res2.0.c1"].train and L["res3.0.c1"].train
    scaled = [s for s in bb.segments if s[4] is not None]
    assert len(scaled) == 4 + 6 + 3                       # the c3 layers of res3..res5
    assert all(float(s[4][0]) == 0.0625 and s[4].numel() * s[3] == s[1] for s in scaled)     # s^2, one per row
    n_bias = sum(1 for s in bb.segments if s[2])
    assert n_bias == 3 + 3 + 2                             # lat x3, out x3, p6, p7
    bw = _count(bb.prog, "backward", "sgd")
    # bias gradients of the three laterals and of P6 / P7 (the other 3x3 layers get theirs from the Winograd
    # filter-gradient launch); no body layer has one
    assert bw[PR.RELU_GRAD_ROWSUM] == 3 + 2
    # P6 / P7 (3x3, stride 2: FPN.py:193-224) run at their own size: no stride-1 layer + subsampling
    assert bw[PR.CONV_KXK_WGRAD] == 2 and bw[PR.CONV_KXK_DGRAD] == 2
    assert bw[PR.SUBSAMPLE_GRAD] == 2                      # res4.0 / res5.0's strided pointwise layers only
    fw = _count(bb.prog, "forward", "backward")
    assert fw[PR.CONV_IMPLICIT_WS] == 2 and fw[PR.SUBSAMPLE] == 3          # res3.0 / res4.0 / res5.0
    assert bw[PR.CONV1X1_WGRAD] == 2 * 13 + 3 + 3           # c1 + c3 of 13 blocks, 3 projections, 3 laterals
    assert _count(bb.prog, "sgd", "end") == {PR.SGD_FLAT: 1}
    # four gradient buckets in the order the backward pass completes them
    assert list(bb.bucket) == ["fpn", "res5", "res4", "res3"]
    assert sum(b.numel() for b in bb.bucket.values()) == bb.grads_flat.numel()


def test_backbone_engines_follow_the_f24_switches(monkeypatch):
    """3x3 layers of >= 128 channels run on the F(2x4) engine (i[4] == 2): trained network forward + data gradient
    under SSAD_STUDENT_F24 bit 8, frozen network forward under SSAD_TEACHER_F24; the 64-wide res2 layers and the
    filter gradients keep their engines."""
    from ssad_amd.backbone_pipeline import NativeResNetFPN
    def engines(train, env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        bb = NativeResNetFPN("r50", 1, (128, 128), "cpu", train=train)
        return [(o.i[4], o.klass) for o in bb.prog.ops if o.code == PR.CONV3X3], bb
    e, bb = engines(True, {"SSAD_STUDENT_F24": "15", "SSAD_SPLIT_CONV": "15"})
    # forward: res2 x3 on F(2x2); res3 x4, res4 x6, res5 x3 and the FPN output launch on F(2x4); backward: their data
    # gradients (res3.0's block is the last one that sends a gradient down)
    assert e.count((1, 48)) == 3 and e.count((2, 46)) == (4 + 6 + 3 + 1) * 2, e
    assert sum(o.code == PR.WINO_PACK_FILTERS and o.i[1] == 2 for o in bb.prog.ops) == 1      # trained packs: per step
    # SSAD_SPLIT_CONV bit 16 (default): the >= 256-wide ones (res4, res5, FPN outputs) on the split-operand engine,
    # one workspace for all of them; res3's 128-wide layers stay on F(2x4)
    e, bb = engines(True, {"SSAD_STUDENT_F24": "15", "SSAD_SPLIT_CONV": "31"})
    assert e.count((1, 48)) == 3 and e.count((2, 46)) == 4 * 2 and e.count((3, 66)) == (6 + 3 + 1) * 2, e
    assert sum(o.code == PR.WINO_PACK_FILTERS and o.i[1] == 3 for o in bb.prog.ops) == 1
    sp = [o for o in bb.prog.ops if o.code == PR.CONV3X3 and o.i[4] == 3]
    assert {o.p[3] for o in sp} == {bb.split_ws.data_ptr()} and max(o.l[0] for o in sp) <= bb.split_ws.numel()
    e, bb = engines(False, {"SSAD_TEACHER_F24": "1", "SSAD_SPLIT_CONV": "31"})
    assert e.count((1, 48)) == 3 and e.count((2, 47)) == 4 and e.count((3, 67)) == 10
    assert sum(o.code == PR.WINO_PACK_FILTERS and o.i[1] == 3 for o in bb.prep.ops) == 1      # frozen: packed once
    # bit 64 (default): the >= 256-wide filter gradients (res4 x6, the three FPN output convolutions, res5 x3) on the
    # split-operand engine, res3's stay on F(3x3, 2x2)
    def wgrads(bits):
        monkeypatch.setenv("SSAD_SPLIT_CONV", str(bits))
        bb = NativeResNetFPN("r50", 1, (128, 128), "cpu", train=True)
        return sorted((o.i[4], o.klass, o.i[1]) for o in bb.prog.ops if o.code == PR.CONV3X3_WGRAD)
    w0, w1 = wgrads(31), wgrads(95)
    assert all(x[0] == 0 for x in w0) and len(w0) == len(w1)
    assert [x for x in w1 if x[0] == 1] == [(1, 70, 256)] * 9 + [(1, 70, 512)] * 3, w1
    assert all(x[2] < 256 or x[1] != 49 for x in w1 if x[0] == 0), w1
    # bit 128 (default): pointwise layers with K, M >= 256 and enough pixels on the split-operand GEMM, sharing the 3x3
    # split engine's workspace; everything narrower keeps the exact-fp32 MFMA GEMM
    from ssad_amd import backbone_pipeline as BP
    def gemms(bits, min_px):
        monkeypatch.setenv("SSAD_SPLIT_CONV", str(bits))
        monkeypatch.setattr(BP, "GEMM_SPLIT_MIN_PIXELS", min_px)
        bb = NativeResNetFPN("r50", 1, (128, 128), "cpu", train=True)
        return bb, [o for o in bb.prog.ops if o.code == PR.GEMM_CONV_SPLIT], [o for o in bb.prog.ops if o.code == PR.GEMM_CONV]
    _, sp0, g0 = gemms(127, 0)
    _, sp1, g1 = gemms(255, 8192)            # a 128 x 128 image: no launch has 8192 pixels from res4 up
    bb, sp2, g2 = gemms(255, 0)
    assert not sp0 and not sp1 and len(g0) == len(g1) == len(sp2) + len(g2) and len(sp2) > 20
    assert {o.klass for o in sp2} == {71} and {o.p[1] for o in sp2} == {bb.split_ws.data_ptr()}
    assert max(o.l[0] for o in sp2) <= bb.split_ws.numel()
    # bit 256 (default): the pointwise filter gradients with C, M >= 256 likewise
    def pw_wgrads(bits):
        monkeypatch.setenv("SSAD_SPLIT_CONV", str(bits))
        monkeypatch.setattr(BP, "GEMM_SPLIT_MIN_PIXELS", 0)
        bb = NativeResNetFPN("r50", 1, (128, 128), "cpu", train=True)
        return [(o.i[5], o.klass, o.i[1], o.i[3]) for o in bb.prog.ops if o.code == PR.CONV1X1_WGRAD]
    p0, p1 = pw_wgrads(255), pw_wgrads(511)
    assert all(x[0] == 0 and x[1] == 52 for x in p0) and len(p0) == len(p1)
    assert all((x[0], x[1]) == ((1, 72) if min(x[2], x[3]) >= 256 else (0, 52)) for x in p1) and any(x[0] for x in p1)
    monkeypatch.setenv("SSAD_SPLIT_CONV", "15")
    e, _ = engines(True, {"SSAD_STUDENT_F24": "7"})
    assert all(x == (1, 48) for x in e) and len(e) == 3 + 14 * 2
    e, bb = engines(False, {"SSAD_TEACHER_F24": "1"})
    assert e.count((1, 48)) == 3 and e.count((2, 47)) == 14
    assert sum(o.code == PR.WINO_PACK_FILTERS and o.i[1] == 2 for o in bb.prep.ops) == 1      # frozen packs: once
    e, _ = engines(False, {"SSAD_TEACHER_F24": "0"})
    assert all(x == (1, 48) for x in e)


def test_fp16_backbone_program_uses_only_fp16_convolutions_after_the_stem():
    """BASELINE config 5: every convolution of the net with fp16 storage (conv_op_cudnn.cc:631-636);
    the 7x7 stem on the fp32 image is the one fp32-MFMA launch."""
    from ssad_amd.backbone_f16 import NativeResNetFPNF16
    from ssad_amd import synth
    hw, shapes = (128, 128), [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    heads = DistillHeadsF16(HeadConfig(num_gpus=1), N=1, shapes=shapes, device="cpu", blocked_io=True)
    assert heads.blocked_io
    codes = [o.code for o in heads.prog.ops]
    assert PR.F16_PACK_ACT in codes                        # (the prediction gradients are still packed)
    fwd = heads.prog.ops[heads.prog.marks["forward"]:heads.prog.marks["losses"]]
    assert all(o.code != PR.F16_PACK_ACT for o in fwd)     # no fp32 -> fp16 conversion of the FPN levels
    assert all(o.code != PR.F16_UNPACK_ACT for o in heads.prog.ops)
    st = NativeResNetFPNF16("r50", 1, hw, "cpu", train=True,
                            heads_io=dict(fpn_out=heads.in_blk["student"], inv_scale=heads.ls_state[1:2],
                                          d_fpn_in=(heads.dbuf["cls"][0], heads.dbuf["bbox"][0])),
                            skip_flag=heads.ls_counters)
    fw = _count(st.prog, "forward", "backward")
    assert fw[PR.CONV_IMPLICIT] == 1 and fw[PR.GEMM_CONV] == 0 and fw[PR.CONV3X3] == 0
    assert fw[PR.PW_F16] == 16 * 2 + 4 + 3 and fw[PR.F16_CONV3X3] == 16 + 1 + 2
    assert [t.data_ptr() for t in st.fpn] == [t.data_ptr() for t in heads.in_blk["student"]]
    bw = _count(st.prog, "backward", "sgd")
    assert bw[PR.GEMM_CONV] == bw[PR.CONV3X3] == bw[PR.CONV3X3_WGRAD] == bw[PR.CONV1X1_WGRAD] == 0
    assert bw[PR.PW_F16_WGRAD] == 2 * 13 + 3 + 3 and bw[PR.F16_WGRAD] == 13 + 3 + 2
    # every filter gradient is unscaled by the subnets' 1 / loss-scale, read on the device
    inv = heads.ls_state.data_ptr() + 4
    assert {o.p[2] for o in st.prog.ops if o.code == PR.PW_F16_WGRAD} == {inv}
    assert {o.p[1] for o in st.prog.ops if o.code == PR.F16_WGRAD} == {inv}
    sgd = [o for o in st.prog.ops if o.code == PR.SGD_FLAT]
    assert len(sgd) == 1 and sgd[0].p[5] == heads.ls_counters.data_ptr()     # dropped on overflow with the subnets'
    # the frozen ResNeXt teacher: grouped 3x3 layers, no backward, nothing to train
    te = NativeResNetFPNF16("x101-64x4d", 1, hw, "cpu", train=False, heads_io=dict(fpn_out=heads.in_blk["teacher"]))
    cw = _count(te.prog, "forward", "backward")
    assert cw[PR.GROUPED_F16] == 33 and te.params_flat.numel() == 0 and "sgd" not in te.prog.marks
