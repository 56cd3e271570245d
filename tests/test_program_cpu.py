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
    assert _codes(h, "pack", "forward") == [PR.WINO_PACK_FILTERS]        # 10 filters x {fwd, dgrad}: 1 launch
    # 4 tower depths (teacher+student x cls+bbox in one launch each), teacher cls_pred (sigmoid),
    # student cls_pred, bbox_pred (student + teacher)
    assert _codes(h, "forward", "losses") == [PR.CONV3X3] * 7
    assert [o.i[0] for o in h.prog.ops[m["forward"]:m["losses"]]] == [8, 8, 8, 8, 2, 2, 4]
    assert _codes(h, "losses", "backward") == [PR.POW_SUM, PR.CLS_LOSSES_FUSED, PR.SMOOTH_L1]
    bw = _codes(h, "backward", "sgd")
    assert bw.count(PR.CONV3X3_WGRAD) == 10 and bw.count(PR.CONV3X3) == 6
    assert _codes(h, "sgd", "end") == [PR.SGD_FLAT]
    # the "late" bucket (predictions + upper tower half) is complete at the cut
    late = [o for o in h.prog.ops[m["backward"]:m["backward_late_done"]] if o.code == PR.CONV3X3_WGRAD]
    assert len(late) == 6
    # direct-form flops of SURVEY 8d: 2*9*Cout*Cin per output pixel
    px = sum(hh * ww for hh, ww in SHAPES)
    first = h.prog.ops[m["forward"]]
    assert first.work == 2.0 * 9 * 256 * 256 * px * 4 and first.klass == 2
    # every wgrad shares the one workspace, sized for the largest
    ws = {o.p[3] for o in h.prog.ops if o.code == PR.CONV3X3_WGRAD}
    assert ws == {h.wgrad_ws.data_ptr()}


def test_student_only_program_has_no_teacher_and_no_distillation():
    h = DistillHeads(HeadConfig(num_gpus=1), N=1, shapes=SHAPES, device="cpu", distill=False)
    assert h.teacher is None and not hasattr(h, "t_prob")
    assert [o.i[0] for o in h.prog.ops[h.prog.marks["forward"]:h.prog.marks["losses"]]] == [4, 4, 4, 4, 2, 2]
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
    first = ctypes.cast(h.prog.ops[h.prog.marks["forward"]].p[0], ctypes.POINTER(K.ConvLevel))
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
