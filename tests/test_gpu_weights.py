"""Detectron weights files driving the native backbones (row f3: detectron/lib/utils/net.py:50-182).

A synthetic `model_final.pkl` in the REFERENCE's layout -- every blob of the student's body + FPN +
subnets, the teacher's under `teacher/` (net.py:71-78), pickle protocol 2 as the Python 2 reference
writes it -- is loaded into backbone_pipeline.NativeDistillModel by utils/net.py, which folds each
`<conv>_w` / `<conv>_bn_s` / `<conv>_bn_b` triple into the filter + bias the kernels run on.

Reference for the numbers = the reference's OWN graph, op for op: tests/golden/backbone_graph_*.json
is the operator list the imported reference builder emitted (tests/golden/make_backbone_graph.py:
ResNet.py:85-130,221-283 + FPN.py:116-250 under a recording model), evaluated here by a few lines of
torch in float64 on the blobs of the file -- Conv with no bias, then AffineChannel as its own op,
exactly as the reference runs them (no folding anywhere on this side).  Bar: 1e-4.
"""
import json
import os
import pickle

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ssad_amd  # noqa: F401
from ssad_amd import synth
from ssad_amd.modeling.retinanet_heads import HeadConfig
from ssad_amd.utils import net

from test_weights_file import reference_backbone_blobs, GOLDEN, GRAPHS

pytestmark = pytest.mark.gpu
DEV = "cuda"


def run_reference_graph(graph, blobs, images, prefix=""):
    """The captured operator list on float64 torch tensors.  -> the FPN levels, finest first."""
    ws = {"data": images.double()}

    def get(name):
        if name not in ws:
            ws[name] = torch.from_numpy(np.asarray(blobs[prefix + name])).to(images.device).double()
        return ws[name]
    for op in graph["ops"]:
        t, a, ins, outs = op["type"], op["args"], op["input"], op["output"]
        if t == "Conv":
            b = get(ins[2]) if len(ins) == 3 else None
            y = F.conv2d(get(ins[0]), get(ins[1]), b, a.get("stride", 1), a.get("pad", 0), a.get("dilation", 1),
                         a.get("group", 1))
        elif t == "AffineChannel":
            y = get(ins[0]) * get(ins[1]).view(1, -1, 1, 1) + get(ins[2]).view(1, -1, 1, 1)
        elif t == "Relu":
            y = F.relu(get(ins[0]))
        elif t == "MaxPool":
            y = F.max_pool2d(get(ins[0]), a["kernel"], a["stride"], a["pad"])
        elif t == "Sum":
            y = get(ins[0]) + get(ins[1])
        elif t == "UpsampleNearest":
            y = F.interpolate(get(ins[0]), scale_factor=a["scale"], mode="nearest")
        elif t == "StopGradient":
            y = get(ins[0])
        else:
            raise AssertionError("operator %s in the captured graph" % t)
        ws[outs[0]] = y
    return [ws[n] for n in reversed(graph["fpn_blobs"])]


def close_1e4(got, want, what):
    got, want = got.double(), want.double()
    tol = 1e-4 * want.abs() + 1e-5 * float(want.abs().max())
    bad = (got - want).abs() > tol
    assert not bool(bad.any()), "%s: %d outside 1e-4, worst %.3e of max %.3e" % (
        what, int(bad.sum()), float((got - want).abs().max()), float(want.abs().max()))


def _head_blobs(rng, prefix=""):
    return {prefix + k: v for k, v in synth.head_params(rng).items()}


def test_reference_layout_weights_file_drives_both_native_backbones(tmp_path):
    """Student R-50-FPN from the file's own blobs, ResNeXt-101-64x4d teacher from its `teacher/`
    blobs (second file, as TRAIN.WEIGHTS of the teacher config): FPN levels of both networks against
    the reference's captured graph in float64; then one training step, a checkpoint, and the
    checkpoint reloaded into a fresh model: same parameters, same update history, same teacher."""
    from ssad_amd.head_pipeline import DistillHeads
    rng = np.random.default_rng(17)
    s_blobs, s_graph = reference_backbone_blobs("r50", rng)
    t_blobs, t_graph = reference_backbone_blobs("x101-64x4d", rng, momentum=False)
    s_blobs.update(_head_blobs(rng))
    t_blobs.update(_head_blobs(rng))
    s_path, t_path = str(tmp_path / "R-50.pkl"), str(tmp_path / "X-101-64x4d.pkl")
    with open(s_path, "wb") as f:
        pickle.dump(dict(blobs=dict(s_blobs), cfg="NUM_GPUS: 8\n"), f, protocol=2)
    with open(t_path, "wb") as f:
        pickle.dump(dict(blobs=dict(t_blobs), cfg=""), f, protocol=2)

    N, hw = 2, (128, 256)
    shapes = [(16, 32), (8, 16), (4, 8), (2, 4), (1, 2)]
    cfg = HeadConfig(num_gpus=1)

    def heads():
        return DistillHeads(cfg, N=N, shapes=shapes, device=DEV, lr=1e-3)
    model, loaded, missing = net.native_model_from_weights_files(
        heads(), s_path, t_path, student_arch="r50", teacher_arch="x101-64x4d", N=N, image_hw=hw, device=DEV, lr=1e-3)
    assert not missing
    assert "conv1_w" in loaded and "teacher/res4_22_branch2c_bn_s" in loaded and "retnet_cls_pred_fpn3_w" in loaded
    assert not any(k.startswith(("res", "conv1", "fpn_")) for k in model.heads.preserved)
    # subnets: the file's blobs, the teacher's from the teacher file
    assert np.array_equal(model.heads.params["retnet_cls_conv_n0_fpn3_w"].cpu().numpy(),
                          s_blobs["retnet_cls_conv_n0_fpn3_w"])
    assert np.array_equal(model.heads.teacher["retnet_cls_conv_n0_fpn3_w"].cpu().numpy(),
                          t_blobs["retnet_cls_conv_n0_fpn3_w"])

    images = torch.randn((N, 3) + hw, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    model.student.pack()
    got_s = model.student.forward(images)
    got_t = model.teacher.forward(images)
    with torch.no_grad():
        want_s = run_reference_graph(s_graph, s_blobs, images)
        want_t = run_reference_graph(t_graph, t_blobs, images)
    for l in range(5):
        close_1e4(got_s[l], want_s[l], "student P%d" % (l + 3))
        close_1e4(got_t[l], want_t[l], "teacher P%d" % (l + 3))

    # the update history came with the file (folded: m' = s m) and the update uses it
    lay = model.student._layers["res4.2.c2"]
    off = (lay.w.data_ptr() - model.student.params_flat.data_ptr()) // 4
    m_file = s_blobs["res4_2_branch2b_w_momentum"] * s_blobs["res4_2_branch2b_bn_s"].reshape(-1, 1, 1, 1)
    assert np.array_equal(model.student.moms_flat[off:off + lay.w.numel()].cpu().numpy(), m_file.ravel())

    # one training iteration, then a checkpoint in the reference's layout
    lab_rng = np.random.default_rng(3)
    labs = [synth.distill_inputs(lab_rng, N, 9, 80, h, w)[2] for h, w in shapes]
    tg = [synth.bbox_targets(lab_rng, l) for l in labs]
    fg = torch.tensor([float(max(1, sum(t[0].shape[0] for t in tg)))], device=DEV)
    to = lambda a: torch.from_numpy(a).to(DEV)
    model.step(images, [to(a) for a in labs], [(to(y), to(l)) for y, l in tg], fg)
    torch.cuda.synchronize()
    assert torch.isfinite(model.heads.losses).all() and torch.isfinite(model.student.params_flat).all()
    ckpt = str(tmp_path / "model_iter0.pkl")
    net.save_model_to_weights_file(ckpt, model, cfg_yaml="NUM_GPUS: 1\n")
    saved = pickle.load(open(ckpt, "rb"))
    assert set(saved) == {"blobs", "cfg"}
    sb = saved["blobs"]
    # every blob the two source files held is in the checkpoint under the reference's names
    for k in s_blobs:
        assert k in sb, k
    for k in t_blobs:
        assert "teacher/" + k in sb, k
    # frozen things did not move: the AffineChannel blobs exactly, the frozen filters to the rounding of W' / s
    for k in ("res_conv1_bn_s", "res4_2_branch2b_bn_s", "res4_2_branch2b_bn_b", "res2_1_branch2a_bn_b"):
        assert np.array_equal(sb[k], s_blobs[k]), k
    assert np.allclose(sb["res2_1_branch2a_w"], s_blobs["res2_1_branch2a_w"], rtol=3e-7, atol=1e-12)
    assert np.allclose(sb["teacher/res3_0_branch2b_w"], t_blobs["res3_0_branch2b_w"], rtol=3e-7, atol=1e-12)
    # trained things did
    assert not np.allclose(sb["res4_2_branch2b_w"], s_blobs["res4_2_branch2b_w"], rtol=1e-6, atol=0)
    assert not np.array_equal(sb["fpn_6_b"], s_blobs["fpn_6_b"])
    # the un-folded filter the reference would hold after ITS update of W: W - (lr (s dW' + wd W) + mu m)
    #   = (W' - m') / s, m' the folded history after the step
    m_new = model.student.moms_flat[off:off + lay.w.numel()].view_as(lay.w).cpu().numpy()
    s_ = s_blobs["res4_2_branch2b_bn_s"].reshape(-1, 1, 1, 1)
    want_w = s_blobs["res4_2_branch2b_w"] - m_new / s_
    assert np.allclose(sb["res4_2_branch2b_w"], want_w, rtol=1e-5, atol=1e-8)
    assert np.allclose(sb["res4_2_branch2b_w_momentum"], m_new / s_, rtol=3e-7, atol=1e-12)

    # the checkpoint alone (it carries the teacher/ scope) resumes: same state in a fresh model
    again, _, missing2 = net.native_model_from_weights_files(
        heads(), ckpt, None, student_arch="r50", teacher_arch="x101-64x4d", N=N, image_hw=hw, device=DEV, lr=1e-3)
    assert not missing2

    def same(a, b, what, rtol=3e-7):
        a, b = a.double(), b.double()
        assert float((a - b).abs().max()) <= rtol * float(b.abs().max()) + 1e-12, what
    same(again.student.params_flat, model.student.params_flat, "backbone parameters")
    same(again.student.moms_flat, model.student.moms_flat, "backbone update history")
    same(again.student.frozen_flat, model.student.frozen_flat, "frozen stem / res2 / affine biases")
    same(again.teacher.frozen_flat, model.teacher.frozen_flat, "teacher")
    assert torch.equal(again.heads.params.flat, model.heads.params.flat)
    assert torch.equal(again.heads.moms.flat, model.heads.moms.flat)
    assert torch.equal(again.heads.teacher.flat, model.heads.teacher.flat)
    for name, lyr in model.student._layers.items():
        if lyr.s2 is not None:
            assert torch.equal(again.student._layers[name].s2, lyr.s2), name


def test_fp16_backbones_take_the_same_weights_file(tmp_path):
    """Config 5's networks (fp32 master parameters, fp16 storage in the kernels) from a
    reference-layout file: R-50 student here for size; FPN levels against the captured graph in
    float64 at the fp16 route's tolerance."""
    from ssad_amd.backbone_f16 import NativeResNetFPNF16
    rng = np.random.default_rng(23)
    blobs, graph = reference_backbone_blobs("r50", rng)
    state, scales, _, _ = net.backbone_from_blobs(blobs, "r50")
    N, hw = 2, (128, 256)
    nat = NativeResNetFPNF16("r50", N, hw, DEV, train=True, src=state, affine_scales=scales, lr=1e-3)
    net.load_backbone(nat, blobs)
    images = torch.randn((N, 3) + hw, device=DEV, generator=torch.Generator(device=DEV).manual_seed(6))
    nat.pack()
    nat.forward(images)
    with torch.no_grad():
        want = run_reference_graph(graph, blobs, images)
    for l, (a, b) in enumerate(zip(nat.fpn_f32(), want)):
        rel = float((a.double() - b).norm() / b.norm())
        assert rel < 1e-2, ("P%d" % (l + 3), rel)
    out = net.backbone_to_blobs(nat)
    assert set(out) == set(blobs)
    assert np.allclose(out["res5_2_branch2c_w"], blobs["res5_2_branch2c_w"], rtol=3e-7, atol=1e-12)
