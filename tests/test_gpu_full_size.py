"""The objects bench.py measures, AT THE SIZE it measures them, checked for stream races.

BASELINE config 3 (R-50-FPN student + R-101-FPN teacher, bs 16, 3x640x896, fp32) and config 5
(R-101-FPN student + ResNeXt-101-64x4d-FPN teacher, bs 16, 3x512x768, every convolution in fp16
storage / fp32 accumulation, conv_op_cudnn.cc:631-636) through backbone_pipeline.NativeDistillModel
= build_generic_retinanet_model_dissstillation (detectron/lib/modeling/model_builder.py:373-411) on
one GPU, with the schedule of bench.py's timed region: step() on a high-priority stream, the frozen
teacher on a side stream (for fp32 ordered only after the previous step's last reader of its
outputs), filter gradients on the executor's auxiliary streams, the launching thread several
iterations ahead of the GPU (nothing synchronises between the iterations).

The arithmetic is pinned elsewhere (tests/test_gpu_native_model.py against the composed oracle at
256x384; per-kernel tests at the full level shapes); what only exists at full size is the TIMING --
hundreds of workgroups per launch, kernels of several streams really sharing the chip.  Every
kernel is deterministic, so the same iterations enqueued on ONE stream with every overlap switched
off must give the same bits in every loss, parameter, momentum and gradient buffer; scratch buffers
are poisoned with NaN first, so a read-before-write shows up as well.
"""
import numpy as np
import pytest
import torch

import ssad_amd  # noqa: F401
from ssad_amd import synth
from ssad_amd.modeling.retinanet_heads import HeadConfig

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _inputs(N, shapes, image_hw, seed):
    """bench.py's synthetic batch: labels 5 % ignore / 2 % foreground, box targets for every foreground anchor."""
    gen = torch.Generator(device=DEV).manual_seed(seed)
    labels = []
    for h, w in shapes:
        u = torch.rand((N, 9, h, w), device=DEV, generator=gen)
        lab = torch.zeros((N, 9, h, w), dtype=torch.int32, device=DEV)
        lab[u < 0.05] = -1
        fg = (u >= 0.05) & (u < 0.07)
        lab[fg] = torch.randint(1, 81, (int(fg.sum()),), device=DEV, generator=gen, dtype=torch.int32)
        labels.append(lab)
    targets, n_fg = [], 0
    for lab in labels:
        idx = torch.nonzero(lab > 0)
        Lc = torch.stack([idx[:, 0], 4 * idx[:, 1], idx[:, 2], idx[:, 3]], dim=1).float().contiguous()
        Y = (torch.randn((Lc.shape[0], 4), device=DEV, generator=gen) * 0.5).contiguous()
        targets.append((Y, Lc))
        n_fg += Lc.shape[0]
    fg_num = torch.tensor([float(max(n_fg, 1))], device=DEV)
    images = torch.randn((N, 3) + image_hw, device=DEV, generator=gen)
    return images, labels, targets, fg_num


def _build(student, teacher, N, image_hw, shapes, f16, overlap):
    from ssad_amd.head_pipeline import DistillHeads, DistillHeadsF16
    from ssad_amd.backbone_pipeline import NativeDistillModel
    cfg = HeadConfig(num_gpus=1)
    kw = dict(N=N, shapes=shapes, device=DEV, student_init=synth.head_params(np.random.default_rng(1)),
              teacher_init=synth.head_params(np.random.default_rng(2)), lr=1e-4, overlap_wgrad=overlap)
    heads = DistillHeadsF16(cfg, blocked_io=True, **kw) if f16 else DistillHeads(cfg, **kw)
    return NativeDistillModel(heads, student, teacher, N, image_hw, DEV, two_streams=overlap, overlap_wgrad=overlap)


def _run(model, batch, steps, high_priority):
    images, labels, targets, fg_num = batch
    model.student.poison()
    model.teacher.poison()
    torch.cuda.synchronize()
    if high_priority:                      # bench.py: the step's critical path on a high-priority stream
        main = torch.cuda.Stream(priority=-1)
        main.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(main):
            for _ in range(steps):
                model.step(images, labels, targets, fg_num)
    else:
        for _ in range(steps):
            model.step(images, labels, targets, fg_num)
    torch.cuda.synchronize()


def _same_bits(a, b, levels):
    for name in ("losses", "focal_losses", "bbox_losses"):
        x, y = getattr(a.heads, name), getattr(b.heads, name)
        assert torch.isfinite(x).all(), (name, x)
        assert torch.equal(x, y), (name, x, y)
    for x, y, what in ((a.heads.params.flat, b.heads.params.flat, "subnet parameters"),
                       (a.heads.moms.flat, b.heads.moms.flat, "subnet momentum"),
                       (a.heads.grads.flat, b.heads.grads.flat, "subnet update"),
                       (a.student.params_flat, b.student.params_flat, "backbone parameters"),
                       (a.student.moms_flat, b.student.moms_flat, "backbone momentum"),
                       (a.student.grads_flat, b.student.grads_flat, "backbone update")):
        assert torch.isfinite(x).all(), what
        assert torch.equal(x, y), (what, float((x - y).abs().max()))
    for l in range(levels):
        assert torch.equal(a.teacher.fpn[l], b.teacher.fpn[l]), ("teacher FPN level", l)
        assert torch.equal(a.student.fpn[l], b.student.fpn[l]), ("student FPN level", l)


def test_config3_headline_object_at_full_size_is_race_free():
    """R-50 student + R-101 teacher, bs 16, 640x896, fp32: three iterations enqueued back to back
    under the final schedule against the serial program."""
    N, hw, shapes = 16, (640, 896), synth.LEVEL_SHAPES_600
    batch = _inputs(N, shapes, hw, seed=1234)
    a = _build("r50", "r101", N, hw, shapes, f16=False, overlap=True)
    assert a.side is not None and a._teacher_ahead and a.student._wstreams == [1, 2] and a.heads._wstream == 1
    _run(a, batch, 3, high_priority=True)
    assert a._t_fpn_read is not None and a._images_ref is batch[0]      # the teacher did run ahead (steps 2, 3)
    b = _build("r50", "r101", N, hw, shapes, f16=False, overlap=False)
    assert b.side is None and b.student._wstreams == [0] and b.heads._wstream == 0
    _run(b, batch, 3, high_priority=False)
    _same_bits(a, b, len(shapes))
    for l in range(len(shapes)):
        assert torch.equal(a.student.d_fpn[l], b.student.d_fpn[l]), ("d_fpn", l)
    # the step moved every trained tensor
    fresh = _build("r50", "r101", N, hw, shapes, f16=False, overlap=False)
    assert not torch.equal(fresh.student.params_flat, a.student.params_flat)
    assert not torch.equal(fresh.heads.params.flat, a.heads.params.flat)


def test_config5_fp16_step_at_full_size_is_race_free_and_keeps_its_loss_scale():
    """R-101 student + ResNeXt-101-64x4d teacher, bs 16, 512x768, fp16 storage: three iterations under
    bench.py's schedule against the serial program, bit for bit; no iteration overflowed (the dynamic
    loss scale still has its initial value, the overflow flag is clear, every update was applied)."""
    from ssad_amd.head_pipeline import DistillHeadsF16
    N, hw, shapes = 16, (512, 768), synth.LEVEL_SHAPES_500
    batch = _inputs(N, shapes, hw, seed=4321)
    a = _build("r101", "x101-64x4d", N, hw, shapes, f16=True, overlap=True)
    assert a.backbone_f16 and type(a.student).__name__ == "NativeResNetFPNF16" and a.side is not None
    assert a.teacher._layers["res2.0.c2"].group == 64
    p_h0, p_b0 = a.heads.params.flat.clone(), a.student.params_flat.clone()
    _run(a, batch, 3, high_priority=True)
    b = _build("r101", "x101-64x4d", N, hw, shapes, f16=True, overlap=False)
    assert b.side is None
    _run(b, batch, 3, high_priority=False)
    for m in (a, b):
        assert float(m.heads.ls_state[0]) == DistillHeadsF16.LOSS_SCALE, float(m.heads.ls_state[0])
        assert float(m.heads.ls_state[0]) * float(m.heads.ls_state[1]) == pytest.approx(1.0)
        assert int(m.heads.ls_counters[0]) == 0                         # overflow flag clear
        assert int(m.heads.ls_counters[1]) == 3, m.heads.ls_counters    # three clean steps counted
    for name in ("losses", "focal_losses", "bbox_losses"):
        x, y = getattr(a.heads, name), getattr(b.heads, name)
        assert torch.isfinite(x).all() and torch.equal(x, y), (name, x, y)
    for x, y, what in ((a.heads.params.flat, b.heads.params.flat, "subnet parameters"),
                       (a.heads.moms.flat, b.heads.moms.flat, "subnet momentum"),
                       (a.student.params_flat, b.student.params_flat, "backbone parameters"),
                       (a.student.moms_flat, b.student.moms_flat, "backbone momentum"),
                       (a.student.grads_flat, b.student.grads_flat, "backbone update")):
        assert torch.isfinite(x).all(), what
        assert torch.equal(x, y), (what, float((x - y).abs().max()))
    assert not torch.equal(p_h0, a.heads.params.flat) and not torch.equal(p_b0, a.student.params_flat)
