"""The native ResNet-FPN backbone program (backbone_pipeline.NativeResNetFPN, row f1) against
the same network written with torch's own float64 convolutions on the same weights
(tests/torch_ref.py): FPN outputs, every parameter gradient of res3..res5 + FPN, and the SGD
update (incl. the s^2 row factor of filters that carry a folded AffineChannel scale)."""
import pytest
import torch

import ssad_amd  # noqa: F401
from torch_ref import RefResNetFPN

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))


def _run(arch, mask_safe):
    from ssad_amd.backbone_pipeline import NativeResNetFPN
    N, hw = 2, (256, 384)
    gen = torch.Generator(device="cuda").manual_seed(5)
    images = torch.randn((N, 3) + hw, device="cuda", generator=gen)
    ref = RefResNetFPN(arch, seed=11)
    if mask_safe:
        ref.calibrate(images)
    nat = NativeResNetFPN(arch, N, hw, "cuda", train=True, src=ref.state_dict(), lr=0.01,
                          affine_scales=ref.scales)
    nat.pack()
    got = nat.forward(images)
    want = ref(images)
    assert [tuple(t.shape) for t in got] == [tuple(t.shape) for t in want]
    for g, w in zip(got, want):
        assert rel(g, w.detach()) < 2e-5, rel(g, w.detach())
    d_fpn = [torch.randn(t.shape, device="cuda", generator=gen) for t in want]
    torch.autograd.backward(want, [d.double() for d in d_fpn])
    nat.backward(d_fpn)
    torch.cuda.synchronize()
    errs = {}
    for name, p in ref.p.items():
        lname, kind = name.rsplit(".", 1)
        layer = nat._layers[lname]
        g = layer.gw if kind == "weight" else layer.gb
        if not p.requires_grad:
            # frozen layers, and the folded AffineChannel biases of the trainable body
            # (affine_channel_op.cc: never trained): no gradient buffer exists for them
            assert g is None and p.grad is None, name
            continue
        errs[name] = rel(g, p.grad)
    return nat, ref, errs


def test_native_backbone_forward_backward_vs_torch():
    nat, ref, errs = _run("r50", mask_safe=False)
    worst = max(errs, key=errs.get)
    # The FPN's own parameters sit above every ReLU of the body: no mask can flip underneath them
    fpn = [v for k, v in errs.items() if k.split(".")[0] in ("lat", "out", "p6")]
    assert max(fpn) < 2e-5, sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    # Body, free-running activations: an activation within fp32 round-off of zero takes the other
    # side of a ReLU mask than in the float64 reference, and on these small maps (res5 is 8 x 12) one
    # flipped element moves every gradient below it by ~sqrt(1 / elements) ~ 1e-3 (measured 6e-4 ..
    # 4e-3 over seeds).  Sanity bound 1e-2 per tensor here; the mask-safe variant below holds the same
    # tensors to 1e-4.
    assert errs[worst] < 1e-2, (worst, errs[worst])
    # SGD: weights s^2 g + wd * w (s = the folded AffineChannel scale), biases 2 g (optimizer.py:115-130)
    p0, g0 = nat.params_flat.clone(), nat.grads_flat.clone()
    nat.sgd_step()
    want = torch.empty_like(p0)
    scaled = 0
    for off, n, is_bias, row_len, s2 in nat.segments:
        g = g0[off:off + n]
        if s2 is not None:
            g = (g.view(-1, row_len) * s2.view(-1, 1)).reshape(-1)
            scaled += 1
        want[off:off + n] = p0[off:off + n] - 0.01 * (2.0 * g if is_bias else g + 1e-4 * p0[off:off + n])
    # a network built from given weights has a scale slot for every trainable folded filter (ones where
    # the source gave no scale), so that a checkpoint's AffineChannel scales can be installed later
    assert scaled == sum(1 for l in nat._layers.values() if l.train and l.affine)
    assert float(nat._layers["res3.0.c3"].s2[0]) == pytest.approx(0.0625)
    assert torch.allclose(nat.params_flat, want, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("small_launches_on_the_split_gemm", [False, True])
def test_native_backbone_meets_1e4_when_masks_cannot_flip(monkeypatch, small_launches_on_the_split_gemm):
    """Every backbone gradient at north_star's 1e-4 once no pre-activation of the trainable part
    can sit within round-off of zero (torch_ref.calibrate).  The split-operand pointwise GEMM serves launches of
    >= 8192 pixels by default (the bench's res4 / res5 / laterals); True sends this test's small maps through it too."""
    if small_launches_on_the_split_gemm:
        from ssad_amd import backbone_pipeline as BP
        monkeypatch.setattr(BP, "GEMM_SPLIT_MIN_PIXELS", 0)
    nat, _, errs = _run("r50", mask_safe=True)
    from ssad_amd import program as PR
    assert any(o.code == PR.GEMM_CONV_SPLIT for o in nat.prog.ops) == small_launches_on_the_split_gemm
    worst = max(errs, key=errs.get)
    assert errs[worst] < 1e-4, sorted(errs.items(), key=lambda kv: -kv[1])[:5]


def test_native_backbone_default_initialisation_needs_no_harness():
    """Random weights of the network's shapes without importing anything outside the package."""
    import sys
    from ssad_amd.backbone_pipeline import NativeResNetFPN
    nat = NativeResNetFPN("r50", 1, (128, 128), "cuda", train=True)
    assert not any(m.startswith("tools.harness") for m in sys.modules)
    assert float(nat._layers["res3.0.c3"].s2[0]) == pytest.approx(nat.INIT_C3_SCALE ** 2)
    nat.pack()
    got = nat.forward(torch.randn(1, 3, 128, 128, device="cuda"))
    assert all(torch.isfinite(t).all() for t in got)
    # activations stay O(1) through the 16 blocks
    assert 1e-3 < float(got[0].abs().mean()) < 1e3


@pytest.mark.parametrize("small_launches_on_the_split_gemm", [False, True])
def test_native_backbone_frozen_teacher_matches_torch_r101_small(monkeypatch, small_launches_on_the_split_gemm):
    from ssad_amd.backbone_pipeline import NativeResNetFPN
    if small_launches_on_the_split_gemm:
        from ssad_amd import backbone_pipeline as BP
        monkeypatch.setattr(BP, "GEMM_SPLIT_MIN_PIXELS", 0)
    ref = RefResNetFPN("r101", seed=12)
    N, hw = 1, (128, 256)
    nat = NativeResNetFPN("r101", N, hw, "cuda", train=False, src=ref.state_dict())
    assert nat.params_flat.numel() == 0 and "sgd" not in nat.prog.marks
    images = torch.randn((N, 3) + hw, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    got = nat.forward(images)
    with torch.no_grad():
        want = ref(images)
    for g, w in zip(got, want):
        assert rel(g, w) < 3e-5


def test_native_resnext_teacher_matches_torch():
    """ResNeXt-101-64x4d (BASELINE config 5's teacher): cardinality-64 grouped 3x3 layers with the
    stride on the 3x3 (grouped_conv3x3.hip), forward only, against torch's float64 network."""
    from ssad_amd.backbone_pipeline import NativeResNetFPN
    from ssad_amd import kernels as K
    ref = RefResNetFPN("x101-64x4d", seed=13)
    N, hw = 1, (128, 256)
    nat = NativeResNetFPN("x101-64x4d", N, hw, "cuda", train=False, src=ref.state_dict())
    assert nat._layers["res2.0.c2"].group == 64 and nat._layers["res3.0.c2"].stride == 2
    images = torch.randn((N, 3) + hw, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9))
    got = nat.forward(images)
    with torch.no_grad():
        want = ref(images)
    for g, w in zip(got, want):
        assert tuple(g.shape) == tuple(w.shape)
        assert rel(g, w) < 2e-5, rel(g, w)
    with pytest.raises(K.KernelError):
        NativeResNetFPN("x101-64x4d", N, hw, "cuda", train=True)
