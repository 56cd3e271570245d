"""The native ResNet-FPN backbone program (backbone_pipeline.NativeResNetFPN, row f1) against
the same network written with torch's own convolutions on the same weights: FPN outputs,
every parameter gradient of res3..res5 + FPN, and the SGD update."""
import numpy as np
import pytest
import torch

import ssad_amd  # noqa: F401

pytestmark = pytest.mark.gpu


def _torch_reference(arch, seed=11):
    """harness.full_model.ResNetFPN on plain torch operators (MIOpen / rocBLAS), no kernel of
    this repo: the independent implementation."""
    from ssad_amd.harness import full_model as fm
    fm._HIP3X3 = fm._FUSE_TAIL = fm._GEMM_1X1 = fm._FUSED_PW = False
    with torch.random.fork_rng():
        torch.manual_seed(seed)
        m = fm.ResNetFPN(arch).cuda()
        # biases away from zero so that every bias path is exercised (inside the forked RNG: the
        # test's data must not depend on what ran before it)
        with torch.no_grad():
            for name, p in m.named_parameters():
                if name.endswith("bias"):
                    p.normal_(0.0, 0.05)
    return m


def rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))


@pytest.mark.parametrize("arch", ["r50"])
def test_native_backbone_forward_backward_vs_torch(arch):
    from ssad_amd.backbone_pipeline import NativeResNetFPN
    ref = _torch_reference(arch)
    N, hw = 2, (256, 384)
    nat = NativeResNetFPN(arch, N, hw, "cuda", train=True, src=ref, lr=0.01)
    gen = torch.Generator(device="cuda").manual_seed(5)
    images = torch.randn((N, 3) + hw, device="cuda", generator=gen)
    nat.pack()
    got = nat.forward(images)
    # the reference in float64 (torch's own double-precision convolutions): what is left is this
    # repo's fp32 arithmetic, not the difference between two fp32 algorithms
    ref = ref.double()
    want = ref(images.double())
    assert [tuple(t.shape) for t in got] == [tuple(t.shape) for t in want]
    for g, w in zip(got, want):
        assert rel(g, w.detach()) < 2e-5, rel(g, w.detach())
    d_fpn = [torch.randn(t.shape, device="cuda", generator=gen) for t in want]
    torch.autograd.backward(want, [d.double() for d in d_fpn])
    nat.backward(d_fpn)
    torch.cuda.synchronize()
    errs = {}
    for name, p in ref.named_parameters():
        lname, kind = name.rsplit(".", 1)
        layer = nat._layers[lname]
        if not layer.train:
            assert p.grad is None, name
            continue
        g = layer.gw if kind == "weight" else layer.gb
        errs[name] = rel(g, p.grad)
    worst = max(errs, key=errs.get)
    # The FPN's own parameters sit above every ReLU of the body: no mask can flip underneath them
    fpn = [v for k, v in errs.items() if k.split(".")[0] in ("lat", "out", "p6")]
    assert max(fpn) < 2e-5, sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    # Body: an activation within fp32 round-off of zero takes the other side of a ReLU mask than in
    # the float64 reference, and on these small maps (res5 is 8 x 12) one flipped element of the last
    # block's gradient moves every gradient below it by ~sqrt(1 / elements) ~ 1e-3 (measured 6e-4 with
    # one seed, 1e-6 with another; tests/test_gpu_operators.py:make_mask_safe shows 1e-4 is met when
    # no mask can flip).  Bound: 3e-3 per tensor.
    assert errs[worst] < 3e-3, (worst, errs[worst])
    # SGD: weights g + wd * w, biases 2 g, momentum (optimizer.py:115-130)
    p0 = nat.params_flat.clone()
    g0 = nat.grads_flat.clone()
    nat.sgd_step()
    isb = torch.zeros_like(p0, dtype=torch.bool)
    for off, n, b in nat.segments:
        if b:
            isb[off:off + n] = True
    gg = torch.where(isb, 2.0 * g0, g0 + 1e-4 * p0)
    assert torch.allclose(nat.params_flat, p0 - 0.01 * gg, rtol=1e-5, atol=1e-8)


def test_native_backbone_frozen_teacher_matches_torch_r101_small():
    from ssad_amd.backbone_pipeline import NativeResNetFPN
    ref = _torch_reference("r101")
    N, hw = 1, (128, 256)
    nat = NativeResNetFPN("r101", N, hw, "cuda", train=False, src=ref)
    assert nat.params_flat.numel() == 0 and "sgd" not in nat.prog.marks
    images = torch.randn((N, 3) + hw, device="cuda")
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
    ref = _torch_reference("x101-64x4d")
    N, hw = 1, (128, 256)
    nat = NativeResNetFPN("x101-64x4d", N, hw, "cuda", train=False, src=ref)
    assert nat._layers["res2.0.c2"].group == 64 and nat._layers["res3.0.c2"].stride == 2
    images = torch.randn((N, 3) + hw, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9))
    got = nat.forward(images)
    with torch.no_grad():
        want = ref.double()(images.double())
    for g, w in zip(got, want):
        assert tuple(g.shape) == tuple(w.shape)
        assert rel(g, w) < 2e-5, rel(g, w)
    with pytest.raises(K.KernelError):
        NativeResNetFPN("x101-64x4d", N, hw, "cuda", train=True)
