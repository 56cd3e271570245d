"""The reference's OWN loss kernels on the GPU as the checker: oracle/_ref/libref_kernels_hip.so holds the __global__
kernels of sigmoid_adaptive_distillation_loss_op.cu:28-105, sigmoid_focal_loss_op.cu:26-109 and
select_smooth_l1_loss_op.cu:23-86 compiled by hipcc for gfx950 from the reference's text (oracle/Makefile `ref`,
oracle/ref_driver_hip.cc: the reference's loop macro and launch geometry, no definition supplied in CUDA's place).
It is built in the container that has /root/reference and travels to the GPU box as a binary; the tests skip when
it is absent.  Checked against it, live, on the same device buffers:

  * the committed golden fixtures (tests/golden/*.npz, made from the HOST compile of the same text): the two
    compiles of the reference agree, NaN positions included;
  * oracle/ssad_oracle.c (the CPU restatement) element by element;
  * this repo's kernels at BASELINE config 1 and at config 3's full size (bs 16, all five levels, 600 px) --
    a size the CPU oracle cannot finish in seconds, so until now only slices and invariants were checked there.

Tolerances: the reference kernels compute in float with double promotions (`1.`, `-1.*x`) and libdevice / ocml
transcendental functions; 1e-5 relative (floor 1e-7 of the tensor's scale) on gradients, 2e-6 on float64 sums."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import ssad_amd  # noqa: F401
from ssad_amd import synth
from oracle import oracle
from test_gpu_kernels import close, dev

pytestmark = pytest.mark.gpu

_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref",
                   "libref_kernels_hip.so")
DX_RTOL, DX_FLOOR = 1e-5, 1e-7


@pytest.fixture(scope="module")
def K():
    from ssad_amd import kernels
    kernels.lib()
    return kernels


@pytest.fixture(scope="module")
def R():
    if not os.path.exists(_SO):
        pytest.skip("oracle/_ref/libref_kernels_hip.so not built (needs /root/reference at build time)")
    L = C.CDLL(_SO)
    i32, f32, vp = C.c_int, C.c_float, C.c_void_p
    L.ref_hip_distill_loss.argtypes = [i32] * 5 + [vp] * 4 + [f32] * 3 + [i32, vp, vp]
    L.ref_hip_distill_grad.argtypes = [i32] * 5 + [vp] * 5 + [f32] * 3 + [i32, vp, vp]
    L.ref_hip_focal_loss.argtypes = [i32] * 4 + [vp] * 3 + [f32] * 2 + [i32, vp, vp]
    L.ref_hip_focal_grad.argtypes = [i32] * 4 + [vp] * 4 + [f32] * 2 + [i32, vp, vp]
    L.ref_hip_smoothl1.argtypes = [i32] * 5 + [vp] * 5 + [f32, vp]
    L.ref_hip_smoothl1_grad.argtypes = [i32] * 5 + [vp] * 5 + [f32, vp, f32, vp]
    return L


def _p(t):
    return C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ref_distill(R, x, q, lab, wp, dloss, *, gamma, alpha, beta, num_classes, ignored_label):
    """(per-element losses, dX before the operator's *scale pass) from the reference kernels, device tensors."""
    N, D, H, W = x.shape
    wp_t = torch.full((1,), float(wp), device="cuda") if not torch.is_tensor(wp) else wp.reshape(1).float()
    go = torch.full((1,), float(dloss), device="cuda")
    le, dx = torch.empty_like(x), torch.empty_like(x)
    assert R.ref_hip_distill_loss(N, D, H, W, ignored_label, _p(x), _p(q), _p(lab), _p(wp_t), gamma, alpha, beta,
                                  num_classes, _p(le), _st()) == 0
    assert R.ref_hip_distill_grad(N, D, H, W, ignored_label, _p(x), _p(q), _p(lab), _p(dx), _p(wp_t), gamma, alpha,
                                  beta, num_classes, _p(go), _st()) == 0
    torch.cuda.synchronize()
    return le, dx


def ref_focal(R, x, lab, wp, dloss, *, gamma, alpha, num_classes):
    N, D, H, W = x.shape
    wp_t = torch.full((1,), float(wp), device="cuda")
    go = torch.full((1,), float(dloss), device="cuda")
    le, dx = torch.empty_like(x), torch.empty_like(x)
    assert R.ref_hip_focal_loss(N, D, H, W, _p(x), _p(lab), _p(wp_t), gamma, alpha, num_classes, _p(le), _st()) == 0
    assert R.ref_hip_focal_grad(N, D, H, W, _p(x), _p(lab), _p(dx), _p(wp_t), gamma, alpha, num_classes, _p(go),
                                _st()) == 0
    torch.cuda.synchronize()
    return le, dx


def tclose(got, ref, rtol, floor_frac, what):
    """|got - ref| <= rtol * |ref| + floor_frac * max|ref| on device tensors; NaN positions must coincide."""
    got, ref = got.double(), ref.double()
    ng, nr = torch.isnan(got), torch.isnan(ref)
    assert bool((ng == nr).all()), "%s: NaN positions differ" % what
    ok = ~nr
    if not bool(ok.any()):
        return
    scale = float(ref[ok].abs().max())
    err = (got[ok] - ref[ok]).abs() - rtol * ref[ok].abs()
    worst = float(err.max())
    assert worst <= floor_frac * scale, "%s: worst excess %.3e (scale %.3e)" % (what, worst, scale)


# ---------------------------------------------------------------------------------------------------------------

def test_device_compile_reproduces_the_golden_fixtures(R, golden_dir):
    """tests/golden/distill_small.npz / distill_edges.npz were written by the HOST compile of the same reference
    text: the device compile gives the same per-element losses and gradients (NaN positions too)."""
    g = np.load(os.path.join(golden_dir, "distill_small.npz"))
    x, q, lab = dev(g["logits"]), dev(g["teacher"]), dev(g["labels"])
    n = 0
    for beta in (0.0, 0.3):
        for wp in (0.5, 123.4):
            for gamma, alpha in ((2.0, 0.5), (1.0, 0.25), (1.5, 0.75)):
                key = "b%g_n%g_g%g_a%g" % (beta, wp, gamma, alpha)
                le, dx = ref_distill(R, x, q, lab, wp, 0.7, gamma=gamma, alpha=alpha, beta=beta, num_classes=3,
                                     ignored_label=-1)
                close(le.cpu().numpy(), g["loss_" + key], 2e-6, 1e-7, "loss elements " + key)
                close(dx.cpu().numpy(), g["dx_" + key], DX_RTOL, DX_FLOOR, "dx " + key)
                n += 1
    assert n == 12
    e = np.load(os.path.join(golden_dir, "distill_edges.npz"))
    x, q, lab = dev(e["logits"]), dev(e["teacher"]), dev(e["labels"])
    for beta in (0.0, 1.0):
        le, dx = ref_distill(R, x, q, lab, 10.0, 1.0, gamma=2.0, alpha=0.5, beta=beta, num_classes=1,
                             ignored_label=-1)
        close(le.cpu().numpy(), e["loss_b%g" % beta], 2e-6, 1e-7, "edge loss elements")
        close(dx.cpu().numpy(), e["dx_b%g" % beta], DX_RTOL, DX_FLOOR, "edge dx")


def test_oracle_and_kernels_against_reference_kernels_cfg1(R, K):
    """BASELINE config 1 (N=2, A=9, C=80, 64x64): oracle/ssad_oracle.c element by element, then this repo's
    kernels, against the reference kernels run on the same buffers."""
    N, A, Cc, H, W = 2, 9, 80, 64, 64
    x, q, lab = synth.distill_inputs(np.random.default_rng(11), N, A, Cc, H, W)
    tx, tq, tl = dev(x), dev(q), dev(lab)
    for beta, wp, dloss in ((0.0, 123.4, 1.0), (0.3, 0.5, 0.7)):
        kw = dict(gamma=2.0, alpha=0.5, beta=beta, num_classes=Cc, ignored_label=-1)
        le, dx = ref_distill(R, tx, tq, tl, wp, dloss, **kw)
        # the CPU restatement
        _, s64, elems = oracle.distill_loss_forward(x, q, lab, wp, scale=1.0, want_elems=True, **kw)
        close(elems, le.cpu().numpy(), 2e-6, 1e-7, "oracle loss elements")
        close(oracle.distill_loss_backward(x, q, lab, wp, dloss, scale=1.0, **kw), dx.cpu().numpy(), DX_RTOL,
              DX_FLOOR, "oracle dx")
        ref_sum = float(le.double().sum())
        assert abs(s64 - ref_sum) <= 2e-6 * abs(ref_sum)
        # this repo's kernels (the operator's *scale pass folded in: compare at scale 1 and 0.125)
        norm = dev(np.array([wp], np.float32))
        for scale in (1.0, 0.125):
            loss = K.distill_loss_forward([(tx, tq, tl)], norm, scale=scale, **kw)
            assert abs(float(loss[0]) - scale * ref_sum) <= 2e-6 * abs(ref_sum)
            got = K.distill_loss_backward([(tx, tq, tl)], norm, dev(np.array([dloss], np.float32)), scale=scale,
                                          **kw)[0]
            tclose(got, dx * scale, DX_RTOL, DX_FLOOR, "kernel dx, scale %g" % scale)


def test_kernels_against_reference_kernels_at_config3_size(R, K):
    """BASELINE config 3's loss input -- bs 16, 600 px, all five levels (190 960 anchor positions x 720 logits per
    image) -- in one launch of this repo's kernels against the reference kernels level by level: every gradient
    element and the float64 sum of the reference's per-element losses.  Also the fused one-pass kernel the bench
    step runs (cls_losses_fused_kernel: distillation + focal, gradients summed)."""
    N, A, Cc = 16, 9, 80
    gen = torch.Generator(device="cuda").manual_seed(5)
    levels = []
    for (H, W) in synth.LEVEL_SHAPES_600:
        x = torch.randn((N, A * Cc, H, W), device="cuda", generator=gen) * 2 - 4
        q = torch.sigmoid(torch.randn((N, A * Cc, H, W), device="cuda", generator=gen) * 2 - 4).clamp_(1e-6, 1 - 1e-6)
        u = torch.rand((N, A, H, W), device="cuda", generator=gen)
        lab = torch.zeros((N, A, H, W), dtype=torch.int32, device="cuda")
        lab[u < 0.05] = -1
        fg = (u >= 0.05) & (u < 0.07)
        lab[fg] = torch.randint(1, Cc + 1, (int(fg.sum()),), device="cuda", generator=gen, dtype=torch.int32)
        levels.append((x, q, lab))
    norm = K.pow_sum([q for _, q, _ in levels], 1.8).reshape(1)
    kw = dict(gamma=2.0, alpha=0.5, beta=0.0, num_classes=Cc, ignored_label=-1)
    one = torch.ones(1, device="cuda")
    losses = K.distill_loss_forward(levels, norm, scale=1.0, **kw)
    dxs = K.distill_loss_backward(levels, norm, one, scale=1.0, **kw)
    fkw = dict(gamma=2.0, alpha=0.25, num_classes=Cc)
    fgn = torch.full((1,), 5000.0, device="cuda")
    fused_d, fused_f, fused_dx = K.cls_losses_fused(levels, norm, fgn, dict(scale=1.0, **kw), dict(scale=1.0, **fkw))
    for i, (x, q, lab) in enumerate(levels):
        le, dx = ref_distill(R, x, q, lab, norm, 1.0, **kw)
        ref_sum = float(le.double().sum())
        assert abs(float(losses[i]) - ref_sum) <= 2e-6 * abs(ref_sum), ("level", i, float(losses[i]), ref_sum)
        tclose(dxs[i], dx, DX_RTOL, DX_FLOOR, "distill dx level %d" % i)
        fle, fdx = ref_focal(R, x, lab, 5000.0, 1.0, **fkw)
        fsum = float(fle.double().sum())
        assert abs(float(fused_d[i]) - ref_sum) <= 2e-6 * abs(ref_sum)
        assert abs(float(fused_f[i]) - fsum) <= 2e-6 * abs(fsum)
        tclose(fused_dx[i], dx + fdx, DX_RTOL, 2 * DX_FLOOR, "fused dx level %d" % i)
        del le, dx, fle, fdx


def test_focal_and_smooth_l1_against_reference_kernels(R, K):
    """SigmoidFocalLoss and SelectSmoothL1Loss (row f2): the reference kernels on the device against this repo's
    kernels and the oracle."""
    rng = np.random.default_rng(3)
    N, A, Cc, H, W = 2, 9, 80, 20, 28
    x, _, lab = synth.distill_inputs(rng, N, A, Cc, H, W)
    tx, tl = dev(x), dev(lab)
    for gamma, alpha, wp, dloss in ((2.0, 0.25, 37.0, 1.0), (1.5, 0.4, 1.0, 0.3)):
        le, dx = ref_focal(R, tx, tl, wp, dloss, gamma=gamma, alpha=alpha, num_classes=Cc)
        _, s64, elems = oracle.focal_loss_forward(x, lab, wp, gamma=gamma, alpha=alpha, num_classes=Cc, scale=1.0,
                                                  want_elems=True)
        close(elems, le.cpu().numpy(), 2e-6, 1e-7, "oracle focal elements")
        close(oracle.focal_loss_backward(x, lab, wp, dloss, gamma=gamma, alpha=alpha, num_classes=Cc, scale=1.0),
              dx.cpu().numpy(), DX_RTOL, DX_FLOOR, "oracle focal dx")
        fg = dev(np.array([wp], np.float32))
        loss = K.focal_loss_forward([(tx, tl)], fg, gamma=gamma, alpha=alpha, num_classes=Cc, scale=1.0)
        ref_sum = float(le.double().sum())
        assert abs(float(loss[0]) - ref_sum) <= 2e-6 * abs(ref_sum)
        got = K.focal_loss_backward([(tx, tl)], fg, dev(np.array([dloss], np.float32)), gamma=gamma, alpha=alpha,
                                    num_classes=Cc, scale=1.0)[0]
        tclose(got, dx, DX_RTOL, DX_FLOOR, "kernel focal dx")
    # SelectSmoothL1Loss: Y_hat [N][4A][H][W], M selected boxes
    Yh = rng.standard_normal((N, 4 * A, H, W)).astype(np.float32)
    tg = synth.bbox_targets(rng, lab)
    Y, Lc = tg[0], tg[1]
    M = Y.shape[0]
    assert M > 0
    tYh, tY, tL = dev(Yh), dev(Y), dev(Lc)
    S = torch.full((1,), 7.0, device="cuda")
    for beta, scale, dloss in ((0.11, 1.0, 1.0), (1.0, 0.25, 0.6)):
        buf, dy = torch.zeros_like(tYh), torch.zeros_like(tYh)
        go = torch.full((1,), dloss, device="cuda")
        assert R.ref_hip_smoothl1(tYh.numel(), 4 * A, H, W, M, _p(tYh), _p(tY), _p(tL), _p(buf), _p(S), beta,
                                  _st()) == 0
        assert R.ref_hip_smoothl1_grad(tYh.numel(), 4 * A, H, W, M, _p(tYh), _p(tY), _p(tL), _p(dy), _p(go), scale,
                                       _p(S), beta, _st()) == 0
        torch.cuda.synchronize()
        ref_loss = float(buf.double().sum()) * scale            # the operator: Sum, then Scale by scale_
        _, o64 = oracle.select_smooth_l1_forward(Yh, Y, Lc, 7.0, beta=beta, scale=scale)
        assert abs(o64 - ref_loss) <= 2e-6 * abs(ref_loss)
        close(oracle.select_smooth_l1_backward(Yh, Y, Lc, 7.0, dloss, beta=beta, scale=scale), dy.cpu().numpy(),
              DX_RTOL, DX_FLOOR, "oracle smooth-L1 dY")
        got = K.select_smooth_l1_forward(tYh, tY, tL, S, beta=beta, scale=scale)
        assert abs(float(got) - ref_loss) <= 2e-6 * abs(ref_loss)
        gdy = K.select_smooth_l1_backward(tYh, tY, tL, S, go, beta=beta, scale=scale)
        tclose(gdy, dy, DX_RTOL, DX_FLOOR, "kernel smooth-L1 dY")
