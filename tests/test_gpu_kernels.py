"""GPU parity: the hand-written gfx950 kernels, called through the raw C-ABI
(include/ssad_kernels.h), against the CPU oracle and the golden fixtures.

Tolerances (BASELINE.json: loss/grad within 1e-4 rel, fp32):
  loss scalars            |a-b| <= 1e-4 |b|
  per-element dX          |a-b| <= 1e-4 |b| + 1e-6 max|b|   (the abs floor covers
                          elements where 1-exp(-DL) cancels in fp32 in BOTH the
                          reference and here)
  conv outputs / grads    |a-b| <= 1e-4 |b| + 1e-5 max|b|   (K = 2304..6480 fp32
                          accumulations)
NaN positions must coincide exactly.
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import oracle  # noqa: E402
from ssad_amd import synth  # noqa: E402
import make_golden as mg  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from ssad_amd import kernels
    kernels.lib()  # raises if the HIP extension is missing: no fallback
    return kernels


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def close(got, ref, rtol, afloor, what=""):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    ng, nr = np.isnan(got), np.isnan(ref)
    assert np.array_equal(ng, nr), "%s: NaN positions differ (%d vs %d)" % (what, ng.sum(), nr.sum())
    ok = ~nr
    if not ok.any():
        return
    scale = float(np.max(np.abs(ref[ok]))) if ok.any() else 0.0
    err = np.abs(got[ok] - ref[ok])
    lim = rtol * np.abs(ref[ok]) + afloor * scale
    bad = err > lim
    assert not bad.any(), "%s: %d/%d outside tol, worst err %.3e (ref %.3e, scale %.3e)" % (
        what, int(bad.sum()), bad.size, float(err[bad].max()),
        float(np.abs(ref[ok])[bad][np.argmax(err[bad])]), scale)


LOSS_RTOL, DX_RTOL, DX_FLOOR = 1e-4, 1e-4, 1e-6
CONV_RTOL, CONV_FLOOR = 1e-4, 1e-5


# ---------------------------------------------------------------------------
# SigmoidAdaptiveDistillLoss
# ---------------------------------------------------------------------------

def run_distill(K, levels_np, wp, dloss, **kw):
    levels = [(dev(x), dev(q), dev(g)) for (x, q, g) in levels_np]
    norm = dev(np.array([wp], np.float32))
    loss = K.distill_loss_forward(levels, norm, **kw).cpu().numpy()
    dl = dev(np.full(len(levels), dloss, np.float32))
    dxs = [d.cpu().numpy() for d in K.distill_loss_backward(levels, norm, dl, **kw)]
    return loss, dxs


def test_distill_small_golden(K, golden_dir):
    g = np.load(os.path.join(golden_dir, "distill_small.npz"))
    x, q, lab = g["logits"], g["teacher"], g["labels"]
    for beta in (0.0, 0.3):
        for wp in (0.5, 123.4):
            for gamma, alpha in ((2.0, 0.5), (1.0, 0.25), (1.5, 0.75)):
                key = "b%g_n%g_g%g_a%g" % (beta, wp, gamma, alpha)
                kw = dict(gamma=gamma, alpha=alpha, beta=beta, num_classes=3,
                          ignored_label=-1, scale=1.0)
                loss, dxs = run_distill(K, [(x, q, lab)], wp, 0.7, **kw)
                ref_sum = np.sum(g["loss_" + key].astype(np.float64))
                close(loss[0], ref_sum, LOSS_RTOL, 0, "loss " + key)
                close(dxs[0], g["dx_" + key], DX_RTOL, DX_FLOOR, "dx " + key)
                ign = np.repeat(lab == -1, 3, axis=1)
                assert np.all(dxs[0][ign] == 0)
    kw = dict(gamma=2.0, alpha=0.5, beta=0.0, num_classes=3, ignored_label=2, scale=1.0)
    loss, dxs = run_distill(K, [(x, q, lab)], 3.0, 1.0, **kw)
    close(loss[0], np.sum(g["loss_ign2"].astype(np.float64)), LOSS_RTOL, 0, "ign2")
    close(dxs[0], g["dx_ign2"], DX_RTOL, DX_FLOOR, "dx ign2")


def test_distill_edges_nan_parity(K, golden_dir):
    """teacher prob 0 / 1 -> NaN (even ignored, even beta=0); p underflow clamp."""
    g = np.load(os.path.join(golden_dir, "distill_edges.npz"))
    x, q, lab = g["logits"], g["teacher"], g["labels"]
    for beta in (0.0, 1.0):
        kw = dict(gamma=2.0, alpha=0.5, beta=beta, num_classes=1, ignored_label=-1, scale=1.0)
        # per-element check through the gradient; the forward sum is NaN
        loss, dxs = run_distill(K, [(x, q, lab)], 10.0, 1.0, **kw)
        assert np.isnan(loss[0])
        close(dxs[0], g["dx_b%g" % beta], DX_RTOL, DX_FLOOR, "edge dx")
        # forward per element: one element per launch would be slow; instead
        # check the non-NaN columns as their own tensors
        cols = [1, 2, 3, 4]
        xs, qs, ls = x[..., cols], q[..., cols], lab[..., cols]
        loss2, _ = run_distill(K, [(np.ascontiguousarray(xs), np.ascontiguousarray(qs),
                                    np.ascontiguousarray(ls))], 10.0, 1.0, **kw)
        ref = np.sum(g["loss_b%g" % beta][..., cols].astype(np.float64))
        close(loss2[0], ref, LOSS_RTOL, 0, "edge loss")


def test_distill_cfg1_golden_and_oracle(K, golden_dir):
    """BASELINE config 1: H=W=64, A=9, C=80, N=2."""
    g = np.load(os.path.join(golden_dir, "distill_cfg1.npz"))
    N, A, C, H, W = [int(v) for v in g["shape"]]
    x, q, lab = synth.distill_inputs(np.random.default_rng(int(g["seed"])), N, A, C, H, W)
    idx = g["sample_idx"]
    ps = K.pow_sum([dev(q)], 1.8).cpu().numpy()
    close(ps, g["normalizer_powsum_f64"], 1e-5, 0, "powsum normalizer")
    for beta in (0.0, 0.3):
        for wp_name, wp in (("ps", float(g["normalizer_powsum"])), ("fix", 123.4)):
            key = "b%g_%s" % (beta, wp_name)
            for scale in (1.0, 0.125):
                kw = dict(gamma=2.0, alpha=0.5, beta=beta, num_classes=C,
                          ignored_label=-1, scale=scale)
                loss, dxs = run_distill(K, [(x, q, lab)], wp, 1.0, **kw)
                close(loss[0], g["sum64_" + key] * scale, LOSS_RTOL, 0, "loss " + key)
                close(dxs[0].ravel()[idx], g["dx_s_" + key] * np.float32(scale),
                      DX_RTOL, DX_FLOOR, "dx samples " + key)
                close(np.sum(np.abs(dxs[0].astype(np.float64))),
                      g["sumabs_dx_" + key] * scale, 1e-5, 0, "sum|dx| " + key)
            # full tensor against the oracle
            ref = oracle.distill_loss_backward(x, q, lab, wp, gamma=2.0, alpha=0.5, beta=beta,
                                               num_classes=C, ignored_label=-1, scale=0.125)
            close(dxs[0], ref, DX_RTOL, DX_FLOOR, "dx full " + key)


def test_distill_multilevel_one_launch(K):
    """All five FPN levels (incl. the odd 5x7 map -> scalar path) in one launch."""
    rng = np.random.default_rng(42)
    N, A, C = 2, 9, 80
    shapes = [(20, 28), (10, 14), (5, 7), (3, 5), (1, 1)]
    levels = [synth.distill_inputs(rng, N, A, C, h, w) for (h, w) in shapes]
    kw = dict(gamma=2.0, alpha=0.5, beta=0.0, num_classes=C, ignored_label=-1, scale=0.5)
    loss, dxs = run_distill(K, levels, 77.0, 1.0, **kw)
    for i, (x, q, g) in enumerate(levels):
        _, s64, _ = oracle.distill_loss_forward(x, q, g, 77.0, **kw)
        close(loss[i], s64, LOSS_RTOL, 0, "level %d" % i)
        close(dxs[i], oracle.distill_loss_backward(x, q, g, 77.0, **kw), DX_RTOL, DX_FLOOR,
              "dx level %d" % i)


def test_distill_general_gamma_and_empty(K):
    rng = np.random.default_rng(1)
    x, q, g = synth.distill_inputs(rng, 1, 2, 5, 6, 8)
    kw = dict(gamma=1.7, alpha=0.3, beta=0.2, num_classes=5, ignored_label=-1, scale=2.0)
    loss, dxs = run_distill(K, [(x, q, g)], 0.1, 0.3, **kw)   # normalizer < 1 -> clamped to 1
    _, s64, _ = oracle.distill_loss_forward(x, q, g, 0.1, **kw)
    close(loss[0], s64, LOSS_RTOL, 0, "gamma 1.7")
    close(dxs[0], oracle.distill_loss_backward(x, q, g, 0.1, 0.3, **kw), DX_RTOL, DX_FLOOR, "dx")
    e = (np.zeros((0, 10, 4, 4), np.float32), np.zeros((0, 10, 4, 4), np.float32),
         np.zeros((0, 2, 4, 4), np.int32))
    loss, _ = run_distill(K, [e], 5.0, 1.0, **kw)
    assert loss[0] == 0.0


def test_distill_full_size_properties(K):
    """BASELINE size (bs=16, P3 80x112): size-independent properties --
    scale linearity, ignored labels give exact zeros, determinism, and a
    checksum against a chunked oracle on a slice."""
    N, A, C, H, W = 16, 9, 80, 80, 112
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((N, A * C, H, W), device="cuda", generator=gen) * 2 - 4
    q = torch.sigmoid(torch.randn((N, A * C, H, W), device="cuda", generator=gen) * 2 - 4)
    q = q.clamp_(1e-6, 1 - 1e-6)
    u = torch.rand((N, A, H, W), device="cuda", generator=gen)
    lab = torch.where(u < 0.05, -1, 0).to(torch.int32)
    norm = K.pow_sum([q], 1.8).reshape(1)
    kw = dict(gamma=2.0, alpha=0.5, beta=0.0, num_classes=C, ignored_label=-1)
    l1 = K.distill_loss_forward([(x, q, lab)], norm, scale=1.0, **kw)
    l2 = K.distill_loss_forward([(x, q, lab)], norm, scale=0.125, **kw)
    l3 = K.distill_loss_forward([(x, q, lab)], norm, scale=1.0, **kw)
    assert float(l1[0]) == float(l3[0]), "not deterministic"
    assert abs(float(l2[0]) - 0.125 * float(l1[0])) <= 1e-6 * abs(float(l1[0]))
    one = torch.ones(1, device="cuda")
    dx = K.distill_loss_backward([(x, q, lab)], norm, one, scale=1.0, **kw)[0]
    ign = (lab == -1).repeat_interleave(C, dim=1)
    assert bool((dx[ign] == 0).all())
    assert bool(torch.isfinite(dx).all())
    # slice check against the oracle: image 3 only (same normalizer)
    xs, qs, ls = x[3:4].cpu().numpy(), q[3:4].cpu().numpy(), lab[3:4].cpu().numpy()
    ref = oracle.distill_loss_backward(xs, qs, ls, float(norm[0]), scale=1.0, **kw)
    close(dx[3:4].cpu().numpy(), ref, DX_RTOL, DX_FLOOR, "full-size slice")
    # sum over images of per-image losses == whole loss (checksum of checksums)
    parts = [float(K.distill_loss_forward([(x[i:i + 1], q[i:i + 1], lab[i:i + 1])], norm,
                                          scale=1.0, **kw)[0]) for i in range(N)]
    assert abs(sum(parts) - float(l1[0])) <= 1e-5 * abs(float(l1[0]))


def test_ticket_launchers_heal_a_dirty_workspace(K):
    """ssad_pow_sum / ssad_cls_losses_fused finish their sums in the last-arriving workgroup through
    arrival counters in the caller's workspace; the launchers zero the counters themselves (ABI 3),
    so a workspace full of garbage -- never zeroed, or left dirty by an aborted launch -- gives the
    same bits as a clean one.  The *_prezeroed forms rely on the caller's zero fill and leave the
    counters zero for the next launch."""
    import ctypes as C
    L = K.lib()
    g = torch.Generator(device="cuda").manual_seed(5)
    xs = [torch.rand(n, device="cuda", generator=g) for n in (100000, 4097, 33)]
    want = K.pow_sum(xs, 1.8).clone()
    n = len(xs)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in xs])
    sizes = (C.c_int64 * n)(*[t.numel() for t in xs])
    nb = L.ssad_pow_sum_workspace_bytes(n)
    st = torch.cuda.current_stream().cuda_stream
    dirty = torch.full((nb,), 0xA5, dtype=torch.uint8, device="cuda")
    out = torch.full((), float("nan"), device="cuda")
    for _ in range(2):
        assert L.ssad_pow_sum(ptrs, sizes, n, 1.8, out.data_ptr(), dirty.data_ptr(), nb, st) == 0
        assert torch.equal(out, want)
        out.fill_(float("nan"))
    clean = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    for _ in range(3):          # every launch leaves the counters zero
        assert L.ssad_pow_sum_prezeroed(ptrs, sizes, n, 1.8, out.data_ptr(), clean.data_ptr(), nb, st) == 0
        assert torch.equal(out, want)
        out.fill_(float("nan"))
    # the fused classification losses: two levels
    levels = []
    for (h, w) in ((12, 20), (6, 10)):
        x = torch.randn((2, 720, h, w), device="cuda", generator=g) * 2 - 4
        q = torch.sigmoid(torch.randn((2, 720, h, w), device="cuda", generator=g) * 2 - 4).clamp(1e-6, 1 - 1e-6)
        t = torch.randint(-1, 81, (2, 9, h, w), device="cuda", generator=g, dtype=torch.int32)
        levels.append((x, q, t))
    norm, fg = torch.tensor([37.5], device="cuda"), torch.tensor([11.0], device="cuda")
    dkw = dict(gamma=2.0, alpha=0.5, beta=0.0, num_classes=80, scale=0.5)
    fkw = dict(gamma=2.0, alpha=0.25, num_classes=80, scale=1.0)
    dl0, fl0, dx0 = K.cls_losses_fused(levels, norm, fg, dkw, fkw)
    dl0, fl0, dx0 = dl0.clone(), fl0.clone(), [d.clone() for d in dx0]
    nb = L.ssad_cls_losses_fused_workspace_bytes(2)
    K._ws_cache[(torch.cuda.current_device(), st, "clsfused")] = torch.full((nb,), 0x5A, dtype=torch.uint8,
                                                                              device="cuda")
    dl1, fl1, dx1 = K.cls_losses_fused(levels, norm, fg, dkw, fkw)
    assert torch.equal(dl0, dl1) and torch.equal(fl0, fl1)
    assert all(torch.equal(a, b) for a, b in zip(dx0, dx1))


# ---------------------------------------------------------------------------
# PowSum
# ---------------------------------------------------------------------------

def test_powsum_golden(K, golden_dir):
    g = np.load(os.path.join(golden_dir, "powsum.npz"))
    arrs = [dev(g["in%d" % i].ravel()) for i in range(5)]
    for power in (1.8, 1.0, 2.0, 0.5):
        got = K.pow_sum(arrs, power).cpu().numpy()
        close(got, g["sum_p%g" % power], 1e-5, 0, "powsum %g" % power)


def test_powsum_many_inputs_and_specials(K):
    rng = np.random.default_rng(2)
    arrs = [rng.random(n).astype(np.float32) for n in (1, 7, 64, 1000, 5, 3, 129, 4097, 33, 2, 11)]
    got = K.pow_sum([dev(a) for a in arrs], 1.8).cpu().numpy()
    close(got, sum(np.sum(a.astype(np.float64) ** 1.8) for a in arrs), 1e-5, 0, "11 inputs")
    # zeros, negatives (NaN for non-integer power), unaligned views
    z = np.array([0.0, 0.5, 0.0, 2.0], np.float32)
    close(K.pow_sum([dev(z)], 1.8).cpu().numpy(), 0.5 ** 1.8 + 2.0 ** 1.8, 1e-6, 0, "zeros")
    neg = np.array([0.5, -0.5], np.float32)
    assert np.isnan(K.pow_sum([dev(neg)], 1.8).cpu().numpy())
    base = dev(rng.random(1001).astype(np.float32))
    view = base[1:]   # 4-byte aligned only
    close(K.pow_sum([view], 2.0).cpu().numpy(), float((view.double() ** 2).sum()), 1e-6, 0,
          "unaligned")
    assert float(K.pow_sum([dev(np.zeros(0, np.float32))], 1.8)) == 0.0


# ---------------------------------------------------------------------------
# elementwise
# ---------------------------------------------------------------------------

def test_elementwise(K):
    rng = np.random.default_rng(3)
    for n in (1, 5, 1023, 4096, 100003):
        x = rng.standard_normal(n).astype(np.float32)
        dy = rng.standard_normal(n).astype(np.float32)
        y = K.relu(dev(x)).cpu().numpy()
        assert np.array_equal(y, oracle.relu(x))
        assert np.array_equal(K.relu_grad(dev(y), dev(dy)).cpu().numpy(), oracle.relu_grad(y, dy))
        close(K.sigmoid(dev(x)).cpu().numpy(), oracle.sigmoid(x), 1e-6, 0, "sigmoid")
        xs = [rng.standard_normal(n).astype(np.float32) for _ in range(5)]
        ref = xs[0].copy()
        for a in xs[1:]:
            ref = ref + a
        assert np.array_equal(K.sum_n([dev(a) for a in xs]).cpu().numpy(), ref)
        for is_bias in (False, True):
            w, gr, m = (rng.standard_normal(n).astype(np.float32) for _ in range(3))
            tw, tg, tm = dev(w), dev(gr), dev(m)
            K.momentum_sgd_update_(tw, tg, tm, dev(np.array([0.01], np.float32)), 0.9, 1e-4, is_bias)
            rw, rg, rm = oracle.sgd_update(w, gr, m, 0.01, 0.9, 1e-4, is_bias)
            close(tw.cpu().numpy(), rw, 1e-6, 1e-7, "sgd w")
            close(tm.cpu().numpy(), rm, 1e-6, 1e-7, "sgd m")
            close(tg.cpu().numpy(), rg, 1e-6, 1e-7, "sgd g")


# ---------------------------------------------------------------------------
# conv3x3
# ---------------------------------------------------------------------------

def conv_all(K, Xs, Wt, b, dYs, relu=False):
    """forward, dgrad, wgrad through the raw launchers for a list of levels."""
    tX = [dev(X) for X in Xs]
    tW, tb = dev(Wt), dev(b) if b is not None else None
    M = Wt.shape[0]
    pf, pd = K.conv_pack_filter(tW)
    Y = K.conv3x3_forward(tX, pf, tb, M, relu=relu)
    tdY = [dev(d) for d in dYs]
    dX = K.conv3x3_forward(tdY, pd, None, Wt.shape[1])
    dW, db = K.conv3x3_wgrad(tX, tdY, M)
    return ([y.cpu().numpy() for y in Y], [d.cpu().numpy() for d in dX],
            dW.cpu().numpy(), db.cpu().numpy())


@pytest.mark.parametrize("case", mg.CONV_CASES, ids=[c[0] for c in mg.CONV_CASES])
def test_conv_golden_float64(K, golden_dir, case):
    g = np.load(os.path.join(golden_dir, "conv_small.npz"))
    name = case[0]
    seed, N, Cin, M, H, W = [int(v) for v in g[name + "_dims"]]
    X, Wt, b, dY = mg.conv_case_inputs(seed, N, Cin, M, H, W)
    Y, dX, dW, db = conv_all(K, [X], Wt, b, [dY])
    for key, arr in (("Y", Y[0]), ("dW", dW), ("dX", dX[0])):
        ref = g["%s_%s" % (name, key)]
        got = arr.ravel()[g["%s_%s_idx" % (name, key)]]
        close(got, ref, CONV_RTOL, CONV_FLOOR, name + " " + key)
    close(db, g[name + "_db"], CONV_RTOL, CONV_FLOOR, name + " db")


@pytest.mark.parametrize("shape", [
    (1, 8, 32, 1, 1), (2, 5, 7, 3, 4), (1, 16, 40, 9, 17), (2, 36, 256, 5, 7),
    (1, 256, 36, 10, 14), (1, 24, 64, 17, 33), (3, 8, 65, 2, 31), (1, 720, 256, 5, 7)],
    ids=lambda s: "N%d_C%d_M%d_%dx%d" % s)
def test_conv_vs_oracle_ragged(K, shape):
    N, Cin, M, H, W = shape
    rng = np.random.default_rng(sum(shape))
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((M, Cin, 3, 3)) * 0.05).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    dY = rng.standard_normal((N, M, H, W)).astype(np.float32)
    Y, dX, dW, db = conv_all(K, [X], Wt, b, [dY])
    rY = oracle.conv_forward(X, Wt, b)
    rdW, rdb, rdX = oracle.conv_backward(X, Wt, dY)
    close(Y[0], rY, CONV_RTOL, CONV_FLOOR, "Y")
    close(dX[0], rdX, CONV_RTOL, CONV_FLOOR, "dX")
    close(dW, rdW, CONV_RTOL, CONV_FLOOR, "dW")
    close(db, rdb, CONV_RTOL, CONV_FLOOR, "db")


def test_conv_multilevel_shared_filter(K):
    """Five pyramid levels through one launch; dW/db are the sum over levels
    (the autograd Sum of caffe2/python/core.py:706-741)."""
    rng = np.random.default_rng(8)
    N, Cin, M = 2, 32, 48
    shapes = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
    Xs = [rng.standard_normal((N, Cin, h, w)).astype(np.float32) for h, w in shapes]
    dYs = [rng.standard_normal((N, M, h, w)).astype(np.float32) for h, w in shapes]
    Wt = (rng.standard_normal((M, Cin, 3, 3)) * 0.05).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    Y, dX, dW, db = conv_all(K, Xs, Wt, b, dYs, relu=True)
    rdW = np.zeros_like(Wt)
    rdb = np.zeros_like(b)
    for i in range(len(shapes)):
        close(Y[i], oracle.relu(oracle.conv_forward(Xs[i], Wt, b)), CONV_RTOL, CONV_FLOOR, "Y%d" % i)
        w_, b_, x_ = oracle.conv_backward(Xs[i], Wt, dYs[i])
        close(dX[i], x_, CONV_RTOL, CONV_FLOOR, "dX%d" % i)
        rdW += w_
        rdb += b_
    close(dW, rdW, CONV_RTOL, CONV_FLOOR, "dW sum")
    close(db, rdb, CONV_RTOL, CONV_FLOOR, "db sum")


def test_conv_fused_relu_grad_mask(K):
    rng = np.random.default_rng(12)
    N, Cin, M, H, W = 2, 16, 24, 9, 11
    Yprev = rng.standard_normal((N, Cin, H, W)).astype(np.float32)   # forward output (post-relu source)
    Yprev = np.maximum(Yprev, 0)
    dY = rng.standard_normal((N, M, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((M, Cin, 3, 3)) * 0.05).astype(np.float32)
    _, pd = K.conv_pack_filter(dev(Wt))
    got = K.conv3x3_forward([dev(dY)], pd, None, Cin, mask_by=[dev(Yprev)])[0].cpu().numpy()
    _, _, rdX = oracle.conv_backward(Yprev, Wt, dY)
    close(got, oracle.relu_grad(Yprev, rdX), CONV_RTOL, CONV_FLOOR, "masked dX")


def test_conv_full_size_adjoint_identities(K):
    """BASELINE size (bs=16, 256->256 at P3+P4): <conv(X,W),dY> = <X,dgrad(dY)>
    = <W,wgrad(X,dY)> -- size-independent properties of the three kernels --
    plus a slice against the oracle."""
    gen = torch.Generator(device="cuda").manual_seed(9)
    N, C, M = 16, 256, 256
    shapes = [(80, 112), (40, 56)]
    Xs = [torch.randn((N, C, h, w), device="cuda", generator=gen) for h, w in shapes]
    dYs = [torch.randn((N, M, h, w), device="cuda", generator=gen) for h, w in shapes]
    Wt = torch.randn((M, C, 3, 3), device="cuda", generator=gen) * 0.02
    pf, pd = K.conv_pack_filter(Wt)
    Ys = K.conv3x3_forward(Xs, pf, None, M)
    dXs = K.conv3x3_forward(dYs, pd, None, C)
    dW, _ = K.conv3x3_wgrad(Xs, dYs, M, want_db=False)
    a = sum(float((y.double() * d.double()).sum()) for y, d in zip(Ys, dYs))
    b = sum(float((x.double() * d.double()).sum()) for x, d in zip(Xs, dXs))
    c = float((Wt.double() * dW.double()).sum())
    scale = sum(float((y.double().abs() * d.double().abs()).sum()) for y, d in zip(Ys, dYs))
    assert abs(a - b) <= 1e-5 * scale and abs(a - c) <= 1e-5 * scale, (a, b, c, scale)
    # slice: one image of the P4 level against the oracle
    x1 = Xs[1][5:6].cpu().numpy()
    ref = oracle.conv_forward(x1, Wt.cpu().numpy(), None)
    close(Ys[1][5:6].cpu().numpy(), ref, CONV_RTOL, CONV_FLOOR, "full-size slice")


# ---------------------------------------------------------------------------
# Row f2: SigmoidFocalLoss, SelectSmoothL1Loss, fused classification losses
# ---------------------------------------------------------------------------

def test_focal_loss_golden_and_oracle(K, golden_dir):
    g = np.load(os.path.join(golden_dir, "focal_smoothl1.npz"))
    x, lab = g["logits"], g["labels"]
    C = 5
    for wp in (0.5, 37.0):
        for gamma, alpha in ((2.0, 0.25), (1.0, 0.5), (1.5, 0.75)):
            key = "n%g_g%g_a%g" % (wp, gamma, alpha)
            kw = dict(gamma=gamma, alpha=alpha, num_classes=C, scale=0.5)
            fg = dev(np.array([wp], np.float32))
            loss = K.focal_loss_forward([(dev(x), dev(lab))], fg, **kw).cpu().numpy()
            close(loss[0], 0.5 * np.sum(g["fl_" + key].astype(np.float64)), LOSS_RTOL, 0, "focal " + key)
            dx = K.focal_loss_backward([(dev(x), dev(lab))], fg, dev(np.array([0.7], np.float32)),
                                       **kw)[0].cpu().numpy()
            close(dx, g["fdx_" + key] * np.float32(0.5), DX_RTOL, DX_FLOOR, "focal dx " + key)
    fg = dev(np.array([4.0], np.float32))
    kw = dict(gamma=2.0, alpha=0.25, num_classes=3, scale=1.0)
    xe, le = g["e_logits"], g["e_labels"]
    loss = K.focal_loss_forward([(dev(xe), dev(le))], fg, **kw).cpu().numpy()
    close(loss[0], np.sum(g["e_fl"].astype(np.float64)), LOSS_RTOL, 0, "focal extremes")
    dx = K.focal_loss_backward([(dev(xe), dev(le))], fg, dev(np.ones(1, np.float32)), **kw)[0]
    close(dx.cpu().numpy(), g["e_fdx"], DX_RTOL, DX_FLOOR, "focal dx extremes")


def test_select_smooth_l1_golden(K, golden_dir):
    g = np.load(os.path.join(golden_dir, "focal_smoothl1.npz"))
    Yh, Y, L = g["Y_hat"], g["Y"], g["L"]
    for S in (0.5, float(L.shape[0])):
        for beta in (0.11, 1.0):
            key = "s%g_b%g" % (S, beta)
            tS = dev(np.array([S], np.float32))
            loss = K.select_smooth_l1_forward(dev(Yh), dev(Y), dev(L), tS, beta=beta, scale=2.0)
            close(loss.cpu().numpy(), 2.0 * np.sum(g["sl_" + key].astype(np.float64)), 1e-5, 0, key)
            dy = K.select_smooth_l1_backward(dev(Yh), dev(Y), dev(L), tS,
                                             dev(np.array([0.7], np.float32)), beta=beta, scale=0.125)
            close(dy.cpu().numpy(), g["sdy_" + key], 1e-5, 1e-7, "dy " + key)
    e = dev(np.zeros((0, 4), np.float32))
    assert float(K.select_smooth_l1_forward(dev(Yh), e, e, dev(np.ones(1, np.float32)))) == 0.0
    assert not bool(K.select_smooth_l1_backward(dev(Yh), e, e, dev(np.ones(1, np.float32)),
                                                dev(np.ones(1, np.float32))).any())


def test_fused_cls_losses_equal_separate_ops(K):
    """distill + focal in one pass == the two operators + the autograd Sum."""
    rng = np.random.default_rng(77)
    N, A, C = 2, 9, 80
    shapes = [(10, 14), (5, 7), (3, 4)]
    levels_np = []
    for h, w in shapes:
        x, q, _ = synth.distill_inputs(rng, N, A, C, h, w)
        lab = np.zeros((N, A, h, w), np.int32)
        u = rng.random(lab.shape)
        lab[u < 0.05] = -1
        fgm = (u >= 0.05) & (u < 0.12)
        lab[fgm] = rng.integers(1, C + 1, size=int(fgm.sum()))
        levels_np.append((x, q, lab))
    dkw = dict(gamma=2.0, alpha=0.5, beta=0.0, num_classes=C, ignored_label=-1, scale=0.125)
    fkw = dict(gamma=2.0, alpha=0.25, num_classes=C, scale=0.125)
    norm, fgn = 41.5, 17.0
    levels = [(dev(x), dev(q), dev(g)) for x, q, g in levels_np]
    dl, fl, dxs = K.cls_losses_fused(levels, dev(np.array([norm], np.float32)),
                                     dev(np.array([fgn], np.float32)), dkw, fkw)
    for i, (x, q, g) in enumerate(levels_np):
        _, d64, _ = oracle.distill_loss_forward(x, q, g, norm, **dkw)
        _, f64, _ = oracle.focal_loss_forward(x, g, fgn, **fkw)
        close(dl[i].cpu().numpy(), d64, LOSS_RTOL, 0, "fused distill loss")
        close(fl[i].cpu().numpy(), f64, LOSS_RTOL, 0, "fused focal loss")
        ref = oracle.distill_loss_backward(x, q, g, norm, 1.0, **dkw) + \
            oracle.focal_loss_backward(x, g, fgn, 1.0, **fkw)
        close(dxs[i].cpu().numpy(), ref, DX_RTOL, 2 * DX_FLOOR, "fused dX")


# ---------------------------------------------------------------------------
# Winograd F(2x2,3x3) engine: same contract, same tolerance as the direct kernel
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("pairs", ["0", "1"])
def test_winograd_both_staging_geometries_forced(pairs):
    """The persistent kernel stages a work item as one 8 x 16 patch or as two 8 x 8 sub-patches (which may sit in
    different images; an odd count leaves an absent partner), chosen per level by the launcher.  The library reads
    SSAD_WINO_PAIRS once per process, so each forced geometry runs the Winograd parity tests in a child process."""
    import subprocess
    import sys
    if os.environ.get("SSAD_WINO_PAIRS_CHILD"):
        pytest.skip("child run")
    env = dict(os.environ, SSAD_WINO_PAIRS=pairs, SSAD_WINO_PAIRS_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "winograd_vs_oracle or winograd_multilevel_matches or winograd_persistent_short or "
                        "winograd_multiproblem"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("shape", [
    (1, 8, 128, 8, 16), (2, 16, 40, 9, 17), (2, 36, 256, 5, 7), (1, 256, 256, 10, 14),
    (1, 24, 130, 17, 33), (3, 8, 65, 2, 31), (1, 720, 256, 5, 7), (2, 256, 720, 3, 4)],
    ids=lambda s: "N%d_C%d_M%d_%dx%d" % s)
def test_winograd_vs_oracle(K, shape):
    N, Cin, M, H, W = shape
    rng = np.random.default_rng(1000 + sum(shape))
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((M, Cin, 3, 3)) * 0.05).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    dY = rng.standard_normal((N, M, H, W)).astype(np.float32)
    pf, pd = K.conv_wino_pack_filter(dev(Wt))
    Y = K.conv3x3_forward([dev(X)], pf, dev(b), M, wino=True)[0].cpu().numpy()
    close(Y, oracle.conv_forward(X, Wt, b), CONV_RTOL, CONV_FLOOR, "wino Y")
    dX = K.conv3x3_forward([dev(dY)], pd, None, Cin, wino=True)[0].cpu().numpy()
    close(dX, oracle.conv_backward(X, Wt, dY, want_db=False)[2], CONV_RTOL, CONV_FLOOR, "wino dX")
    # epilogues
    Yr = K.conv3x3_forward([dev(X)], pf, dev(b), M, relu=True, wino=True)[0].cpu().numpy()
    close(Yr, oracle.relu(oracle.conv_forward(X, Wt, b)), CONV_RTOL, CONV_FLOOR, "wino relu")
    Xp = np.maximum(X, 0)
    dXm = K.conv3x3_forward([dev(dY)], pd, None, Cin, mask_by=[dev(Xp)], wino=True)[0].cpu().numpy()
    close(dXm, oracle.relu_grad(Xp, oracle.conv_backward(Xp, Wt, dY, want_db=False)[2]),
          CONV_RTOL, CONV_FLOOR, "wino masked dX")


def test_winograd_multilevel_matches_direct_full_size(K):
    gen = torch.Generator(device="cuda").manual_seed(19)
    N, C, M = 4, 256, 256
    shapes = [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)]
    Xs = [torch.randn((N, C, h, w), device="cuda", generator=gen) for h, w in shapes]
    Wt = torch.randn((M, C, 3, 3), device="cuda", generator=gen) * 0.02
    b = torch.randn(M, device="cuda", generator=gen)
    pf, _ = K.conv_pack_filter(Wt, True, False)
    wf, _ = K.conv_wino_pack_filter(Wt, True, False)
    Yd = K.conv3x3_forward(Xs, pf, b, M, relu=True)
    Yw = K.conv3x3_forward(Xs, wf, b, M, relu=True, wino=True)
    for a, c in zip(Yd, Yw):
        close(c.cpu().numpy(), a.cpu().numpy(), CONV_RTOL, CONV_FLOOR, "wino vs direct")


@pytest.mark.parametrize("cin,cout", [(16, 32), (32, 144), (48, 16)])
def test_winograd_persistent_short_reductions(K, cin, cout):
    """More tiles than CUs with 1-3 channel chunks per tile: the persistent
    kernel's cross-tile staging / filter-ring hand-over on every chunk."""
    gen = torch.Generator(device="cuda").manual_seed(23 + cin)
    N, H, W = 16, 40, 56
    X = torch.randn((N, cin, H, W), device="cuda", generator=gen)
    Wt = torch.randn((cout, cin, 3, 3), device="cuda", generator=gen) * 0.05
    b = torch.randn(cout, device="cuda", generator=gen)
    pf, _ = K.conv_pack_filter(Wt, True, False)
    wf, _ = K.conv_wino_pack_filter(Wt, True, False)
    Yd = K.conv3x3_forward([X], pf, b, cout, relu=True)[0]
    Yw = K.conv3x3_forward([X], wf, b, cout, relu=True, wino=True)[0]
    close(Yw.cpu().numpy(), Yd.cpu().numpy(), CONV_RTOL, CONV_FLOOR, "wino persistent vs direct")


def test_winograd_multiproblem_masked_dgrad_full_size(K):
    """Four independent towers x five levels in one launch (the head pipeline's
    dgrad shape): per-problem filters, ReLU-gradient mask, 45-chunk reduction."""
    gen = torch.Generator(device="cuda").manual_seed(29)
    N, Cin, M = 2, 256, 720
    shapes = [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)]
    probs_w, probs_d = [], []
    for t in range(3):
        dYs = [torch.randn((N, M, h, w), device="cuda", generator=gen) for h, w in shapes]
        Xs = [torch.relu(torch.randn((N, Cin, h, w), device="cuda", generator=gen)) for h, w in shapes]
        Wt = torch.randn((M, Cin, 3, 3), device="cuda", generator=gen) * 0.02
        _, pd = K.conv_pack_filter(Wt, False, True)
        _, wd = K.conv_wino_pack_filter(Wt, False, True)
        o1 = [torch.empty_like(x) for x in Xs]
        o2 = [torch.empty_like(x) for x in Xs]
        probs_d.append(dict(xs=dYs, packed=pd, out=o1, mask_by=Xs))
        probs_w.append(dict(xs=dYs, packed=wd, out=o2, mask_by=Xs))
    K.conv3x3_forward_multi(probs_d, Cin)
    K.conv3x3_forward_multi(probs_w, Cin, wino=True)
    for pd_, pw_ in zip(probs_d, probs_w):
        for a, c in zip(pd_["out"], pw_["out"]):
            close(c.cpu().numpy(), a.cpu().numpy(), CONV_RTOL, CONV_FLOOR, "wino multi dgrad vs direct")


# ---------------------------------------------------------------------------
# Winograd F(3x3,2x2) filter gradient: same contract / tolerance as the direct wgrad
# ---------------------------------------------------------------------------

@pytest.fixture
def wgrad_engine(monkeypatch):
    def choose(name):
        monkeypatch.setenv("SSAD_WGRAD_ENGINE", name)
    return choose


@pytest.mark.parametrize("shape", [
    (1, 64, 64, 8, 16), (2, 16, 40, 9, 17), (2, 36, 256, 5, 7), (1, 256, 256, 10, 14),
    (1, 24, 130, 17, 33), (3, 8, 65, 2, 31), (1, 720, 256, 5, 7), (2, 256, 36, 13, 20)],
    ids=lambda s: "N%d_C%d_M%d_%dx%d" % s)
def test_winograd_wgrad_vs_oracle(K, wgrad_engine, shape):
    N, Cin, M, H, W = shape
    rng = np.random.default_rng(2000 + sum(shape))
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((M, Cin, 3, 3)) * 0.05).astype(np.float32)
    dY = rng.standard_normal((N, M, H, W)).astype(np.float32)
    ref_dW, ref_db, _ = oracle.conv_backward(X, Wt, dY, want_db=True)
    wgrad_engine("winograd")
    dW, db = K.conv3x3_wgrad([dev(X)], [dev(dY)], M)
    close(dW.cpu().numpy(), ref_dW, CONV_RTOL, CONV_FLOOR, "wino dW")
    close(db.cpu().numpy(), ref_db, CONV_RTOL, CONV_FLOOR, "db")
    # accumulate: dW += on top of a known tensor
    base = dev(np.full_like(ref_dW, 0.25))
    K.conv3x3_wgrad([dev(X)], [dev(dY)], M, dW=base, want_db=False, accumulate=True)
    close(base.cpu().numpy(), ref_dW + 0.25, CONV_RTOL, CONV_FLOOR, "wino dW accumulate")


def test_winograd_wgrad_multilevel_matches_direct_full_size(K, wgrad_engine):
    gen = torch.Generator(device="cuda").manual_seed(31)
    N, C, M = 4, 256, 256
    shapes = [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)]
    Xs = [torch.randn((N, C, h, w), device="cuda", generator=gen) for h, w in shapes]
    dYs = [torch.randn((N, M, h, w), device="cuda", generator=gen) for h, w in shapes]
    wgrad_engine("direct")
    dWd, _ = K.conv3x3_wgrad(Xs, dYs, M, want_db=False)
    dWd = dWd.clone()
    wgrad_engine("winograd")
    dWw, _ = K.conv3x3_wgrad(Xs, dYs, M, want_db=False)
    close(dWw.cpu().numpy(), dWd.cpu().numpy(), CONV_RTOL, CONV_FLOOR, "wino wgrad vs direct")
    # deterministic
    dW2, _ = K.conv3x3_wgrad(Xs, dYs, M, want_db=False)
    assert torch.equal(dW2, dWw)


# ---------------------------------------------------------------------------
# Split-operand filter gradient (conv3x3_wgrad_split.hip): the same contract at the same (unwidened) tolerance
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("shape", [
    (1, 64, 64, 8, 16), (2, 16, 40, 9, 17), (2, 36, 256, 5, 7), (1, 256, 256, 10, 14),
    (1, 24, 130, 17, 33), (3, 8, 65, 2, 31), (1, 720, 256, 5, 7), (2, 256, 36, 13, 20), (1, 70, 130, 1, 1),
    (2, 128, 128, 20, 28)],
    ids=lambda s: "N%d_C%d_M%d_%dx%d" % s)
def test_split_wgrad_vs_oracle(K, shape):
    N, Cin, M, H, W = shape
    rng = np.random.default_rng(2100 + sum(shape))
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((M, Cin, 3, 3)) * 0.05).astype(np.float32)
    dY = rng.standard_normal((N, M, H, W)).astype(np.float32)
    ref_dW, ref_db, _ = oracle.conv_backward(X, Wt, dY, want_db=True)
    dW, db = K.conv3x3_wgrad([dev(X)], [dev(dY)], M, split=True)
    close(dW.cpu().numpy(), ref_dW, CONV_RTOL, CONV_FLOOR, "split dW")
    close(db.cpu().numpy(), ref_db, CONV_RTOL, CONV_FLOOR, "db")
    base = dev(np.full_like(ref_dW, 0.25))
    K.conv3x3_wgrad([dev(X)], [dev(dY)], M, dW=base, want_db=False, accumulate=True, split=True)
    close(base.cpu().numpy(), ref_dW + 0.25, CONV_RTOL, CONV_FLOOR, "split dW accumulate")


@pytest.mark.parametrize("scales", [(1e-6, 1e4), (3e7, 1e-9), (1.0, 1.0)], ids=lambda s: "x%g_dy%g" % s)
def test_split_wgrad_extreme_magnitudes_and_wide_dynamic_range(K, scales):
    """The power-of-two scales come from the measured |max|: tiny gradients against large activations (and the
    reverse) must not lose anything, and elements 2^-12 below the tensor's largest must still count."""
    sx, sdy = scales
    rng = np.random.default_rng(77)
    N, Cin, M, H, W = 2, 64, 128, 12, 20
    X = (rng.standard_normal((N, Cin, H, W)) * sx).astype(np.float32)
    dY = (rng.standard_normal((N, M, H, W)) * sdy).astype(np.float32)
    X[:, ::2] *= np.float32(2.0 ** -12)          # half the channels far below the |max|
    dY[:, 1::2] *= np.float32(2.0 ** -12)
    Wt = np.zeros((M, Cin, 3, 3), np.float32)
    ref_dW, _, _ = oracle.conv_backward(X, Wt, dY, want_db=False)
    dW, _ = K.conv3x3_wgrad([dev(X)], [dev(dY)], M, want_db=False, split=True)
    got = dW.cpu().numpy()
    close(got, ref_dW, CONV_RTOL, CONV_FLOOR, "split dW, scaled")
    # the quiet block on its own scale: (even input channel, odd output channel) is 2^-24 of the loudest
    q_ref, q_got = ref_dW[1::2, ::2], got[1::2, ::2]
    assert np.abs(q_got - q_ref).max() <= 2e-3 * np.abs(q_ref).max(), np.abs(q_got - q_ref).max() / np.abs(q_ref).max()


def test_split_wgrad_multilevel_full_size_vs_winograd_and_deterministic(K, wgrad_engine):
    gen = torch.Generator(device="cuda").manual_seed(33)
    N, C, M = 4, 256, 256
    shapes = [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)]
    Xs = [torch.randn((N, C, h, w), device="cuda", generator=gen) for h, w in shapes]
    dYs = [torch.randn((N, M, h, w), device="cuda", generator=gen) for h, w in shapes]
    wgrad_engine("direct")
    dWd, dbd = K.conv3x3_wgrad(Xs, dYs, M)
    dWd, dbd = dWd.clone(), dbd.clone()
    dWs, dbs = K.conv3x3_wgrad(Xs, dYs, M, split=True)
    dWs = dWs.clone()
    close(dWs.cpu().numpy(), dWd.cpu().numpy(), CONV_RTOL, CONV_FLOOR, "split wgrad vs direct")
    assert torch.equal(dbs, dbd)
    dW2, _ = K.conv3x3_wgrad(Xs, dYs, M, want_db=False, split=True)
    assert torch.equal(dW2, dWs)


# ---------------------------------------------------------------------------
# The DEFAULT engine (Winograd forward / data gradient / filter gradient) at the headline
# size, bs 16 on P3 (80 x 112), against the oracle -- not against another HIP kernel
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("M", [256, 720])
def test_default_engine_headline_size_vs_oracle_and_adjoints(K, wgrad_engine, M):
    """BASELINE config 3's dominant launches (bs = 16, 256 -> 256 tower layer and 256 -> 720
    cls_pred at P3) on the engine the step uses by default:
      * forward and data gradient of one image against the oracle (the launch covers all 16);
      * the filter gradient against the oracle by linearity: with dY zero outside image `n0` the
        bs-16 launch must return exactly the oracle's one-image gradient;
      * the adjoint identities <conv(X,W),dY> = <X,dgrad(dY)> = <W,wgrad(X,dY)> on full random
        data, which tie all 16 images of the three kernels together."""
    wgrad_engine("winograd")
    gen = torch.Generator(device="cuda").manual_seed(40 + M)
    N, C, H, W = 16, 256, 80, 112
    n0 = 11
    X = torch.randn((N, C, H, W), device="cuda", generator=gen)
    dY = torch.randn((N, M, H, W), device="cuda", generator=gen)
    Wt = torch.randn((M, C, 3, 3), device="cuda", generator=gen) * 0.02
    b = torch.randn(M, device="cuda", generator=gen)
    wf, wd = K.conv_wino_pack_filter(Wt)
    Y = K.conv3x3_forward([X], wf, b, M, wino=True)[0]
    dX = K.conv3x3_forward([dY], wd, None, C, wino=True)[0]
    dW, db = K.conv3x3_wgrad([X], [dY], M)
    # adjoints (bias excluded: subtract it)
    Yl = Y.double() - b.double().view(1, M, 1, 1)
    a = float((Yl * dY.double()).sum())
    bb = float((X.double() * dX.double()).sum())
    c = float((Wt.double() * dW.double()).sum())
    scale = float((Yl.abs() * dY.double().abs()).sum())
    assert abs(a - bb) <= 1e-5 * scale and abs(a - c) <= 1e-5 * scale, (a, bb, c, scale)
    close(db.cpu().numpy(), dY.double().sum((0, 2, 3)).cpu().numpy(), CONV_RTOL, CONV_FLOOR, "db")
    # one image against the oracle
    x1, dy1, w_np = X[n0:n0 + 1].cpu().numpy(), dY[n0:n0 + 1].cpu().numpy(), Wt.cpu().numpy()
    close(Y[n0:n0 + 1].cpu().numpy(), oracle.conv_forward(x1, w_np, b.cpu().numpy()), CONV_RTOL, CONV_FLOOR,
          "headline fwd slice")
    ref_dW, ref_db, ref_dX = oracle.conv_backward(x1, w_np, dy1)
    close(dX[n0:n0 + 1].cpu().numpy(), ref_dX, CONV_RTOL, CONV_FLOOR, "headline dgrad slice")
    # filter gradient of the full launch with only image n0 contributing
    dY0 = torch.zeros_like(dY)
    dY0[n0].copy_(dY[n0])
    dW0, db0 = K.conv3x3_wgrad([X], [dY0], M)
    close(dW0.cpu().numpy(), ref_dW, CONV_RTOL, CONV_FLOOR, "headline wgrad (one contributing image)")
    close(db0.cpu().numpy(), ref_db, CONV_RTOL, CONV_FLOOR, "headline db")


def test_smooth_l1_levels_launcher_matches_per_level_calls(K):
    """ssad_select_smooth_l1_levels (every FPN level in one launch each: forward, finalize, zero
    fill, scatter; caller-provided workspace) against the per-level launchers and the oracle,
    with an empty level and list entries outside the map."""
    rng = np.random.default_rng(77)
    N, A = 2, 9
    shapes = [(10, 14), (5, 7), (3, 4)]
    preds = [rng.standard_normal((N, 4 * A, h, w)).astype(np.float32) for h, w in shapes]
    tg = []
    for l, (h, w) in enumerate(shapes):
        if l == 2:
            tg.append((np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32)))      # no foreground
            continue
        M = 23 + l
        flat = rng.choice(N * A * h * w, size=M, replace=False)        # distinct anchors
        n_, a_, y_, x_ = np.unravel_index(flat, (N, A, h, w))
        Lc = np.stack([n_, 4 * a_, y_, x_], 1).astype(np.float32)
        Lc[3] = [0, 0, h + 2, 1]          # outside the map: contributes nothing
        Lc[4] = [N, 0, 0, 0]              # image index out of range: skipped, not read
        tg.append(((rng.standard_normal((M, 4)) * 0.5).astype(np.float32), Lc))
    S = dev(np.array([41.0], np.float32))
    one = dev(np.array([1.0], np.float32))
    tp = [dev(p) for p in preds]
    tt = [(dev(y), dev(l)) for y, l in tg]
    losses, dps = K.select_smooth_l1_levels(tp, tt, S, one, beta=0.11, scale=0.5)
    for l in range(len(shapes)):
        Y, Lc = tg[l]
        ok = np.ones(len(Y), bool)
        if len(Y):
            ok = (Lc[:, 0] < N) & (Lc[:, 2] < shapes[l][0]) & (Lc[:, 3] < shapes[l][1])
        _, l64 = oracle.select_smooth_l1_forward(preds[l], Y[ok], Lc[ok], np.array([41.0], np.float32),
                                                 beta=0.11, scale=0.5)
        close(float(losses[l]), l64, LOSS_RTOL, 1e-9, "levels loss %d" % l)
        ref = oracle.select_smooth_l1_backward(preds[l], Y[ok], Lc[ok], np.array([41.0], np.float32), 1.0,
                                               beta=0.11, scale=0.5)
        close(dps[l].cpu().numpy(), ref, DX_RTOL, DX_FLOOR, "levels dY_hat %d" % l)


# ---------------------------------------------------------------------------
# HIP convolution engines against the REFERENCE operator's own outputs
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("engine", ["winograd", "direct"])
@pytest.mark.parametrize("case", mg.CONV_CASES, ids=[c[0] for c in mg.CONV_CASES])
def test_conv_engines_vs_reference_operator_golden(K, golden_dir, wgrad_engine, case, engine):
    """Forward, data gradient and filter gradient of both 3x3 engines against
    tests/golden/conv_ref.npz = outputs of the reference's compiled ConvOp / ConvGradientOp
    <float, CPUContext> on the same seeded inputs (1e-4 rel + 1e-5 of the tensor's scale)."""
    g = np.load(os.path.join(golden_dir, "conv_ref.npz"))
    name = case[0]
    seed, N, Cin, M, H, W, k, s, p, grp = [int(v) for v in g[name + "_dims"]]
    X, Wt, b, dY = mg.conv_ref_inputs(seed, N, Cin, M, H, W, k, s, p, grp)
    wino = engine == "winograd"
    wgrad_engine(engine)
    pf, pd = (K.conv_wino_pack_filter if wino else K.conv_pack_filter)(dev(Wt))
    Y = K.conv3x3_forward([dev(X)], pf, dev(b), M, wino=wino)[0].cpu().numpy()
    dX = K.conv3x3_forward([dev(dY)], pd, None, Cin, wino=wino)[0].cpu().numpy()
    dW, db = K.conv3x3_wgrad([dev(X)], [dev(dY)], M)
    for key, arr in (("Y", Y), ("dW", dW.cpu().numpy()), ("dX", dX)):
        ref = g["%s_%s" % (name, key)]
        close(arr.ravel()[g["%s_%s_idx" % (name, key)]], ref, CONV_RTOL, CONV_FLOOR, "%s %s %s" % (engine, name, key))
    close(db.cpu().numpy(), g[name + "_db"], CONV_RTOL, CONV_FLOOR, name + " db")


# ---------------------------------------------------------------------------
# Round 5: the split tail of the persistent Winograd kernel
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("case", [
    # (Cin, Cout, H, W, relu, masked, bias)            grid rounds at bs 16 on 256 CUs -> units per tail item
    (256, 256, 40, 56, True, False, True),             # res4 / P4: 560 pair items = 2.19 rounds -> 48 x 4 units
    (256, 256, 40, 56, False, True, False),            # ... its masked data-gradient form
    (512, 512, 20, 28, True, False, True),             # res5: 384 patch items = 1.5 rounds -> 128 x 2
    (128, 128, 80, 112, True, False, True),            # res3: 1120 = 4.4 rounds -> 96 x 2 (8 chunks: 4 each)
    (256, 720, 80, 112, False, False, True),           # cls_pred on P3: 6 channel blocks (the last 80 wide), 64 x 4 units
    (256, 256, 10, 14, True, False, True),             # P6: 64 items, no full round at all -> 64 x 4 units
], ids=lambda c: "%d_%d_%dx%d_%s%s" % (c[0], c[1], c[2], c[3], "relu" if c[4] else "lin", "_mask" if c[5] else ""))
def test_wino_split_tail_vs_unsplit_vs_oracle(K, case):
    """The tail items of the persistent grid cut along the input channels (conv3x3_winograd.hip, SPLIT): same
    results as the unsplit kernel to fp32 round-off (another summation order), bit-reproducible, and one image
    of the bs-16 launch against the oracle."""
    Cin, Cout, H, W, relu, masked, use_bias = case
    N, n0 = 16, 5
    gen = torch.Generator(device="cuda").manual_seed(Cin + Cout + H)
    X = torch.randn((N, Cin, H, W), device="cuda", generator=gen)
    Wt = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=gen) * float(1.0 / (3 * np.sqrt(Cin)))
    b = torch.randn(Cout, device="cuda", generator=gen) if use_bias else None
    mask = [torch.randn((N, Cout, H, W), device="cuda", generator=gen)] if masked else None
    wf, _ = K.conv_wino_pack_filter(Wt, True, False)
    L = K.lib()
    arr = K._conv_levels([X], [torch.empty((N, Cout, H, W), device="cuda")], mask)
    prev = L.ssad_conv_wino_split_tail(2)         # 2: every partial round (the default, 1, splits only launches without a full round)
    try:
        with_split = L.ssad_conv3x3_forward_wino_launches_for(arr, 1, Cout, Cin, 0)
        Ys = K.conv3x3_forward([X], wf, b, Cout, relu=relu, mask_by=mask, wino=True)[0].clone()
        Ys2 = K.conv3x3_forward([X], wf, b, Cout, relu=relu, mask_by=mask, wino=True)[0]
        assert torch.equal(Ys, Ys2)                                  # whoever arrives last: the same bits
        L.ssad_conv_wino_split_tail(0)
        assert L.ssad_conv3x3_forward_wino_launches_for(arr, 1, Cout, Cin, 0) == 1
        assert with_split == (2 if N * ((H + 7) // 8) * ((W + 7) // 8) * ((Cout + 127) // 128) >= 512 else 1)
        Yu = K.conv3x3_forward([X], wf, b, Cout, relu=relu, mask_by=mask, wino=True)[0]
    finally:
        L.ssad_conv_wino_split_tail(prev)
    d = (Ys - Yu).abs().max().item()
    assert d <= 2e-5 * Yu.abs().max().item(), d                      # round-off of another summation order
    assert not torch.equal(Ys, Yu) or Cin <= 32                      # ... and it IS another order
    ref = oracle.conv_forward(X[n0:n0 + 1].cpu().numpy(), Wt.cpu().numpy(), b.cpu().numpy() if use_bias else None)
    if relu:
        ref = np.maximum(ref, 0)
    if masked:
        ref = np.where(mask[0][n0:n0 + 1].cpu().numpy() > 0, ref, 0)
    close(Ys[n0:n0 + 1].cpu().numpy(), ref, CONV_RTOL, CONV_FLOOR, "split tail vs oracle")
    close(Yu[n0:n0 + 1].cpu().numpy(), ref, CONV_RTOL, CONV_FLOOR, "unsplit vs oracle")


# ---------------------------------------------------------------------------
# Round 5: Winograd F(2x4, 3x3), the forward engine of frozen (evaluated-only) networks
# ---------------------------------------------------------------------------

F24_RTOL, F24_FLOOR = 1e-4, 1e-5      # the direct kernel's bar, unwidened (round 5 ran this engine at a 2e-5 floor)
F24_REL_OVER = 1e-2                   # north_star's "1e-4 rel": asserted on every element >= 1e-2 of the tensor's scale


def rel_err_over(got, ref, frac=F24_REL_OVER):
    """max |got - ref| / |ref| over the elements with |ref| >= frac * max|ref| (and how many there are)"""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    big = np.abs(ref) >= frac * np.abs(ref).max()
    return float((np.abs(got - ref)[big] / np.abs(ref)[big]).max()), int(big.sum())


def close24(got, ref, what):
    close(got, ref, F24_RTOL, F24_FLOOR, what)
    rel, n = rel_err_over(got, ref)
    assert rel <= 1e-4, "%s: max relative error %.3e over the %d elements >= %g of the scale" % (what, rel, n, F24_REL_OVER)


# Reported, not widened: with K = 2304 products per output the F(2x4) engine's transform round-off
# (1.5-2.2e-6 of the scale) exceeds 1e-4 RELATIVE on elements down to 1e-2 of the scale (measured 1.9e-4 on MI355X,
# gpurun r6a / r6b); it meets the 1e-5-of-scale floor.  The split-operand engine passes the same assertion on the same
# shapes (test_split_engine_*), which is why the 720-wide prediction layer runs on it.
_F24_REL = pytest.mark.xfail(strict=False, reason="F(2x4) fp32 Winograd: max relative error ~1.9e-4 over elements >= 1e-2 "
                             "of the scale at K = 2304 (floor 1e-5 of the scale holds)")


@pytest.mark.parametrize("shape", [
    (1, 16, 128, 8, 16), (2, 36, 256, 5, 7), pytest.param((1, 256, 256, 10, 14), marks=_F24_REL), (1, 24, 130, 17, 33),
    (3, 40, 129, 2, 31), pytest.param((2, 256, 720, 3, 4), marks=_F24_REL), (2, 128, 128, 9, 12)],
    ids=lambda s: "N%d_C%d_M%d_%dx%d" % s)
def test_winograd24_vs_oracle(K, shape):
    """conv3x3_winograd24.hip against the oracle (conv_op_impl.h:31-202): bias, ReLU and Sigmoid epilogues,
    channel tails (Cin, Cout not multiples of 16 / 128), maps that need the scalar edge stores."""
    N, Cin, M, H, W = shape
    rng = np.random.default_rng(2400 + sum(shape))
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((M, Cin, 3, 3)) * 0.05).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    pf = K.conv_wino24_pack_filter(dev(Wt))
    ref = oracle.conv_forward(X, Wt, b)
    close24(K.conv3x3_forward_wino24([dev(X)], pf, dev(b), M)[0].cpu().numpy(), ref, "wino24 Y")
    close24(K.conv3x3_forward_wino24([dev(X)], pf, dev(b), M, relu=True)[0].cpu().numpy(), oracle.relu(ref), "wino24 relu")
    sig = 1.0 / (1.0 + np.exp(-ref.astype(np.float64)))
    close24(K.conv3x3_forward_wino24([dev(X)], pf, dev(b), M, sigmoid=True)[0].cpu().numpy(), sig, "wino24 sigmoid")
    close24(K.conv3x3_forward_wino24([dev(X)], pf, None, M)[0].cpu().numpy(), oracle.conv_forward(X, Wt, None), "wino24 no bias")


def test_winograd24_levels_full_size_vs_winograd22_and_oracle(K):
    """The teacher's tower layer at config 3's size (bs 16, all five levels in one call: both staging geometries):
    against the F(2x2) engine on every element (both are fp32 Winograd: agreement to ~3e-6 of the scale), one image of
    P3 against the oracle, bit-reproducible."""
    gen = torch.Generator(device="cuda").manual_seed(24)
    N, C, M = 16, 256, 256
    shapes = [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)]
    Xs = [torch.randn((N, C, h, w), device="cuda", generator=gen).clamp_(min=0) for h, w in shapes]
    Wt = torch.randn((M, C, 3, 3), device="cuda", generator=gen) * 0.02
    b = torch.randn(M, device="cuda", generator=gen)
    p24 = K.conv_wino24_pack_filter(Wt)
    p22, _ = K.conv_wino_pack_filter(Wt, True, False)
    Y24 = K.conv3x3_forward_wino24(Xs, p24, b, M, relu=True)
    Y22 = K.conv3x3_forward(Xs, p22, b, M, relu=True, wino=True)
    for a, c in zip(Y24, Y22):
        assert (a - c).abs().max().item() <= 1e-5 * c.abs().max().item()
    again = K.conv3x3_forward_wino24(Xs, p24, b, M, relu=True)
    assert all(torch.equal(a, c) for a, c in zip(Y24, again))
    n0 = 7
    ref = oracle.relu(oracle.conv_forward(Xs[0][n0:n0 + 1].cpu().numpy(), Wt.cpu().numpy(), b.cpu().numpy()))
    close24(Y24[0][n0:n0 + 1].cpu().numpy(), ref, "wino24 P3 slice")


@pytest.mark.parametrize("shape", [
    (1, 128, 128, 8, 16), (2, 256, 36, 5, 7), pytest.param((1, 256, 256, 10, 14), marks=_F24_REL), (1, 130, 24, 17, 33),
    pytest.param((2, 256, 720, 3, 4), marks=_F24_REL), (3, 129, 40, 2, 31)], ids=lambda s: "N%d_C%d_M%d_%dx%d" % s)
def test_winograd24_data_gradient_vs_oracle(K, shape):
    """The F(2x4) engine's data-gradient form (flipped + transposed pack, fused ReluGradient mask; SSAD_STUDENT_F24):
    dX of conv_op_impl.h:358-577 for a layer of Cin inputs and M outputs, unmasked and masked."""
    N, Cin, M, H, W = shape
    rng = np.random.default_rng(2450 + sum(shape))
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)          # doubles as the mask (the layer's ReLU'd input)
    Wt = (rng.standard_normal((M, Cin, 3, 3)) * 0.05).astype(np.float32)
    dY = rng.standard_normal((N, M, H, W)).astype(np.float32)
    _, pd = K.conv_wino24_pack_filter(dev(Wt), want_dgrad=True)
    dX = oracle.conv_backward(X, Wt, dY, want_db=False)[2]
    got = K.conv3x3_forward_wino24([dev(dY)], pd, None, Cin)[0].cpu().numpy()
    close24(got, dX, "wino24 dX")
    got = K.conv3x3_forward_wino24([dev(dY)], pd, None, Cin, mask_by=[dev(X)])[0].cpu().numpy()
    close24(got, np.where(X > 0, dX, 0), "wino24 masked dX")


def test_winograd24_data_gradient_full_size_vs_winograd22(K):
    """Tower data gradient at config 3's size (two filters x five levels in one call, masked) against the F(2x2)
    engine element by element, and bit-reproducible."""
    gen = torch.Generator(device="cuda").manual_seed(25)
    N, C = 16, 256
    shapes = [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)]
    dYs = [torch.randn((N, C, h, w), device="cuda", generator=gen) * 1e-3 for h, w in shapes]
    Ms = [torch.randn((N, C, h, w), device="cuda", generator=gen) for h, w in shapes]
    Wt = torch.randn((C, C, 3, 3), device="cuda", generator=gen) * 0.02
    _, d24 = K.conv_wino24_pack_filter(Wt, want_dgrad=True)
    _, d22 = K.conv_wino_pack_filter(Wt, False, True)
    A = K.conv3x3_forward_wino24(dYs, d24, None, C, mask_by=Ms)
    B = K.conv3x3_forward(dYs, d22, None, C, mask_by=Ms, wino=True)
    for a, c, m in zip(A, B, Ms):
        assert (a - c).abs().max().item() <= 1e-5 * c.abs().max().item()
        assert torch.equal(a == 0, (m <= 0) | (a == 0))
    again = K.conv3x3_forward_wino24(dYs, d24, None, C, mask_by=Ms)
    assert all(torch.equal(a, c) for a, c in zip(A, again))


# ---------------------------------------------------------------------------
# Round 6: the split-operand engine (fp32 operands as hi + lo fp16, three fp16 MFMAs per pair: conv3x3_split.hip)
# held to the DIRECT fp32 kernel's bar (CONV_FLOOR, unwidened) plus the relative-error assertion
# ---------------------------------------------------------------------------

def close_split(got, ref, what):
    close(got, ref, CONV_RTOL, CONV_FLOOR, what)
    rel, n = rel_err_over(got, ref)
    assert rel <= 1e-4, "%s: max relative error %.3e over the %d elements >= %g of the scale" % (what, rel, n, F24_REL_OVER)


@pytest.mark.parametrize("shape", [
    (1, 16, 128, 8, 16), (2, 36, 256, 5, 7), (1, 256, 256, 10, 14), (1, 24, 130, 17, 33), (3, 40, 129, 2, 31),
    (2, 256, 720, 3, 4), (2, 128, 128, 9, 12), (1, 8, 36, 20, 28), (2, 3, 5, 6, 6)], ids=lambda s: "N%d_C%d_M%d_%dx%d" % s)
def test_split_engine_vs_oracle(K, shape):
    """conv3x3_split.hip against the oracle (conv_op_impl.h:31-202): bias, ReLU, Sigmoid epilogues, channel tails
    (Cin not a multiple of 8 / 16, Cout not a multiple of 32 / 128), ragged maps."""
    N, Cin, M, H, W = shape
    rng = np.random.default_rng(2600 + sum(shape))
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((M, Cin, 3, 3)) * 0.05).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    pf = K.conv_split_pack_filter(dev(Wt))
    ref = oracle.conv_forward(X, Wt, b)
    close_split(K.conv3x3_forward_split([dev(X)], pf, dev(b), M)[0].cpu().numpy(), ref, "split Y")
    close_split(K.conv3x3_forward_split([dev(X)], pf, dev(b), M, relu=True)[0].cpu().numpy(), oracle.relu(ref), "split relu")
    sig = 1.0 / (1.0 + np.exp(-ref.astype(np.float64)))
    close_split(K.conv3x3_forward_split([dev(X)], pf, dev(b), M, sigmoid=True)[0].cpu().numpy(), sig, "split sigmoid")
    close_split(K.conv3x3_forward_split([dev(X)], pf, None, M)[0].cpu().numpy(), oracle.conv_forward(X, Wt, None), "split no bias")


@pytest.mark.parametrize("shape", [
    (1, 128, 128, 8, 16), (2, 256, 36, 5, 7), (1, 256, 256, 10, 14), (1, 130, 24, 17, 33), (2, 256, 720, 3, 4),
    (3, 129, 40, 2, 31)], ids=lambda s: "N%d_C%d_M%d_%dx%d" % s)
def test_split_engine_data_gradient_vs_oracle(K, shape):
    """The data-gradient form (flipped + transposed pack, fused ReluGradient mask): dX of conv_op_impl.h:358-577."""
    N, Cin, M, H, W = shape
    rng = np.random.default_rng(2650 + sum(shape))
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((M, Cin, 3, 3)) * 0.05).astype(np.float32)
    dY = rng.standard_normal((N, M, H, W)).astype(np.float32)
    _, pd = K.conv_split_pack_filter(dev(Wt), want_dgrad=True)
    dX = oracle.conv_backward(X, Wt, dY, want_db=False)[2]
    close_split(K.conv3x3_forward_split([dev(dY)], pd, None, Cin)[0].cpu().numpy(), dX, "split dX")
    got = K.conv3x3_forward_split([dev(dY)], pd, None, Cin, mask_by=[dev(X)])[0].cpu().numpy()
    close_split(got, np.where(X > 0, dX, 0), "split masked dX")


@pytest.mark.parametrize("xs,ws", [(1e-30, 1.0), (1e30, 1e-3), (1.0, 1e-25), (3e-39, 1.0), (1e18, 1e18), (65504.0, 65504.0)],
                         ids=lambda v: "%g" % v)
def test_split_engine_extreme_magnitudes(K, xs, ws):
    """Per-tensor power-of-two scales: tiny / huge / denormal-range operands keep the fp32 bar (the fp16 halves never
    see the tensors' magnitudes).  Reference in float64 (the products leave fp32's range in one case)."""
    rng = np.random.default_rng(77)
    N, Cin, M, H, W = 1, 32, 64, 9, 11
    X = (rng.standard_normal((N, Cin, H, W)) * xs).astype(np.float32)
    Wt = (rng.standard_normal((M, Cin, 3, 3)) * ws).astype(np.float32)
    ref = oracle.conv_forward(X.astype(np.float64), Wt.astype(np.float64), None)
    got = K.conv3x3_forward_split([dev(X)], K.conv_split_pack_filter(dev(Wt)), None, M)[0].cpu().numpy()
    assert np.all(np.isfinite(got))
    if float(np.abs(ref).max()) < 1e-37:            # the result itself is in fp32's denormal range: absolute bar
        assert np.abs(got - ref).max() <= 2e-45 + 1e-5 * np.abs(ref).max()
    else:
        close_split(got, ref, "split extreme %g x %g" % (xs, ws))


def test_split_engine_wide_dynamic_range_within_a_tensor(K):
    """What is NOT fp32-like, pinned: elements far below the tensor's |max| lose bits gracefully -- absolute error
    <= 2^-40 of |max| per operand, i.e. the output error stays under 1e-5 of the output scale; and a tensor with an
    Inf passes through unscaled (Inf / NaN where fp32 has them)."""
    rng = np.random.default_rng(78)
    N, Cin, M, H, W = 1, 16, 32, 8, 8
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    X[0, :, :4] *= 1e-7                                  # half the map 2^-23 below the other half
    Wt = (rng.standard_normal((M, Cin, 3, 3)) * 0.1).astype(np.float32)
    ref = oracle.conv_forward(X.astype(np.float64), Wt.astype(np.float64), None)
    pf = K.conv_split_pack_filter(dev(Wt))
    got = K.conv3x3_forward_split([dev(X)], pf, None, M)[0].cpu().numpy()
    close(got, ref, CONV_RTOL, CONV_FLOOR, "wide range, whole tensor")
    small = np.abs(got[0, :, :2] - ref[0, :, :2]).max() / np.abs(ref[0, :, :2]).max()
    assert small <= 1e-4, small                          # rows that see only the small half: still 1e-4 of THEIR scale
    X[0, 3, 5, 5] = np.inf
    got = K.conv3x3_forward_split([dev(X)], pf, None, M)[0].cpu().numpy()
    want = oracle.conv_forward(X, Wt, None)
    assert np.array_equal(np.isfinite(got), np.isfinite(want))


def test_split_engine_levels_full_size_vs_float64_and_winograd(K):
    """A tower layer at config 3's size (bs 16, five levels, one call): one image of P3 against a float64 convolution
    -- and the error of the Winograd F(2x4) fp32 engine on the same data beside it (this engine must be the more
    accurate one); every level against F(2x2) on all elements; bit-reproducible."""
    gen = torch.Generator(device="cuda").manual_seed(26)
    N, C, M = 16, 256, 256
    shapes = [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)]
    Xs = [torch.randn((N, C, h, w), device="cuda", generator=gen).clamp_(min=0) for h, w in shapes]
    Wt = torch.randn((M, C, 3, 3), device="cuda", generator=gen) * 0.02
    b = torch.randn(M, device="cuda", generator=gen)
    ps = K.conv_split_pack_filter(Wt)
    p24 = K.conv_wino24_pack_filter(Wt)
    p22, _ = K.conv_wino_pack_filter(Wt, True, False)
    Ys = K.conv3x3_forward_split(Xs, ps, b, M, relu=True)
    Y24 = K.conv3x3_forward_wino24(Xs, p24, b, M, relu=True)
    Y22 = K.conv3x3_forward(Xs, p22, b, M, relu=True, wino=True)
    for a, c in zip(Ys, Y22):
        assert (a - c).abs().max().item() <= 1e-5 * c.abs().max().item()
    again = K.conv3x3_forward_split(Xs, ps, b, M, relu=True)
    assert all(torch.equal(a, c) for a, c in zip(Ys, again))
    n0 = 7
    ref = np.maximum(oracle.conv_forward(Xs[0][n0:n0 + 1].double().cpu().numpy(), Wt.double().cpu().numpy(),
                                         b.double().cpu().numpy()), 0)
    scale = np.abs(ref).max()
    e_split = np.abs(Ys[0][n0:n0 + 1].double().cpu().numpy() - ref).max() / scale
    e_24 = np.abs(Y24[0][n0:n0 + 1].double().cpu().numpy() - ref).max() / scale
    e_22 = np.abs(Y22[0][n0:n0 + 1].double().cpu().numpy() - ref).max() / scale
    print("max error / scale vs float64: split %.3e, F(2x4) %.3e, F(2x2) %.3e" % (e_split, e_24, e_22))
    # (fp32 accumulation over K = 2304 non-negative terms is what is left: the direct fp32 kernel measures the same)
    assert e_split <= 2e-6 and e_split <= 1.05 * e_24, (e_split, e_24, e_22)
    close_split(Ys[0][n0:n0 + 1].cpu().numpy(), ref, "split P3 slice vs float64")
