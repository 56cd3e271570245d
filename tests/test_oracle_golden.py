"""The CPU oracle (oracle/ssad_oracle.c) against the committed golden vectors.

The distill_* fixtures are outputs of the reference's own kernel bodies
host-compiled in the build container (tests/golden/make_golden.py); the conv
fixture is an independent float64 implementation; powsum is float64 numpy.
"""
import os

import numpy as np
import pytest

from oracle import oracle
from ssad_amd import synth
import make_golden as mg


def rel_close(a, b, rtol, atol):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), "NaN positions differ"
    ok = ~nan_a
    err = np.abs(a[ok] - b[ok])
    lim = atol + rtol * np.abs(b[ok])
    assert np.all(err <= lim), "max excess %g" % float(np.max(err - lim))


def test_sum128_order():
    rng = np.random.default_rng(3)
    x = rng.standard_normal(100003).astype(np.float32)
    # independent numpy statement of the same order
    lanes = np.zeros(128, np.float32)
    for j in range(128):
        acc = np.float32(0)
        for v in x[j::128]:
            acc = np.float32(acc + v)
        lanes[j] = acc
    for j in range(32):
        lanes[j] = np.float32(lanes[j] + np.float32(np.float32(lanes[j + 32] + lanes[j + 64]) + lanes[j + 96]))
    tot = np.float32(0)
    for j in range(32):
        tot = np.float32(tot + lanes[j])
    assert np.float32(oracle.sum128(x)) == tot
    assert oracle.sum128(np.zeros(0, np.float32)) == 0.0


def test_distill_small(golden_dir):
    g = np.load(os.path.join(golden_dir, "distill_small.npz"))
    x, q, lab = g["logits"], g["teacher"], g["labels"]
    for beta in (0.0, 0.3):
        for wp in (0.5, 123.4):
            for gamma, alpha in ((2.0, 0.5), (1.0, 0.25), (1.5, 0.75)):
                key = "b%g_n%g_g%g_a%g" % (beta, wp, gamma, alpha)
                kw = dict(gamma=gamma, alpha=alpha, beta=beta, num_classes=3,
                          ignored_label=-1)
                s128, s64, elems = oracle.distill_loss_forward(
                    x, q, lab, wp, scale=1.0, want_elems=True, **kw)
                rel_close(elems, g["loss_" + key], 2e-5, 1e-9)
                rel_close(s128, g["sum128_" + key], 2e-5, 1e-9)
                dx = oracle.distill_loss_backward(x, q, lab, wp, 0.7, scale=1.0, **kw)
                rel_close(dx, g["dx_" + key], 5e-5, 1e-9)
                # exact zeros where ignored
                ign = np.repeat(lab == -1, 3, axis=1)
                assert np.all(elems[ign] == 0) and np.all(dx[ign] == 0)
    kw = dict(gamma=2.0, alpha=0.5, beta=0.0, num_classes=3, ignored_label=2)
    _, _, elems = oracle.distill_loss_forward(x, q, lab, 3.0, want_elems=True, **kw)
    rel_close(elems, g["loss_ign2"], 2e-5, 1e-9)
    rel_close(oracle.distill_loss_backward(x, q, lab, 3.0, **kw), g["dx_ign2"], 5e-5, 1e-9)


def test_distill_scale_is_linear(golden_dir):
    g = np.load(os.path.join(golden_dir, "distill_small.npz"))
    x, q, lab = g["logits"], g["teacher"], g["labels"]
    kw = dict(gamma=2.0, alpha=0.5, beta=0.0, num_classes=3, ignored_label=-1)
    a, _, _ = oracle.distill_loss_forward(x, q, lab, 5.0, scale=1.0, **kw)
    b, _, _ = oracle.distill_loss_forward(x, q, lab, 5.0, scale=0.125, **kw)
    assert np.float32(a * np.float32(0.125)) == b
    da = oracle.distill_loss_backward(x, q, lab, 5.0, scale=1.0, **kw)
    db = oracle.distill_loss_backward(x, q, lab, 5.0, scale=0.125, **kw)
    assert np.array_equal(da * np.float32(0.125), db)


def test_distill_edges(golden_dir):
    g = np.load(os.path.join(golden_dir, "distill_edges.npz"))
    x, q, lab = g["logits"], g["teacher"], g["labels"]
    for beta in (0.0, 1.0):
        kw = dict(gamma=2.0, alpha=0.5, beta=beta, num_classes=1, ignored_label=-1)
        _, _, elems = oracle.distill_loss_forward(x, q, lab, 10.0, want_elems=True, **kw)
        rel_close(elems, g["loss_b%g" % beta], 2e-5, 1e-12)
        dx = oracle.distill_loss_backward(x, q, lab, 10.0, **kw)
        rel_close(dx, g["dx_b%g" % beta], 5e-5, 1e-12)
        # q == 0 and q == 1 rows are NaN even with beta == 0 and when ignored
        assert np.all(np.isnan(elems[0, 0, :, 0])) and np.all(np.isnan(elems[0, 0, :, 5]))


def test_distill_cfg1(golden_dir):
    """BASELINE config 1: one synthetic FPN level, H=W=64, A=9, C=80."""
    g = np.load(os.path.join(golden_dir, "distill_cfg1.npz"))
    N, A, C, H, W = [int(v) for v in g["shape"]]
    x, q, lab = synth.distill_inputs(np.random.default_rng(int(g["seed"])), N, A, C, H, W)
    pw32, pw64 = oracle.pow_sum([q], 1.8)
    assert np.float32(pw32) == g["normalizer_powsum"]
    idx = g["sample_idx"]
    for beta in (0.0, 0.3):
        for wp_name, wp in (("ps", float(pw32)), ("fix", 123.4)):
            key = "b%g_%s" % (beta, wp_name)
            kw = dict(gamma=2.0, alpha=0.5, beta=beta, num_classes=C, ignored_label=-1)
            s128, s64, elems = oracle.distill_loss_forward(x, q, lab, wp, want_elems=True, **kw)
            rel_close(elems.ravel()[idx], g["loss_s_" + key], 2e-5, 1e-12)
            rel_close(s128, g["sum128_" + key], 1e-5, 0)
            rel_close(s64, g["sum64_" + key], 1e-6, 0)
            dx = oracle.distill_loss_backward(x, q, lab, wp, **kw)
            rel_close(dx.ravel()[idx], g["dx_s_" + key], 5e-5, 1e-12)
            rel_close(np.sum(np.abs(dx.astype(np.float64))), g["sumabs_dx_" + key], 1e-6, 0)


def test_powsum(golden_dir):
    g = np.load(os.path.join(golden_dir, "powsum.npz"))
    arrs = [g["in%d" % i] for i in range(5)]
    for power in (1.8, 1.0, 2.0, 0.5):
        s32, s64 = oracle.pow_sum(arrs, power)
        rel_close(s64, g["sum_p%g" % power], 1e-6, 0)
        rel_close(s32, g["sum_p%g" % power], 1e-5, 0)
    s32, _ = oracle.pow_sum([np.zeros(0, np.float32)], 1.8)
    assert s32 == 0.0


@pytest.mark.parametrize("case", mg.CONV_CASES, ids=[c[0] for c in mg.CONV_CASES])
def test_conv_oracle_vs_float64(golden_dir, case):
    g = np.load(os.path.join(golden_dir, "conv_small.npz"))
    name = case[0]
    seed, N, Cin, M, H, W = [int(v) for v in g[name + "_dims"]]
    X, Wt, b, dY = mg.conv_case_inputs(seed, N, Cin, M, H, W)
    Y = oracle.conv_forward(X, Wt, b)
    dW, db, dX = oracle.conv_backward(X, Wt, dY)
    scale = lambda a: float(np.max(np.abs(a)))
    for key, arr in (("Y", Y), ("dW", dW), ("dX", dX)):
        ref = g["%s_%s" % (name, key)]
        got = arr.ravel()[g["%s_%s_idx" % (name, key)]]
        rel_close(got, ref, 1e-4, 1e-5 * scale(ref))
    rel_close(db, g[name + "_db"], 1e-4, 1e-5 * scale(g[name + "_db"]))


def test_relu_sigmoid_sgd():
    rng = np.random.default_rng(9)
    x = rng.standard_normal(1000).astype(np.float32)
    y = oracle.relu(x)
    assert np.array_equal(y, np.maximum(x, 0))
    dy = rng.standard_normal(1000).astype(np.float32)
    assert np.array_equal(oracle.relu_grad(y, dy), np.where(y > 0, dy, 0))
    rel_close(oracle.sigmoid(x), 1 / (1 + np.exp(-x.astype(np.float64))), 1e-6, 0)
    w = rng.standard_normal(100).astype(np.float32)
    gr = rng.standard_normal(100).astype(np.float32)
    m = rng.standard_normal(100).astype(np.float32)
    w2, g2, m2 = oracle.sgd_update(w, gr, m, 0.01, 0.9, 1e-4, is_bias=False)
    gg = gr + np.float32(1e-4) * w
    mm = np.float32(0.01) * gg + np.float32(0.9) * m
    rel_close(m2, mm, 1e-6, 1e-8)
    rel_close(w2, w - mm, 1e-6, 1e-8)
    w3, g3, m3 = oracle.sgd_update(w, gr, m, 0.01, 0.9, 1e-4, is_bias=True)
    rel_close(m3, np.float32(0.01) * (2 * gr) + np.float32(0.9) * m, 1e-6, 1e-8)


def test_focal_loss_oracle_vs_reference_kernels(golden_dir):
    """Row f2: SigmoidFocalLoss (sigmoid_focal_loss_op.cu:26-109)."""
    g = np.load(os.path.join(golden_dir, "focal_smoothl1.npz"))
    x, lab = g["logits"], g["labels"]
    C = 5
    for wp in (0.5, 37.0):
        for gamma, alpha in ((2.0, 0.25), (1.0, 0.5), (1.5, 0.75)):
            key = "n%g_g%g_a%g" % (wp, gamma, alpha)
            s128, s64, elems = oracle.focal_loss_forward(x, lab, wp, gamma=gamma, alpha=alpha,
                                                         num_classes=C, want_elems=True)
            rel_close(elems, g["fl_" + key], 2e-5, 1e-9)
            rel_close(s128, np.float32(oracle.sum128(g["fl_" + key])), 2e-5, 1e-9)
            dx = oracle.focal_loss_backward(x, lab, wp, 0.7, gamma=gamma, alpha=alpha, num_classes=C)
            rel_close(dx, g["fdx_" + key], 5e-5, 1e-9)
            ign = np.repeat(lab == -1, C, axis=1)
            assert np.all(elems[ign] == 0) and np.all(dx[ign] == 0)
    _, _, e = oracle.focal_loss_forward(g["e_logits"], g["e_labels"], 4.0, gamma=2.0, alpha=0.25,
                                        num_classes=3, want_elems=True)
    rel_close(e, g["e_fl"], 2e-5, 1e-12)
    rel_close(oracle.focal_loss_backward(g["e_logits"], g["e_labels"], 4.0, gamma=2.0, alpha=0.25,
                                         num_classes=3), g["e_fdx"], 5e-5, 1e-12)


def test_select_smooth_l1_oracle_vs_reference_kernels(golden_dir):
    """Row f2: SelectSmoothL1Loss (select_smooth_l1_loss_op.cu:23-86)."""
    g = np.load(os.path.join(golden_dir, "focal_smoothl1.npz"))
    Yh, Y, L = g["Y_hat"], g["Y"], g["L"]
    assert Y.shape[0] > 10
    for S in (0.5, float(L.shape[0])):
        for beta in (0.11, 1.0):
            key = "s%g_b%g" % (S, beta)
            s128, s64 = oracle.select_smooth_l1_forward(Yh, Y, L, S, beta=beta, scale=1.0)
            rel_close(s64, np.sum(g["sl_" + key].astype(np.float64)), 1e-6, 0)
            rel_close(s128, np.float32(oracle.sum128(g["sl_" + key])), 1e-5, 0)
            dy = oracle.select_smooth_l1_backward(Yh, Y, L, S, 0.7, beta=beta, scale=0.125)
            rel_close(dy, g["sdy_" + key], 1e-6, 0)
    # no foreground boxes: zero loss, zero gradient (.cu:101-105,146-149)
    e = np.zeros((0, 4), np.float32)
    assert oracle.select_smooth_l1_forward(Yh, e, e, 3.0)[1] == 0.0
    assert not oracle.select_smooth_l1_backward(Yh, e, e, 3.0).any()


# ---------------------------------------------------------------------------
# The convolution oracle PINNED by the reference's own compiled CPU operators
# ---------------------------------------------------------------------------
# tests/golden/conv_ref.npz holds outputs of ConvOp<float, CPUContext> /
# ConvGradientOp<float, CPUContext> (caffe2/operators/conv_op_impl.h + utils/math_cpu.cc)
# built from /root/reference by oracle/build_ref_conv.sh and run on seeded inputs
# (tests/golden/make_golden.py:make_conv_ref).

def _ref_cases():
    cases = [(n, N, Ci, M, H, W, 3, 1, 1, 1) for (n, N, Ci, M, H, W) in mg.CONV_CASES] + list(mg.CONV_REF_GEOMS)
    return cases


@pytest.mark.parametrize("case", [c for c in _ref_cases() if c[-1] == 1], ids=lambda c: c[0])
def test_conv_oracle_vs_reference_operator_outputs(golden_dir, case):
    """oracle.conv_forward / conv_backward against the stored outputs of the reference's own
    operators: same algorithm (per-image im2col + GEMM, conv_op_impl.h:126-173), so the
    agreement is fp32 round-off of a differently ordered dot product -- 1e-5 of the tensor's
    scale -- for the 3x3 subnet cases and the backbone's 1x1 / strided / 7x7 geometries."""
    g = np.load(os.path.join(golden_dir, "conv_ref.npz"))
    name = case[0]
    seed, N, Cin, M, H, W, k, s, p, grp = [int(v) for v in g[name + "_dims"]]
    X, Wt, b, dY = mg.conv_ref_inputs(seed, N, Cin, M, H, W, k, s, p, grp)
    Y = oracle.conv_forward(X, Wt, b, kernel=k, stride=s, pad=p)
    dW, db, dX = oracle.conv_backward(X, Wt, dY, kernel=k, stride=s, pad=p)
    for key, arr in (("Y", Y), ("dW", dW), ("dX", dX)):
        ref = g["%s_%s" % (name, key)]
        got = arr.ravel()[g["%s_%s_idx" % (name, key)]]
        assert np.max(np.abs(got - ref)) <= 1e-5 * np.max(np.abs(ref)), (name, key)
    assert np.max(np.abs(db - g[name + "_db"])) <= 1e-5 * np.max(np.abs(g[name + "_db"]))


def test_conv_golden_is_what_the_reference_operators_compute(golden_dir):
    """Container only (needs oracle/_ref/libref_conv.so): the committed fixture is reproduced
    bit for bit by running the reference's operators again, and on fresh random shapes the
    oracle tracks them to fp32 round-off."""
    if oracle.load_ref_conv() is None:
        pytest.skip("oracle/_ref/libref_conv.so not built (needs /root/reference)")
    g = np.load(os.path.join(golden_dir, "conv_ref.npz"))
    for case in _ref_cases():
        name = case[0]
        seed, N, Cin, M, H, W, k, s, p, grp = [int(v) for v in g[name + "_dims"]]
        X, Wt, b, dY = mg.conv_ref_inputs(seed, N, Cin, M, H, W, k, s, p, grp)
        Y = oracle.ref_conv_forward(X, Wt, b, kernel=k, stride=s, pad=p, group=grp)
        dW, db, dX = oracle.ref_conv_backward(X, Wt, dY, kernel=k, stride=s, pad=p, group=grp)
        for key, arr in (("Y", Y), ("dW", dW), ("dX", dX)):
            assert np.array_equal(arr.ravel()[g["%s_%s_idx" % (name, key)]], g["%s_%s" % (name, key)]), (name, key)
        assert np.array_equal(db, g[name + "_db"])
    rng = np.random.default_rng(31)
    for _ in range(6):
        N, Cin, M = int(rng.integers(1, 3)), int(rng.integers(1, 40)), int(rng.integers(1, 40))
        H, W = int(rng.integers(3, 20)), int(rng.integers(3, 20))
        k, s = [(3, 1), (1, 1), (3, 2), (1, 2)][int(rng.integers(0, 4))]
        p = k // 2
        X, Wt, b, dY = mg.conv_ref_inputs(int(rng.integers(1 << 30)), N, Cin, M, H, W, k, s, p, 1)
        Y = oracle.conv_forward(X, Wt, b, kernel=k, stride=s, pad=p)
        Yr = oracle.ref_conv_forward(X, Wt, b, kernel=k, stride=s, pad=p)
        assert np.max(np.abs(Y - Yr)) <= 1e-5 * np.max(np.abs(Yr))
        got = oracle.conv_backward(X, Wt, dY, kernel=k, stride=s, pad=p)
        ref = oracle.ref_conv_backward(X, Wt, dY, kernel=k, stride=s, pad=p)
        for a, r in zip(got, ref):
            assert np.max(np.abs(a - r)) <= 1e-5 * max(np.max(np.abs(r)), 1e-30)
