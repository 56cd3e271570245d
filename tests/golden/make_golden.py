#!/usr/bin/env python3
"""Generate the committed golden fixtures in tests/golden/.

Runs ONLY in the build container (needs /root/reference to build
oracle/_ref/libref_kernels.so = the reference's own kernel bodies compiled
for the host; see oracle/ref_driver.cc).  The fixtures hold inputs (or the
seed that regenerates them) and the outputs of the reference kernels; no
reference source text is stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixtures:
  distill_small.npz   N=1,A=2,C=3,H=4,W=5; beta in {0, 0.3}; every label kind;
                      per-element loss, dX, scalar (reference kernels)
  distill_cfg1.npz    BASELINE config 1 (N=2,A=9,C=80,H=W=64, seed 0): scalar
                      loss (128-lane order and float64), sum|dX|, 4096 sampled
                      (index, loss_i, dX_i) triples (reference kernels)
  distill_edges.npz   x in {-100,-30,-5,0,5,30,100} x q in {0,1e-30,1e-6,.5,
                      1-1e-6,1} x beta in {0,1}, incl. NaN positions
  focal_smoothl1.npz  SigmoidFocalLoss / SelectSmoothL1Loss per-element outputs of the
                      reference kernels (row f2)
  powsum.npz          5 random tensors, power 1.8: float64 numpy answer
  conv_small.npz      3x3 convs fwd/bwd, float64 torch (independent impl.)
  conv_ref.npz        the same cases + the backbone's geometries (1x1, strided, 7x7, grouped)
                      through the reference's own compiled Conv / ConvGradient CPU operators
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import oracle  # noqa: E402
import ssad_amd  # noqa: E402
from ssad_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def ref_fwd_bwd(x, q, lab, wp, gamma, alpha, beta, C, ign, dloss=1.0):
    le = oracle.ref_distill_loss_elems(x, q, lab, wp, gamma=gamma, alpha=alpha,
                                       beta=beta, num_classes=C, ignored_label=ign)
    dx = oracle.ref_distill_grad_elems(x, q, lab, wp, dloss, gamma=gamma,
                                       alpha=alpha, beta=beta, num_classes=C,
                                       ignored_label=ign)
    return le, dx


def make_small():
    rng = np.random.default_rng(7)
    N, A, C, H, W = 1, 2, 3, 4, 5
    x, q, lab = synth.distill_inputs(rng, N, A, C, H, W)
    lab = rng.integers(-1, C + 1, size=lab.shape).astype(np.int32)  # -1,0,1,2,3
    out = dict(logits=x, teacher=q, labels=lab)
    for beta in (0.0, 0.3):
        for wp in (0.5, 123.4):
            for gamma, alpha in ((2.0, 0.5), (1.0, 0.25), (1.5, 0.75)):
                le, dx = ref_fwd_bwd(x, q, lab, wp, gamma, alpha, beta, C, -1, dloss=0.7)
                key = "b%g_n%g_g%g_a%g" % (beta, wp, gamma, alpha)
                out["loss_" + key] = le
                out["dx_" + key] = dx
                out["sum128_" + key] = np.float32(oracle.sum128(le))
    # a different ignored_label value
    le, dx = ref_fwd_bwd(x, q, lab, 3.0, 2.0, 0.5, 0.0, C, 2)
    out["loss_ign2"] = le
    out["dx_ign2"] = dx
    np.savez_compressed(os.path.join(OUT, "distill_small.npz"), **out)


def make_cfg1():
    rng = np.random.default_rng(0)
    N, A, C, H, W = 2, 9, 80, 64, 64
    x, q, lab = synth.distill_inputs(rng, N, A, C, H, W)
    pw32, pw64 = oracle.pow_sum([q], 1.8)
    out = dict(seed=0, shape=np.array([N, A, C, H, W]), normalizer_powsum=np.float32(pw32),
               normalizer_powsum_f64=pw64)
    idx = np.random.default_rng(1).choice(x.size, 4096, replace=False).astype(np.int64)
    out["sample_idx"] = idx
    for beta in (0.0, 0.3):
        for wp_name, wp in (("ps", float(pw32)), ("fix", 123.4)):
            le, dx = ref_fwd_bwd(x, q, lab, wp, 2.0, 0.5, beta, C, -1)
            key = "b%g_%s" % (beta, wp_name)
            out["sum128_" + key] = np.float32(oracle.sum128(le))
            out["sum64_" + key] = np.sum(le.astype(np.float64))
            out["sumabs_dx_" + key] = np.sum(np.abs(dx.astype(np.float64)))
            out["loss_s_" + key] = le.ravel()[idx]
            out["dx_s_" + key] = dx.ravel()[idx]
    np.savez_compressed(os.path.join(OUT, "distill_cfg1.npz"), **out)


def make_edges():
    xs = np.array([-100, -30, -5, 0, 5, 30, 100], np.float32)
    qs = np.array([0, 1e-30, 1e-6, .5, 1 - 1e-6, 1], np.float32)
    X, Q = np.meshgrid(xs, qs, indexing="ij")
    # shape N=1, D=A*C with A=1, C=len(xs)*len(qs)... keep it 4-D: 1 x 1 x 7 x 6
    x = X.reshape(1, 1, 7, 6).astype(np.float32)
    q = Q.reshape(1, 1, 7, 6).astype(np.float32)
    lab = np.zeros((1, 1, 7, 6), np.int32)
    lab[0, 0, :, 3] = -1  # an ignored column: NaN must still propagate (0*NaN)
    out = dict(logits=x, teacher=q, labels=lab)
    for beta in (0.0, 1.0):
        le, dx = ref_fwd_bwd(x, q, lab, 10.0, 2.0, 0.5, beta, 1, -1)
        out["loss_b%g" % beta] = le
        out["dx_b%g" % beta] = dx
    np.savez_compressed(os.path.join(OUT, "distill_edges.npz"), **out)


def make_powsum():
    rng = np.random.default_rng(11)
    shapes = [(2, 18, 8, 12), (2, 18, 4, 6), (2, 18, 2, 3), (1, 5), (7,)]
    arrs = [np.clip(synth.sigmoid(rng.standard_normal(s) * 2 - 1), 1e-6, 1).astype(np.float32)
            for s in shapes]
    out = {"in%d" % i: a for i, a in enumerate(arrs)}
    for power in (1.8, 1.0, 2.0, 0.5):
        out["sum_p%g" % power] = sum(
            np.sum(np.power(a.astype(np.float64), power)) for a in arrs)
    np.savez_compressed(os.path.join(OUT, "powsum.npz"), **out)


def make_focal_smoothl1():
    """Row f2: SigmoidFocalLoss and SelectSmoothL1Loss from the reference kernels."""
    rng = np.random.default_rng(17)
    N, A, C, H, W = 2, 3, 5, 6, 7
    x, _, _ = synth.distill_inputs(rng, N, A, C, H, W)
    x = (x + 3.0).astype(np.float32)                 # both signs
    lab = rng.integers(-1, C + 1, size=(N, A, H, W)).astype(np.int32)
    out = dict(logits=x, labels=lab)
    for wp in (0.5, 37.0):
        for gamma, alpha in ((2.0, 0.25), (1.0, 0.5), (1.5, 0.75)):
            le, dx = oracle.ref_focal_elems(x, lab, wp, 0.7, gamma=gamma, alpha=alpha, num_classes=C)
            key = "n%g_g%g_a%g" % (wp, gamma, alpha)
            out["fl_" + key], out["fdx_" + key] = le, dx
    # extreme logits: log(max(p, FLT_MIN)) clamp and 1-p cancellation
    xe = np.array([-100, -30, -5, 0, 5, 30, 100], np.float32).reshape(1, 1, 1, 7)
    xe = np.repeat(xe, 3, axis=1)                    # A=1, C=3
    labe = np.array([[[[1, 2, 3, 0, -1, 1, 2]]]], np.int32)
    le, dx = oracle.ref_focal_elems(xe, labe, 4.0, 1.0, gamma=2.0, alpha=0.25, num_classes=3)
    out.update(e_logits=xe, e_labels=labe, e_fl=le, e_fdx=dx)
    # SelectSmoothL1Loss
    Yh = rng.standard_normal((N, 4 * A, H, W)).astype(np.float32)
    Y, L = synth.bbox_targets(rng, lab, 4 * A)
    out.update(Y_hat=Yh, Y=Y, L=L)
    for S in (0.5, float(L.shape[0])):
        for beta in (0.11, 1.0):
            buf, dy = oracle.ref_smoothl1_elems(Yh, Y, L, S, 0.7, beta=beta, norm=0.125)
            key = "s%g_b%g" % (S, beta)
            out["sl_" + key], out["sdy_" + key] = buf, dy
    np.savez_compressed(os.path.join(OUT, "focal_smoothl1.npz"), **out)


def conv_case_inputs(seed, N, Cin, M, H, W):
    """Seeded conv inputs shared by the fixture generator and the tests."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((M, Cin, 3, 3)) * 0.05).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    dY = rng.standard_normal((N, M, H, W)).astype(np.float32)
    return X, Wt, b, dY


CONV_CASES = [("c8m8", 2, 8, 8, 7, 9), ("c8m36", 2, 8, 36, 7, 9),
              ("c12m20", 3, 12, 20, 5, 7), ("c64m48", 1, 64, 48, 10, 14),
              ("c256m256", 1, 256, 256, 10, 14), ("c256m720", 2, 256, 720, 5, 7)]


def make_conv():
    """float64 torch conv2d + autograd = independent implementation.  Inputs
    are regenerated from the seed; outputs are stored as 2048 samples each."""
    import torch
    import torch.nn.functional as F
    out = {}
    for ci, (name, N, Cin, M, H, W) in enumerate(CONV_CASES):
        X, Wt, b, dY = conv_case_inputs(100 + ci, N, Cin, M, H, W)
        tx = torch.tensor(X, dtype=torch.float64, requires_grad=True)
        tw = torch.tensor(Wt, dtype=torch.float64, requires_grad=True)
        tb = torch.tensor(b, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(tx, tw, tb, stride=1, padding=1)
        y.backward(torch.tensor(dY, dtype=torch.float64))
        srng = np.random.default_rng(1000 + ci)
        for key, arr in (("Y", y.detach().numpy()), ("dW", tw.grad.numpy()),
                         ("dX", tx.grad.numpy())):
            k = min(2048, arr.size)
            idx = srng.choice(arr.size, k, replace=False).astype(np.int64)
            out["%s_%s_idx" % (name, key)] = idx
            out["%s_%s" % (name, key)] = arr.ravel()[idx]
        out[name + "_db"] = tb.grad.numpy()
        out[name + "_dims"] = np.array([100 + ci, N, Cin, M, H, W])
    np.savez_compressed(os.path.join(OUT, "conv_small.npz"), **out)


# geometries beyond the subnets' 3x3: the backbone's layers (row f1) -- pointwise, strided
# pointwise, 3x3 / stride 2 (P6, P7), the 7x7 / stride 2 stem, ResNeXt's grouped 3x3
# (name, N, Cin, M, H, W, kernel, stride, pad, group)
CONV_REF_GEOMS = [("k1s1", 2, 24, 40, 9, 13, 1, 1, 0, 1), ("k1s2", 2, 32, 16, 10, 14, 1, 2, 0, 1),
                  ("k3s2", 2, 16, 24, 10, 13, 3, 2, 1, 1), ("k7s2", 1, 3, 16, 21, 29, 7, 2, 3, 1),
                  ("k3g4", 2, 32, 32, 9, 11, 3, 1, 1, 4), ("k3g8s2", 1, 64, 64, 12, 10, 3, 2, 1, 8),
                  # pointwise layers with 16-pixel-multiple maps (what the GEMM kernels take)
                  ("k1s1p96", 2, 24, 40, 8, 12, 1, 1, 0, 1), ("k1s2p96", 2, 32, 16, 16, 24, 1, 2, 0, 1),
                  ("k1s1c160", 3, 160, 72, 4, 8, 1, 1, 0, 1),
                  # ResNeXt's grouped 3x3 layer at every channels-per-group width (4, 16, 32; 8 above)
                  ("k3g64c4", 1, 256, 256, 9, 20, 3, 1, 1, 64), ("k3g4c16s2", 2, 64, 64, 11, 18, 3, 2, 1, 4),
                  ("k3g2c32", 1, 64, 64, 10, 17, 3, 1, 1, 2), ("k3g2c32s2", 1, 64, 64, 16, 9, 3, 2, 1, 2),
                  # strided k x k with 4-pixel-multiple output maps (what the implicit GEMM takes)
                  ("k7s2p176", 1, 3, 16, 22, 32, 7, 2, 3, 1), ("k3s2p80", 2, 16, 24, 16, 20, 3, 2, 1, 1),
                  # the stem itself (3 -> 64 channels): two images, 22 x 36 outputs = partial 8 x 32 tiles of stem.hip
                  ("k7s2m64", 2, 3, 64, 44, 72, 7, 2, 3, 1)]


def conv_ref_inputs(seed, N, Cin, M, H, W, kernel=3, stride=1, pad=1, group=1):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((M, Cin // group, kernel, kernel)) * 0.05).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    oh, ow = (H + 2 * pad - kernel) // stride + 1, (W + 2 * pad - kernel) // stride + 1
    dY = rng.standard_normal((N, M, oh, ow)).astype(np.float32)
    return X, Wt, b, dY


def make_conv_ref():
    """conv_ref.npz: outputs of the REFERENCE'S OWN Conv / ConvGradient CPU operators
    (oracle/_ref/libref_conv.so, built from /root/reference by oracle/build_ref_conv.sh) on
    seeded inputs: the subnets' 3x3 cases of CONV_CASES and the backbone geometries of
    CONV_REF_GEOMS.  float32, up to 4096 sampled entries per tensor (indices stored)."""
    assert oracle.load_ref_conv() is not None, "make -C oracle refconv first"
    out = {}
    cases = [(n, N, Ci, M, H, W, 3, 1, 1, 1) for (n, N, Ci, M, H, W) in CONV_CASES] + CONV_REF_GEOMS
    for ci, (name, N, Cin, M, H, W, k, s, p, g) in enumerate(cases):
        seed = 100 + ci if ci < len(CONV_CASES) else 500 + ci
        X, Wt, b, dY = conv_ref_inputs(seed, N, Cin, M, H, W, k, s, p, g)
        Y = oracle.ref_conv_forward(X, Wt, b, kernel=k, stride=s, pad=p, group=g)
        dW, db, dX = oracle.ref_conv_backward(X, Wt, dY, kernel=k, stride=s, pad=p, group=g)
        srng = np.random.default_rng(2000 + ci)
        for key, arr in (("Y", Y), ("dW", dW), ("dX", dX)):
            kk = min(4096, arr.size)
            idx = np.sort(srng.choice(arr.size, kk, replace=False)).astype(np.int64)
            out["%s_%s_idx" % (name, key)] = idx
            out["%s_%s" % (name, key)] = arr.ravel()[idx].astype(np.float32)
        out[name + "_db"] = db
        out[name + "_dims"] = np.array([seed, N, Cin, M, H, W, k, s, p, g])
    np.savez_compressed(os.path.join(OUT, "conv_ref.npz"), **out)


if __name__ == "__main__":
    oracle.build(ref=True)
    assert oracle.load_ref() is not None
    make_small()
    make_cfg1()
    make_edges()
    make_powsum()
    make_focal_smoothl1()
    make_conv()
    make_conv_ref()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
