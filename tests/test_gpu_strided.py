"""k x k / strided convolutions at their own size (FPN's P6 / P7: 3x3, stride 2 -- detectron/lib/modeling/FPN.py:193-224):
forward as an implicit GEMM with split-K (gemm_conv.hip), filter and data gradient on flattened-batch GEMMs
(conv_strided.hip) -- against the oracle (restates caffe2/operators/conv_op_impl.h:126-173, 358-577; pinned by the
reference's compiled ConvOp / ConvGradientOp, tests/test_oracle_golden.py), against the stored outputs of the
reference operators, and against the stride-1 + subsample route they replace."""
import os

import numpy as np
import pytest
import torch

import ssad_amd  # noqa: F401
from oracle import oracle
import make_golden as mg
from test_gpu_kernels import CONV_FLOOR, CONV_RTOL, close, dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from ssad_amd import kernels
    kernels.lib()
    return kernels


GEOMS = [
    (2, 64, 32, 8, 12, 3, 2, 1),         # P6-like: even map
    (2, 32, 256, 5, 7, 3, 2, 1),         # P7-like: odd map, OH x OW = 3 x 4, Q = 24 (padded to 32 columns)
    (3, 2048, 256, 4, 6, 3, 2, 1),       # P6's channel counts: K = 18 432, split-K forward
    (1, 16, 24, 9, 11, 3, 2, 1),         # odd map, M tail
    (2, 8, 40, 11, 9, 5, 3, 2),          # 5x5 / 3
    (1, 12, 16, 6, 6, 2, 2, 0),          # 2x2 / 2, no padding
    (2, 16, 16, 7, 10, 3, 1, 1),         # stride 1 (every tap of every input pixel)
]


@pytest.mark.parametrize("geom", GEOMS, ids=lambda g: "N%d_C%d_M%d_%dx%d_k%ds%dp%d" % g)
def test_strided_conv_forward_splitk_vs_oracle(K, geom, monkeypatch):
    N, Cin, M, H, W, k, st, pad = geom
    rng = np.random.default_rng(sum(geom))
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((M, Cin, k, k)) * (1.0 / np.sqrt(Cin * k * k))).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    ref = oracle.conv_forward(X, Wt, b, kernel=k, stride=st, pad=pad)
    got = K.conv_implicit_gemm(dev(X), dev(Wt), dev(b), stride=st, pad=pad, split_k=True)
    assert tuple(got.shape) == ref.shape
    close(got.cpu().numpy(), ref, CONV_RTOL, CONV_FLOOR, "Y (split-K plan)")
    again = K.conv_implicit_gemm(dev(X), dev(Wt), dev(b), stride=st, pad=pad, split_k=True)
    assert torch.equal(got, again), "run-to-run bits (splits are added in index order)"
    close(K.conv_implicit_gemm(dev(X), dev(Wt), None, stride=st, pad=pad, relu=True, split_k=True).cpu().numpy(),
          np.maximum(oracle.conv_forward(X, Wt, None, kernel=k, stride=st, pad=pad), 0), CONV_RTOL, CONV_FLOOR,
          "relu(Y) through the split-K epilogue")


def test_strided_conv_forward_forced_splits(K):
    """Every split count on one geometry (child processes: the plan's override is read once per process),
    including more splits than chunks."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import ssad_amd
from ssad_amd import kernels as K
from oracle import oracle
rng = np.random.default_rng(5)
X = rng.standard_normal((2, 40, 8, 12)).astype(np.float32)
W = (rng.standard_normal((136, 40, 3, 3)) / 19.0).astype(np.float32)
b = rng.standard_normal(136).astype(np.float32)
ref = oracle.conv_forward(X, W, b, kernel=3, stride=2, pad=1)
got = K.conv_implicit_gemm(torch.from_numpy(X).cuda(), torch.from_numpy(W).cuda(), torch.from_numpy(b).cuda(),
                           stride=2, pad=1, split_k=True).cpu().numpy()
err = np.abs(got - ref).max() / np.abs(ref).max()
assert err < 1e-5, err
print("ok", err)
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for splits in ("1", "2", "3", "7", "23", "64"):          # K = 360 = 22.5 chunks
        env = dict(os.environ, SSAD_IMPLICIT_SPLITS=splits)
        r = subprocess.run([sys.executable, "-c", code % (root, os.path.join(root, "tests"))], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (splits, r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.parametrize("geom", GEOMS, ids=lambda g: "N%d_C%d_M%d_%dx%d_k%ds%dp%d" % g)
def test_strided_conv_gradients_vs_oracle(K, geom):
    N, Cin, M, H, W, k, st, pad = geom
    rng = np.random.default_rng(sum(geom) + 1)
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((M, Cin, k, k)) * (1.0 / np.sqrt(Cin * k * k))).astype(np.float32)
    oh, ow = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    dY = rng.standard_normal((N, M, oh, ow)).astype(np.float32)
    rdW, rdb, rdX = oracle.conv_backward(X, Wt, dY, kernel=k, stride=st, pad=pad)
    tx, tw, tdy = dev(X), dev(Wt), dev(dY)
    dW = K.conv_kxk_wgrad(tx, tdy, k, st, pad)
    close(dW.cpu().numpy(), rdW, CONV_RTOL, CONV_FLOOR, "dW")
    assert torch.equal(dW, K.conv_kxk_wgrad(tx, tdy, k, st, pad)), "run-to-run bits"
    base = rng.standard_normal(Wt.shape).astype(np.float32)
    close(K.conv_kxk_wgrad(tx, tdy, k, st, pad, out=dev(base), accumulate=True).cpu().numpy(), base + rdW, CONV_RTOL,
          CONV_FLOOR, "dW accumulated")
    dX = K.conv_kxk_dgrad(tw, tdy, H, W, st, pad)
    close(dX.cpu().numpy(), rdX, CONV_RTOL, CONV_FLOOR, "dX")
    mask = np.maximum(rng.standard_normal(X.shape), 0).astype(np.float32)
    close(K.conv_kxk_dgrad(tw, tdy, H, W, st, pad, mask=dev(mask)).cpu().numpy(), np.where(mask > 0, rdX, 0),
          CONV_RTOL, CONV_FLOOR, "masked dX")
    bx = rng.standard_normal(X.shape).astype(np.float32)
    close(K.conv_kxk_dgrad(tw, tdy, H, W, st, pad, out=dev(bx), accumulate=True).cpu().numpy(), bx + rdX, CONV_RTOL,
          CONV_FLOOR, "dX accumulated")


def test_strided_conv_vs_reference_operator_golden(K, golden_dir):
    """tests/golden/conv_ref.npz: outputs of the reference's compiled ConvOp / ConvGradientOp on its strided and
    k x k geometries (group 1) -- forward through the split-K plan, dW and dX through conv_strided.hip."""
    g = np.load(os.path.join(golden_dir, "conv_ref.npz"))
    done = 0
    for key in g.files:
        if not key.endswith("_dims"):
            continue
        name = key[:-5]
        seed, N, Cin, M, H, W, k, s, p, grp = [int(v) for v in g[key]]
        if grp != 1 or (Cin * k * k) % 4:
            continue
        X, Wt, b, dY = mg.conv_ref_inputs(seed, N, Cin, M, H, W, k, s, p, grp)
        Y = K.conv_implicit_gemm(dev(X), dev(Wt), dev(b), stride=s, pad=p, split_k=True).cpu().numpy()
        close(Y.ravel()[g[name + "_Y_idx"]], g[name + "_Y"], CONV_RTOL, CONV_FLOOR, name + " Y")
        if name + "_dW" in g.files:
            dW = K.conv_kxk_wgrad(dev(X), dev(dY), k, s, p).cpu().numpy()
            close(dW.ravel()[g[name + "_dW_idx"]], g[name + "_dW"], CONV_RTOL, CONV_FLOOR, name + " dW")
            dX = K.conv_kxk_dgrad(dev(Wt), dev(dY), H, W, s, p).cpu().numpy()
            close(dX.ravel()[g[name + "_dX_idx"]], g[name + "_dX"], CONV_RTOL, CONV_FLOOR, name + " dX")
        done += 1
    assert done >= 6


def test_strided_equals_stride1_plus_subsample_at_p6_size(K):
    """P6 at the bench's own size (2048 -> 256 on 20 x 28, bs 16): the layer at its own size against the route it
    replaces -- the stride-1 Winograd layer followed by subsampling -- forward and both gradients (a size the CPU
    oracle does not finish in seconds; the two routes share no kernel)."""
    g = torch.Generator(device="cuda").manual_seed(3)
    N, Cin, M, H, W = 16, 2048, 256, 20, 28
    x = torch.randn((N, Cin, H, W), device="cuda", generator=g)
    w = torch.randn((M, Cin, 3, 3), device="cuda", generator=g) * (1.0 / np.sqrt(Cin * 9))
    b = torch.randn((M,), device="cuda", generator=g)
    y = K.conv_implicit_gemm(x, w, b, stride=2, pad=1, split_k=True)
    wf, wd = K.conv_wino_pack_filter(w, True, True)
    full = K.conv3x3_forward([x], wf, b, M, wino=True)[0]
    ref = K.subsample(full, 2)
    scale = float(ref.abs().max())
    assert float((y - ref).abs().max()) <= 2e-5 * scale
    dy = torch.randn((N, M, H // 2, W // 2), device="cuda", generator=g)
    dyf = K.subsample_grad(dy, H, W, 2)
    dw_ref = K.conv3x3_wgrad([x], [dyf], M)
    dw_ref = dw_ref[0] if isinstance(dw_ref, (tuple, list)) else dw_ref
    dw = K.conv_kxk_wgrad(x, dy, 3, 2, 1)
    assert float((dw - dw_ref.view_as(dw)).abs().max()) <= 2e-5 * float(dw_ref.abs().max())
    dx_ref = K.conv3x3_forward([dyf], wd, None, Cin, wino=True)[0]
    dx = K.conv_kxk_dgrad(w, dy, H, W, 2, 1)
    assert float((dx - dx_ref).abs().max()) <= 2e-5 * float(dx_ref.abs().max())


def test_strided_argument_validation(K):
    import ctypes as C
    L = K.lib()
    x = torch.zeros((1, 8, 6, 6), device="cuda")
    dy = torch.zeros((1, 4, 3, 3), device="cuda")
    dw = torch.zeros((4, 8, 3, 3), device="cuda")
    nb = L.ssad_conv_kxk_wgrad_workspace_bytes(1, 8, 6, 6, 4, 3, 2, 1)
    assert nb > 0
    ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    # workspace too small / missing
    assert L.ssad_conv_kxk_wgrad(p(x), p(dy), 1, 8, 6, 6, 4, 3, 2, 1, p(dw), 0, p(ws), nb - 1, None) == -2
    assert L.ssad_conv_kxk_wgrad(p(x), p(dy), 1, 8, 6, 6, 4, 3, 2, 1, p(dw), 0, None, nb, None) == -2
    # kernel larger than the padded map, stride 0
    assert L.ssad_conv_kxk_wgrad(p(x), p(dy), 1, 8, 1, 1, 4, 5, 2, 1, p(dw), 0, p(ws), nb, None) == -1
    assert L.ssad_conv_kxk_wgrad(p(x), p(dy), 1, 8, 6, 6, 4, 3, 0, 1, p(dw), 0, p(ws), nb, None) == -1
    assert L.ssad_conv_kxk_dgrad_workspace_bytes(1, 8, 6, 6, 4, 3, 0, 1) == 0
    nd = L.ssad_conv_kxk_dgrad_workspace_bytes(1, 8, 6, 6, 4, 3, 2, 1)
    wsd = torch.zeros(nd, dtype=torch.uint8, device="cuda")
    assert L.ssad_conv_kxk_dgrad(p(dw), p(dy), 1, 8, 6, 6, 4, 3, 2, 1, p(x), None, 0, p(wsd), nd - 1, None) == -2
    # Cin * k * k must be a multiple of 4 (the filter is a GEMM operand with 16-byte rows)
    assert L.ssad_conv_kxk_dgrad(p(dw), p(dy), 1, 3, 6, 6, 4, 3, 2, 1, p(x), None, 0, p(wsd), nd, None) == -1


@pytest.mark.parametrize("shape", [(2, 5, 8, 12), (1, 3, 7, 16), (3, 4, 9, 10), (2, 2, 6, 4), (1, 1, 1, 4)],
                         ids=lambda s: "N%d_C%d_%dx%d" % s)
def test_subsample_stride2_fast_path_and_general_path(K, shape):
    """ssad_subsample / ssad_subsample_grad: widths that are multiples of 4 take the 16-byte kernels, the others the
    general one; both are exact copies (bit equality with the slice / the zero-stuffed scatter)."""
    N, Cc, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = torch.randn((N, Cc, H, W), device="cuda", generator=g)
    y = K.subsample(x, 2)
    assert torch.equal(y, x[:, :, ::2, ::2])
    dy = torch.randn_like(y)
    ref = torch.zeros_like(x)
    ref[:, :, ::2, ::2] = dy
    assert torch.equal(K.subsample_grad(dy, H, W, 2), ref)
    base = torch.randn_like(x)
    want = base + ref
    assert torch.equal(K.subsample_grad(dy, H, W, 2, accumulate_into=base.clone()), want)


@pytest.mark.parametrize("shape", [(2, 5, 8, 12), (1, 3, 7, 6), (3, 4, 5, 7), (1, 2, 1, 2)],
                         ids=lambda s: "N%d_C%d_%dx%d" % s)
def test_upsample_nearest_fast_path_and_general_path(K, shape):
    """ssad_upsample_nearest (+ lateral Sum, also in place) / its gradient (upsample_nearest_op.cu:62-151): even
    widths take the 16-byte kernels, odd ones the general kernel; same values, same summation order."""
    N, Cc, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    x = torch.randn((N, Cc, H, W), device="cuda", generator=g)
    up = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    assert torch.equal(K.upsample_nearest(x, 2), up)
    add = torch.randn_like(up)
    assert torch.equal(K.upsample_nearest(x, 2, addend=add), up + add)
    inplace = add.clone()
    K.upsample_nearest(x, 2, addend=inplace, out=inplace)
    assert torch.equal(inplace, up + add)
    dy = torch.randn_like(up)
    d = dy.view(N, Cc, H, 2, W, 2)
    want = ((d[:, :, :, 0, :, 0] + d[:, :, :, 0, :, 1]) + d[:, :, :, 1, :, 0]) + d[:, :, :, 1, :, 1]
    assert torch.equal(K.upsample_nearest_grad(dy, 2), want)
