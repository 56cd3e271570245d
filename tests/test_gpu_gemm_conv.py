"""Pointwise convolution as an fp32-MFMA GEMM (csrc/kernels/gemm_conv.hip; row f1): forward with
the bottleneck tail in the epilogue, data gradient (mask / accumulate), filter gradient, the
strided variant through ssad_subsample -- against the oracle (which restates
caffe2/operators/conv_op_impl.h and is pinned by the reference's compiled operators,
tests/test_oracle_golden.py) and against the stored outputs of the reference operators."""
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


@pytest.mark.parametrize("shape", [
    (2, 64, 64, 12, 16),       # 64-wide output: the 64 x 128 tile
    (2, 24, 40, 9, 12),        # K tail (24), M tail (40), P = 108: columns flattened across images
    (3, 256, 128, 10, 14),     # P = 140: a tile spans two images
    (1, 128, 512, 20, 28),     # res3 c3 at small size, 4 M tiles
    (2, 2048, 512, 5, 8),      # res5 c1: 128 K chunks
    (1, 16, 8, 2, 2),          # tiny: one partial tile, P = 4
], ids=lambda s: "N%d_C%d_M%d_%dx%d" % s)
def test_gemm_conv_forward_dgrad_wgrad_vs_oracle(K, shape):
    N, Cin, M, H, W = shape
    rng = np.random.default_rng(sum(shape))
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((M, Cin, 1, 1)) * (1.0 / np.sqrt(Cin))).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    R = rng.standard_normal((N, M, H, W)).astype(np.float32)
    dY = rng.standard_normal((N, M, H, W)).astype(np.float32)
    tw, tx = dev(Wt), dev(X)
    wt = K.transpose_filter(tw)
    ref = oracle.conv_forward(X, Wt, b, kernel=1, stride=1, pad=0)
    close(K.conv1x1_forward(tx, wt, M, dev(b)).cpu().numpy(), ref, CONV_RTOL, CONV_FLOOR, "Y")
    close(K.conv1x1_forward(tx, wt, M, dev(b), dev(R), relu=True).cpu().numpy(), np.maximum(ref + R, 0), CONV_RTOL,
          CONV_FLOOR, "relu(Y + R)")
    close(K.conv1x1_forward(tx, wt, M).cpu().numpy(), oracle.conv_forward(X, Wt, None, kernel=1, stride=1, pad=0),
          CONV_RTOL, CONV_FLOOR, "Y no bias")
    if (H * W) % 16 == 0 or True:
        rdW, rdb, rdX = oracle.conv_backward(X, Wt, dY, kernel=1, stride=1, pad=0)
        tdy = dev(dY)
        close(K.conv1x1_dgrad(tdy, tw).cpu().numpy(), rdX, CONV_RTOL, CONV_FLOOR, "dX")
        # fused ReluGradient mask and accumulation onto an existing gradient
        mask = np.maximum(rng.standard_normal(X.shape), 0).astype(np.float32)
        close(K.conv1x1_dgrad(tdy, tw, mask=dev(mask)).cpu().numpy(), np.where(mask > 0, rdX, 0), CONV_RTOL,
              CONV_FLOOR, "masked dX")
        base = rng.standard_normal(X.shape).astype(np.float32)
        got = K.conv1x1_dgrad(tdy, tw, accumulate_into=dev(base)).cpu().numpy()
        close(got, base + rdX, CONV_RTOL, CONV_FLOOR, "dX accumulated")
        if (H * W) % 16 == 0:
            dW = K.conv1x1_wgrad(tx, tdy)
            close(dW.cpu().numpy(), rdW.reshape(M, Cin), CONV_RTOL, CONV_FLOOR, "dW")
            dW2 = K.conv1x1_wgrad(tx, tdy, out=dW.clone(), accumulate=True)
            close(dW2.cpu().numpy(), 2 * rdW.reshape(M, Cin), CONV_RTOL, CONV_FLOOR, "dW accumulate")
            assert torch.equal(K.conv1x1_wgrad(tx, tdy), dW)            # deterministic


def test_gemm_conv_vs_reference_operator_golden(K, golden_dir):
    """k1s1 and k1s2 of tests/golden/conv_ref.npz (outputs of the reference's compiled
    ConvOp / ConvGradientOp): the strided layer = the pointwise layer on the subsampled map."""
    g = np.load(os.path.join(golden_dir, "conv_ref.npz"))
    for name in ("k1s1p96", "k1s2p96", "k1s1c160"):
        seed, N, Cin, M, H, W, k, s, p, grp = [int(v) for v in g[name + "_dims"]]
        X, Wt, b, dY = mg.conv_ref_inputs(seed, N, Cin, M, H, W, k, s, p, grp)
        tx, tw = dev(X), dev(Wt)
        xs = K.subsample(tx, s) if s > 1 else tx
        if s > 1:
            assert np.array_equal(xs.cpu().numpy(), X[:, :, ::s, ::s])
        wt = K.transpose_filter(tw)
        Y = K.conv1x1_forward(xs, wt, M, dev(b)).cpu().numpy()
        close(Y.ravel()[g[name + "_Y_idx"]], g[name + "_Y"], CONV_RTOL, CONV_FLOOR, name + " Y")
        dxs = K.conv1x1_dgrad(dev(dY), tw)
        dX = K.subsample_grad(dxs, H, W, s) if s > 1 else dxs
        close(dX.cpu().numpy().ravel()[g[name + "_dX_idx"]], g[name + "_dX"], CONV_RTOL, CONV_FLOOR, name + " dX")
        dW = K.conv1x1_wgrad(xs, dev(dY))
        close(dW.cpu().numpy().ravel()[g[name + "_dW_idx"]], g[name + "_dW"], CONV_RTOL, CONV_FLOOR, name + " dW")
    # maps whose pixel count is not a multiple of 4 are refused (no backbone layer has one)
    with pytest.raises(K.KernelError):
        K.conv1x1_forward(torch.zeros(1, 8, 3, 3, device="cuda"), torch.zeros(8, 8, device="cuda"), 8)


def test_gemm_conv_full_size_adjoints(K):
    """res4's 1024 -> 256 layer at bs 16 (P = 40 x 56): <conv(X), dY> = <X, dgrad(dY)> = <W, wgrad>,
    plus one image against the oracle."""
    gen = torch.Generator(device="cuda").manual_seed(3)
    N, Cin, M, H, W = 16, 1024, 256, 40, 56
    X = torch.randn((N, Cin, H, W), device="cuda", generator=gen)
    dY = torch.randn((N, M, H, W), device="cuda", generator=gen)
    Wt = torch.randn((M, Cin, 1, 1), device="cuda", generator=gen) * 0.03
    wt = K.transpose_filter(Wt)
    Y = K.conv1x1_forward(X, wt, M)
    dX = K.conv1x1_dgrad(dY, Wt)
    dW = K.conv1x1_wgrad(X, dY)
    a = float((Y.double() * dY.double()).sum())
    b = float((X.double() * dX.double()).sum())
    c = float((Wt.view(M, Cin).double() * dW.double()).sum())
    scale = float((Y.double().abs() * dY.double().abs()).sum())
    assert abs(a - b) <= 1e-5 * scale and abs(a - c) <= 1e-5 * scale, (a, b, c, scale)
    n0 = 13
    ref = oracle.conv_forward(X[n0:n0 + 1].cpu().numpy(), Wt.cpu().numpy(), None, kernel=1, stride=1, pad=0)
    close(Y[n0:n0 + 1].cpu().numpy(), ref, CONV_RTOL, CONV_FLOOR, "full-size slice")


def _grouped_oracle(X, Wt, b, group, stride):
    """The groups are independent convolutions over contiguous channel blocks
    (conv_op_impl.h:93-98,126-173): the oracle's dense conv per group."""
    cg = X.shape[1] // group
    outs = [oracle.conv_forward(np.ascontiguousarray(X[:, g * cg:(g + 1) * cg]),
                                np.ascontiguousarray(Wt[g * cg:(g + 1) * cg]),
                                None if b is None else b[g * cg:(g + 1) * cg], kernel=3, stride=stride, pad=1)
            for g in range(group)]
    return np.concatenate(outs, axis=1)


@pytest.mark.parametrize("geom", [
    (2, 4, 16, 19, 23, 1), (2, 4, 16, 19, 23, 2),       # cg = 4 (res2), ragged tiles
    (1, 8, 8, 16, 32, 1), (2, 8, 4, 9, 17, 2),          # cg = 8 (res3)
    (2, 16, 4, 12, 24, 1), (1, 16, 64, 8, 16, 2),       # cg = 16 (res4), 64 groups
    (1, 32, 2, 10, 21, 1), (2, 32, 3, 16, 24, 2),       # cg = 32 (res5)
], ids=lambda g: "N%d_cg%d_G%d_%dx%d_s%d" % g)
def test_grouped_conv3x3_vs_oracle(K, geom):
    N, cg, G, H, W, s = geom
    C = cg * G
    rng = np.random.default_rng(sum(geom))
    X = rng.standard_normal((N, C, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((C, cg, 3, 3)) * (1.0 / np.sqrt(9 * cg))).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    ref = _grouped_oracle(X, Wt, b, G, s)
    got = K.grouped_conv3x3_forward(dev(X), dev(Wt), dev(b), group=G, stride=s)
    assert tuple(got.shape) == ref.shape
    close(got.cpu().numpy(), ref, CONV_RTOL, CONV_FLOOR, "Y")
    got = K.grouped_conv3x3_forward(dev(X), dev(Wt), None, group=G, stride=s, relu=True)
    close(got.cpu().numpy(), np.maximum(_grouped_oracle(X, Wt, None, G, s), 0), CONV_RTOL, CONV_FLOOR, "relu(Y)")


def test_grouped_conv3x3_vs_reference_operator_golden(K, golden_dir):
    """Every grouped case of tests/golden/conv_ref.npz (outputs of the reference's compiled ConvOp
    with group > 1)."""
    g = np.load(os.path.join(golden_dir, "conv_ref.npz"))
    names = [k[:-5] for k in g.files if k.endswith("_dims") and int(g[k][9]) > 1]
    assert len(names) >= 6
    for name in names:
        seed, N, Cin, M, H, W, k, s, p, grp = [int(v) for v in g[name + "_dims"]]
        X, Wt, b, _ = mg.conv_ref_inputs(seed, N, Cin, M, H, W, k, s, p, grp)
        Y = K.grouped_conv3x3_forward(dev(X), dev(Wt), dev(b), group=grp, stride=s).cpu().numpy()
        close(Y.ravel()[g[name + "_Y_idx"]], g[name + "_Y"], CONV_RTOL, CONV_FLOOR, name + " Y")
    with pytest.raises(K.KernelError):          # 12 channels per group: not a ResNeXt width
        K.grouped_conv3x3_forward(torch.zeros(1, 24, 4, 4, device="cuda"), torch.zeros(24, 12, 3, 3, device="cuda"),
                                  group=2)
    assert K.lib().ssad_grouped_conv3x3_filter_floats(256, 64) == 64 * 9 * 64


@pytest.mark.parametrize("geom", [
    (2, 3, 64, 32, 40, 7, 2, 3),        # the stem: 7x7 / 2, pad 3, three input channels
    (2, 3, 64, 128, 192, 7, 2, 3),      # the stem at the body-graph test's image size
    (1, 3, 16, 22, 32, 7, 2, 3),        # ragged borders on every side (11 x 16 outputs)
    (1, 3, 64, 22, 32, 7, 2, 3),        # the same map through the stem kernel (one partial 8 x 32 tile)
    (3, 3, 64, 75, 141, 7, 2, 3),       # odd image, 38 x 71 outputs: partial tiles right and below, 3 images
    (2, 16, 24, 12, 16, 3, 2, 1),       # 3x3 / 2 (P6 / P7)
    (1, 8, 136, 8, 12, 3, 1, 1),        # 3x3 / 1, two 128-row tiles
    (2, 24, 40, 8, 12, 1, 1, 0),        # degenerates to the pointwise GEMM
], ids=lambda g: "N%d_C%d_M%d_%dx%d_k%ds%dp%d" % g)
def test_conv_implicit_gemm_vs_oracle(K, geom):
    N, Cin, M, H, W, k, st, pad = geom
    rng = np.random.default_rng(sum(geom))
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((M, Cin, k, k)) * (1.0 / np.sqrt(Cin * k * k))).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    ref = oracle.conv_forward(X, Wt, b, kernel=k, stride=st, pad=pad)
    got = K.conv_implicit_gemm(dev(X), dev(Wt), dev(b), stride=st, pad=pad)
    assert tuple(got.shape) == ref.shape
    close(got.cpu().numpy(), ref, CONV_RTOL, CONV_FLOOR, "Y")
    close(K.conv_implicit_gemm(dev(X), dev(Wt), None, stride=st, pad=pad, relu=True).cpu().numpy(),
          np.maximum(oracle.conv_forward(X, Wt, None, kernel=k, stride=st, pad=pad), 0), CONV_RTOL, CONV_FLOOR,
          "relu(Y)")
    # no epilogue term at all: for the stem geometry (3 -> 64 channels, 7x7 / 2, pad 3) this is the dedicated kernel
    # of stem.hip (raw patch staged in LDS), which the backbones call; any other geometry stays on the general path
    plain = K.conv_implicit_gemm(dev(X), dev(Wt), None, stride=st, pad=pad)
    close(plain.cpu().numpy(), oracle.conv_forward(X, Wt, None, kernel=k, stride=st, pad=pad), CONV_RTOL, CONV_FLOOR,
          "Y without bias")
    assert torch.equal(plain, K.conv_implicit_gemm(dev(X), dev(Wt), None, stride=st, pad=pad)), "run-to-run bits"


def test_conv_implicit_gemm_vs_reference_operator_golden(K, golden_dir):
    """The strided / 7x7 geometries of tests/golden/conv_ref.npz (outputs of the reference's compiled
    ConvOp) whose output map is a multiple of 4 pixels."""
    g = np.load(os.path.join(golden_dir, "conv_ref.npz"))
    done = 0
    stem_seen = False
    for key in g.files:
        if not key.endswith("_dims"):
            continue
        name = key[:-5]
        seed, N, Cin, M, H, W, k, s, p, grp = [int(v) for v in g[key]]
        oh, ow = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        if grp != 1 or (oh * ow) % 4:
            continue
        X, Wt, b, _ = mg.conv_ref_inputs(seed, N, Cin, M, H, W, k, s, p, grp)
        Y = K.conv_implicit_gemm(dev(X), dev(Wt), dev(b), stride=s, pad=p).cpu().numpy()
        close(Y.ravel()[g[name + "_Y_idx"]], g[name + "_Y"], CONV_RTOL, CONV_FLOOR, name + " Y")
        if (Cin, M, k, s, p) == (3, 64, 7, 2, 3):
            # the stem geometry without epilogue terms is stem.hip's kernel: the reference operator's output
            # minus its bias
            Y0 = K.conv_implicit_gemm(dev(X), dev(Wt), None, stride=s, pad=p).cpu().numpy()
            want = g[name + "_Y"] - np.broadcast_to(b.reshape(1, M, 1, 1), Y0.shape).ravel()[g[name + "_Y_idx"]]
            close(Y0.ravel()[g[name + "_Y_idx"]], want, CONV_RTOL, CONV_FLOOR, name + " Y (stem kernel)")
            stem_seen = True
        done += 1
    assert done >= 8 and stem_seen


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_general_gemm_matches_float64(K, ta, tb):
    """ssad_gemm_f32 (kernels/gemm_general.hip): math::Gemm / GemmStridedBatched of the default
    convolution engine's im2col route -- every transposition, sizes that are multiples of nothing,
    padded leading dimensions, alpha / beta, a batch; beta = 0 must ignore what C held (NaN)."""
    import ctypes as C
    L = K.lib()
    L.ssad_gemm_f32.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int,
                                C.c_longlong, C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_void_p, C.c_int,
                                C.c_longlong, C.c_int, C.c_void_p]
    g = torch.Generator(device="cuda").manual_seed(17)
    M, N, Kk, batch = 75, 131, 147, 3
    lda = (M if ta else Kk) + 5
    ldb = (Kk if tb else N) + 3
    ldc = N + 2
    A = torch.randn((batch, Kk if ta else M, lda), device="cuda", generator=g)
    B = torch.randn((batch, N if tb else Kk, ldb), device="cuda", generator=g)
    Cm = torch.full((batch, M, ldc), float("nan"), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rc = L.ssad_gemm_f32(ta, tb, M, N, Kk, 1.5, A.data_ptr(), lda, A.stride(0), B.data_ptr(), ldb, B.stride(0), 0.0,
                         Cm.data_ptr(), ldc, Cm.stride(0), batch, st)
    assert rc == 0
    opA = (A[:, :, :M].transpose(1, 2) if ta else A[:, :, :Kk]).double()
    opB = (B[:, :, :Kk].transpose(1, 2) if tb else B[:, :, :N]).double()
    want = 1.5 * torch.bmm(opA, opB)
    got = Cm[:, :, :N].double()
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    assert torch.isnan(Cm[:, :, N:]).all()                      # the padding of C was not touched
    # beta = 1 accumulates
    rc = L.ssad_gemm_f32(ta, tb, M, N, Kk, -0.5, A.data_ptr(), lda, A.stride(0), B.data_ptr(), ldb, B.stride(0), 1.0,
                         Cm.data_ptr(), ldc, Cm.stride(0), batch, st)
    assert rc == 0
    got = Cm[:, :, :N].double()
    assert float((got - want * (1.0 / 1.5)).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.parametrize("shape", [
    (16, 256, 64, 160, 224),     # res2 c1 at config 3's real size: the HBM-bound 64-wide-output tile
    (16, 1024, 256, 40, 56),     # res4 c1: 560 tiles of 128 x 128 (the workgroup-quantisation case of DESIGN 3.5)
    (16, 512, 2048, 20, 28),     # res5 c3: P = 560 does not fill the last 128-column tile of an image
], ids=lambda s: "N%d_C%d_M%d_%dx%d" % s)
def test_gemm_conv_full_size_vs_oracle_and_adjoints(K, shape):
    """the pointwise GEMM kernels at the backbone's REAL bs-16 shapes (640 x 896 input)
    against the oracle, as test_default_engine_headline_size_vs_oracle_and_adjoints does for the 3x3 engines:
    forward (bias + shortcut + ReLU) and data gradient (mask) of one image of the bs-16 launch against
    oracle.conv_forward / conv_backward; the filter gradient by linearity (only image n0's dY non-zero in the
    bs-16 launch); and the three adjoint identities over the whole batch."""
    N, Cin, M, H, W = shape
    n0 = 9
    gen = torch.Generator(device="cuda").manual_seed(sum(shape))
    X = torch.randn((N, Cin, H, W), device="cuda", generator=gen)
    dY = torch.randn((N, M, H, W), device="cuda", generator=gen)
    Wt = torch.randn((M, Cin, 1, 1), device="cuda", generator=gen) * float(1.0 / np.sqrt(Cin))
    b = torch.randn(M, device="cuda", generator=gen)
    wt = K.transpose_filter(Wt)
    Y = K.conv1x1_forward(X, wt, M, b)
    dX = K.conv1x1_dgrad(dY, Wt)
    dW = K.conv1x1_wgrad(X, dY)
    Yl = Y.double() - b.double().view(1, M, 1, 1)
    a = float((Yl * dY.double()).sum())
    bb = float((X.double() * dX.double()).sum())
    c = float((Wt.double().view(M, Cin) * dW.double()).sum())
    scale = float((Yl.abs() * dY.double().abs()).sum())
    assert abs(a - bb) <= 1e-5 * scale and abs(a - c) <= 1e-5 * scale, (a, bb, c, scale)
    # one image of the launch against the oracle
    x1, dy1, w_np, b_np = X[n0:n0 + 1].cpu().numpy(), dY[n0:n0 + 1].cpu().numpy(), Wt.cpu().numpy(), b.cpu().numpy()
    ref = oracle.conv_forward(x1, w_np, b_np, kernel=1, stride=1, pad=0)
    close(Y[n0:n0 + 1].cpu().numpy(), ref, CONV_RTOL, CONV_FLOOR, "full-size fwd slice")
    R = torch.randn((N, M, H, W), device="cuda", generator=gen)
    Yr = K.conv1x1_forward(X, wt, M, b, R, relu=True)
    close(Yr[n0:n0 + 1].cpu().numpy(), np.maximum(ref + R[n0:n0 + 1].cpu().numpy(), 0), CONV_RTOL, CONV_FLOOR,
          "full-size relu(Y + R) slice")
    del R, Yr
    ref_dW, _, ref_dX = oracle.conv_backward(x1, w_np, dy1, kernel=1, stride=1, pad=0)
    close(dX[n0:n0 + 1].cpu().numpy(), ref_dX, CONV_RTOL, CONV_FLOOR, "full-size dgrad slice")
    mask = torch.randn((N, Cin, H, W), device="cuda", generator=gen).clamp_(min=0)
    dXm = K.conv1x1_dgrad(dY, Wt, mask=mask)
    close(dXm[n0:n0 + 1].cpu().numpy(), np.where(mask[n0:n0 + 1].cpu().numpy() > 0, ref_dX, 0), CONV_RTOL, CONV_FLOOR,
          "full-size masked dgrad slice")
    del mask, dXm
    dY0 = torch.zeros_like(dY)
    dY0[n0].copy_(dY[n0])
    dW0 = K.conv1x1_wgrad(X, dY0)
    close(dW0.cpu().numpy(), ref_dW.reshape(M, Cin), CONV_RTOL, CONV_FLOOR, "full-size wgrad (one contributing image)")
    assert torch.equal(K.conv1x1_wgrad(X, dY), dW)            # deterministic at this size too


# ---------------------------------------------------------------------------
# Round 6: the same descriptor on the split-operand engine (gemm_split.hip), at the exact-fp32 GEMM's bar
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("shape", [
    (2, 64, 64, 12, 16), (2, 24, 40, 9, 12), (3, 256, 128, 10, 14), (1, 128, 512, 20, 28), (2, 2048, 512, 5, 8),
    (1, 16, 8, 2, 2), (2, 256, 1024, 6, 7), (1, 1024, 256, 11, 13), (2, 72, 300, 5, 5),
], ids=lambda s: "N%d_C%d_M%d_%dx%d" % s)
def test_gemm_split_forward_and_dgrad_vs_oracle(K, shape):
    """ssad_conv1x1_gemm_split against the oracle: bias, shortcut + ReLU, data gradient with the fused ReluGradient
    mask and with accumulation -- K / M tails (not multiples of 8 / 64 / 256), pixel tiles that end inside an image."""
    N, Cin, M, H, W = shape
    rng = np.random.default_rng(600 + sum(shape))
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    Wt = (rng.standard_normal((M, Cin, 1, 1)) * (1.0 / np.sqrt(Cin))).astype(np.float32)
    b = rng.standard_normal(M).astype(np.float32)
    R = rng.standard_normal((N, M, H, W)).astype(np.float32)
    dY = rng.standard_normal((N, M, H, W)).astype(np.float32)
    tw, tx = dev(Wt), dev(X)
    wt = K.transpose_filter(tw)
    ref = oracle.conv_forward(X, Wt, b, kernel=1, stride=1, pad=0)
    close(K.conv1x1_forward(tx, wt, M, dev(b), split=True).cpu().numpy(), ref, CONV_RTOL, CONV_FLOOR, "Y")
    close(K.conv1x1_forward(tx, wt, M, dev(b), dev(R), relu=True, split=True).cpu().numpy(), np.maximum(ref + R, 0),
          CONV_RTOL, CONV_FLOOR, "relu(Y + R)")
    close(K.conv1x1_forward(tx, wt, M, split=True).cpu().numpy(), oracle.conv_forward(X, Wt, None, kernel=1, stride=1, pad=0),
          CONV_RTOL, CONV_FLOOR, "Y no bias")
    rdW, rdb, rdX = oracle.conv_backward(X, Wt, dY, kernel=1, stride=1, pad=0)
    tdy = dev(dY)
    close(K.conv1x1_dgrad(tdy, tw, split=True).cpu().numpy(), rdX, CONV_RTOL, CONV_FLOOR, "dX")
    mask = np.maximum(rng.standard_normal(X.shape), 0).astype(np.float32)
    close(K.conv1x1_dgrad(tdy, tw, mask=dev(mask), split=True).cpu().numpy(), np.where(mask > 0, rdX, 0), CONV_RTOL,
          CONV_FLOOR, "masked dX")
    base = rng.standard_normal(X.shape).astype(np.float32)
    got = K.conv1x1_dgrad(tdy, tw, accumulate_into=dev(base), split=True).cpu().numpy()
    close(got, base + rdX, CONV_RTOL, CONV_FLOOR, "dX accumulated")


def test_gemm_split_full_size_vs_exact_fp32_gemm_and_float64(K):
    """res4's two pointwise shapes at config 3's size (bs 16, 40 x 56): every element against the exact-fp32 MFMA GEMM
    (1e-5 of the scale), one image against float64, bit-reproducible, many work items per workgroup."""
    gen = torch.Generator(device="cuda").manual_seed(61)
    for Cin, M in ((1024, 256), (256, 1024)):
        N, H, W = 16, 40, 56
        X = torch.randn((N, Cin, H, W), device="cuda", generator=gen).clamp_(min=0)
        Wt = torch.randn((M, Cin, 1, 1), device="cuda", generator=gen) * (1.0 / np.sqrt(Cin))
        b = torch.randn(M, device="cuda", generator=gen)
        R = torch.randn((N, M, H, W), device="cuda", generator=gen)
        wt = K.transpose_filter(Wt)
        want = K.conv1x1_forward(X, wt, M, b, R, relu=True)
        got = K.conv1x1_forward(X, wt, M, b, R, relu=True, split=True)
        assert (got - want).abs().max().item() <= 1e-5 * want.abs().max().item()
        assert torch.equal(got, K.conv1x1_forward(X, wt, M, b, R, relu=True, split=True))
        n0 = 9
        ref = torch.einsum("mc,chw->mhw", Wt[:, :, 0, 0].double(), X[n0].double()) + b.double()[:, None, None] + R[n0].double()
        ref = ref.clamp_(min=0).cpu().numpy()
        e_s = np.abs(got[n0].double().cpu().numpy() - ref).max() / np.abs(ref).max()
        e_f = np.abs(want[n0].double().cpu().numpy() - ref).max() / np.abs(ref).max()
        assert e_s <= 2e-6 and e_s <= 4 * e_f + 1e-7, (e_s, e_f)


# ---------------------------------------------------------------------------
# Split-operand pointwise filter gradient (gemm_split.hip, wpoint_split_kernel): conv1x1_wgrad's contract, same tolerance
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("shape", [(2, 256, 256, 8, 16), (3, 264, 520, 5, 8), (1, 64, 72, 4, 6), (2, 512, 256, 20, 28),
                                   (1, 300, 40, 2, 4)], ids=lambda s: "N%d_C%d_M%d_%dx%d" % s)
def test_split_pointwise_wgrad_vs_oracle(K, shape):
    N, Cin, M, H, W = shape
    rng = np.random.default_rng(3100 + sum(shape))
    X = rng.standard_normal((N, Cin, H, W)).astype(np.float32)
    dY = (rng.standard_normal((N, M, H, W)) * 1e-3).astype(np.float32)
    Wt = np.zeros((M, Cin, 1, 1), np.float32)
    ref_dW, _, _ = oracle.conv_backward(X, Wt, dY, kernel=1, stride=1, pad=0)
    ref_dW = ref_dW.reshape(M, Cin)
    dW = K.conv1x1_wgrad(dev(X), dev(dY), split=True)
    close(dW.cpu().numpy(), ref_dW, CONV_RTOL, CONV_FLOOR, "split pointwise dW")
    base = dev(np.full_like(ref_dW, 0.25))
    K.conv1x1_wgrad(dev(X), dev(dY), out=base, accumulate=True, split=True)
    close(base.cpu().numpy(), ref_dW + 0.25, CONV_RTOL, CONV_FLOOR, "split pointwise dW accumulate")
    assert torch.equal(K.conv1x1_wgrad(dev(X), dev(dY), split=True), dW)        # deterministic


def test_split_pointwise_wgrad_full_size_vs_exact_engine(K):
    """res4's 1024 <- 256 layer at the bench's size (bs 16, 40 x 56) against the exact-fp32 MFMA engine."""
    gen = torch.Generator(device="cuda").manual_seed(17)
    N, Cin, M, H, W = 16, 256, 1024, 40, 56
    X = torch.randn((N, Cin, H, W), device="cuda", generator=gen).clamp_(min=0)
    dY = torch.randn((N, M, H, W), device="cuda", generator=gen) * 1e-4
    want = K.conv1x1_wgrad(X, dY).clone()
    got = K.conv1x1_wgrad(X, dY, split=True)
    close(got.cpu().numpy(), want.cpu().numpy(), CONV_RTOL, CONV_FLOOR, "split pointwise dW, full size")


def test_split_engines_with_handed_over_max_words_match_the_self_measuring_calls(K):
    """The |max| words measured once (ssad_split_absmax / _levels, or folded in by a producer's epilogue) and the filter
    split by the table op give bit-identical results to the calls that measure and split for themselves."""
    gen = torch.Generator(device="cuda").manual_seed(23)
    N, Cin, M, H, W = 2, 264, 300, 12, 20
    X = torch.randn((N, Cin, H, W), device="cuda", generator=gen).clamp_(min=0) * 3.0
    Wt = torch.randn((M, Cin, 1, 1), device="cuda", generator=gen) * 0.05
    b = torch.randn(M, device="cuda", generator=gen)
    wt = K.transpose_filter(Wt)
    want = K.conv1x1_forward(X, wt, M, b, relu=True, split=True)
    xw = K.split_absmax(X)
    assert int(xw.item()) == int(X.abs().max().view(torch.int32).item())
    pa = K.gemm_split_pack_filter(wt, wt.shape[1], Cin, M)
    yw = torch.zeros(1, dtype=torch.int32, device="cuda")
    got = K.conv1x1_forward_split_amax(X, wt, M, b, relu=True, packed_a=pa, x_amax=xw, y_amax=yw)
    assert torch.equal(got, want)
    assert int(yw.item()) == int(want.abs().max().view(torch.int32).item())
    # only one of the two handed over
    assert torch.equal(K.conv1x1_forward_split_amax(X, wt, M, b, relu=True, packed_a=pa), want)
    assert torch.equal(K.conv1x1_forward_split_amax(X, wt, M, b, relu=True, x_amax=xw), want)
    # pointwise filter gradient
    dY = torch.randn((N, M, H, W), device="cuda", generator=gen) * 1e-3
    dw = K.conv1x1_wgrad(X, dY, split=True).clone()
    assert torch.equal(K.conv1x1_wgrad(X, dY, split=True, x_amax=xw, dy_amax=K.split_absmax(dY)), dw)
    # 3x3: the producer's epilogue folds its output's |max| in (amax_out); the consumer takes the words (amax_in); the
    # filter gradient takes one word per level
    Cc = 256
    shapes = [(10, 14), (5, 7)]
    Xs = [torch.randn((N, Cc, h, w), device="cuda", generator=gen) for h, w in shapes]
    W3 = torch.randn((Cc, Cc, 3, 3), device="cuda", generator=gen) * 0.02
    pf = K.conv_split_pack_filter(W3, want_dgrad=False)
    yw3 = torch.zeros(len(Xs), dtype=torch.int32, device="cuda")
    Y1 = K.conv3x3_forward_split(Xs, pf, None, Cc, relu=True, amax_out=yw3)
    assert [int(v) for v in yw3.tolist()] == [int(y.abs().max().view(torch.int32).item()) for y in Y1]
    lw = K.split_absmax_levels(Y1)
    assert torch.equal(lw, yw3)
    Y2 = K.conv3x3_forward_split(Y1, pf, None, Cc)
    Y2k = K.conv3x3_forward_split(Y1, pf, None, Cc, amax_in=yw3)
    assert all(torch.equal(a, c) for a, c in zip(Y2, Y2k))
    dYs = [torch.randn(y.shape, device="cuda", generator=gen) * 1e-2 for y in Y1]
    dW, db = K.conv3x3_wgrad(Y1, dYs, Cc, split=True)
    dW, db = dW.clone(), db.clone()
    dWk, dbk = K.conv3x3_wgrad(Y1, dYs, Cc, split=True, x_amax=yw3, dy_amax=K.split_absmax_levels(dYs))
    assert torch.equal(dWk, dW) and torch.equal(dbk, db)
