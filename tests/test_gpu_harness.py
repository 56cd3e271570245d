"""The harness' HIP 3x3 convolution (autograd wrapper over the Winograd forward /
data-gradient and the direct weight-gradient kernels) against torch's conv2d."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fm():
    import ssad_amd  # noqa: F401
    from ssad_amd.harness import full_model
    return full_model


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("channels_last", [False, True])
def test_hip_conv3x3_autograd_matches_torch(fm, relu, channels_last):
    torch.manual_seed(5)
    N, C, M, H, W = 2, 128, 256, 13, 22
    x = torch.randn(N, C, H, W, device="cuda")
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    w = torch.randn(M, C, 3, 3, device="cuda") * 0.03
    b = torch.randn(M, device="cuda")
    dy = torch.randn(N, M, H, W, device="cuda")

    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = torch.nn.functional.conv2d(xr, wr, br, padding=1)
    if relu:
        yr = torch.relu(yr)
    yr.backward(dy)

    m = fm.HipConv3x3(C, M, relu=relu).cuda()
    with torch.no_grad():
        m.weight.copy_(w)
        m.bias.copy_(b)
    xh = x.clone().requires_grad_(True)
    yh = m(xh)
    assert yh.is_contiguous(memory_format=torch.channels_last) == channels_last or not channels_last
    yh.backward(dy)
    assert _rel(yh, yr) < 1e-4
    assert _rel(xh.grad, xr.grad) < 1e-4
    assert _rel(m.weight.grad, wr.grad) < 1e-4
    assert _rel(m.bias.grad, br.grad) < 1e-4
    # frozen path (teacher): packed filter cached, no autograd node
    for p in m.parameters():
        p.requires_grad_(False)
    with torch.no_grad():
        yf = m(x)
    assert _rel(yf, yr) < 1e-4
    assert np.isfinite(float(yf.sum()))


@pytest.mark.parametrize("relu,with_res", [(True, True), (True, False), (False, True)])
def test_fused_bias_residual_relu_matches_torch(fm, relu, with_res):
    torch.manual_seed(9)
    N, C, H, W = 3, 70, 9, 13                      # HW = 117: scalar tail path
    z = torch.randn(N, C, H, W, device="cuda")
    b = torch.randn(C, device="cuda")
    r = torch.randn(N, C, H, W, device="cuda") if with_res else None
    dy = torch.randn(N, C, H, W, device="cuda")

    zr, br = z.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if with_res else None
    yr = zr + br.view(1, C, 1, 1)
    if with_res:
        yr = yr + rr
    if relu:
        yr = torch.relu(yr)
    yr.backward(dy)

    zs = z.clone().requires_grad_(True)
    bh = b.clone().requires_grad_(True)
    rh = r.clone().requires_grad_(True) if with_res else None
    yh = fm.bias_act(zs * 1.0, bh, residual=rh, relu=relu)     # zs*1.0: a non-leaf, like a conv output
    yh.backward(dy)
    assert torch.allclose(yh, yr, rtol=1e-6, atol=1e-6)
    assert torch.allclose(zs.grad, zr.grad, rtol=1e-6, atol=1e-6)
    assert torch.allclose(bh.grad, br.grad, rtol=1e-5, atol=1e-4)
    if with_res:
        assert torch.allclose(rh.grad, rr.grad, rtol=1e-6, atol=1e-6)
    # vectorised path (HW % 4 == 0), no autograd
    z4 = torch.randn(2, 8, 4, 8, device="cuda")
    want = torch.relu(z4 + b[:8].view(1, 8, 1, 1))
    got = fm.bias_act(z4.clone(), b[:8].clone(), relu=True)
    assert torch.equal(got, want)


def test_conv1x1_gemm_weight_gradient_matches_torch(fm):
    torch.manual_seed(11)
    N, C, M, H, W = 3, 48, 80, 7, 9
    x = torch.randn(N, C, H, W, device="cuda")
    w = torch.randn(M, C, 1, 1, device="cuda") * 0.1
    dy = torch.randn(N, M, H, W, device="cuda")
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    torch.nn.functional.conv2d(xr, wr).backward(dy)
    xh, wh = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yh = fm.conv1x1(xh, wh)
    yh.backward(dy)
    assert _rel(yh, torch.nn.functional.conv2d(x, w)) < 1e-5
    assert _rel(fm.conv1x1(x, w), yh) < 1e-6          # no-grad (teacher) route
    assert _rel(xh.grad, xr.grad) < 1e-5
    assert _rel(wh.grad, wr.grad) < 1e-5


@pytest.mark.parametrize("cin,cmid,cout,stride", [(64, 64, 256, 1), (256, 64, 256, 1), (256, 128, 512, 2)])
def test_bottleneck_fused_route_matches_plain_torch(fm, monkeypatch, cin, cmid, cout, stride):
    """The block as the harness runs it (GEMM 1x1s, HIP 3x3, bias/residual/ReLU tails,
    projection bias merged into the last pass) against the plain nn.Conv2d + F.relu
    route over the same parameters: output and every parameter / input gradient."""
    torch.manual_seed(5)
    blk = fm.Bottleneck(cin, cmid, cout, stride).cuda()
    with torch.no_grad():
        for p in blk.parameters():
            p.copy_(torch.randn_like(p) * (0.05 if p.dim() == 4 else 0.5))
    x = torch.randn(2, cin, 12, 20, device="cuda")
    dy = torch.randn(2, cout, 12 // stride, 20 // stride, device="cuda")

    def run(fused):
        monkeypatch.setattr(fm, "_FUSE_TAIL", fused)
        blk.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        y = blk(xi)
        y.backward(dy)
        return y.detach().clone(), xi.grad.clone(), [p.grad.clone() for p in blk.parameters()]

    y1, dx1, g1 = run(True)
    y0, dx0, g0 = run(False)
    assert _rel(y1, y0) < 2e-5
    assert _rel(dx1, dx0) < 2e-5
    for a, b in zip(g1, g0):
        assert _rel(a, b) < 2e-5
