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
    a, b = a.double(), b.double()
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
