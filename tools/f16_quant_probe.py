"""fp16 forward kernel vs workgroup count (slots = 2 x 256 CUs): throughput as a function of rounds."""
import sys, time, torch
sys.path.insert(0, ".")
import ssad_amd  # noqa
from ssad_amd import kernels as K
from tools.f16_probe import timeit

for N in (7, 8, 14, 15, 16, 21, 22, 29, 30):
    ci = co = 256; H, W = 80, 112
    x = torch.randn(N, ci, H, W, device="cuda")
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    b = torch.randn(co, device="cuda")
    xb = K.f16_pack_activations(x)
    w16, _ = K.f16_pack_filter(w, True, False)
    t = timeit(lambda: K.conv3x3_forward_f16(xb, w16, b, ci, co, relu=True))
    fl = 2.0 * 9 * N * H * W * ci * co / 1e9
    wgs = N * 35 * 2
    print("N=%2d workgroups %4d = %.2f rounds of 512: %.3f ms %5.0f TF/s" % (N, wgs, wgs / 512.0, t, fl / t))
