"""Round 4: stress the Winograd forward kernel for the timing-dependent corruption the full-size race
test found in res2's 64 -> 64 layers.  The layer runs repeatedly on one stream while other streams keep
the chip busy (GEMM-shaped and streaming work); every output is compared bit for bit with the output
of a quiet run.  usage: r4_wino_stress.py [Cin Cout H W iters]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ssad_amd  # noqa
from ssad_amd import kernels as K

Cin, Cout, H, W, iters = (int(v) for v in (sys.argv[1:6] + ["64", "64", "160", "224", "300"][len(sys.argv) - 1:]))
N = 16
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((N, Cin, H, W), device="cuda", generator=g)
w = torch.randn((Cout, Cin, 3, 3), device="cuda", generator=g) / (3 * Cin ** 0.5)
b = torch.randn((Cout,), device="cuda", generator=g)
pf, _ = K.conv_wino_pack_filter(w, True, False)
ref = K.conv3x3_forward([x], pf, b, Cout, relu=True, wino=True)[0].clone()
torch.cuda.synchronize()
for _ in range(3):
    y = K.conv3x3_forward([x], pf, b, Cout, relu=True, wino=True)[0]
    assert torch.equal(y, ref), "quiet run not reproducible"
print("quiet: reproducible; NHALF env", os.environ.get("SSAD_WINO_NHALF"), flush=True)

# background load: a pointwise GEMM (gemm_conv_nn) and a big elementwise pass on two other streams
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
xa = torch.randn((N, 256, H // 2, W // 2), device="cuda", generator=g)
wa = torch.randn((1024, 256), device="cuda", generator=g)
big = torch.randn((64, 1024, 1024), device="cuda", generator=g)
x2 = torch.randn((N, 64, H, W), device="cuda", generator=g)
pf2, _ = K.conv_wino_pack_filter(torch.randn((64, 64, 3, 3), device="cuda", generator=g), True, False)
out = torch.empty_like(ref)
bad = 0
first = None
t0 = time.time()
for it in range(iters):
    with torch.cuda.stream(s1):
        for _ in range(3):
            K.conv1x1_bias_act(xa, wa.view(1024, 256, 1, 1), None, None, relu=True)
            K.conv3x3_forward([x2], pf2, None, 64, relu=True, wino=True)      # the other network's same layer
    with torch.cuda.stream(s2):
        big.mul_(1.0000001)
    K.conv3x3_forward([x], pf, b, Cout, relu=True, wino=True, out=[out])
    if (it & 7) == 7 or True:
        eq = torch.equal(out, ref)
        if not eq:
            bad += 1
            if first is None:
                nz = torch.nonzero(out != ref)
                first = (it, int((out != ref).sum()), nz.min(0).values.tolist(), nz.max(0).values.tolist(),
                         float((out - ref).abs().max()))
torch.cuda.synchronize()
print("Cin %d Cout %d %dx%d: %d / %d iterations differ (%.1f s); first: %s" % (Cin, Cout, H, W, bad, iters,
                                                                              time.time() - t0, first), flush=True)
