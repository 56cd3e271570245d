"""Is the chip clock- / power-limited under these kernels?  Runs one launch shape back to back for a few seconds
and samples rocm-smi (sclk, power) from a second thread while it runs; then idle."""
import subprocess, sys, threading, time
import torch
sys.path.insert(0, ".")
import ssad_amd
from ssad_amd import kernels as K

def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if ("sclk" in l or "Power" in l or "mclk" in l)]
        return " | ".join(keep)[:400]
    except Exception as e:
        return repr(e)

g = torch.Generator(device="cuda").manual_seed(3)
N = 16
lv = [(80, 112), (40, 56), (20, 28), (10, 14), (5, 7)]
def tower(zero=False):
    xs, packs, outs = [], [], []
    for _ in range(4):
        w = torch.randn(256, 256, 3, 3, device="cuda", generator=g) * 0.02
        pf = K.conv_wino24_pack_filter(w)
        for h, wd in lv:
            x = torch.randn(N, 256, h, wd, device="cuda", generator=g)
            if zero: x.zero_()
            xs.append(x); packs.append(pf); outs.append(torch.empty(N, 256, h, wd, device="cuda"))
    arr = K._conv_levels(xs, outs, None, packs, [None] * len(xs))
    L = K.lib()
    return lambda: K._check(L.ssad_conv3x3_forward_wino24(arr, len(xs), K._ptr(packs[0]), None, 256, 256, K.CONV_RELU, K._stream()), "w24")
def gemm():
    x = torch.randn(N, 256, 40, 56, device="cuda", generator=g)
    wt = K.transpose_filter(torch.randn(1024, 256, 1, 1, device="cuda", generator=g) * 0.05)
    y = torch.empty(N, 1024, 40, 56, device="cuda")
    return lambda: K.conv1x1_forward(x, wt, 1024, out=y)
for name, fn in (("wino24 tower depth", tower()), ("wino24 tower depth, zero activations", tower(True)), ("gemm_conv_nn 256->1024", gemm())):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    samples = []
    stop = False
    def sampler():
        time.sleep(1.0)
        while not stop:
            samples.append(smi()); time.sleep(0.7)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 4.0:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    dt = time.perf_counter() - t0
    stop = True; th.join()
    print("%s: %.3f ms per launch sustained" % (name, dt / n * 1e3))
    for s in samples[:3]: print("   ", s)
time.sleep(2)
print("idle:", smi())
