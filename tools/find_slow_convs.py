#!/usr/bin/env python3
"""Harness diagnostics: time fwd / dgrad / wgrad of every distinct conv config
of the R-50-FPN student backbone under MIOpen (NHWC and NCHW)."""
import sys, os, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssad_amd
from tools.harness.full_model import ResNetFPN

def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

m = ResNetFPN(50).cuda()
x = torch.randn(16, 3, 640, 896, device="cuda")
shapes = {}
hooks = []
def mk(name, mod):
    def hook(mod_, inp, out):
        key = (mod.in_channels, mod.out_channels, mod.kernel_size, mod.stride, mod.padding, tuple(inp[0].shape[2:]))
        shapes.setdefault(key, []).append(name)
    return hook
for name, mod in m.named_modules():
    if isinstance(mod, torch.nn.Conv2d):
        hooks.append(mod.register_forward_hook(mk(name, mod)))
with torch.no_grad():
    m(x)
for fmt_name, fmt in (("NHWC", torch.channels_last), ("NCHW", torch.contiguous_format)):
    print("==", fmt_name)
    for key, names in sorted(shapes.items(), key=lambda kv: -kv[0][5][0]):
        cin, cout, ks, st, pad, hw = key
        xi = torch.randn(16, cin, *hw, device="cuda").contiguous(memory_format=fmt).requires_grad_(True)
        w = torch.randn(cout, cin, *ks, device="cuda").contiguous(memory_format=fmt).requires_grad_(True)
        y = F.conv2d(xi, w, None, st, pad)
        gy = torch.randn_like(y)
        tf = t(lambda: F.conv2d(xi, w, None, st, pad))
        tw = t(lambda: torch.autograd.grad(F.conv2d(xi, w, None, st, pad), w, gy))
        td = t(lambda: torch.autograd.grad(F.conv2d(xi, w, None, st, pad), xi, gy))
        fl = 2.0 * cin * cout * ks[0] * ks[1] * 16 * y.shape[2] * y.shape[3] / 1e9
        flag = "  <== SLOW" if max(tw - tf, td - tf) > 5 * max(tf, 0.05) else ""
        print("%4d->%4d k%d s%d in %3dx%3d x%2d  %7.1f GF  fwd %7.2f  fwd+wgrad %7.2f  fwd+dgrad %7.2f ms%s" % (
            cin, cout, ks[0], st[0], hw[0], hw[1], len(names), fl, tf, tw, td, flag))
