"""Round 4: soak -- config 3 at full size, STEPS unsynchronised iterations under bench.py's schedule against the
same iterations on one stream, bit for bit (tests/test_gpu_full_size.py does three).  usage: r4_soak.py [steps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ssad_amd  # noqa
from ssad_amd import synth
import test_gpu_full_size as T

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
f16 = len(sys.argv) > 2 and sys.argv[2] == "f16"
if f16:
    N, hw, shapes, archs = 16, (512, 768), synth.LEVEL_SHAPES_500, ("r101", "x101-64x4d")
else:
    N, hw, shapes, archs = 16, (640, 896), synth.LEVEL_SHAPES_600, ("r50", "r101")
batch = T._inputs(N, shapes, hw, seed=99)
a = T._build(archs[0], archs[1], N, hw, shapes, f16=f16, overlap=True)
T._run(a, batch, steps, high_priority=True)
b = T._build(archs[0], archs[1], N, hw, shapes, f16=f16, overlap=False)
T._run(b, batch, steps, high_priority=False)
ok = True
for x, y, what in ((a.heads.losses, b.heads.losses, "losses"), (a.heads.params.flat, b.heads.params.flat, "subnet parameters"),
                   (a.student.params_flat, b.student.params_flat, "backbone parameters"),
                   (a.student.moms_flat, b.student.moms_flat, "backbone momentum")):
    same = torch.equal(x, y)
    ok &= same and bool(torch.isfinite(x).all())
    print(what, "SAME" if same else "DIFFER max %.3e" % float((x - y).abs().max()), "finite", bool(torch.isfinite(x).all()))
print("soak %s, %d steps: %s; losses %s" % ("cfg5 f16" if f16 else "cfg3 f32", steps, "OK" if ok else "FAILED",
                                          [round(float(v), 5) for v in a.heads.losses]))
