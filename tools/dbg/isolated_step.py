"""Steady-state vs isolated step time (power / clock effects): runs bench-like steps back to back, then
one at a time with idle gaps."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "7")
import numpy as np, torch
import ssad_amd
from ssad_amd import synth
from ssad_amd.head_pipeline import DistillHeads, DistillHeadsF16
from ssad_amd.backbone_pipeline import NativeDistillModel
from ssad_amd.modeling.retinanet_heads import HeadConfig
f16 = len(sys.argv) > 1 and sys.argv[1] == "f16"
dev = "cuda"; N = 16; shapes = synth.LEVEL_SHAPES_600
kw = dict(blocked_io=True) if f16 else {}
heads = (DistillHeadsF16 if f16 else DistillHeads)(HeadConfig(num_gpus=1), N=N, shapes=shapes, device=dev,
        student_init=synth.head_params(np.random.default_rng(1)), teacher_init=synth.head_params(np.random.default_rng(2)), lr=1e-4, **kw)
model = NativeDistillModel(heads, "r50", "r101", N, (640, 896), dev)
gen = torch.Generator(device=dev).manual_seed(1)
labels = []
for h, w in shapes:
    u = torch.rand((N, 9, h, w), device=dev, generator=gen)
    lab = torch.zeros((N, 9, h, w), dtype=torch.int32, device=dev); lab[u < 0.05] = -1
    fg = (u >= 0.05) & (u < 0.07); lab[fg] = torch.randint(1, 81, (int(fg.sum()),), device=dev, generator=gen, dtype=torch.int32)
    labels.append(lab)
tg = []; nfg = 0
for lab in labels:
    idx = torch.nonzero(lab > 0)
    Lc = torch.stack([idx[:, 0], 4 * idx[:, 1], idx[:, 2], idx[:, 3]], dim=1).float().contiguous()
    tg.append(((torch.randn((Lc.shape[0], 4), device=dev, generator=gen) * 0.5).contiguous(), Lc)); nfg += Lc.shape[0]
fgn = torch.tensor([float(nfg)], device=dev)
images = torch.randn((N, 3, 640, 896), device=dev, generator=gen)
def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        return " | ".join(l.strip() for l in out.splitlines() if ("sclk" in l or "Power" in l))[:300]
    except Exception as e:
        return str(e)
for _ in range(5): model.step(images, labels, tg, fgn)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): model.step(images, labels, tg, fgn)
print("during sustained run:", smi())
torch.cuda.synchronize()
print("steady state: %.2f ms/step" % ((time.perf_counter() - t0) / 30 * 1e3))
for gap in (0.0, 0.05, 0.3):
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); time.sleep(gap)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); model.step(images, labels, tg, fgn); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("isolated steps after %.2f s idle: %s ms (GPU time first launch -> last)" % (gap, [round(t, 1) for t in ts]))
print("idle:", smi())
