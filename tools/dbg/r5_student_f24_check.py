"""What does SSAD_STUDENT_F24 (the trained networks' 3x3 forward / data gradient on the F(2x4) engine) do to one whole
config-3 step?  Runs the step once per mode in a subprocess (the switch is read when the programs are built), same
seeded inputs, and compares losses, every gradient buffer and the updated parameters against mode 0.

    python tools/dbg/r5_student_f24_check.py [modes...]      (default: 0 8 9 11 15)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch
    import ssad_amd
    from ssad_amd import synth, program as PR
    from ssad_amd.head_pipeline import DistillHeads
    from ssad_amd.backbone_pipeline import NativeDistillModel
    from ssad_amd.modeling.retinanet_heads import HeadConfig
    mode = os.environ.get("SSAD_STUDENT_F24", "0")
    dev = "cuda"; N = int(os.environ.get("CHECK_N", "16")); shapes = synth.LEVEL_SHAPES_600
    heads = DistillHeads(HeadConfig(num_gpus=1), N=N, shapes=shapes, device=dev,
                         student_init=synth.head_params(np.random.default_rng(1)),
                         teacher_init=synth.head_params(np.random.default_rng(2)), lr=1e-4)
    model = NativeDistillModel(heads, "r50", "r101", N, (640, 896), dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    labels = []
    for h, w in shapes:
        u = torch.rand((N, 9, h, w), device=dev, generator=gen)
        lab = torch.zeros((N, 9, h, w), dtype=torch.int32, device=dev); lab[u < 0.05] = -1
        fg = (u >= 0.05) & (u < 0.07)
        lab[fg] = torch.randint(1, 81, (int(fg.sum()),), device=dev, generator=gen, dtype=torch.int32)
        labels.append(lab)
    tg = []; nfg = 0
    for lab in labels:
        idx = torch.nonzero(lab > 0)
        Lc = torch.stack([idx[:, 0], 4 * idx[:, 1], idx[:, 2], idx[:, 3]], dim=1).float().contiguous()
        tg.append(((torch.randn((Lc.shape[0], 4), device=dev, generator=gen) * 0.5).contiguous(), Lc)); nfg += Lc.shape[0]
    fgn = torch.tensor([float(nfg)], device=dev)
    images = torch.randn((N, 3, 640, 896), device=dev, generator=gen)
    model.step(images, labels, tg, fgn)
    torch.cuda.synchronize()
    eng_h = [op.i[4] for op in heads.prog.ops if op.code == PR.CONV3X3]
    eng_b = [op.i[4] for op in model.student.prog.ops if op.code == PR.CONV3X3]
    print("mode %s: subnet conv engines %s; backbone %d launches on F(2x4), %d on F(2x2)" % (
        mode, "".join(str(e) for e in eng_h), sum(e == 2 for e in eng_b), sum(e == 1 for e in eng_b)), flush=True)
    out = dict(losses=heads.losses.cpu(), focal=heads.focal_losses.cpu(), bbox=heads.bbox_losses.cpu(),
               g_heads=heads.grads.flat.cpu(), p_heads=heads.params.flat.cpu(),
               g_body=model.student.grads_flat.cpu(), p_body=model.student.params_flat.cpu(),
               d_fpn=[t.cpu() for t in heads.d_fpn["cls"]])
    torch.save(out, "/tmp/sf24_%s.pt" % os.environ.get("CHECK_TAG", mode))
    sys.exit(0)

import torch
modes = sys.argv[1:] or ["0", "8", "9", "11", "15"]
for m in modes:
    env = dict(os.environ, SSAD_STUDENT_F24=m.split("_")[0], CHECK_TAG=m)      # "0_again": the same mode twice
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, cwd=ROOT)
    if r.returncode:
        print("mode", m, "FAILED", r.returncode)
ref = torch.load("/tmp/sf24_0.pt")


def cmp(a, b):
    a, b = a.double(), b.double()
    d = (a - b)
    return "rel L2 %.2e, max|d|/max|ref| %.2e" % ((d.norm() / b.norm().clamp_min(1e-300)).item(),
                                                  (d.abs().max() / b.abs().max().clamp_min(1e-300)).item())


for m in modes:
    if m == "0" or not os.path.exists("/tmp/sf24_%s.pt" % m):
        continue
    o = torch.load("/tmp/sf24_%s.pt" % m)
    print("== mode %s against mode 0" % m)
    for k in ("losses", "focal", "bbox"):
        print("  %-8s %s   %s" % (k, cmp(o[k], ref[k]), [float(v) for v in o[k][:3]]))
    for k in ("g_heads", "g_body", "p_heads", "p_body"):
        print("  %-8s %s" % (k, cmp(o[k], ref[k])))
    print("  d_fpn(P3) %s" % cmp(o["d_fpn"][0], ref["d_fpn"][0]))
