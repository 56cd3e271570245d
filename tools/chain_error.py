"""Diagnostics: relative L2 error of every head gradient of the fused pipeline
vs the CPU oracle, for the direct and the Winograd engine (tiny problem)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import ssad_amd
from oracle import head_step
from ssad_amd.head_pipeline import DistillHeads
import test_gpu_operators as T
for seed, N in ((33, 1), (31, 2), (5, 2)):
    cfg, S, Tt, fs, ft, labs, tg, fg = T.small_problem(seed=seed, N=N)
    ref = head_step.head_step(S, Tt, fs, ft, labs, scale=cfg.loss_scale, bbox_targets=tg, fg_num=fg,
                              focal_gamma=cfg.focal_gamma, focal_alpha=cfg.focal_alpha, bbox_beta=cfg.bbox_reg_beta)
    dev = torch.device("cuda", 0)
    t = lambda arrs: [torch.from_numpy(a).to(dev) for a in arrs]
    for eng in ("direct", "winograd"):
        os.environ["SSAD_CONV_ENGINE"] = eng
        h = DistillHeads(cfg, N=N, shapes=T.SHAPES, device=dev, student_init=S, teacher_init=Tt)
        h.step(t(fs), t(ft), t(labs), update=False, bbox_targets=[tuple(t(p)) for p in tg],
               fg_num=torch.from_numpy(fg).to(dev))
        errs = {k: float(np.linalg.norm(h.grads[k].cpu().numpy() - g) / np.linalg.norm(g)) for k, g in ref["grads"].items()}
        worst = max(errs, key=errs.get)
        act = [float(np.linalg.norm(h.act["cls"][i][0].cpu().numpy() - 0) ) for i in range(4)]
        print("seed", seed, eng, "worst", worst, "%.2e" % errs[worst], " cls_n0_w %.2e  cls_pred_w %.2e  bbox_n0_w %.2e" % (
            errs["retnet_cls_conv_n0_fpn3_w"], errs["retnet_cls_pred_fpn3_w"], errs["retnet_bbox_conv_n0_fpn3_w"]))
