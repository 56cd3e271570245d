"""One image at config 3's full map sizes (P3 80x112 ... P7 5x7): the subnets step on F(2x4) and on F(2x2) against the
oracle composition -- activations on the other side of zero, losses, logits, every gradient tensor."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")): sys.path.insert(0, p)
import ssad_amd
from ssad_amd import synth
from oracle import head_step
from ssad_amd.head_pipeline import DistillHeads
import test_gpu_operators as T
import ssad_amd.modeling.retinanet_heads as rh
shapes = synth.LEVEL_SHAPES_600
rng = np.random.default_rng(77)
cfg = rh.HeadConfig(num_gpus=1)
S, Tt = synth.head_params(rng), synth.head_params(rng)
for P in (S, Tt):
    for k in P:
        if k.endswith("_w"): P[k] = (P[k] * 3).astype(np.float32)
N = 1
fs = synth.fpn_features(rng, N, shapes); ft = synth.fpn_features(rng, N, shapes)
labs = []
for h, w in shapes:
    lab = synth.distill_inputs(rng, N, 9, 80, h, w)[2]
    u = rng.random(lab.shape); lab[u < 0.02] = rng.integers(1, 81, size=int((u < 0.02).sum()))
    labs.append(lab)
tg = [synth.bbox_targets(rng, l) for l in labs]
fg = np.array([float(sum(t[0].shape[0] for t in tg))], np.float32)
t0 = time.time()
ref = head_step.head_step(S, Tt, fs, ft, labs, scale=cfg.loss_scale, bbox_targets=tg, fg_num=fg, focal_gamma=cfg.focal_gamma, focal_alpha=cfg.focal_alpha, bbox_beta=cfg.bbox_reg_beta)
acts = T.oracle_tower_acts(S, fs)
print("oracle %.1f s" % (time.time() - t0), flush=True)
dev = torch.device("cuda", 0)
t = lambda arrs: [torch.from_numpy(a).to(dev) for a in arrs]
for mode in ("15", "0"):
    os.environ["SSAD_STUDENT_F24"] = mode
    h = DistillHeads(cfg, N=N, shapes=shapes, device=dev, student_init=S, teacher_init=Tt)
    losses = h.step(t(fs), t(ft), t(labs), update=False, bbox_targets=[tuple(t(p)) for p in tg], fg_num=torch.from_numpy(fg).to(dev))
    flips = T.count_flips(lambda tw, d, l: h.act[tw][d][l].cpu().numpy(), acts)
    tot = sum(a.size for tw in acts.values() for d in tw for a in d)
    rel = {k: float(np.linalg.norm(h.grads[k].cpu().numpy() - g) / np.linalg.norm(g)) for k, g in ref["grads"].items()}
    mx = {k: float(np.abs(h.grads[k].cpu().numpy() - g).max() / np.abs(g).max()) for k, g in ref["grads"].items()}
    lrel = np.abs(losses.cpu().numpy() - ref["losses"]) / np.abs(ref["losses"])
    lg = max(float(np.abs(h.cls_logits[i].cpu().numpy() - ref["cls_logits"][i]).max() / np.abs(ref["cls_logits"][i]).max()) for i in range(5))
    dfp = max(float(np.linalg.norm(h.d_fpn[tw][i].cpu().numpy() - ref["d_fpn"][tw][i]) / np.linalg.norm(ref["d_fpn"][tw][i])) for tw in ("cls", "bbox") for i in range(5))
    print("mode %s: flips %d of %d; losses rel %.1e; logits max err %.1e of max; grads rel L2 worst %.2e (%s) median %.2e; worst entry %.2e; d_fpn worst rel L2 %.2e" % (
        mode, flips, tot, lrel.max(), lg, max(rel.values()), max(rel, key=rel.get), float(np.median(list(rel.values()))), max(mx.values()), dfp), flush=True)
