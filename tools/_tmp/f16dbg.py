import sys
import numpy as np, torch
sys.path.insert(0, ".")
import ssad_amd
from ssad_amd import synth, kernels as K
from ssad_amd.head_pipeline import DistillHeads, DistillHeadsF16
from ssad_amd.modeling import retinanet_heads as rh
rng = np.random.default_rng(77)
shapes = [(20, 28), (10, 14), (5, 7), (3, 4), (2, 2)]
N = 2
cfg = rh.HeadConfig(num_gpus=1)
S, T = synth.head_params(rng), synth.head_params(rng)
for P in (S, T):
    for k in P:
        if k.endswith("_w"):
            P[k] = (P[k] * 3).astype(np.float32)
fs, ft = synth.fpn_features(rng, N, shapes), synth.fpn_features(rng, N, shapes)
print("fpn feature stats", [float(np.abs(f).max()) for f in fs], float(fs[0].std()))
labs = []
for h, w in shapes:
    lab = synth.distill_inputs(rng, N, 9, 80, h, w)[2]
    u = rng.random(lab.shape)
    lab[u < 0.1] = rng.integers(1, 81, size=int((u < 0.1).sum()))
    labs.append(lab)
tg = [synth.bbox_targets(rng, l) for l in labs]
fg = np.array([float(sum(t[0].shape[0] for t in tg))], np.float32)
dev = torch.device("cuda", 0)
t = lambda arrs: [torch.from_numpy(a).to(dev) for a in arrs]
H = {}
for name, cls in (("f32", DistillHeads), ("f16", DistillHeadsF16)):
    h = cls(cfg, N=N, shapes=shapes, device=dev, student_init=S, teacher_init=T)
    h.step(t(fs), t(ft), t(labs), update=False, bbox_targets=[tuple(t(p)) for p in tg],
           fg_num=torch.from_numpy(fg).to(dev))
    H[name] = h
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for k, _, _, _ in H["f32"].params.specs:
    print("%-34s %.5f   |g| %.3e" % (k, rel(H["f16"].grads[k], H["f32"].grads[k]), float(H["f32"].grads[k].abs().max())))
for tw in ("cls", "bbox"):
    for l in range(5):
        print("d_fpn", tw, l, rel(H["f16"].d_fpn[tw][l], H["f32"].d_fpn[tw][l]))
# direct check: n0 wgrad from the f16 pipeline's own tensors, in float64 on the host
h = H["f16"]
name = "retnet_cls_conv_n0_fpn3"
# dy entering layer 0 is not kept; recompute dW with the fp32 wgrad from unpacked tensors of level 0 only as a sanity check
xb = h.in_blk["student"][0]
x = K.f16_unpack_activations(xb, 256)
print("pack error of fpn level 0:", rel(x, torch.from_numpy(fs[0]).to(dev)))
