import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "7")
import numpy as np, torch
import ssad_amd
from ssad_amd import synth
from ssad_amd.head_pipeline import DistillHeads
from ssad_amd.backbone_pipeline import NativeDistillModel
from ssad_amd.modeling.retinanet_heads import HeadConfig
dev = "cuda"; N = 16; shapes = synth.LEVEL_SHAPES_600
heads = DistillHeads(HeadConfig(num_gpus=1), N=N, shapes=shapes, device=dev, student_init=synth.head_params(np.random.default_rng(1)),
                     teacher_init=synth.head_params(np.random.default_rng(2)), lr=1e-4)
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
for _ in range(3): model.step(images, labels, tg, fgn)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); model.step(images, labels, tg, fgn); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print("host enqueue with an empty queue, ms:", [round(t * 1e3, 2) for t in ts])
# piecewise
h, st, te = model.heads, model.student, model.teacher
def T(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0; torch.cuda.synchronize(); return round(dt * 1e3, 2)
print("pack", T(lambda: (h.pack_student(), st.pack())), "teacher fwd", T(lambda: te.forward(images)), "student fwd", T(lambda: st.forward(images)),
      "heads fwd", T(lambda: h.forward_all(te.fpn, st.fpn)))
# steady state (queue full, no synchronize between steps): where does the launching thread spend its time, and
# when, relative to the start of step(), does it reach each phase -- a phase the host reaches late is a phase the
# GPU cannot have started early
import collections
acc = collections.OrderedDict(); at = collections.OrderedDict()
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); t1 = time.perf_counter()
        acc[label] = acc.get(label, 0.0) + (t1 - t0); at[label] = at.get(label, 0.0) + (t0 - step_t0[0])
        return r
    setattr(obj, name, g)
step_t0 = [0.0]
wrap(te, "forward", "teacher.forward"); wrap(h, "pack_student", "heads.pack"); wrap(st, "pack", "student.pack")
wrap(st, "forward", "student.forward"); wrap(h, "forward_all", "heads.forward"); wrap(h, "bbox_losses_fwd_bwd", "losses")
wrap(h, "backward", "heads.backward"); wrap(st, "backward", "student.backward"); wrap(h, "sgd_step", "heads.sgd")
wrap(st, "sgd_step", "student.sgd")
for _ in range(3): model.step(images, labels, tg, fgn)
acc.clear(); at.clear()
K_ = 20
t_all = time.perf_counter()
for _ in range(K_):
    step_t0[0] = time.perf_counter(); model.step(images, labels, tg, fgn)
host_total = time.perf_counter() - t_all
torch.cuda.synchronize()
wall = time.perf_counter() - t_all
print("steady state: %.2f ms/step wall, host inside step() %.2f ms/step" % (wall / K_ * 1e3, host_total / K_ * 1e3))
for k in acc:
    print("  %-18s reached at %7.2f ms, takes %7.2f ms" % (k, at[k] / K_ * 1e3, acc[k] / K_ * 1e3))
