"""Round 4: first differing backbone buffer (forward order) between the two-stream and the serial step."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools", "dbg"))
import ssad_amd  # noqa
import test_gpu_full_size as T
from r4_race_hunt import build, batch  # noqa


def fwd_buffers(net, tag):
    out = [(tag + ".image", net.image), (tag + ".stem_z", net.stem_z)]
    for pre, sv in net.saved.items():
        for k in ("xs", "y1", "y2", "y"):
            out.append(("%s.%s.%s" % (tag, pre, k), sv[k]))
    for k in ("t5", "t4", "t3"):
        out.append(("%s.%s" % (tag, k), net._fpn_saved[k]))
    for l in range(5):
        out.append(("%s.fpn%d" % (tag, l), net.fpn[l]))
    return out


ref = build(False, False, False)
T._run(ref, batch, 1, high_priority=False)
rb = fwd_buffers(ref.teacher, "T") + fwd_buffers(ref.student, "S")
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    a = build(True, False, False)
    T._run(a, batch, 1, high_priority=False)
    ab = fwd_buffers(a.teacher, "T") + fwd_buffers(a.student, "S")
    msgs = []
    for (n1, x), (n2, y) in zip(ab, rb):
        if x.shape != y.shape or torch.equal(x, y):
            continue
        nz = x != y
        idx = torch.nonzero(nz)
        lo, hi = idx.min(0).values.tolist(), idx.max(0).values.tolist()
        d = (x - y).abs()
        msgs.append("%s: %d/%d differ, box %s..%s, max|d| %.3e (max|ref| %.3e), nan %d" % (
            n1, int(nz.sum()), x.numel(), lo, hi, float(torch.nan_to_num(d).max()), float(y.abs().max()),
            int(torch.isnan(x).sum())))
    print("trial", trial, "SAME" if not msgs else "DIFF (%d buffers), first 3 in forward order:\n   %s" % (
        len(msgs), "\n   ".join(msgs[:3])), flush=True)
    del a
    torch.cuda.empty_cache()
