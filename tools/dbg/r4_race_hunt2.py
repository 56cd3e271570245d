"""Round 4: locate the first differing buffer of the intermittent full-size difference (two_streams, 1 step)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools", "dbg"))
import ssad_amd  # noqa
from ssad_amd import synth
import test_gpu_full_size as T
from r4_race_hunt import build, batch  # noqa


def buffers(m):
    h, st, te = m.heads, m.student, m.teacher
    out = []
    for l in range(5):
        out.append(("teacher.fpn%d" % l, te.fpn[l]))
        out.append(("student.fpn%d" % l, st.fpn[l]))
    out.append(("normalizer", h.normalizer))
    for t in ("cls", "bbox"):
        for i, lv in enumerate(h.act[t]):
            for l, x in enumerate(lv):
                out.append(("act.%s.n%d.L%d" % (t, i, l), x))
    for nm in ("cls_logits", "bbox_pred", "t_prob", "d_cls_logits", "d_bbox_pred"):
        for l, x in enumerate(getattr(h, nm)):
            out.append(("%s.L%d" % (nm, l), x))
    for t in ("cls", "bbox"):
        for i, lv in enumerate(h.dbuf[t]):
            for l, x in enumerate(lv):
                out.append(("dbuf.%s.%d.L%d" % (t, i, l), x))
        for l, x in enumerate(h.d_fpn[t]):
            out.append(("d_fpn.%s.L%d" % (t, l), x))
    for name, _, _, _ in h.params.specs:
        out.append(("headgrad." + name, h.grads[name]))
    for l in range(5):
        out.append(("student.d_fpn%d" % l, st.d_fpn[l]))
    for pre, sv in st.saved.items():
        for k in ("y1", "y2", "y"):
            out.append(("saved.%s.%s" % (pre, k), sv[k]))
    for lname, la in st._layers.items():
        if la.train:
            out.append(("bbgrad." + lname, la.gw))
    return out


ref = build(False, False, False)
T._run(ref, batch, 1, high_priority=False)
rb = buffers(ref)
mode = sys.argv[1] if len(sys.argv) > 1 else "ts"
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    a = build(True, mode == "all", mode == "all")
    T._run(a, batch, 1, high_priority=(mode == "all"))
    bad = []
    for (n1, x), (n2, y) in zip(buffers(a), rb):
        assert n1 == n2
        if not torch.equal(x, y):
            nz = (x != y)
            idx = torch.nonzero(nz)
            bad.append("%s:%d/%d first=%s last=%s" % (n1, int(nz.sum()), x.numel(), idx[0].tolist(), idx[-1].tolist()))
    print("trial", trial, "SAME" if not bad else "DIFF %d buffers; first 10: %s" % (len(bad), bad[:10]), flush=True)
    del a
    torch.cuda.empty_cache()
