#!/usr/bin/env python3
"""Coarse timeline of ONE step of a rocprofv3 rocpd capture: per hardware queue, the busy share of every time bin
(which stream runs alone when -- e.g. the teacher's forward pass after the student's has finished).
usage: step_timeline.py x.db [bin_ms] [marker] [back]
(a step = the span between two consecutive `marker` kernels; `back` = how many steps before the last one -- bench.py
ends with three synchronised steps for its host-enqueue figure, so the steady state is e.g. back = 6)"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
bin_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
marker = sys.argv[3] if len(sys.argv) > 3 else "cls_losses_fused_kernel"
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = cur.execute("select name, start, end, %s from kernels" % qcol).fetchall()
marks = sorted(r[1] for r in rows if marker in r[0])
back = int(sys.argv[4]) if len(sys.argv) > 4 else 0
lo, hi = marks[-2 - back], marks[-1 - back]
rows = [r for r in rows if r[2] > lo and r[1] < hi]
nb = int((hi - lo) / 1e6 / bin_ms) + 1
busy = defaultdict(lambda: [0.0] * nb)
first = {}
for name, s, e, q in rows:
    s, e = max(s, lo), min(e, hi)
    b0, b1 = int((s - lo) / 1e6 / bin_ms), int((e - lo) / 1e6 / bin_ms)
    for b in range(b0, min(b1, nb - 1) + 1):
        bs, be = lo + b * bin_ms * 1e6, lo + (b + 1) * bin_ms * 1e6
        busy[q][b] += max(0.0, min(e, be) - max(s, bs)) / 1e6
    first.setdefault(q, name.split("(")[0][-40:])
print("step %.2f ms (from one %s to the next), bins of %.1f ms, busy %% per queue" % ((hi - lo) / 1e6, marker, bin_ms))
for q in sorted(busy, key=lambda q: -sum(busy[q])):
    print("queue %-3s %6.1f ms | %s" % (q, sum(busy[q]), " ".join("%3d" % round(100 * v / bin_ms) for v in busy[q])))
tot = [sum(busy[q][b] for q in busy) for b in range(nb)]
print("sum           | %s" % " ".join("%3d" % round(100 * v / bin_ms) for v in tot))
# per bin: the kernel names with the most time (what phase is this)
import re


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"[<(].*", "", n)[:28]


top = defaultdict(lambda: defaultdict(float))
for name, s, e, q in rows:
    b = int((max(s, lo) - lo) / 1e6 / bin_ms)
    top[b][short(name)] += (e - s) / 1e6
for b in range(nb):
    t = sorted(top[b].items(), key=lambda kv: -kv[1])[:2]
    print("bin %2d: %s" % (b, ", ".join("%s %.1f" % kv for kv in t)))

# the busiest queue's kernels by total time in this step (the critical path's make-up)
main_q = max(busy, key=lambda q: sum(busy[q]))
acc = defaultdict(lambda: [0.0, 0])
for name, s, e, q in rows:
    if q == main_q:
        acc[short(name)][0] += (min(e, hi) - max(s, lo)) / 1e6
        acc[short(name)][1] += 1
print("queue %s by kernel:" % main_q)
for k, (ms, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:24]:
    print("  %-30s %4d launches %7.2f ms" % (k, n, ms))
