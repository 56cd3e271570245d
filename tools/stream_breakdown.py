#!/usr/bin/env python3
"""Per-stream (queue) kernel time of a rocprofv3 rocpd capture, for the last K step periods:
which stream is the critical path of an overlapped step.  usage: stream_breakdown.py x.db [marker] [K]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "sgd_flat_kernel"
K = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
print("kernels view columns:", cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = cur.execute("select name, start, end%s from kernels" % ((", " + qcol) if qcol else "")).fetchall()
ends = sorted(r[2] for r in rows if marker in r[0])
lo, hi = ends[-2 * K - 1], ends[-1]          # the heads' and the backbone's SGD per step: 2 markers per step
rows = [r for r in rows if r[1] >= lo and r[2] <= hi]
span = (hi - lo) / 1e6
per = defaultdict(lambda: [0.0, 0, defaultdict(float)])
for r in rows:
    q = r[3] if qcol else 0
    per[q][0] += (r[2] - r[1]) / 1e6
    per[q][1] += 1
    per[q][2][r[0].split("(")[0][-60:]] += (r[2] - r[1]) / 1e6
print("window %.2f ms (%d steps)" % (span, K))
for q, (ms, n, names) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    top = sorted(names.items(), key=lambda kv: -kv[1])[:6]
    print("queue %s: %.2f ms busy (%.0f %% of window), %d kernels; top: %s" % (
        q, ms, 100 * ms / span, n, ", ".join("%s %.1f" % (k.strip(), v) for k, v in top)))
