#!/usr/bin/env python3
"""Every dispatch of the kernels whose name contains PATTERN inside the last step of a rocprofv3 capture
(step = period of a once-per-step marker kernel), in start order: start (us from the step's begin), duration, grid.
  python tools/dbg/list_launches.py x.db wino_wgrad_kernel [marker=cls_losses_fused_kernel]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else "cls_losses_fused_kernel"
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
q = "select name, start, end%s from kernels order by start" % ((", " + gx) if gx else "")
rows = cur.execute(q).fetchall()
ends = [r[2] for r in rows if marker in r[0]]
lo, hi = ends[-2], ends[-1]
sel = [r for r in rows if r[1] >= lo and r[2] <= hi]
t0 = sel[0][1]
tot = 0.0
for r in sel:
    if pat in r[0]:
        d = (r[2] - r[1]) / 1e3
        tot += d
        print("%9.1f us  +%8.1f us  grid %s  %s" % ((r[1] - t0) / 1e3, d, r[3] if gx else "?", r[0][:60]))
print("total %.1f us in a step of %.1f us" % (tot, (hi - lo) / 1e3))
