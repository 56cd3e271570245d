"""Launches of one hardware queue inside one step period, with the idle gaps in front of them (rocprofv3 rocpd capture).
  python tools/dbg/queue_gaps.py results.db [marker=cls_losses_fused_kernel] [min gap us=100] [steps back=6]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "cls_losses_fused_kernel"
ming = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
back = int(sys.argv[4]) if len(sys.argv) > 4 else 6      # bench.py ends with three synchronised steps: stay clear of them
rows = db.execute("select name, start, end, queue_id from kernels order by start").fetchall()
marks = [r for r in rows if marker in r[0]]
t0, t1 = marks[-2 - back][1], marks[-1 - back][1]
main_q = marks[-2 - back][3]
short = lambda n: re.sub(r"[<(].*", "", re.sub(r"\(anonymous namespace\)::|^void ", "", n))[:34]
qs = sorted({r[3] for r in rows if t0 <= r[1] < t1})
for q in qs:
    cur = [r for r in rows if t0 <= r[1] < t1 and r[3] == q]
    print("queue %s%s: %d launches" % (q, " (main)" if q == main_q else "", len(cur)))
    prev = t0
    for n, s, e, _ in cur:
        if (s - prev) / 1e3 >= ming:
            print("   idle %8.1f us before %-34s at %7.2f ms" % ((s - prev) / 1e3, short(n), (s - t0) / 1e6))
        prev = max(prev, e)
