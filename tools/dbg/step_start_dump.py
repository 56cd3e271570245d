"""List the kernels around the start of the last step of a rocpd capture (every queue): name, queue, start and end in
ms relative to the step's backbone SGD launch.  usage: step_start_dump.py x.db [before_ms] [after_ms]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
before = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
after = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = cur.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
sg = [r for r in rows if "sgd_flat_kernel" in r[0]]
t0 = sg[-3][1]        # the backbone SGD of the step before the last one
import re
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*", "", n)[:44]
for n, s, e, q in rows:
    if t0 - before * 1e6 <= s <= t0 + after * 1e6:
        print("q%-2s %8.3f .. %8.3f  (%7.1f us)  %s" % (q, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, short(n)))
