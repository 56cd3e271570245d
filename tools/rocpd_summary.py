#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) capture: per-kernel calls / total / avg /
min / max duration (the `--stats` view), and per-kernel PMC counter sums when
the capture holds counters.  Usage: rocpd_summary.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (
            n, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / total))
    try:
        pm = cur.execute("select * from counters_collection limit 1").fetchall()
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        if pm:
            kn = [c for c in ccols if "kernel" in c and "name" in c] or [c for c in ccols if c == "name"]
            cn = [c for c in ccols if "counter" in c and "name" in c]
            vn = [c for c in ccols if c in ("value", "counter_value")]
            if kn and cn and vn:
                q = "select %s, %s, sum(%s), count(*) from counters_collection group by 1, 2" % (kn[0], cn[0], vn[0])
                lines += ["", "| kernel | counter | sum | dispatches |", "|---|---|---|---|"]
                for k, c, v, n in cur.execute(q):
                    lines.append("| %s | %s | %.6g | %d |" % (short(k), c, v, n))
            else:
                lines.append("\ncounters_collection columns: %s" % ccols)
    except sqlite3.Error as e:
        lines.append("\n(no counters: %s)" % e)
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
