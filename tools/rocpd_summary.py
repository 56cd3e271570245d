#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) capture.

  rocpd_summary.py results.db [out.md] [--json out.json] [--title "..."]

Writes the `--stats` view (per-kernel calls / total / avg / min / max
duration) and, when the capture holds PMC counters, the per-kernel,
per-dispatch average of every counter."""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:80]


def class_table(rows, f16):
    """One kernel serves several timing classes (wino24_conv_kernel: tower forward = 23, data gradients = 24, ...): the
    launches of a subnets-only capture are told apart by their order inside a step, the way tools/pmc_by_class.py
    attributes counters (the program of the step gives, per kernel name, the class of its 1st, 2nd, ... launch).  The
    avg per CALL of class 23 is what bench.py reports as roofline.avg_launch_ms."""
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import pmc_by_class as P
    from ssad_amd import program as PR
    seq = P.class_sequences(f16)
    steps, cur = [], []
    for n, s, e in sorted(rows, key=lambda r: r[1]):
        cur.append((P.short(n), e - s))
        if cur[-1][0] == "sgd_flat_kernel":
            steps.append(cur)
            cur = []
    acc, used = {}, 0
    for st in steps:
        per = {}
        for n, d in st:
            if n in seq:
                per.setdefault(n, []).append(d)
        if any(len(per.get(n, [])) != len(q) for n, q in seq.items()):
            continue
        used += 1
        for n, q in seq.items():
            for d, (k, first) in zip(per[n], q):
                a = acc.setdefault(k, [n, 0, 0])
                if first:
                    a[0] = n
                a[1] += int(first)
                a[2] += d
    out = ["", "By timing class (%d of %d captured steps matched the program's launch sequence; a call's launches are "
           "summed):" % (used, len(steps)), "", "| class | kernel | what | calls / step | avg ms per call |", "|---|---|---|---|---|"]
    for k in sorted(acc):
        n, calls, tot = acc[k]
        out.append("| %d | %s | %s | %.1f | %.4f |" % (k, n, PR.KLASS.get(k, {}).get("name", "")[:70], calls / max(used, 1),
                                                        tot / max(calls, 1) / 1e6))
    return out


def main():
    args = [a for a in sys.argv[1:] if a != "--busy"]
    js = title = None
    if "--json" in args:
        i = args.index("--json"); js = args[i + 1]; del args[i:i + 2]
    if "--title" in args:
        i = args.index("--title"); title = args[i + 1]; del args[i:i + 2]
    tail = None
    if "--tail-fraction" in args:   # only kernels in the last fraction of the timeline
        i = args.index("--tail-fraction"); tail = float(args[i + 1]); del args[i:i + 2]
    marker, last = None, 0
    if "--marker" in args:          # keep only the last K periods of a once-per-step kernel
        i = args.index("--marker"); marker = args[i + 1]; del args[i:i + 2]
        i = args.index("--last"); last = int(args[i + 1]); del args[i:i + 2]
    by_class = None
    if "--classes" in args:         # heads | heads-f16: split the rows of kernels that serve several timing classes
        i = args.index("--classes"); by_class = args[i + 1]; del args[i:i + 2]
    db = sqlite3.connect(args[0])
    cur = db.cursor()
    rows = cur.execute("select name, start, end from kernels").fetchall()
    if tail and rows:
        t0, t1 = min(r[1] for r in rows), max(r[2] for r in rows)
        cut = t1 - (t1 - t0) * tail
        rows = [r for r in rows if r[1] >= cut]
    if marker and rows:
        # steady state only: from the end of the (K+1)-th last marker dispatch to the
        # end of the last one = exactly K step periods (library warm-up / MIOpen
        # find-phase candidates of the first steps are excluded)
        ends = sorted(r[2] for r in rows if marker in r[0])
        if len(ends) > last:
            lo, hi = ends[-last - 1], ends[-1]
            rows = [r for r in rows if r[1] >= lo and r[2] <= hi]
    busy_note = None
    if rows:
        # GPU occupancy of the window: union of the kernel intervals against its span,
        # and the kernels that precede / follow the longest idle gaps
        iv = sorted((r[1], r[2], r[0]) for r in rows)
        span = max(r[1] for r in iv) - iv[0][0]
        busy, cur_end, gaps, prev_name = 0, iv[0][0], [], ""
        for st, en, nm in iv:
            if st > cur_end:
                gaps.append((st - cur_end, prev_name, nm))
            busy += max(0, en - max(st, cur_end))
            if en > cur_end:
                cur_end, prev_name = en, nm
        gaps.sort(reverse=True)
        tot_gap = sum(g[0] for g in gaps)
        busy_note = ["", "GPU busy (union of kernel intervals) %.3f ms of %.3f ms window = %.1f %%; "
                     "%d idle gaps, %.3f ms total, %d of them > 20 us (%.3f ms)" % (
                         busy / 1e6, span / 1e6, 100.0 * busy / max(span, 1), len(gaps), tot_gap / 1e6,
                         sum(1 for g in gaps if g[0] > 20000), sum(g[0] for g in gaps if g[0] > 20000) / 1e6),
                     "", "| longest idle gaps (us) | after | before |", "|---|---|---|"]
        for g in gaps[:12]:
            busy_note.append("| %.1f | %s | %s |" % (g[0] / 1e3, short(g[1])[:50], short(g[2])[:50]))
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    lines = []
    if title:
        lines += ["# " + title, ""]
    lines += ["| kernel | calls | total ms | avg us | min us | max us | % |",
              "|---|---|---|---|---|---|---|"]
    out = {"kernels": {}, "counters": {}}
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if a[1] / total < 0.0005 and len(lines) > 30:
            continue
        lines.append("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (
            n, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / total))
        out["kernels"][n] = {"calls": a[0], "avg_us": a[1] / a[0] / 1e3}
    if busy_note and "--busy" in sys.argv:
        lines += busy_note
    if by_class:
        lines += class_table(rows, by_class == "heads-f16")
    try:
        q = ("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) "
             "from counters_collection group by 1, 2")
        crow = cur.execute(q).fetchall()
        if crow:
            lines += ["", "| kernel | counter | per-dispatch avg | dispatches |", "|---|---|---|---|"]
            for k, c, v, n in sorted(crow, key=lambda r: (short(r[0]), r[1])):
                k = short(k)
                if k.startswith("at::") or k.startswith("rocprim") or k.startswith("__amd"):
                    continue
                lines.append("| %s | %s | %.6g | %d |" % (k, c, v / max(n, 1), n))
                out["counters"].setdefault(k, {})[c] = v / max(n, 1)
    except sqlite3.Error as e:
        lines.append("\n(no counters: %s)" % e)
    text = "\n".join(lines) + "\n"
    if len(args) > 1:
        open(args[1], "w").write(text)
    if js:
        json.dump(out, open(js, "w"), indent=1, sort_keys=True)
    print(text)


if __name__ == "__main__":
    main()
