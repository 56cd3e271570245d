#!/bin/bash
# diagnostic: L2 hit rate and fabric line fills per kernel NAME over a whole bench run
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
name=$1; shift
rm -rf /tmp/prof_$name
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d /tmp/prof_$name -o t -- python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > /tmp/$name.log 2>&1
python - <<PY
import sqlite3, re, glob
db = sqlite3.connect(glob.glob("/tmp/prof_$name/*.db")[0])
rows = db.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by 1, 2").fetchall()
agg = {}
for k, c, v, n in rows:
    k = re.sub(r"\(anonymous namespace\)::", "", k); k = re.sub(r"^void ", "", k); k = re.sub(r"\(.*", "", k)[:60]
    agg.setdefault(k, {})[c] = v; agg[k]["n"] = n
print("%-60s %6s %8s %10s %10s" % ("kernel", "calls", "hit %", "rd GB", "wr GB"))
for k, e in sorted(agg.items(), key=lambda kv: -kv[1].get("TCC_EA0_RDREQ_sum", 0)):
    h, m = e.get("TCC_HIT_sum", 0), e.get("TCC_MISS_sum", 0)
    if e.get("TCC_EA0_RDREQ_sum", 0) * 128 < 5e7: continue
    print("%-60s %6d %8.1f %10.2f %10.2f" % (k, e["n"], 100 * h / max(h + m, 1), e.get("TCC_EA0_RDREQ_sum", 0) * 128 / 1e9, e.get("TCC_EA0_WRREQ_sum", 0) * 64 / 1e9))
PY
