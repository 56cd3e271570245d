#!/bin/bash
# HBM bytes of a WHOLE bench step by kernel name: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
# (MI355X_MICROARCH.md: counters in their own runs, with --kernel-trace only) over `bench.py <args> --steps 3
# --warmup 1`; read side x 2 (the gfx950 correction calibrated in profiles/r02_pmc_fetch_calib.md).  Prints a
# table and the step's total against its wall time.     tools/hbm_by_kernel.sh NAME [bench args...]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
name=$1; shift
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_${name}_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/prof_${name}_$c -o t -- python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-also --profile-steps 0 > /tmp/${name}_$c.log 2>&1
done
ms=$(python bench.py "$@" --steps 10 --warmup 3 --no-cpu-baseline --no-also --profile-steps 0 2>/dev/null | python -c "
import sys,json
print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])")
python - <<PY
import sqlite3, re, glob
def load(c):
    db = sqlite3.connect(glob.glob("/tmp/prof_${name}_%s/*.db" % c)[0])
    return db.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection where counter_name = ? group by 1", (c,)).fetchall()
agg = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, v, n in load(c):
        k = re.sub(r"\(anonymous namespace\)::", "", k); k = re.sub(r"^void ", "", k); k = re.sub(r"\(.*", "", k)[:56]
        e = agg.setdefault(k, {"n": 0}); e[c] = v; e["n"] = max(e["n"], n)
steps = 7.0            # bench.py ran 1 warm-up + 3 timed + 3 enqueue-timing steps under the tracer
tot_r = tot_w = 0.0
print("| kernel | launches / step | read GB / step | written GB / step |"); print("|---|---|---|---|")
for k, e in sorted(agg.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0))):
    r, w = 2 * e.get("FETCH_SIZE", 0) * 1024 / steps / 1e9, e.get("WRITE_SIZE", 0) * 1024 / steps / 1e9
    tot_r += r; tot_w += w
    if r + w >= 0.05:
        print("| %s | %.1f | %.2f | %.2f |" % (k, e["n"] / steps, r, w))
ms = float("$ms")
print("\nstep total: %.1f GB read + %.1f GB written = %.1f GB in %.2f ms = %.2f TB/s (%.0f %% of 8 TB/s)" % (
    tot_r, tot_w, tot_r + tot_w, ms, (tot_r + tot_w) / ms, 100 * (tot_r + tot_w) / ms / 8.0))
PY
