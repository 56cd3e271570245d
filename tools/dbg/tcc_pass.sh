#!/bin/bash
# one-off diagnostic: L2 (TCC) counters of the subnets' launches by timing class
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/profiles_r03
mkdir -p $O
run() { name=$1; shift; rm -rf /tmp/prof_$name; rocprofv3 --kernel-trace --pmc "$@" -d /tmp/prof_$name -o t -- python bench.py --workload heads --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > /tmp/$name.log 2>&1; tail -1 /tmp/$name.log | cut -c1-200; }
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
python tools/pmc_by_class.py --out $O/tcc_classes.json --md $O/tcc_classes.md tcc1=$(ls /tmp/prof_tcc1/*.db | head -1) tcc2=$(ls /tmp/prof_tcc2/*.db | head -1) > /dev/null 2> $O/tcc.err
tail -3 $O/tcc.err
python - <<PY
import json
d=json.load(open("$O/tcc_classes.json"))
for k,e in sorted(d["classes"].items(), key=lambda kv:int(kv[0])):
    print(k, e["kernel"], {c: round(v/1e6,2) for c,v in e.items() if c.startswith("TCC")})
PY
