#!/bin/bash
# in-step A/B of the Winograd split tail: 0 = off, 1 = launches without a full round only, 2 = every partial round
run() { python bench.py --no-also --no-cpu-baseline --profile-steps 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2; do for m in 0 1 2; do echo "overlapped mode $m: $(SSAD_WINO_SPLIT_TAIL=$m run)"; done; done
for m in 0 2; do echo "serial mode $m: $(SSAD_WINO_SPLIT_TAIL=$m SSAD_NATIVE_TWO_STREAMS=0 SSAD_OVERLAP_WGRAD=0 run)"; done
for m in 0 1 2; do echo "cfg2 mode $m: $(SSAD_WINO_SPLIT_TAIL=$m run --teacher none --batch-per-gpu 2)"; done
for m in 0 1 2; do echo "heads mode $m: $(SSAD_WINO_SPLIT_TAIL=$m run --workload heads)"; done
