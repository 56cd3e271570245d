#!/bin/bash
# Round 4: the final schedule (high-priority main stream, teacher ahead on a side stream, two auxiliary
# streams) with and without collectives in flight (a forced one-rank RCCL communicator), for several
# stream -> hardware-queue mappings.  Two rounds, alternating, so that box drift shows.
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
mkdir -p gpurun_out
out=gpurun_out/r04_hw_queues.txt; : > $out
for round in 1 2; do
 for q in 4 6 7 8; do
  for force in 0 1; do
    ms=$(GPU_MAX_HW_QUEUES=$q SSAD_DP_FORCE=$force python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --profile-steps 0 2>/dev/null | python -c "
import sys,json
l=[x for x in sys.stdin if x.startswith('{')]
print(json.loads(l[-1])['ms_per_step'] if l else 'failed')")
    echo "round $round GPU_MAX_HW_QUEUES=$q collectives=$force ms_per_step=$ms" | tee -a $out
  done
 done
done
