run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --profile-steps 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
run X=1
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=5
run X=2
