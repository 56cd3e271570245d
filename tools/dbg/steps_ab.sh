for a in "--steps 10 --warmup 3" "--steps 20 --warmup 5" "--steps 40 --warmup 5" "--steps 20 --warmup 5"; do echo -n "$a: "; python bench.py --no-cpu-baseline --profile-steps 0 $a 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], 'host_in_step', d.get('host_in_step_ms_per_step'))"; done
echo -n "driver cmd: "; python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], 'host_in_step', d.get('host_in_step_ms_per_step'))"
