for rep in 1 2; do for v in 0 -1; do echo -n "fp32 main priority=$v: "; SSAD_MAIN_PRIORITY=$v python bench.py --no-cpu-baseline --profile-steps 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['roofline']['frac'])"; done; done
for v in 0 -1; do echo -n "cfg5 main priority=$v: "; SSAD_MAIN_PRIORITY=$v python bench.py --no-cpu-baseline --profile-steps 0 --steps 20 --warmup 5 --student r101 --teacher x101-64x4d --px 500 --precision f16 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])"; done
