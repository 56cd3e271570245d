# same-call A/B of SSAD_SPLIT_CONV values over the default bench
run() { echo -n "$*: "; env "$@" python bench.py --no-cpu-baseline --no-also --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
for v in ${AB_VALUES:-31 127 255 127 255}; do run SSAD_SPLIT_CONV=$v; done
