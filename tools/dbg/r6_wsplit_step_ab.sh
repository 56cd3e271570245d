run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-also --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
run "SSAD_SPLIT_CONV=31"
run "SSAD_SPLIT_CONV=63"
run "SSAD_SPLIT_CONV=127"
run "SSAD_SPLIT_CONV=95"
run "SSAD_SPLIT_CONV=31"
run "SSAD_SPLIT_CONV=63"
run "SSAD_SPLIT_CONV=127"
