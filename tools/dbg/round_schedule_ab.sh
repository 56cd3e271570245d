# the round-start schedule (teacher after the packs, after the previous update, default-priority stream) against the
# final default, same call, two repeats each; then a second config-5 pair for the high-priority stream
run() { echo -n "$1 | $2: "; env $2 python bench.py --no-cpu-baseline --profile-steps 0 --steps 20 --warmup 5 $3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
OLD="SSAD_TEACHER_AHEAD=0 SSAD_MAIN_PRIORITY=0"
for rep in 1 2; do run "cfg3 final" "X=1" ""; run "cfg3 round-start" "$OLD" ""; done
C5="--student r101 --teacher x101-64x4d --px 500 --precision f16"
for rep in 1 2; do run "cfg5 prio -1" "SSAD_MAIN_PRIORITY=-1" "$C5"; run "cfg5 prio 0" "SSAD_MAIN_PRIORITY=0" "$C5"; done
