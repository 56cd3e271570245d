#!/bin/bash
# schedule switches under the F(2x4) default: teacher side stream (SSAD_NATIVE_TWO_STREAMS), filter gradients on
# auxiliary streams (SSAD_OVERLAP_WGRAD), teacher ahead (SSAD_TEACHER_AHEAD)
run() { python bench.py --no-also --no-cpu-baseline --profile-steps 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])"; }
for rep in 1 2; do
echo "default:            $(run)"
echo "two_streams=0:      $(SSAD_NATIVE_TWO_STREAMS=0 run)"
echo "overlap_wgrad=0:    $(SSAD_OVERLAP_WGRAD=0 run)"
echo "both 0 (serial):    $(SSAD_NATIVE_TWO_STREAMS=0 SSAD_OVERLAP_WGRAD=0 run)"
echo "teacher_ahead=0:    $(SSAD_TEACHER_AHEAD=0 run)"
done
