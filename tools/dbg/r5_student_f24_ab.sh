#!/bin/bash
# in-step A/B of SSAD_STUDENT_F24 (bit mask: 1 subnet data gradients, 2 cls_pred forward, 4 tower forward, 8 backbone)
run() { python bench.py --no-also --no-cpu-baseline --profile-steps 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['distill_loss'][:2])"; }
for rep in 1 2; do for m in ${MODES:-0 8 9 11 15}; do echo "student F24 mode $m: $(SSAD_STUDENT_F24=$m run)"; done; done
for m in ${MODES:-0 1 3 7}; do echo "heads, student F24 mode $m: $(SSAD_STUDENT_F24=$m run --workload heads)"; done
