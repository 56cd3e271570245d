#!/bin/bash
# in-step A/B of the teacher's F(2x4, 3x3) engine: 0 = off, 1 = cls_pred + backbone, 2 = + towers
run() { python bench.py --no-also --no-cpu-baseline --profile-steps 0 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['distill_loss'][:2])"; }
for rep in 1 2; do for m in 0 1 2; do echo "F24 mode $m: $(SSAD_TEACHER_F24=$m run)"; done; done
for m in 0 1 2; do echo "heads F24 mode $m: $(SSAD_TEACHER_F24=$m run --workload heads)"; done
