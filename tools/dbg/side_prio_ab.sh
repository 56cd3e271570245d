python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
for rep in 1 2; do for v in 0 1 -1; do echo -n "fp32 side priority=$v: "; SSAD_SIDE_PRIORITY=$v python bench.py --no-cpu-baseline --profile-steps 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])"; done; done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl && rocprofv3 --kernel-trace -d /tmp/tl -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python tools/step_timeline.py $(ls /tmp/tl/*.db | head -1) 2.0 | head -7
