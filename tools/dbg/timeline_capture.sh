# step timelines (tools/step_timeline.py) of the default bench with and without the teacher running ahead
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/timeline
for v in 1 0; do
  rm -rf /tmp/tl$v
  (cd /tmp && SSAD_TEACHER_AHEAD=$v rocprofv3 --kernel-trace -d /tmp/tl$v -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 > /tmp/tl$v.log 2>&1)
  grep -o '"ms_per_step": [0-9.]*' /tmp/tl$v.log | head -1
  python $R/tools/step_timeline.py $(ls /tmp/tl$v/*.db | head -1) 2.0 cls_losses_fused_kernel 6 > $R/gpurun_out/timeline/ahead$v.txt
  head -7 $R/gpurun_out/timeline/ahead$v.txt
done
