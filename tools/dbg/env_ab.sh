# bench.py under ROCclr knobs that bound how far the launching thread may run ahead of the GPU
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], 'host_in_step', d.get('host_in_step_ms_per_step'))"; }
run "X=1"
run "ROC_SIGNAL_POOL_SIZE=4096"
run "ROC_AQL_QUEUE_SIZE=65536"
run "HIP_FORCE_DEV_KERNARG=0"
run "GPU_MAX_HW_QUEUES=8"
run "GPU_MAX_HW_QUEUES=4"
run "ROC_CPU_WAIT_FOR_SIGNAL=0"
run "SSAD_TEACHER_FIRST=0"
run "X=2"
