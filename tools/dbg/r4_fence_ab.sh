mkdir -p gpurun_out
for f in 0 1 2 0 1; do
  SSAD_TICKET_FENCES=$f python bench.py --workload heads --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 > gpurun_out/fence_$f.json 2>/dev/null
  python - <<P
import json
l=[x for x in open('gpurun_out/fence_$f.json') if x.startswith('{"metric')][-1]
p=json.loads(l)
print('fences=$f', 'ms/step', p['ms_per_step'], 'loss', p['roofline_loss']['avg_launch_ms'], p['roofline_loss']['frac'], 'powsum', p['roofline_pow_sum']['avg_launch_ms'], p['roofline_pow_sum']['frac'])
P
done
python -m pytest tests/test_gpu_backbone.py tests/test_gpu_weights.py tests/test_gpu_full_size.py tests/test_gpu_native_model.py tests/test_gpu_rccl.py tests/test_gpu_kernels.py tests/test_gpu_f16.py tests/test_gpu_operators.py -m gpu -q 2>&1 | tail -40
