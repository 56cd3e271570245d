#!/bin/bash
# GPU box: config 5 (fp16 storage) with the filter-gradient REDUCE launches of conv3x3_f16.hip compiled out (wrong
# gradients, timing only): the most that folding those 111 launches per step into the last-arriving workgroup could save.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/semi-supervised-adaptive-distillation_amd/csrc
sed -e 's/    hipLaunchKernelGGL(f16_wgrad_reduce_pw_kernel,/    if (0) hipLaunchKernelGGL(f16_wgrad_reduce_pw_kernel,/' \
    -e 's/    hipLaunchKernelGGL(f16_wgrad_reduce_kernel,/    if (0) hipLaunchKernelGGL(f16_wgrad_reduce_kernel,/' kernels/conv3x3_f16.hip > /tmp/conv3x3_f16.hip
grep -c "if (0) hipLaunch" /tmp/conv3x3_f16.hip
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -Ikernels -fvisibility=hidden -Wno-unused-function"
run() { echo -n "$1: "; python $R/bench.py --student r101 --teacher x101-64x4d --px 500 --precision f16 --no-cpu-baseline --no-also --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
cd $R; run "as built"; run "as built"
cd $R/semi-supervised-adaptive-distillation_amd/csrc
/opt/rocm/bin/hipcc $FLAGS -x hip -c /tmp/conv3x3_f16.hip -o build/kernels/conv3x3_f16.o || exit 1
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libcaffe2_detectron_ops_hip.so $(find build -name '*.o') || exit 1
cd $R; run "no reduce launches"; run "no reduce launches"
