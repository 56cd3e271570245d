#!/bin/bash
# GPU box: upper bound of what chaining |max| words from producer to consumer could save -- a throw-away build in which
# the split engines' |max| passes over ACTIVATIONS are not launched (all scales 1: results differ, timing only).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/semi-supervised-adaptive-distillation_amd/csrc
for f in conv3x3_split conv3x3_wgrad_split gemm_split; do
  sed -e 's/hipLaunchKernelGGL(split_absmax_kernel,/if (0) hipLaunchKernelGGL(split_absmax_kernel,/' \
      -e 's/hipLaunchKernelGGL(wsplit_absmax_kernel,/if (0) hipLaunchKernelGGL(wsplit_absmax_kernel,/' kernels/$f.hip > /tmp/$f.hip
done
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -Ikernels -fvisibility=hidden -Wno-unused-function"
run() { echo -n "$1: "; python $R/bench.py --no-cpu-baseline --no-also --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
cd $R; run "with |max| passes"; run "with |max| passes"
cd $R/semi-supervised-adaptive-distillation_amd/csrc
for f in conv3x3_split conv3x3_wgrad_split gemm_split; do
  /opt/rocm/bin/hipcc $FLAGS -x hip -c /tmp/$f.hip -o build/kernels/$f.o || exit 1
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libcaffe2_detectron_ops_hip.so $(find build -name '*.o') || exit 1
cd $R; run "without"; run "without"
