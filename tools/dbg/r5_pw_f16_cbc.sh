#!/bin/bash
# GPU box: the fp16 pointwise GEMM with 64-channel K chunks (PW_CBC=8) at ring depths that fit the LDS (VERDICT r4 item 4's
# prescription), throw-away rebuilds; isolated shapes of config 5 (tools/pw_f16_probe.py)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/semi-supervised-adaptive-distillation_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -fvisibility=hidden -Wno-unused-function"
for cfg in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS $cfg -c kernels/gemm_f16.hip -o build/kernels/gemm_f16.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libcaffe2_detectron_ops_hip.so $(find build -name '*.o') || exit 1
  echo "== $cfg"
  (cd $R && timeout 300 python tools/pw_f16_probe.py 2>&1 | grep -v amdgpu.ids)
done
