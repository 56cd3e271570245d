#!/bin/bash
# GPU box: gemm_fly_kernel under the GFLY_ABLATE debug switches (throw-away rebuilds inside the box's scratch copy)
#   bash tools/dbg/r6_gfly_ablate.sh 0 1 2 3 7 11
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/semi-supervised-adaptive-distillation_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -fvisibility=hidden -Wno-unused-function"
for ab in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS ${EXTRA:-} -DGFLY_ABLATE=$ab -c kernels/gemm_split.hip -o build/kernels/gemm_split.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libcaffe2_detectron_ops_hip.so $(find build -name '*.o') || exit 1
  echo "== GFLY_ABLATE=$ab ${EXTRA:-}"
  (cd $R && timeout 300 python tools/gemm_split_bench.py 2>&1 | grep -E "res4|res5 2048" | cut -c1-120)
done
