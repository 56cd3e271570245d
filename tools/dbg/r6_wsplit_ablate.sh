#!/bin/bash
# GPU box: wsplit_kernel under the WSPLIT_ABLATE debug switches (throw-away rebuilds inside the box's scratch copy)
#   bash tools/dbg/r6_wsplit_ablate.sh 0 1 2 4 ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/semi-supervised-adaptive-distillation_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -fvisibility=hidden -Wno-unused-function"
for ab in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS ${EXTRA:-} -DWSPLIT_ABLATE=$ab -c kernels/conv3x3_wgrad_split.hip -o build/kernels/conv3x3_wgrad_split.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libcaffe2_detectron_ops_hip.so $(find build -name '*.o') || exit 1
  echo "== WSPLIT_ABLATE=$ab ${EXTRA:-}"
  (cd $R && WSPLIT_ONLY=1 timeout 300 python tools/dbg/r6_wsplit_time.py 2>&1 | grep -v amdgpu.ids)
done
