#!/bin/bash
# GPU box: per-unit cycle stamps of wino_wgrad_kernel under the WGRAD_ABLATE debug switches
# (throw-away rebuilds of the in-tree library inside the box's scratch copy).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/semi-supervised-adaptive-distillation_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -fvisibility=hidden -Wno-unused-function"
for ab in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DWGRAD_TIMELINE -DWGRAD_ABLATE=$ab -c kernels/conv3x3_wgrad_winograd.hip -o build/kernels/conv3x3_wgrad_winograd.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libcaffe2_detectron_ops_hip.so $(find build -name '*.o') || exit 1
  echo "== WGRAD_ABLATE=$ab"
  (cd $R && timeout 300 python tools/wgrad_timeline.py | grep -E "SUMMARY|FINE")
done
