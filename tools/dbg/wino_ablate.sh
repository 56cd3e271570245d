#!/bin/bash
# GPU box: per-chunk cycle stamps of wino_conv_z_kernel under the WINO_ABLATE debug switches
# (throw-away rebuilds of the in-tree library inside the box's scratch copy).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/semi-supervised-adaptive-distillation_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -fvisibility=hidden -Wno-unused-function"
for ab in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DWINO_TIMELINE -DWINO_ABLATE=$ab -c kernels/conv3x3_winograd.hip -o build/kernels/conv3x3_winograd.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libcaffe2_detectron_ops_hip.so $(find build -name '*.o') || exit 1
  echo "== WINO_ABLATE=$ab"
  (cd $R && timeout 300 python tools/wino_timeline.py | grep -E "SUMMARY|WAVES")
done
