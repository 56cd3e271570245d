#!/bin/bash
# GPU box: A/B of compile-time switches of conv3x3_winograd.hip on the tower layer (tools/kbench.py); each argument
# is one set of extra compiler flags ("-" = none).  Throw-away rebuilds inside the box's scratch copy.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/semi-supervised-adaptive-distillation_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -fvisibility=hidden -Wno-unused-function"
for ex in "$@"; do
  [ "$ex" = "-" ] && ex=""
  /opt/rocm/bin/hipcc $FLAGS $ex -c kernels/conv3x3_winograd.hip -o build/kernels/conv3x3_winograd.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libcaffe2_detectron_ops_hip.so $(find build -name '*.o') || exit 1
  echo "== flags: $ex"
  (cd $R && timeout 300 python tools/kbench.py 2>&1 | grep -E "WINO (fwd|dgrad)")
done
