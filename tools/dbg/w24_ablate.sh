#!/bin/bash
# GPU box: wino24_conv_kernel under the W24_ABLATE debug switches (throw-away rebuilds of the in-tree library inside
# the box's scratch copy): what a chunk's time is made of.   bash tools/dbg/w24_ablate.sh 0 1 2 4 8 16 32 ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/semi-supervised-adaptive-distillation_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -fvisibility=hidden -Wno-unused-function"
for ab in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS ${EXTRA:-} -DW24_ABLATE=$ab -c kernels/conv3x3_winograd24.hip -o build/kernels/conv3x3_winograd24.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libcaffe2_detectron_ops_hip.so $(find build -name '*.o') || exit 1
  echo "== W24_ABLATE=$ab ${EXTRA:-}"
  (cd $R && timeout 300 python tools/dbg/w24_time.py 2>&1 | grep -v amdgpu.ids)
done
