#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/semi-supervised-adaptive-distillation_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -fvisibility=hidden -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS ${EXTRA:-} -DSPLIT_TIMELINE -c kernels/conv3x3_split.hip -o build/kernels/conv3x3_split.o || exit 1
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libcaffe2_detectron_ops_hip.so $(find build -name '*.o') || exit 1
cd $R && python tools/dbg/r6_split_timeline.py 2>&1 | grep -v amdgpu
