#!/bin/bash
# GPU box, round 6: (1) what an F(3x3, 2x4) filter-gradient engine could gain (WGRAD_ABLATE 16: its MFMA and VALU work per
# pixel on today's skeleton -- a lower bound on its time); (2) what A-operand reuse 2 could gain in wino24_conv_kernel
# (W24_ABLATE 512: half the filter-operand traffic, everything else unchanged -- the best case of that design).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/semi-supervised-adaptive-distillation_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -fvisibility=hidden -Wno-unused-function"
link() { /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libcaffe2_detectron_ops_hip.so $(find build -name '*.o') || exit 1; }
for ab in 0 16 2 18; do
  /opt/rocm/bin/hipcc $FLAGS -DWGRAD_ABLATE=$ab -c kernels/conv3x3_wgrad_winograd.hip -o build/kernels/conv3x3_wgrad_winograd.o || exit 1
  link
  echo "== WGRAD_ABLATE=$ab"
  (cd $R && timeout 300 python tools/dbg/r6_wgrad_time.py 2>&1 | grep -v amdgpu.ids)
done
/opt/rocm/bin/hipcc $FLAGS -c kernels/conv3x3_wgrad_winograd.hip -o build/kernels/conv3x3_wgrad_winograd.o || exit 1
for ab in 0 512 128 4; do
  /opt/rocm/bin/hipcc $FLAGS -DW24_ABLATE=$ab -c kernels/conv3x3_winograd24.hip -o build/kernels/conv3x3_winograd24.o || exit 1
  link
  echo "== W24_ABLATE=$ab"
  (cd $R && timeout 300 python tools/dbg/w24_time.py 2>&1 | grep -v amdgpu.ids | head -2)
done
