#!/bin/bash
# GPU box: A/B of compile-time switches of ONE kernel source on tools/kbench.py lines.
#   tools/dbg/kernel_ab.sh conv3x3_f16 "F16 wgrad" "--what f16" "-" "-DF16_ABLATE=8"
# arguments: source (kernels/<name>.hip), grep pattern over kbench's output, kbench arguments, then one set of
# extra compiler flags per run ("-" = none).  Throw-away rebuilds inside the box's scratch copy.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
SRC=$1; PAT=$2; KARGS=$3; shift 3
cd $R/semi-supervised-adaptive-distillation_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -fvisibility=hidden -Wno-unused-function"
for ex in "$@"; do
  [ "$ex" = "-" ] && ex=""
  /opt/rocm/bin/hipcc $FLAGS $ex -c kernels/$SRC.hip -o build/kernels/$SRC.o || exit 1
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libcaffe2_detectron_ops_hip.so $(find build -name '*.o') || exit 1
  echo "== $SRC flags: $ex"
  (cd $R && timeout 300 python tools/kbench.py $KARGS 2>&1 | grep -E "$PAT")
done
