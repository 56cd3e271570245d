#!/bin/bash
# oracle/build_ref_conv.sh -- builds oracle/_ref/libref_conv.so: the REFERENCE's own CPU
# convolution operators, compiled from the sources where they lie under /root/reference.
#
# TEST INFRASTRUCTURE ONLY (container with /root/reference; `make -C oracle refconv`).
# Nothing of the reference is copied into this repository and nothing is stood in for:
#   1. protoc 3.4.x and libprotobuf are compiled with g++ from the reference's vendored
#      caffe2/third_party/protobuf (source lists read from its own cmake/*.cmake files; the
#      one generated source of protoc, js/well_known_types_embed.cc, is produced by the
#      reference's own js/embed.cc tool exactly as its src/Makefile.am:518-527 does);
#   2. that protoc generates caffe2.pb.{h,cc} / caffe2_legacy.pb.{h,cc} from
#      caffe2/caffe2/proto/*.proto;
#   3. g++ compiles caffe2/caffe2/core/*.cc (CPU part), utils/{math_cpu,proto_utils,...}.cc
#      and operators/conv_{op,op_shared,gradient_op,op_eigen}.cc against the vendored Eigen,
#      with the reference's own build options passed as -D flags (USE_EIGEN_FOR_BLAS = its
#      cmake option CAFFE2_USE_EIGEN_FOR_BLAS; version numbers from caffe2/VERSION_NUMBER;
#      caffe2/core/macros.h is the placeholder the reference ships);
#   4. oracle/ref_conv_driver.cc (ours: feeds blobs, creates the operators through the
#      reference's registry from an OperatorDef, fetches the outputs) is linked with them.
# Intermediates live in a mktemp directory; the only output is oracle/_ref/libref_conv.so
# (git-ignored; it travels to the GPU box with the snapshot).  ~1.5 CPU-minutes x 8 cores.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${REFROOT:-/root/reference}"
C2="$REF/caffe2"
PB="$C2/third_party/protobuf"
OUT="$HERE/_ref/libref_conv.so"
JOBS="${JOBS:-$(nproc)}"
[ -d "$C2/caffe2/core" ] || { echo "no reference tree at $REF" >&2; exit 1; }
mkdir -p "$HERE/_ref"
T="$(mktemp -d /tmp/ref_conv.XXXXXX)"
trap 'rm -rf "$T"' EXIT
cd "$T"

grep -ho 'src/google/protobuf/[a-zA-Z0-9_/.]*\.cc' "$PB/cmake/libprotobuf-lite.cmake" "$PB/cmake/libprotobuf.cmake" \
  | sort -u | grep -v 'msvc\|io_win32' > pb_files.txt
grep -ho 'src/google/protobuf/[a-zA-Z0-9_/.]*\.cc' "$PB/cmake/libprotoc.cmake" | sort -u | grep -v 'js/embed.cc' > pc_files.txt
VER=$(cat "$C2/VERSION_NUMBER")
IFS=. read -r VMAJ VMIN VPAT <<< "$VER"

cat > build.mk <<EOF2
C2 := $C2
PB := $PB
PBF := \$(shell cat pb_files.txt)
PCF := \$(shell cat pc_files.txt)
OBJ_PB := \$(patsubst %.cc,obj/%.o,\$(PBF))
OBJ_PC := \$(patsubst %.cc,obj/%.o,\$(PCF))
PBFLAGS := -O1 -std=c++11 -fPIC -w -DHAVE_PTHREAD -I\$(PB)/src
CORE := allocator blob_serialization blob_stats common context db event flags init init_intrinsics_check \\
        logging module net net_simple net_dag net_dag_utils net_async_base net_async_polling net_async_scheduling \\
        net_simple_async operator operator_schema stats tensor typeid types workspace plan_executor graph memonger \\
        transform qtensor qtensor_serialization
UTILS := math_cpu proto_utils string_utils cpuid signal_handler threadpool/ThreadPool threadpool/pthreadpool \\
         threadpool/pthreadpool_impl
OPS := conv_op conv_op_shared conv_gradient_op conv_op_eigen
SRCS := \$(addprefix caffe2/core/,\$(addsuffix .cc,\$(CORE))) \$(addprefix caffe2/utils/,\$(addsuffix .cc,\$(UTILS))) \\
        \$(addprefix caffe2/operators/,\$(addsuffix .cc,\$(OPS)))
OBJ_C2 := \$(patsubst %.cc,c2obj/%.o,\$(SRCS)) c2obj/gen/caffe2/proto/caffe2.pb.o c2obj/gen/caffe2/proto/caffe2_legacy.pb.o
C2FLAGS := -O2 -std=c++11 -fPIC -w -fno-fast-math -ffp-contract=off -DCAFFE2_USE_EIGEN_FOR_BLAS -DEIGEN_MPL2_ONLY \\
        -DCAFFE2_VERSION_MAJOR=$VMAJ -DCAFFE2_VERSION_MINOR=$VMIN -DCAFFE2_VERSION_PATCH=$VPAT \\
        -DCAFFE2_GIT_VERSION=\\"none\\" -DHAVE_PTHREAD -Igen -I\$(C2) -I\$(PB)/src -I\$(C2)/third_party/eigen

protoc: \$(OBJ_PB) \$(OBJ_PC) obj/main.o obj/src/google/protobuf/compiler/js/well_known_types_embed.o
	g++ -o \$@ \$^ -lpthread
obj/%.o: \$(PB)/%.cc
	@mkdir -p \$(dir \$@)
	g++ \$(PBFLAGS) -c \$< -o \$@
obj/main.o: \$(PB)/src/google/protobuf/compiler/main.cc
	@mkdir -p obj
	g++ \$(PBFLAGS) -c \$< -o \$@
obj/src/google/protobuf/compiler/js/well_known_types_embed.o: gen/google/protobuf/compiler/js/well_known_types_embed.cc
	@mkdir -p \$(dir \$@)
	g++ \$(PBFLAGS) -c \$< -o \$@
libprotobuf.a: \$(OBJ_PB)
	ar rcs \$@ \$^

c2: \$(OBJ_C2) c2obj/driver.o
c2obj/%.o: \$(C2)/%.cc
	@mkdir -p \$(dir \$@)
	g++ \$(C2FLAGS) -c \$< -o \$@
c2obj/gen/%.o: gen/%.cc
	@mkdir -p \$(dir \$@)
	g++ \$(C2FLAGS) -c \$< -o \$@
c2obj/driver.o: $HERE/ref_conv_driver.cc
	@mkdir -p c2obj
	g++ \$(C2FLAGS) -c \$< -o \$@
libref_conv.so: c2 libprotobuf.a
	g++ -shared -o \$@ -Wl,--no-undefined \$(OBJ_C2) c2obj/driver.o libprotobuf.a -lpthread
EOF2

# the one generated source of protoc, made by the reference's own tool (src/Makefile.am:518-527)
g++ -o js_embed "$PB/src/google/protobuf/compiler/js/embed.cc"
mkdir -p gen/google/protobuf/compiler/js
( cd "$PB/src" && "$T/js_embed" google/protobuf/compiler/js/well_known_types/any.js \
    google/protobuf/compiler/js/well_known_types/struct.js \
    google/protobuf/compiler/js/well_known_types/timestamp.js ) > gen/google/protobuf/compiler/js/well_known_types_embed.cc

make -s -f build.mk -j"$JOBS" protoc libprotobuf.a
./protoc -I"$C2" --cpp_out=gen "$C2/caffe2/proto/caffe2.proto" "$C2/caffe2/proto/caffe2_legacy.proto"
make -s -f build.mk -j"$JOBS" libref_conv.so
cp libref_conv.so "$OUT"
echo "built $OUT"
