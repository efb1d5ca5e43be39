#!/bin/bash
# AddressSanitizer pass over the HOST code (CPU only; GPU ASan is not available on the pool): the .cpp sources of
# the product library (C ABI, host builders, relayout) and the oracle are rebuilt with -fsanitize=address and the
# CPU test suite runs against them.  Usage: tools/asan_cpu.sh [pytest args]; outputs under /tmp/nnbvh_asan.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/nnbvh_asan
LLVM=/opt/rocm/lib/llvm
RT=$(ls $LLVM/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
mkdir -p $OUT
python -c "from nn_bvh_amd import build; build.build()" >/dev/null
SAN="-fsanitize=address -shared-libasan -fno-omit-frame-pointer -g"
objs=""
for f in bvh_capi bvh_layout bvh_build kd_build; do
  /opt/rocm/bin/hipcc -O1 $SAN -ffp-contract=off -fPIC -std=c++17 -I$ROOT/include -c $ROOT/nn_bvh_amd/csrc/$f.cpp -o $OUT/$f.o
  objs="$objs $OUT/$f.o"
done
for o in $ROOT/nn_bvh_amd/_obj/product/*.o; do
  case $(basename $o .o) in bvh_capi|bvh_layout|bvh_build|kd_build) ;; *) objs="$objs $o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $SAN -o $ROOT/nn_bvh_amd/libnnbvh_hip_asan.so $objs
$LLVM/bin/clang -O1 $SAN -std=gnu11 -fPIC -ffp-contract=off -mfma -shared -o $OUT/libnnbvh_oracle_asan.so $ROOT/oracle/nnbvh_oracle.c -lm -lpthread
cd $ROOT
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:symbolize=1 ASAN_SYMBOLIZER_PATH=$LLVM/bin/llvm-symbolizer \
  NNBVH_LIB=libnnbvh_hip_asan.so NNBVH_ORACLE_LIB=$OUT/libnnbvh_oracle_asan.so \
  python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@"
rm -f $ROOT/nn_bvh_amd/libnnbvh_hip_asan.so
