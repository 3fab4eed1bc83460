#!/bin/bash
# Library variants that differ in the -D switches of ONE object of the product build:
#   tools/obj_variant.sh <name> <source.hip> "<extra flags>" [<name> <source.hip> "<flags>" ...]
# (sources compiled with the Makefile's plain flags: everything except the hierarchical / k-buffer render kernels, for which
# tools/flag_variants.sh exists).  Output: gpurun_ab/libstp_<name>.so (git-ignored, travels with gpurun); A/B with tools/abn.sh.
set -e
cd "$(dirname "$0")/../stopthepop-rasterization_amd/csrc"
HIPCC=/opt/rocm/bin/hipcc
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-unused-variable -I/opt/rocm/include"
mkdir -p ../../gpurun_ab
make -j8 > /dev/null
while [ $# -ge 3 ]; do
  name=$1; src=$2; fl=$3; shift 3
  d=/tmp/objvar_$name; mkdir -p $d
  ob=$(basename $src .hip).o
  [ -f build/$ob ] || { echo "no object build/$ob in the product build"; exit 1; }
  $HIPCC $BASE $fl -c $src -o $d/$ob
  objs=""
  for o in build/*.o; do
    if [ "$(basename $o)" = "$ob" ]; then objs="$objs $d/$ob"; else objs="$objs $o"; fi
  done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_ab/libstp_$name.so $objs
  echo "built gpurun_ab/libstp_$name.so"
done
