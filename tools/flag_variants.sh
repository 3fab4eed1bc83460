#!/bin/bash
# Builds library variants that differ only in the compiler flags (or -D switches) of the two hot kernels -- the recording
# hierarchical forward (hier_r8.o) and the replay backward (stp_render_replay.o) -- and links each with the other objects of
# the product build:   tools/flag_variants.sh <name> "<flags for the forward>" "<flags for the replay>" [...]
# ("-" = leave that kernel as the product builds it).  Output: gpurun_ab/libstp_<name>.so (git-ignored, travels with gpurun).
# Used for the round-3 scheduler-flag sweep; A/B on one box with tools/abn.sh, bitwise check with
# tools/compare_builds.py.
set -e
cd "$(dirname "$0")/../stopthepop-rasterization_amd/csrc"
HIPCC=/opt/rocm/bin/hipcc
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-unused-variable -I/opt/rocm/include"
mkdir -p ../../gpurun_ab
[ -f build/stp_api.o ] || make -j8 > /dev/null
while [ $# -ge 3 ]; do
  name=$1; ffl=$2; rfl=$3; shift 3
  d=/tmp/flagvar_$name; mkdir -p $d
  objs=""
  for o in build/*.o; do
    b=$(basename $o)
    if [ "$b" = hier_r8.o ] && [ "$ffl" != "-" ]; then objs="$objs $d/hier_r8.o"
    elif [ "$b" = stp_render_replay.o ] && [ "$rfl" != "-" ]; then objs="$objs $d/stp_render_replay.o"
    else objs="$objs $o"; fi
  done
  ( [ "$ffl" = "-" ] || $HIPCC $BASE -fno-slp-vectorize -DSTP_INST_MID=8 -DSTP_INST_MODE=2 $ffl -c stp_render_hier_inst.hip -o $d/hier_r8.o ) &
  ( [ "$rfl" = "-" ] || $HIPCC $BASE $rfl -c stp_render_replay.hip -o $d/stp_render_replay.o ) &
  wait
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_ab/libstp_$name.so $objs
  echo "built gpurun_ab/libstp_$name.so"
done
