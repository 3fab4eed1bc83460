#!/usr/bin/env bash
# oracle/ref_build/build_ref.sh -- compiles THE REFERENCE ITSELF (its own rasterizer sources, where they lie
# under /root/reference) for gfx950, into oracle/_ref/ (git-ignored; travels to the GPU box like our own .so).
#
# TEST INFRASTRUCTURE ONLY.  The result is the checker that pins oracle/stp_oracle.cpp (and, on the GPU box,
# the product's kernels) to the reference, and the "reference on MI355X" timing beside bench.py's numbers.
# Nothing under stopthepop-rasterization_amd/ loads it.
#
# Recipe (no reference source is copied into the repository; the translated files live in a mktemp
# directory that is deleted on exit):
#   1. /opt/rocm/bin/hipify-perl (the ROCm image's own CUDA->HIP source translator) on the 3 .cu files and
#      the 11 headers of cuda_rasterizer/.  The reference's build system (setup.py / CMake) is not run.
#   2. Four textual fix-ups of things hipify-perl leaves behind (none touches an expression or a statement
#      of the algorithm):
#        a. `#include ""`   (what hipify-perl makes of device_launch_parameters.h)            -> removed
#        b. `#include <cub/...>` lines it does not translate                                   -> removed
#           (hipcub/hipcub.hpp is already included, via hip_adapter.h)
#        c. kernel launches written `<< <...>> >` (a CUDA-compiler tolerance)                  -> `<<<...>>>`
#        d. launch configuration `{16, 4, 4}` as a braced list                                 -> `dim3(16, 4, 4)`
#   3. hipcc --offload-arch=gfx950 with oracle/ref_build/hip_adapter.h force-included (vector-type
#      tile.shfl, 32-bit-mask *_sync builtins, __trap: see that file) and glm from the reference's own
#      third_party/glm.
#   4. oracle/ref_build/ref_driver.cpp (ours: C ABI + memory movement, no rasterizer logic) linked in.
# Two variants:
#   libstp_ref.so        hipcc defaults (-ffp-contract=fast, like nvcc's --fmad=true): "as a user builds it";
#                        used for the reference-on-MI355X timing and tolerance-level parity.
#   libstp_ref_ieee.so   -ffp-contract=off: every a*b+c is two IEEE roundings, so all integer/index results
#                        and all +,-,*,/,sqrt arithmetic have ONE meaning that a CPU restatement can match
#                        bit for bit.  This is the variant the oracle is pinned against.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../_ref"
ref="${STP_REFERENCE_ROOT:-/root/reference}"
hipify="${HIPIFY_PERL:-/opt/rocm/bin/hipify-perl}"
hipcc="${HIPCC:-/opt/rocm/bin/hipcc}"

if [ ! -d "$ref/cuda_rasterizer" ]; then
    echo "build_ref: $ref not present (GPU box?) -- keeping whatever is prebuilt in $out" >&2
    exit 0
fi
[ -x "$hipify" ] || { echo "build_ref: $hipify missing: the reference is unbuildable here" >&2; exit 1; }

mkdir -p "$out"
stamp="$out/.built_from"
sig="$( { cat "$here/build_ref.sh" "$here/hip_adapter.h" "$here/ref_driver.cpp"; \
          find "$ref/cuda_rasterizer" -maxdepth 2 -type f \( -name '*.cu' -o -name '*.h' -o -name '*.cuh' \) -print0 | sort -z | xargs -0 cat; } | sha256sum | cut -d' ' -f1)"
if [ "${1:-}" != "--force" ] && [ -f "$stamp" ] && [ "$(cat "$stamp")" = "$sig" ] \
     && [ -f "$out/libstp_ref.so" ] && [ -f "$out/libstp_ref_ieee.so" ]; then
    exit 0
fi

tmp="$(mktemp -d)"
trap 'rm -rf "$tmp"' EXIT
mkdir -p "$tmp/src/stopthepop"
files="forward.cu backward.cu rasterizer_impl.cu forward.h backward.h forward_common.h rasterizer.h rasterizer_impl.h
       auxiliary.h config.h stopthepop/hierarchical_render.cuh stopthepop/resorted_render.cuh
       stopthepop/rasterizer_debug.h stopthepop/stopthepop_common.cuh"
for f in $files; do
    "$hipify" "$ref/cuda_rasterizer/$f" 2>/dev/null \
      | sed -e '/^#include ""$/d' \
            -e '/^#include <cub\//d' \
            -e 's/<< *</<<</g' -e 's/>> *>/>>>/g' \
            -e 's/<<<grid, {16, 4, 4}>>>/<<<grid, dim3(16, 4, 4)>>>/' \
      > "$tmp/src/$f"
done

common="-x hip --offload-arch=gfx950 -std=c++17 -O3 -fPIC -w -include $here/hip_adapter.h
        -I$tmp/src -I$ref/cuda_rasterizer -I$ref/third_party/glm"
build_variant() {   # name, extra flags
    local name="$1"; shift
    local pids=()
    for tu in forward backward rasterizer_impl; do
        $hipcc $common "$@" -c "$tmp/src/$tu.cu" -o "$tmp/$name.$tu.o" & pids+=($!)
    done
    $hipcc $common "$@" "-DSTP_REF_BUILD_INFO=\"hipify-perl + hipcc -O3 --offload-arch=gfx950 $*\"" \
        -c "$here/ref_driver.cpp" -o "$tmp/$name.driver.o" & pids+=($!)
    for p in "${pids[@]}"; do wait "$p"; done
    $hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/$name.so" \
        "$tmp/$name.forward.o" "$tmp/$name.backward.o" "$tmp/$name.rasterizer_impl.o" "$tmp/$name.driver.o"
}
build_variant libstp_ref -ffp-contract=fast &
v1=$!
build_variant libstp_ref_ieee -ffp-contract=off &
v2=$!
wait $v1; wait $v2
echo "$sig" > "$stamp"
echo "build_ref: built $out/libstp_ref.so and $out/libstp_ref_ieee.so"
