#!/bin/bash
# Per-kernel A/B of library builds on ONE box (box-to-box variance is +-3 %): rocprofv3 kernel trace of a short bench run with
# each library, average duration of the kernels matching a pattern.   tools/ab_kernels.sh '<egrep pattern>' <lib...>
pat=$1; shift
export TMPDIR=/tmp
for L in "$@"; do
  d=/tmp/abk_$(basename $L); rm -rf $d
  (cd /tmp && STP_RASTER_LIB=$(realpath $OLDPWD/$L) rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python $OLDPWD/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "== $(basename $L)"; python - "$f" "$pat" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("   %9.1f us  x%-4s %s" % (float(r["AverageNs"]) / 1000, r["Calls"], r["Name"].replace("stp::(anonymous namespace)::", "")[:80]))
PY
done
