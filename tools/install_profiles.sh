#!/bin/bash
# Copy what tools/round_profiles.sh <tag> left under gpurun_out/ into profiles/<dst>_* (the committed evidence), with traffic.json
# pointing at the committed directories.   usage: tools/install_profiles.sh r06 [r06]
tag=$1; dst=${2:-$1}
for w in full full_detail min full_fwd c1 c3 c4_fwd c5 l1 c2l c2h; do [ -s gpurun_out/$tag/bench_$w.json ] && cp gpurun_out/$tag/bench_$w.json profiles/${dst}_bench_$w.json; done
for v in full min c3 c4 c5; do
  [ -d gpurun_out/${tag}_$v ] || continue
  mkdir -p profiles/${dst}_$v
  for f in kernel_stats.csv pmc1.txt pmc2.txt pmc3.txt pmc4.txt pmc5.txt pmc6.txt; do [ -s gpurun_out/${tag}_$v/$f ] && cp gpurun_out/${tag}_$v/$f profiles/${dst}_$v/$f; done
done
python - "$tag" "$dst" <<'PY'
import json, sys
tag, dst = sys.argv[1], sys.argv[2]
sys.path.insert(0, '.')
import bench
d = json.load(open(f'gpurun_out/{tag}/traffic.json'))
names = {'C2-full': 'full', 'C2-min': 'min', 'C3-full': 'c3', 'C4-full': 'c4', 'C5-full': 'c5'}
d['_dirs'] = {k: f'{dst}_{v}' for k, v in names.items() if k in d}
for v in names.values():
    d['_taken'] = d['_taken'].replace(f'{tag}_{v}', f'{dst}_{v}')
json.dump(d, open('profiles/traffic.json', 'w'), indent=1)
print('traffic.json matches the current csrc:', d['_csrc_sha256'] == bench.csrc_sha256())
PY
