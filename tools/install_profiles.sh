#!/bin/bash
# Copy what tools/round_profiles.sh <tag> left under gpurun_out/ into profiles/<dst>_* (the committed evidence), with traffic.json
# pointing at the committed directories.   usage: tools/install_profiles.sh r02f [r02]
tag=$1; dst=${2:-$1}
for w in full min full_fwd c1 c3 c4_fwd c5 l1 c2l; do [ -f gpurun_out/$tag/bench_$w.json ] && cp gpurun_out/$tag/bench_$w.json profiles/${dst}_bench_$w.json; done
for v in full min; do mkdir -p profiles/${dst}_$v; for f in kernel_stats.csv pmc1.txt pmc2.txt pmc3.txt pmc4.txt pmc5.txt pmc6.txt; do cp gpurun_out/${tag}_$v/$f profiles/${dst}_$v/$f; done; done
python - "$tag" "$dst" <<'PY'
import json, sys
tag, dst = sys.argv[1], sys.argv[2]
sys.path.insert(0, '.')
import bench
d = json.load(open(f'gpurun_out/{tag}/traffic.json'))
d['_dirs'] = {'C2-full': f'{dst}_full', 'C2-min': f'{dst}_min'}
d['_taken'] = d['_taken'].replace(f'{tag}_full', f'{dst}_full').replace(f'{tag}_min', f'{dst}_min')
json.dump(d, open('profiles/traffic.json', 'w'), indent=1)
print('traffic.json matches the current csrc:', d['_csrc_sha256'] == bench.csrc_sha256())
PY
