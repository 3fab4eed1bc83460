#!/usr/bin/env python3
"""Root-cause probe for image differences under hierarchical_4x4_culling (fuzz14 case 603 of round 1):
runs product and oracle on one scene, lists every 4x4 sub-tile in which values moved by more than 2e-6, and for each prints the
tile-list entries whose culling alpha (opacity * exp(-power), evaluated in double by the oracle) is closest to 1/255, the
relative distance in fp32 ulps, and the largest pixel difference inside the sub-tile.

    python tools/cull_flip_probe.py            # the recorded reproducer
"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "stopthepop-rasterization_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
from helpers import GpuRun, oracle_run, settings_dict  # noqa: E402
from diff_gaussian_rasterization import scenes  # noqa: E402

SC = {'P': 9000, 'W': 33, 'H': 48, 'sigma_min': 4.0, 'sigma_max': 6.0, 'seed': 742987, 'camera': 'orbit', 'use_sh': True,
      'opacity_range': (0.004, 0.05), 'z_range': (0.5, 3.0)}
SD = {'mode': 3, 'order': 2, 'rect': True, 'tight': False, 'tbc': False, 'h44': True, 'lb': True, 'ewa': True, 'per_pixel': 4, 'tile_2x2': 12}

scene = scenes.make_scene(**SC)
sd = settings_dict(**SD)
g = GpuRun(scene, sd, backward=False)
f, _ = oracle_run(scene, sd, backward=False)
d = np.abs(g.color.astype(np.float64) - f.color.astype(np.float64))
moved = d > 2e-6
print(json.dumps({"values_moved": int(moved.sum()), "max_abs": float(d.max()), "one_over_255": 1 / 255}))
gx = (scene.W + 15) // 16
ys, xs = np.nonzero(moved.any(axis=0))
ULP = 2.0 ** -23
for sy, sx in sorted({(int(y) // 4, int(x) // 4) for y, x in zip(ys, xs)}):
    al = f.cull_alpha((sy // 4) * gx + (sx // 4), 4 * sx, 4 * sy)
    k = int(np.argmin(np.abs(al * 255.0 - 1.0)))
    blk = d[:, 4 * sy:4 * sy + 4, 4 * sx:4 * sx + 4]
    print(json.dumps({"sub_tile_corner": [4 * sx, 4 * sy], "values_moved": int((blk > 2e-6).sum()), "max_abs_in_sub_tile": float(blk.max()),
                      "nearest_entry_list_pos": k, "alpha_exact": float(al[k]), "alpha_minus_threshold_rel": float(al[k] * 255.0 - 1.0),
                      "distance_in_fp32_ulps": float(abs(al[k] * 255.0 - 1.0) / ULP)}))
