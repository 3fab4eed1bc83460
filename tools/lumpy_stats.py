#!/usr/bin/env python3
"""Blend-log behaviour on a LUMPY scene (workload C2L: 40 % of C2's Gaussians in 12 clusters): tile-list lengths, blends per pixel,
tiles whose log overflowed (they take the re-sorting backward), and the stage times with the replay and with the re-sorting
backward.   python tools/lumpy_stats.py [workload]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "stopthepop-rasterization_amd"), ROOT): sys.path.insert(0, p)
import numpy as np
from helpers import GpuRun, settings_dict, FULL_STP
from diff_gaussian_rasterization import scenes
wl = sys.argv[1] if len(sys.argv) > 1 else "C2L"
sc = scenes.config(wl)
for name, sd in (("full", settings_dict(**FULL_STP)), ("min", settings_dict(3))):
    g = GpuRun(sc, sd, backward=True)
    r = g.image_array("ranges").view(np.uint32).reshape(-1, 2); lens = (r[:, 1] - r[:, 0]).astype(np.int64)
    T = ((sc.W + 15) // 16) * ((sc.H + 15) // 16); lens = lens[:T]
    nc = g.image_array("n_contrib").view(np.uint32).astype(np.int64)
    flags = g._C.image_array(g.img, sc.W, sc.H, "tile_flags").cpu().numpy().view(np.uint32)[:T]
    print(json.dumps({"workload": wl, "variant": name, "num_rendered": int(g.num_rendered), "list_len_mean": float(lens.mean()), "list_len_p99": float(np.percentile(lens, 99)),
                      "list_len_max": int(lens.max()), "blends_per_pixel_mean": float(nc.mean()), "blends_p99": float(np.percentile(nc, 99)), "blends_max": int(nc.max()),
                      "pixels_over_256": int((nc > 256).sum()), "tiles_overflowed": int((flags != 0).sum()), "tiles": int(T)}))
for env in ({}, {"STP_BACKWARD": "resort"}):
    for v in ("full", "min"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--variant", v, "--no-cpu-baseline", "--steps", "10"],
                             env={**os.environ, **env}, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print(env.get("STP_BACKWARD", "replay"), v, d["value"], d["ms_per_step"], d["stage_ms"])
