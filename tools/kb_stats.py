#!/usr/bin/env python3
"""Debug: where the k-buffer forward's candidates land in the per-pixel window (library built with -DSTP_KB_STATS, selected through STP_RASTER_LIB).
usage: STP_RASTER_LIB=<lib> tools/kb_stats.py [workload]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stopthepop-rasterization_amd"))
import torch
import bench
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _C, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
dev = torch.device("cuda:0")
scene = scenes.config(name, 1.0)
es = bench.settings_for("full", name)
t = lambda x: torch.tensor(x, device=dev)
rs = dgr.GaussianRasterizationSettings(image_height=scene.H, image_width=scene.W, tanfovx=scene.tanfovx, tanfovy=scene.tanfovy, bg=t(scene.bg),
    scale_modifier=1.0, viewmatrix=t(scene.viewmatrix), projmatrix=t(scene.projmatrix), inv_viewprojmatrix=t(scene.inv_viewprojmatrix),
    sh_degree=scene.sh_degree, campos=t(scene.campos), prefiltered=False, settings=es, render_depth=False, debug=False)
means3D = t(scene.means3D)
L = _C._load()
out = (ctypes.c_ulonglong * 40)()
L.stp_debug_kb_stats(out)
color, radii = dgr.GaussianRasterizer(rs)(means3D, torch.zeros_like(means3D), t(scene.opacities), shs=t(scene.shs), scales=t(scene.scales), rotations=t(scene.rotations))
torch.cuda.synchronize()
L.stp_debug_kb_stats(out)
steps, passing, live = out[32], out[33], out[34]
print(f"{name}: {steps} candidate steps (per wave), {passing} passing (lane, candidate) pairs = {passing / max(steps, 1):.1f} of 64 lanes, live lanes {live / max(steps, 1):.1f}")
tot = sum(out[0:16])
print("passing candidates by slots passed from the back:", " ".join(f"{i}:{out[i] / tot:.3f}" for i in range(16)))
print("mean per-lane distance:", sum(i * out[i] for i in range(16)) / tot)
ts = sum(out[16:32])
print("steps by the wave's LARGEST distance:          ", " ".join(f"{i}:{out[16 + i] / ts:.3f}" for i in range(16)))
print("mean of the wave's largest distance:", sum(i * out[16 + i] for i in range(16)) / ts)
