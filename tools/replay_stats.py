#!/usr/bin/env python3
"""Debug: per-wave-step lane statistics of the replay backward (needs a library built with -DSTP_REPLAY_STATS,
selected through STP_RASTER_LIB)."""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stopthepop-rasterization_amd"))
import torch
import bench
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _C, scenes
variant = sys.argv[1] if len(sys.argv) > 1 else "full"
workload = sys.argv[2] if len(sys.argv) > 2 else "C2"
dev = torch.device("cuda:0")
scene = scenes.config(workload, 1.0)
es = bench.settings_for(variant, workload)
t = lambda x: torch.tensor(x, device=dev)
rs = dgr.GaussianRasterizationSettings(image_height=scene.H, image_width=scene.W, tanfovx=scene.tanfovx, tanfovy=scene.tanfovy, bg=t(scene.bg),
    scale_modifier=1.0, viewmatrix=t(scene.viewmatrix), projmatrix=t(scene.projmatrix), inv_viewprojmatrix=t(scene.inv_viewprojmatrix),
    sh_degree=scene.sh_degree, campos=t(scene.campos), prefiltered=False, settings=es, render_depth=False, debug=False)
means3D = t(scene.means3D).requires_grad_(True)
means2D = torch.zeros_like(means3D, requires_grad=True)
color, radii = dgr.GaussianRasterizer(rs)(means3D, means2D, t(scene.opacities), shs=t(scene.shs), scales=t(scene.scales), rotations=t(scene.rotations))
L = _C._load()
out = (ctypes.c_ulonglong * 16)()
L.stp_debug_replay_stats(out)
(color * t(scene.dL_dout)).sum().backward()
torch.cuda.synchronize()
L.stp_debug_replay_stats(out)
steps, nw, ns = out[0], out[2], out[3]
print(f"{workload}-{variant}: wave-steps {steps}; per step: lanes that add to LDS after the pairwise merge {nw/steps:.1f}, "
      f"of which stragglers behind their window (global atomics) {ns/steps:.3f}")
print(f"   of the adding lanes: {out[4]/steps:.1f} hold a position that a lower adding lane holds too ({out[5]/steps:.1f} of them in the same 16-lane row); "
      f"distinct positions per step {(nw-out[4])/steps:.1f}; largest group on one position {out[6]/steps:.2f}")
