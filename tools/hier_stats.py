#!/usr/bin/env python3
"""Debug: consumption-round statistics of the recording hierarchical forward (library built with -DSTP_HIER_STATS,
selected through STP_RASTER_LIB)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stopthepop-rasterization_amd"))
import torch
import bench
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _C, scenes
variant = sys.argv[1] if len(sys.argv) > 1 else "full"
dev = torch.device("cuda:0")
scene = scenes.config("C2", 1.0)
es = bench.settings_for(variant, "C2")
t = lambda x: torch.tensor(x, device=dev)
rs = dgr.GaussianRasterizationSettings(image_height=scene.H, image_width=scene.W, tanfovx=scene.tanfovx, tanfovy=scene.tanfovy, bg=t(scene.bg),
    scale_modifier=1.0, viewmatrix=t(scene.viewmatrix), projmatrix=t(scene.projmatrix), inv_viewprojmatrix=t(scene.inv_viewprojmatrix),
    sh_degree=scene.sh_degree, campos=t(scene.campos), prefiltered=False, settings=es, render_depth=False, debug=False)
means3D = t(scene.means3D).requires_grad_(True)
means2D = torch.zeros_like(means3D, requires_grad=True)
L = _C._load()
out = (ctypes.c_ulonglong * 16)()
L.stp_debug_hier_stats(out)
color, radii = dgr.GaussianRasterizer(rs)(means3D, means2D, t(scene.opacities), shs=t(scene.shs), scales=t(scene.scales), rotations=t(scene.rotations))
torch.cuda.synchronize()
L.stp_debug_hier_stats(out)
rounds = sum(out[1:5])
print(f"{variant}: consumption rounds {rounds} ({rounds / 32640:.1f} per wave); by participating rows 1/2/3/4: "
      f"{out[1]/rounds:.3f} {out[2]/rounds:.3f} {out[3]/rounds:.3f} {out[4]/rounds:.3f}; mean rows {sum(i*out[i] for i in range(1,5))/rounds:.2f}; "
      f"live lanes per round {out[8]/rounds:.1f}; draining rounds {out[9]/rounds:.3f}")
if out[10]:
    print(f"head pre-test: {out[10]} candidates shown (per quad), {out[11]} kept ({out[11]/out[10]:.3f}); {out[12]} group steps "
          f"({out[12] / 32640:.1f} per wave) carrying {out[13]} candidates = {out[13]/out[12]/64:.3f} of their 64 slots; {out[14]/out[12]:.3f} of them in the final drain")
