#!/usr/bin/env python3
"""Counts that size the render kernels' work for one bench workload (run on the GPU box):
tile-list entries R, blended (pixel, entry) pairs B (from the blend log), replay wave-steps, per-tile list lengths."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stopthepop-rasterization_amd"))
import bench  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402
from diff_gaussian_rasterization import _C, scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C2")
ap.add_argument("--variant", default="full")
a = ap.parse_args()
dev = torch.device("cuda:0")
scene = scenes.config(a.workload, 1.0)
es = bench.settings_for(a.variant, a.workload)
t = lambda x: torch.tensor(x, device=dev)
rs = dgr.GaussianRasterizationSettings(
    image_height=scene.H, image_width=scene.W, tanfovx=scene.tanfovx, tanfovy=scene.tanfovy, bg=t(scene.bg), scale_modifier=1.0,
    viewmatrix=t(scene.viewmatrix), projmatrix=t(scene.projmatrix), inv_viewprojmatrix=t(scene.inv_viewprojmatrix),
    sh_degree=scene.sh_degree, campos=t(scene.campos), prefiltered=False, settings=es, render_depth=False, debug=False)
means3D = t(scene.means3D).requires_grad_(True)
means2D = torch.zeros_like(means3D, requires_grad=True)
color, radii = dgr.GaussianRasterizer(rs)(means3D, means2D, t(scene.opacities), shs=t(scene.shs), scales=t(scene.scales), rotations=t(scene.rotations))
fn = color.grad_fn
R = fn.num_rendered
img = fn.saved_tensors[11]
W, H = scene.W, scene.H
gx, gy = (W + 15) // 16, (H + 15) // 16
ranges = _C.image_array(img, W, H, "ranges").cpu().numpy().reshape(-1, 2)
lens = (ranges[:, 1] - ranges[:, 0]).astype(np.int64)
print(f"P={scene.P} visible={(radii > 0).sum().item()} R={R} tiles={gx * gy} list len mean={lens.mean():.1f} p50={np.median(lens):.0f} p99={np.percentile(lens, 99):.0f} max={lens.max()}")
if es.sort_settings.sort_mode == dgr.SortMode.HIER:
    n = _C.image_array(img, W, H, "n_contrib").cpu().numpy().reshape(H, W).astype(np.int64)
    flags = _C.image_array(img, W, H, "tile_flags").cpu().numpy()
    B = int(n.sum())
    # replay wave = 4 rows x 16 columns of a tile
    Hp, Wp = gy * 16, gx * 16
    pad = np.zeros((Hp, Wp), np.int64); pad[:H, :W] = n
    wv = pad.reshape(gy, 4, 4, gx, 16).max(axis=(2, 4))  # [tile_y][wave][tile_x]
    print(f"blended pairs B={B} ({B / (W * H):.1f} per pixel, max {n.max()})  replay wave-steps={int(wv.sum())} (x64 = {int(wv.sum()) * 64}; lane utilisation {B / (wv.sum() * 64):.2f})  overflow tiles={int((flags != 0).sum())}")
