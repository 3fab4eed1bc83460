#!/usr/bin/env python3
"""Where the HOST time of a small frame goes: cProfile over N fwd+bwd steps of C1 (1k Gaussians, 256x256: 0.19 ms of GPU
stages per step, the rest is Python / ctypes / allocator / launch overhead).  python tools/host_profile.py [steps]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stopthepop-rasterization_amd")); sys.path.insert(0, ROOT)
import torch
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import scenes
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
wl = sys.argv[2] if len(sys.argv) > 2 else "C1"
dev = torch.device("cuda:0")
scene = scenes.config(wl)
es = bench.settings_for("full", wl)
t = lambda a, rg=False: None if a is None else torch.tensor(a, device=dev).requires_grad_(rg)
means3D, opac, scales, rots, shs = t(scene.means3D, True), t(scene.opacities, True), t(scene.scales, True), t(scene.rotations, True), t(scene.shs, True)
means2D = torch.zeros_like(means3D, requires_grad=True)
w_img = t(scene.dL_dout)
rs = dgr.GaussianRasterizationSettings(image_height=scene.H, image_width=scene.W, tanfovx=scene.tanfovx, tanfovy=scene.tanfovy, bg=t(scene.bg),
    scale_modifier=1.0, viewmatrix=t(scene.viewmatrix), projmatrix=t(scene.projmatrix), inv_viewprojmatrix=t(scene.inv_viewprojmatrix),
    sh_degree=scene.sh_degree, campos=t(scene.campos), prefiltered=False, settings=es, render_depth=False, debug=False)
raster = dgr.GaussianRasterizer(rs)
leaves = [means3D, means2D, opac, scales, rots, shs]
def step():
    for x in leaves: x.grad = None
    color, radii = raster(means3D, means2D, opac, shs=shs, scales=scales, rotations=rots)
    (color * w_img).sum().backward()
for _ in range(200): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
print(f"{wl}: {1000 * (time.perf_counter() - t0) / steps:.4f} ms/step unprofiled")
pr = cProfile.Profile(); pr.enable()
for _ in range(steps): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
