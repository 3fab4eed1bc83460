#!/usr/bin/env python3
"""Per-pixel blend counts of the BASELINE workloads (GPU box): what a blend log of depth D would hold -- percentiles of the records per
pixel and the share of TILES with a pixel above D (such a tile's backward falls back to the re-sorting kernel).
   usage: tools/log_depth_stats.py [workload ...]      (default: C2-full C2-min C3 C5)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stopthepop-rasterization_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from diff_gaussian_rasterization import _C, scenes

dev = torch.device("cuda:0")
for label in (sys.argv[1:] or ["C2-full", "C2-min", "C3", "C5"]):
    name, _, variant = label.partition("-")
    sc = scenes.config(name)
    sd = bench.settings_for(variant or "full", name).to_dict()
    sd["_record_blend_log"] = True
    t = lambda a: torch.tensor(a, device=dev)
    empty = torch.Tensor([])
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(
        t(sc.bg), t(sc.means3D), empty, t(sc.opacities), t(sc.scales), t(sc.rotations), 1.0, empty, t(sc.viewmatrix), t(sc.projmatrix),
        t(sc.inv_viewprojmatrix), sc.tanfovx, sc.tanfovy, sc.H, sc.W, t(sc.shs), 3, t(sc.campos), False, sd, False, False)
    n = _C.image_array(img, sc.W, sc.H, "n_contrib").view(sc.H, sc.W).to(torch.int64)
    gx, gy = (sc.W + 15) // 16, (sc.H + 15) // 16
    pad = torch.zeros(gy * 16, gx * 16, dtype=torch.int64, device=dev)
    pad[:sc.H, :sc.W] = n
    tmax = pad.view(gy, 16, gx, 16).amax(dim=(1, 3)).flatten()
    q = torch.quantile(n.flatten().float(), torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev)).tolist()
    shares = {d: float((tmax > d).float().mean()) for d in (64, 96, 128, 160, 192, 256)}
    print(f"{label}: R={R} blends/pixel mean {n.float().mean():.1f} median {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} p99.9 {q[3]:.0f} max {int(n.max())}; "
          f"tiles with a pixel above depth D: " + ", ".join(f"D={d}: {100 * s:.2f} %" for d, s in shares.items()))
    del geom, binning, img
    _C.clear_scratch_pool(dev)
