#!/usr/bin/env python3
"""Bitwise comparison of two builds of the library on one frame (GPU box): tile lists, image, final_T, per-pixel blend counts and the
whole blend log -- what "the kernel work changed nothing" means.   usage: tools/compare_builds.py <libA.so> <libB.so> [workload] [variant] [NAME=VALUE ...]
(NAME=VALUE: environment switches set just before library B is loaded -- the library reads its switches once, at its first forward)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stopthepop-rasterization_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from diff_gaussian_rasterization import _C, scenes

libA, libB = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
workload = sys.argv[3] if len(sys.argv) > 3 else "C2"
variant = sys.argv[4] if len(sys.argv) > 4 else "full"
dev = torch.device("cuda:0")
sc = scenes.config(workload)
sd = bench.settings_for(variant, workload).to_dict()
sd["_record_blend_log"] = True
t = lambda a: torch.tensor(a, device=dev)
empty = torch.Tensor([])
args = (t(sc.bg), t(sc.means3D), empty, t(sc.opacities), t(sc.scales), t(sc.rotations), 1.0, empty, t(sc.viewmatrix), t(sc.projmatrix),
        t(sc.inv_viewprojmatrix), sc.tanfovx, sc.tanfovy, sc.H, sc.W, t(sc.shs), 3, t(sc.campos), False, sd, False, False)
out = {}
for name, lib in (("A", libA), ("B", libB)):
    if name == "B":
        for kv in sys.argv[5:]:
            os.environ[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
    _C.use_library(lib)
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(*args)
    torch.cuda.synchronize()
    n = _C.image_array(img, sc.W, sc.H, "n_contrib").clone()
    depth = _C.blend_log_depth(img)   # (the first forward of a kind in a library: the default depth, the one stp_image_layout reports "blend_log" for)
    out[name] = dict(R=R, color=color.clone(), list=_C.binning_array(binning, R, "point_list").clone(), keys=_C.binning_array(binning, R, "keys").clone(),
                     final_T=_C.image_array(img, sc.W, sc.H, "final_T").clone(), n=n, flags=_C.image_array(img, sc.W, sc.H, "tile_flags").clone(),
                     log=_C.image_array(img, sc.W, sc.H, "blend_log").clone())
    del geom, binning, img
    _C.clear_scratch_pool(dev)
a, b = out["A"], out["B"]
ok = a["R"] == b["R"]
for k in ("list", "keys", "color", "final_T", "n", "flags"):
    same = torch.equal(a[k].view(torch.uint8), b[k].view(torch.uint8))
    ok &= same
    print(f"{k:8s} identical: {same}")
# the log is only defined where records were written: record k of a pixel for k < n (layout [tile][wave][record][lane])
T = ((sc.W + 15) // 16) * ((sc.H + 15) // 16)
la, lb = a["log"].view(T, 4, -1, 64), b["log"].view(T, 4, -1, 64)   # (rows per wave: 256, or 257 in a build with STP_LOG_UNCOND)
gx = (sc.W + 15) // 16
py, px = torch.meshgrid(torch.arange(sc.H, device=dev), torch.arange(sc.W, device=dev), indexing="ij")
tile = (py // 16) * gx + (px // 16)
lx, ly = px % 16, py % 16
wave = ly // 4
sub, qx, qy = lx // 4, lx % 4, ly % 4
lane = sub * 16 + ((qy // 2) * 2 + (qx // 2)) * 4 + (qy % 2) * 2 + (qx % 2)
nrec = a["n"].view(sc.H, sc.W).clamp(max=depth).to(torch.int64)
diff = 0
for k in range(0, depth):
    m = nrec > k
    if not bool(m.any()):
        break
    ra = la[tile[m], wave[m], k, lane[m]]
    rb = lb[tile[m], wave[m], k, lane[m]]
    diff += int((ra != rb).sum())
print(f"blend log records compared: {int(nrec.sum())}, differing: {diff}")
ok &= diff == 0
print("IDENTICAL" if ok else "DIFFERENT")
sys.exit(0 if ok else 1)
