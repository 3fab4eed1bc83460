"""Generates tests/golden/ref/*.npz: outputs of THE REFERENCE ITSELF on seeded scenes.

What runs: oracle/_ref/libstp_ref_ieee.so = the reference's own cuda_rasterizer sources, translated by the ROCm
image's hipify-perl and compiled by hipcc for gfx950 with -ffp-contract=off (recipe, fix-ups and the adapter
header: oracle/ref_build/).  It needs a GPU, so this script is run on the MI355X box:

    gpurun -- 'python tests/golden/make_ref_golden.py gpurun_out/ref_golden'      # then copy into tests/golden/ref/

Each fixture holds DATA only: the scene's generator arguments + a hash of the generated inputs (inputs are a
pure function of the seed, diff_gaussian_rasterization/scenes.py), the settings dict, and what the reference
returned: num_rendered, radii, per-Gaussian state, sort keys, sorted list, tile ranges, image, final_T,
n_contrib, all nine gradient tensors.  tests/test_reference_golden.py (CPU) holds the oracle to them.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: F401,E402
from helpers import FULL_STP, settings_dict  # noqa: E402
from diff_gaussian_rasterization import scenes  # noqa: E402

GOLD = dict(P=1000, W=64, H=48, sigma_min=2.0, sigma_max=12.0, seed=31, camera="orbit")      # ~800 entries per tile
SPARSE = dict(P=600, W=128, H=96, sigma_min=1.0, sigma_max=10.0, seed=32, camera="origin")
NOLB = {**FULL_STP, "lb": False}

CASES = {
    "gold_global_z": dict(scene=GOLD, settings=settings_dict(0)),
    "gold_global_dist": dict(scene=GOLD, settings=settings_dict(0, order=1)),
    "gold_ptd_center": dict(scene=GOLD, settings=settings_dict(0, order=2)),
    "gold_ptd_max": dict(scene=GOLD, settings=settings_dict(0, order=3)),
    "gold_kbuffer16": dict(scene=GOLD, settings=settings_dict(2, per_pixel=16)),
    "gold_kbuffer4_ewa": dict(scene=GOLD, settings=settings_dict(2, per_pixel=4, ewa=True)),
    "gold_hier": dict(scene=GOLD, settings=settings_dict(3)),
    "gold_hier_cull": dict(scene=GOLD, settings=settings_dict(3, h44=True)),
    "gold_full_stp": dict(scene=GOLD, settings=settings_dict(**NOLB)),
    "gold_full_stp_lb": dict(scene=GOLD, settings=settings_dict(**FULL_STP)),
    "gold_full_stp_ewa": dict(scene=GOLD, settings=settings_dict(**{**NOLB, "ewa": True})),
    "gold_global_all_culling": dict(scene=GOLD, settings=settings_dict(0, rect=True, tight=True, tbc=True)),
    "gold_hier_h8_m12": dict(scene=GOLD, settings=settings_dict(3, per_pixel=8, tile_2x2=12, h44=True)),
    "gold_hier_h16_m20": dict(scene=GOLD, settings=settings_dict(3, per_pixel=16, tile_2x2=20)),
    "sparse_global_colors": dict(scene={**SPARSE, "use_sh": False}, settings=settings_dict(0)),
    "sparse_global_cov3d": dict(scene=SPARSE, settings=settings_dict(0), cov3D_precomp=True),
    "sparse_global_cov3d_ewa": dict(scene=SPARSE, settings=settings_dict(0, ewa=True), cov3D_precomp=True),
    "sparse_full_stp": dict(scene=SPARSE, settings=settings_dict(**NOLB)),
    "gold_depth_global": dict(scene=GOLD, settings=settings_dict(0), render_depth=True),
    "gold_depth_hier": dict(scene=GOLD, settings=settings_dict(3), render_depth=True),
    "gold_depth_kbuffer": dict(scene=GOLD, settings=settings_dict(2, per_pixel=16), render_depth=True),
}

STATE = ("depths", "means2D", "rects2D", "conic_opacity", "cov3D", "rgb", "clamped", "tiles_touched", "point_offsets")


def scene_hash(sc) -> str:
    h = hashlib.sha256()
    for a in (sc.means3D, sc.scales, sc.rotations, sc.opacities, sc.shs, sc.colors_precomp, sc.viewmatrix, sc.projmatrix,
              sc.inv_viewprojmatrix, sc.campos, sc.bg, sc.dL_dout):
        if a is not None:
            h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def run_case(case):
    from oracle import reference as ref
    sc = scenes.make_scene(**case["scene"])
    sd = case["settings"]
    c3 = scenes.covariance_from_scale_rotation(sc) if case.get("cov3D_precomp") else None
    depth = bool(case.get("render_depth"))
    f = ref.forward_scene(sc, sd, variant="ieee", cov3D_precomp=c3, render_depth=depth)
    out = dict(scene_json=json.dumps(case["scene"]), settings_json=json.dumps(sd), scene_sha256=scene_hash(sc),
               reference_build=ref.build_info("ieee"), render_depth=depth,
               num_rendered=f.num_rendered, radii=f.radii, color=f.color)
    if c3 is not None:
        out["cov3D_precomp"] = c3
    vis = f.radii > 0
    inv = sd["sort_settings"]["sort_mode"] != 0 or sd["sort_settings"]["sort_order"] >= 2
    for nm in STATE + (("cov3D_inv",) if inv else ()):
        a = f.array(nm)
        if nm not in ("tiles_touched", "point_offsets"):
            a = a.reshape(sc.P, -1).copy()
            a[~vis] = 0          # never written for culled Gaussians (uninitialised in the reference)
        out["state_" + nm] = a
    if f.num_rendered > 0:
        for nm in ("keys", "point_list", "ranges"):
            out[nm] = f.array(nm)
    out["final_T"] = f.array("final_T")
    if sd["sort_settings"]["sort_mode"] != 3:
        out["n_contrib"] = f.array("n_contrib")
    if not depth:
        g = f.backward(sc.dL_dout)
        for k, v in g.items():
            out["grad_" + k] = v
    out["mark_visible"] = ref.mark_visible(sc.means3D, sc.viewmatrix, sc.projmatrix)
    f.free()
    return out


if __name__ == "__main__":
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref")
    os.makedirs(dst, exist_ok=True)
    for name, case in CASES.items():
        out = run_case(case)
        path = os.path.join(dst, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, out["num_rendered"], os.path.getsize(path), flush=True)
