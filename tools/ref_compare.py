#!/usr/bin/env python3
"""Side-by-side of THE REFERENCE compiled for gfx950 (oracle/_ref, see oracle/ref_build/build_ref.sh), the CPU
oracle and -- with --product -- the HIP product, on the same seeded scenes.  Runs on the GPU box:

    python tools/ref_compare.py [--variant ieee|fast] [--product] [--cases small|all] [--json out.json]

Prints one line per (scene, settings, array) with the number of differing elements / the max error, and a
summary.  Test infrastructure; the product is only ever the thing compared.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stopthepop-rasterization_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from diff_gaussian_rasterization import scenes  # noqa: E402
from helpers import FULL_STP, settings_dict  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle import reference as ref  # noqa: E402

GRADS = ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
         "dL_drotations")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def ulps(a, b):
    """max distance in units of the last place between two float32 arrays (same sign assumed for finite values)"""
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return int(np.max(np.abs(a - b))) if a.size else 0


def compare(scene, sd, variant, backward=True, out=None, name=""):
    rf = ref.forward_scene(scene, sd, variant=variant)
    of = orc.forward_scene(scene, sd)
    rec = {"case": name, "variant": variant, "R_ref": rf.num_rendered, "R_orc": of.num_rendered}
    vis = rf.radii > 0
    rec["radii_diff"] = int((rf.radii != of.radii).sum())
    rec["tiles_diff"] = int((rf.array("tiles_touched") != of.array("tiles_touched")).sum())
    rec["offsets_equal"] = bool(np.array_equal(rf.array("point_offsets"), of.array("point_offsets")))
    both = vis & (of.radii > 0)
    inv = sd["sort_settings"]["sort_mode"] != 0 or sd["sort_settings"]["sort_order"] >= 2
    for nm, per in (("depths", 1), ("means2D", 2), ("rects2D", 2), ("conic_opacity", 4), ("cov3D", 6), ("rgb", 3)) + \
            ((("cov3D_inv", 12),) if inv else ()):
        a, b = rf.array(nm).reshape(-1, per)[both], of.array(nm).reshape(-1, per)[both]
        rec[nm + "_bitdiff"] = int((a.view(np.uint32) != b.view(np.uint32)).sum())
        rec[nm + "_ulps"] = ulps(a, b)
    a, b = rf.array("clamped").reshape(-1, 3)[both], of.array("clamped").reshape(-1, 3)[both]
    rec["clamped_diff"] = int((a != b).sum())
    if rf.num_rendered == of.num_rendered and rf.num_rendered > 0:
        for nm in ("keys_unsorted", "values_unsorted", "keys", "point_list"):
            rec[nm + "_diff"] = int((rf.array(nm) != of.array(nm)).sum())
        rec["ranges_diff"] = int((rf.array("ranges") != of.array("ranges")).sum())
    d = np.abs(rf.color.astype(np.float64) - of.color.astype(np.float64))
    rec["img_max_abs"] = float(d.max())
    rec["img_gt_2e-6"] = int((d > 2e-6).sum())
    rec["img_bitdiff"] = int((rf.color.view(np.uint32) != of.color.view(np.uint32)).sum())
    mse = float(np.mean(d ** 2))
    rec["img_psnr"] = 200.0 if mse == 0 else float(10 * np.log10(1.0 / mse))
    a, b = rf.array("final_T"), of.array("final_T")
    rec["final_T_max_abs"] = float(np.max(np.abs(a - b)))
    if sd["sort_settings"]["sort_mode"] != 3:
        rec["n_contrib_diff"] = int((rf.array("n_contrib") != of.array("n_contrib")).sum())
    if backward and sd["sort_settings"]["sort_mode"] != 1:
        rg, og = rf.backward(scene.dL_dout), of.backward(scene.dL_dout)
        for k in GRADS:
            a, b = rg[k], og[k]
            if k == "dL_dmeans2D":
                a, b = a[:, :2], b[:, :2]
            if k == "dL_dconic":
                a, b = a.reshape(-1, 4)[:, [0, 1, 3]], b.reshape(-1, 4)[:, [0, 1, 3]]
            rec[k + "_rel"] = rel(a, b)
    if out is not None:
        out.append(rec)
    rf.free(); of.free()
    return rec


def cases(which):
    C1 = dict(P=1000, W=256, H=256, sigma_min=1.0, sigma_max=12.0, seed=1)
    DENSE = dict(P=6000, W=96, H=80, sigma_min=2.0, sigma_max=14.0, seed=11, camera="orbit")
    sds = {"global_z": settings_dict(0), "global_dist": settings_dict(0, order=1), "ptd_center": settings_dict(0, order=2),
           "ptd_max": settings_dict(0, order=3), "kbuffer16": settings_dict(2, per_pixel=16), "kbuffer4_ewa": settings_dict(2, per_pixel=4, ewa=True),
           "hier": settings_dict(3), "hier_cull": settings_dict(3, h44=True), "full_stp": settings_dict(**FULL_STP),
           "full_stp_nolb": settings_dict(**{**FULL_STP, "lb": False}),
           "global_all_culling": settings_dict(0, rect=True, tight=True, tbc=True),
           "full_stp_ewa": settings_dict(**{**FULL_STP, "ewa": True, "lb": False}),
           }
    if which == "ppx":      # the reference's PER_PIXEL_FULL kernel, on its own (it faulted on wave64 in the first survey)
        return [("dense/ppx_full", DENSE, settings_dict(1)), ("c1/ppx_full", C1, settings_dict(1))]
    out = []
    for sname, skw in (("c1", C1), ("dense", DENSE)):
        for k, sd in sds.items():
            out.append((f"{sname}/{k}", skw, sd))
    if which == "all":
        for head, mid in ((8, 8), (16, 8), (4, 12), (8, 12), (16, 20), (4, 20)):
            out.append((f"dense/hier_h{head}_m{mid}", DENSE, settings_dict(3, per_pixel=head, tile_2x2=mid, h44=True)))
        for w in (1, 2, 4, 8, 12, 20, 24):
            out.append((f"dense/kbuffer{w}", DENSE, settings_dict(2, per_pixel=w)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="ieee")
    ap.add_argument("--cases", default="small")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    print("reference build:", ref.build_info(args.variant), flush=True)
    recs = []
    for name, skw, sd in cases(args.cases):
        t0 = time.time()
        try:
            r = compare(scenes.make_scene(**skw), sd, args.variant, out=recs, name=name)
            print(json.dumps(r), f"# {time.time() - t0:.1f}s", flush=True)
        except Exception as e:  # keep going: this is a survey
            print(json.dumps({"case": name, "error": repr(e)}), flush=True)
            recs.append({"case": name, "error": repr(e)})
    if args.json:
        with open(args.json, "w") as f:
            json.dump(recs, f, indent=1)


if __name__ == "__main__":
    main()
