"""Stage-by-stage HIP-vs-oracle diagnostic (a script, not a pytest module): run on the GPU box as
`python tests/gpu_diag.py [quick]`; prints one line per stage and mode so that a single gpurun call
shows where a mismatch starts."""
import sys, os, time, traceback
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: F401  (sys.path)
from helpers import settings_dict, FULL_STP, GpuRun, oracle_run, psnr, max_abs
from diff_gaussian_rasterization import scenes


def cmp_stage(tag, name, a, b, exact=False):
    a = np.asarray(a); b = np.asarray(b)
    if a.shape != b.shape:
        print(f"   {tag} {name:16s} SHAPE MISMATCH {a.shape} vs {b.shape}"); return False
    if a.dtype.kind in "iu" or exact:
        bad = int(np.sum(a != b))
        print(f"   {tag} {name:16s} n={a.size} mismatches={bad}")
        return bad == 0
    d = max_abs(a, b); m = float(np.max(np.abs(b))) if b.size else 0
    print(f"   {tag} {name:16s} n={a.size} maxabs={d:.3e} (max|ref|={m:.3e}) bitdiff={int(np.sum(a.view(np.uint32)!=b.view(np.uint32)))}")
    return d <= 1e-4 * max(m, 1e-6)


def one(tag, scene, sd, backward=True):
    print(f"== {tag}: P={scene.P} {scene.W}x{scene.H} settings={sd['sort_settings']} cull={sd['culling_settings']} ewa={sd['proper_ewa_scaling']}")
    t0 = time.time(); f, og = oracle_run(scene, sd, backward); t1 = time.time()
    try:
        g = GpuRun(scene, sd, backward=backward)
    except Exception:
        traceback.print_exc(); return
    t2 = time.time()
    print(f"   oracle {t1-t0:.2f}s gpu {t2-t1:.2f}s  R oracle={f.num_rendered} gpu={g.num_rendered}")
    cmp_stage(tag, "radii", g.radii, f.radii)
    vis = f.radii > 0
    for nm, per in (("tiles_touched", 1), ("point_offsets", 1)):
        cmp_stage(tag, nm, g.geom_array(nm).view(np.uint32), f.array(nm))
    for nm, per in (("depths", 1), ("means2D", 2), ("rects2D", 2), ("conic_opacity", 4), ("rgb", 3), ("cov3D", 6), ("cov3D_inv", 12)):
        try:
            ga = g.geom_array(nm).reshape(-1, per)[vis]; oa = f.array(nm).reshape(-1, per)[vis]
        except KeyError:
            continue
        if nm == "rgb" and scene.shs is None: continue
        cmp_stage(tag, nm, ga, oa)
    if f.num_rendered == g.num_rendered and f.num_rendered > 0:
        cmp_stage(tag, "keys", g.binning_array("keys"), f.array("keys"))
        cmp_stage(tag, "point_list", g.binning_array("point_list"), f.array("point_list"))
        cmp_stage(tag, "ranges", g.image_array("ranges").view(np.uint32), f.array("ranges"))
    print(f"   {tag} IMAGE psnr={psnr(g.color, f.color):.2f} dB maxabs={max_abs(g.color, f.color):.3e} bitdiff={int(np.sum(g.color.view(np.uint32)!=f.color.view(np.uint32)))}/{g.color.size}")
    cmp_stage(tag, "final_T", g.image_array("final_T"), f.array("final_T"))
    if backward and g.grads is not None:
        for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dcolors"):
            if g.grads.get(k) is None or og.get(k) is None or og[k].size == 0: continue
            a, b = g.grads[k], og[k]
            if k == "dL_dmeans2D": a, b = a[:, :2], b[:, :2]
            d = max_abs(a, b); m = float(np.max(np.abs(b)))
            print(f"   {tag} {k:16s} maxabs={d:.3e} max|ref|={m:.3e} rel={d/max(m,1e-30):.2e}")


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    import torch
    print("torch", torch.__version__, "cuda", torch.cuda.is_available(), torch.cuda.get_device_name(0) if torch.cuda.is_available() else None)
    c1 = scenes.config("C1")
    c1rgb = scenes.config("C1", use_sh=False, camera="orbit")
    dense = scenes.make_scene(P=6000, W=96, H=80, sigma_min=2.0, sigma_max=14.0, seed=11, camera="orbit")
    one("C1-global", c1, settings_dict(0))
    one("C1rgb-global-dist", c1rgb, settings_dict(0, order=1))
    one("C1-kbuf16", c1, settings_dict(2, per_pixel=16))
    one("C1-hier", c1, settings_dict(3))
    one("dense-hier", dense, settings_dict(3))
    one("dense-hier-cull", dense, settings_dict(3, h44=True))
    one("dense-full-stp", dense, settings_dict(**FULL_STP))
    if not quick:
        one("dense-global", dense, settings_dict(0))
        one("dense-kbuf16", dense, settings_dict(2, per_pixel=16))
        one("dense-kbuf4-ewa", dense, settings_dict(2, per_pixel=4, ewa=True))
        one("dense-ptd-center", dense, settings_dict(0, order=2))
        one("dense-tbc-rect-tight", dense, settings_dict(0, rect=True, tight=True, tbc=True))
        one("C1-full", scenes.make_scene(P=400, W=48, H=32, sigma_min=1, sigma_max=8, seed=5), settings_dict(1), backward=False)
