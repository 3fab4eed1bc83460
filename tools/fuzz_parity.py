#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: random small scenes x random settings, every check of
tests/test_gpu_parity.py::check_against_oracle (bit-exact state, image, gradients vs the CPU oracle).
    python tools/fuzz_parity.py [--cases 150] [--seed 1] [--seconds 600]
Prints one line per failure (with the exact reproducer) and a summary; exit code 1 on any failure."""
import argparse, os, random, sys, time, traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "stopthepop-rasterization_amd"), ROOT):
    sys.path.insert(0, p)
import conftest  # noqa: F401,E402  (path setup)
import numpy as np  # noqa: E402
from helpers import GpuRun, oracle_run, settings_dict  # noqa: E402
from test_gpu_parity import check_against_oracle  # noqa: E402
from diff_gaussian_rasterization import scenes  # noqa: E402


def heavy_case(rng):
    """Dense scenes: hundreds to thousands of entries per tile, many blends per pixel (blend-log overflow, lists beyond the LDS
    sort capacity, several replay windows)."""
    mode = rng.choice([0, 2, 3, 3, 3])
    sd = dict(mode=mode, order=rng.choice([0, 1, 2, 3]), rect=rng.random() < 0.5, tight=rng.random() < 0.5,
              tbc=rng.random() < 0.5, h44=(mode == 3 and rng.random() < 0.6), lb=True, ewa=rng.random() < 0.2)
    if mode == 3:
        sd["per_pixel"], sd["tile_2x2"] = rng.choice([(4, 8), (4, 8), (8, 12), (16, 20)])
    else:
        sd["per_pixel"] = rng.choice([4, 16, 24])
    smin = rng.choice([2.0, 5.0, 10.0])
    sc = dict(P=rng.choice([8000, 20000, 40000]), W=rng.choice([48, 96, 200]), H=rng.choice([32, 64, 120]),
              sigma_min=smin, sigma_max=smin * rng.choice([2.0, 4.0]), seed=rng.randrange(1, 10**6), camera="orbit",
              opacity_range=rng.choice([(0.05, 0.6), (0.01, 0.08), (0.004, 0.03)]))
    return sc, sd


def random_case(rng):
    mode = rng.choice([0, 0, 2, 2, 3, 3, 3, 3])
    sd = dict(mode=mode, order=rng.choice([0, 1, 2, 3]), rect=rng.random() < 0.5, tight=rng.random() < 0.5,
              tbc=rng.random() < 0.5, h44=(mode == 3 and rng.random() < 0.6), lb=rng.random() < 0.5, ewa=rng.random() < 0.3)
    if mode == 3:
        sd["per_pixel"], sd["tile_2x2"] = rng.choice([(4, 8), (4, 8), (8, 8), (16, 8), (4, 12), (8, 12), (16, 20), (4, 20)])
    elif mode == 2:
        sd["per_pixel"] = rng.choice([1, 2, 4, 8, 12, 16, 16, 20, 24])
    smin = rng.choice([0.4, 1.0, 2.0, 4.0])
    sc = dict(P=rng.choice([1, 7, 300, 1500, 4000, 9000]), W=rng.choice([16, 33, 64, 100, 160, 250]), H=rng.choice([16, 31, 48, 96, 130]),
              sigma_min=smin, sigma_max=smin * rng.choice([1.5, 4.0, 10.0]), seed=rng.randrange(1, 10**6),
              camera=rng.choice(["orbit", "orbit", "origin"]), use_sh=rng.random() < 0.8,
              opacity_range=rng.choice([(0.05, 0.6), (0.3, 0.99), (0.004, 0.05)]), z_range=rng.choice([(2.0, 12.0), (0.5, 3.0), (0.05, 40.0)]))
    return sc, sd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--heavy", action="store_true", help="dense scenes (overflowing blend logs, long lists) instead of the small ones")
    ap.add_argument("--fma", action="store_true", help="sweep the second shipped library (libstp_raster_fma.so) against the oracle's fma depth order")
    args = ap.parse_args()
    if args.fma:
        from diff_gaussian_rasterization import _C
        from oracle import oracle as orc
        _C.use_library("fma")
        orc.set_flag("ieee_depth", 0)
    rng = random.Random(args.seed)
    t0, done, bad = time.time(), 0, 0
    for i in range(args.cases):
        if time.time() - t0 > args.seconds:
            break
        sc, sd = heavy_case(rng) if args.heavy else random_case(rng)
        scene = scenes.make_scene(**sc)
        if not args.heavy and rng.random() < 0.3:  # lower active SH degree / different scale modifier (Scene fields the API passes through)
            scene.sh_degree = rng.choice([0, 1, 2]) if scene.shs is not None else 0
        if not args.heavy and rng.random() < 0.3:
            scene.scale_modifier = rng.choice([0.5, 0.8, 1.7])
        try:
            kind = rng.random()
            if not args.heavy and kind < 0.12:    # depth visualisation (forward only, every sort mode incl. PPX_FULL)
                sdv = dict(sd); sdv["mode"] = rng.choice([0, 1, 2, 3]) if scene.P <= 1500 and scene.W * scene.H <= 6400 else sd["mode"]
                if sdv["mode"] != 3: sdv.pop("tile_2x2", None); sdv["h44"] = False
                elif sd["mode"] != 3: sdv["per_pixel"], sdv["tile_2x2"] = 4, 8
                if sdv["mode"] in (0, 1): sdv["per_pixel"] = 4
                if sdv["mode"] == 2 and sdv.get("per_pixel", 4) not in (1, 2, 4, 8, 12, 16, 20, 24): sdv["per_pixel"] = 16
                g = GpuRun(scene, settings_dict(**sdv), backward=False, render_depth=True)
                f, _ = oracle_run(scene, settings_dict(**sdv), backward=False, render_depth=True)
                # (an empty frame has min == max: 0/0 in the reference's normalisation, NaN on both sides)
                assert np.array_equal(np.isnan(g.color), np.isnan(f.color)), "render_depth: NaN pattern"
                dd = np.nan_to_num(np.abs(g.color.astype(np.float64) - f.color))
                # (the normalisation divides by the frame's depth range: a nearly flat frame amplifies 1e-6 arbitrarily)
                assert dd.max() <= 2e-3 or not (np.nanmax(f.color) - np.nanmin(f.color) >= 0.05), ("render_depth", float(dd.max()))
            elif not args.heavy and kind < 0.30:  # inference forward (no blend log) of the same settings
                check_against_oracle(scene, settings_dict(**sd), backward=False)
            elif not args.heavy and kind < 0.40:  # a tile-row window (what a rank of the tile-row sharding renders)
                gy = (scene.H + 15) // 16
                y0 = rng.randrange(0, gy); y1 = rng.randrange(y0 + 1, gy + 1)
                g = GpuRun(scene, settings_dict(**sd), backward=False, tile_rows=(y0, y1))
                f, _ = oracle_run(scene, settings_dict(**sd), backward=False, tile_rows=(y0, y1))
                rows = slice(16 * y0, min(16 * y1, scene.H))
                dd = np.abs(g.color[:, rows].astype(np.float64) - f.color[:, rows])
                assert g.num_rendered == f.num_rendered and dd.max() <= 1.0 / 255.0 + 1e-6 and int((dd > 2e-6).sum()) <= 6, ("tile rows", y0, y1, float(dd.max()), int(np.isnan(g.color[:, rows]).sum()), int(np.isnan(f.color[:, rows]).sum()), g.num_rendered)
            elif args.heavy:  # ~500-1000 blends per pixel: rounding of the transmittance product accumulates (seen: 2e-5), faint
                # Gaussians (opacity at the 1/255 threshold) have gradients that hang on single threshold decisions
                # and the 4x4 culling test is such a decision for a whole sub-tile (16 pixels x 3 channels at once)
                check_against_oracle(scene, settings_dict(**sd), backward=True, img_tol=4e-5, grad_tol=2e-3, flip_grad_tol=5e-2)
            elif scene.P < 50:  # a handful of Gaussians: "relative to the largest entry" is relative to values that are themselves
                # the result of cancellation (rotation gradient of a near-isotropic splat: seen 2e-3 with P = 1)
                check_against_oracle(scene, settings_dict(**sd), backward=True, grad_tol=1e-2)
            else:  # (flip_grad_tol: with opacities down to 0.004 a single flipped blend is a visible part of a Gaussian's gradient)
                # (a 4x4-culling decision on the threshold moves a whole sub-tile: accepted by check_against_oracle only when the
                # oracle shows an entry of that tile within 6e-7 (relative) of the 1/255 threshold -- no blanket allowance)
                check_against_oracle(scene, settings_dict(**sd), backward=True, flip_grad_tol=2e-2)
        except Exception as e:  # noqa: BLE001
            bad += 1
            tb = traceback.extract_tb(e.__traceback__)[-1]
            print(f"FAIL case {i}: scene={sc} settings={sd}: {type(e).__name__}: {str(e)[:200]} @ {tb.filename.split('/')[-1]}:{tb.lineno}", flush=True)
            if os.environ.get("FUZZ_TRACE"):
                print(f"   sh_degree={scene.sh_degree} scale_modifier={scene.scale_modifier}")
                print("   " + "\n   ".join(traceback.format_exc().splitlines()[-14:]), flush=True)
        done += 1
    import test_gpu_parity as tgp
    print(f"fuzz: {done} cases, {bad} failures, {time.time() - t0:.0f} s; frames with a blend decision on its threshold: {tgp.FLIP_STATS['flipped_frames']}, "
          f"of which the oracle with that threshold nudged reproduces the product to the plain tolerances: {tgp.FLIP_STATS['closed_by_nudge']}")
    sys.exit(1 if bad else 0)


main()
