"""GPU tests (-m gpu) at BASELINE.json's FULL sizes for C3 / C4 / C5 (C2 has its own in test_gpu_parity.py):

  * a window of tile rows of the full frame against the CPU oracle on the same window (all Gaussians preprocessed on
    both sides; only the window's tiles are binned, sorted and blended -- the oracle finishes in seconds): image,
    radii, num_rendered and gradients;
  * size-independent properties of the complete frame: checksum of the duplicate counts, prefix sums, sortedness of
    the keys on the sorted bit range, every Gaussian listed tiles_touched times, ranges tiling the list, finite outputs,
    forward determinism, backward linearity in dL/dimage.

C4's defining leg (tile rows sharded over 8 GPUs + RCCL gather) is covered separately (test_multi_gpu.py); here its
4K frame runs on one GPU.
"""
import numpy as np
import pytest

from helpers import FULL_STP, GpuRun, max_abs, oracle_run, psnr, settings_dict
from diff_gaussian_rasterization import scenes

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return max_abs(a, b) / max(float(np.max(np.abs(b))), 1e-30) if np.size(b) else 0.0


CONFIGS = {
    "C3": dict(sd=settings_dict(2, per_pixel=16), backward=True, rows=(30, 33)),
    "C4": dict(sd=settings_dict(**FULL_STP), backward=False, rows=(60, 66)),
    "C5": dict(sd=settings_dict(**FULL_STP), backward=True, rows=(30, 33)),
}
_scene_cache = {}


def scene_of(name):
    if name not in _scene_cache:
        _scene_cache.clear()             # one multi-GB scene at a time
        _scene_cache[name] = scenes.config(name)
    return _scene_cache[name]


@pytest.mark.parametrize("name", ["C3", "C4", "C5"])
def test_full_size_tile_rows_against_oracle(name):
    c = CONFIGS[name]
    sc, sd, rows, backward = scene_of(name), c["sd"], c["rows"], c["backward"]
    g = GpuRun(sc, sd, backward=backward, tile_rows=rows)
    f, og = oracle_run(sc, sd, backward=backward, tile_rows=rows)
    assert g.num_rendered == f.num_rendered and np.array_equal(g.radii, f.radii)
    assert np.array_equal(g.binning_array("point_list"), f.array("point_list"))
    sl = slice(rows[0] * 16, min(rows[1] * 16, sc.H))
    d = np.abs(g.color[:, sl].astype(np.float64) - f.color[:, sl].astype(np.float64))
    # 1000+ blends per pixel in C3 / C5: fp32 rounding accumulates (3e-6).  A pixel beyond that moved because a decision of the blend loop sits on its
    # threshold -- it has to be EXPLAINED (oracle/explain.py, as in test_gpu_parity.check_against_oracle: no allowance by count), and the window then has to
    # match the oracle re-run with that decision taken the other way at the plain tolerances; only a window that mixes several decisions falls back to 2e-3
    IMG_TOL = 4e-6
    assert d.max() <= 1.0 / 255.0 + 1e-6
    moved = np.zeros((sc.H, sc.W), bool)
    moved[sl] = (d > IMG_TOL).any(axis=0)
    flipped = int(moved.sum())
    closed = False
    if flipped:
        from oracle import explain
        from test_gpu_parity import _matches_a_nudged_oracle
        ex = explain.explain_moved_pixels(moved, W=sc.W, H=sc.H, ranges=f.array("ranges").reshape(-1), point_list=f.array("point_list"),
                                          conic_opacity=f.array("conic_opacity").reshape(-1), means2D=f.array("means2D").reshape(-1),
                                          final_T_a=g.image_array("final_T").reshape(-1), final_T_b=f.array("final_T").reshape(-1),
                                          cull_4x4=bool(sd["culling_settings"]["hierarchical_4x4_culling"]) and sd["sort_settings"]["sort_mode"] == 3)
        assert not ex["unexplained"], (flipped, ex["by"], ex["unexplained"][:5])
        closed = _matches_a_nudged_oracle(sc, sd, g, backward, IMG_TOL, 1e-4, ex, tile_rows=rows, max_tries=7)
    assert psnr(g.color[:, sl], f.color[:, sl]) >= (100.0 if flipped == 0 else 75.0)
    if backward and not closed:
        tol = 1e-4 if flipped == 0 else 2e-3
        for k in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D"):
            a, b = g.grads[k], og[k]
            if k == "dL_dmeans2D":
                a, b = a[:, :2], b[:, :2]
            assert _rel(a, b) < tol, k


@pytest.mark.parametrize("name", ["C3", "C4", "C5"])
def test_full_size_properties(name):
    c = CONFIGS[name]
    sc, sd, backward = scene_of(name), c["sd"], c["backward"]
    g = GpuRun(sc, sd, backward=backward)
    R = g.num_rendered
    tiles = g.geom_array("tiles_touched").view(np.uint32)
    assert int(tiles.astype(np.int64).sum()) == R
    assert np.array_equal(np.cumsum(tiles.astype(np.int64)).astype(np.uint32), g.geom_array("point_offsets").view(np.uint32))
    keys = g.binning_array("keys")
    T = ((sc.W + 15) // 16) * ((sc.H + 15) // 16)
    bit = int(np.ceil(np.log2(T + 1)))
    masked = keys & np.uint64((1 << (32 + bit)) - 1)
    assert np.all(masked[1:] >= masked[:-1])
    ids = g.binning_array("point_list")
    vis_ids = np.nonzero(tiles)[0]
    assert np.array_equal(np.bincount(ids[ids != 0xFFFFFFFF].astype(np.int64), minlength=sc.P)[vis_ids], tiles[vis_ids])
    ranges = g.image_array("ranges").view(np.uint32).reshape(-1, 2)[:T]
    assert int((ranges[:, 1] - ranges[:, 0]).astype(np.int64).sum()) == int(((keys >> np.uint64(32)) < T).sum())
    fT = g.image_array("final_T")
    assert np.all(fT >= 0) and np.all(fT <= 1) and np.isfinite(g.color).all()
    g2 = GpuRun(sc, sd, backward=False)
    assert np.array_equal(g.color, g2.color)
    if backward:
        assert all(np.isfinite(v).all() for v in g.grads.values() if v is not None)
        keep = sc.dL_dout
        try:
            sc.dL_dout = keep * np.float32(2.0)
            g3 = GpuRun(sc, sd, backward=True)
        finally:
            sc.dL_dout = keep
        assert _rel(g3.grads["dL_dmeans3D"], 2.0 * g.grads["dL_dmeans3D"]) < 1e-4
        assert _rel(g3.grads["dL_dsh"], 2.0 * g.grads["dL_dsh"]) < 1e-4
