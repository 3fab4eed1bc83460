"""CPU tests (-m "not gpu") of the oracle itself: the independent float64 torch reference pins its value and
gradient maths, cross-mode consistency pins the resorting machinery, golden fixtures pin it against
regressions.  These pins are independent of the reference; the pin to the reference itself (its own sources compiled
for gfx950) is tests/test_reference_golden.py."""
import os

import numpy as np
import pytest

from helpers import FULL_STP, max_abs, psnr, settings_dict
from diff_gaussian_rasterization import scenes
from oracle import oracle as orc
import torch_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rel(a, b):
    return max_abs(a, b) / max(float(np.max(np.abs(b))), 1e-30)


@pytest.mark.parametrize("camera", ["origin", "orbit"])
@pytest.mark.parametrize("use_sh", [True, False])
@pytest.mark.parametrize("mode,order", [(0, "global"), (2, "exact"), (3, "exact")])
def test_oracle_matches_float64_autograd(camera, use_sh, mode, order):
    """Image and every gradient of the fp32 oracle agree with an independently written float64 autograd
    renderer (tolerance: fp32 round-off, 5e-5 relative to the largest gradient entry)."""
    sc = scenes.make_scene(P=150, W=40, H=36, sigma_min=1.0, sigma_max=8.0, seed=7, use_sh=use_sh, camera=camera)
    f = orc.forward_scene(sc, settings_dict(mode, per_pixel=16 if mode == 2 else 4))
    g = f.backward(sc.dL_dout)
    img, tg = torch_ref.loss_and_grads(sc, order=order)
    assert max_abs(f.color, img) < 2e-6
    assert _rel(g["dL_dmeans3D"], tg["means3D"]) < 5e-5
    assert _rel(g["dL_dopacity"], tg["opacities"]) < 5e-5
    assert _rel(g["dL_dscales"], tg["scales"]) < 5e-5
    assert _rel(g["dL_drotations"], tg["rotations"]) < 5e-5
    assert _rel(g["dL_dmeans2D"][:, :2], tg["means2D"]) < 5e-5
    if use_sh:
        assert _rel(g["dL_dsh"], tg["shs"]) < 5e-5
    else:
        assert _rel(g["dL_dcolors"], tg["colors_precomp"]) < 5e-5


def test_oracle_ewa_scaling_gradient_quirk():
    """proper_ewa_scaling: forward + dL/dopacity match autograd; the covariance-path gradients match only
    with the test switch that evaluates the reference's closed form with undilated entries (the reference,
    backward.cu:227-234, plugs in the dilated ones; the oracle and the kernels reproduce the reference)."""
    sc = scenes.make_scene(P=150, W=40, H=36, sigma_min=1.0, sigma_max=8.0, seed=7, camera="orbit")
    img, tg = torch_ref.loss_and_grads(sc, order="global", proper_ewa_scaling=True)
    f = orc.forward_scene(sc, settings_dict(0, ewa=True))
    g_ref_style = f.backward(sc.dL_dout)
    assert max_abs(f.color, img) < 2e-6
    assert _rel(g_ref_style["dL_dopacity"], tg["opacities"]) < 5e-5
    assert _rel(g_ref_style["dL_dscales"], tg["scales"]) > 1e-2  # the quirk is visible
    try:
        orc.set_flag("ewa_exact_grad", 1)
        g_exact = f.backward(sc.dL_dout)
    finally:
        orc.set_flag("ewa_exact_grad", 0)
    assert _rel(g_exact["dL_dscales"], tg["scales"]) < 5e-5
    assert _rel(g_exact["dL_drotations"], tg["rotations"]) < 5e-5
    assert _rel(g_exact["dL_dmeans3D"], tg["means3D"]) < 5e-5


def test_sorted_modes_agree_at_low_density():
    """At ~25 Gaussians per tile k-buffer(16), PPX_FULL and every hierarchical queue size give the exact
    per-pixel order, bit for bit (SURVEY.md section 6 observation)."""
    sc = scenes.config("C1")
    ref = orc.forward_scene(sc, settings_dict(1)).color
    for sd in (settings_dict(2, per_pixel=16), settings_dict(3), settings_dict(3, h44=True),
               settings_dict(3, per_pixel=8, tile_2x2=12), settings_dict(3, per_pixel=16, tile_2x2=20)):
        assert np.array_equal(orc.forward_scene(sc, sd).color, ref)
    assert psnr(orc.forward_scene(sc, settings_dict(0)).color, ref) < 80.0  # the global order does differ


def test_modes_differ_at_high_density_in_the_expected_ranking():
    """~700 per tile: hierarchical+culling and k-buffer(16) stay close to the exact sort, global is far."""
    sc = scenes.make_scene(P=6000, W=96, H=80, sigma_min=2.0, sigma_max=14.0, seed=11, camera="orbit")
    exact = orc.forward_scene(sc, settings_dict(1)).color
    p_global = psnr(orc.forward_scene(sc, settings_dict(0)).color, exact)
    p_hier = psnr(orc.forward_scene(sc, settings_dict(3, h44=True)).color, exact)
    p_kbuf = psnr(orc.forward_scene(sc, settings_dict(2, per_pixel=16)).color, exact)
    assert p_hier > p_global + 10 and p_kbuf > p_global + 10


@pytest.mark.parametrize("sd", [settings_dict(3), settings_dict(3, h44=True), settings_dict(**FULL_STP), settings_dict(3, per_pixel=8, tile_2x2=12),
                                settings_dict(3, per_pixel=16, tile_2x2=20, h44=True), settings_dict(2, per_pixel=1), settings_dict(2, per_pixel=4),
                                settings_dict(2, per_pixel=16), settings_dict(2, per_pixel=24)],
                         ids=["hier", "hier_cull", "full_stp", "hier_8_12", "hier_16_20_cull", "kbuf1", "kbuf4", "kbuf16", "kbuf24"])
def test_failing_candidates_are_no_ops_for_the_blend_order(sd):
    """The claim the wave64 render kernels rest on (stp_render_hier.inc "filter_push", stp_render_kbuf.hip): the reference
    pops a full head queue / k-buffer window BEFORE it looks at a candidate (ref: hierarchical_render.cuh:455-458,
    resorted_render.cuh:17-221), but a candidate that then FAILS its tests might as well never have been shown to the pixel --
    the next passing candidate pops the same front entry before it is inserted.  The oracle's test switch "lazy_pop" walks the
    lists that way; image, final_T and every gradient must be bit-identical to the reference walk, in dense scenes where
    most pixels saturate (in the list and in the final drain) as well as in sparse ones."""
    for kw in (dict(P=6000, W=96, H=80, sigma_min=2.0, sigma_max=14.0, seed=11, camera="orbit"),
               dict(P=1500, W=112, H=64, sigma_min=1.0, sigma_max=6.0, seed=23),
               dict(P=9000, W=64, H=48, sigma_min=3.0, sigma_max=20.0, seed=5, camera="orbit", opacity_range=(0.01, 0.08))):
        sc = scenes.make_scene(**kw)
        a = orc.forward_scene(sc, sd)
        ga = a.backward(sc.dL_dout)
        fa_T, fa_c = a.array("final_T").copy(), a.color.copy()
        orc.set_flag("lazy_pop", 1)
        try:
            b = orc.forward_scene(sc, sd)
            gb = b.backward(sc.dL_dout)
        finally:
            orc.set_flag("lazy_pop", 0)
        assert np.array_equal(fa_c.view(np.uint32), b.color.view(np.uint32))
        assert np.array_equal(fa_T.view(np.uint32), b.array("final_T").view(np.uint32))
        for k in ga:
            if ga[k] is not None and np.size(ga[k]):
                assert np.array_equal(np.ascontiguousarray(ga[k]).view(np.uint8), np.ascontiguousarray(gb[k]).view(np.uint8)), k


def test_load_balancing_flag_changes_nothing():
    sc = scenes.make_scene(P=800, W=128, H=96, sigma_min=2.0, sigma_max=30.0, seed=3)
    a = orc.forward_scene(sc, settings_dict(**{**FULL_STP, "lb": False}))
    b = orc.forward_scene(sc, settings_dict(**FULL_STP))
    assert a.num_rendered == b.num_rendered and np.array_equal(a.color, b.color)


def test_tile_based_culling_only_removes_non_contributors():
    """TBC shrinks the duplicate list but (with a global order) never the image."""
    sc = scenes.make_scene(P=2000, W=128, H=96, sigma_min=1.0, sigma_max=12.0, seed=9, camera="orbit")
    a = orc.forward_scene(sc, settings_dict(0))
    b = orc.forward_scene(sc, settings_dict(0, tbc=True, rect=True))
    assert b.num_rendered < a.num_rendered
    assert psnr(a.color, b.color) > 70.0


def test_edge_cases():
    sc = scenes.make_scene(P=50, W=40, H=24, sigma_min=1.0, sigma_max=4.0, seed=2)
    # all Gaussians behind the near plane: empty lists, background image, zero gradients
    behind = scenes.make_scene(P=50, W=40, H=24, sigma_min=1.0, sigma_max=4.0, seed=2)
    behind.means3D[:, 2] = 0.1
    for mode in (0, 2, 3):
        f = orc.forward_scene(behind, settings_dict(mode, per_pixel=16 if mode == 2 else 4))
        assert f.num_rendered == 0 and not f.radii.any()
        assert np.allclose(f.color, np.asarray(behind.bg).reshape(3, 1, 1))
        g = f.backward(behind.dL_dout)
        assert all(not np.any(v) for v in g.values())
    # image size that is not a multiple of the tile size, single Gaussian, unsupported queue sizes
    odd = scenes.make_scene(P=1, W=37, H=19, sigma_min=3.0, sigma_max=3.1, seed=4)
    for mode in (0, 1, 2, 3):
        assert orc.forward_scene(odd, settings_dict(mode)).color.shape == (3, 19, 37)
    with pytest.raises(RuntimeError):
        orc.forward_scene(sc, settings_dict(3, per_pixel=5))
    with pytest.raises(RuntimeError):
        orc.forward_scene(sc, settings_dict(3, tile_2x2=16))
    with pytest.raises(RuntimeError):
        orc.forward_scene(sc, settings_dict(1)).backward(sc.dL_dout)  # PPX_FULL has no backward


def test_mark_visible():
    sc = scenes.make_scene(P=100, W=32, H=32, sigma_min=1.0, sigma_max=4.0, seed=6, camera="orbit")
    sc.means3D[::3] = sc.campos  # at the camera: view z = 0 <= 0.2
    vis = orc.mark_visible(sc.means3D, sc.viewmatrix, sc.projmatrix)
    assert not vis[::3].any() and vis[1::3].all()


def test_tile_row_windows_tile_the_frame():
    """Rendering disjoint tile-row windows and pasting the strips reproduces the full frame exactly; radii
    are those of the full frame in every window; partial gradients of the render half add up."""
    sc = scenes.make_scene(P=1500, W=96, H=112, sigma_min=1.5, sigma_max=14.0, seed=8, camera="orbit")
    sd = settings_dict(**FULL_STP)
    full = orc.forward_scene(sc, sd)
    gfull = full.backward(sc.dL_dout)
    img = np.zeros_like(full.color)
    acc = None
    for rows in ((0, 3), (3, 5), (5, 7)):
        part = orc.forward_scene(sc, sd, tile_rows=rows)
        assert np.array_equal(part.radii, full.radii)
        img[:, rows[0] * 16:rows[1] * 16] = part.color[:, rows[0] * 16:rows[1] * 16]
        g = part.backward(sc.dL_dout)
        acc = g if acc is None else {k: acc[k] + g[k] for k in g}
    assert np.array_equal(img, full.color)
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dcolors"):
        assert _rel(acc[k], gfull[k]) < 1e-5


@pytest.mark.parametrize("name", ["c1_global", "dense_hier_full", "dense_kbuffer"])
def test_golden_fixtures(name):
    """Regression fixtures: oracle outputs captured by tests/golden/make_golden.py (oracle-generated, NOT
    reference-generated -- the reference cannot run here)."""
    from golden.make_golden import CASES, run_case
    ref = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = run_case(CASES[name])
    assert int(ref["num_rendered"]) == out["num_rendered"]
    assert np.array_equal(ref["radii"], out["radii"])
    assert max_abs(ref["color"], out["color"]) < 1e-6
    for k in ("dL_dmeans3D", "dL_dopacity", "dL_dscales"):
        assert _rel(out[k], ref[k]) < 1e-5


def test_render_depth_visualisation_oracle():
    """DebugVisualization::Depth (`render_depth=True`): an RGB image from the Turbo table, identical for the exact
    per-pixel sorts at low density, and with the frame's uncovered pixels at the top of the colormap."""
    sc = scenes.make_scene(P=25, W=64, H=48, sigma_min=1.5, sigma_max=4.0, seed=17, opacity_range=(0.5, 0.95))
    imgs = {m: orc.forward_scene(sc, settings_dict(m, per_pixel=16), render_depth=True).color for m in (0, 1, 2, 3)}
    for img in imgs.values():
        assert img.shape == (3, sc.H, sc.W) and img.min() >= 0.0 and img.max() <= 1.0
    assert np.array_equal(imgs[1], imgs[2])              # PPX_FULL == k-buffer(16) here
    assert np.max(np.abs(imgs[3] - imgs[1])) < 1e-4       # hierarchical: same order, another summation path
    empty = orc.forward_scene(sc, settings_dict(1)).array("final_T").reshape(sc.H, sc.W) == 1.0
    assert empty.any()
    top = np.array([0.47960, 0.01583, 0.01055], np.float32).reshape(3, 1)  # last entry of the Turbo table
    assert np.allclose(imgs[1][:, empty], top, atol=1e-6)


@pytest.mark.parametrize("mode", [0, 2, 3])
def test_blend_nudge_moves_only_the_blend_decisions_and_resets(mode):
    """The checker's knob (oracle.blend_nudge, used by check_against_oracle for frames with a decision on its threshold): a large nudge changes the image,
    never the state in front of the blend loop; a nudge far below the distance of any decision from its threshold changes nothing; leaving the block restores
    the reference's constants."""
    sc = scenes.make_scene(P=1500, W=96, H=64, sigma_min=1.5, sigma_max=9.0, seed=21, camera="orbit", opacity_range=(0.004, 0.3))
    sd = settings_dict(mode, per_pixel=4 if mode == 3 else 8)
    base = orc.forward_scene(sc, sd)
    gb = base.backward(sc.dL_dout)
    with orc.blend_nudge(alpha=2e-3):
        big = orc.forward_scene(sc, sd)
    assert big.num_rendered == base.num_rendered
    for nm in ("keys", "point_list", "ranges", "conic_opacity", "means2D"):
        assert np.array_equal(big.array(nm), base.array(nm)), nm
    assert not np.array_equal(big.color, base.color)       # alphas between 1/255 and 1/255 + 2e-3 no longer blend
    assert float(big.color.sum()) < float(base.color.sum())  # (black background: fewer blends, less light)
    with orc.blend_nudge(T=0.5):
        early = orc.forward_scene(sc, sd)                   # pixels stop at half transmittance
    assert float(early.array("final_T").min()) >= 0.5 - 1e-6
    with orc.blend_nudge(alpha=1e-12, T=1e-12):
        tiny = orc.forward_scene(sc, sd)                    # (1/255 + 1e-12 == 1/255 in fp32)
    assert np.array_equal(tiny.color, base.color)
    again = orc.forward_scene(sc, sd)
    ga = again.backward(sc.dL_dout)
    assert np.array_equal(again.color, base.color)
    for k in gb:
        assert np.array_equal(ga[k], gb[k]), k


def test_forced_alpha_flip_changes_one_pixel_and_resets():
    """oracle.forced_alpha_flips: ONE per-pixel alpha test taken the other way moves that pixel and no other; a Gaussian's gradient changes only if it is
    blended there; leaving the block restores the oracle."""
    from oracle import explain
    sc = scenes.make_scene(P=600, W=64, H=48, sigma_min=2.0, sigma_max=8.0, seed=5, camera="orbit")
    sd = settings_dict(0)
    base = orc.forward_scene(sc, sd)
    gb = base.backward(sc.dL_dout)
    x, y = 30, 20
    ids = explain._entries(base.array("ranges").reshape(-1), base.array("point_list"), (sc.W + 15) // 16, x, y)
    a = explain.pixel_alphas(ids, base.array("conic_opacity").reshape(-1), base.array("means2D").reshape(-1), x, y)
    blended = ids[a > 0.05]
    assert blended.size, "pick another pixel: nothing blends visibly here"
    gid = int(blended[0])
    with orc.forced_alpha_flips([(x, y, gid)], sc.W):
        f2 = orc.forward_scene(sc, sd)
        g2 = f2.backward(sc.dL_dout)
    d = np.abs(f2.color - base.color).max(axis=0)
    assert d[y, x] > 1e-4
    d[y, x] = 0.0
    assert d.max() == 0.0
    changed = np.nonzero(np.abs(g2["dL_dopacity"] - gb["dL_dopacity"]).reshape(sc.P, -1).max(axis=1) > 0)[0]
    assert set(changed.tolist()) <= set(ids.tolist()) and gid in changed
    again = orc.forward_scene(sc, sd)
    assert np.array_equal(again.color, base.color)


def test_forced_T_flip_touches_only_a_step_on_the_threshold():
    """The transmittance side of oracle.forced_alpha_flips: a pixel named in T_pixels changes only if one of its steps has test_T within 2e-6 (relative) of 1e-4 --
    naming pixels that are nowhere near saturation changes nothing; the state is restored afterwards."""
    sc = scenes.make_scene(P=600, W=64, H=48, sigma_min=2.0, sigma_max=8.0, seed=5, camera="orbit")
    sd = settings_dict(3)
    base = orc.forward_scene(sc, sd)
    with orc.forced_alpha_flips([], sc.W, T_pixels=[(x, y) for y in range(0, sc.H, 3) for x in range(0, sc.W, 3)]):
        f2 = orc.forward_scene(sc, sd)
    assert np.array_equal(f2.color, base.color)   # (no pixel of this scene ends within 2e-6 of the threshold)
    assert np.array_equal(orc.forward_scene(sc, sd).color, base.color)
