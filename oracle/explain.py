"""Why does a pixel differ?  TEST INFRASTRUCTURE ONLY (tests/, bench.py's checker legs): never imported by the product.

The product and its checkers (the CPU oracle, the reference's own kernels) agree bit for bit in every integer / index result and in the
per-Gaussian state; what is left are a handful of pixels per frame that move by up to 1/255.  The claim has always been: a DISCRETE decision
of the blend loop sits exactly on its threshold and two exponential implementations (the device's exp_blend, libm's expf, ocml's) land on
different sides.  This module makes the claim checkable per pixel, from arrays both sides share bit for bit (tile ranges, sorted list,
conic + opacity, 2D means):

  alpha threshold   an entry of the pixel's tile list whose alpha = opacity * exp(power) -- power in fp32 exactly as the kernels evaluate it
                    (reference forward.cu:306-318 / hierarchical_render.cuh:491-506), the exponential in DOUBLE -- lies within ALPHA_TOL
                    (relative) of 1/255: the entry is blended by one side and skipped by the other;
  sub-tile culling  (hierarchical_4x4_culling) the same for the culling test's alpha at the sub-tile's point of maximum contribution
                    (reference hierarchical_render.cuh:722-743): one decision removes the entry from 16 pixels;
  T threshold       one side's final transmittance lies within T_TOL (relative) of 1e-4: `T (1 - alpha) < 1e-4` stopped one side one entry
                    earlier (reference forward.cu:319-324).

ALPHA_TOL = 6e-7: 2 + 1 + 1/2 ulp of fp32 for the two exponentials and the product (the figure tests/test_gpu_parity.py has used for the
sub-tile flips since round 2).  T_TOL = 1e-6: the transmittance is a product of up to a few hundred (1 - alpha) factors.
"""
from __future__ import annotations

import numpy as np

ALPHA_TOL = 6e-7
T_TOL = 1e-6
_THR = 1.0 / 255.0
f32 = np.float32


def _entries(ranges, point_list, gx, px, py):
    t = (py // 16) * gx + (px // 16)
    lo, hi = int(ranges[2 * t]), int(ranges[2 * t + 1])
    return point_list[lo:hi].astype(np.int64)


def pixel_alphas(ids, conic_opacity, means2D, px, py):
    """alpha of every entry `ids` at pixel (px, py): fp32 exponent as in the kernels (no contraction), exponential in double."""
    co = conic_opacity.reshape(-1, 4)[ids].astype(f32)
    xy = means2D.reshape(-1, 2)[ids].astype(f32)
    dx = xy[:, 0] - f32(px)
    dy = xy[:, 1] - f32(py)
    power = -(f32(0.5) * (co[:, 0] * dx * dx + co[:, 2] * dy * dy) + co[:, 1] * dx * dy)   # every operation rounds to fp32 on its own
    a = co[:, 3].astype(np.float64) * np.exp(power.astype(np.float64))
    a[power > 0] = 0.0
    return a


def _max_contrib_power_rect(co, xy, rmin, rmax, patch):
    """numpy restatement of the culling test's point of maximum contribution (reference stopthepop_common.cuh:130-174), fp32 step by step;
    co (n, 4), xy (n, 2); returns the fp32 `power` (positive form)."""
    one, zero = f32(1.0), f32(0.0)
    x_min_diff = f32(rmin[0]) - xy[:, 0]
    x_left = (x_min_diff > 0).astype(f32)
    not_in_x = x_left + (xy[:, 0] > f32(rmax[0])).astype(f32)
    y_min_diff = f32(rmin[1]) - xy[:, 1]
    y_above = (y_min_diff > 0).astype(f32)
    not_in_y = y_above + (xy[:, 1] > f32(rmax[1])).astype(f32)
    outside = (not_in_y + not_in_x) > 0
    pxs = x_left * f32(rmin[0]) + (one - x_left) * f32(rmax[0])
    pys = y_above * f32(rmin[1]) + (one - y_above) * f32(rmax[1])
    dxs = np.copysign(f32(patch), x_min_diff).astype(f32)
    dys = np.copysign(f32(patch), y_min_diff).astype(f32)
    diffx, diffy = xy[:, 0] - pxs, xy[:, 1] - pys
    with np.errstate(divide="ignore", invalid="ignore"):
        rcp_x = one / (f32(patch) * f32(patch) * co[:, 0])
        rcp_y = one / (f32(patch) * f32(patch) * co[:, 2])
        sat = lambda v: np.where(np.isnan(v), zero, np.minimum(np.maximum(v, zero), one)).astype(f32)
        tx = not_in_y * sat((dxs * co[:, 0] * diffx + dxs * co[:, 1] * diffy) * rcp_x)
        ty = not_in_x * sat((dys * co[:, 1] * diffx + dys * co[:, 2] * diffy) * rcp_y)
    mx, my = pxs + tx * dxs, pys + ty * dys
    ddx, ddy = xy[:, 0] - mx, xy[:, 1] - my
    power = f32(0.5) * (co[:, 0] * ddx * ddx + co[:, 2] * ddy * ddy) + co[:, 1] * ddx * ddy
    return np.where(outside, power, zero).astype(f32)


def subtile_cull_alphas(ids, conic_opacity, means2D, cx, cy):
    """opacity * exp(-power) of the 4x4-culling test of every entry `ids` for the sub-tile whose corner pixel is (cx, cy) (double exponential)."""
    co = conic_opacity.reshape(-1, 4)[ids].astype(f32)
    xy = means2D.reshape(-1, 2)[ids].astype(f32)
    power = _max_contrib_power_rect(co, xy, (cx, cy), (cx + 3, cy + 3), 3.0)
    return co[:, 3].astype(np.float64) * np.exp(-power.astype(np.float64))


def explain_moved_pixels(moved, *, W, H, ranges, point_list, conic_opacity, means2D, final_T_a=None, final_T_b=None, cull_4x4=False, limit=4096):
    """moved: (H, W) bool -- pixels that differ between two renderings of the same frame.  Every other array is shared by the two sides bit for
    bit (flat numpy arrays in the layouts of the oracle / the reference: ranges 2 per tile, conic_opacity 4 per Gaussian, means2D 2 per
    Gaussian); final_T_a / final_T_b: the two sides' final transmittance (H * W), when they are at hand.
    Returns {"pixels": n, "explained": m, "by": {...}, "unexplained": [(x, y), ...], "gaussians": ids blended at an explained pixel,
    "decisions": [(x, y, gaussian id), ...] the alpha test nearest 1/255 at every pixel explained by one, "T_pixels": [(x, y), ...] the pixels explained by their
    transmittance (oracle.forced_alpha_flips takes both)}."""
    gx = (W + 15) // 16
    ys, xs = np.nonzero(moved)
    out = {"pixels": int(ys.size), "explained": 0, "by": {"alpha_threshold": 0, "subtile_cull": 0, "T_threshold": 0}, "unexplained": [], "gaussians": set(),
           "decisions": [], "T_pixels": []}  # decisions: (x, y, gaussian id) of the per-pixel alpha test nearest its threshold at every pixel explained that way
    cull_cache = {}
    pix = list(zip(ys.tolist(), xs.tolist()))
    # pixels beyond `limit` are not examined: they count as UNEXPLAINED (a gate of the form "every moved pixel is explained" must not pass on a prefix)
    out["unexplained"].extend((x, y) for y, x in pix[limit:])
    for y, x in pix[:limit]:
        ids = _entries(ranges, point_list, gx, x, y)
        why = None
        if ids.size:
            a = pixel_alphas(ids, conic_opacity, means2D, x, y)
            near = np.abs(a * 255.0 - 1.0) <= ALPHA_TOL
            if near.any():
                why = "alpha_threshold"
                out["decisions"].append((int(x), int(y), int(ids[int(np.argmin(np.abs(a * 255.0 - 1.0)))])))
                out["gaussians"].update(ids[a >= _THR * (1.0 - ALPHA_TOL)].tolist())
            if why is None and cull_4x4:
                key = (x // 4, y // 4)
                if key not in cull_cache:
                    ca = subtile_cull_alphas(ids, conic_opacity, means2D, 4 * key[0], 4 * key[1])
                    cull_cache[key] = bool((np.abs(ca * 255.0 - 1.0) <= ALPHA_TOL).any())
                if cull_cache[key]:
                    why = "subtile_cull"
                    out["gaussians"].update(ids[a >= _THR * (1.0 - ALPHA_TOL)].tolist())
            if why is None:
                for ft in (final_T_a, final_T_b):
                    if ft is not None and abs(float(ft[y * W + x]) / 1e-4 - 1.0) <= T_TOL:
                        why = "T_threshold"
                        out["T_pixels"].append((int(x), int(y)))
                        out["gaussians"].update(ids[a >= _THR * (1.0 - ALPHA_TOL)].tolist())
                        break
        if why is None:
            out["unexplained"].append((x, y))
        else:
            out["explained"] += 1
            out["by"][why] += 1
    return out
