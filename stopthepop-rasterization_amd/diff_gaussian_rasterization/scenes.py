"""Deterministic synthetic scenes for the parity tests and bench.py.

Pure numpy, counter-based hash RNG (splitmix64 of (seed, stream, index)), so the same arrays come
out on every machine and numpy version.  Nothing here touches the GPU; callers upload the arrays.

The five BASELINE.json configurations are available through ``config(name)``.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

_MASK = (1 << 64) - 1


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """n float64 uniforms in [0,1), a pure function of (seed, stream, index)."""
    with np.errstate(over="ignore"):
        base = np.uint64(((seed * 0x9E3779B97F4A7C15) ^ (stream * 0xD1B54A32D192ED03)) & _MASK)
        idx = np.arange(n, dtype=np.uint64)
        bits = _splitmix64(_splitmix64(idx + base) ^ np.uint64(stream & _MASK))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def normal(seed: int, stream: int, n: int) -> np.ndarray:
    """Box-Muller on two uniform streams."""
    u1 = np.maximum(uniform(seed, 2 * stream + 1000, n), 1e-300)
    u2 = uniform(seed, 2 * stream + 1001, n)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)


def _each(fn, items, n):
    """fn(item) for every item; large scenes on a thread pool (numpy's element-wise loops release the GIL: the 48 SH streams of a 6M-Gaussian
    scene take a minute on one core)."""
    items = list(items)
    workers = min(len(items), os.cpu_count() or 1, 16)
    if n < 200_000 or workers < 2:
        for it in items:
            fn(it)
        return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(fn, items))


@dataclass
class Scene:
    """Host-side (numpy float32) inputs of one frame, in the reference's tensor conventions."""
    W: int
    H: int
    tanfovx: float
    tanfovy: float
    bg: np.ndarray
    viewmatrix: np.ndarray        # (4,4) row-vector convention (transposed), as 3DGS passes it
    projmatrix: np.ndarray        # (4,4) full view-projection, same convention
    inv_viewprojmatrix: np.ndarray
    campos: np.ndarray
    means3D: np.ndarray           # (P,3)
    scales: np.ndarray            # (P,3)
    rotations: np.ndarray         # (P,4) normalised (r,x,y,z)
    opacities: np.ndarray         # (P,1)
    shs: Optional[np.ndarray]     # (P,16,3) or None
    colors_precomp: Optional[np.ndarray]  # (P,3) or None
    sh_degree: int = 3
    scale_modifier: float = 1.0
    dL_dout: Optional[np.ndarray] = None  # (3,H,W) fixed N(0,1) image: loss = sum(w * img)
    meta: Dict = field(default_factory=dict)

    @property
    def P(self) -> int:
        return int(self.means3D.shape[0])


def perspective(tanfovx: float, tanfovy: float, znear: float = 0.01, zfar: float = 100.0) -> np.ndarray:
    """3DGS projection matrix in the row-vector layout (p_hom = [x,y,z,1] @ P)."""
    P = np.zeros((4, 4), dtype=np.float64)
    P[0, 0] = 1.0 / tanfovx
    P[1, 1] = 1.0 / tanfovy
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = 1.0
    P[3, 2] = -(zfar * znear) / (zfar - znear)
    return P


def look_at_view(eye, target, up=(0.0, -1.0, 0.0)) -> np.ndarray:
    """World->view matrix (row-vector layout) of a camera at `eye` looking at `target` (+z forward)."""
    eye = np.asarray(eye, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(np.asarray(up, dtype=np.float64), fwd)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    Rm = np.stack([right, down, fwd], axis=0)      # rows = camera axes in world coordinates
    V = np.eye(4)
    V[:3, :3] = Rm.T                                # row-vector layout: p_view = p_world @ V[:3,:3] + t
    V[3, :3] = -(Rm @ eye)
    return V


def make_scene(P: int, W: int, H: int, sigma_min: float, sigma_max: float, seed: int,
               use_sh: bool = True, camera: str = "origin", z_range=(2.0, 12.0),
               opacity_range=(0.05, 0.6), with_grad_image: bool = True, clusters=None) -> Scene:
    """clusters = (count, fraction, spread): `fraction` of the Gaussians are gathered around `count` random screen positions
    with a normal spread of `spread` x the frame size (a lumpy scene: tile lists and per-pixel blend counts far above the
    homogeneous average in the clusters, empty sky between them); None = homogeneous."""
    tanfovy = 0.5
    tanfovx = 0.5 * W / H
    focal_x = W / (2.0 * tanfovx)
    proj = perspective(tanfovx, tanfovy)

    u = uniform(seed, 1, P) * 2.0 - 1.0
    v = uniform(seed, 2, P) * 2.0 - 1.0
    if clusters is not None:
        n_c, frac, spread = int(clusters[0]), float(clusters[1]), float(clusters[2])
        member = uniform(seed, 30, P) < frac
        which = np.minimum((uniform(seed, 31, P) * n_c).astype(np.int64), n_c - 1)
        cu = uniform(seed, 32, n_c) * 1.6 - 0.8
        cv = uniform(seed, 33, n_c) * 1.6 - 0.8
        u = np.where(member, cu[which] + 2.0 * spread * normal(seed, 34, P), u)
        v = np.where(member, cv[which] + 2.0 * spread * normal(seed, 35, P), v)
    z = z_range[0] + (z_range[1] - z_range[0]) * uniform(seed, 3, P)
    x = u * 1.05 * tanfovx * z
    y = v * 1.05 * tanfovy * z
    cam_pts = np.stack([x, y, z], axis=1)           # positions in camera space

    sig = np.exp(math.log(sigma_min) + (math.log(sigma_max) - math.log(sigma_min)) * uniform(seed, 4, P))
    base = sig * z / focal_x
    scales = np.stack([base * (0.3 + 0.7 * uniform(seed, 5 + k, P)) for k in range(3)], axis=1)
    q = np.stack([normal(seed, 10 + k, P) for k in range(4)], axis=1)
    q /= np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)
    opac = opacity_range[0] + (opacity_range[1] - opacity_range[0]) * uniform(seed, 20, P)

    if camera == "origin":
        view = np.eye(4)
        campos = np.zeros(3)
        world = cam_pts
    elif camera == "orbit":
        # camera off-axis, rotated: exercises the general matrix paths.  World points are the
        # camera-space points mapped back through the inverse view transform.
        eye = np.array([1.3, -0.7, -2.1])
        view = look_at_view(eye, target=(0.2, 0.1, 6.0))
        Rv = view[:3, :3]
        world = (cam_pts - view[3, :3]) @ np.linalg.inv(Rv)
        campos = eye
    else:
        raise ValueError(camera)

    full = view @ proj
    inv_full = np.linalg.inv(full)

    shs = None
    colors = None
    if use_sh:
        M = 16
        shs = np.empty((P, M, 3), dtype=np.float64)

        def fill(kc):   # (streams are pure functions of (seed, stream, index): the order they are evaluated in changes nothing)
            k, ch = divmod(kc, 3)
            shs[:, k, ch] = (0.5 if k == 0 else 0.1) * normal(seed, 100 + 3 * k + ch, P)
        _each(fill, range(3 * M), P)
    else:
        colors = np.stack([uniform(seed, 200 + ch, P) for ch in range(3)], axis=1)

    dL = None
    if with_grad_image:
        dL = normal(seed, 300, 3 * H * W).reshape(3, H, W)

    f32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
    return Scene(W=W, H=H, tanfovx=tanfovx, tanfovy=tanfovy, bg=f32(np.array([0.1, 0.2, 0.3])),
                 viewmatrix=f32(view), projmatrix=f32(full), inv_viewprojmatrix=f32(inv_full), campos=f32(campos),
                 means3D=f32(world), scales=f32(scales), rotations=f32(q), opacities=f32(opac.reshape(P, 1)),
                 shs=f32(shs), colors_precomp=f32(colors), sh_degree=3 if use_sh else 0, dL_dout=f32(dL),
                 meta=dict(seed=seed, sigma=(sigma_min, sigma_max), camera=camera))


def covariance_from_scale_rotation(scene: "Scene") -> np.ndarray:
    """(P,6) float32 world-space covariances [xx, xy, xz, yy, yz, zz] = R diag((mod*s)^2) R^T of the scene's scales and
    rotations, evaluated in float64 and rounded: a valid `cov3D_precomp` INPUT for the same Gaussians (ref:
    forward_common.h:149-183 computes the same matrix in fp32; as an input, bit equality with that is not needed)."""
    q = scene.rotations.astype(np.float64)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                  2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                  2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], axis=1).reshape(-1, 3, 3)
    s2 = (scene.scale_modifier * scene.scales.astype(np.float64)) ** 2
    S = np.einsum("pij,pj,pkj->pik", R, s2, R)
    return np.ascontiguousarray(np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], axis=1),
                                dtype=np.float32)


# BASELINE.json configs made concrete
_CONFIGS = {
    "C1": dict(P=1_000, W=256, H=256, sigma_min=1.0, sigma_max=12.0, seed=1),
    "C2": dict(P=1_000_000, W=1920, H=1080, sigma_min=0.7, sigma_max=7.0, seed=2),
    "C3": dict(P=3_000_000, W=1920, H=1080, sigma_min=0.5, sigma_max=5.0, seed=3),
    "C4": dict(P=1_000_000, W=3840, H=2160, sigma_min=1.0, sigma_max=12.0, seed=4),
    "C5": dict(P=6_000_000, W=1600, H=1063, sigma_min=0.5, sigma_max=5.0, seed=5),
    # not a BASELINE config: LARGE splats (pixel sigma up to 200, i.e. rectangles of hundreds to thousands of tiles per
    # Gaussian) -- the regime the reference's warp-cooperative tile loops exist for (stopthepop_common.cuh:207-259, 510-621)
    # not a BASELINE config: C2's Gaussians, 40 % of them gathered in 12 clusters (what a trained scene looks like more than
    # the homogeneous C2 does): long tile lists, pixels with hundreds of blends, blend-log overflow
    "C2L": dict(P=1_000_000, W=1920, H=1080, sigma_min=0.7, sigma_max=7.0, seed=2, clusters=(12, 0.4, 0.03)),
    # ... and a heavier one (round 6): 60 % in 6 tight clusters -- tile lists beyond the LDS sort's 4096 entries, blend-log overflow on the clusters' tiles
    "C2H": dict(P=1_000_000, W=1920, H=1080, sigma_min=0.7, sigma_max=7.0, seed=2, clusters=(6, 0.6, 0.012)),
    "L1": dict(P=20_000, W=1920, H=1080, sigma_min=10.0, sigma_max=200.0, seed=6, opacity_range=(0.02, 0.3)),
}


def config(name: str, scale: float = 1.0, **overrides) -> Scene:
    """Scene for BASELINE config `name`.  scale<1 shrinks the Gaussian count and the image area
    together (same density per tile) -- used by parity tests that must finish in seconds."""
    c = dict(_CONFIGS[name])
    if scale != 1.0:
        lin = math.sqrt(scale)
        c["P"] = max(1, int(round(c["P"] * scale)))
        c["W"] = max(16, int(round(c["W"] * lin)))
        c["H"] = max(16, int(round(c["H"] * lin)))
    c.update(overrides)
    return make_scene(**c)
