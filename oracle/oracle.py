"""ctypes front end of the CPU oracle (oracle/libstp_oracle.so).

TEST INFRASTRUCTURE ONLY -- see oracle/stp_oracle.h.  Imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg; never by the product package.
Pinned against the reference's own sources compiled for gfx950 (oracle/ref_build/, tests/golden/ref/): see stp_oracle.h.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libstp_oracle.so")
_lib = None


class OrcSettings(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "sort_mode", "sort_order", "queue_tile_4x4", "queue_tile_2x2", "queue_per_pixel",
        "rect_bounding", "tight_opacity_bounding", "tile_based_culling", "hierarchical_4x4_culling",
        "load_balancing", "proper_ewa_scaling", "tile_y0", "tile_y1", "debug_visualization")]


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("stp_oracle.cpp", "stp_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.c_void_p
        L.orc_forward.restype = ctypes.c_int
        L.orc_forward.argtypes = [ctypes.c_int] * 3 + [fp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(OrcSettings),
                                  fp, fp, fp, fp, fp, ctypes.c_float, fp, fp, fp, fp, fp, fp,
                                  ctypes.c_float, ctypes.c_float, ctypes.c_int, fp, fp, ctypes.POINTER(ctypes.c_void_p)]
        L.orc_backward.restype = ctypes.c_int
        L.orc_backward.argtypes = [fp, fp, fp, fp, fp, fp, fp, ctypes.c_float, fp, fp, fp, fp, fp, fp,
                                   ctypes.c_float, ctypes.c_float, fp, fp] + [fp] * 9
        L.orc_mark_visible.restype = None
        L.orc_mark_visible.argtypes = [ctypes.c_int, fp, fp, fp, fp]
        L.orc_frame_free.restype = None
        L.orc_frame_free.argtypes = [fp]
        L.orc_frame_array.restype = ctypes.c_int64
        L.orc_frame_array.argtypes = [fp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
        L.orc_frame_num_rendered.restype = ctypes.c_int
        L.orc_frame_num_rendered.argtypes = [fp]
        L.orc_num_threads.restype = ctypes.c_int
        L.orc_set_num_threads.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def settings_struct(d: Optional[dict] = None, tile_rows=None) -> OrcSettings:
    """Build the POD settings from the reference's settings dict (ExtendedSettings.to_dict())."""
    d = d or {}
    ss = d.get("sort_settings", {})
    q = ss.get("queue_sizes", {})
    cs = d.get("culling_settings", {})
    s = OrcSettings()
    s.sort_mode = int(ss.get("sort_mode", 0))
    s.sort_order = int(ss.get("sort_order", 0))
    s.queue_tile_4x4 = int(q.get("tile_4x4", 64))
    s.queue_tile_2x2 = int(q.get("tile_2x2", 8))
    s.queue_per_pixel = int(q.get("per_pixel", 4))
    s.rect_bounding = int(bool(cs.get("rect_bounding", False)))
    s.tight_opacity_bounding = int(bool(cs.get("tight_opacity_bounding", False)))
    s.tile_based_culling = int(bool(cs.get("tile_based_culling", False)))
    s.hierarchical_4x4_culling = int(bool(cs.get("hierarchical_4x4_culling", False)))
    s.load_balancing = int(bool(d.get("load_balancing", False)))
    s.proper_ewa_scaling = int(bool(d.get("proper_ewa_scaling", False)))
    if tile_rows is not None:
        s.tile_y0, s.tile_y1 = int(tile_rows[0]), int(tile_rows[1])
    s.debug_visualization = 1 if d.get("_render_depth") else 0  # private key: the reference's `render_depth` flag
    return s


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


_ARRAY_TYPES = {
    "depths": np.float32, "clamped": np.uint8, "radii": np.int32, "rects2D": np.float32, "means2D": np.float32,
    "cov3D": np.float32, "cov3D_inv": np.float32, "conic_opacity": np.float32, "rgb": np.float32,
    "tiles_touched": np.uint32, "point_offsets": np.uint32, "keys_unsorted": np.uint64,
    "values_unsorted": np.uint32, "keys": np.uint64, "point_list": np.uint32, "ranges": np.uint32,
    "final_T": np.float32, "n_contrib": np.uint32,
}


class Frame:
    """Forward result + handle to the oracle's retained state."""

    def __init__(self, handle, color, radii, num_rendered, inputs):
        self._h = handle
        self.color = color
        self.radii = radii
        self.num_rendered = num_rendered
        self._in = inputs

    def array(self, name: str) -> np.ndarray:
        ptr = ctypes.c_void_p()
        n = lib().orc_frame_array(self._h, name.encode(), ctypes.byref(ptr))
        if n < 0:
            raise KeyError(name)
        dt = np.dtype(_ARRAY_TYPES[name])
        if n == 0:
            return np.zeros(0, dtype=dt)
        buf = (ctypes.c_char * (n * dt.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dt).copy()

    def cull_alpha(self, tile: int, cx: int, cy: int) -> np.ndarray:
        """opacity*exp(-power) of the 4x4-culling test (double) of every entry of `tile` vs the sub-tile at pixel (cx, cy)."""
        L = lib()
        L.orc_cull_alpha.restype = ctypes.c_int
        L.orc_cull_alpha.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        r = self.array("ranges").reshape(-1, 2)[tile]
        out = np.zeros(max(1, int(r[1] - r[0])), np.float64)
        n = L.orc_cull_alpha(self._h, int(tile), int(cx), int(cy), out.ctypes.data_as(ctypes.c_void_p), out.size)
        return out[:max(n, 0)]

    def backward(self, dL_dout: np.ndarray, pixel_colors: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
        i = self._in
        P, M = i["P"], i["M"]
        dL = _f32(dL_dout)
        pc = _f32(self.color if pixel_colors is None else pixel_colors)
        g = dict(dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 2, 2), np.float32),
                 dL_dopacity=np.zeros((P, 1), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
                 dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dcov3D=np.zeros((P, 6), np.float32),
                 dL_dsh=np.zeros((P, M, 3), np.float32), dL_dscales=np.zeros((P, 3), np.float32),
                 dL_drotations=np.zeros((P, 4), np.float32))
        rc = lib().orc_backward(self._h, _p(i["bg"]), _p(i["means3D"]), _p(i["shs"]), _p(i["colors"]), _p(i["opac"]),
                                _p(i["scales"]), ctypes.c_float(i["mod"]), _p(i["rots"]), _p(i["cov3D"]), _p(i["view"]),
                                _p(i["proj"]), _p(i["inv"]), _p(i["cam"]), ctypes.c_float(i["tfx"]), ctypes.c_float(i["tfy"]),
                                _p(pc), _p(dL), _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]),
                                _p(g["dL_dcolors"]), _p(g["dL_dmeans3D"]), _p(g["dL_dcov3D"]), _p(g["dL_dsh"]),
                                _p(g["dL_dscales"]), _p(g["dL_drotations"]))
        if rc != 0:
            raise RuntimeError(f"oracle backward failed with code {rc}")
        return g

    def free(self):
        if self._h:
            lib().orc_frame_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def forward(*, bg, means3D, opacities, viewmatrix, projmatrix, inv_viewprojmatrix, campos, tanfovx, tanfovy,
            W, H, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
            sh_degree=3, scale_modifier=1.0, settings: Optional[dict] = None, tile_rows=None,
            prefiltered=False) -> Frame:
    means3D = _f32(means3D)
    P = int(means3D.shape[0])
    shs_ = _f32(shs) if shs is not None and np.size(shs) else None
    M = int(shs_.shape[1]) if shs_ is not None else 0
    col = _f32(colors_precomp) if colors_precomp is not None and np.size(colors_precomp) else None
    sc = _f32(scales) if scales is not None and np.size(scales) else None
    ro = _f32(rotations) if rotations is not None and np.size(rotations) else None
    c3 = _f32(cov3D_precomp) if cov3D_precomp is not None and np.size(cov3D_precomp) else None
    inputs = dict(P=P, M=M, bg=_f32(bg), means3D=means3D, shs=shs_, colors=col, opac=_f32(opacities), scales=sc,
                  mod=float(scale_modifier), rots=ro, cov3D=c3, view=_f32(viewmatrix), proj=_f32(projmatrix),
                  inv=_f32(inv_viewprojmatrix), cam=_f32(campos), tfx=float(tanfovx), tfy=float(tanfovy))
    s = settings_struct(settings, tile_rows)
    out = np.zeros((3, H, W), np.float32)
    radii = np.zeros(P, np.int32)
    handle = ctypes.c_void_p()
    i = inputs
    rc = lib().orc_forward(P, int(sh_degree), M, _p(i["bg"]), int(W), int(H), ctypes.byref(s), _p(means3D), _p(shs_),
                           _p(col), _p(i["opac"]), _p(sc), ctypes.c_float(i["mod"]), _p(ro), _p(c3), _p(i["view"]),
                           _p(i["proj"]), _p(i["inv"]), _p(i["cam"]), ctypes.c_float(i["tfx"]), ctypes.c_float(i["tfy"]),
                           int(bool(prefiltered)), _p(out), _p(radii), ctypes.byref(handle))
    if rc < 0:
        raise RuntimeError({-2: "sorted modes need scales and rotations", -3: "Not supported queue size",
                            -4: "invalid sort mode"}.get(rc, f"oracle forward failed with code {rc}"))
    return Frame(handle, out, radii, rc, inputs)


def forward_scene(scene, settings: Optional[dict] = None, tile_rows=None, cov3D_precomp=None,
                  render_depth: bool = False, prefiltered: bool = False) -> Frame:
    """Convenience: run the oracle on a diff_gaussian_rasterization.scenes.Scene."""
    if render_depth:
        settings = {**(settings or {}), "_render_depth": True}
    return forward(bg=scene.bg, means3D=scene.means3D, opacities=scene.opacities, viewmatrix=scene.viewmatrix,
                   projmatrix=scene.projmatrix, inv_viewprojmatrix=scene.inv_viewprojmatrix, campos=scene.campos,
                   tanfovx=scene.tanfovx, tanfovy=scene.tanfovy, W=scene.W, H=scene.H, shs=scene.shs,
                   colors_precomp=scene.colors_precomp, scales=scene.scales, rotations=scene.rotations,
                   cov3D_precomp=cov3D_precomp, sh_degree=scene.sh_degree, scale_modifier=scene.scale_modifier,
                   settings=settings, tile_rows=tile_rows, prefiltered=prefiltered)


def mark_visible(means3D, viewmatrix, projmatrix) -> np.ndarray:
    m = _f32(means3D)
    out = np.zeros(m.shape[0], np.uint8)
    lib().orc_mark_visible(int(m.shape[0]), _p(m), _p(_f32(viewmatrix)), _p(_f32(projmatrix)), _p(out))
    return out.astype(bool)


def set_flag(name: str, value: int) -> None:
    L = lib()
    L.orc_set_flag.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.orc_set_flag.restype = None
    L.orc_set_flag(name.encode(), int(value))


@contextlib.contextmanager
def blend_nudge(alpha: float = 0.0, T: float = 0.0, cull_alpha: float = 0.0):
    """Inside the block the oracle's per-pixel blend decisions use thresholds moved by these amounts (alpha < 1/255 + alpha: skip; test_T < 1e-4 + T: stop;
    the 4x4 sub-tile culling's alpha < 1/255 + cull_alpha): the checker's question "is this the reference's result with a decision that sits on its
    threshold taken the other way?" (tests/test_gpu_parity.py).  Everything else -- preprocessing, tile culling, sorting -- is untouched."""
    L = lib()
    L.orc_set_blend_nudge.argtypes = [ctypes.c_float] * 3
    L.orc_set_blend_nudge.restype = None
    L.orc_set_blend_nudge(float(alpha), float(T), float(cull_alpha))
    try:
        yield
    finally:
        L.orc_set_blend_nudge(0.0, 0.0, 0.0)


@contextlib.contextmanager
def forced_alpha_flips(decisions, W: int, T_pixels=()):
    """Inside the block the oracle takes the per-pixel alpha test of every (x, y, gaussian_id) in `decisions` the OTHER way (blends what it would skip,
    skips what it would blend), and at every (x, y) of `T_pixels` the "test_T < 1e-4: stop" decision of the step whose test_T lies within 2e-6 (relative) of
    1e-4 -- and nothing else differently: the exact form of blend_nudge for single decisions that sit on their threshold."""
    L = lib()
    L.orc_set_forced_alpha_flips.argtypes = [ctypes.c_int, ctypes.c_void_p]
    L.orc_set_forced_alpha_flips.restype = None
    L.orc_set_forced_T_flips.argtypes = [ctypes.c_int, ctypes.c_void_p]
    L.orc_set_forced_T_flips.restype = None
    keys = np.ascontiguousarray([((int(y) * int(W) + int(x)) << 32) | (int(g) & 0xFFFFFFFF) for x, y, g in decisions], dtype=np.uint64)
    pix = np.ascontiguousarray([int(y) * int(W) + int(x) for x, y in T_pixels], dtype=np.uint32)
    L.orc_set_forced_alpha_flips(int(keys.size), keys.ctypes.data_as(ctypes.c_void_p))
    L.orc_set_forced_T_flips(int(pix.size), pix.ctypes.data_as(ctypes.c_void_p))
    try:
        yield
    finally:
        L.orc_set_forced_alpha_flips(0, None)
        L.orc_set_forced_T_flips(0, None)


def num_threads() -> int:
    return int(lib().orc_num_threads())
