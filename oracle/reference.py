"""ctypes front end of THE REFERENCE ITSELF compiled for gfx950 (oracle/_ref/libstp_ref*.so).

TEST INFRASTRUCTURE ONLY -- see oracle/ref_build/build_ref.sh for what the library is (the reference's
own cuda_rasterizer sources, translated by the ROCm image's hipify-perl and compiled by hipcc, plus our
C-ABI driver) and oracle/stp_oracle.h for who may load it (tests/, tools/, bench.py's baseline leg;
never the product).  It needs a GPU: it is the reference's kernels that run.

The API mirrors oracle/oracle.py (`forward_scene`, `Frame.array`, `Frame.backward`) so a test can put the
CPU oracle, the reference and the product side by side.  Two builds:
  variant "ieee"  (-ffp-contract=off)  every +,-,*,/ and sqrt has one IEEE meaning -> the CPU oracle must
                                         match its integer/index results and per-Gaussian state BIT FOR BIT;
  variant "fast"  (hipcc defaults)      what a user's build of the reference on ROCm would be; used for the
                                         reference-on-MI355X timing and tolerance-level comparisons.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

from . import oracle as _orc

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_DIR = os.path.join(_HERE, "_ref")
_LIBS = {"ieee": "libstp_ref_ieee.so", "fast": "libstp_ref.so"}
_loaded: Dict[str, ctypes.CDLL] = {}


def build() -> None:
    """Runs the recipe when /root/reference is present (this container); a no-op on the GPU box."""
    subprocess.check_call(["bash", os.path.join(_HERE, "ref_build", "build_ref.sh")])


def available(variant: str = "ieee") -> bool:
    return os.path.exists(os.path.join(_REF_DIR, _LIBS[variant]))


def lib(variant: str = "ieee") -> ctypes.CDLL:
    if variant not in _loaded:
        path = os.path.join(_REF_DIR, _LIBS[variant])
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: run oracle/ref_build/build_ref.sh where /root/reference exists")
        L = ctypes.CDLL(path)
        fp = ctypes.c_void_p
        L.ref_forward.restype = ctypes.c_int
        L.ref_forward.argtypes = [ctypes.c_int] * 3 + [fp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_orc.OrcSettings),
                                  fp, fp, fp, fp, fp, ctypes.c_float, fp, fp, fp, fp, fp, fp,
                                  ctypes.c_float, ctypes.c_float, ctypes.c_int, fp, fp, ctypes.POINTER(ctypes.c_void_p)]
        L.ref_backward.restype = ctypes.c_int
        L.ref_backward.argtypes = [fp] * 12
        L.ref_time_steps.restype = ctypes.c_int
        L.ref_time_steps.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
        L.ref_mark_visible.restype = None
        L.ref_mark_visible.argtypes = [ctypes.c_int, fp, fp, fp, fp]
        L.ref_frame_free.restype = None
        L.ref_frame_free.argtypes = [fp]
        L.ref_frame_array.restype = ctypes.c_int64
        L.ref_frame_array.argtypes = [fp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
        L.ref_frame_num_rendered.restype = ctypes.c_int
        L.ref_frame_num_rendered.argtypes = [fp]
        L.ref_last_error.restype = ctypes.c_char_p
        L.ref_build_info.restype = ctypes.c_char_p
        _loaded[variant] = L
    return _loaded[variant]


def build_info(variant: str = "ieee") -> str:
    return lib(variant).ref_build_info().decode()


class Frame:
    """Forward result of the reference + handle to its retained device state."""

    def __init__(self, L, handle, color, radii, num_rendered, P, M):
        self._L, self._h = L, handle
        self.color, self.radii, self.num_rendered = color, radii, num_rendered
        self.P, self.M = P, M

    def array(self, name: str) -> np.ndarray:
        ptr = ctypes.c_void_p()
        n = self._L.ref_frame_array(self._h, name.encode(), ctypes.byref(ptr))
        if n < 0:
            raise KeyError(name)
        dt = np.dtype(_orc._ARRAY_TYPES[name])
        if n == 0:
            return np.zeros(0, dtype=dt)
        buf = (ctypes.c_char * (n * dt.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dt).copy()

    def backward(self, dL_dout: np.ndarray, pixel_colors: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
        P, M = self.P, self.M
        dL = _orc._f32(dL_dout)
        pc = _orc._f32(self.color if pixel_colors is None else pixel_colors)
        g = dict(dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 2, 2), np.float32),
                 dL_dopacity=np.zeros((P, 1), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
                 dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dcov3D=np.zeros((P, 6), np.float32),
                 dL_dsh=np.zeros((P, M, 3), np.float32), dL_dscales=np.zeros((P, 3), np.float32),
                 dL_drotations=np.zeros((P, 4), np.float32))
        p = _orc._p
        rc = self._L.ref_backward(self._h, p(pc), p(dL), p(g["dL_dmeans2D"]), p(g["dL_dconic"]), p(g["dL_dopacity"]),
                                  p(g["dL_dcolors"]), p(g["dL_dmeans3D"]), p(g["dL_dcov3D"]), p(g["dL_dsh"]),
                                  p(g["dL_dscales"]), p(g["dL_drotations"]))
        if rc != 0:
            raise RuntimeError(self._L.ref_last_error().decode())
        return g

    def time_steps(self, dL_dout: Optional[np.ndarray], warmup: int = 3, steps: int = 10):
        """(forward ms, backward ms) per step of the reference on this frame's resident inputs."""
        a, b = ctypes.c_float(), ctypes.c_float()
        dL = None if dL_dout is None else _orc._f32(dL_dout)
        rc = self._L.ref_time_steps(self._h, _orc._p(dL), int(warmup), int(steps), ctypes.byref(a), ctypes.byref(b))
        if rc != 0:
            raise RuntimeError(self._L.ref_last_error().decode())
        return float(a.value), float(b.value)

    def free(self):
        if self._h:
            self._L.ref_frame_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def forward(*, bg, means3D, opacities, viewmatrix, projmatrix, inv_viewprojmatrix, campos, tanfovx, tanfovy,
            W, H, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
            sh_degree=3, scale_modifier=1.0, settings: Optional[dict] = None, prefiltered=False,
            variant: str = "ieee") -> Frame:
    L = lib(variant)
    f32, p = _orc._f32, _orc._p
    means3D = f32(means3D)
    P = int(means3D.shape[0])
    opt = lambda a: f32(a) if a is not None and np.size(a) else None
    shs_, col, sc, ro, c3 = opt(shs), opt(colors_precomp), opt(scales), opt(rotations), opt(cov3D_precomp)
    M = int(shs_.shape[1]) if shs_ is not None else 0
    s = _orc.settings_struct(settings)
    out = np.zeros((3, H, W), np.float32)
    radii = np.zeros(P, np.int32)
    handle = ctypes.c_void_p()
    keep = [f32(bg), f32(opacities), f32(viewmatrix), f32(projmatrix), f32(inv_viewprojmatrix), f32(campos)]
    rc = L.ref_forward(P, int(sh_degree), M, p(keep[0]), int(W), int(H), ctypes.byref(s), p(means3D), p(shs_), p(col),
                       p(keep[1]), p(sc), ctypes.c_float(float(scale_modifier)), p(ro), p(c3), p(keep[2]), p(keep[3]),
                       p(keep[4]), p(keep[5]), ctypes.c_float(float(tanfovx)), ctypes.c_float(float(tanfovy)),
                       int(bool(prefiltered)), p(out), p(radii), ctypes.byref(handle))
    if rc < 0:
        raise RuntimeError(L.ref_last_error().decode())
    return Frame(L, handle, out, radii, rc, P, M)


def forward_scene(scene, settings: Optional[dict] = None, cov3D_precomp=None, render_depth: bool = False,
                  variant: str = "ieee", prefiltered: bool = False) -> Frame:
    if render_depth:
        settings = {**(settings or {}), "_render_depth": True}
    return forward(bg=scene.bg, means3D=scene.means3D, opacities=scene.opacities, viewmatrix=scene.viewmatrix,
                   projmatrix=scene.projmatrix, inv_viewprojmatrix=scene.inv_viewprojmatrix, campos=scene.campos,
                   tanfovx=scene.tanfovx, tanfovy=scene.tanfovy, W=scene.W, H=scene.H, shs=scene.shs,
                   colors_precomp=scene.colors_precomp, scales=scene.scales, rotations=scene.rotations,
                   cov3D_precomp=cov3D_precomp, sh_degree=scene.sh_degree, scale_modifier=scene.scale_modifier,
                   settings=settings, variant=variant, prefiltered=prefiltered)


def mark_visible(means3D, viewmatrix, projmatrix, variant: str = "ieee") -> np.ndarray:
    m = _orc._f32(means3D)
    out = np.zeros(m.shape[0], np.uint8)
    lib(variant).ref_mark_visible(int(m.shape[0]), _orc._p(m), _orc._p(_orc._f32(viewmatrix)),
                                  _orc._p(_orc._f32(projmatrix)), _orc._p(out))
    return out.astype(bool)
