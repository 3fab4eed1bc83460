"""`diff_gaussian_rasterization._C` -- host binding of libstp_raster.so.

Mirrors the reference's pybind module of the same name (reference ext.cpp:15-19,
rasterize_points.cu:43-253): the three functions below take and return exactly the tensors the
reference's do, in the same order.  PyTorch is used for device memory and the current HIP stream only;
all compute is in the hand-written HIP library reached through its C ABI (include/stp_raster.h) with
ctypes -- no torch C++ extension, hence no hipify pass over our sources.

There is NO fallback: if the library is missing or the tensors are not on a GPU, these functions raise.
"""
from __future__ import annotations

import ctypes
import os
from typing import Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libstp_raster.so"
_lib = None


GRAD_RECORD_FLOATS = 16  # == STP_GRAD_RECORD_FLOATS (include/stp_raster.h): floats per Gaussian in the backward's hand-over buffer


class StpSettings(ctypes.Structure):
    """POD mirror of StpSettings (include/stp_raster.h) == reference SplattingSettings (rasterizer.h:129-135)."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "sort_mode", "sort_order", "queue_tile_4x4", "queue_tile_2x2", "queue_per_pixel",
        "rect_bounding", "tight_opacity_bounding", "tile_based_culling", "hierarchical_4x4_culling",
        "load_balancing", "proper_ewa_scaling", "tile_y0", "tile_y1", "record_blend_log", "debug_visualization")]


_ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)


def library_path() -> str:
    return os.environ.get("STP_RASTER_LIB", os.path.join(_HERE, _LIB_NAME))


def _load():
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `make -C stopthepop-rasterization_amd/csrc` "
            f"(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    L = ctypes.CDLL(path)
    vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    L.stp_forward.restype = ci
    L.stp_forward.argtypes = [_ALLOC_FN, vp, _ALLOC_FN, vp, _ALLOC_FN, vp, ci, ci, ci, vp, ci, ci,
                              ctypes.POINTER(StpSettings), vp, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, vp, cf, cf, ci,
                              vp, vp, ci, vp]
    L.stp_backward.restype = ci
    L.stp_backward.argtypes = [ci, ci, ci, ci, vp, ci, ci, ctypes.POINTER(StpSettings), vp, vp, vp, vp, vp, cf, vp, vp,
                               vp, vp, vp, vp, cf, cf, vp, vp, vp, vp, vp, vp] + [vp] * 9 + [ci, vp]
    L.stp_backward_phases.restype = ci
    L.stp_backward_phases.argtypes = [ci] + L.stp_backward.argtypes
    L.stp_mark_visible.restype = ci
    L.stp_mark_visible.argtypes = [ci, vp, vp, vp, vp, vp]
    L.stp_last_error.restype = ctypes.c_char_p
    L.stp_abi_version.restype = ci
    for name in ("stp_geometry_buffer_size", "stp_binning_buffer_size", "stp_image_buffer_size"):
        getattr(L, name).restype = ctypes.c_size_t
    L.stp_geometry_buffer_size.argtypes = [ci, ctypes.POINTER(StpSettings)]
    L.stp_binning_buffer_size.argtypes = [ci]
    L.stp_image_buffer_size.argtypes = [ci, ci]
    szp = ctypes.POINTER(ctypes.c_size_t)
    L.stp_geometry_layout.argtypes = [ci, ctypes.POINTER(StpSettings), ctypes.c_char_p, szp, szp]
    L.stp_binning_layout.argtypes = [ci, ctypes.c_char_p, szp, szp]
    L.stp_image_layout.argtypes = [ci, ci, ctypes.c_char_p, szp, szp]
    L.stp_timing_enable.argtypes = [ci]
    L.stp_timing_enable.restype = None
    L.stp_timing_read.argtypes = [ctypes.POINTER(ctypes.c_float)]
    L.stp_timing_read.restype = ci
    L.stp_timing_text.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    L.stp_timing_text.restype = ctypes.c_size_t
    if L.stp_abi_version() != 4:
        raise ImportError("libstp_raster.so ABI version mismatch")
    _lib = L
    return L


def settings_from_dict(d: dict, tile_rows=None) -> StpSettings:
    """dict -> POD.  All keys are mandatory, as in the reference's json parser (rasterizer.h:160-182,
    every field read with .at()); a missing key raises KeyError where the reference raised json::out_of_range."""
    ss, cs = d["sort_settings"], d["culling_settings"]
    q = ss["queue_sizes"]
    s = StpSettings()
    s.sort_mode = int(ss["sort_mode"])
    s.sort_order = int(ss["sort_order"])
    s.queue_tile_4x4 = int(q["tile_4x4"])
    s.queue_tile_2x2 = int(q["tile_2x2"])
    s.queue_per_pixel = int(q["per_pixel"])
    s.rect_bounding = int(bool(cs["rect_bounding"]))
    s.tight_opacity_bounding = int(bool(cs["tight_opacity_bounding"]))
    s.tile_based_culling = int(bool(cs["tile_based_culling"]))
    s.hierarchical_4x4_culling = int(bool(cs["hierarchical_4x4_culling"]))
    s.load_balancing = int(bool(d["load_balancing"]))
    s.proper_ewa_scaling = int(bool(d["proper_ewa_scaling"]))
    tr = tile_rows if tile_rows is not None else d.get("_tile_rows")
    if tr is not None:
        s.tile_y0, s.tile_y1 = int(tr[0]), int(tr[1])
    # private key set by the autograd function for training forwards (see __init__._RasterizeGaussians.forward);
    # STP_BACKWARD=resort in the environment forces the reference-style re-sorting backward everywhere
    if d.get("_record_blend_log") and os.environ.get("STP_BACKWARD", "replay") != "resort":
        s.record_blend_log = 1
    return s


def _raise_last(rc: int):
    msg = _load().stp_last_error()
    raise RuntimeError((msg.decode() if msg else "") or f"libstp_raster error {rc}")


def _ptr(t: torch.Tensor):
    """Device pointer of a contiguous fp32/int32 tensor; empty tensor -> NULL (reference convention:
    `torch.Tensor([])` marks an absent optional input and its data_ptr is null)."""
    if t is None or t.numel() == 0:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _prep(t: torch.Tensor, device) -> torch.Tensor:
    if t is None or t.numel() == 0:
        return t
    if t.device != device:
        raise RuntimeError(f"expected all tensors on {device}, got one on {t.device}")
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected float32 tensor, got {t.dtype}")
    return t.contiguous()


def _stream_ptr(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


# The binning buffer (tile lists + list-ordered entry records, ~100 B per entry) and the training forward's image
# buffer are large.  The latter carries the blend log (512 B per pixel of the tile grid, 1.07 GB at 1080p).
# Cycling a block of that size through torch's caching allocator every step invites splitting: smaller requests
# carve pieces off the free block, the next forward finds no 2 GB hole and the allocator falls back to hipMalloc
# (tens of ms per step, reserved memory growing by 2 GB a step -- observed on MI355X).  Buffers of this class are
# therefore kept on a small free list of our own: handed out by the forward, handed back by the backward.
_BIG_BYTES = 256 << 20
_BIG_STEP = 64 << 20
_BIG_KEEP = 4                # free buffers kept per device (see also set_scratch_pool_limit)
_big_max_bytes = None        # optional cap on the bytes the free list may hold per device
_big_free = {}               # device index -> [(tensor, event recorded on the releasing stream), ...]
_big_generation = {}         # data_ptr -> how many times the buffer at this address was handed out


def clear_scratch_pool(device=None) -> int:
    """Drops the pooled scratch buffers (tile lists / blend logs waiting for reuse) of `device` (all devices when None) so
    that torch.cuda.empty_cache() can hand their memory back; returns the number of bytes released.  Buffers that a live
    autograd graph still holds are not affected."""
    freed = 0
    for d in list(_big_free) if device is None else [torch.device(device).index if not isinstance(device, int) else device]:
        for t, _ in _big_free.pop(d, []):
            freed += t.numel()
    return freed


def set_scratch_pool_limit(max_buffers: int = 4, max_bytes=None) -> None:
    """Bounds the free list: at most `max_buffers` buffers and (optionally) `max_bytes` bytes per device stay pooled."""
    global _BIG_KEEP, _big_max_bytes
    _BIG_KEEP, _big_max_bytes = int(max_buffers), (None if max_bytes is None else int(max_bytes))


class _Resizer:
    """The reference's resizeFunctional (rasterize_points.cu:33-41): grows a byte tensor on request.  The library may
    call it twice per forward (a size guess before the num_rendered hand-over, the exact size afterwards if the guess
    was short): a request the current buffer already covers returns the same pointer."""

    def __init__(self, device, pooled=False):
        self.tensor = torch.empty(0, dtype=torch.uint8, device=device)
        self.pooled = pooled
        self.from_pool = False
        self.cb = _ALLOC_FN(self._alloc)

    def _alloc(self, _user, nbytes):
        try:
            nbytes = int(nbytes)
            if nbytes <= self.tensor.numel() and nbytes > 0:
                return self.tensor.data_ptr()
            if self.pooled and nbytes >= _BIG_BYTES:
                if self.from_pool:       # the guess was too small: the buffer goes back, a larger one comes
                    _put_back(self.tensor)
                # capacity in steps of 64 MiB, so that a buffer whose size follows the number of tile-list entries
                # (it changes a little from view to view) finds its predecessor on the free list
                cap = (nbytes + _BIG_STEP - 1) // _BIG_STEP * _BIG_STEP
                free = _big_free.setdefault(self.tensor.device.index, [])
                fits = [i for i, (t, _) in enumerate(free) if nbytes <= t.numel() <= cap + cap // 4]
                hit = min(fits, key=lambda i: free[i][0].numel()) if fits else None
                if hit is not None:
                    self.tensor, ev = free.pop(hit)
                    if ev is not None:   # the releasing stream's kernels may still be reading it: order this stream behind them
                        torch.cuda.current_stream(self.tensor.device).wait_event(ev)
                else:
                    self.tensor = torch.empty(cap, dtype=torch.uint8, device=self.tensor.device)
                self.from_pool = True
                ptr = self.tensor.data_ptr()
                _big_generation[ptr] = _big_generation.get(ptr, 0) + 1
                return ptr
            self.tensor.resize_(nbytes)
            return self.tensor.data_ptr() if nbytes else 0
        except Exception:  # surfaces as STP_ERR_ALLOC on the C side
            return 0


def scratch_generation(buf: torch.Tensor) -> int:
    """Token the autograd function keeps with a pooled buffer (0 for ordinary ones); see check_scratch."""
    return _big_generation.get(buf.data_ptr(), 0) if buf.numel() >= _BIG_BYTES else 0


def check_scratch(buf: torch.Tensor, generation: int) -> None:
    """A second backward through a retained graph after a later forward reused the buffer must not read that
    forward's blend log: fail loudly instead."""
    if generation and _big_generation.get(buf.data_ptr(), 0) != generation:
        raise RuntimeError("a scratch buffer of this forward (tile lists / blend log) was recycled by a later forward "
                           "pass; run the forward again before this backward")


def _put_back(buf: torch.Tensor) -> None:
    free = _big_free.setdefault(buf.device.index, [])
    if any(t.data_ptr() == buf.data_ptr() for t, _ in free):
        return
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(buf.device))
    free.append((buf, ev))
    while len(free) > _BIG_KEEP or (_big_max_bytes is not None and len(free) > 1 and sum(t.numel() for t, _ in free) > _big_max_bytes):
        free.pop(0)


def release_scratch(buf: torch.Tensor) -> None:
    """Hand a pooled buffer back after the backward that consumed it (reuse is ordered behind the releasing stream)."""
    if buf.numel() < _BIG_BYTES or not buf.is_cuda:
        return
    _put_back(buf)


def _require_gpu(means3D: torch.Tensor):
    if not means3D.is_cuda:
        raise RuntimeError("diff_gaussian_rasterization (MI355X build) needs tensors on a GPU device; "
                           "there is no CPU path in the product")


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, inv_viewprojmatrix, tan_fovx, tan_fovy, image_height, image_width,
                        sh, degree, campos, prefiltered, settings_dict, render_depth, debug
                        ) -> Tuple[int, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """== RasterizeGaussiansCUDA (reference rasterize_points.cu:43-138).
    Returns (num_rendered, out_color (3,H,W), radii (P,) int32, geomBuffer, binningBuffer, imgBuffer)."""
    L = _load()
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _require_gpu(means3D)
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    # the render kernels write every pixel of the tile rows they cover and preprocess writes every Gaussian's radius:
    # zero-filled outputs are only needed when nothing runs (P == 0) or when a tile-row window leaves rows untouched
    windowed = settings_dict.get("_tile_rows") is not None
    make = torch.zeros if (P == 0 or windowed) else torch.empty
    out_color = make((3, H, W), dtype=torch.float32, device=dev)
    radii = make((P,), dtype=torch.int32, device=dev)
    geom, binning, img = _Resizer(dev), _Resizer(dev, pooled=True), _Resizer(dev, pooled=True)
    rendered = 0
    if P != 0:
        M = int(sh.size(1)) if sh.numel() != 0 else 0
        s = settings_from_dict(settings_dict)
        if render_depth:  # DebugVisualization::Depth (reference rasterize_points.cu:104-107); no log: it has no backward
            s.debug_visualization = 1
            s.record_blend_log = 0
        t = [_prep(x, dev) for x in (background, means3D, sh, colors, opacity, scales, rotations, cov3D_precomp,
                                     viewmatrix, projmatrix, inv_viewprojmatrix, campos)]
        bg_, m3_, sh_, col_, op_, sc_, ro_, c3_, vm_, pm_, inv_, cam_ = t
        with torch.cuda.device(dev):
            rc = L.stp_forward(geom.cb, None, binning.cb, None, img.cb, None, P, int(degree), M, _ptr(bg_), W, H,
                               ctypes.byref(s), _ptr(m3_), _ptr(sh_), _ptr(col_), _ptr(op_), _ptr(sc_),
                               ctypes.c_float(scale_modifier), _ptr(ro_), _ptr(c3_), _ptr(vm_), _ptr(pm_), _ptr(inv_),
                               _ptr(cam_), ctypes.c_float(tan_fovx), ctypes.c_float(tan_fovy), int(bool(prefiltered)),
                               _ptr(out_color), _ptr(radii), int(bool(debug)), _stream_ptr(dev))
        for r in (geom, binning, img):
            r.cb = None  # the callback object refers to its resizer: a cycle that would keep the buffers alive until a gc run
        if rc < 0:
            _raise_last(rc)
        rendered = rc
    return rendered, out_color, radii, geom.tensor, binning.tensor, img.tensor


def rasterize_gaussians_backward(background, means3D, radii, opacities, colors, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, inv_viewprojmatrix, tan_fovx, tan_fovy,
                                 pixel_colors, dL_dout_color, sh, degree, campos, geomBuffer, R, binningBuffer,
                                 imageBuffer, settings_dict, debug, phases=3, partial=None):
    """== RasterizeGaussiansBackwardCUDA (reference rasterize_points.cu:140-232).
    Returns (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations).

    Extension for tile-row sharding (not in the reference): phases=1 runs only the render half and
    returns its per-Gaussian partial sums as the library's (P,16) gradient records (stp_raster.h);
    phases=2 takes the records (after the caller's all-reduce) as `partial` and runs the preprocess half."""
    L = _load()
    _require_gpu(means3D)
    dev = means3D.device
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if sh.numel() != 0 else 0
    z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
    records = partial if partial is not None else z(P, GRAD_RECORD_FLOATS)
    if records.shape != (P, GRAD_RECORD_FLOATS) or records.dtype != torch.float32 or not records.is_contiguous():
        raise ValueError("partial must be a contiguous float32 (P,%d) tensor" % GRAD_RECORD_FLOATS)
    if phases & 2:
        # the per-Gaussian half writes every row of its outputs (zeros for invisible Gaussians): no zero-fill needed,
        # except for the scale/rotation gradients when a precomputed covariance is used (then they are not touched)
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        zs = e if scales.numel() != 0 else z
        dL_dmeans2D, dL_dcolors, dL_dopacity = e(P, 3), e(P, 3), e(P, 1)
        dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations = e(P, 3), e(P, 6), e(P, M, 3), zs(P, 3), zs(P, 4)
    else:
        dL_dmeans2D = dL_dcolors = dL_dopacity = None
        dL_dmeans3D = dL_dcov3D = dL_dsh = dL_dscales = dL_drotations = None
    s = settings_from_dict(settings_dict)
    if P != 0:
        t = [_prep(x, dev) for x in (background, means3D, sh, colors, opacities, scales, rotations, cov3D_precomp,
                                     viewmatrix, projmatrix, inv_viewprojmatrix, campos, pixel_colors, dL_dout_color)]
        bg_, m3_, sh_, col_, op_, sc_, ro_, c3_, vm_, pm_, inv_, cam_, pix_, dl_ = t
        radii_ = radii.contiguous()
        with torch.cuda.device(dev):
            rc = L.stp_backward_phases(int(phases), P, int(degree), M, int(R), _ptr(bg_), W, H, ctypes.byref(s), _ptr(m3_),
                                       _ptr(sh_), _ptr(op_), _ptr(col_), _ptr(sc_), ctypes.c_float(scale_modifier), _ptr(ro_),
                                       _ptr(c3_), _ptr(vm_), _ptr(pm_), _ptr(inv_), _ptr(cam_), ctypes.c_float(tan_fovx),
                                       ctypes.c_float(tan_fovy), _ptr(pix_), _ptr(radii_), _ptr(geomBuffer),
                                       _ptr(binningBuffer), _ptr(imageBuffer), _ptr(dl_), _ptr(dL_dmeans2D), _ptr(records),
                                       _ptr(dL_dopacity), _ptr(dL_dcolors), _ptr(dL_dmeans3D), _ptr(dL_dcov3D), _ptr(dL_dsh),
                                       _ptr(dL_dscales), _ptr(dL_drotations), int(bool(debug)), _stream_ptr(dev))
        if rc < 0:
            _raise_last(rc)
    if phases == 1:
        return records
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def mark_visible(means3D, viewmatrix, projmatrix) -> torch.Tensor:
    """== markVisible (reference rasterize_points.cu:234-253)."""
    L = _load()
    _require_gpu(means3D)
    dev = means3D.device
    P = int(means3D.size(0))
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        m3_, vm_, pm_ = (_prep(x, dev) for x in (means3D, viewmatrix, projmatrix))
        with torch.cuda.device(dev):
            rc = L.stp_mark_visible(P, _ptr(m3_), _ptr(vm_), _ptr(pm_), ctypes.c_void_p(present.data_ptr()), _stream_ptr(dev))
        if rc < 0:
            _raise_last(rc)
    return present


# ---- introspection helpers (tests / bench; not part of the reference surface) -----------------------
_GEOM_TYPES = {"depths": torch.float32, "clamped": torch.uint8, "radii": torch.int32, "rects2D": torch.float32,
               "means2D": torch.float32, "cov3D": torch.float32, "cov3D_inv": torch.float32,
               "conic_opacity": torch.float32, "rgb": torch.float32, "tiles_touched": torch.int32,
               "point_offsets": torch.int32}
_BIN_TYPES = {"point_list": torch.int32, "point_list_unsorted": torch.int32, "keys": torch.int64, "keys_unsorted": torch.int64}
_IMG_TYPES = {"final_T": torch.float32, "n_contrib": torch.int32, "ranges": torch.int32, "tile_flags": torch.int32,
              "blend_log": torch.int16}  # the last two exist only in a buffer of a recording forward


def timing_text() -> str:
    """The reference's `timings_text` (what its viewer displays), from the stages measured since timing_enable(True)."""
    buf = ctypes.create_string_buffer(512)
    _load().stp_timing_text(buf, 512)
    return buf.value.decode()


def _view(buf: torch.Tensor, off: int, count: int, dtype) -> torch.Tensor:
    nbytes = count * torch.empty((), dtype=dtype).element_size()
    return buf[off:off + nbytes].view(dtype)


def geometry_array(geomBuffer, P, settings_dict, name):
    s = settings_from_dict(settings_dict)
    off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
    if _load().stp_geometry_layout(int(P), ctypes.byref(s), name.encode(), ctypes.byref(off), ctypes.byref(cnt)) != 0:
        raise KeyError(name)
    return _view(geomBuffer, off.value, cnt.value, _GEOM_TYPES[name])


def binning_array(binningBuffer, R, name):
    off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
    if _load().stp_binning_layout(int(R), name.encode(), ctypes.byref(off), ctypes.byref(cnt)) != 0:
        raise KeyError(name)
    return _view(binningBuffer, off.value, cnt.value, _BIN_TYPES[name])


def image_array(imgBuffer, W, H, name):
    off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
    if _load().stp_image_layout(int(W), int(H), name.encode(), ctypes.byref(off), ctypes.byref(cnt)) != 0:
        raise KeyError(name)
    return _view(imgBuffer, off.value, cnt.value, _IMG_TYPES[name])


def timing_enable(flag: bool) -> None:
    _load().stp_timing_enable(int(bool(flag)))


def timing_read():
    """Milliseconds of the last call's stages: Preprocess, Duplicate, Sort, Render, BwdRender, BwdPreprocess."""
    arr = (ctypes.c_float * 6)()
    _load().stp_timing_read(arr)
    names = ("Preprocess", "Duplicate", "Sort", "Render", "BwdRender", "BwdPreprocess")
    return {n: float(v) for n, v in zip(names, arr)}
