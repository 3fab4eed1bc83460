"""`diff_gaussian_rasterization._C` -- host binding of libstp_raster.so.

Mirrors the reference's pybind module of the same name (reference ext.cpp:15-19,
rasterize_points.cu:43-253): the three functions below take and return exactly the tensors the
reference's do, in the same order.  PyTorch is used for device memory and the current HIP stream only;
all compute is in the hand-written HIP library, reached through its C ABI (include/stp_raster.h).

The hot functions -- rasterize_gaussians, rasterize_gaussians_backward, mark_visible and the scratch pool -- live in
the native module `_stp_host` (csrc/host/stp_torch_binding.cpp: a plain C++ torch extension built by g++, no kernels,
no hipify pass; it allocates through ATen, takes the current stream from c10 and calls the same C ABI).  What stays here
is cold: the settings POD for the introspection helpers, the backward-mode policy, the stage timer's readers (ctypes).

There is NO fallback: if the library or the native module is missing, or the tensors are not on a GPU, these functions raise.
"""
from __future__ import annotations

import ctypes
import os
from typing import Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libstp_raster.so"
_lib = None
_host = None   # the native module _stp_host, bound to the same library file as _lib


GRAD_RECORD_FLOATS = 16  # == STP_GRAD_RECORD_FLOATS (include/stp_raster.h): floats per Gaussian in the backward's hand-over buffer


class StpSettings(ctypes.Structure):
    """POD mirror of StpSettings (include/stp_raster.h) == reference SplattingSettings (rasterizer.h:129-135)."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "sort_mode", "sort_order", "queue_tile_4x4", "queue_tile_2x2", "queue_per_pixel",
        "rect_bounding", "tight_opacity_bounding", "tile_based_culling", "hierarchical_4x4_culling",
        "load_balancing", "proper_ewa_scaling", "tile_y0", "tile_y1", "record_blend_log", "debug_visualization")]


_lib_override = None


def library_path() -> str:
    p = _lib_override or os.environ.get("STP_RASTER_LIB", os.path.join(_HERE, _LIB_NAME))
    return os.path.join(_HERE, "libstp_raster_fma.so") if p == "fma" else p   # (STP_RASTER_LIB=fma: the shipped second build, by name)


def use_library(path=None) -> None:
    """The next call loads `path` instead of the default library (None: back to the default).  `path` may be a file or one of the
    shipped builds by name: "fma" = libstp_raster_fma.so (depth keys as fused multiply-add chains: 1.4-3 % faster, keys an ulp off the
    reference's now and then), "default" = libstp_raster.so (the reference's uncontracted depth keys)."""
    global _lib, _host, _lib_override
    if path in ("fma", "default"):
        path = os.path.join(_HERE, "libstp_raster_fma.so" if path == "fma" else _LIB_NAME)
    _lib, _host, _lib_override = None, None, path


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
    # (stp_forward / stp_backward_phases / stp_mark_visible are called by the native module _stp_host, not from here)
    L.stp_last_error.restype = ctypes.c_char_p
    L.stp_abi_version.restype = ci
    for name in ("stp_geometry_buffer_size", "stp_binning_buffer_size", "stp_image_buffer_size"):
        getattr(L, name).restype = ctypes.c_size_t
    L.stp_geometry_buffer_size.argtypes = [ci, ctypes.POINTER(StpSettings)]
    L.stp_binning_buffer_size.argtypes = [ci]
    L.stp_image_buffer_size.argtypes = [ci, ci]
    L.stp_blend_log_bytes.argtypes = [ci, ci]
    L.stp_blend_log_bytes.restype = ctypes.c_size_t
    szp = ctypes.POINTER(ctypes.c_size_t)
    L.stp_geometry_layout.argtypes = [ci, ctypes.POINTER(StpSettings), ctypes.c_char_p, szp, szp]
    L.stp_binning_layout.argtypes = [ci, ctypes.c_char_p, szp, szp]
    L.stp_image_layout.argtypes = [ci, ci, ctypes.c_char_p, szp, szp]
    L.stp_timing_enable.argtypes = [ci]
    L.stp_timing_enable.restype = None
    L.stp_timing_read.argtypes = [ctypes.POINTER(ctypes.c_float)]
    L.stp_timing_read.restype = ci
    L.stp_hbm_probe.argtypes = [ci, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ci, ctypes.c_void_p]
    L.stp_hbm_probe.restype = ci
    L.stp_timing_history.argtypes = [ctypes.POINTER(ctypes.c_float), ci]
    L.stp_timing_history.restype = ci
    L.stp_timing_history_host.argtypes = [ctypes.POINTER(ctypes.c_float), ci]
    L.stp_timing_history_host.restype = ci
    L.stp_timing_text.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    L.stp_timing_text.restype = ctypes.c_size_t
    L.stp_binning_layout_count.argtypes = [vp, ci]
    L.stp_binning_layout_count.restype = ci
    L.stp_forget_buffer.argtypes = [vp]
    L.stp_forget_buffer.restype = None
    if L.stp_abi_version() != 7:
        raise ImportError("libstp_raster.so ABI version mismatch")
    _lib = L
    return L


def _native():
    """The native module, bound to the library file _load() resolved."""
    global _host
    if _host is not None:
        return _host
    _load()
    try:
        from . import _stp_host
    except ImportError as ex:
        raise ImportError(f"the native host binding diff_gaussian_rasterization._stp_host is missing ({ex}): build it with "
                          f"`python stopthepop-rasterization_amd/csrc/host/build_host.py` (or python -c 'import __graft_entry__ as g; "
                          f"g.build()').  There is no pure-Python fallback.") from ex
    _stp_host.load_library(library_path())
    global _atexit_registered
    if not _atexit_registered:   # pooled tensors must not outlive the HIP runtime at interpreter shutdown
        import atexit
        atexit.register(lambda: _stp_host.clear_scratch_pool(-1))
        _atexit_registered = True
    _host = _stp_host
    return _host


_atexit_registered = False


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
    if _records_log(d):
        s.record_blend_log = 1
    return s


# ---- backward-mode policy (blend log: 512 B per pixel of the tile grid held between forward and backward) ----------
# "replay"  every training forward records the blend log, the backward replays it (fastest; default)
# "resort"  no log: the backward re-runs the per-pixel resort like the reference's (12 B per pixel held instead of 512)
# "auto"    record while the logs held by live autograd graphs + the pooled free buffers + the new log stay within
#           the byte budget, otherwise this forward falls back to "resort" (a trainer that sums K views before one
#           backward holds K logs)
# The process-wide default comes from STP_BACKWARD, read ONCE at import; set_backward_mode() changes it at run time and a
# settings object can override it per call (ExtendedSettings._backward_mode, see __init__.py).
_BACKWARD_MODES = ("replay", "resort", "auto")
_backward_mode = os.environ.get("STP_BACKWARD", "replay")
if _backward_mode not in _BACKWARD_MODES:
    raise ImportError(f"STP_BACKWARD={_backward_mode!r}: expected one of {_BACKWARD_MODES}")
_log_budget_bytes = int(float(os.environ.get("STP_LOG_BUDGET_GB", "16")) * (1 << 30))
_log_live = {}               # device index -> bytes of blend logs held by forwards whose backward has not run yet


def set_backward_mode(mode: str, log_budget_bytes=None) -> None:
    """Process-wide default of the policy above; `log_budget_bytes` (mode "auto") bounds live + pooled + new log bytes per device."""
    global _backward_mode, _log_budget_bytes
    if mode not in _BACKWARD_MODES:
        raise ValueError(f"backward mode {mode!r}: expected one of {_BACKWARD_MODES}")
    _backward_mode = mode
    if log_budget_bytes is not None:
        _log_budget_bytes = int(log_budget_bytes)


def backward_mode() -> str:
    return _backward_mode


def blend_log_bytes(width: int, height: int, tile_rows=None, depth=None) -> int:
    """Bytes of the blend log of one forward at this resolution.  depth = None: of a frame nothing is known about (400 B per pixel of the
    16x16 tile grid; later frames of the same kind get the depth their predecessors needed: blend_log_depth); depth = n: of a log of n
    records per pixel; depth = 0: of the DEEPEST log a forward may carve (what a memory policy has to budget before the forward has run).
    tile_rows = (y0, y1): of a forward restricted to that tile-row window (a rank of a tile-row shard holds its rows' log only)."""
    L = _load()
    y0, y1 = (0, 0) if tile_rows is None else (int(tile_rows[0]), int(tile_rows[1]))
    if depth is not None:
        L.stp_blend_log_bytes_depth.argtypes = [ctypes.c_int] * 5
        L.stp_blend_log_bytes_depth.restype = ctypes.c_size_t
        return int(L.stp_blend_log_bytes_depth(int(width), int(height), y0, y1, int(depth)))
    if tile_rows is None:
        return int(L.stp_blend_log_bytes(int(width), int(height)))
    L.stp_blend_log_bytes_rows.argtypes = [ctypes.c_int] * 4
    L.stp_blend_log_bytes_rows.restype = ctypes.c_size_t
    return int(L.stp_blend_log_bytes_rows(int(width), int(height), y0, y1))


class LogLease:
    """Accounts one forward's blend log against the device's budget until its backward has run -- or until the autograd
    graph that holds it is dropped without one (the lease is an attribute of the graph node: it dies with it)."""

    def __init__(self, index: int, nbytes: int):
        self.index, self.nbytes = index, int(nbytes)
        _log_live[index] = _log_live.get(index, 0) + self.nbytes

    def resize(self, nbytes: int) -> None:
        """The forward has run: account what its log really takes (the library chooses the depth per frame)."""
        if self.nbytes:
            _log_live[self.index] = _log_live.get(self.index, 0) - self.nbytes + int(nbytes)
            self.nbytes = int(nbytes)

    def release(self) -> None:
        if self.nbytes:
            _log_live[self.index] = _log_live.get(self.index, 0) - self.nbytes
            self.nbytes = 0

    __del__ = release


def live_log_bytes(device=None) -> int:
    return _log_live.get(_device_index(device), 0)


def decide_recording(mode, device, width: int, height: int, tile_rows=None) -> bool:
    """Does a training forward on `device` record the blend log under policy `mode` (None = the process default)?"""
    mode = mode or _backward_mode
    if mode not in _BACKWARD_MODES:
        raise ValueError(f"backward mode {mode!r}: expected one of {_BACKWARD_MODES}")
    if mode != "auto":
        return mode == "replay"
    idx = _device_index(device)
    need = blend_log_bytes(width, height, tile_rows, depth=0)   # (the deepest log the forward may carve: its depth is the library's choice)
    pooled = list(_native().pooled_sizes(idx))
    reuse = any(need <= n for n in pooled)   # a pooled buffer takes the new log: no new memory
    return _log_live.get(idx, 0) + sum(pooled) + (0 if reuse else need) <= _log_budget_bytes


def _on_device(device):
    """Context that makes `device` the calling thread's current HIP device (nothing to switch on a box without a GPU)."""
    import contextlib
    return torch.cuda.device(_device_index(device)) if torch.cuda.is_available() else contextlib.nullcontext()


def _device_index(device) -> int:
    if device is None:
        return torch.cuda.current_device()
    if isinstance(device, int):
        return device
    d = torch.device(device)
    return torch.cuda.current_device() if d.index is None else d.index


def _raise_last(rc: int):
    msg = _load().stp_last_error()
    raise RuntimeError((msg.decode() if msg else "") or f"libstp_raster error {rc}")


# ---- scratch pool (csrc/host/stp_torch_binding.cpp: buffers of >= 256 MiB -- tile lists with their entry records, image
# ---- state with the blend log -- are kept on a free list per device instead of cycling through torch's caching allocator)
def clear_scratch_pool(device=None) -> int:
    """Drops the pooled scratch buffers (tile lists / blend logs waiting for reuse) of `device` (all devices when None) so
    that torch.cuda.empty_cache() can hand their memory back; returns the number of bytes released.  Buffers that a live
    autograd graph still holds are not affected."""
    return int(_native().clear_scratch_pool(-1 if device is None else _device_index(device)))   # ('cuda' = the current device)


def set_scratch_pool_limit(max_buffers: int = 4, max_bytes=None) -> None:
    """Bounds the free list: at most `max_buffers` buffers and (optionally) `max_bytes` bytes per device stay pooled."""
    _native().set_scratch_pool_limit(int(max_buffers), -1 if max_bytes is None else int(max_bytes))


def scratch_generation(buf: torch.Tensor) -> int:
    """Token the autograd function keeps with a pooled buffer (0 for ordinary ones); see check_scratch."""
    return int(_native().scratch_generation(buf))


def check_scratch(buf: torch.Tensor, generation: int) -> None:
    """A second backward through a retained graph after a later forward reused the buffer must not read that
    forward's blend log: fail loudly instead."""
    _native().check_scratch(buf, int(generation))


def set_forward_split(tile_row: int, event) -> None:
    """The next rasterize_gaussians of this thread launches its render kernel for the tile rows below `tile_row` first and records `event`
    (a torch.cuda.Event that has been recorded once, so that its handle exists) on the current stream before it launches the rest
    (include/stp_raster.h: stp_set_forward_split; tile_shard.py sends the first half of a strip behind it)."""
    _native().set_forward_split(int(tile_row), 0 if event is None else int(event.cuda_event))   # (None: clear a pending request)


def release_scratch(buf: torch.Tensor) -> None:
    """Hand a pooled buffer back after the backward that consumed it (reuse is ordered behind the releasing stream)."""
    _native().release_scratch(buf)


def _records_log(d: dict) -> bool:
    # private key set by the autograd function for training forwards (see __init__._RasterizeGaussians.forward, which
    # applies the backward-mode policy above); mode "resort" forces the reference-style re-sorting backward everywhere
    return bool(d.get("_record_blend_log")) and d.get("_backward_mode", _backward_mode) != "resort"


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, inv_viewprojmatrix, tan_fovx, tan_fovy, image_height, image_width,
                        sh, degree, campos, prefiltered, settings_dict, render_depth, debug
                        ) -> Tuple[int, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """== RasterizeGaussiansCUDA (reference rasterize_points.cu:43-138).
    Returns (num_rendered, out_color (3,H,W), radii (P,) int32, geomBuffer, binningBuffer, imgBuffer)."""
    if means3D.dim() != 2 or means3D.size(1) != 3:   # (the reference's first check, rasterize_points.cu:69-71)
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    return (_host or _native()).rasterize_gaussians(
        background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
        inv_viewprojmatrix, tan_fovx, tan_fovy, int(image_height), int(image_width), sh, int(degree), campos, bool(prefiltered),
        settings_dict, bool(render_depth), bool(debug), _records_log(settings_dict))


def rasterize_gaussians_backward(background, means3D, radii, opacities, colors, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, inv_viewprojmatrix, tan_fovx, tan_fovy,
                                 pixel_colors, dL_dout_color, sh, degree, campos, geomBuffer, R, binningBuffer,
                                 imageBuffer, settings_dict, debug, phases=3, partial=None, chunk=None, outputs=None):
    """== RasterizeGaussiansBackwardCUDA (reference rasterize_points.cu:140-232).
    Returns (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations).

    Extension for tile-row sharding (not in the reference): phases=1 runs only the render half and
    returns its per-Gaussian partial sums as the library's (P,16) gradient records (stp_raster.h);
    phases=2 takes the records (after the caller's all-reduce) as `partial` and runs the preprocess half.
    Adding 4 to either selects COMPACT records, (P,9) with no padding: the tensor that is all-reduced as it is.
    chunk = (k, K) with phases=2: the per-Gaussian half on chunk k of K equal ranges of Gaussian ids (the shard all-reduces the records in
    K pieces and runs chunk k as soon as piece k has arrived); outputs = the tuple an earlier chunk returned (every chunk writes its rows)."""
    if chunk is not None:
        k, K = int(chunk[0]), int(chunk[1])
        if not (0 <= k < K <= 255) or (int(phases) & 1):
            raise ValueError("chunk = (k, K) with 0 <= k < K <= 255, per-Gaussian half only")
        phases = int(phases) | (K << 8) | (k << 16)
    out = (_host or _native()).rasterize_gaussians_backward(
        background, means3D, radii, opacities, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
        inv_viewprojmatrix, tan_fovx, tan_fovy, pixel_colors, dL_dout_color, sh, int(degree), campos, geomBuffer, int(R), binningBuffer,
        imageBuffer, settings_dict, bool(debug), _records_log(settings_dict), int(phases), partial, None if outputs is None else list(outputs))
    return out[0] if (int(phases) & 3) == 1 else tuple(out)


def mark_visible(means3D, viewmatrix, projmatrix) -> torch.Tensor:
    """== markVisible (reference rasterize_points.cu:234-253)."""
    return (_host or _native()).mark_visible(means3D, viewmatrix, projmatrix)


# ---- introspection helpers (tests / bench; not part of the reference surface) -----------------------
_GEOM_TYPES = {"depths": torch.float32, "clamped": torch.uint8, "radii": torch.int32, "rects2D": torch.float32,
               "means2D": torch.float32, "cov3D": torch.float32, "cov3D_inv": torch.float32,
               "conic_opacity": torch.float32, "rgb": torch.float32, "tiles_touched": torch.int32,
               "point_offsets": torch.int32}
_BIN_TYPES = {"point_list": torch.int32, "point_list_unsorted": torch.int32, "keys": torch.int64, "keys_unsorted": torch.int64}
_IMG_TYPES = {"final_T": torch.float32, "n_contrib": torch.int32, "ranges": torch.int32, "tile_flags": torch.int32,
              "blend_log": torch.int16}  # the last two exist only in a buffer of a recording forward


def timing_text(device=None) -> str:
    """The reference's `timings_text` (what its viewer displays), from the stages measured on `device` since timing_enable(True)."""
    buf = ctypes.create_string_buffer(512)
    with _on_device(device):
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
    """Named view of the first R entries of a sub-array.  (A run-ahead forward carves the buffer for the capacity it guessed, not for
    num_rendered: the library remembers which, stp_binning_layout_count.)"""
    off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
    L = _load()
    if binningBuffer.is_cuda:
        _native().forget_unless_same_storage(binningBuffer)   # (a clone / recycled address is described by the header it carries, not by the address)
    lay = int(L.stp_binning_layout_count(ctypes.c_void_p(binningBuffer.data_ptr()), int(R))) if int(R) > 0 else 0
    if lay < 0:
        _raise_last(lay)
    if L.stp_binning_layout(lay, name.encode(), ctypes.byref(off), ctypes.byref(cnt)) != 0:
        raise KeyError(name)
    n = cnt.value // lay * int(R) if lay > 0 else cnt.value
    return _view(binningBuffer, off.value, n, _BIN_TYPES[name])


def image_array(imgBuffer, W, H, name, tile_rows=None):
    """Named view into an image buffer.  tile_rows = (y0, y1): the buffer of a forward restricted to that tile-row window -- it holds the
    window's pixel rows / tiles only (element 0 = the window's first pixel / tile).  "blend_log" is reported at the depth the buffer was
    carved with (its header: blend_log_depth)."""
    off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
    L = _load()
    L.stp_image_layout_depth.argtypes = [ctypes.c_int] * 5 + [ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    y0, y1 = (0, 0) if tile_rows is None else (int(tile_rows[0]), int(tile_rows[1]))
    depth = blend_log_depth(imgBuffer) if name == "blend_log" else 0
    if L.stp_image_layout_depth(int(W), int(H), y0, y1, depth, name.encode(), ctypes.byref(off), ctypes.byref(cnt)) != 0:
        raise KeyError(name)
    return _view(imgBuffer, off.value, cnt.value, _IMG_TYPES[name])


def blend_log_depth(imgBuffer) -> int:
    """Records per pixel of the blend log in the image buffer of a recording forward (adaptive: include/stp_raster.h, stp_blend_log_bytes)."""
    L = _load()
    L.stp_blend_log_depth.argtypes = [ctypes.c_void_p]
    L.stp_blend_log_depth.restype = ctypes.c_int
    if imgBuffer.is_cuda:
        _native().forget_unless_same_storage(imgBuffer)
    d = int(L.stp_blend_log_depth(ctypes.c_void_p(imgBuffer.data_ptr())))
    if d < 0:
        _raise_last(d)
    return d


def set_run_ahead(mode) -> int:
    """Run-ahead forward: False / 0 = never, True / 1 = always, 2 = auto (small frames only; the default) -- include/stp_raster.h:
    stp_set_run_ahead.  Returns the previous mode."""
    L = _load()
    L.stp_set_run_ahead.argtypes = [ctypes.c_int]
    L.stp_set_run_ahead.restype = None
    L.stp_get_run_ahead.restype = ctypes.c_int
    prev = int(L.stp_get_run_ahead())
    L.stp_set_run_ahead(int(mode))
    return prev


def reset_size_guesses() -> None:
    """The next forward of every kind runs without a size guess (hand-over in the middle of the frame, exact binning request) -- tests."""
    L = _load()
    L.stp_reset_size_guesses.restype = None
    L.stp_reset_size_guesses()


def timing_enable(flag: bool) -> None:
    _load().stp_timing_enable(int(bool(flag)))


def hbm_probe(kind: str, dst, src, blocks: int = 4096, nontemporal: bool = False) -> None:
    """One launch of the library's float4 streaming kernel (`read` src, `write` dst, `copy` src -> dst) on torch's current stream: the
    box's HBM ceiling for a plain stream (bench.py `hbm_measured`)."""
    import torch
    t = src if src is not None else dst
    k = {"read": 0, "write": 1, "copy": 2}[kind] + (4 if nontemporal else 0)
    with _on_device(t.device):
        rc = _load().stp_hbm_probe(k, None if dst is None else dst.data_ptr(), None if src is None else src.data_ptr(),
                                   t.numel() * t.element_size(), int(blocks), torch.cuda.current_stream(t.device).cuda_stream)
    if rc < 0:
        raise RuntimeError(f"stp_hbm_probe failed ({rc})")


def timing_history(device=None, capacity=1024, host=False):
    """Stage times of the last (up to `capacity`) timed calls on `device`, one dict per forward(+backward) in chronological order;
    a stage that was not measured in a call is missing from its dict.  host=True: the launching thread's own time between the stage's two
    event records instead of the GPU interval (a long GPU interval with an equally long host one = the launches came late)."""
    arr = (ctypes.c_float * (6 * capacity))()
    with _on_device(device):
        n = (_load().stp_timing_history_host if host else _load().stp_timing_history)(arr, capacity)
    if n < 0:
        _raise_last(n)
    names = ("Preprocess", "Duplicate", "Sort", "Render", "BwdRender", "BwdPreprocess")
    return [{nm: float(arr[6 * k + i]) for i, nm in enumerate(names) if arr[6 * k + i] >= 0.0} for k in range(n)]


def timing_read(device=None):
    """Mean milliseconds of the stages timed on `device` (default: the current one) since timing_enable(True): Preprocess,
    Duplicate, Sort, Render, BwdRender, BwdPreprocess; -1 = not measured.  The library keeps one timer per device."""
    arr = (ctypes.c_float * 6)()
    with _on_device(device):
        rc = _load().stp_timing_read(arr)
    if rc < 0:
        _raise_last(rc)
    names = ("Preprocess", "Duplicate", "Sort", "Render", "BwdRender", "BwdPreprocess")
    return {n: float(v) for n, v in zip(names, arr)}
