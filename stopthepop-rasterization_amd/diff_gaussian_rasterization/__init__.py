"""diff_gaussian_rasterization -- MI355X-native drop-in for the StopThePop rasterizer's Python API.

Public surface (names, argument order, field order, defaults, error texts) follows the reference
package of the same name (reference diff_gaussian_rasterization/__init__.py):
  rasterize_gaussians            :32-53      _RasterizeGaussians (autograd.Function)  :55-172
  SortMode / GlobalSortOrder     :175-191    SortQueueSizes / SortSettings / CullingSettings / ExtendedSettings :193-246
  GaussianRasterizationSettings  :248-263    GaussianRasterizer (nn.Module)           :265-314
A trainer written against the reference imports this package unchanged.  The compute behind `_C` is the
hand-written HIP library (csrc/), reached through its C ABI; see _C.py.

Differences that do not change behaviour: the settings dataclasses use default_factory (the
reference's shared mutable defaults are rejected by Python >= 3.11); `dacite` is optional (only
ExtendedSettings.from_dict used it).
"""
from __future__ import annotations

import json
from dataclasses import asdict, dataclass, field, fields
from enum import IntEnum
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C

__all__ = ["rasterize_gaussians", "SortMode", "GlobalSortOrder", "SortQueueSizes", "SortSettings", "CullingSettings",
           "ExtendedSettings", "GaussianRasterizationSettings", "GaussianRasterizer"]


def enum_dict_factory(data):
    """asdict() factory that stores IntEnum members as plain ints (the C side reads ints)."""
    return {k: (v.value if isinstance(v, IntEnum) else v) for k, v in data}


def cpu_deep_copy_tuple(input_tuple):
    return tuple(item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        rs = raster_settings
        sdict = rs.settings.to_dict()
        ctx.log_lease = None
        if any(ctx.needs_input_grad) and not rs.render_depth:
            # a backward can follow: let the hierarchical / k-buffer forward record each pixel's blend order so that the
            # backward replays it instead of re-sorting (extension of ours; ignored by the other sort modes) -- unless the
            # backward-mode policy says otherwise (_C.set_backward_mode / STP_BACKWARD / settings._backward_mode: "resort",
            # or "auto" with the device's blend-log budget used up by forwards that still wait for their backward).
            # Not with render_depth: the depth-visualisation forward records no log, and a backward through it
            # (meaningless in the reference too, but memory-safe there) must take the re-sorting path.
            mode = sdict.get("_backward_mode")
            uses_log = int(sdict["sort_settings"]["sort_mode"]) in (2, 3)
            if uses_log and means3D.is_cuda and means3D.size(0) != 0 and _C.decide_recording(mode, means3D.device, rs.image_width, rs.image_height):
                sdict["_record_blend_log"] = True
                sdict["_backward_mode"] = "replay"
                ctx.log_lease = _C.LogLease(_C._device_index(means3D.device), _C.blend_log_bytes(rs.image_width, rs.image_height))
            else:
                sdict["_backward_mode"] = "resort"
        ctx.settings_dict = sdict
        # positional layout of _C.rasterize_gaussians (22 arguments)
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.inv_viewprojmatrix, rs.tanfovx, rs.tanfovy, rs.image_height,
                rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, sdict, rs.render_depth,
                rs.debug)
        if rs.debug:
            cpu_args = cpu_deep_copy_tuple(args)  # snapshot before anything can corrupt them
            try:
                num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians(*args)

        if ctx.log_lease is not None:   # the library chose the log's depth for this frame: account what the buffer really holds
            ctx.log_lease.resize(_C.blend_log_bytes(rs.image_width, rs.image_height, depth=_C.blend_log_depth(imgBuffer)))
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.img_generation = _C.scratch_generation(imgBuffer)
        ctx.bin_generation = _C.scratch_generation(binningBuffer)
        ctx.save_for_backward(colors_precomp, means3D, opacities, scales, rotations, cov3Ds_precomp, radii, sh, color,
                              geomBuffer, binningBuffer, imgBuffer)
        # radii is an integer output: without these two lines autograd materialises a (P,) zero "gradient" for it in
        # every backward (a 4 MB fill kernel per step at 1 M Gaussians)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        num_rendered = ctx.num_rendered
        rs = ctx.raster_settings
        (colors_precomp, means3D, opacities, scales, rotations, cov3Ds_precomp, radii, sh, color, geomBuffer,
         binningBuffer, imgBuffer) = ctx.saved_tensors
        _C.check_scratch(imgBuffer, ctx.img_generation)
        _C.check_scratch(binningBuffer, ctx.bin_generation)
        if grad_out_color is None:  # (set_materialize_grads(False): cannot happen while the image is the only differentiable output)
            grad_out_color = torch.zeros_like(color)
        # positional layout of _C.rasterize_gaussians_backward (25 arguments)
        args = (rs.bg, means3D, radii, opacities, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.inv_viewprojmatrix, rs.tanfovx, rs.tanfovy, color, grad_out_color, sh,
                rs.sh_degree, rs.campos, geomBuffer, num_rendered, binningBuffer, imgBuffer, ctx.settings_dict,
                rs.debug)
        if rs.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                out = _C.rasterize_gaussians_backward(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            out = _C.rasterize_gaussians_backward(*args)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = out
        _C.release_scratch(imgBuffer); _C.release_scratch(binningBuffer)  # the blend log goes back to the library's free list
        if ctx.log_lease is not None:
            ctx.log_lease.release()
        # one gradient per forward input, in forward's order
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, None)


class SortMode(IntEnum):
    GLOBAL = 0
    PPX_FULL = 1
    PPX_KBUFFER = 2
    HIER = 3

    def __str__(self):
        return self.name


class GlobalSortOrder(IntEnum):
    Z_DEPTH = 0
    DISTANCE = 1
    PTD_CENTER = 2
    PTD_MAX = 3

    def __str__(self):
        return self.name


class _Settable:
    """set_value(key, value): set an own field, otherwise hand the key down (reference :199-246)."""
    _children = ()

    def set_value(self, key, value):
        if key in {f.name for f in fields(self)}:
            setattr(self, key, value)
        else:
            for child in self._children:
                getattr(self, child).set_value(key, value)


@dataclass
class SortQueueSizes(_Settable):
    tile_4x4: int = 64
    tile_2x2: int = 8
    per_pixel: int = 4


@dataclass
class SortSettings(_Settable):
    queue_sizes: SortQueueSizes = field(default_factory=SortQueueSizes)
    sort_mode: SortMode = SortMode.GLOBAL
    sort_order: GlobalSortOrder = GlobalSortOrder.Z_DEPTH
    _children = ("queue_sizes",)


@dataclass
class CullingSettings(_Settable):
    rect_bounding: bool = False
    tight_opacity_bounding: bool = False
    tile_based_culling: bool = False
    hierarchical_4x4_culling: bool = False


@dataclass
class ExtendedSettings(_Settable):
    sort_settings: SortSettings = field(default_factory=SortSettings)
    culling_settings: CullingSettings = field(default_factory=CullingSettings)
    load_balancing: bool = False
    proper_ewa_scaling: bool = False
    _children = ("culling_settings", "sort_settings")

    def to_dict(self):
        # same result as asdict(self, dict_factory=enum_dict_factory) (the reference's form, __init__.py:231-233), written out:
        # dataclasses.asdict deep-copies every leaf, 50 us per call -- a visible part of a small frame's host time
        ss, cs, q = self.sort_settings, self.culling_settings, self.sort_settings.queue_sizes
        as_int = lambda v: v.value if isinstance(v, IntEnum) else v
        return {"sort_settings": {"queue_sizes": {"tile_4x4": q.tile_4x4, "tile_2x2": q.tile_2x2, "per_pixel": q.per_pixel},
                                  "sort_mode": as_int(ss.sort_mode), "sort_order": as_int(ss.sort_order)},
                "culling_settings": {"rect_bounding": cs.rect_bounding, "tight_opacity_bounding": cs.tight_opacity_bounding,
                                     "tile_based_culling": cs.tile_based_culling, "hierarchical_4x4_culling": cs.hierarchical_4x4_culling},
                "load_balancing": self.load_balancing, "proper_ewa_scaling": self.proper_ewa_scaling,
                # (extension, not a dataclass field: `settings._backward_mode = "replay" | "resort" | "auto"` overrides the
                # process-wide backward-mode policy of _C.set_backward_mode for the calls made with this settings object)
                **({"_backward_mode": self._backward_mode} if getattr(self, "_backward_mode", None) else {})}

    def to_json(self):
        return json.dumps(self.to_dict())

    @staticmethod
    def from_dict(dict):
        try:
            import dacite
            return dacite.from_dict(data_class=ExtendedSettings, data=dict, config=dacite.Config(cast=[IntEnum]))
        except ImportError:
            ss, cs = dict["sort_settings"], dict["culling_settings"]
            return ExtendedSettings(
                sort_settings=SortSettings(queue_sizes=SortQueueSizes(**ss["queue_sizes"]), sort_mode=SortMode(ss["sort_mode"]),
                                           sort_order=GlobalSortOrder(ss["sort_order"])),
                culling_settings=CullingSettings(**cs), load_balancing=dict["load_balancing"],
                proper_ewa_scaling=dict["proper_ewa_scaling"])

    @staticmethod
    def from_json(json_filename):
        return ExtendedSettings.from_dict(json.load(open(json_filename)))


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    inv_viewprojmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    settings: ExtendedSettings
    render_depth: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean mask of the points that pass the camera's near-plane test."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = lambda t: torch.Tensor([]) if t is None else t  # absent optional input == empty CPU tensor
        return rasterize_gaussians(means3D, means2D, empty(shs), empty(colors_precomp), opacities, empty(scales),
                                   empty(rotations), empty(cov3D_precomp), rs)
