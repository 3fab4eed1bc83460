#!/usr/bin/env python3
"""Trainer-side caller: the `render()` function a 3DGS / StopThePop trainer wraps around the rasterizer (upstream
`gaussian_renderer/__init__.py`), written against this package, plus a tiny optimisation
loop on a synthetic scene that shows the forward + backward of the hot path in its natural habitat.

    PYTHONPATH=stopthepop-rasterization_amd python examples/train_render.py [--iters 30] [--config full|min|kbuffer|global]

`render()` takes the trainer's usual objects by duck typing:
  camera : image_width, image_height, FoVx, FoVy, world_view_transform, full_proj_transform, camera_center
  model  : get_xyz, get_opacity, get_scaling, get_rotation, get_features, active_sh_degree
and returns the trainer's usual dict (render, viewspace_points, visibility_filter, radii).
"""
from __future__ import annotations

import argparse
import math
import os
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stopthepop-rasterization_amd"))
from diff_gaussian_rasterization import (CullingSettings, ExtendedSettings, GaussianRasterizationSettings,  # noqa: E402
                                         GaussianRasterizer, GlobalSortOrder, SortMode, SortQueueSizes, SortSettings, scenes)


def render(camera, model, bg_color: torch.Tensor, splat_args: ExtendedSettings, scaling_modifier: float = 1.0,
           override_color: torch.Tensor | None = None, render_depth: bool = False, debug: bool = False):
    """One frame.  Gradients flow to every model tensor; `viewspace_points.grad` is the screen-space positional
    gradient densification uses."""
    screenspace_points = torch.zeros_like(model.get_xyz, requires_grad=True)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(camera.image_height), image_width=int(camera.image_width),
        tanfovx=math.tan(camera.FoVx * 0.5), tanfovy=math.tan(camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=camera.world_view_transform, projmatrix=camera.full_proj_transform,
        inv_viewprojmatrix=camera.full_proj_transform.inverse(), sh_degree=model.active_sh_degree,
        campos=camera.camera_center, prefiltered=False, settings=splat_args, render_depth=render_depth, debug=debug)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    shs, colors = (None, override_color) if override_color is not None else (model.get_features, None)
    image, radii = rasterizer(means3D=model.get_xyz, means2D=screenspace_points, shs=shs, colors_precomp=colors,
                              opacities=model.get_opacity, scales=model.get_scaling, rotations=model.get_rotation)
    return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}


def splat_config(name: str) -> ExtendedSettings:
    """The settings files the reference ships (configs/*.json), as objects."""
    hier = lambda order: SortSettings(queue_sizes=SortQueueSizes(64, 8, 4), sort_mode=SortMode.HIER, sort_order=order)
    if name == "full":  # hierarchical resort + every culling option + per-tile depth (the paper's default)
        return ExtendedSettings(sort_settings=hier(GlobalSortOrder.PTD_MAX), culling_settings=CullingSettings(True, True, True, True),
                                load_balancing=True, proper_ewa_scaling=False)
    if name == "min":
        return ExtendedSettings(sort_settings=hier(GlobalSortOrder.Z_DEPTH))
    if name == "kbuffer":
        return ExtendedSettings(sort_settings=SortSettings(queue_sizes=SortQueueSizes(64, 8, 16), sort_mode=SortMode.PPX_KBUFFER))
    return ExtendedSettings()  # plain 3DGS: global sort by view-space z


class ToyGaussians(torch.nn.Module):
    """The part of the trainer's GaussianModel that render() touches (activations included)."""

    def __init__(self, sc: scenes.Scene, device):
        super().__init__()
        t = lambda a: torch.nn.Parameter(torch.tensor(a, device=device))
        self._xyz = t(sc.means3D)
        self._scaling = torch.nn.Parameter(torch.log(torch.tensor(sc.scales, device=device)))
        self._rotation = t(sc.rotations)
        op = torch.tensor(sc.opacities, device=device).clamp(1e-4, 1 - 1e-4)
        self._opacity = torch.nn.Parameter(torch.log(op / (1 - op)))
        self._features = t(sc.shs)
        self.active_sh_degree = sc.sh_degree

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_features = property(lambda s: s._features)


def camera_of(sc: scenes.Scene, device):
    t = lambda a: torch.tensor(a, device=device)
    return SimpleNamespace(image_width=sc.W, image_height=sc.H, FoVx=2 * math.atan(sc.tanfovx), FoVy=2 * math.atan(sc.tanfovy),
                           world_view_transform=t(sc.viewmatrix), full_proj_transform=t(sc.projmatrix), camera_center=t(sc.campos))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--config", default="full", choices=["full", "min", "kbuffer", "global"])
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--size", type=int, nargs=2, default=[320, 240], metavar=("W", "H"))
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("this example needs a GPU (the rasterizer has no CPU path)")
    dev = torch.device("cuda:0")
    W, H = args.size
    target_scene = scenes.make_scene(P=args.points, W=W, H=H, sigma_min=1.0, sigma_max=9.0, seed=3, camera="orbit")
    cam, bg, cfg = camera_of(target_scene, dev), torch.tensor(target_scene.bg, device=dev), splat_config(args.config)
    with torch.no_grad():
        target = render(cam, ToyGaussians(target_scene, dev), bg, cfg)["render"]

    # start from perturbed colours and opacities and fit them back
    model = ToyGaussians(target_scene, dev)
    with torch.no_grad():
        model._features.mul_(0.3)
        model._opacity.sub_(1.0)
    opt = torch.optim.Adam([{"params": [model._features], "lr": 2e-2}, {"params": [model._opacity], "lr": 5e-2},
                            {"params": [model._xyz, model._scaling, model._rotation], "lr": 0.0}])
    first = last = None
    for it in range(args.iters):
        out = render(cam, model, bg, cfg)
        loss = (out["render"] - target).abs().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        grad2d = out["viewspace_points"].grad  # what densification accumulates
        opt.step()
        last = float(loss.detach())
        first = last if first is None else first
        if it % 10 == 0 or it == args.iters - 1:
            print(f"iter {it:3d}  L1 {last:.5f}  visible {int(out['visibility_filter'].sum())}  |grad2D| max {float(grad2d.norm(dim=1).max()):.3e}")
    with torch.no_grad():
        depth = render(cam, model, bg, cfg, render_depth=True)["render"]
    print(f"L1 {first:.5f} -> {last:.5f}; depth visualisation {tuple(depth.shape)} in [{float(depth.min()):.3f}, {float(depth.max()):.3f}]")
    return first, last


if __name__ == "__main__":
    main()
