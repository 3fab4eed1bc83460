"""Shared helpers of the parity tests: settings dicts, running the product (GPU) and the oracle (CPU)
on the same synthetic scene, and comparison metrics."""
from __future__ import annotations

import numpy as np


def settings_dict(mode=0, order=0, per_pixel=4, tile_2x2=8, rect=False, tight=False, tbc=False, h44=False,
                  lb=False, ewa=False):
    """The reference's ExtendedSettings.to_dict() layout."""
    return {"sort_settings": {"queue_sizes": {"tile_4x4": 64, "tile_2x2": tile_2x2, "per_pixel": per_pixel},
                              "sort_mode": mode, "sort_order": order},
            "culling_settings": {"rect_bounding": rect, "tight_opacity_bounding": tight, "tile_based_culling": tbc,
                                 "hierarchical_4x4_culling": h44},
            "load_balancing": lb, "proper_ewa_scaling": ewa}


FULL_STP = dict(mode=3, order=3, rect=True, tight=True, tbc=True, h44=True, lb=True)  # "full StopThePop" (C2-full)


def psnr(a, b):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 200.0 if mse == 0 else 10.0 * np.log10(1.0 / mse)


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if np.size(a) else 0.0


def ext_settings(d):
    """dict -> diff_gaussian_rasterization.ExtendedSettings"""
    import diff_gaussian_rasterization as dgr
    es = dgr.ExtendedSettings.from_dict(d)
    if d.get("_backward_mode"):   # (our per-call extension rides on the settings object, see ExtendedSettings.to_dict)
        es._backward_mode = d["_backward_mode"]
    return es


class GpuRun:
    """Runs one scene through the product's public API on cuda:0 and keeps what the tests inspect."""

    def __init__(self, scene, sdict, backward=True, device="cuda:0", tile_rows=None, debug=False, render_depth=False,
                 cov3D_precomp=None, prefiltered=False, warm=True, run_ahead=None):
        """cov3D_precomp: (P,6) array -> passed INSTEAD of scales/rotations (ref: __init__.py:286-289 allows exactly one).
        warm: the frame is rendered TWICE.  The library's default forward takes the reference's path -- hand-over of num_rendered in the
        middle of the frame, exact binning size; with run-ahead switched on (_C.set_run_ahead) every forward but the first of its kind is
        launched as a whole on a capacity guessed from the frames before (stp_api.hip).  The untracked first pass runs the default path,
        the second -- the one the tests inspect -- the run-ahead path; both must return the same image, radii and count, bit for bit.
        run_ahead (with warm=False): run this one frame with the switch set like this."""
        import torch
        import diff_gaussian_rasterization as dgr
        from diff_gaussian_rasterization import _C
        self.scene, self.sdict = scene, sdict
        dev = torch.device(device)
        t = lambda a, rg=False: None if a is None else torch.tensor(a, device=dev).requires_grad_(rg and backward)
        self.means3D, self.opac = t(scene.means3D, True), t(scene.opacities, True)
        self.scales, self.rots = t(scene.scales, True), t(scene.rotations, True)
        self.shs, self.colors = t(scene.shs, True), t(scene.colors_precomp, True)
        self.means2D = torch.zeros_like(self.means3D, requires_grad=backward)
        self.cov3D = t(cov3D_precomp, True)
        if self.cov3D is not None:
            self.scales = self.rots = None
        es = ext_settings(sdict)
        if tile_rows is not None:
            # tile-row window rides along in the dict through a private key (see _C.settings_from_dict)
            base = es.to_dict
            es.to_dict = lambda: {**base(), "_tile_rows": tuple(tile_rows)}
        rs = dgr.GaussianRasterizationSettings(
            image_height=scene.H, image_width=scene.W, tanfovx=scene.tanfovx, tanfovy=scene.tanfovy, bg=t(scene.bg),
            scale_modifier=scene.scale_modifier, viewmatrix=t(scene.viewmatrix), projmatrix=t(scene.projmatrix),
            inv_viewprojmatrix=t(scene.inv_viewprojmatrix), sh_degree=scene.sh_degree, campos=t(scene.campos),
            prefiltered=prefiltered, settings=es, render_depth=render_depth, debug=debug)
        self.rs = rs
        rast = dgr.GaussianRasterizer(rs)
        self.exact_pass = None
        self.run_ahead_before = None
        if warm and not debug:
            _C.reset_size_guesses()
            self.run_ahead_before = _C.set_run_ahead(0)
            empty_ = torch.Tensor([])
            e_ = lambda x: empty_ if x is None else x.detach()
            o = _C.rasterize_gaussians(rs.bg, self.means3D.detach(), e_(self.colors), self.opac.detach(), e_(self.scales), e_(self.rots),
                                       rs.scale_modifier, e_(self.cov3D), rs.viewmatrix, rs.projmatrix, rs.inv_viewprojmatrix, rs.tanfovx,
                                       rs.tanfovy, rs.image_height, rs.image_width, e_(self.shs), rs.sh_degree, rs.campos, prefiltered,
                                       es.to_dict(), render_depth, debug)
            self.exact_pass = (int(o[0]), o[1].clone(), o[2].clone())
            _C.release_scratch(o[5]); _C.release_scratch(o[4])
            del o
            _C.set_run_ahead(1)
        elif run_ahead is not None:
            self.run_ahead_before = _C.set_run_ahead(1 if run_ahead else 0)
        try:
            self._run(rast, rs, es, backward, dev, prefiltered, render_depth, debug)
        finally:
            if self.run_ahead_before is not None:
                _C.set_run_ahead(self.run_ahead_before)

    def _run(self, rast, rs, es, backward, dev, prefiltered, render_depth, debug):
        import torch
        from diff_gaussian_rasterization import _C
        scene = self.scene
        color, radii = rast(self.means3D, self.means2D, self.opac, shs=self.shs, colors_precomp=self.colors,
                            scales=self.scales, rotations=self.rots, cov3D_precomp=self.cov3D)
        self.color_t = color
        self.color = color.detach().cpu().numpy()
        self.radii = radii.cpu().numpy()
        if self.exact_pass is not None:   # exact path == run-ahead path, bit for bit (NaN == NaN: render_depth of an empty frame)
            if not torch.equal(torch.nan_to_num(self.exact_pass[1], nan=-1.0), torch.nan_to_num(color.detach(), nan=-1.0)):
                a_, b_ = self.exact_pass[1].cpu().numpy(), color.detach().cpu().numpy()
                bad = np.argwhere((a_ != b_) & ~(np.isnan(a_) & np.isnan(b_)))
                raise AssertionError(f"run-ahead forward differs from the exact one: {len(bad)} values, rows {bad[:, 1].min()}..{bad[:, 1].max()}, cols {bad[:, 2].min()}..{bad[:, 2].max()}, "
                                     f"exact {a_[tuple(bad[0])]} run-ahead {b_[tuple(bad[0])]}, counts {self.exact_pass[0]} / {getattr(color.grad_fn, 'num_rendered', None)}")
            assert torch.equal(self.exact_pass[2], radii)
        fn = color.grad_fn
        if fn is not None:
            self.num_rendered = fn.num_rendered
            assert self.exact_pass is None or self.exact_pass[0] == self.num_rendered
            saved = fn.saved_tensors
            self.geom, self.binning, self.img = saved[9], saved[10], saved[11]
        else:  # forward-only run (no tensor requires grad): fetch the scratch buffers with a direct _C call
            empty = torch.Tensor([])
            e = lambda x: empty if x is None else x
            out = _C.rasterize_gaussians(rs.bg, self.means3D, e(self.colors), self.opac, e(self.scales), e(self.rots), rs.scale_modifier,
                                         e(self.cov3D), rs.viewmatrix, rs.projmatrix, rs.inv_viewprojmatrix, rs.tanfovx, rs.tanfovy,
                                         rs.image_height, rs.image_width, e(self.shs), rs.sh_degree, rs.campos, prefiltered,
                                         es.to_dict(), render_depth, debug)
            self.num_rendered, color2 = out[0], out[1]
            assert torch.equal(torch.nan_to_num(color2, nan=-1.0), torch.nan_to_num(color, nan=-1.0))  # (render_depth of an empty frame is NaN, as in the reference)
            self.geom, self.binning, self.img = out[3], out[4], out[5]
        self._C = _C
        self.grads = None
        if backward:
            w = torch.tensor(scene.dL_dout, device=dev)
            (color * w).sum().backward()
            g = lambda x: None if x is None or x.grad is None else x.grad.detach().cpu().numpy()
            self.grads = dict(dL_dmeans3D=g(self.means3D), dL_dmeans2D=g(self.means2D), dL_dopacity=g(self.opac),
                              dL_dscales=g(self.scales), dL_drotations=g(self.rots), dL_dsh=g(self.shs),
                              dL_dcolors=g(self.colors), dL_dcov3D=g(self.cov3D))

    def geom_array(self, name):
        return self._C.geometry_array(self.geom, self.scene.P, self.sdict, name).cpu().numpy()

    def binning_array(self, name):
        a = self._C.binning_array(self.binning, self.num_rendered, name).cpu().numpy()
        return a.view(np.uint64) if a.dtype == np.int64 else a.view(np.uint32)

    def image_array(self, name):
        return self._C.image_array(self.img, self.scene.W, self.scene.H, name).cpu().numpy()


def oracle_run(scene, sdict, backward=True, tile_rows=None, render_depth=False, cov3D_precomp=None):
    from oracle import oracle as orc
    f = orc.forward_scene(scene, sdict, tile_rows=tile_rows, render_depth=render_depth, cov3D_precomp=cov3D_precomp)
    g = f.backward(scene.dL_dout) if backward else None
    return f, g
