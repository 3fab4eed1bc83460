"""Independent float64 PyTorch reference of the differentiable splatting maths (dense, tiny scenes).

This is NOT the oracle and shares no code with it or with the HIP kernels: it is written from the
textbook formulation (Sigma = R S^2 R^T, EWA projection J W Sigma W^T J^T, real SH basis, alpha
compositing with a per-pixel order) and differentiated by torch.autograd.  It pins the *value and
gradient maths* of oracle and kernels; discrete behaviour (tile binning, queue cadence) is pinned
elsewhere.  Ordering modes: "global" (sort by the global depth key) and "exact" (sort by each
pixel's own depth-along-ray, i.e. what the resorting modes converge to at low density).
"""
from __future__ import annotations

import math

import numpy as np
import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    return R  # textbook rotation matrix (row-major)


def eval_sh(deg, sh, dirs):
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
               + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
               + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res + 0.5


def render(scene, order="global", proper_ewa_scaling=False, use_cov3D_precomp=False, depth_key="z"):
    """Returns (image (3,H,W) float64 tensor, dict of leaf tensors with requires_grad)."""
    dd = torch.float64
    t = lambda a: torch.tensor(np.asarray(a), dtype=dd)
    W, H = scene.W, scene.H
    V, PM, INV = t(scene.viewmatrix), t(scene.projmatrix), t(scene.inv_viewprojmatrix)
    cam, bg = t(scene.campos), t(scene.bg)
    leaves = {}
    means = t(scene.means3D).requires_grad_(True); leaves["means3D"] = means
    opac = t(scene.opacities).requires_grad_(True); leaves["opacities"] = opac
    scales = t(scene.scales).requires_grad_(True); leaves["scales"] = scales
    rots = t(scene.rotations).requires_grad_(True); leaves["rotations"] = rots
    P = means.shape[0]
    ndc_shift = torch.zeros(P, 2, dtype=dd, requires_grad=True); leaves["means2D"] = ndc_shift

    Rm = quat_to_rot(rots)
    Sd = torch.diag_embed((scene.scale_modifier * scales) ** 2)
    Sigma = Rm @ Sd @ Rm.transpose(1, 2)
    if use_cov3D_precomp:
        c6 = torch.stack([Sigma[:, 0, 0], Sigma[:, 0, 1], Sigma[:, 0, 2], Sigma[:, 1, 1], Sigma[:, 1, 2], Sigma[:, 2, 2]], 1)
        c6 = c6.detach().clone().requires_grad_(True); leaves["cov3D_precomp"] = c6
        Sigma_used = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4], c6[:, 2], c6[:, 4], c6[:, 5]], 1).reshape(-1, 3, 3)
    else:
        Sigma_used = Sigma

    pv = means @ V[:3, :3] + V[3, :3]
    tz = pv[:, 2]
    near_ok = (tz > 0.2).detach()
    fx, fy = W / (2 * scene.tanfovx), H / (2 * scene.tanfovy)
    limx, limy = 1.3 * scene.tanfovx, 1.3 * scene.tanfovy
    txc = torch.clamp(pv[:, 0] / tz, -limx, limx) * tz
    tyc = torch.clamp(pv[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * txc / (tz * tz), zero, fy / tz, -fy * tyc / (tz * tz)], 1).reshape(-1, 2, 3)
    Wm = V[:3, :3].T  # p_view = Wm p + t
    JW = J @ Wm
    cov2 = JW @ Sigma_used @ JW.transpose(1, 2)
    a0, b0, c0 = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 1]
    a, b, c = a0 + 0.3, b0, c0 + 0.3
    det = a * c - b * b
    o = opac[:, 0]
    if proper_ewa_scaling:
        o = o * torch.sqrt(torch.clamp((a0 * c0 - b0 * b0) / det, min=0.000025))
    cA, cB, cC = c / det, -b / det, a / det

    ph = torch.cat([means, torch.ones(P, 1, dtype=dd)], 1) @ PM
    ndc = ph[:, :2] / (ph[:, 3:4] + 1e-7) + ndc_shift
    mx = ((ndc[:, 0] + 1) * W - 1) * 0.5
    my = ((ndc[:, 1] + 1) * H - 1) * 0.5

    if scene.shs is not None:
        shs = t(scene.shs).requires_grad_(True); leaves["shs"] = shs
        d = means - cam
        d = d / d.norm(dim=1, keepdim=True)
        col = torch.clamp(eval_sh(scene.sh_degree, shs, d), min=0.0)
    else:
        col = t(scene.colors_precomp).requires_grad_(True); leaves["colors_precomp"] = col

    # binning, non-differentiable: 3.33 sigma rectangle of tiles (no culling options here)
    with torch.no_grad():
        mid = 0.5 * (a + c)
        lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.01))
        radius = 3.33 * torch.sqrt(lam)
        visible = near_ok & (det != 0) & (o >= 1.0 / 255.0)
        x0 = torch.clamp(torch.floor((mx - radius) / 16), 0, (W + 15) // 16)
        x1 = torch.clamp(torch.ceil((mx + radius) / 16), 0, (W + 15) // 16)
        y0 = torch.clamp(torch.floor((my - radius) / 16), 0, (H + 15) // 16)
        y1 = torch.clamp(torch.ceil((my + radius) / 16), 0, (H + 15) // 16)
        visible &= ((x1 - x0) * (y1 - y0)) > 0

    ys, xs = torch.meshgrid(torch.arange(H, dtype=dd), torch.arange(W, dtype=dd), indexing="ij")
    px, py = xs.reshape(-1), ys.reshape(-1)             # N
    dx = mx[None, :] - px[:, None]                      # N x P
    dy = my[None, :] - py[:, None]
    power = -0.5 * (cA[None] * dx * dx + cC[None] * dy * dy) - cB[None] * dx * dy
    G = torch.exp(torch.clamp(power, max=0.0))
    alpha = torch.clamp(o[None] * G, max=0.99)
    with torch.no_grad():
        tx_, ty_ = torch.floor(px / 16), torch.floor(py / 16)
        in_rect = (tx_[:, None] >= x0[None]) & (tx_[:, None] < x1[None]) & (ty_[:, None] >= y0[None]) & (ty_[:, None] < y1[None])
        keep = in_rect & visible[None] & (power <= 0) & (alpha >= 1.0 / 255.0)
        if order == "global":
            key = (tz if depth_key == "z" else (means - cam).norm(dim=1))[None].expand(px.shape[0], P)
        else:
            # depth along each pixel's ray: (Sigma^-1 (mu - cam)) . v / (v^T Sigma^-1 v)
            s_cl = torch.clamp(scales, min=1e-3) * scene.scale_modifier
            Sinv = Rm @ torch.diag_embed(1.0 / (s_cl ** 2)) @ Rm.transpose(1, 2)
            ndcx, ndcy = px * (2.0 / W) - 1.0, py * (2.0 / H) - 1.0
            pw = ndcx[:, None] * INV[0][None] + ndcy[:, None] * INV[1][None] + INV[3][None]
            pw = pw[:, :3] / pw[:, 3:4]
            v = pw - cam
            v = v / v.norm(dim=1, keepdim=True)            # N x 3
            u = torch.einsum("pij,pj->pi", Sinv, means - cam)
            num = v @ u.T                                   # N x P
            den = torch.einsum("ni,pij,nj->np", v, Sinv, v)
            key = num / torch.clamp(den, min=1e-5)
            keep &= key >= 0
        key = torch.where(keep, key, torch.full_like(key, float("inf")))
        idx = torch.argsort(key, dim=1, stable=True)
    a_s = torch.gather(torch.where(keep, alpha, torch.zeros_like(alpha)), 1, idx)
    one_m = 1 - a_s
    Tbefore = torch.cumprod(torch.cat([torch.ones(a_s.shape[0], 1, dtype=dd), one_m[:, :-1]], 1), 1)
    with torch.no_grad():
        stop = (Tbefore * one_m) < 1e-4                   # first saturating entry ends the pixel
        alive = torch.cumsum(stop.to(torch.int64), 1) == 0
    wgt = torch.where(alive, a_s * Tbefore, torch.zeros_like(a_s))
    col_s = col[idx]                                      # N x P x 3
    C = (wgt[..., None] * col_s).sum(1)
    T_final = torch.where(alive, one_m, torch.ones_like(one_m)).prod(1)
    img = C + T_final[:, None] * bg[None]
    return img.T.reshape(3, H, W), leaves


def loss_and_grads(scene, **kw):
    img, leaves = render(scene, **kw)
    w = torch.tensor(scene.dL_dout, dtype=torch.float64)
    loss = (img * w).sum()
    names = list(leaves.keys())
    grads = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    out = {n: (None if g is None else g.detach().numpy()) for n, g in zip(names, grads)}
    return img.detach().numpy(), out
