"""Worker of tests/test_multi_gpu.py (one process per rank): the HIP product under TileRowShardedRasterizer against the
unsharded HIP run of the same frame in the same process.
    argv: backend ("gloo": both ranks on cuda:0, exchange staged through the host | "nccl": one GPU per rank, RCCL)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import conftest  # noqa: F401,E402
from helpers import FULL_STP, GpuRun, ext_settings, settings_dict  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402
from diff_gaussian_rasterization import scenes, tile_shard  # noqa: E402

backend = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", rank if backend == "nccl" else 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
rel = lambda a, b: float(np.max(np.abs(a - b))) / max(float(np.max(np.abs(b))), 1e-30)


def run_sharded(sc, sd, backward, rebalance=False, frames=1):
    t = lambda a, rg=False: None if a is None else torch.tensor(a, device=dev).requires_grad_(rg and backward)
    leaves = dict(means3D=t(sc.means3D, True), opac=t(sc.opacities, True), scales=t(sc.scales, True), rots=t(sc.rotations, True), shs=t(sc.shs, True))
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=backward)
    rs = dgr.GaussianRasterizationSettings(
        image_height=sc.H, image_width=sc.W, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, bg=t(sc.bg), scale_modifier=sc.scale_modifier,
        viewmatrix=t(sc.viewmatrix), projmatrix=t(sc.projmatrix), inv_viewprojmatrix=t(sc.inv_viewprojmatrix), sh_degree=sc.sh_degree,
        campos=t(sc.campos), prefiltered=False, settings=ext_settings(sd), render_depth=False, debug=False)
    r = tile_shard.TileRowShardedRasterizer(rs, dist, rank, world, rebalance=rebalance)
    for _ in range(frames):
        for x in list(leaves.values()) + [means2D]:
            x.grad = None
        color, radii = r(leaves["means3D"], means2D, leaves["opac"], shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rots"])
        if backward:  # every rank back-propagates the loss gradient of ITS rows (here: the same dL/dimage everywhere; rows of
            # other ranks are not in this rank's graph -- the backward only walks its own tiles)
            (color * torch.tensor(sc.dL_dout, device=dev)).sum().backward()
    g = None
    if backward:
        gr = lambda x: None if x.grad is None else x.grad.detach().cpu().numpy()
        g = dict(dL_dmeans3D=gr(leaves["means3D"]), dL_dmeans2D=gr(means2D), dL_dopacity=gr(leaves["opac"]), dL_dscales=gr(leaves["scales"]),
                 dL_drotations=gr(leaves["rots"]), dL_dsh=gr(leaves["shs"]))
    return color.detach().cpu().numpy(), radii.cpu().numpy(), g, r.parts


cases = [("hier_full_bwd", dict(P=5000, W=160, H=200, sigma_min=1.5, sigma_max=12.0, seed=51, camera="orbit"), settings_dict(**FULL_STP), True, False, 1),
         ("kbuffer_bwd", dict(P=3000, W=100, H=90, sigma_min=1.5, sigma_max=10.0, seed=52), settings_dict(2, per_pixel=16), True, False, 1),
         ("c4_like_fwd", dict(P=4000, W=384, H=216, sigma_min=1.0, sigma_max=12.0, seed=4), settings_dict(**FULL_STP), False, False, 1),
         ("rebalanced_bwd", dict(P=4000, W=128, H=256, sigma_min=1.5, sigma_max=10.0, seed=53, camera="orbit"), settings_dict(**FULL_STP), True, True, 3)]
for name, skw, sd, backward, rebalance, frames in cases:
    sc = scenes.make_scene(**skw)
    if rebalance:  # content only in the upper part of the frame: equal-height blocks would be very unequal work
        sc.opacities[(sc.means3D @ sc.viewmatrix[:3, 1] + sc.viewmatrix[3, 1]) > 0.0] = 0.0   # everything below the optical axis becomes invisible
    img, radii, g, parts = run_sharded(sc, sd, backward, rebalance, frames)
    full = GpuRun(sc, sd, backward=backward, device=str(dev))
    assert np.array_equal(radii, full.radii), name
    if rank == 0:
        assert np.array_equal(img, full.color), (name, float(np.abs(img - full.color).max()))   # the blend of a tile does not depend on who runs it
    if backward:
        for k, b in full.grads.items():
            if b is None or g.get(k) is None:
                continue
            a = g[k]
            if k == "dL_dmeans2D":
                a, b = a[:, :2], b[:, :2]
            assert rel(a, b) < 1e-5, (name, k, rel(a, b))
    if rebalance and rank == 0:
        h = [b - a for a, b in parts]
        assert h[0] != h[-1] or world == 1, (name, parts)   # the partition moved away from equal heights
    dist.barrier()
try:   # the depth visualisation is refused (seams), loudly
    sc = scenes.make_scene(P=200, W=64, H=64, sigma_min=1.0, sigma_max=6.0, seed=3)
    t = lambda a: torch.tensor(a, device=dev)
    rs = dgr.GaussianRasterizationSettings(image_height=64, image_width=64, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, bg=t(sc.bg), scale_modifier=1.0,
        viewmatrix=t(sc.viewmatrix), projmatrix=t(sc.projmatrix), inv_viewprojmatrix=t(sc.inv_viewprojmatrix), sh_degree=3, campos=t(sc.campos),
        prefiltered=False, settings=ext_settings(settings_dict(3)), render_depth=True, debug=False)
    tile_shard.TileRowShardedRasterizer(rs, dist, rank, world)(t(sc.means3D), torch.zeros(200, 3, device=dev), t(sc.opacities), shs=t(sc.shs), scales=t(sc.scales), rotations=t(sc.rotations))
    raise SystemExit("render_depth under sharding did not raise")
except RuntimeError as e:
    assert "render_depth is not available with tile-row sharding" in str(e)
if rank == 0:
    print("SHARD_GPU_OK", backend)
dist.destroy_process_group()
