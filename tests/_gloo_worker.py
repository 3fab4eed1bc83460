"""Worker of test_two_rank_gloo_gather_and_gradient_allreduce (launched once per rank)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import conftest  # noqa: F401,E402
from helpers import FULL_STP, settings_dict  # noqa: E402
from diff_gaussian_rasterization import scenes, tile_shard  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (test-side compute only)

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
sc = scenes.make_scene(P=1200, W=80, H=104, sigma_min=1.5, sigma_max=12.0, seed=13, camera="orbit")
sd = settings_dict(**FULL_STP)
parts = tile_shard.row_partition(tile_shard.tile_rows(sc.H), world)
local = orc.forward_scene(sc, sd, tile_rows=parts[rank])
img = tile_shard.gather_image(torch.from_numpy(local.color), parts, rank, world, dist, dst=0)
img_all = tile_shard.gather_image(torch.from_numpy(local.color), parts, rank, world, dist, to_all=True)
# the strips in two halves (what the GPU path does around its split render launch; on host tensors: two batches, nothing to overlap)
img_halves = tile_shard.gather_image(torch.from_numpy(local.color.copy()), parts, rank, world, dist, dst=0, halves=(None, None))
assert all(0 < tile_shard.split_row(p) - p[0] < p[1] - p[0] for p in parts)
g = local.backward(sc.dL_dout)  # with a window the oracle's render half only sees this rank's rows
buf = tile_shard.pack_partials(*(torch.from_numpy(g[k]) for k in ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors")))
dist.all_reduce(buf)
m2d, conic, opac, col = tile_shard.unpack_partials(buf)
full = orc.forward_scene(sc, sd)
gfull = full.backward(sc.dL_dout)
assert np.array_equal(img_all.numpy(), full.color)
assert np.array_equal(local.radii, full.radii)
rel = lambda a, b: float(np.max(np.abs(a - b))) / max(float(np.max(np.abs(b))), 1e-30)
assert rel(m2d.numpy(), gfull["dL_dmeans2D"]) < 1e-5 and rel(opac.numpy(), gfull["dL_dopacity"]) < 1e-5
assert rel(col.numpy(), gfull["dL_dcolors"]) < 1e-5 and rel(conic.numpy(), gfull["dL_dconic"]) < 1e-5
if rank == 0:
    assert img is not None and np.array_equal(img.numpy(), full.color)
    assert img_halves is not None and np.array_equal(img_halves.numpy(), full.color)
    print("GLOO_SHARD_OK")
else:
    assert img is None
dist.destroy_process_group()
