"""Tile-row sharding of ONE frame across the GPUs of a node (BASELINE.json north_star / SURVEY.md 8(e)).

After binning, screen tiles are independent, so rank g of G owns the tile rows
[g*ceil(Ty/G), min(Ty, (g+1)*ceil(Ty/G))).  Every rank holds the full Gaussian set and runs the
per-Gaussian preprocess on all of it (cheap, HBM-streaming), but bins, sorts and blends only the tiles
of its rows (StpSettings.tile_y0/tile_y1).  Exchange steps:

  forward  : the image strips go to rank 0 as grouped point-to-point sends (three contiguous channel segments per
             peer, received straight into the rows of the root's own output tensor: every peer->root transfer rides
             its own xGMI link; a ring would be bound by one link) -- or are all-gathered when every rank needs the frame;
  backward : the render half runs on the rank's rows only and yields PARTIAL per-Gaussian sums
             (one 64-byte gradient record per Gaussian, see include/stp_raster.h); they are linear
             inputs of the per-Gaussian backward, so ONE all-reduce(sum) of the (P,16) record buffer precedes
             the (replicated) preprocess half.  The forward's per-Gaussian state is identical on all ranks
             because visibility is decided on the full frame.

No reference counterpart exists (the reference is single-GPU); the pure partition/assembly logic below is
device-agnostic and is exercised by world_size-2 gloo tests on CPU.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from . import _C


def tile_rows(height: int, tile: int = 16) -> int:
    return (height + tile - 1) // tile


def row_partition(n_rows: int, world: int) -> List[Tuple[int, int]]:
    """[(y0, y1)] per rank: contiguous blocks of ceil(n_rows/world) rows, last ranks may be short/empty."""
    per = (n_rows + world - 1) // world
    return [(min(n_rows, r * per), min(n_rows, (r + 1) * per)) for r in range(world)]


def strip_pixels(rows: Tuple[int, int], height: int, tile: int = 16) -> Tuple[int, int]:
    """Pixel-row interval [py0, py1) covered by a tile-row interval."""
    return min(height, rows[0] * tile), min(height, rows[1] * tile)


def pack_strip(image: torch.Tensor, rows: Tuple[int, int], rows_max: int, tile: int = 16) -> torch.Tensor:
    """(3,H,W) -> contiguous (3, rows_max*tile, W) strip of this rank's pixel rows, zero padded."""
    C, H, W = image.shape
    py0, py1 = strip_pixels(rows, H, tile)
    out = image.new_zeros((C, rows_max * tile, W))
    if py1 > py0:
        out[:, : py1 - py0, :] = image[:, py0:py1, :]
    return out


def assemble(strips: List[torch.Tensor], parts: List[Tuple[int, int]], height: int, tile: int = 16) -> torch.Tensor:
    """Inverse of pack_strip over all ranks -> (3,H,W)."""
    C, _, W = strips[0].shape
    out = strips[0].new_zeros((C, height, W))
    for strip, rows in zip(strips, parts):
        py0, py1 = strip_pixels(rows, height, tile)
        if py1 > py0:
            out[:, py0:py1, :] = strip[:, : py1 - py0, :]
    return out


def balanced_partition(row_cost, world: int) -> List[Tuple[int, int]]:
    """Contiguous tile-row blocks of (nearly) equal COST instead of equal height: row_cost[y] = work of tile row y (the
    number of tile-list entries of its tiles is a good proxy for the blend time).  Cut k goes where the running sum
    first reaches k/world of the total; every rank gets at least zero rows, the blocks tile [0, n_rows) in order.
    SURVEY.md 8(e): "optional re-partition by per-row duplicate histogram"."""
    cost = [max(0.0, float(c)) for c in row_cost]
    n = len(cost)
    total = sum(cost)
    if total <= 0.0 or world <= 1:
        return row_partition(n, world)
    cuts, run, y = [0], 0.0, 0
    for k in range(1, world):
        target = total * k / world
        while y < n and run + 0.5 * cost[y] < target:   # a row goes to the side its midpoint falls on
            run += cost[y]
            y += 1
        cuts.append(y)
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def _host_staged(dist) -> bool:
    """gloo moves host memory only: device tensors are staged through the host (CPU tests, and the 2-process test that runs
    the HIP path with both ranks on ONE GPU).  RCCL (backend "nccl") takes device pointers and uses xGMI."""
    try:
        return dist.get_backend() == "gloo"
    except Exception:
        return False


def split_row(part: Tuple[int, int]) -> int:
    """Tile row at which a rank's block is rendered in two launches (0: one launch).  The same function on every rank: the root needs the
    peers' split rows to post its receives."""
    y0, y1 = part
    return (y0 + y1) // 2 if y1 - y0 >= 2 else 0


_comm_streams = {}


def _comm_stream(device):
    """One side stream per device for the strip exchange: the collective library orders an operation behind the stream that is CURRENT when it
    is posted, so operations posted under this stream wait for what this stream waited for -- not for the whole render."""
    key = torch.device(device).index
    if key not in _comm_streams:
        _comm_streams[key] = torch.cuda.Stream(device=device)
    return _comm_streams[key]


def gather_image(local_image: torch.Tensor, parts, rank: int, world: int, dist, dst: int = 0, to_all: bool = False, halves=None):
    """Exchange step of the forward.  `local_image` is this rank's (3,H,W) output with its own pixel rows rendered.
    Root gather (default): in CHW layout a rank's strip is THREE contiguous segments (one per channel); every peer
    sends its three segments and the root receives them straight into the rows of ITS OWN output tensor -- one grouped
    batch of point-to-point operations (ncclGroupStart/End under batch_isend_irecv), every peer -> root transfer on its
    own xGMI link, no padding, no staging copy, no assembly pass.  Returns the assembled frame (the
    root's local_image itself, completed in place) on `dst`, None elsewhere.
    halves = (start_event, first_half_event) (round 6): every rank rendered its block in two launches split at split_row(part) and
    first_half_event fires between them (_C.set_forward_split).  A peer then sends the upper half of its strip as soon as that event has
    fired -- while the lower half is still being blended -- and the root posts ALL its receives at once on the side stream (they land in rows
    its own render never touches; start_event, recorded before the forward, keeps them behind whatever used the tensor's memory before).
    Two batches per peer instead of one; the bytes, the landing addresses and the assembled frame are the same.
    to_all: every rank needs the frame -> all-gather of equal-size padded strips + one assembly copy."""
    H = local_image.shape[1]
    if to_all:
        rows_max = max(1, max(b - a for a, b in parts))
        strip = pack_strip(local_image, parts[rank], rows_max)
        staged = _host_staged(dist) and strip.is_cuda
        s_ = strip.cpu() if staged else strip
        bufs = [torch.empty_like(s_) for _ in range(world)]
        dist.all_gather(bufs, s_)
        out = assemble(bufs, parts, H)
        return out.to(local_image.device) if staged else out
    staged = _host_staged(dist) and local_image.is_cuda

    def pieces(part):
        """Pixel-row ranges of a rank's strip in sending order: [upper half, lower half] of a split block, else the whole strip."""
        py0, py1 = strip_pixels(part, H)
        ym = split_row(part) if halves is not None else 0
        pm = min(ym * 16, H)
        if ym and py0 < pm < py1:
            return [(py0, pm), (pm, py1)]
        return [(py0, py1)] if py1 > py0 else []

    batches, landing = [[], []], []   # batch 0: upper halves (or whole strips), batch 1: lower halves
    if rank == dst:
        for r in range(world):
            if r == dst:
                continue
            for n, (a, b) in enumerate(pieces(parts[r])):
                for c in range(local_image.shape[0]):
                    seg = local_image[c, a:b, :]                      # contiguous: rows of one channel
                    buf = torch.empty(seg.shape, dtype=seg.dtype) if staged else seg
                    landing.append((seg, buf))
                    batches[n].append(dist.P2POp(dist.irecv, buf, r))
    else:
        for n, (a, b) in enumerate(pieces(parts[rank])):
            for c in range(local_image.shape[0]):
                seg = local_image[c, a:b, :]
                batches[n].append(dist.P2POp(dist.isend, seg.cpu() if staged else seg, dst))
    reqs = []
    overlapped = halves is not None and not staged and local_image.is_cuda
    if overlapped:
        start_event, first_half_event = halves
        comm = _comm_stream(local_image.device)
        comm.wait_event(start_event)
        if rank == dst:
            with torch.cuda.stream(comm):           # every receive now: none of them waits for the root's own render
                for ops in batches:
                    if ops:
                        reqs += dist.batch_isend_irecv(ops)
        else:
            comm.wait_event(first_half_event)
            with torch.cuda.stream(comm):           # the upper half behind the first launch ...
                if batches[0]:
                    reqs += dist.batch_isend_irecv(batches[0])
            if batches[1]:                          # ... the lower half behind the second (the current stream)
                reqs += dist.batch_isend_irecv(batches[1])
    else:
        for ops in batches:
            if ops:
                reqs += dist.batch_isend_irecv(ops)
    for req in reqs:
        req.wait()                                   # (RCCL: the current stream waits for the operation; gloo: the host does)
    if staged:
        for seg, buf in landing:
            seg.copy_(buf)
    return local_image if rank == dst else None


SPLIT_FORWARD = True   # (module switch: False restores the single launch + one batch of rounds 1-5)
RECORD_USED = 9  # floats of a gradient record that carry data (include/stp_raster.h, stp_backward)
RECORD_CHUNKS = 2  # pieces the record all-reduce is pipelined in against the per-Gaussian half of the backward (_ShardedRasterize.backward)


def record_chunk_bounds(P: int, K: int) -> List[int]:
    """Row bounds of the K id ranges the library's chunked per-Gaussian half works on (stp_backward_phases: equal ranges of 256-Gaussian
    blocks): chunk k = rows [bounds[k], bounds[k + 1])."""
    n_blocks = (P + 255) // 256
    return [min(P, 256 * (n_blocks * k // K)) for k in range(K)] + [P]


def pack_partials(dL_dmeans2D, dL_dconic, dL_dopacity, dL_dcolors) -> torch.Tensor:
    """The reference's four partial-sum tensors (P,3),(P,2,2),(P,1),(P,3) -> the library's (P,16) gradient records
    (include/stp_raster.h, stp_backward: colour rgb | mean2D xy | conic xx xy yy | opacity | 7 unused).
    The product never needs this (the render half of the backward writes records directly and the all-reduce runs
    on them as they are); it documents the layout and lets CPU tests build records from the oracle's tensors."""
    P = dL_dmeans2D.shape[0]
    rec = torch.zeros((P, _C.GRAD_RECORD_FLOATS), dtype=dL_dmeans2D.dtype, device=dL_dmeans2D.device)
    rec[:, 0:3] = dL_dcolors.reshape(P, 3)
    rec[:, 3:5] = dL_dmeans2D.reshape(P, 3)[:, 0:2]
    rec[:, 5:8] = dL_dconic.reshape(P, 4)[:, [0, 1, 3]]
    rec[:, 8:9] = dL_dopacity.reshape(P, 1)
    return rec


def unpack_partials(rec: torch.Tensor):
    """Inverse of pack_partials: (dL_dmeans2D (P,3), dL_dconic (P,2,2), dL_dopacity (P,1), dL_dcolors (P,3))."""
    P = rec.shape[0]
    m2d = torch.zeros((P, 3), dtype=rec.dtype, device=rec.device)
    m2d[:, 0:2] = rec[:, 3:5]
    conic = torch.zeros((P, 4), dtype=rec.dtype, device=rec.device)
    conic[:, [0, 1, 3]] = rec[:, 5:8]
    return m2d, conic.reshape(P, 2, 2), rec[:, 8:9].contiguous(), rec[:, 0:3].contiguous()


class _ShardedRasterize(torch.autograd.Function):
    """rasterize_gaussians for one rank's tile rows + the two exchange steps."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs, shard, shard_state=None):
        dist, rank, world, parts, to_all = shard
        if rs.render_depth:
            # the depth visualisation normalises by the FRAME's depth extrema (reference rasterizer_impl.cu:54-109); a rank only
            # sees its rows, so the strips would be normalised differently and the assembled picture would have seams
            raise RuntimeError("render_depth is not available with tile-row sharding (the depth colormap is normalised by the "
                               "whole frame's extrema); render it on one GPU")
        sdict = dict(rs.settings.to_dict())
        y0, y1 = parts[rank]
        n_rows = tile_rows(rs.image_height)
        rows = (y0, y1) if y1 > y0 else (n_rows, n_rows)   # (an empty block; (0, 0) would mean "all rows" to the library)
        sdict["_tile_rows"] = rows
        ctx.log_lease = None
        if any(ctx.needs_input_grad) and not rs.render_depth:   # the backward-mode policy of the unsharded autograd function (__init__.py)
            uses_log = int(sdict["sort_settings"]["sort_mode"]) in (2, 3)
            if uses_log and means3D.is_cuda and means3D.size(0) != 0 and \
                    _C.decide_recording(sdict.get("_backward_mode"), means3D.device, rs.image_width, rs.image_height, rows):
                sdict["_record_blend_log"] = True
                sdict["_backward_mode"] = "replay"
                # (the image buffer and the blend log of a rank cover ITS tile rows only: 1 / world of the frame's log)
                ctx.log_lease = _C.LogLease(_C._device_index(means3D.device), _C.blend_log_bytes(rs.image_width, rs.image_height, rows))
            else:
                sdict["_backward_mode"] = "resort"
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.inv_viewprojmatrix, rs.tanfovx, rs.tanfovy, rs.image_height,
                rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, sdict, rs.render_depth, rs.debug)
        # the render in two launches with an event between them, so that the first half of the strip can leave while the second is blended
        # (gather_image); the assembled frame is the same with or without
        halves = None
        ym = split_row((y0, y1))
        if world > 1 and not to_all and means3D.is_cuda and means3D.size(0) != 0 and SPLIT_FORWARD:
            start_ev, half_ev = torch.cuda.Event(), torch.cuda.Event()
            start_ev.record()     # (also what keeps the root's early receives behind earlier users of the output tensor's memory)
            half_ev.record()      # (torch creates the hipEvent at the first record: the library needs its handle)
            if ym:
                _C.set_forward_split(ym, half_ev)
            halves = (start_ev, half_ev)
        try:
            num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians(*args)
        finally:
            if halves is not None and ym:
                _C.set_forward_split(0, None)   # (a forward that raised before the library saw the request must not leave it to the next one)
        if halves is not None and not ym:
            halves[1].record()    # (a block of one tile row is one launch: its "first half" is the whole strip)
        if ctx.log_lease is not None:   # the library chose the log's depth for this frame: account what the buffer really holds
            ctx.log_lease.resize(_C.blend_log_bytes(rs.image_width, rs.image_height, rows, depth=_C.blend_log_depth(imgBuffer)))
        ctx.rs, ctx.sdict, ctx.shard, ctx.num_rendered = rs, sdict, shard, num_rendered
        ctx.img_generation = _C.scratch_generation(imgBuffer)
        ctx.bin_generation = _C.scratch_generation(binningBuffer)
        ctx.save_for_backward(colors_precomp, means3D, opacities, scales, rotations, cov3Ds_precomp, radii, sh, color,
                              geomBuffer, binningBuffer, imgBuffer)
        if shard_state is not None:   # per-row work of this frame (own rows) for the next frame's partition
            # reduced HERE, while the image buffer is certainly this frame's: the backward hands the buffer back to the scratch
            # pool, and another rasterizer on the device (an eval render, a second sharded module) may have reused it by the
            # time the next frame is partitioned -- only the small per-row tensor is kept
            gx = (rs.image_width + 15) // 16
            # (the buffer holds the ranges of this rank's rows only; the other rows of the per-row vector stay zero and come from their owners)
            r = _C.image_array(imgBuffer, rs.image_width, rs.image_height, "ranges", tile_rows=rows).reshape(-1, 2)[: gx * (rows[1] - rows[0])].to(torch.int64)
            pending = torch.zeros(n_rows, dtype=torch.float32, device=r.device)
            pending[rows[0]:rows[1]] = (r[:, 1] - r[:, 0]).reshape(rows[1] - rows[0], gx).sum(dim=1).to(torch.float32)
            shard_state["pending"] = pending
        full = gather_image(color, parts, rank, world, dist, dst=0, to_all=to_all, halves=halves)
        ctx.mark_non_differentiable(radii)
        # every rank returns a (3,H,W) tensor: the assembled frame where it is available, else the local strip image
        return (full if full is not None else color), radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        rs, sdict = ctx.rs, ctx.sdict
        dist, rank, world, parts, to_all = ctx.shard
        (colors_precomp, means3D, opacities, scales, rotations, cov3Ds_precomp, radii, sh, color, geomBuffer,
         binningBuffer, imgBuffer) = ctx.saved_tensors
        _C.check_scratch(imgBuffer, ctx.img_generation)
        _C.check_scratch(binningBuffer, ctx.bin_generation)
        args = (rs.bg, means3D, radii, opacities, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.inv_viewprojmatrix, rs.tanfovx, rs.tanfovy, color, grad_out_color, sh,
                rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, sdict, rs.debug)
        # The render half writes COMPACT records -- (P, 9) floats, no padding (phases bit 2, include/stp_raster.h) -- and that tensor is summed
        # over the ranks AS IT IS: 36 bytes per Gaussian on the wire, no pack / unpack pass on either side (rounds 1-3 sliced the used
        # columns out of the padded (P, 16) records and copied them back: two strided 36 MB copies per step at C2).
        records = _C.rasterize_gaussians_backward(*args, phases=1 | 4)
        # The exchange, taken off the critical path as far as the data flow allows.  A Gaussian's record is complete only when EVERY tile of the
        # rank has been replayed (any tile may hold any Gaussian), so nothing of the all-reduce can start before the render half has ended; but
        # the per-Gaussian half is independent per Gaussian: the records are summed in K pieces by id range (asynchronous collectives on RCCL's
        # own stream) and the per-Gaussian half runs on range k as soon as piece k has arrived, while piece k + 1 is still on the links.
        K = RECORD_CHUNKS if records.shape[0] >= 256 * RECORD_CHUNKS else 1
        if _host_staged(dist) and records.is_cuda:   # (gloo: host memory only)
            host = records.cpu()
            dist.all_reduce(host)
            records = host.to(records.device)
            K = 1
        if K == 1:
            if not (_host_staged(dist) and records.is_cuda):
                dist.all_reduce(records)
            out = _C.rasterize_gaussians_backward(*args, phases=2 | 4, partial=records)
        else:
            bounds = record_chunk_bounds(records.shape[0], K)
            works = [dist.all_reduce(records[bounds[k]:bounds[k + 1]], async_op=True) for k in range(K)]   # (row ranges of a contiguous (P, 9) tensor)
            out = None
            for k in range(K):
                works[k].wait()   # (the current stream waits for piece k; the host does not)
                out = _C.rasterize_gaussians_backward(*args, phases=2 | 4, partial=records, chunk=(k, K), outputs=out)
        _C.release_scratch(imgBuffer); _C.release_scratch(binningBuffer)
        if ctx.log_lease is not None:
            ctx.log_lease.release()
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = out
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, None, None, None)


class TileRowShardedRasterizer(torch.nn.Module):
    """Drop-in for GaussianRasterizer when one frame is split over `world` ranks by tile row.
    forward(...) has GaussianRasterizer.forward's signature; it returns (image, radii) where `image` is the
    assembled frame on rank 0 (on every rank with to_all=True) and this rank's strip image elsewhere.
    The loss must be evaluated so that each rank back-propagates the gradient of ITS rows (e.g. a per-pixel
    loss evaluated redundantly, or dL/dimage scattered from rank 0)."""

    def __init__(self, raster_settings, dist, rank: int, world: int, to_all: bool = False, rebalance: bool = False):
        """rebalance: re-partition the tile rows before every frame by the PREVIOUS frame's per-row tile-list lengths (one
        all-reduce of n_rows integers per frame) so that ranks get equal work, not equal height -- for scenes whose
        content is not uniform over the image.  Off by default: the partition is then the fixed ceil(Ty/G) blocks."""
        super().__init__()
        self.raster_settings = raster_settings
        self.n_rows = tile_rows(raster_settings.image_height)
        self.parts = row_partition(self.n_rows, world)
        self._comm = (dist, rank, world, to_all)
        self.shard = (dist, rank, world, self.parts, to_all)
        self.state = {"pending": None} if rebalance else None

    def _repartition(self):
        """per-row entry counts of the last frame (own rows) -> summed over ranks -> balanced_partition"""
        dist, rank, world, to_all = self._comm
        cost = self.state["pending"]   # (n_rows,) tile-list entries per tile row of the last frame, this rank's rows (reduced in forward)
        self.state["pending"] = None
        if _host_staged(dist) and cost.is_cuda:
            host = cost.cpu()
            dist.all_reduce(host)
            cost = host
        else:
            dist.all_reduce(cost)
        self.parts = balanced_partition(cost.tolist(), world)
        self.shard = (dist, rank, world, self.parts, to_all)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = lambda t: torch.Tensor([]) if t is None else t
        if self.state is not None and self.state["pending"] is not None:
            self._repartition()
        return _ShardedRasterize.apply(means3D, means2D, empty(shs), empty(colors_precomp), opacities, empty(scales),
                                       empty(rotations), empty(cov3D_precomp), self.raster_settings, self.shard, self.state)
