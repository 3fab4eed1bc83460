#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X rasterizer hot path.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one forward + backward pass of the hot path over one synthetic frame of BASELINE.json's
headline configuration (C2: 1M Gaussians, 1920x1080, hierarchical sort; `--variant full` = PTD_MAX order +
rect/tight/tile-based/4x4 culling + load balancing, `--variant min` = plain Z order, no culling), inputs
resident in HBM, through the public drop-in API (GaussianRasterizer -> autograd -> _C -> C ABI).
Rank 0 prints ONE JSON line (contract in the task statement / DESIGN.md section "Measurement").

N > 1: the unit of the metric is a frame, and frames are independent, so the headline shards FRAMES over the ranks
(`--shard frames`, the default): every rank renders its own frame, no data-path collective, weak scaling.  The same
run then also times the north star's optional mode -- ONE frame partitioned by screen-tile row, image strips gathered
to rank 0 over RCCL, per-Gaussian gradient records all-reduced (strong scaling) -- and reports it in the `tile_shard`
object of the JSON line.  `--shard tilerows` makes that mode the headline instead.

The `cpu_baseline` leg (rank 0, N == 1 only) times the CPU oracle on a bounded sample of the same frame:
it is a reported, non-target baseline.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "stopthepop-rasterization_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s


def settings_for(variant: str, workload: str):
    import diff_gaussian_rasterization as dgr
    es = dgr.ExtendedSettings()
    if workload == "C3":
        es.sort_settings.sort_mode = dgr.SortMode.PPX_KBUFFER
        es.sort_settings.queue_sizes.per_pixel = 16
        return es
    if workload == "C1":
        return es
    es.sort_settings.sort_mode = dgr.SortMode.HIER
    if variant == "full":
        es.sort_settings.sort_order = dgr.GlobalSortOrder.PTD_MAX
        es.culling_settings.rect_bounding = True
        es.culling_settings.tight_opacity_bounding = True
        es.culling_settings.tile_based_culling = True
        es.culling_settings.hierarchical_4x4_culling = True
        es.load_balancing = True
    return es


def algorithmic_bytes(P, P_v, R, N, T, M, S, mode_hier, K, E, sh, B=0):
    """Compulsory HBM traffic per stage, bytes: every array element the stage must read or
    write counted once.  P Gaussians, P_v visible, R tile-list entries, N pixels, T tiles, M SH coefficients,
    S = 1 if the mode needs Sigma^-1, K/E = per-tile-depth / culling extras of duplicate, B = blended
    (pixel, entry) pairs recorded in the blend log (hierarchical training forward; 0 otherwise)."""
    b = {}
    b["preprocess"] = P * (44 + 8) + P_v * (12 * M * sh + 60 + (48 + 64) * S + 15 * sh)
    b["scan"] = 8 * P
    b["duplicate"] = 8 * P + P_v * (20 + 16 * E + 48 * K) + 12 * R
    b["sort"] = 24 * R
    b["ranges"] = 8 * R + 16 * T
    # per-pixel-sort modes (S): the list-ordered entry records (80 B each) are gathered once (id + the Gaussian's packed
    # 64-byte line + colour in, record out) and are what the render kernels read; GLOBAL reads by Gaussian id
    b["gather"] = R * (4 + 64 + 12 + 80) if S else 0
    per_entry_fwd = 80 if S else (4 + 24 + 12)
    # forward render: every list entry's data once per tile, the pixel outputs, and (recording forward) the 2-byte log
    # record of every blended pair + n_contrib + tile flags
    b["render_fwd"] = 8 * T + R * per_entry_fwd + N * (16 + (0 if mode_hier else 4)) + (2 * B + 4 * N + 4 * T if B else 0)
    b["zero_fill"] = P * 64
    if B:   # replay backward: the log, the blended entries' records (mean + id, conic/opacity, colour: 48 B), the pixel state,
            # and one read-modify-write of every visible Gaussian's 64-byte gradient record
        b["render_bwd"] = 8 * T + 2 * B + R * 48 + N * (4 + 4 + 12 + 12) + 128 * P_v
    else:
        b["render_bwd"] = 8 * T + R * (4 + 24 + 48 * S + 12) + N * (16 + (12 if S else 4)) + 128 * P_v
    b["bwd_preprocess"] = 4 * P + P_v * (64 + 36) + 4 * P + P_v * (36 + sh * (12 * M + 15) + 52) + P_v * (12 + sh * 12 * M + 28 + 28)
    return b


def measured_traffic(kernel: str, workload: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    tools/profile.sh + tools/traffic_json.py on the GPU box: FETCH_SIZE and WRITE_SIZE collected in separate passes,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None when no profile of this kernel exists."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        e = t.get(workload, {}).get(kernel)
        return int(e["hbm_bytes_per_launch"]) if e else None
    except (OSError, ValueError, KeyError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C2", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--variant", default="full", choices=["full", "min"])
    ap.add_argument("--shard", default="frames", choices=["tilerows", "frames"])
    ap.add_argument("--tile-shard-probe", action="store_true",
                    help="N > 1 with --shard frames: additionally time one frame sharded by tile row over all ranks (extra \"tile_shard\" "
                         "object; off by default so that the headline run contains no collective beyond its barriers)")
    ap.add_argument("--no-tile-shard-probe", action="store_true", help="(accepted, no effect: the probe is opt-in)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; invalidates the headline number)")
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--prewarm-seconds", type=float, default=0.5, help="untimed steps before the W warm-up steps (device clock ramp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0, help="tile rows in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--train-forward-only", action="store_true",
                    help="debug: forward passes that expect a backward (recording forward) without running it; read stage_ms only")
    ap.add_argument("--per-step", action="store_true", help="debug: print cumulative wall time after each timed step")
    ap.add_argument("--force-shard", action="store_true", help="use the tile-row sharded path even with one rank (self-test of the exchange code)")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the result of rank 0.  Libraries write there too (RCCL prints a five-line version
    # banner through C stdio, which is flushed at exit, i.e. after our line): file descriptor 1 is pointed at stderr for
    # the whole run and the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or args.force_shard:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist = dist_mod

    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _C, scenes, tile_shard

    fwd_only = args.fwd_only or args.workload == "C4"
    scene = scenes.config(args.workload, scale=args.scale)
    es = settings_for(args.variant, args.workload)
    sdict = es.to_dict()
    t = lambda a, rg=False: None if a is None else torch.tensor(a, device=dev).requires_grad_(rg and not fwd_only)
    means3D, opac = t(scene.means3D, True), t(scene.opacities, True)
    scales, rots, shs = t(scene.scales, True), t(scene.rotations, True), t(scene.shs, True)
    means2D = torch.zeros_like(means3D, requires_grad=not fwd_only)
    w_img = t(scene.dL_dout)
    rs = dgr.GaussianRasterizationSettings(
        image_height=scene.H, image_width=scene.W, tanfovx=scene.tanfovx, tanfovy=scene.tanfovy, bg=t(scene.bg),
        scale_modifier=1.0, viewmatrix=t(scene.viewmatrix), projmatrix=t(scene.projmatrix),
        inv_viewprojmatrix=t(scene.inv_viewprojmatrix), sh_degree=scene.sh_degree, campos=t(scene.campos),
        prefiltered=False, settings=es, render_depth=False, debug=False)

    gy = (scene.H + 15) // 16
    sharded = (world > 1 or args.force_shard) and args.shard == "tilerows"
    if sharded:
        raster = tile_shard.TileRowShardedRasterizer(rs, dist, rank, world)
    else:
        raster = dgr.GaussianRasterizer(rs)
    leaves = [means3D, means2D, opac, scales, rots, shs]
    state = {}

    def step():
        for x in leaves:
            if x is not None and x.grad is not None:
                x.grad = None
        color, radii = raster(means3D, means2D, opac, shs=shs, scales=scales, rotations=rots)
        state["color"], state["radii"] = color, radii
        if not fwd_only and not args.train_forward_only:
            (color * w_img).sum().backward()
        elif args.train_forward_only:
            _C.release_scratch(color.grad_fn.saved_tensors[11]); _C.release_scratch(color.grad_fn.saved_tensors[10])  # nobody will replay this log: hand the buffers back

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # bring the device out of its idle power state before the contract's W warm-up steps: the first process on a fresh box
    # otherwise measures the clock ramp (seen: 299 instead of 337 frames/s); untimed, like the warm-up itself
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm_seconds:
        step()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    barrier()
    import gc
    gc.collect()
    gc.disable()  # (a collection inside the timed region is a multi-millisecond stall of the launching thread)
    _C.timing_enable(True)  # hipEvents around every stage of every timed step, on the launch stream, no extra sync
    t0 = time.perf_counter()
    per_step = []
    for _ in range(args.steps):
        step()
        if args.per_step:  # debug: per-step wall time (adds a sync per step, invalidates the headline number)
            torch.cuda.synchronize(dev)
            per_step.append(round(1000.0 * (time.perf_counter() - t0), 3))
    barrier()
    if args.per_step and rank == 0:
        print("cumulative ms after each step:", per_step, "reserved GB:", round(torch.cuda.memory_reserved(dev) / 2**30, 2),
              "num_alloc_retries:", torch.cuda.memory_stats(dev).get("num_alloc_retries"), "segments:",
              torch.cuda.memory_stats(dev).get("segment.all.allocated"), file=sys.stderr, flush=True)
    dt = time.perf_counter() - t0
    gc.enable()
    stage_ms = {k: v for k, v in _C.timing_read().items() if v >= 0}  # means over the timed region
    _C.timing_enable(False)

    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    frames_per_step = world if (world > 1 and not sharded) else 1
    value = frames_per_step * args.steps / dt
    ms_per_step = 1000.0 * dt / args.steps

    # auxiliary measurement (N > 1, or --force-shard handled above): one frame sharded by tile row over all ranks
    tile_probe = None
    if world > 1 and not sharded and args.tile_shard_probe:
        try:
            r2 = tile_shard.TileRowShardedRasterizer(rs, dist, rank, world)

            def step2():
                for x in leaves:
                    if x is not None and x.grad is not None:
                        x.grad = None
                color, _ = r2(means3D, means2D, opac, shs=shs, scales=scales, rotations=rots)
                if not fwd_only:
                    (color * w_img).sum().backward()

            k2 = max(1, min(args.steps, 10))
            for _ in range(2):
                step2()
            barrier()
            t2 = time.perf_counter()
            for _ in range(k2):
                step2()
            barrier()
            dt2 = time.perf_counter() - t2
            tt2 = torch.tensor([dt2], device=dev, dtype=torch.float64)
            dist.all_reduce(tt2, op=dist.ReduceOp.MAX)
            dt2 = float(tt2.item())
            tile_probe = {"value": round(k2 / dt2, 3), "unit": "frames/s", "ms_per_frame": round(1000.0 * dt2 / k2, 4), "steps": k2,
                          "scaling": "strong", "parallelism": f"tilerows{world}",
                          "exchange": "gather of image strips to rank 0 + all-reduce of 36 B per Gaussian (RCCL)"}
        except Exception as ex:  # the headline above stands on its own
            tile_probe = {"error": repr(ex)[:300]}

    if rank == 0:
        # measured sizes for the byte model
        radii = state["radii"]
        P = scene.P
        P_v = int((radii > 0).sum().item())
        fn = state["color"].grad_fn
        R = int(getattr(fn, "num_rendered", 0)) if fn is not None else 0
        N = scene.W * scene.H
        T = ((scene.W + 15) // 16) * gy
        mode = int(sdict["sort_settings"]["sort_mode"])
        order = int(sdict["sort_settings"]["sort_order"])
        S = 1 if (mode != 0 or order >= 2) else 0
        Kf = 1 if order >= 2 else 0
        E = 1 if (sdict["culling_settings"]["tile_based_culling"] or order == 3) else 0
        # blended pairs recorded by the training forward (one untimed forward; the timed graphs are gone)
        B = 0
        head, mid = int(sdict["sort_settings"]["queue_sizes"]["per_pixel"]), int(sdict["sort_settings"]["queue_sizes"]["tile_2x2"])
        cull = bool(sdict["culling_settings"]["hierarchical_4x4_culling"])
        recording = mode in (2, 3) and not fwd_only and os.environ.get("STP_BACKWARD", "replay") != "resort" and not sharded
        if recording:
            c2, _ = raster(means3D, means2D, opac, shs=shs, scales=scales, rotations=rots)
            B = int(_C.image_array(c2.grad_fn.saved_tensors[11], scene.W, scene.H, "n_contrib").to(torch.int64).clamp_(max=256).sum().item())
            del c2
        bts = algorithmic_bytes(P, P_v, R, N, T, 16, S, mode == 3, Kf, E, 1, B)
        dom = "BwdRender" if (not fwd_only and stage_ms.get("BwdRender", 0) >= stage_ms.get("Render", 0)) else "Render"
        dom_bytes = bts["render_bwd"] if dom == "BwdRender" else bts["render_fwd"]
        dom_ms = stage_ms.get(dom, float("nan"))
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms and dom_ms > 0 else float("nan")
        if mode == 3:
            kname = ("render_hier_replay_kernel" if recording else f"render_hier_kernel<{head}, {mid}, {str(cull).lower()}, 1>") if dom == "BwdRender" \
                else f"render_hier_kernel<{head}, {mid}, {str(cull).lower()}, {2 if recording else 0}>"
        else:
            kname = {0: "render_global", 1: "render_full", 2: "render_kbuffer"}[mode] + ("_bwd_kernel" if dom == "BwdRender" else "_fwd_kernel")
            if mode == 2:
                kname = "render_hier_replay_kernel" if (dom == "BwdRender" and recording) else \
                    f"render_kbuffer_kernel<{head}, {1 if dom == 'BwdRender' else (2 if recording else 0)}>"
        traffic = measured_traffic(kname, f"{args.workload}-{args.variant}")
        fwd_bytes = sum(bts[k] for k in ("preprocess", "scan", "duplicate", "sort", "ranges", "gather", "render_fwd"))
        bwd_bytes = sum(bts[k] for k in ("zero_fill", "render_bwd", "bwd_preprocess"))
        out = {
            "metric": "fwd+bwd frames/sec at 1920×1080, 1M Gaussians; PSNR vs reference",
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}-{args.variant}: {P} Gaussians, {scene.W}x{scene.H}, SH degree 3, "
                                   f"sort_mode={mode} sort_order={order} culling={sdict['culling_settings']} "
                                   f"{'fwd' if fwd_only else 'fwd+bwd'}",
                       "P": P, "P_visible": P_v, "num_rendered": R, "tiles": T, "blended_pairs": B,
                       "parallelism": (f"tilerows{world}" if sharded else f"frames{world}") if world > 1 else "single",
                       "scale": args.scale},
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "algorithmic_bytes": {"forward": int(fwd_bytes), "backward": int(bwd_bytes)},
            "roofline": {"bound": "hbm", "kernel": kname,
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_ms, 4),
                         "whole_step_frac": round(((fwd_bytes + (0 if fwd_only else bwd_bytes)) / (ms_per_step * 1e-3) / 1e9) / HBM_PEAK_GBS, 5)},
        }
        if tile_probe is not None:
            out["tile_shard"] = tile_probe
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, sdict, gy, fwd_only, args.cpu_rows)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    os.close(result_fd)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(scene, sdict, gy, fwd_only, rows):
    """Oracle (our CPU restatement, OpenMP over all host cores) on a bounded sample of the same frame:
    a centred window of tile rows, sized by a 2-row calibration run so that the sample takes ~15 s, the time
    extrapolated linearly to the full frame."""
    from oracle import oracle as orc
    cores = orc.num_threads()

    def run(r):
        y0 = max(0, (gy - r) // 2)
        t0 = time.perf_counter()
        f = orc.forward_scene(scene, sdict, tile_rows=(y0, y0 + r))
        if not fwd_only:
            f.backward(scene.dL_dout)
        dt = time.perf_counter() - t0
        f.free()
        return y0, dt

    if rows <= 0:
        cal_rows = min(2, gy)
        _, t_cal = run(cal_rows)                       # includes the per-Gaussian stages of all P once
        _, t_cal0 = run(0) if False else (0, 0.0)
        per_row = max(t_cal / cal_rows, 1e-3)
        rows = int(max(cal_rows, min(gy, round(15.0 / per_row))))
    y0, dt = run(rows)
    reps = 1
    while dt < 10.0 and reps < 16:   # a many-core host finishes the whole frame in seconds: repeat to ~10 s of CPU work
        dt += run(rows)[1]
        reps += 1
    est_frame_s = (dt / reps) * gy / rows
    return {"value": round(1.0 / est_frame_s, 5), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"tile rows {y0}..{y0 + rows - 1} of {gy} of the same frame ({'fwd' if fwd_only else 'fwd+bwd'}), "
                      f"{reps} repetition(s), {dt:.1f} s measured on {cores} threads, extrapolated x{gy / rows:.2f}"}


if __name__ == "__main__":
    main()
