#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X rasterizer hot path.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one forward + backward pass of the hot path over one synthetic frame of BASELINE.json's
headline configuration (C2: 1M Gaussians, 1920x1080, hierarchical sort; `--variant full` = PTD_MAX order +
rect/tight/tile-based/4x4 culling + load balancing, `--variant min` = plain Z order, no culling), inputs
resident in HBM, through the public drop-in API (GaussianRasterizer -> autograd -> _C -> C ABI).
Rank 0 prints ONE JSON line (contract in the task statement / DESIGN.md section "Measurement").

N > 1: the north star's tile-row mode is the headline -- ONE frame partitioned by screen-tile row over the N ranks, image
strips sent to rank 0 over RCCL, per-Gaussian gradient records all-reduced: `"scaling": "strong"`, `value` = frames/s of
that one frame stream (`--shard tilerows`, the default for every workload).  The same run then also times the
frame-sharded mode (every rank its own frame, no data-path collective, weak scaling) and reports it in the
`frame_shard` object of the JSON line (`--no-frame-shard-probe` leaves it out); `--shard frames` swaps the two.
`rccl_ranks` is what the first collective of the run saw.

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

# (multi-process GPU work on this pool: the host driver only supports dmabuf IPC -- without this RCCL fails with `hipIpcGetMemHandle: invalid argument`.
# The boxes export it already; a launcher that builds its own environment may not)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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


def survey_bytes(P, P_v, R, N, T, M, S, mode, K, E, sh, ewa):
    """ALGORITHMIC bytes per stage, SURVEY.md section 8(d) verbatim (the compulsory traffic of the REFERENCE's data flow;
    the figure `roofline.achieved` is priced on).  P Gaussians, P_v visible, R tile-list entries, N pixels, T tiles, M SH
    coefficients, S = 1 if the mode needs Sigma^-1, K = per-tile depth order, E = TBC or PTD_MAX, sh = SH used."""
    hier, glob = mode == 3, mode == 0
    b = {}
    b["preprocess"] = P * (44 + 8) + P_v * (12 * M * sh + 60 + 48 * S + 15 * sh)
    b["scan"] = 8 * P
    b["duplicate"] = 8 * P + P_v * (20 + 16 * E + 48 * K) + 12 * R
    b["sort"] = 24 * R
    b["ranges"] = 8 * R + 16 * T
    b["render_fwd"] = 8 * T + R * (4 + 24 + 48 * S + 12) + N * (16 + (0 if hier else 4))
    b["zero_fill"] = P * (108 + 12 * M)
    b["render_bwd"] = 8 * T + R * (4 + 24 + 48 * S + 12) + N * (16 + (4 if glob else 12)) + 88 * P_v
    b["bwd_cov2D"] = 4 * P + P_v * (52 + 36) + (8 * P_v if ewa else 0)
    b["bwd_preprocess"] = 4 * P + P_v * (36 + sh * (12 * M + 15) + 52) + P_v * (12 + sh * 12 * M + 28)
    return b


def design_bytes(P, P_v, R, N, T, M, S, mode, K, E, sh, B=0):
    """Bytes of OUR data flow per stage: the same stages plus what this design adds on purpose -- the
    list-ordered 80-byte entry records and their gather, the packed 64-byte Gaussian line, the 2-byte blend-log record of
    every blended (pixel, entry) pair (B), the 64-byte gradient records.  Reported beside the algorithmic bytes; never used
    for `roofline.achieved`."""
    hier = mode == 3
    b = {}
    b["preprocess"] = P * (44 + 8) + P_v * (12 * M * sh + 60 + (48 + 64) * S + 15 * sh)
    b["scan"] = 8 * P
    b["duplicate"] = 8 * P + P_v * (20 + 16 * E + 48 * K) + 12 * R
    b["sort"] = 24 * R
    b["ranges"] = 8 * R + 16 * T
    b["gather"] = R * (4 + 64 + 12 + 80) if S else 0
    per_entry_fwd = 80 if S else (4 + 24 + 12)
    b["render_fwd"] = 8 * T + R * per_entry_fwd + N * (16 + (0 if hier else 4)) + (2 * B + 4 * N + 4 * T if B else 0)
    b["zero_fill"] = P * 64
    if B:
        b["render_bwd"] = 8 * T + 2 * B + R * 48 + N * (4 + 4 + 12 + 12) + 128 * P_v
    else:
        b["render_bwd"] = 8 * T + R * (4 + 24 + 48 * S + 12) + N * (16 + (12 if S else 4)) + 128 * P_v
    b["bwd_preprocess"] = 4 * P + P_v * (64 + 36) + 4 * P + P_v * (36 + sh * (12 * M + 15) + 52) + P_v * (12 + sh * 12 * M + 28 + 28)
    return b


def csrc_sha256():
    """Hash of the kernel sources: profiles/traffic.json is stamped with it, and a stale profile is not quoted."""
    import hashlib
    d = os.path.join(ROOT, "stopthepop-rasterization_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h", ".inc")) or fn == "Makefile":
            h.update(fn.encode()); h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()


def profile_entry(kernel: str, workload: str):
    """(entry, note) for `kernel` from the committed rocprofv3 PMC passes (profiles/traffic.json, written by tools/profile.sh +
    tools/traffic_json.py on the GPU box: FETCH_SIZE and WRITE_SIZE collected in separate passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950; SQ counters of pass 1).  The file carries the hash of csrc/ it was taken on:
    when the kernels have changed since, nothing is quoted (entry None, note says so)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, "no profiles/traffic.json"
    if t.get("_csrc_sha256") != csrc_sha256():
        return None, "profiles/traffic.json was taken on different kernel sources (csrc hash mismatch): not quoted"
    e = t.get(workload, {}).get(kernel)
    return (e, t.get("_taken", "")) if e else (None, f"no PMC profile of {kernel} for {workload}")


_SCENE_CACHE = {}


class Workload:
    """One BASELINE configuration resident in HBM: tensors, the drop-in rasterizer module, one step() = one pass of the hot path."""

    def __init__(self, name, variant, dev, fwd_only=False, scale=1.0, dist=None, rank=0, world=1, sharded=False, train_forward_only=False, loss=False):
        import diff_gaussian_rasterization as dgr
        from diff_gaussian_rasterization import _C, scenes, tile_shard
        self.name, self.variant, self.dev = name, variant, dev
        self.fwd_only = fwd_only or name == "C4"
        self.scale, self.world, self.sharded = scale, world, sharded
        self.train_forward_only = train_forward_only
        self.loss = loss   # True: the step also holds torch's kernels of a weighted-sum loss (the round-1/2 definition of a step)
        self._C = _C
        # (C2's scene serves the headline, the fma / loss side legs and C2-min: generated once per process)
        key = (name, scale)
        if key not in _SCENE_CACHE:
            if name != "C2":
                _SCENE_CACHE.pop(next((k for k in _SCENE_CACHE if k[0] != "C2"), None), None)
            _SCENE_CACHE[key] = scenes.config(name, scale=scale)
        self.scene = scene = _SCENE_CACHE[key]
        self.es = es = settings_for(variant, name)
        self.sdict = es.to_dict()
        fo = self.fwd_only
        t = lambda a, rg=False: None if a is None else torch.tensor(a, device=dev).requires_grad_(rg and not fo)
        self.means3D, self.opac = t(scene.means3D, True), t(scene.opacities, True)
        self.scales, self.rots, self.shs = t(scene.scales, True), t(scene.rotations, True), t(scene.shs, True)
        self.means2D = torch.zeros_like(self.means3D, requires_grad=not fo)
        self.w_img = t(scene.dL_dout)
        self.rs = dgr.GaussianRasterizationSettings(
            image_height=scene.H, image_width=scene.W, tanfovx=scene.tanfovx, tanfovy=scene.tanfovy, bg=t(scene.bg),
            scale_modifier=1.0, viewmatrix=t(scene.viewmatrix), projmatrix=t(scene.projmatrix),
            inv_viewprojmatrix=t(scene.inv_viewprojmatrix), sh_degree=scene.sh_degree, campos=t(scene.campos),
            prefiltered=False, settings=es, render_depth=False, debug=False)
        self.gy = (scene.H + 15) // 16
        self.raster = tile_shard.TileRowShardedRasterizer(self.rs, dist, rank, world) if sharded else dgr.GaussianRasterizer(self.rs)
        self.leaves = [self.means3D, self.means2D, self.opac, self.scales, self.rots, self.shs]
        self.state = {}
        self.label = f"{name}-{variant}" if name in ("C2", "C2L", "C2H", "C4", "C5", "L1") else name

    def tensors(self):
        return (self.means3D, self.means2D, self.opac, self.shs, self.scales, self.rots)

    def step(self):
        for x in self.leaves:
            if x is not None and x.grad is not None:
                x.grad = None
        color, radii = self.raster(self.means3D, self.means2D, self.opac, shs=self.shs, scales=self.scales, rotations=self.rots)
        self.state["color"], self.state["radii"] = color, radii
        if not self.fwd_only and not self.train_forward_only:
            if self.loss:
                (color * self.w_img).sum().backward()   # + three torch kernels (mul, sum, mul): about 0.06 ms at 1080p, none of them the hot path
            else:
                color.backward(self.w_img)               # dL/dimage is resident in HBM like every other input of the step
        elif self.train_forward_only:  # nobody will replay this log: hand the buffers back
            self._C.release_scratch(color.grad_fn.saved_tensors[11]); self._C.release_scratch(color.grad_fn.saved_tensors[10])

    def free(self):
        self.state.clear()
        for x in self.leaves:
            if x is not None:
                x.grad = None
        self.leaves = []
        self.raster = None


def timed_region(wl, steps, warmup, prewarm_seconds, barrier, per_step=False):
    """W untimed warm-up steps, then EXACTLY `steps` steps between two barriers (+ device synchronisation).  Returns wall
    seconds, the library's per-stage hipEvent means over the region, and the per-step intervals between hipEvents recorded on
    the launch stream after every step (no synchronisation inside the region)."""
    _C, dev = wl._C, wl.dev
    # bring the device out of its idle power state before the contract's W warm-up steps: the first process on a fresh box
    # otherwise measures the clock ramp (seen: 299 instead of 337 frames/s); untimed, like the warm-up itself
    # The untimed steps run WITH the stage timer switched on, like the timed ones (an event gets its signal from the runtime at its first
    # record, not at hipEventCreate: the timer's 512 events should have been recorded once before the timed region).  The counters are reset
    # at the start of the timed region.  (This did NOT remove the sporadic slow step -- one step of +0.5 .. +2 ms in about one default run in
    # four on SOME boxes, none in thirty runs on others, with or without this: profiles/r03_experiments/step_outliers.txt)
    _C.timing_enable(True)
    if wl.sharded:
        # (a tile-row step contains collectives: every rank must run the SAME number of steps -- a time-based loop would let the
        # ranks disagree and deadlock in the exchange; found by the one-GPU self-test of round 3, `--test-one-gpu`)
        for _ in range(10 if prewarm_seconds > 0 else 0):
            wl.step()
        torch.cuda.synchronize(dev)
    else:
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < prewarm_seconds:
            wl.step()
            torch.cuda.synchronize(dev)
    # Everything that takes host time without GPU work happens BEFORE the warm-up steps, so that the timed region follows the warm-up with
    # nothing but the contract's barrier + synchronise in between: a gap of tens of milliseconds there (a gc pass, 512 + K hipEvent
    # creations) lets the device fall back to its idle clocks, and the first ten timed steps then measure the ramp (seen with --per-step:
    # 2.89, 2.73, 2.67, 2.65, ... 2.50 ms).
    import gc
    gc.collect()
    gc.disable()  # (a collection inside the timed region is a multi-millisecond stall of the launching thread)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    stream = torch.cuda.current_stream(dev)
    for m in marks:   # torch creates the hipEvent at the first record(): do that HERE, not inside the timed region
        m.record(stream)
    for _ in range(warmup):
        wl.step()
    barrier()
    _C.timing_enable(True)  # (resets the stage means: hipEvents around every stage of every timed step, on the launch stream, no extra sync)
    t0 = time.perf_counter()
    marks[0].record(stream)
    cum = []
    for i in range(steps):
        wl.step()
        marks[i + 1].record(stream)
        if per_step:  # debug: per-step wall time (adds a sync per step, invalidates the headline number)
            torch.cuda.synchronize(dev)
            cum.append(round(1000.0 * (time.perf_counter() - t0), 3))
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    stage_ms = {k: v for k, v in _C.timing_read(dev).items() if v >= 0}  # means over the timed region
    hist = _C.timing_history(dev, capacity=steps)                        # ... and per call (one forward + backward per step), chronological
    hist_host = _C.timing_history(dev, capacity=steps, host=True)        # ... and the launching thread's own time inside each stage
    _C.timing_enable(False)
    seq = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    per = sorted(seq)
    stats = {"median": round(per[len(per) // 2] if len(per) % 2 else 0.5 * (per[len(per) // 2 - 1] + per[len(per) // 2]), 4),
             "min": round(per[0], 4), "max": round(per[-1], 4),
             "how": "intervals between hipEvents recorded on the launch stream after every step of the timed region"} if per else {}
    if per and len(hist) == steps:
        # which step was the slowest, and where inside it: its six stage times beside those of the median step (the stage events sit on the launch
        # stream, so a stall of the launching thread shows up in the stage it interrupted, and `outside_stages` is what no stage interval covers)
        iw = max(range(steps), key=lambda i: seq[i])
        im = min(range(steps), key=lambda i: abs(seq[i] - stats["median"]))
        rec = lambda i: {"index": i, "ms": round(seq[i], 4), "stage_ms": {k: round(v, 4) for k, v in hist[i].items()},
                         "outside_stages": round(seq[i] - sum(hist[i].values()), 4),
                         "host_ms_in_stage": {k: round(v, 4) for k, v in hist_host[i].items()} if len(hist_host) == steps else None}
        stats["worst_step"] = rec(iw)
        stats["median_step"] = rec(im)
        stats["reading"] = ("stage_ms = GPU interval between the stage's events on the launch stream; host_ms_in_stage = the launching thread's own time between "
                            "recording them (microseconds when the launches are asynchronous).  A stage that is long in BOTH was waiting for its launches: a late host "
                            "(descheduled thread, blocking driver call), not a slow kernel")
    return dt, stage_ms, stats, cum


def describe(wl, stage_ms, ms_per_step, recording_allowed=True):
    """Sizes of the frame (from one extra untimed forward), SURVEY 8(d) byte models, the dominant kernel and its roofline."""
    _C, scene, sdict, rs, fwd_only = wl._C, wl.scene, wl.sdict, wl.rs, wl.fwd_only
    radii = wl.state["radii"]
    P = scene.P
    P_v = int((radii > 0).sum().item())
    N = scene.W * scene.H
    T = ((scene.W + 15) // 16) * wl.gy
    mode = int(sdict["sort_settings"]["sort_mode"])
    order = int(sdict["sort_settings"]["sort_order"])
    S = 1 if (mode != 0 or order >= 2) else 0
    Kf = 1 if order >= 2 else 0
    E = 1 if (sdict["culling_settings"]["tile_based_culling"] or order == 3) else 0
    ewa = bool(sdict["proper_ewa_scaling"])
    head, mid = int(sdict["sort_settings"]["queue_sizes"]["per_pixel"]), int(sdict["sort_settings"]["queue_sizes"]["tile_2x2"])
    cull = bool(sdict["culling_settings"]["hierarchical_4x4_culling"])
    recording = mode in (2, 3) and not fwd_only and _C.backward_mode() != "resort" and not wl.sharded and recording_allowed
    # R (tile-list entries) and B (blended (pixel, entry) pairs) from ONE extra untimed forward through _C directly: a
    # forward-only run has no grad_fn to ask, and the timed graphs are gone
    empty = torch.Tensor([])
    d1 = dict(sdict)
    if recording:
        d1["_record_blend_log"] = True
    o1 = _C.rasterize_gaussians(rs.bg, wl.means3D.detach(), empty, wl.opac.detach(), wl.scales.detach(), wl.rots.detach(), 1.0, empty,
                                rs.viewmatrix, rs.projmatrix, rs.inv_viewprojmatrix, rs.tanfovx, rs.tanfovy, rs.image_height,
                                rs.image_width, wl.shs.detach(), rs.sh_degree, rs.campos, False, d1, False, False)
    R = int(o1[0])
    B = 0
    if recording or mode in (0, 2):   # n_contrib = blends per pixel (recording forwards), list positions visited (GLOBAL / k-buffer)
        B = int(_C.image_array(o1[5], scene.W, scene.H, "n_contrib").to(torch.int64).sum().item())
    _C.release_scratch(o1[5]); _C.release_scratch(o1[4])
    del o1
    M = 16
    alg = survey_bytes(P, P_v, R, N, T, M, S, mode, Kf, E, 1, ewa)
    des = design_bytes(P, P_v, R, N, T, M, S, mode, Kf, E, 1, B if recording else 0)
    fwd_keys, bwd_keys = ("preprocess", "scan", "duplicate", "sort", "ranges", "render_fwd"), ("zero_fill", "render_bwd", "bwd_cov2D", "bwd_preprocess")
    fwd_bytes = sum(alg[k] for k in fwd_keys)
    bwd_bytes = sum(alg[k] for k in bwd_keys)
    step_bytes = fwd_bytes + (0 if fwd_only else bwd_bytes)

    def roof_for(dom):
        """The roofline object of one of the two render kernels ("Render" / "BwdRender")."""
        dom_key = "render_bwd" if dom == "BwdRender" else "render_fwd"
        dom_ms = stage_ms.get(dom, float("nan"))
        achieved = alg[dom_key] / (dom_ms * 1e-3) / 1e9 if dom_ms and dom_ms > 0 else float("nan")
        if mode == 3:
            # (last template argument: the depth keys' reciprocal without its domain check -- the default queue sizes'
            # forward passes on a frame whose Sigma^-1 is tame, which every synthetic workload is)
            frcp = "true" if (head == 4 and mid == 8) else "false"
            kname = ("render_replay_kernel<false>" if recording else f"render_hier_kernel<{head}, {mid}, {str(cull).lower()}, 1, false>") if dom == "BwdRender" \
                else f"render_hier_kernel<{head}, {mid}, {str(cull).lower()}, {2 if recording else 0}, {frcp}>"
        elif mode == 2:
            win = next(w for w in (1, 2, 4, 8, 12, 16, 20, 24) if head <= w or w == 24)
            # (launch_kbuffer_wave: windows >= 8 run the LDS-ring kernel unless STP_KBUFFER=wave asks for the register window)
            fwd_kernel = "render_kbuffer_ring_kernel" if (win >= 8 and os.environ.get("STP_KBUFFER", "") != "wave") else "render_kbuffer_wave_kernel"
            kname = "render_replay_kernel<true>" if (dom == "BwdRender" and recording) else \
                (f"render_kbuffer_kernel<{win}, 1>" if dom == "BwdRender" else
                 f"{fwd_kernel}<{win}, {2 if recording else 0}, true>")
        else:
            kname = {0: "render_global", 1: "render_full"}[mode] + ("_bwd_kernel" if dom == "BwdRender" else "_fwd_kernel")
        prof, prof_note = profile_entry(kname, f"{wl.name}-{wl.variant}")
        # What bounds the kernel (PMC passes under profiles/): the re-sorting forwards issue VALU instructions 93-96 % of the time; the replay waits for the LDS atomic unit
        # (sums on shared addresses: VALU 68 %, LDS 51 % busy, and neither more waves nor spreading the adds out helps -- profiles/EXPERIMENTS.md, round 5)
        bound = "hbm" if mode not in (2, 3) else ("lds" if kname.startswith("render_replay_kernel") else "valu")
        note = {"hbm": "streaming kernel: achieved / peak / frac are SURVEY 8(d) bytes per launch over the launch duration against 8 TB/s",
                "valu": "what bounds this kernel is VALU issue (see \"roofline_valu\" / \"valu\"): achieved / peak / frac here are its HBM figures -- SURVEY 8(d) "
                        "bytes per launch over the launch duration against 8 TB/s -- which the contract asks for",
                "lds": "what bounds this kernel is the LDS atomic unit (64-bit fixed-point sums of lanes that share list positions; see \"valu\" for its VALU side): achieved / peak / "
                       "frac here are its HBM figures -- SURVEY 8(d) bytes per launch over the launch duration against 8 TB/s -- which the contract asks for"}[bound]
        roofline = {"bound": bound, "priced_on": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": int(prof["hbm_bytes_per_launch"]) if prof else None,
                    "traffic_source": prof_note,
                    "algorithmic_bytes_per_launch": int(alg[dom_key]), "bytes_model": "SURVEY.md section 8(d), verbatim",
                    "design_bytes_per_launch": int(des[dom_key]), "avg_launch_ms": round(dom_ms, 4),
                    "whole_step_frac": round((step_bytes / (ms_per_step * 1e-3) / 1e9) / HBM_PEAK_GBS, 5),
                    "note": note}
        return roofline, kname, dom_ms, prof, prof_note

    # the dominant kernel is the longer of the two render kernels; at C2-full they are within 1 % of each other and which one leads changes from box to box: the other one's
    # object rides along as "second" so that both are on every line
    dom = "BwdRender" if (not fwd_only and stage_ms.get("BwdRender", 0) >= stage_ms.get("Render", 0)) else "Render"
    roofline, kname, dom_ms, prof, prof_note = roof_for(dom)
    if not fwd_only and "BwdRender" in stage_ms:
        second = roof_for("Render" if dom == "BwdRender" else "BwdRender")[0]
        roofline["second"] = {k: second[k] for k in ("bound", "kernel", "achieved", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms")}
    info = {"P": P, "P_visible": P_v, "num_rendered": R, "tiles": T, "blended_pairs": B, "mode": mode, "order": order,
            "alg": alg, "des": des, "fwd_bytes": fwd_bytes, "bwd_bytes": bwd_bytes, "kname": kname, "dom_ms": dom_ms,
            "prof": prof, "prof_note": prof_note}
    return roofline, info


def hbm_ceilings(dev):
    """Measured HBM ceilings of THIS box in THIS run: the library's own float4 streaming kernels (stp_hbm_probe: read / write / copy, eight
    16-byte accesses in flight per thread, non-temporal) on 1 GiB, best of 10 over a few grid sizes -- what a streaming kernel can reach
    here, next to the 8 TB/s spec peak the roofline fractions are priced on (MI355X_MICROARCH.md measures 6.29 TB/s for a copy).  torch's
    generic kernels on the same buffers are kept as a side object: `sum` is a reduction kernel, not a read ceiling."""
    from diff_gaussian_rasterization import _C
    n = 1 << 28
    x = torch.ones(n, device=dev); y = torch.empty_like(x)

    def best(f, reps=10):
        for _ in range(2):
            f()
        torch.cuda.synchronize(dev)
        b = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); e1.synchronize()
            b = min(b, e0.elapsed_time(e1))
        return b
    gb = n * 4 / 1e9
    out = {}
    grids = (2048, 4096, 8192, 16384)
    for kind, args, factor in (("read", (None, x), 1), ("write", (y, None), 1), ("copy", (y, x), 2)):
        t = {(g, nt): best(lambda g=g, nt=nt: _C.hbm_probe(kind, args[0], args[1], blocks=g, nontemporal=nt), reps=6) for g in grids for nt in (False, True)}
        g = min(t, key=t.get)
        out[f"{kind}_GBps"] = round(1e3 * factor * gb / t[g], 1)
        out[f"{kind}_config"] = {"blocks": g[0], "nontemporal": g[1]}
    out["how"] = ("stp_hbm_probe (csrc/stp_hbm_probe.hip): float4 streaming kernels of the library, 1 GiB fp32, best of 6 launches over grids of "
                  "2048..16384 workgroups x plain / non-temporal accesses, this box, this run; copy counts read + write bytes")
    out["torch_kernels"] = {"sum_GBps": round(1e3 * gb / best(lambda: x.sum()), 1), "fill_GBps": round(1e3 * gb / best(lambda: y.fill_(2.0)), 1),
                            "copy_GBps": round(1e3 * 2 * gb / best(lambda: y.copy_(x)), 1),
                            "how": "torch sum / fill_ / copy_ on the same buffers (the round 2-4 definition of hbm_measured; sum is a reduction kernel, not a read ceiling)"}
    del x, y
    torch.cuda.empty_cache()
    return out


FMA_LIB_NAME = "libstp_raster_fma.so"
GRAD_NAMES = ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh")


def product_frame(wl):
    """One untimed step of `wl` with the library currently loaded + one direct _C forward: what the whole-frame parity records
    compare -- image, gradients, num_rendered, the 64-bit sort keys and the sorted Gaussian list (host copies)."""
    _C, rs = wl._C, wl.rs
    for x in wl.leaves:
        if x is not None:
            x.grad = None
    wl.step()
    color = wl.state["color"].detach().cpu().numpy()
    grads = None
    if not wl.fwd_only:
        grads = {n: (None if x.grad is None else x.grad.detach().cpu().numpy())
                 for n, x in zip(GRAD_NAMES, (wl.means3D, wl.means2D, wl.opac, wl.scales, wl.rots, wl.shs))}
    empty = torch.Tensor([])
    o1 = _C.rasterize_gaussians(rs.bg, wl.means3D.detach(), empty, wl.opac.detach(), wl.scales.detach(), wl.rots.detach(), 1.0, empty,
                                rs.viewmatrix, rs.projmatrix, rs.inv_viewprojmatrix, rs.tanfovx, rs.tanfovy, rs.image_height,
                                rs.image_width, wl.shs.detach(), rs.sh_degree, rs.campos, False, dict(wl.sdict), False, False)
    R = int(o1[0])
    keys = _C.binning_array(o1[4], R, "keys").cpu().numpy().view(np.uint64) if R else np.zeros(0, np.uint64)
    plist = _C.binning_array(o1[4], R, "point_list").cpu().numpy().view(np.uint32) if R else np.zeros(0, np.uint32)
    sc = wl.scene
    # what explains a moved pixel (oracle/explain.py): state both sides share bit for bit + the product's final transmittance
    state = {"W": sc.W, "H": sc.H, "point_list": plist,
             "ranges": _C.image_array(o1[5], sc.W, sc.H, "ranges").cpu().numpy().view(np.uint32).reshape(-1),
             "conic_opacity": _C.geometry_array(o1[3], sc.P, wl.sdict, "conic_opacity").cpu().numpy().reshape(-1),
             "means2D": _C.geometry_array(o1[3], sc.P, wl.sdict, "means2D").cpu().numpy().reshape(-1),
             "final_T": _C.image_array(o1[5], sc.W, sc.H, "final_T").cpu().numpy().reshape(-1),
             "cull_4x4": bool(wl.sdict["culling_settings"]["hierarchical_4x4_culling"]) and int(wl.sdict["sort_settings"]["sort_mode"]) == 3}
    _C.release_scratch(o1[5]); _C.release_scratch(o1[4])
    del o1
    return {"color": color, "grads": grads, "num_rendered": R, "keys": keys, "point_list": plist, "state": state}


def explain_differences(prod, other_color, other_final_T, other_grads=None, y0=0, y1=None):
    """The residual of a parity record, explained or not (oracle/explain.py: checker-side code): every pixel that moved by more than 2e-6 must
    have an entry of its tile list whose alpha -- fp32 exponent, exponential in double -- sits within 6e-7 of 1/255 (per pixel or, with 4x4
    culling, at its sub-tile's point of maximum contribution), or a final transmittance within 1e-6 of 1e-4; every Gaussian whose gradient
    is off by more than 1e-4 must be blended at such a pixel.  y0 / y1: pixel rows the comparison covers."""
    from oracle import explain
    st = prod["state"]
    W, H = st["W"], st["H"]
    y1 = H if y1 is None else y1
    d = np.abs(np.asarray(prod["color"], np.float64)[:, y0:y1] - np.asarray(other_color, np.float64)[:, y0:y1])
    moved = np.zeros((H, W), bool)
    moved[y0:y1] = (d > 2e-6).any(axis=0)
    ex = explain.explain_moved_pixels(moved, W=W, H=H, ranges=st["ranges"], point_list=st["point_list"], conic_opacity=st["conic_opacity"],
                                      means2D=st["means2D"], final_T_a=st["final_T"], final_T_b=other_final_T, cull_4x4=st["cull_4x4"])
    out = {"pixels_moved_gt_2e-6": ex["pixels"], "explained": f"{ex['explained']}/{ex['pixels']}", "by": {k: v for k, v in ex["by"].items() if v},
           "decisions": [list(d) for d in ex["decisions"][:16]], "T_pixels": [list(d) for d in ex["T_pixels"][:16]],
           "criterion": "an entry's alpha (fp32 exponent, exp in double) within 6e-7 of 1/255 at the pixel or at its 4x4 sub-tile's culling point, or a final T within 1e-6 of 1e-4"}
    if ex["unexplained"]:
        out["unexplained_pixels"] = ex["unexplained"][:8]
    if other_grads is not None and prod.get("grads") is not None:
        per_g = None
        for k, a in prod["grads"].items():
            b = other_grads.get(k)
            if a is None or b is None or np.size(b) == 0:
                continue
            a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
            if k == "dL_dmeans2D":
                a, b = a[:, :2], b[:, :2]
            e = np.abs(a - b).reshape(a.shape[0], -1).max(axis=1) / max(float(np.max(np.abs(b))), 1e-30)
            per_g = e if per_g is None else np.maximum(per_g, e)
        if per_g is not None:
            over = np.nonzero(per_g > 1e-4)[0]
            out["gaussians_over_1e-4_explained"] = f"{sum(int(i) in ex['gaussians'] for i in over)}/{over.size}"
    return out



class ReferenceFrame:
    """THE REFERENCE'S OWN KERNELS (oracle/_ref: hipify-perl + hipcc build, test infrastructure) on the whole frame of a
    workload: image, gradients, keys, list -- run once per workload, compared with as many product builds as wanted."""

    def __init__(self, scene, sdict, fwd_only, variant="ieee"):
        from oracle import reference as ref
        self.note = None
        rf = ref.forward_scene(scene, sdict, variant=variant)
        self.build = ref.build_info(variant)
        self.num_rendered = int(rf.num_rendered)
        self.color = rf.color
        self.keys = rf.array("keys").view(np.uint64) if self.num_rendered else np.zeros(0, np.uint64)
        self.point_list = rf.array("point_list").view(np.uint32) if self.num_rendered else np.zeros(0, np.uint32)
        self.grads = None if fwd_only else rf.backward(scene.dL_dout)
        self.final_T = rf.array("final_T").reshape(-1)
        self.rf = rf

    def compare(self, prod):
        rec = {"build": self.build, "frame": "whole frame", "num_rendered_equal": bool(prod["num_rendered"] == self.num_rendered)}
        if rec["num_rendered_equal"]:
            kd = int((prod["keys"] != self.keys).sum())
            ld = int((prod["point_list"] != self.point_list).sum())
            rec.update({"keys_bit_equal": kd == 0, "keys_differing": kd, "list_bit_equal": ld == 0, "list_entries_differing": ld,
                        "list_entries": int(self.keys.size)})
        rec.update(_img_err(prod["color"], self.color))
        if prod["grads"] is not None and self.grads is not None:
            rec.update(_grad_err(prod["grads"], self.grads))
        if rec["num_rendered_equal"] and rec.get("list_bit_equal") and (rec["pixels_moved_gt_2e-6"] or rec.get("gaussians_over_1e-4")):
            try:   # (only meaningful when both sides walk the same lists)
                rec["residual"] = explain_differences(prod, self.color, self.final_T, self.grads)
            except Exception as ex:
                rec["residual"] = {"error": repr(ex)[:200]}
        if self.note:
            rec["note"] = self.note
        return rec

    def free(self):
        if self.rf is not None:
            self.rf.free()
            self.rf = None


def forced_oracle_closure(scene, sdict, fwd_only, prod, res):
    """A whole-frame residual against the reference whose every moved pixel is explained by a per-pixel alpha or transmittance decision on its threshold: the CPU oracle
    is run ONCE on the whole frame with exactly those decisions taken the other way (oracle.forced_alpha_flips) and the product compared with it -- a complete
    explanation leaves no pixel above 2e-6 and no gradient above 1e-4.  Checker-side only; None when there is nothing of that kind to force."""
    from oracle import oracle as orc
    by = res.get("by") or {}
    dec = [tuple(d) for d in res.get("decisions", [])]
    tpx = [tuple(d) for d in res.get("T_pixels", [])]
    n_moved = int(res.get("pixels_moved_gt_2e-6", 0))
    if not (dec or tpx) or by.get("subtile_cull") or res.get("explained") != f"{n_moved}/{n_moved}" or n_moved > 16:
        return None
    t0 = time.perf_counter()
    with orc.forced_alpha_flips(dec, scene.W, T_pixels=tpx):
        f2 = orc.forward_scene(scene, sdict)
        g2 = None if (fwd_only or prod.get("grads") is None) else f2.backward(scene.dL_dout)
    rec = {"oracle_run_with": {"forced_alpha_flips": [list(d) for d in dec], "forced_T_flips": [list(d) for d in tpx]}, **_img_err(prod["color"], f2.color)}
    if g2 is not None:
        rec.update(_grad_err(prod["grads"], g2))
    f2.free()
    rec["closes_the_residual"] = bool(rec["pixels_moved_gt_2e-6"] == 0 and not rec.get("gaussians_over_1e-4"))
    rec["oracle_seconds"] = round(time.perf_counter() - t0, 1)
    rec["what"] = "product vs the CPU oracle run on the whole frame with the explained decision(s) taken the other way and nothing else changed"
    return rec


def reference_parity(wl, dev, timing_steps=10, timing_warmup=3, with_fma=True, time_fma=True):
    """Whole-frame parity of one workload against the reference's IEEE build, for the default library and for the second shipped
    library (libstp_raster_fma.so), and the latter's speed in the same harness.  Returns {} when oracle/_ref is absent."""
    from oracle import reference as ref
    from diff_gaussian_rasterization import _C
    if not ref.available("ieee"):
        return {}
    out = {}
    rfr = ReferenceFrame(wl.scene, wl.sdict, wl.fwd_only)
    try:
        prod = product_frame(wl)
        first = rfr.compare(prod)
        if not first.get("keys_bit_equal", False) and wl.sdict.get("load_balancing") and wl.sdict["culling_settings"]["tile_based_culling"]:
            # load_balancing + tile_based_culling: the reference computes its write offsets with `0xFFFFFFFFU >> (32 - lane)`
            # (stopthepop_common.cuh:519-520), a shift by 32 for lane 0 -- undefined; NVIDIA hardware clamps it to 0, gfx950 wraps it to a shift
            # by 0, so THIS build of the reference emits a different list wherever a Gaussian takes that path (more than 32 tiles).  On CUDA
            # the flag changes no result: the reference is re-run without it and both records are kept.
            rfr.free()
            rfr = ReferenceFrame(wl.scene, {**wl.sdict, "load_balancing": False}, wl.fwd_only)
            rfr.note = ("reference run with load_balancing=false: with it, stopthepop_common.cuh:519-520 shifts by 32 (undefined; this build of the "
                        "reference then emits another list, see as_configured); the flag changes no result on CUDA")
            out["vs_reference_ieee_build"] = rfr.compare(prod)
            out["vs_reference_ieee_build"]["as_configured"] = {k: first[k] for k in first if k != "build"}
        else:
            out["vs_reference_ieee_build"] = first
        try:   # the residual, closed (see forced_oracle_closure)
            res = out["vs_reference_ieee_build"].get("residual")
            if isinstance(res, dict) and "error" not in res:
                closure = forced_oracle_closure(wl.scene, wl.sdict, wl.fwd_only, prod, res)
                if closure is not None:
                    res["forced_oracle"] = closure
        except Exception as ex:
            out["vs_reference_ieee_build"]["residual"]["forced_oracle"] = {"error": repr(ex)[:200]}
        fma = os.path.join(os.path.dirname(_C.library_path()), FMA_LIB_NAME)
        if with_fma and os.path.exists(fma) and os.path.basename(_C.library_path()) != FMA_LIB_NAME:
            _C.use_library(fma)
            wl._C = _C
            try:
                rec = {"library": FMA_LIB_NAME, "what": "the second shipped build: depthAlongRay as fused multiply-add chains (the default of rounds 1-3)"}
                rec["vs_reference_ieee_build"] = rfr.compare(product_frame(wl))
                if time_fma:
                    dt, stage_ms, stats, _ = timed_region(wl, timing_steps, timing_warmup, 0.0, lambda: torch.cuda.synchronize(dev))
                    rec.update({"value": round(timing_steps / dt, 3), "unit": "frames/s", "ms_per_step": round(1000.0 * dt / timing_steps, 4),
                                "steps": timing_steps, "step_ms": {k: stats[k] for k in ("median", "min", "max")} if stats else {},
                                "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()}})
                out["fma_depth_build"] = rec
            finally:
                _C.use_library(None)
                _C.clear_scratch_pool(dev)
    finally:
        rfr.free()
    return out


OTHER_WORKLOADS = (("C2-min", "C2", "min", False), ("C3", "C3", "full", False), ("C4-1gpu-fwd", "C4", "full", True), ("C5", "C5", "full", False))


def other_workloads(dev, steps=10, warmup=3, parity=True, detail=False, scale=1.0):
    """The BASELINE configurations that are not the headline, `steps` timed steps each on the same code in the same process
    (same harness as the headline: wall clock between two synchronisations, stage hipEvents, SURVEY 8(d) bytes)."""
    out = {}
    for label, name, variant, fwd_only in OTHER_WORKLOADS:
        try:
            wl = Workload(name, variant, dev, fwd_only=fwd_only, scale=scale)
            dt, stage_ms, stats, _ = timed_region(wl, steps, warmup, 0.2, lambda: torch.cuda.synchronize(dev))
            ms = 1000.0 * dt / steps
            roof, info = describe(wl, stage_ms, ms)
            out[label] = {"value": round(steps / dt, 3), "unit": "frames/s", "ms_per_step": round(ms, 4), "steps": steps, "warmup": warmup,
                          "step_ms": {k: stats[k] for k in ("median", "min", "max", "worst_step", "median_step") if k in stats} if stats else {},
                          "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
                          "P": info["P"], "num_rendered": info["num_rendered"], "blended_pairs": info["blended_pairs"],
                          "resolution": f"{wl.scene.W}x{wl.scene.H}", "passes": "fwd" if wl.fwd_only else "fwd+bwd",
                          "dominant_kernel": roof["kernel"], "dominant_ms": roof["avg_launch_ms"],
                          "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"], "achieved_GBps": roof["achieved"],
                          "frac": roof["frac"], "traffic": roof["traffic"], "traffic_source": roof["traffic_source"]}
            t_leg = time.perf_counter()
            if parity:
                try:   # whole frame against the reference's own kernels (IEEE build), default library and libstp_raster_fma.so (+ its speed)
                    out[label].update(reference_parity(wl, dev, timing_steps=steps, timing_warmup=warmup, with_fma=detail, time_fma=detail and label in ("C3", "C5")))
                except Exception as ex:
                    out[label]["vs_reference_error"] = repr(ex)[:300]
            out[label]["checker_seconds"] = round(time.perf_counter() - t_leg, 1)
            wl.free()
            del wl
        except Exception as ex:  # the headline stands on its own
            out[label] = {"error": repr(ex)[:300]}
        wl = None
        from diff_gaussian_rasterization import _C
        _C.clear_scratch_pool(dev)
        torch.cuda.empty_cache()
    return out

LINE_CAP_BYTES = 8192       # the driver keeps a bounded tail of stdout: round 5's 24.7 KB line could not be parsed
STR_CAP = 140               # (and shortens long strings)


def _short(x, n=STR_CAP):
    return x if not isinstance(x, str) or len(x) <= n else x[:n - 1] + "~"


def _parity_summary(rec):
    """<= 300 bytes of one parity record (oracle window or whole frame vs the reference build): what a reader needs to see that the
    frame is the reference algorithm's; the decisions, criteria and records behind it stay in bench_detail.json."""
    if not isinstance(rec, dict):
        return None
    o = {}
    for k_in, k_out in (("keys_bit_equal", "keys_equal"), ("list_bit_equal", "list_equal"), ("psnr_db", "psnr_db")):
        if k_in in rec:
            o[k_out] = rec[k_in]
    if "max_abs" in rec:
        o["max_abs"] = float(f"{rec['max_abs']:.3g}")
    if "pixels_moved_gt_2e-6" in rec:
        o["moved"] = rec["pixels_moved_gt_2e-6"]
    if "grad_rel_max" in rec:
        o["grad_rel_max"] = float(f"{rec['grad_rel_max']:.3g}")
    res = rec.get("residual")
    if isinstance(res, dict):
        if "explained" in res:
            o["explained"] = res["explained"]
        closure = res.get("forced_oracle") or res.get("nudged_oracle")
        if isinstance(closure, dict) and "closes_the_residual" in closure:
            o["closed"] = closure["closes_the_residual"]
            if "max_abs" in closure:
                o["closed_max_abs"] = float(f"{closure['max_abs']:.3g}")
            if "grad_rel_max" in closure:
                o["closed_grad_rel_max"] = float(f"{closure['grad_rel_max']:.3g}")
        if "error" in res:
            o["residual_error"] = _short(res["error"], 80)
    return o


def compact_line(out):
    """The ONE stdout line: the contract's keys + config + stage_ms + step_ms{median,min,max} + roofline (with `second`) + roofline_valu +
    cpu_baseline + a short parity summary + one compact object per other workload.  Everything else the run produced (worst / median step
    records, decisions, criteria, as_configured, side legs) is in bench_detail.json next to this script and on stderr."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    o = {k: out[k] for k in keep if k in out}
    cfg = dict(out.get("config", {}))
    cfg["workload"] = _short(cfg.get("workload", ""))
    cfg.pop("step", None)
    o["config"] = cfg
    o["stage_ms"] = out.get("stage_ms")
    sm = out.get("step_ms") or {}
    o["step_ms"] = {k: sm[k] for k in ("median", "min", "max") if k in sm}
    rf = out.get("roofline")
    if isinstance(rf, dict):
        r = {k: rf[k] for k in ("bound", "priced_on", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                                "design_bytes_per_launch", "avg_launch_ms", "whole_step_frac") if k in rf}
        if rf.get("traffic") is None:
            r["traffic_note"] = _short(rf.get("traffic_source", ""), 100)
        if "second" in rf:
            r["second"] = rf["second"]
        o["roofline"] = r
    rv = out.get("roofline_valu")
    if isinstance(rv, dict):
        o["roofline_valu"] = {k: rv[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "limits_the_kernel") if k in rv}
    if "cpu_baseline" in out:
        o["cpu_baseline"] = {k: _short(v) for k, v in out["cpu_baseline"].items()}
    hm = out.get("hbm_measured")
    if isinstance(hm, dict):
        o["hbm_measured"] = {k: hm[k] for k in ("read_GBps", "write_GBps", "copy_GBps") if k in hm}
    par = out.get("parity")
    if isinstance(par, dict):
        p = {"against": "CPU oracle (window) + the reference's sources built by hipify-perl+hipcc (whole frame): by the tier rules 'parity unpinned'",
             "window": par.get("window"), "oracle": _parity_summary(par)}
        if "vs_reference_ieee_build" in par:
            p["reference_build"] = _parity_summary(par["vs_reference_ieee_build"])
        if "vs_reference_error" in par:
            p["reference_error"] = _short(par["vs_reference_error"], 100)
        o["parity"] = p
    for k in ("rccl_ranks", "tile_shard_exchange"):
        if k in out:
            o[k] = _short(out[k])
    for k in ("frame_shard", "tile_shard"):
        if isinstance(out.get(k), dict):
            o[k] = {kk: _short(vv) for kk, vv in out[k].items() if kk != "what"}
    ow = out.get("other_workloads")
    if isinstance(ow, dict):
        c = {}
        for label, w in ow.items():
            if "error" in w:
                c[label] = {"error": _short(w["error"], 100)}
                continue
            e = {k: w[k] for k in ("value", "ms_per_step", "dominant_kernel", "dominant_ms", "frac") if k in w}
            e["traffic"] = w.get("traffic")
            e["stage_ms"] = w.get("stage_ms")
            rec = w.get("vs_reference_ieee_build")
            if isinstance(rec, dict):
                e.update(_parity_summary(rec))
            if "vs_reference_error" in w:
                e["reference_error"] = _short(w["vs_reference_error"], 100)
            c[label] = e
        o["other_workloads"] = c
    if "leg_seconds" in out:
        o["leg_seconds"] = out["leg_seconds"]
    o["detail"] = "bench_detail.json (+ stderr): step records, decisions, criteria, side legs"
    return o


def emit(result_fd, out):
    """Full record -> bench_detail.json + stderr; compact record -> the one stdout line, hard-capped."""
    full = json.dumps(out)
    for d in (ROOT, os.environ.get("TMPDIR", "/tmp")):
        try:
            with open(os.path.join(d, "bench_detail.json"), "w") as f:
                f.write(full + "\n")
            break
        except OSError:
            continue
    print("[bench detail] " + full, file=sys.stderr, flush=True)
    o = compact_line(out)
    text = json.dumps(o)
    # shed the optional objects, cheapest first, rather than lose the line (cannot happen with today's field set: a guard, not a plan)
    for k in ("leg_seconds", "hbm_measured", "roofline_valu", "other_workloads"):
        if len(text) < LINE_CAP_BYTES:
            break
        o.pop(k, None)
        text = json.dumps(o)
    assert len(text) < LINE_CAP_BYTES, f"bench line is {len(text)} bytes: the driver cannot parse more than {LINE_CAP_BYTES}"
    os.write(result_fd, (text + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C2", choices=["C1", "C2", "C3", "C4", "C5", "L1", "C2L", "C2H"])
    ap.add_argument("--variant", default="full", choices=["full", "min"])
    ap.add_argument("--shard", default=None, choices=["tilerows", "frames"],
                    help="N > 1: what the ranks share.  Default: tilerows (north star: ONE frame partitioned by screen-tile row, strips "
                         "gathered over RCCL, strong scaling); frames = every rank renders its own frame, no collective, weak scaling")
    ap.add_argument("--no-frame-shard-probe", action="store_true",
                    help="N > 1 with --shard tilerows: do NOT additionally time the frame-sharded mode (the \"frame_shard\" object of the JSON line)")
    ap.add_argument("--tile-shard-probe", action="store_true", help="(accepted, no effect)")
    ap.add_argument("--no-tile-shard-probe", action="store_true", help="(accepted; N > 1 with --shard frames: no tile-row side measurement)")
    ap.add_argument("--probe-timeout", type=float, default=120.0, help="seconds the side measurement may take before the headline is printed without it")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; invalidates the headline number)")
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--with-loss", action="store_true",
                    help="the step also computes a weighted-sum loss with torch ((image * w).sum().backward(), the step of rounds 1-2) instead of "
                         "handing dL/dimage, resident in HBM, to the rasterizer's backward")
    ap.add_argument("--prewarm-seconds", type=float, default=0.5, help="untimed steps before the W warm-up steps (device clock ramp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--detail", action="store_true",
                    help="also run the side legs of rounds 3-5 (libstp_raster_fma.so parity + speed, the reference's default-contraction build, the "
                         "reference's kernels timed on this GPU, the step with a torch loss): minutes of checker time, all of it in bench_detail.json")
    ap.add_argument("--no-other-workloads", action="store_true", help="N == 1: leave out the other_workloads object (C2-min, C3, C4 on one GPU, C5; 10 steps each)")
    ap.add_argument("--cpu-rows", type=int, default=0, help="tile rows in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--train-forward-only", action="store_true",
                    help="debug: forward passes that expect a backward (recording forward) without running it; read stage_ms only")
    ap.add_argument("--per-step", action="store_true", help="debug: print cumulative wall time after each timed step")
    ap.add_argument("--force-shard", action="store_true", help="use the tile-row sharded path even with one rank (self-test of the exchange code)")
    ap.add_argument("--test-one-gpu", action="store_true",
                    help="self-test of the N > 1 code path on a box with ONE GPU: every rank uses cuda:0 and the process group is gloo (host-staged "
                         "exchange); the numbers mean nothing")
    args = ap.parse_args()
    t_start = time.perf_counter()

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
    dev = torch.device("cuda", 0 if args.test_one_gpu else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    rccl_ranks = None
    if world > 1 or args.force_shard:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.test_one_gpu:
            dist_mod.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist = dist_mod
        coll_dev = torch.device("cpu") if args.test_one_gpu else dev   # (gloo: small control tensors live on the host)
        # the first collective: every rank contributes 1, so the sum IS the number of ranks RCCL connected
        ones = torch.ones(1, device=coll_dev, dtype=torch.int32)
        dist.all_reduce(ones)
        rccl_ranks = {"world_size": int(dist.get_world_size()), "all_reduce_of_ones": int(ones.item()), "backend": dist.get_backend()}

    from diff_gaussian_rasterization import _C

    fwd_only = args.fwd_only or args.workload == "C4"
    if args.shard is None:
        args.shard = "tilerows"
    sharded = (world > 1 or args.force_shard) and args.shard == "tilerows"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def measure(shard_rows, steps, warmup, prewarm):
        """One Workload + one timed region (contract: W warm-up steps, EXACTLY K steps between barrier + synchronise, MAX over ranks)."""
        w = Workload(args.workload, args.variant, dev, fwd_only=fwd_only, scale=args.scale, dist=dist, rank=rank, world=world,
                     sharded=shard_rows, train_forward_only=args.train_forward_only, loss=args.with_loss)
        dt_, stage_, stats_, cum_ = timed_region(w, steps, warmup, prewarm, barrier, per_step=args.per_step)
        if args.per_step and rank == 0:
            print("cumulative ms after each step:", cum_, "reserved GB:", round(torch.cuda.memory_reserved(dev) / 2**30, 2),
                  "num_alloc_retries:", torch.cuda.memory_stats(dev).get("num_alloc_retries"), "segments:",
                  torch.cuda.memory_stats(dev).get("segment.all.allocated"), file=sys.stderr, flush=True)
        if dist is not None:
            tt = torch.tensor([dt_], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt.item())
        return w, dt_, stage_, stats_

    def line(w, dt_, stage_ms, step_stats, shard_rows, steps):
        """The JSON line of one measurement (rank 0)."""
        scene, sdict = w.scene, w.sdict
        frames_per_step = world if (world > 1 and not shard_rows) else 1
        value = frames_per_step * steps / dt_
        ms_per_step = 1000.0 * dt_ / steps
        roofline, info = describe(w, stage_ms, ms_per_step)
        P, P_v, R, T, B, mode, order = info["P"], info["P_visible"], info["num_rendered"], info["tiles"], info["blended_pairs"], info["mode"], info["order"]
        des, prof, prof_note, kname, dom_ms = info["des"], info["prof"], info["prof_note"], info["kname"], info["dom_ms"]
        out = {
            "metric": "fwd+bwd frames/sec at 1920×1080, 1M Gaussians; PSNR vs reference",
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if shard_rows else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}-{args.variant}: {P} Gaussians, {scene.W}x{scene.H}, SH degree 3, "
                                   f"sort_mode={mode} sort_order={order} culling={sdict['culling_settings']} "
                                   f"{'fwd' if w.fwd_only else 'fwd+bwd'}",
                       "P": P, "P_visible": P_v, "num_rendered": R, "tiles": T, "blended_pairs": B,
                       "parallelism": (f"tilerows{world}" if shard_rows else f"frames{world}") if world > 1 else "single",
                       "scale": args.scale,
                       "step": ("rasterizer forward" if w.fwd_only else
                                "rasterizer forward + torch weighted-sum loss + backward ((image * w).sum().backward())" if w.loss else
                                "rasterizer forward + backward; dL/dimage resident in HBM like the other inputs (image.backward(dL_dimage))")},
            "step_ms": step_stats,
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "algorithmic_bytes": {"forward": int(info["fwd_bytes"]), "backward": int(info["bwd_bytes"]), "model": "SURVEY.md section 8(d)"},
            "design_bytes": {"forward": int(sum(des[k] for k in ("preprocess", "scan", "duplicate", "sort", "ranges", "gather", "render_fwd"))),
                             "backward": int(sum(des[k] for k in ("zero_fill", "render_bwd", "bwd_preprocess")))},
            "roofline": roofline,
        }
        if rccl_ranks is not None:
            out["rccl_ranks"] = rccl_ranks
        if prof and prof.get("SQ_INSTS_VALU") and dom_ms and dom_ms > 0:
            # VALU view of the dominant kernel, from pass 1 of the same committed profile (SQ counters are sums over the chip):
            # busy = SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (1024 SIMDs x launch duration x 2.4 GHz nominal);
            # lane_util = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64; insts_per_pair = wave-instructions x 64 lanes / B
            simd_cycles = 1024 * (float(prof.get("avg_ms_at_profile", dom_ms)) * 1e-3) * 2.4e9
            out["valu"] = {"kernel": kname, "busy": round(4.0 * prof["SQ_ACTIVE_INST_VALU"] / simd_cycles, 3),
                           "wave_insts_per_launch": int(prof["SQ_INSTS_VALU"]),
                           "insts_per_pair": round(64.0 * prof["SQ_INSTS_VALU"] / B, 1) if B else None,
                           "lane_util": round(prof["SQ_THREAD_CYCLES_VALU"] / prof["SQ_ACTIVE_INST_VALU"] / 64.0, 3),
                           "source": prof_note}
            # the VALU roofline beside the HBM one (the kernel is VALU-issue bound): wave-instructions per second against the chip's full-rate issue,
            # one wave64 fp32 instruction per SIMD every 2 cycles (157 TFLOP/s fp32 vector = 1024 SIMDs x 2.4 GHz x 64 lanes x 2 flop / 2 cycles);
            # most of this kernel's stream issues at half that rate (compares, selects, min/max, DPP operands: profiles/valu_rate_bench)
            t_prof = float(prof.get("avg_ms_at_profile", dom_ms)) * 1e-3
            peak = 1024 * 2.4e9 / 2.0
            out["roofline_valu"] = {"bound": "valu", "kernel": kname, "achieved": round(prof["SQ_INSTS_VALU"] / t_prof / 1e9, 1), "peak": round(peak / 1e9, 1),
                                    "unit": "G wave-instructions/s", "frac": round(prof["SQ_INSTS_VALU"] / t_prof / peak, 4),
                                    "how": "SQ_INSTS_VALU per launch / launch duration of the same rocprofv3 pass, against 1024 SIMDs x 2.4 GHz / 2 cycles",
                                    "limits_the_kernel": roofline["bound"] == "valu"}
        return out

    def summary(o):
        return {k: o[k] for k in ("value", "unit", "ms_per_step", "steps", "scaling", "stage_ms") if k in o} | {"parallelism": o["config"]["parallelism"]}

    # N == 1 (or one explicitly chosen mode): one measurement.  N > 1 with the tile-row headline: the mode WITHOUT a data-path
    # collective -- every rank its own frame -- is measured FIRST and kept as the fallback line; the tile-row exchange (never run by us
    # on more than one RCCL rank) then runs under a watchdog: should it hang or fail, the fallback line is printed with an error field
    # instead of nothing.
    out, wl = None, None
    side_first = world > 1 and ((sharded and not args.no_frame_shard_probe) or (not sharded and not args.no_tile_shard_probe))
    fallback = None
    if side_first and sharded:
        wl_f, dt_f, stage_f, stats_f = measure(False, args.steps, args.warmup, args.prewarm_seconds)
        if rank == 0:
            fallback = line(wl_f, dt_f, stage_f, stats_f, False, args.steps)
        wl_f.free()
        import threading

        def bail():
            if rank == 0:
                o = dict(fallback)
                o["tile_shard"] = {"error": f"the tile-row exchange did not complete within {args.probe_timeout:.0f} s: this line is the frame-sharded measurement"}
                emit(result_fd, o)
            os._exit(0)

        watchdog = threading.Timer(args.probe_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            wl, dt, stage_ms, step_stats = measure(True, args.steps, args.warmup, 0.2)
            if rank == 0:
                out = line(wl, dt, stage_ms, step_stats, True, args.steps)
                out["frame_shard"] = summary(fallback) | {"what": "every rank renders its own frame (the metric's unit is a frame and frames are independent): no data-path collective"}
        except Exception as ex:
            if rank == 0:
                out = dict(fallback)
                out["tile_shard"] = {"error": repr(ex)[:300]}
            wl = None
        watchdog.cancel()
    else:
        wl, dt, stage_ms, step_stats = measure(sharded, args.steps, args.warmup, args.prewarm_seconds)
        if rank == 0:
            out = line(wl, dt, stage_ms, step_stats, sharded, args.steps)
        if side_first:   # (--shard frames: the tile-row mode as the side measurement, under the watchdog)
            import threading

            def bail2():
                if rank == 0:
                    o = dict(out)
                    o["tile_shard"] = {"error": f"timed out after {args.probe_timeout:.0f} s; the headline is unaffected"}
                    emit(result_fd, o)
                os._exit(0)

            watchdog = threading.Timer(args.probe_timeout, bail2)
            watchdog.daemon = True
            watchdog.start()
            try:
                k2 = max(1, min(args.steps, 10))
                wl2, dt2, stage2, stats2 = measure(True, k2, 2, 0.0)
                if rank == 0:
                    out["tile_shard"] = summary(line(wl2, dt2, stage2, stats2, True, k2))
                wl2.free()
            except Exception as ex:
                if rank == 0:
                    out["tile_shard"] = {"error": repr(ex)[:300]}
            watchdog.cancel()
    side = None
    if wl is not None:
        scene, sdict, es, gy = wl.scene, wl.sdict, wl.es, wl.gy
        fwd_only = wl.fwd_only

    if rank == 0:
        if sharded and world > 1 and out.get("scaling") == "strong":
            out["tile_shard_exchange"] = ("forward: every peer sends its strip as three channel segments straight into rank 0's frame (grouped RCCL send/recv); "
                                          "backward: all-reduce of 36 B per Gaussian between the two halves; the per-Gaussian stages run replicated "
                                          "(DESIGN.md section 7 gives the Amdahl ceiling per N)")
        legs = {"headline": round(time.perf_counter() - t_start, 1)}
        if world == 1:
            t_leg = time.perf_counter()
            out["hbm_measured"] = hbm_ceilings(dev)
            legs["hbm_measured"] = round(time.perf_counter() - t_leg, 1)
        if world == 1 and not args.no_cpu_baseline:
            t_leg = time.perf_counter()
            checker = checker_legs(scene, sdict, gy, fwd_only, args.cpu_rows, wl.state, wl.leaves,
                                   raster_factory=lambda e: __import__("diff_gaussian_rasterization").GaussianRasterizer(wl.rs._replace(settings=e)),
                                   es=es, tensors=wl.tensors(), w_img=wl.w_img, wl=wl, dev=dev, detail=args.detail)
            out.update(checker)
            legs["cpu_baseline_and_parity"] = round(time.perf_counter() - t_leg, 1)
        if world == 1 and not args.no_other_workloads and args.workload == "C2" and args.variant == "full" \
                and not args.fwd_only and not args.train_forward_only:
            if not wl.loss and args.detail:
                # the same workload with the step definition of rounds 1-2 (a torch loss inside the step), for continuity
                wl.loss = True
                k3 = max(1, min(args.steps, 20))
                dt3, stage3, stats3, _ = timed_region(wl, k3, 3, 0.0, barrier)
                out["with_torch_loss"] = {"value": round(k3 / dt3, 3), "unit": "frames/s", "ms_per_step": round(1000.0 * dt3 / k3, 4), "steps": k3,
                                          "step_ms": stats3, "what": "the same workload with (image * w).sum().backward() as the step: three torch "
                                          "kernels (mul, sum, mul) inside the timed region, as in rounds 1-2"}
            wl.free()
            _C.clear_scratch_pool(dev)
            torch.cuda.empty_cache()
            t_leg = time.perf_counter()
            out["other_workloads"] = other_workloads(dev, detail=args.detail, scale=args.scale)
            legs["other_workloads"] = round(time.perf_counter() - t_leg, 1)
        legs["total"] = round(time.perf_counter() - t_start, 1)
        out["leg_seconds"] = legs
        emit(result_fd, out)
    os.close(result_fd)
    if dist is not None:
        # the line is out; a rank whose peers have left through a watchdog (or died) must not sit in this barrier until the
        # launcher's own limit expires
        import threading
        last = threading.Timer(60.0, lambda: os._exit(0))
        last.daemon = True
        last.start()
        dist.barrier()
        dist.destroy_process_group()
        last.cancel()


def _psnr(a, b):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 200.0 if mse == 0 else round(10.0 * math.log10(1.0 / mse), 2)


def _grad_err(prod, other):
    """Per gradient tensor: |difference| per Gaussian relative to the tensor's largest entry.  Returns the worst value, the
    number of Gaussians above 1e-4 in any tensor (a blend on the 1/255 threshold flips with the expf implementation and puts
    the weight of one pixel into the gradients of the Gaussians blended there) and the worst value among all the others."""
    per_g = None
    for k, a in prod.items():
        b = other.get(k)
        if a is None or b is None or np.size(b) == 0:
            continue
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        if k == "dL_dmeans2D":
            a, b = a[:, :2], b[:, :2]
        e = np.abs(a - b).reshape(a.shape[0], -1).max(axis=1) / max(float(np.max(np.abs(b))), 1e-30)
        per_g = e if per_g is None else np.maximum(per_g, e)
    if per_g is None:
        return {}
    over = per_g > 1e-4
    return {"grad_rel_max": float(per_g.max()), "gaussians_over_1e-4": int(over.sum()),
            "grad_rel_max_of_the_others": float(per_g[~over].max()) if (~over).any() else 0.0}


def _img_err(a, b):
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    return {"psnr_db": _psnr(a, b), "max_abs": float(d.max()), "pixels_moved_gt_2e-6": int((d > 2e-6).any(axis=0).sum()), "pixels": int(d[0].size)}


def checker_legs(scene, sdict, gy, fwd_only, rows, state, leaves, raster_factory, es, tensors, w_img, wl=None, dev=None, detail=False):
    """The checker / baseline leg (rank 0, N = 1): everything here is test infrastructure from oracle/, used as the thing
    compared AGAINST and as reported non-target baselines, never as the thing measured.
      cpu_baseline            the CPU oracle timed on a bounded window of the same frame
      parity                  product vs that oracle window (image PSNR / max-abs, gradient max error), and -- when
                              oracle/_ref is present -- product vs THE REFERENCE'S OWN KERNELS on the whole timed frame
      reference_on_this_gpu   the reference's own kernels (hipify-perl + hipcc build) timed on this GPU, same frame"""
    means3D, means2D, opac, shs, scales, rots = tensors
    out = {}
    base, y0, nrows, ofr, ograds = cpu_baseline(scene, sdict, gy, fwd_only, rows)
    out["cpu_baseline"] = base
    names = ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh")
    grab = lambda: {n: (None if x.grad is None else x.grad.detach().cpu().numpy()) for n, x in zip(names, (means3D, means2D, opac, scales, rots, shs))}
    # product on the oracle's window (same private tile-row key the tile-row sharding uses), one untimed step
    esw = type(es).from_dict(es.to_dict()) if hasattr(type(es), "from_dict") else es
    base_to_dict = esw.to_dict
    esw.to_dict = lambda: {**base_to_dict(), "_tile_rows": (y0, y0 + nrows)}
    rw = raster_factory(esw)
    for x in leaves:
        if x is not None:
            x.grad = None
    cw, _ = rw(means3D, means2D, opac, shs=shs, scales=scales, rotations=rots)
    sl = slice(16 * y0, min(16 * (y0 + nrows), scene.H))
    img_p = cw.detach().cpu().numpy()[:, sl]
    img_o = ofr.color[:, sl]
    par = {"against": "CPU oracle; the oracle is held bit-for-bit in all integer / state results to the reference's own sources built for gfx950 by "
                      "hipify-perl + a 40-line adapter header + hipcc -ffp-contract=off (tests/test_reference_golden.py) -- NOT an nvcc build: by the "
                      "tier rules that pin counts as 'parity unpinned / partial'.  vs_reference_*: the same build of the reference run on the whole "
                      "timed frame on this GPU; the default library evaluates depthAlongRay uncontracted like that build (keys / lists bit-equal), "
                      "fma_depth_build is the second shipped library",
           "window": f"tile rows {y0}..{y0 + nrows - 1} of {gy} of the timed frame",
           **_img_err(img_p, img_o)}
    gw = None
    if not fwd_only:
        (cw * w_img).sum().backward()
        gw = grab()
        par.update(_grad_err(gw, ograds))
    ofr_final_T = ofr.array("final_T").reshape(-1) if (par["pixels_moved_gt_2e-6"] or par.get("gaussians_over_1e-4")) else None
    ofr.free()
    out["parity"] = par
    # the reference itself on this GPU: whole frame, both builds when present; the default library and the second shipped one
    try:
        from oracle import reference as ref
        from diff_gaussian_rasterization import _C
        prod = product_frame(wl)
        if ofr_final_T is not None:   # the oracle window's residual, explained from the whole frame's shared state
            try:
                full = np.array(prod["color"], copy=True)
                full[:, sl] = img_p            # (the window run's pixels, compared with the oracle's)
                ocol = np.array(prod["color"], copy=True)
                ocol[:, sl] = img_o
                par["residual"] = explain_differences({**prod, "color": full, "grads": gw}, ocol, ofr_final_T if ofr_final_T.size == scene.W * scene.H else None,
                                                      ograds, y0=sl.start, y1=sl.stop)
            except Exception as ex:
                par["residual"] = {"error": repr(ex)[:200]}
            try:
                res = par.get("residual") or {}
                if res.get("by") and res.get("explained", "0/1").split("/")[0] == res.get("explained", "0/1").split("/")[1]:
                    res["nudged_oracle"] = nudged_oracle_check(scene, sdict, fwd_only, y0, nrows, sl, img_p, gw, {"by": res["by"], "decisions": [tuple(d) for d in res.get("decisions", [])], "T_pixels": [tuple(d) for d in res.get("T_pixels", [])]})
            except Exception as ex:
                par["residual"]["nudged_oracle"] = {"error": repr(ex)[:200]}
        for variant, key in (("ieee", "vs_reference_ieee_build"), ("fast", "vs_reference_default_build")):
            if not ref.available(variant) or (variant == "fast" and not detail):
                continue
            rfr = ReferenceFrame(scene, sdict, fwd_only, variant=variant)
            par[key] = rfr.compare(prod)
            if variant == "ieee":
                fma = os.path.join(os.path.dirname(_C.library_path()), FMA_LIB_NAME)
                if detail and os.path.exists(fma) and os.path.basename(_C.library_path()) != FMA_LIB_NAME:
                    # the second shipped build (depth keys as fma chains, the default of rounds 1-3): same frame, same harness
                    _C.use_library(fma)
                    try:
                        rec = {"library": FMA_LIB_NAME, "what": "the second shipped build: depthAlongRay as fused multiply-add chains (the default of rounds 1-3); "
                               "select with STP_RASTER_LIB=fma"}
                        rec["vs_reference_ieee_build"] = rfr.compare(product_frame(wl))
                        k2 = 20
                        dt2, stage2, stats2, _ = timed_region(wl, k2, 5, 0.0, lambda: torch.cuda.synchronize(dev))
                        rec.update({"value": round(k2 / dt2, 3), "unit": "frames/s", "ms_per_step": round(1000.0 * dt2 / k2, 4), "steps": k2,
                                    "step_ms": {k: stats2[k] for k in ("median", "min", "max")} if stats2 else {},
                                    "stage_ms": {k: round(v, 4) for k, v in stage2.items()}})
                        out["fma_depth_build"] = rec
                    finally:
                        _C.use_library(None)
            if variant == "fast":
                f_ms, b_ms = rfr.rf.time_steps(None if fwd_only else scene.dL_dout, warmup=1, steps=3)
                out["reference_on_this_gpu"] = {
                    "value": round(1000.0 / (f_ms + b_ms), 3), "unit": "frames/s", "fwd_ms": round(f_ms, 2), "bwd_ms": round(b_ms, 2),
                    "kind": "the reference's own kernels, hipify-perl + hipcc defaults for gfx950 (oracle/_ref; CUDA-tuned code on wave64, "
                            "NOT a statement about NVIDIA hardware; nothing is published, so vs_baseline stays null)", "steps": 3}
            rfr.free()
    except Exception as ex:  # the headline stands on its own
        par["vs_reference_error"] = repr(ex)[:300]
    return out


def nudged_oracle_check(scene, sdict, fwd_only, y0, nrows, sl, img_p, gw, ex, budget_s=40.0):
    """The explained residual, closed: the oracle re-run on the same window with exactly the per-pixel alpha tests the explanation names taken the other way
    (oracle.forced_alpha_flips) -- or, for the other kinds of decision, with that kind's threshold moved inside the explanation's own band (oracle.blend_nudge) --
    and the product compared with THAT: a complete explanation leaves no pixel above 2e-6 and no gradient above 1e-4.  Checker-side only."""
    from oracle import oracle as orc
    by = ex["by"]
    tries = []
    if ex.get("decisions") or ex.get("T_pixels"):
        tries.append(({"forced_alpha_flips": [list(d) for d in ex.get("decisions", [])[:16]], "forced_T_flips": [list(d) for d in ex.get("T_pixels", [])[:16]]},
                      lambda: orc.forced_alpha_flips(ex.get("decisions", []), scene.W, T_pixels=ex.get("T_pixels", []))))
    for frac in (0.1, 0.25, 0.5, 1.0):
        for sign in (1.0, -1.0):
            if by.get("alpha_threshold") and not ex.get("decisions"):
                v = sign * frac * 6e-7 / 255.0
                tries.append(({"alpha": v}, lambda v=v: orc.blend_nudge(alpha=v)))
            if by.get("subtile_cull"):
                v = sign * frac * 6e-7 / 255.0
                tries.append(({"cull_alpha": v}, lambda v=v: orc.blend_nudge(cull_alpha=v)))
            if by.get("T_threshold") and not ex.get("T_pixels"):
                v = sign * frac * 1e-6 * 1e-4
                tries.append(({"T": v}, lambda v=v: orc.blend_nudge(T=v)))
    t0 = time.perf_counter()
    best = None
    for n, (what, ctx) in enumerate(tries, 1):
        with ctx():
            f2 = orc.forward_scene(scene, sdict, tile_rows=(y0, y0 + nrows))
            g2 = None if (fwd_only or gw is None) else f2.backward(scene.dL_dout)
        rec = {"oracle_run_with": what, "runs": n, **_img_err(img_p, f2.color[:, sl])}
        if g2 is not None:
            rec.update(_grad_err(gw, g2))
        f2.free()
        closed = rec["pixels_moved_gt_2e-6"] == 0 and not rec.get("gaussians_over_1e-4")
        if best is None or closed or (rec["pixels_moved_gt_2e-6"], rec.get("gaussians_over_1e-4", 0)) < (best["pixels_moved_gt_2e-6"], best.get("gaussians_over_1e-4", 0)):
            best = rec
        if closed or time.perf_counter() - t0 > budget_s:
            break
    if best is None:
        return {"note": "no decision of a kind the oracle can take the other way"}
    best["closes_the_residual"] = bool(best["pixels_moved_gt_2e-6"] == 0 and not best.get("gaussians_over_1e-4"))
    best["what"] = "product vs the oracle re-run on the same window with the explained decision(s) taken the other way and nothing else changed"
    return best


def cpu_baseline(scene, sdict, gy, fwd_only, rows):
    """Oracle (our CPU restatement, OpenMP over all host cores) on a bounded sample of the same frame:
    a centred window of tile rows, sized by a 2-row calibration run so that the sample takes ~15 s, the time
    extrapolated linearly to the full frame."""
    from oracle import oracle as orc
    cores = orc.num_threads()

    keep = {}

    def run(r):
        y0 = max(0, (gy - r) // 2)
        t0 = time.perf_counter()
        f = orc.forward_scene(scene, sdict, tile_rows=(y0, y0 + r))
        g = None if fwd_only else f.backward(scene.dL_dout)
        dt = time.perf_counter() - t0
        if keep.get("f") is not None:
            keep["f"].free()
        keep["f"], keep["g"] = f, g     # the last run's outputs double as the parity target (checker_legs)
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
    return ({"value": round(1.0 / est_frame_s, 5), "unit": "frames/s", "cores": cores, "kind": "port",
             "sample": f"tile rows {y0}..{y0 + rows - 1} of {gy} of the same frame ({'fwd' if fwd_only else 'fwd+bwd'}), "
                       f"{reps} repetition(s), {dt:.1f} s measured on {cores} threads, extrapolated x{gy / rows:.2f}"},
            y0, rows, keep["f"], keep["g"])


if __name__ == "__main__":
    main()
