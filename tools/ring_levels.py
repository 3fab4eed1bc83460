#!/usr/bin/env python3
"""The k-buffer ring kernel's per-process speed levels: one process, several TRIALS.  A trial frees every scratch buffer
(binding pool + torch's allocator), optionally shifts the next allocations by a pad of a given size, runs recording forwards of C3 and reports the
Render stage mean plus the device addresses of the frame's three scratch buffers.  If the level changes between trials of ONE process, it follows
the buffers' placement; if it only changes between processes, it follows something the process owns (code object placement, queue, clocks).
   usage: python tools/ring_levels.py [--workload C3] [--trials 6] [--steps 12] [--pads 0,4096,...] [--fwd-bwd]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C3")
ap.add_argument("--variant", default="full")
ap.add_argument("--trials", type=int, default=6)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--pads", default="", help="comma list of pad bytes allocated (and kept) in front of each trial's buffers")
ap.add_argument("--fwd-bwd", action="store_true")
ap.add_argument("--eval", action="store_true", help="forward passes that record no blend log (inference)")
ap.add_argument("--tag", default="")
ap.add_argument("--img-offsets", default="", help="comma list, one per trial: STP_SCRATCH_OFFSET_IMAGE of the trial (bytes behind the allocation's start)")
ap.add_argument("--bin-offsets", default="", help="likewise STP_SCRATCH_OFFSET_BINNING")
ap.add_argument("--geom-offsets", default="", help="likewise STP_SCRATCH_OFFSET_GEOM")
ap.add_argument("--slab", type=float, default=0.0, help="GiB allocated in ONE piece and handed to torch's cache before the first forward: the scratch buffers are split off it")
ap.add_argument("--keep-cache", action="store_true", help="trials hand the buffers back to torch's cache (same physical memory) instead of freeing them to the driver")
ap.add_argument("--hold", default="", help="binning | image: keep that buffer of the FIRST trial out of the free/re-allocate cycle (only the other kinds are re-placed)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
from diff_gaussian_rasterization import _C  # noqa: E402
wl = bench.Workload(args.workload, args.variant, dev, fwd_only=args.eval, train_forward_only=not args.fwd_bwd and not args.eval)
if args.slab > 0:
    slab = torch.empty(int(args.slab * 2**30), dtype=torch.uint8, device=dev)
    del slab
offs = {k: [int(x, 0) for x in v.split(",") if x] for k, v in (("IMAGE", args.img_offsets), ("BINNING", args.bin_offsets), ("GEOM", args.geom_offsets))}
n_off = max(len(v) for v in offs.values())
if n_off:
    args.trials = n_off
pads = [int(x) for x in args.pads.split(",") if x] or [0] * args.trials
keep = []
for trial, pad in enumerate(pads[:args.trials] if args.pads else pads):
    for k, v in offs.items():
        if v:
            os.environ["STP_SCRATCH_OFFSET_" + k] = str(v[trial % len(v)])
    wl.state.clear()
    _C.clear_scratch_pool(dev)
    if not args.keep_cache:
        torch.cuda.empty_cache()
    if pad:
        keep.append(torch.empty(pad, dtype=torch.uint8, device=dev))
    for _ in range(4):
        wl.step()
    torch.cuda.synchronize(dev)
    _C.timing_enable(True)
    for _ in range(args.steps):
        wl.step()
    torch.cuda.synchronize(dev)
    st = {k: round(v, 4) for k, v in _C.timing_read(dev).items() if v >= 0}
    hist = _C.timing_history(dev, capacity=args.steps)
    _C.timing_enable(False)
    # addresses of the LAST forward's buffers: one more forward through _C directly (same pool, same sizes: the pool hands the same buffers out)
    empty = torch.Tensor([])
    rs = wl.rs
    d1 = dict(wl.sdict); d1["_record_blend_log"] = not args.eval
    o1 = _C.rasterize_gaussians(rs.bg, wl.means3D.detach(), empty, wl.opac.detach(), wl.scales.detach(), wl.rots.detach(), 1.0, empty, rs.viewmatrix, rs.projmatrix,
                                rs.inv_viewprojmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, wl.shs.detach(), rs.sh_degree, rs.campos, False, d1, False, False)
    ptrs = {n: (hex(o1[i].data_ptr()), o1[i].numel()) for n, i in (("geom", 3), ("binning", 4), ("image", 5))}
    o1_R = int(o1[0])
    depth = _C.blend_log_depth(o1[5])
    cap = int(_C._load().stp_binning_layout_count(o1[4].data_ptr(), int(o1[0])))
    # a plain streaming read over the trial's own buffers: is the MEMORY slow (placement) or only our access pattern on it?
    probe = {}
    for n, i in (("binning", 4), ("image", 5)):
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); _C.hbm_probe("read", None, o1[i], blocks=8192, nontemporal=True); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        probe[n + "_read_GBps"] = round(o1[i].numel() / best / 1e6, 0)
    _C.release_scratch(o1[5]); _C.release_scratch(o1[4])
    del o1
    r = [h["Render"] for h in hist]
    print(json.dumps({"tag": args.tag, "pid": os.getpid(), "trial": trial, "pad": pad, "Render": st.get("Render"), "Render_min": round(min(r), 4), "Render_max": round(max(r), 4),
                      "Sort": st.get("Sort"), "BwdRender": st.get("BwdRender"), "log_depth": depth, "bin_cap": cap, "R": int(o1_R), **probe, "skew": os.environ.get("STP_CARVE_SKEW"), "off": {k: os.environ.get("STP_SCRATCH_OFFSET_" + k) for k in ("GEOM", "BINNING", "IMAGE")}, "ptrs": ptrs}), flush=True)
