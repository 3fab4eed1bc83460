#!/usr/bin/env python3
"""Times THE REFERENCE ITSELF (oracle/_ref/libstp_ref.so: its own kernels, hipify-perl + hipcc defaults, see
oracle/ref_build/build_ref.sh) on the BASELINE configurations on this GPU: forward and backward ms per step on
resident inputs, hipEvents on the stream the reference launches on (the null stream), the reference's own
`num_rendered` read-back and the binding's gradient zero-fill included.  This is the "reference on MI355X"
number quoted beside bench.py's; it is NOT a published baseline (BASELINE.md holds none).

    python tools/ref_time.py --workload C2 --variant full [--steps 10] [--json out.json]

The reference's load-balancing code path hard-wires a 32-lane warp; on wave64 the adapter keeps its 32-lane
meaning (oracle/ref_build/hip_adapter.h), and `--no-lb` times the same settings without it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "stopthepop-rasterization_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from diff_gaussian_rasterization import scenes  # noqa: E402
from helpers import FULL_STP, settings_dict  # noqa: E402
from oracle import reference as ref  # noqa: E402


def settings_for(workload, variant, lb=True):
    if workload == "C1":
        return settings_dict(0)
    if workload == "C3":
        return settings_dict(2, per_pixel=16)
    if variant == "full":
        return settings_dict(**{**FULL_STP, "lb": lb})
    return settings_dict(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--variant", default="full")
    ap.add_argument("--build", default="fast", choices=["fast", "ieee"])
    ap.add_argument("--no-lb", action="store_true")
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    scene = scenes.config(a.workload, scale=a.scale)
    sd = settings_for(a.workload, a.variant, lb=not a.no_lb)
    f = ref.forward_scene(scene, sd, variant=a.build)
    fwd_only = a.fwd_only or a.workload == "C4"
    fwd, bwd = f.time_steps(None if fwd_only else scene.dL_dout, warmup=a.warmup, steps=a.steps)
    total = fwd + bwd
    rec = {"what": "reference (its own kernels via hipify-perl + hipcc) on this GPU", "build": ref.build_info(a.build),
           "workload": f"{a.workload}-{a.variant}" + ("-nolb" if a.no_lb else ""), "P": scene.P, "W": scene.W, "H": scene.H,
           "num_rendered": f.num_rendered, "fwd_ms": round(fwd, 4), "bwd_ms": round(bwd, 4), "ms_per_step": round(total, 4),
           "frames_per_s": round(1000.0 / total, 2), "steps": a.steps, "warmup": a.warmup, "fwd_only": fwd_only}
    print(json.dumps(rec))
    if a.json:
        with open(a.json, "w") as fh:
            json.dump(rec, fh, indent=1)


if __name__ == "__main__":
    main()
