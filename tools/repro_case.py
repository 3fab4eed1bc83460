#!/usr/bin/env python3
"""Diagnose ONE case of tools/fuzz_parity.py on the GPU box: paste the two dicts of its FAIL line.
    python tools/repro_case.py "{'P': 300, ...scene...}" "{'mode': 2, ...settings...}"
Prints the image difference, where the per-pixel blend counts differ, every gradient tensor's relative error with its worst entry,
and what oracle/explain.py says about the pixels that moved by more than 2e-6 / 1e-7 / 1e-9."""
import ast, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "stopthepop-rasterization_amd"), ROOT):
    sys.path.insert(0, p)
import conftest  # noqa: F401,E402
import numpy as np  # noqa: E402
from helpers import GpuRun, oracle_run, settings_dict  # noqa: E402
from test_gpu_parity import GRAD_KEYS, _rel  # noqa: E402
from diff_gaussian_rasterization import scenes  # noqa: E402
from oracle import explain  # noqa: E402

sc, sdk = ast.literal_eval(sys.argv[1]), ast.literal_eval(sys.argv[2])
scene, sd = scenes.make_scene(**sc), settings_dict(**sdk)
g = GpuRun(scene, sd, backward=True)
f, og = oracle_run(scene, sd, backward=True)
diff = np.abs(g.color.astype(np.float64) - f.color.astype(np.float64))
print("num_rendered", g.num_rendered, f.num_rendered, "| image: max diff %.3e, pixels > 1e-7: %d" % (diff.max(), int((diff > 1e-7).any(axis=0).sum())))
nc_g, nc_f = g.image_array("n_contrib").reshape(-1), f.array("n_contrib").reshape(-1)
bad = np.nonzero(nc_g != nc_f)[0]
print("n_contrib differs at %d pixels" % bad.size, [(int(i), int(nc_g[i]), int(nc_f[i])) for i in bad[:8]])
for k in GRAD_KEYS:
    if g.grads.get(k) is None or og.get(k) is None or og[k].size == 0:
        continue
    a, b = g.grads[k], og[k]
    if k == "dL_dmeans2D":
        a, b = a[:, :2], b[:, :2]
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    i = np.unravel_index(np.argmax(d), d.shape)
    print("%-14s rel %.3e  worst %s: gpu %.6e oracle %.6e  (max |oracle| %.3e)" % (k, _rel(a, b), tuple(int(x) for x in i), a[i], b[i], np.abs(b).max()))
for thr in (2e-6, 1e-7, 1e-9):
    m = (diff > thr).any(axis=0)
    if m.any():
        ex = explain.explain_moved_pixels(m, W=scene.W, H=scene.H, ranges=f.array("ranges").reshape(-1), point_list=f.array("point_list"),
                                          conic_opacity=f.array("conic_opacity").reshape(-1), means2D=f.array("means2D").reshape(-1),
                                          final_T_a=g.image_array("final_T").reshape(-1), final_T_b=f.array("final_T").reshape(-1),
                                          cull_4x4=bool(sd["culling_settings"]["hierarchical_4x4_culling"]) and sd["sort_settings"]["sort_mode"] == 3)
        print("moved by > %g: %d pixels, explained %d, unexplained %d, by %s" % (thr, int(m.sum()), ex["explained"], len(ex["unexplained"]), ex["by"]))
