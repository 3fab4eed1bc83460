"""Generates tests/golden/*.npz: small oracle outputs (image, radii, a few gradients) on seeded scenes.

These are ORACLE-generated regression fixtures (the reference has no fixtures and cannot be executed in this
image -- DESIGN.md "Oracle"); they pin the oracle against accidental change and give the GPU tests a
second, file-based target.  Inputs are not stored: they are a pure function of the seed
(diff_gaussian_rasterization/scenes.py).  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: F401,E402
from helpers import FULL_STP, settings_dict  # noqa: E402
from diff_gaussian_rasterization import scenes  # noqa: E402

CASES = {
    "c1_global": dict(scene=dict(P=1000, W=256, H=256, sigma_min=1.0, sigma_max=12.0, seed=1), settings=settings_dict(0)),
    "dense_hier_full": dict(scene=dict(P=3000, W=96, H=80, sigma_min=2.0, sigma_max=14.0, seed=11, camera="orbit"),
                            settings=settings_dict(**FULL_STP)),
    "dense_kbuffer": dict(scene=dict(P=3000, W=96, H=80, sigma_min=2.0, sigma_max=14.0, seed=12, camera="orbit", use_sh=False),
                          settings=settings_dict(2, per_pixel=16)),
}


def run_case(case):
    from oracle import oracle as orc
    sc = scenes.make_scene(**case["scene"])
    f = orc.forward_scene(sc, case["settings"])
    g = f.backward(sc.dL_dout)
    return dict(num_rendered=f.num_rendered, radii=f.radii, color=f.color, dL_dmeans3D=g["dL_dmeans3D"],
                dL_dopacity=g["dL_dopacity"], dL_dscales=g["dL_dscales"])


if __name__ == "__main__":
    for name, case in CASES.items():
        out = run_case(case)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **{k: (np.float16(v) if False else v) for k, v in out.items()})
        print(name, out["num_rendered"], os.path.getsize(os.path.join(HERE, name + ".npz")))
