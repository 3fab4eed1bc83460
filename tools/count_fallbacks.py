import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests')); import conftest
import numpy as np, torch
from helpers import settings_dict, FULL_STP, GpuRun
from diff_gaussian_rasterization import scenes, _C
sc = scenes.config('C2')
for name, sd in (('min', settings_dict(3)), ('full', settings_dict(**FULL_STP))):
    g = GpuRun(sc, sd, backward=False)
    # zero the counters (n_contrib is unused by the hierarchical mode), run backward, read them
    nc = _C.image_array(g.img, sc.W, sc.H, 'n_contrib')
    with torch.no_grad():
        torch.cuda.memset if False else None; base=nc.data_ptr(); import ctypes as _c; hip=_c.CDLL('libamdhip64.so'); hip.hipMemset(_c.c_void_p(base), 0, 32)
    import ctypes
    torch.cuda.synchronize()
    w = torch.tensor(sc.dL_dout, device='cuda:0'); (g.color_t * w).sum().backward(); torch.cuda.synchronize()
    c = nc[:4].cpu().numpy().astype(np.int64)
    print(name, 'window adds', c[0], 'fallbacks', c[1], f'({100*c[1]/max(c[0]+c[1],1):.2f}%)', 'out-of-ring', c[2], 'slot recycled', c[3])
