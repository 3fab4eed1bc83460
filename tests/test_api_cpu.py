"""CPU tests (-m "not gpu") of the drop-in boundary: the Python surface mirrors the reference's, the C-ABI
library loads and exports every symbol include/stp_raster.h declares (no compute call is made here: the
product has no CPU path), the tile-row sharding host logic is correct (single process + world_size-2 gloo)."""
import ctypes
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import conftest
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _C, tile_shard

ROOT = conftest.ROOT

# ExtendedSettings().to_dict() of the reference package, captured by importing its pure-Python half with
# stubbed `_C`/`dacite`; committed here as data.
REFERENCE_DEFAULT_DICT = {
    'sort_settings': {'queue_sizes': {'tile_4x4': 64, 'tile_2x2': 8, 'per_pixel': 4}, 'sort_mode': 0, 'sort_order': 0},
    'culling_settings': {'rect_bounding': False, 'tight_opacity_bounding': False, 'tile_based_culling': False,
                         'hierarchical_4x4_culling': False},
    'load_balancing': False, 'proper_ewa_scaling': False}


def test_public_names_and_defaults():
    for name in ("rasterize_gaussians", "GaussianRasterizer", "GaussianRasterizationSettings", "ExtendedSettings",
                 "SortSettings", "SortQueueSizes", "CullingSettings", "SortMode", "GlobalSortOrder", "_RasterizeGaussians"):
        assert hasattr(dgr, name), name
    assert dgr.ExtendedSettings().to_dict() == REFERENCE_DEFAULT_DICT
    assert [m.name for m in dgr.SortMode] == ["GLOBAL", "PPX_FULL", "PPX_KBUFFER", "HIER"]
    assert [int(m) for m in dgr.SortMode] == [0, 1, 2, 3]
    assert [m.name for m in dgr.GlobalSortOrder] == ["Z_DEPTH", "DISTANCE", "PTD_CENTER", "PTD_MAX"]
    assert str(dgr.SortMode.HIER) == "HIER"
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "inv_viewprojmatrix", "sh_degree", "campos", "prefiltered", "settings", "render_depth", "debug")


def test_settings_set_value_falls_through_and_round_trips():
    e = dgr.ExtendedSettings()
    e.set_value("per_pixel", 16)                 # -> sort_settings.queue_sizes
    e.set_value("sort_mode", dgr.SortMode.HIER)  # -> sort_settings
    e.set_value("tile_based_culling", True)      # -> culling_settings
    e.set_value("proper_ewa_scaling", True)      # own field
    e.set_value("no_such_key", 1)                # silently ignored, as in the reference
    d = e.to_dict()
    assert d["sort_settings"]["queue_sizes"]["per_pixel"] == 16 and d["sort_settings"]["sort_mode"] == 3
    assert d["culling_settings"]["tile_based_culling"] is True and d["proper_ewa_scaling"] is True
    assert dgr.ExtendedSettings.from_dict(d).to_dict() == d
    assert dgr.ExtendedSettings().to_dict() == REFERENCE_DEFAULT_DICT  # defaults are not shared instances
    import json
    assert json.loads(e.to_json()) == d


def test_forward_argument_checks_raise_the_reference_messages():
    rs = dgr.GaussianRasterizationSettings(16, 16, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), torch.eye(4), 0,
                                           torch.zeros(3), False, dgr.ExtendedSettings(), False, False)
    r = dgr.GaussianRasterizer(rs)
    m, o = torch.zeros(2, 3), torch.ones(2, 1)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, o, scales=m, rotations=torch.zeros(2, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, colors_precomp=m)
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        r(m, m, o, colors_precomp=m, scales=m, rotations=torch.zeros(2, 4), cov3D_precomp=torch.zeros(2, 6))


def test_product_refuses_to_run_without_a_gpu():
    """No silent CPU fallback: CPU tensors are rejected before any native call."""
    rs = dgr.GaussianRasterizationSettings(16, 16, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), torch.eye(4), 0,
                                           torch.zeros(3), False, dgr.ExtendedSettings(), False, False)
    m, o = torch.zeros(2, 3), torch.ones(2, 1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        dgr.GaussianRasterizer(rs)(m, m, o, colors_precomp=m, scales=m, rotations=torch.zeros(2, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        dgr.GaussianRasterizer(rs).markVisible(m)
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        _C.rasterize_gaussians(*([torch.zeros(3), torch.zeros(4)] + [None] * 20))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "stopthepop-rasterization_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath:
            continue
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".inc", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle|stp_oracle|liboracle", text, re.M), os.path.join(dirpath, fn)


def test_c_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "stp_raster.h")).read()
    declared = set(re.findall(r"\b(stp_[a-z_]+)\s*\(", header)) - {"stp_alloc_fn"}
    assert {"stp_forward", "stp_backward", "stp_backward_phases", "stp_mark_visible", "stp_last_error"} <= declared
    lib = ctypes.CDLL(_C.library_path())
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in include/stp_raster.h but not exported"
    nm = subprocess.run(["nm", "-D", "--defined-only", _C.library_path()], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (stp_[a-z_]+)$", nm, re.M))
    assert declared <= exported


def test_native_host_binding_loads_and_binds_the_c_abi():
    """The native torch binding (csrc/host/stp_torch_binding.cpp) imports on a box without a GPU, binds the library file _C resolved
    (run-time dlopen: STP_RASTER_LIB / _C.use_library keep working) and exports the reference's three functions + the pool surface."""
    h = _C._native()
    for name in ("rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible", "scratch_generation", "check_scratch", "release_scratch",
                 "clear_scratch_pool", "set_scratch_pool_limit", "pooled_sizes", "load_library"):
        assert callable(getattr(h, name)), name
    assert h.load_library(_C.library_path()) == 7
    assert _C.clear_scratch_pool() == 0 and list(h.pooled_sizes(0)) == []
    with pytest.raises(RuntimeError, match="cannot load"):
        h.load_library("/nonexistent/libstp_raster.so")
    assert h.load_library(_C.library_path()) == 7   # (a failed load leaves the bound library in place)


def test_c_abi_size_and_layout_queries_without_gpu():
    L = _C._load()
    assert L.stp_abi_version() == 7
    s = _C.settings_from_dict(dgr.ExtendedSettings().to_dict())
    small, big = L.stp_geometry_buffer_size(1000, ctypes.byref(s)), L.stp_geometry_buffer_size(2000, ctypes.byref(s))
    assert 0 < small < big
    s.sort_mode = 3  # sorted modes add the Sigma^-1 pack (48 B per Gaussian)
    assert L.stp_geometry_buffer_size(1000, ctypes.byref(s)) >= small + 48 * 1000
    assert L.stp_image_buffer_size(64, 64) >= 64 * 64 * 8
    assert L.stp_binning_buffer_size(0) > 0 and L.stp_binning_buffer_size(10000) >= 10000 * 24
    off, cnt = ctypes.c_size_t(), ctypes.c_size_t()
    assert L.stp_geometry_layout(1000, ctypes.byref(s), b"cov3D_inv", ctypes.byref(off), ctypes.byref(cnt)) == 0
    assert cnt.value == 12000 and off.value % 256 == 0
    assert L.stp_geometry_layout(1000, ctypes.byref(s), b"nonsense", ctypes.byref(off), ctypes.byref(cnt)) < 0
    assert b"nonsense" in L.stp_last_error()
    # blend log of a frame nothing is known about: 192 records + one spare block of eight, 2 bytes each, per pixel of the tile grid (400 B); a tile-row
    # window holds its rows' share; the image-side layout of a window starts at the window's first pixel / tile
    grid_px = ((1920 + 15) // 16) * 16 * ((1080 + 15) // 16) * 16
    assert _C.blend_log_bytes(1920, 1080) == 400 * grid_px
    assert _C.blend_log_bytes(1920, 1080, (0, 17)) * 4 == _C.blend_log_bytes(1920, 1080)
    L.stp_image_layout_rows.argtypes = [ctypes.c_int] * 4 + [ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    assert L.stp_image_layout_rows(1920, 1080, 17, 34, b"ranges", ctypes.byref(off), ctypes.byref(cnt)) == 0 and cnt.value == 2 * 120 * 17
    assert L.stp_image_layout_rows(1920, 1080, 0, 0, b"final_T", ctypes.byref(off), ctypes.byref(cnt)) == 0 and cnt.value == 1920 * 1080
    # the viewer's timings text (reference rasterizer_impl.cu:391-399): header, four forward stages, their total
    text = _C.timing_text()
    assert text.startswith("Timings: \n - Preprocess: ") and " - Render: " in text and text.rstrip().endswith("ms")
    assert [ln.split(":")[0] for ln in text.splitlines()[1:6]] == [" - Preprocess", " - Duplicate", " - Sort", " - Render", " - Total"]


def test_settings_dict_keys_are_mandatory():
    d = dgr.ExtendedSettings().to_dict()
    del d["culling_settings"]["rect_bounding"]
    with pytest.raises(KeyError):
        _C.settings_from_dict(d)


# ---------------------------------------------------------------- tile-row sharding host logic
def test_row_partition_matches_the_survey_example():
    assert [b - a for a, b in tile_shard.row_partition(135, 8)] == [17] * 7 + [16]   # C4: 2160 px = 135 tile rows
    assert tile_shard.row_partition(68, 8)[-1] == (63, 68)
    assert tile_shard.row_partition(3, 8)[3:] == [(3, 3)] * 5                           # more ranks than rows
    for rows, world in ((68, 1), (68, 2), (68, 4), (68, 8), (7, 3)):
        parts = tile_shard.row_partition(rows, world)
        assert parts[0][0] == 0 and parts[-1][1] == rows
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))


def test_strip_pack_assemble_round_trip():
    H, W = 1063, 40  # height not a multiple of the tile size
    img = torch.arange(3 * H * W, dtype=torch.float32).reshape(3, H, W)
    parts = tile_shard.row_partition(tile_shard.tile_rows(H), 4)
    rows_max = max(b - a for a, b in parts)
    strips = [tile_shard.pack_strip(img, p, rows_max) for p in parts]
    assert all(s.shape == (3, rows_max * 16, W) for s in strips)
    assert torch.equal(tile_shard.assemble(strips, parts, H), img)
    a, b, c, d = torch.rand(5, 3), torch.rand(5, 2, 2), torch.rand(5, 1), torch.rand(5, 3)
    a[:, 2] = 0.0   # the gradient records carry what the reference's tensors really use: mean2D x, y ...
    b[:, 1, 0] = 0  # ... and the three distinct conic terms
    rec = tile_shard.pack_partials(a, b, c, d)
    assert rec.shape == (5, 16) and not rec[:, 9:].any()
    ua, ub, uc, ud = tile_shard.unpack_partials(rec)
    assert torch.equal(ua, a) and torch.equal(ub, b) and torch.equal(uc, c) and torch.equal(ud, d)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_balanced_partition_by_row_cost():
    """Contiguous blocks of (nearly) equal cost that tile [0, n) in order; uniform cost = the equal-height blocks."""
    bp = tile_shard.balanced_partition
    for cost, world in (([1] * 135, 8), ([5, 1, 1, 1, 9, 9, 1, 1, 1, 5], 3), ([0, 0, 7, 7, 0, 0], 3), ([0] * 5, 2), ([100, 1, 1, 1], 4), ([3, 1, 4, 1, 5, 9, 2, 6], 1)):
        parts = bp(cost, world)
        assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == len(cost)
        assert all(parts[i][1] == parts[i + 1][0] and parts[i][0] <= parts[i][1] for i in range(world - 1))
    sizes = [b - a for a, b in bp([1] * 135, 8)]
    assert max(sizes) - min(sizes) <= 1
    heavy = bp([10, 1, 1, 1, 1, 1, 1, 10], 2)
    assert heavy == [(0, 4), (4, 8)]
    parts = bp([8, 8, 1, 1, 1, 1, 1, 1, 1, 1], 2)          # the heavy rows end up alone on one rank
    assert parts[0][1] <= 2


def test_two_rank_gloo_gather_and_gradient_allreduce():
    """world_size 2 over gloo: each rank renders its tile rows with the CPU oracle (test-side compute), the
    product's exchange code gathers the strips / all-reduces the partial gradients; rank 0 checks them against
    the unsharded frame."""
    script = os.path.join(ROOT, "tests", "_gloo_worker.py")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "GLOO_SHARD_OK" in outs[0]


def test_inline_dpp_instructions_have_no_read_after_write_hazard():
    """The hot kernels fold quad broadcasts into arithmetic with inline-asm DPP instructions; the compiler's hazard
    recognizer does not see inside them, so the generated code is scanned: no VALU write of a DPP source register
    in the two issue slots before its use (tools/check_dpp_hazards.py, cross-compiles for gfx950, no GPU needed)."""
    import shutil, subprocess, sys
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazards.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "DPP asm instructions" in r.stdout
