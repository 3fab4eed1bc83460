"""The bench line's contract: ONE stdout line the driver can parse -- under 8 KB, the contract's keys plus `roofline`,
`cpu_baseline`, a short parity summary and one compact object per other workload; everything else goes to bench_detail.json / stderr.

  * CPU: `bench.compact_line` on the committed full record of round 5 (`profiles/r05_bench_full.json`, the 24.7 KB line the driver could not
    parse) and on a record padded far beyond it;
  * GPU (-m gpu): `bench.py --scale 0.05` at N = 1 as the driver runs it, every leg on (CPU baseline, parity, other workloads)."""
import json
import os
import subprocess
import sys

import pytest

import conftest

sys.path.insert(0, conftest.ROOT)
import bench  # noqa: E402

CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"}
ROOFLINE_KEYS = {"bound", "achieved", "peak", "unit", "frac", "traffic"}
CPU_KEYS = {"value", "unit", "cores", "kind", "sample"}


def _check_line(d, text, other_workloads=True):
    assert len(text) < bench.LINE_CAP_BYTES, len(text)
    assert CONTRACT_KEYS <= set(d), CONTRACT_KEYS - set(d)
    assert ROOFLINE_KEYS <= set(d["roofline"]), d["roofline"]
    assert d["roofline"]["bound"] in ("hbm", "valu", "lds") and d["roofline"]["unit"] == "GB/s" and d["roofline"]["peak"] == 8000.0
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-4
    assert CPU_KEYS <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] in ("port", "reference")
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["vs_baseline"] is None and d["dtype"] == "f32"
    assert {"oracle"} <= set(d["parity"]) and "psnr_db" in d["parity"]["oracle"]
    if other_workloads:
        assert set(d["other_workloads"]) == {label for label, *_ in bench.OTHER_WORKLOADS}
        for w in d["other_workloads"].values():
            assert "error" not in w, w
            assert {"value", "ms_per_step", "dominant_kernel", "frac", "traffic"} <= set(w), w

    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, (list, tuple)):
            for v in o:
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(x) for x in strings(d)) <= 160   # (the driver shortens long strings)


def test_compact_line_of_the_round_5_record_fits_the_cap():
    full = json.load(open(os.path.join(conftest.ROOT, "profiles", "r05_bench_full.json")))
    assert len(json.dumps(full)) > 20000          # the record that could not be parsed
    o = bench.compact_line(full)
    text = json.dumps(o)
    _check_line(o, text)
    assert len(text) < 6000
    # the numbers are the record's own
    assert o["value"] == full["value"] and o["roofline"]["frac"] == full["roofline"]["frac"] and o["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    assert o["parity"]["reference_build"]["keys_equal"] is True and o["parity"]["oracle"]["explained"] == "2/2" and o["parity"]["oracle"]["closed"] is True
    assert o["other_workloads"]["C5"]["closed"] is True and o["other_workloads"]["C3"]["list_equal"] is True


def test_emit_sheds_optional_objects_instead_of_losing_the_line(tmp_path, monkeypatch):
    full = json.load(open(os.path.join(conftest.ROOT, "profiles", "r05_bench_full.json")))
    # forty other workloads instead of four: the optional objects go, the contract's stay
    full["other_workloads"] = {f"W{i}": dict(full["other_workloads"]["C5"]) for i in range(40)}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    r, w = os.pipe()
    bench.emit(w, full)
    os.close(w)
    text = os.read(r, 1 << 20).decode()
    os.close(r)
    assert text.endswith("\n") and text.count("\n") == 1 and len(text) < bench.LINE_CAP_BYTES
    d = json.loads(text)
    assert CONTRACT_KEYS <= set(d) and "roofline" in d and "cpu_baseline" in d and "other_workloads" not in d
    assert json.load(open(tmp_path / "bench_detail.json"))["other_workloads"].keys() == full["other_workloads"].keys()


def test_the_committed_round_6_line_is_what_the_driver_can_parse():
    """profiles/r06_bench_full.json is the stdout line of the default run on the round's final kernels: one bounded JSON object with the contract's
    keys, the roofline of both render kernels WITH measured traffic, the CPU baseline, and a traffic figure for every other workload."""
    text = open(os.path.join(conftest.ROOT, "profiles", "r06_bench_full.json")).read().strip()
    assert text.count("\n") == 0
    d = json.loads(text)
    _check_line(d, text)
    assert d["roofline"]["traffic"] and d["roofline"]["second"]["traffic"]
    assert all(w["traffic"] for w in d["other_workloads"].values())
    assert d["other_workloads"]["C3"]["dominant_kernel"].startswith("render_kbuffer_ring_kernel<16")
    assert d["leg_seconds"]["total"] < 60.0


@pytest.mark.gpu
def test_bench_prints_one_bounded_line_at_n1():
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    p = subprocess.run([sys.executable, os.path.join(conftest.ROOT, "bench.py"), "--scale", "0.05", "--steps", "3", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    _check_line(d, lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    assert "[bench detail] {" in p.stderr
