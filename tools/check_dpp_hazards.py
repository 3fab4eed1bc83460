#!/usr/bin/env python3
"""Static check of the inline-asm DPP instructions (quad_mul / quad_fma / quad_sub / partner_fma in stp_device.h):
the DPP source register of each must not be written by a VALU instruction in the two preceding issue slots (the
hazard recognizer does not look inside inline asm).  Usage: tools/check_dpp_hazards.py  (compiles the hot kernels to
assembly with hipcc and scans them; exit code 1 on a finding)."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stopthepop-rasterization_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-I/opt/rocm/include", "-S", "--cuda-device-only"]
JOBS = [("stp_render_replay.hip", []), ("stp_render_hier_inst.hip", ["-DSTP_INST_MID=8", "-DSTP_INST_MODE=0"]),
        ("stp_render_hier_inst.hip", ["-DSTP_INST_MID=8", "-DSTP_INST_MODE=2"]), ("stp_render_hier_inst.hip", ["-DSTP_INST_MID=12", "-DSTP_INST_MODE=2"]),
        ("stp_render_hier_inst.hip", ["-DSTP_INST_MID=20", "-DSTP_INST_MODE=3"])]


def written(ins):
    m = re.match(r"\s*(v_\w+)\s+([^,]+),", ins)
    if not m:
        return set()
    dst = m.group(2).strip()
    mm = re.match(r"v\[(\d+):(\d+)\]", dst)
    if mm:
        return set(range(int(mm.group(1)), int(mm.group(2)) + 1))
    mm = re.match(r"v(\d+)$", dst)
    return {int(mm.group(1))} if mm else set()


def scan(path):
    lines = open(path).read().split("\n")
    total = bad = 0
    for i, l in enumerate(lines):
        m = re.match(r"\s*(v_(?:mul|fmac|sub)_f32_dpp)\s+v(\d+),\s*v(\d+),", l)
        if not m:
            continue
        total += 1
        src = int(m.group(3))
        cnt, j = 0, i - 1
        while j >= 0 and cnt < 2:
            t = lines[j].strip()
            if t and not t.startswith(";") and not t.endswith(":") and not t.startswith("."):
                cnt += 1
                if t.startswith("s_nop"):
                    break
                if src in written(lines[j]):
                    bad += 1
                    print(f"{path}:{j + 1}: {t}  ->  {l.strip()}")
            j -= 1
    return total, bad


def main():
    worst = 0
    with tempfile.TemporaryDirectory() as d:
        for k, (src, defs) in enumerate(JOBS):
            out = os.path.join(d, f"k{k}.s")
            extra = ["-fno-slp-vectorize"] if "hier" in src else []  # (HIERFLAGS of csrc/Makefile)
            subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + defs + ["-o", out, os.path.join(CSRC, src)], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            total, bad = scan(out)
            print(f"{src} {' '.join(defs)}: {total} DPP asm instructions, {bad} possible hazards")
            worst = max(worst, bad)
    sys.exit(1 if worst else 0)


if __name__ == "__main__":
    main()
