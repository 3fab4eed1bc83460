#!/usr/bin/env python3
"""VGPRs / spills / LDS / scratch of the kernels in the product build's objects whose (mangled) name matches a regex -- from the gfx950 code object
embedded in each host object (no GPU needed).   usage: tools/kernel_resources.py [regex] [object ...]"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rx = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
objs = sys.argv[2:] or sorted(glob.glob(os.path.join(ROOT, "stopthepop-rasterization_amd", "csrc", "build", "*.o")))
for ob in objs:
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", ob, "/tmp/_fat.bin"], check=True)
    d = open("/tmp/_fat.bin", "rb").read()
    for k, m in enumerate(re.finditer(b"\x7fELF", d)):
        open("/tmp/_co.elf", "wb").write(d[m.start():])
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", "/tmp/_co.elf"], capture_output=True, text=True).stdout
        cur = {}
        for line in out.splitlines():
            mm = re.match(r"\s+\.(name|vgpr_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size|sgpr_count):\s+(.*)", line)
            if mm:
                cur[mm.group(1)] = mm.group(2)
                if mm.group(1) == "vgpr_spill_count" and rx.search(cur.get("name", "")):
                    print(f"{os.path.basename(ob):24s} {cur.get('name', '')[:84]:84s} vgpr {cur.get('vgpr_count'):>4s} sgpr {cur.get('sgpr_count'):>4s} spill {cur.get('vgpr_spill_count')} "
                          f"lds {cur.get('group_segment_fixed_size')} scratch {cur.get('private_segment_fixed_size')}")
