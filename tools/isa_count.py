#!/usr/bin/env python3
"""Static instruction counts of device kernels (no GPU needed): compiles dskgpu.hip to gfx950 assembly and prints, per kernel whose
mangled name contains <pattern>, the VGPR count / occupancy and the instructions per basic block by class.
   python tools/isa_count.py <pattern> [min block size]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1]
minb = int(sys.argv[2]) if len(sys.argv) > 2 else 12
out = "/tmp/isa_dskgpu.s"
cs = os.path.join(root, "dsk_amd", "csrc")
src = os.path.join(cs, "dskgpu.hip")
if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(os.path.join(cs, f)) for f in os.listdir(cs) if not f.endswith(".so")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-ffp-contract=off", "-w",
                           "-S", "--cuda-device-only", "-o", out, src])
lines = open(out).read().split("\n")
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\w+):", lines[i])
    if m and pat in m.group(1):
        name = m.group(1); j = i + 1
        blocks = []; cur = {"name": "entry", "v": 0, "s": 0, "ds": 0, "vm": 0}
        blocks.append(cur)
        while not lines[j].startswith(".Lfunc_end"):
            l = lines[j]; mb = re.match(r"^(\.LBB\d+_\d+):", l)
            if mb:
                cur = {"name": mb.group(1), "v": 0, "s": 0, "ds": 0, "vm": 0}; blocks.append(cur)
            else:
                t = l.strip()
                if t and not t.startswith(";") and not t.startswith("."):
                    op = t.split()[0]
                    key = "v" if op.startswith("v_") else "s" if op.startswith("s_") else "ds" if op.startswith("ds_") else "vm" if op.split("_")[0] in ("global", "buffer", "flat", "scratch") else None
                    if key: cur[key] += 1
            j += 1
        meta = {}
        for l in lines[j:j + 400]:
            mm = re.match(r"^; (NumVgprs|NumSgprs|Occupancy|ScratchSize|LDSByteSize)\S*: (\d+)", l.strip())
            if mm and mm.group(1) not in meta: meta[mm.group(1)] = int(mm.group(2))
            if l.startswith("_Z"): break
        tot = {k: sum(b[k] for b in blocks) for k in ("v", "s", "ds", "vm")}
        print(name[:100]); print("   ", meta, "static totals", tot)
        for b in blocks:
            if b["v"] + b["s"] + b["ds"] + b["vm"] >= minb: print("     ", b)
        i = j
    i += 1
