#!/usr/bin/env python3
"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks: registers, scratch (spills), occupancy per kernel.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -c --cuda-device-only -Rpass-analysis=kernel-resource-usage parrot_hip.hip -o /dev/null 2> res.txt
    python tools/kernel_resources.py res.txt [substring]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
names, rows = [], []
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    names.append(name)
    rows.append((g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for n, (v, a, sc, occ, lds) in zip(dem, rows):
    if flt in n:
        n = n.replace("parrot::", "").replace("void ", "")
        print(f"{n[:100]:100s} V{v:3d} A{a:3d} scratch {sc:4d} occ {occ}")
