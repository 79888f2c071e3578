#!/usr/bin/env python3
"""SGPR / VGPR / occupancy / spills of every kernel in the last build (csrc/resource_usage.log)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
log = open(os.path.join(ROOT, "volume-rendering_amd", "csrc", "resource_usage.log")).read()
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for b in re.split(r"remark: Function Name: ", log)[1:]:
    name = b.split(" ")[0]
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    if pat and pat not in name:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    occ, lds = g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{name[:90]:90s} S {g('TotalSGPRs'):>3s} V {g(' VGPRs'):>3s} occ {occ} spill {g('VGPRs Spill')} lds {lds}")
