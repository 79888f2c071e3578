#!/usr/bin/env python3
"""Vector-issue utilisation per view from a scripts/pmc_per_view.py table (passes sq1 or sq2, and tcc for GRBM_GUI_ACTIVE):
    busy = SQ_INSTS_VALU (or SQ_ACTIVE_INST_VALU, quad-cycles) x 4 cycles  /  (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
i.e. the share of the SIMDs' issue cycles that carried a vector ALU instruction, at one instruction per 4 cycles (plain fp32 VOP2 issue
faster, ~2.4 cycles: profiles/r01_ubench_valu_ops.txt; the figure is an upper bound of the busy share by that much).  GRBM_GUI_ACTIVE is
the sum over the 8 XCDs (MI355X_MICROARCH.md); clock = GRBM_GUI_ACTIVE / 8 / kernel time.   usage: valu_busy.py <per-view table> [ms per view, comma list]"""
import sys

rows = {}
for line in open(sys.argv[1]):
    f = line.split()
    if len(f) > 3:
        try:
            rows[f[1]] = [float(x) for x in f[2:]]
        except ValueError:
            pass
valu = rows.get("SQ_ACTIVE_INST_VALU") or rows["SQ_INSTS_VALU"]
gui = rows["GRBM_GUI_ACTIVE"]
n = min(len(valu), len(gui))
busy = [valu[i] * 4.0 / (1024.0 * gui[i] / 8.0) for i in range(n)]
print("view            " + " ".join(f"{i:>7d}" for i in range(n)))
print("VALU busy       " + " ".join(f"{b:7.2f}" for b in busy))
if "SQ_INSTS_SALU" in rows:
    print("SALU per VALU   " + " ".join(f"{rows['SQ_INSTS_SALU'][i] / valu[i]:7.2f}" for i in range(n)))
if "SQ_INSTS_VMEM_RD" in rows:
    print("VALU per gather " + " ".join(f"{valu[i] / rows['SQ_INSTS_VMEM_RD'][i]:7.1f}" for i in range(n)))
if len(sys.argv) > 2:
    ms = [float(x) for x in sys.argv[2].split(",")]
    print("clock GHz       " + " ".join(f"{gui[i] / 8.0 / (ms[i] * 1e-3) / 1e9:7.2f}" for i in range(min(n, len(ms)))))
