#!/usr/bin/env python3
"""Copies the summaries of a scripts/profile_bench.sh run (gpurun_out/profile_<tag>/) into profiles/ and rewrites
profiles/<round>_traffic.json: HBM bytes per ray-march launch = 2 x FETCH_SIZE + WRITE_SIZE (see profiles/README.md), stamped
with the commit the profile was taken on and the kernel time of that run, so that a stale figure is visible in bench.py's line.
usage: update_profiles.py <tag> <round> [<commit>]"""
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "profile_" + sys.argv[1])
dst = os.path.join(ROOT, "profiles")
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
commit = sys.argv[3] if len(sys.argv) > 3 else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
shutil.copy(os.path.join(src, "kernel_stats_head.csv"), os.path.join(dst, f"{rnd}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(src, "pmc_summary.txt"), os.path.join(dst, f"{rnd}_bench_pmc_summary.txt"))
shutil.copy(os.path.join(src, "stats_bench.json"), os.path.join(dst, f"{rnd}_bench_under_rocprof.json"))
if os.path.exists(os.path.join(src, "timed_launches.json")):
    shutil.copy(os.path.join(src, "timed_launches.json"), os.path.join(dst, f"{rnd}_bench_timed_launches.json"))
vals = {}
for line in open(os.path.join(src, "pmc_summary.txt")):
    m = re.match(r"\S+\s+(\S+)\s+n=\d+ last=\S+ mean=(\S+)", line)
    if m:
        vals[m.group(1)] = float(m.group(2))
with open(os.path.join(src, "stats_bench.json")) as f:
    bench = json.loads([ln for ln in f if ln.startswith("{")][-1])
traffic = 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024
entry = {"bytes_per_launch": round(traffic), "commit": commit, "kernel_ms": bench["roofline"]["kernel_ms"]}
busy_file = os.path.join(ROOT, "gpurun_out", f"{rnd}_k", "valu_busy_u8.txt")       # scripts/valu_busy.py over the per-view counters of the same evidence run
if os.path.exists(busy_file):
    for line in open(busy_file):
        if line.startswith("VALU busy"):
            entry["valu_busy_per_view"] = [float(x) for x in line.split()[2:]]
            entry["valu_busy_note"] = ("share of the SIMDs' issue cycles that carried a vector ALU instruction, per benchmark view (SQ_ACTIVE_INST_VALU x 4 / "
                                       "(1024 SIMDs x GRBM_GUI_ACTIVE / 8), scripts/valu_busy.py): what bounds the lit march beside the memory path")
json.dump({"nooptims_trilinear_1024_2048_n1": entry,
           "_source": f"profiles/{rnd}_bench_pmc_summary.txt: mean over the 16 TIMED raymarch launches of rocprofv3 --pmc FETCH_SIZE "
                      f"({vals['FETCH_SIZE']:.6g} KB, x2: gfx950 tallies 128-B requests at 64 B, calibrated on minmax_kernel which reads "
                      f"exactly 1 GiB and reports 524312 KB) + WRITE_SIZE ({vals['WRITE_SIZE']:.6g} KB, exact), separate passes",
           "_command": "python bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-extras"},
          open(os.path.join(dst, f"{rnd}_traffic.json"), "w"), indent=1)
print(f"traffic {traffic / 1e9:.3f} GB per launch at commit {commit}; TCC hit rate {vals['TCC_HIT_sum'] / (vals['TCC_HIT_sum'] + vals['TCC_MISS_sum']):.2f}")
