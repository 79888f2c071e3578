#!/usr/bin/env python3
"""The timed ray-march launches of a `rocprofv3 --kernel-trace -- python bench.py ...` run, from the kernel trace: bench.py renders 8
set-up frames (one per view), W warm-up steps, then the K timed steps; prints a JSON object with their durations in start order.
usage: timed_launches.py <dir with *kernel_trace.csv> <warmup> <steps> [setup frames = 8]"""
import csv
import glob
import json
import os
import sys

root, warmup, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
setup = int(sys.argv[4]) if len(sys.argv) > 4 else 8
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "march_kernel" in row["Kernel_Name"]:          # vr::raymarch_kernel<...> and vr::colmarch_kernel<...>
                rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
rows.sort()
ms = [(e - s) / 1e6 for s, e, _ in rows]
timed = ms[setup + warmup: setup + warmup + steps]
names = [("col" if "colmarch" in n else "ray") + n.split("march_kernel")[1].split(">")[0] + ">" for _, _, n in rows[setup + warmup: setup + warmup + steps]]
print(json.dumps({"command": f"rocprofv3 --kernel-trace --stats -- python bench.py --steps {steps} --warmup {warmup} --no-cpu-baseline --no-extras",
                  "raymarch_launches": len(ms), "first_launch_ms": round(ms[0], 3) if ms else None,
                  "timed_steps_ms": [round(x, 4) for x in timed], "timed_instantiations": names,
                  "timed_mean_ms": round(sum(timed) / max(1, len(timed)), 4), "timed_max_ms": round(max(timed), 4) if timed else None,
                  "note": f"launches {setup + warmup + 1}..{setup + warmup + steps} in start order are the timed steps ({setup} set-up frames, one per view, and {warmup} warm-up "
                          "frames precede them); the summary csv averages ALL launches of an instantiation, including the very first one on each brick copy "
                          "(first touch of 4-5 GiB: several ms) - bench.py's set-up pass keeps those out of its timed region"}, indent=1))
