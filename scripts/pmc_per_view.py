#!/usr/bin/env python3
"""Per-view summary of rocprofv3 --pmc passes taken over scripts/perf_probe.py (one directory per pass, as written by
scripts/gpu_pmc.sh): the ray-march dispatches appear in view order, (reps + 1) launches per view; prints the mean of each
counter per view, over the LAST <take last> launches of each view if given (the first launches of a view build copies, record tile costs,
or run on a single run copy before the per-tile choice exists).   usage: pmc_per_view.py <root> <launches per view> [kernel substring] [take last]"""
import collections
import csv
import glob
import os
import sys

root, per_view = sys.argv[1], int(sys.argv[2])
pat = sys.argv[3] if len(sys.argv) > 3 else "raymarch"
last = int(sys.argv[4]) if len(sys.argv) > 4 else 0
for d in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))     # counter -> dispatch -> value
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if pat in row.get("Kernel_Name", ""):
                    acc[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
        for name, per in sorted(acc.items()):
            ids = sorted(per)
            views = [ids[i:i + per_view][-last:] for i in range(0, len(ids), per_view)]
            vals = ["%.4g" % (sum(per[i] for i in g) / len(g)) for g in views]
            print(f"{os.path.basename(d):6s} {name:36s} " + " ".join(f"{v:>10s}" for v in vals))
