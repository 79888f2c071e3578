#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output (one dir per pass) per kernel: mean counter value per dispatch.
usage: pmc_summary.py <root> [kernel substring] [first:count]   — first:count restricts the mean to that range of the kernel's
dispatches in dispatch order (bench.py: 32 set-up + W warm-up launches precede the K timed ones)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "raymarch"
first, count = (int(x) for x in sys.argv[3].split(":")) if len(sys.argv) > 3 else (0, 0)
for d in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if pat in row.get("Kernel_Name", ""):
                    acc[row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
        for k, v in acc.items():
            per = collections.defaultdict(float)
            for did, val in v:
                per[did] += val
            vals = [per[k2] for k2 in sorted(per)]
            total = len(vals)
            if count:
                vals = vals[first:first + count]
            print(f"{os.path.basename(d):8s} {k:40s} n={len(vals)} last={vals[-1]:.6g} mean={sum(vals)/len(vals):.6g}" + (f" (dispatches {first}..{first + len(vals) - 1} of {total})" if count else ""))
