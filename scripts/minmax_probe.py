#!/usr/bin/env python3
"""Times the ESL min/max feeder (vr_hip_volume_minmax) at several volume sizes; prints GB/s against the 8 TB/s HBM peak."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
vr = importlib.import_module("volume-rendering_amd")
r = vr.HipRenderer(0)
r.set_layout(vr.LAYOUT_LINEAR)          # no brick copy needed for this probe
out = {}
for n, bpv in ((512, 1), (1024, 1), (1024, 2), (2048, 1), (2048, 2)):
    r.generate_volume("shell", n, seed=1, bytes_per_voxel=bpv)
    ms = min(r.volume_minmax()[3] for _ in range(6))
    h_ms = min(r.volume_histogram()[1] for _ in range(3))
    gb = n ** 3 * bpv / 1e9
    out[f"{n}^3 u{8*bpv}"] = {"minmax_ms": round(ms, 4), "GBs": round(gb / ms * 1e3, 1), "frac_of_8TBs": round(gb / ms * 1e3 / 8000, 3),
                              "histogram_ms": round(h_ms, 3), "histogram_GBs": round(gb / h_ms * 1e3, 1)}
print(json.dumps(out, indent=1))
