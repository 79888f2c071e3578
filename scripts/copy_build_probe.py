#!/usr/bin/env python3
"""set_volume cost breakdown on the GPU box: generation of the synthetic volume and every brick-copy builder, each with its
HBM roofline (bytes = linear array read once + copy written once).  usage: copy_build_probe.py [n=1024] [bpv=1] [reps=3]
Prints one JSON object (also what bench.py's `set_volume.builders` leg reports)."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def probe(vr, r, n, bpv, reps=3):
    import torch
    out = {"volume": n, "bytes_per_voxel": bpv, "generate": [], "copies": {}}
    lin = n ** 3 * bpv
    for _ in range(reps):
        t0 = time.perf_counter()
        r.generate_volume("shell", n, seed=1, bytes_per_voxel=bpv)
        out["generate"].append(round(r.volume_info().upload_ms, 3))
        out.setdefault("generate_wall_ms", []).append(round((time.perf_counter() - t0) * 1e3, 3))
    g = min(out["generate"])
    out["generate_roofline"] = {"bound": "hbm", "bytes": lin, "ms": g, "achieved_GBs": round(lin / g / 1e6, 1), "frac": round(lin / g / 1e6 / HBM_PEAK_GBS, 4)}
    policy = r.volume_info().copies_in_policy
    for rep in range(reps):
        r.set_layout(vr.LAYOUT_BRICKED)                 # drops every copy
        t0 = time.perf_counter()
        try:
            r.prepare(policy)
        except vr.VrError as e:
            out["prepare_error"] = str(e)
        wall = (time.perf_counter() - t0) * 1e3
        info = r.volume_info()
        for k in range(vr.COPY_KINDS):
            if (info.copies >> k) & 1:
                out["copies"].setdefault(vr.COPY_NAMES[k], []).append(round(info.build_ms[k], 3))
        out.setdefault("prepare_wall_ms", []).append(round(wall, 2))
    info = r.volume_info()
    elems = ((n + 7) // 8) ** 3 * 512
    sizes = {"quad_xy": elems * 4 * bpv, "quad_xz": elems * 4 * bpv, "quad_yz": elems * 4 * bpv, "run_z": ((n + 7) // 8) ** 3 * 2304,
             "run_y": ((n + 7) // 8) ** 3 * 2304, "voxel": elems * bpv, "oct": elems * 8 * bpv,
             "col_x": ((n + 3) // 4) ** 2 * ((n + 2) // 3) * 256, "col_y": ((n + 3) // 4) ** 2 * ((n + 2) // 3) * 256, "col_z": ((n + 3) // 4) ** 2 * ((n + 2) // 3) * 256,
             "colv_x": ((n + 3) // 4) ** 2 * ((n + 15) // 16) * 256, "colv_y": ((n + 3) // 4) ** 2 * ((n + 15) // 16) * 256, "colv_z": ((n + 3) // 4) ** 2 * ((n + 15) // 16) * 256}
    out["rooflines"] = {}
    for name, ms in out["copies"].items():
        best = min(ms)
        b = lin + sizes[name]
        out["rooflines"][name] = {"bound": "hbm", "bytes": b, "ms": best, "first_ms": ms[0], "achieved_GBs": round(b / best / 1e6, 1),
                                  "frac": round(b / best / 1e6 / HBM_PEAK_GBS, 4)}
    out["refused"] = int(info.copies_refused)
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    bpv = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    import torch  # noqa: F401
    vr = importlib.import_module("volume-rendering_amd")
    r = vr.HipRenderer(0)
    print(json.dumps(probe(vr, r, n, bpv, reps)))
    r.close()
