#!/usr/bin/env python3
"""Exhaustive sweep of vr_hip_set_tile_mapping (lane order x wave shape x tile phase) per benchmark view against the automatic choice:
does the per-frame chooser leave anything on the table?  Full march, 1024^3 @ 2048^2.  Tuning aid, run on the GPU box."""
import argparse
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", default="0,1,2,3,4,5,6,7")
    ap.add_argument("--sampling", default="trilinear")
    ap.add_argument("--lane-maps", default="0,1,2,4,5,6,8,9,10")
    ap.add_argument("--phases", default="0,1,2,3,4,5,6,7")
    ap.add_argument("--plane", type=int, default=-1)
    a = ap.parse_args()
    vr = importlib.import_module("volume-rendering_amd")
    n, W = 1024, 2048
    r = vr.HipRenderer(0)
    r.set_brick_plane(a.plane)
    r.generate_volume("shell", n, seed=1)
    scene = vr.Scene().set_volume(dims=(n, n, n), minmax=r.volume_minmax()[0])
    scene.set_modes(esl=False, ray_threshold=1.0)
    r.set_transfer_fn(scene.tf, scene.esl)
    samp = vr.SAMPLE_TRILINEAR if a.sampling == "trilinear" else vr.SAMPLE_NEAREST
    buf = torch.empty((W, W, 4), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    phases = [int(x) for x in a.phases.split(",")]

    def measure(p, reps=3):
        for _ in range(2):
            r.render_volume_device(p, buf.data_ptr(), stream)
        torch.cuda.synchronize()
        r.timing_reset()
        for _ in range(reps):
            r.render_volume_device(p, buf.data_ptr(), stream)
        torch.cuda.synchronize()
        t = r.timing()
        return t.kernel_ms_sum / t.launches

    out = {}
    for v in [int(x) for x in a.views.split(",")]:
        p = scene.frame_params(vr.benchmark_view(W, W, v), samp)
        r.set_tile_mapping(-1, 0, 0)
        auto = measure(p, 5)
        best = []
        for lm in [int(x) for x in a.lane_maps.split(",")]:
            for px in phases:
                for py in phases:
                    r.set_tile_mapping(lm, px, py)
                    best.append((round(measure(p, 2), 4), lm, px, py))
        best.sort()
        top = best[:5]
        # confirm the best few with more repetitions
        confirmed = []
        for _, lm, px, py in top:
            r.set_tile_mapping(lm, px, py)
            confirmed.append((round(measure(p, 6), 4), lm, px, py))
        r.set_tile_mapping(-1, 0, 0)
        auto2 = measure(p, 5)
        out[v] = {"auto_ms": [round(auto, 4), round(auto2, 4)], "best": sorted(confirmed), "worst": best[-1], "median": best[len(best) // 2]}
        print(json.dumps({v: out[v]}), flush=True)


if __name__ == "__main__":
    main()
