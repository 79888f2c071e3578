#!/usr/bin/env python3
"""The automatic tile mapping of a view (vr_hip_last_launch) and the kernel time of every tile phase (8 x 8) under given lane maps.
Full march, 1024^3 @ 2048^2.  Tuning aid, run on the GPU box."""
import argparse
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", default="0,2")
    ap.add_argument("--lane-maps", default="2,0")
    ap.add_argument("--sampling", default="trilinear")
    a = ap.parse_args()
    vr = importlib.import_module("volume-rendering_amd")
    n, W = 1024, 2048
    r = vr.HipRenderer(0)
    r.generate_volume("shell", n, seed=1)
    scene = vr.Scene().set_volume(dims=(n, n, n), minmax=r.volume_minmax()[0])
    scene.set_modes(esl=False, ray_threshold=1.0)
    r.set_transfer_fn(scene.tf, scene.esl)
    samp = vr.SAMPLE_TRILINEAR if a.sampling == "trilinear" else vr.SAMPLE_NEAREST
    buf = torch.empty((W, W, 4), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream

    def measure(p, reps=3):
        for _ in range(2):
            r.render_volume_device(p, buf.data_ptr(), stream)
        torch.cuda.synchronize()
        r.timing_reset()
        for _ in range(reps):
            r.render_volume_device(p, buf.data_ptr(), stream)
        torch.cuda.synchronize()
        t = r.timing()
        return round(t.kernel_ms_sum / t.launches, 3)

    for v in [int(x) for x in a.views.split(",")]:
        view = vr.benchmark_view(W, W, v)
        p = scene.frame_params(view, samp)
        r.set_tile_mapping(-1, 0, 0)
        measure(p, 3)
        auto = measure(p, 5)
        out = {"view": v, "auto_ms": auto, "auto": r.last_launch(),
               "direction": [round(x, 9) for x in view.direction], "right": [round(x * 1024, 6) for x in view.right_plane], "up": [round(x * 1024, 6) for x in view.up_plane],
               "origin": [round(x, 6) for x in view.origin]}
        for lm in [int(x) for x in a.lane_maps.split(",")]:
            grid = []
            for py in range(8):
                row = []
                for px in range(8):
                    r.set_tile_mapping(lm, px, py)
                    row.append(measure(p, 2))
                grid.append(row)
            out[f"lane_map_{lm}_rows_phase_y_cols_phase_x"] = grid
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
