#!/usr/bin/env python3
"""Per-tile cost maps (vr_hip_set_tile_scheduling(2)) of one view under several forced volume copies: which screen regions are slow
with which copy.  Prints, per copy, the kernel time and the mean tile cost of a coarse grid of screen cells; with two copies also what a
per-tile choice of the cheaper copy would add up to.  Tuning aid, run on the GPU box."""
import argparse
import importlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--view", type=int, default=1)
    ap.add_argument("--planes", default="3,4", help="vr_hip_set_brick_plane values to compare (3 / 4 = run bricks along z / y, 0-2 quad planes)")
    ap.add_argument("--grid", type=int, default=8, help="coarse grid the map is averaged to for printing")
    ap.add_argument("--sampling", default="trilinear")
    ap.add_argument("--save", default="")
    a = ap.parse_args()
    vr = importlib.import_module("volume-rendering_amd")
    n, W = 1024, 2048
    r = vr.HipRenderer(0)
    r.generate_volume("shell", n, seed=1)
    scene = vr.Scene().set_volume(dims=(n, n, n), minmax=r.volume_minmax()[0])
    scene.set_modes(esl=False, ray_threshold=1.0)
    r.set_transfer_fn(scene.tf, scene.esl)
    samp = vr.SAMPLE_TRILINEAR if a.sampling == "trilinear" else vr.SAMPLE_NEAREST
    p = scene.frame_params(vr.benchmark_view(W, W, a.view), samp)
    buf = torch.empty((W, W, 4), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    maps, out = {}, {"view": a.view}
    r.set_tile_scheduling(2)
    for plane in [int(x) for x in a.planes.split(",")]:
        r.set_brick_plane(plane)
        for _ in range(3):
            r.render_volume_device(p, buf.data_ptr(), stream)
        torch.cuda.synchronize()
        r.timing_reset()
        acc = None
        for _ in range(4):
            r.render_volume_device(p, buf.data_ptr(), stream)
            torch.cuda.synchronize()
            m = r.tile_costs().astype(np.float64)
            acc = m if acc is None else acc + m
        t = r.timing()
        m = acc / 4
        maps[plane] = m
        g = a.grid
        ty, tx = m.shape
        coarse = m[: ty // g * g, : tx // g * g].reshape(g, ty // g, g, tx // g).mean(axis=(1, 3))
        out[f"plane{plane}"] = {"kernel_ms": round(t.kernel_ms_sum / t.launches, 4), "tiles": [int(ty), int(tx)], "sum_cost": float(m.sum()),
                                "coarse_rows_bottom_to_top": [[int(v) for v in row] for row in coarse]}
    planes = list(maps)
    if len(planes) >= 2:
        best = np.minimum.reduce([maps[k] for k in planes])
        out["per_tile_best_sum_cost"] = float(best.sum())
        out["share_of_tiles_won"] = {f"plane{k}": round(float((maps[k] == best).mean()), 3) for k in planes}
    if a.save:
        np.savez_compressed(a.save, **{f"plane{k}": v for k, v in maps.items()})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
