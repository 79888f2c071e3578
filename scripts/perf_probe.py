#!/usr/bin/env python3
"""Per-view kernel timings of the ray-march kernel (hipEvents inside libvr_hip.so).  Tuning aid, run on the GPU box:
    python scripts/perf_probe.py --volume 1024 --viewport 2048 --views 0,1,2,3,4,5,6,7 --mode nooptims --sampling trilinear
"""
import argparse
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--volume", type=int, default=1024)
    ap.add_argument("--viewport", type=int, default=2048)
    ap.add_argument("--views", default="0,1,2,3,4,5,6,7")
    ap.add_argument("--pose", default="", help="instead of --views: 'ax,ay,az[,p]' camera angles in degrees at distance 2 (p = 1: perspective); several separated by ';'")
    ap.add_argument("--mode", default="nooptims")
    ap.add_argument("--sampling", default="trilinear")
    ap.add_argument("--kind", default="shell")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--light", type=float, default=0.6)
    ap.add_argument("--layout", default="bricked")
    ap.add_argument("--bpv", type=int, default=1, help="bytes per voxel of the generated volume")
    ap.add_argument("--tile-map", default="", help="lane_map,phase_x,phase_y (default: automatic)")
    ap.add_argument("--plane", type=int, default=-1, help="brick chunk plane: -1 per view, 0 xy, 1 xz, 2 yz, 3 / 4 run bricks along z / y, 5 oct bricks for every view (2-byte voxels)")
    ap.add_argument("--each", action="store_true", help="synchronise after every launch and list the kernel time of each")
    ap.add_argument("--sched", type=int, default=1, help="vr_hip_set_tile_scheduling: 0 workgroup order, 1 measured-cost order")
    ap.add_argument("--wide", type=int, default=0, help="vr_hip_set_wide_addressing value (2: 64-bit z tables, 1024-thread workgroups)")
    a = ap.parse_args()
    vr = importlib.import_module("volume-rendering_amd")
    r = vr.HipRenderer(0)
    r.set_layout(vr.LAYOUT_BRICKED if a.layout == "bricked" else vr.LAYOUT_LINEAR)
    r.set_brick_plane(a.plane)
    r.set_tile_scheduling(a.sched)
    if a.wide:
        r.set_wide_addressing(a.wide)
    if a.tile_map:
        r.set_tile_mapping(*[int(x) for x in a.tile_map.split(",")])
    n, W = a.volume, a.viewport
    r.generate_volume(a.kind, n, seed=1, bytes_per_voxel=a.bpv)
    mm, _, _, ms = r.volume_minmax()
    scene = vr.Scene().set_volume(dims=(n, n, n), minmax=mm)
    if a.mode == "nooptims":
        scene.set_modes(esl=False, ray_threshold=1.0)
    elif a.mode == "ertonly":                                # VolR.cpp:288-290: early ray termination without leaping
        scene.set_modes(esl=False)
    scene.set_modes(light_kd=a.light)
    r.set_transfer_fn(scene.tf, scene.esl)
    samp = {"trilinear": vr.SAMPLE_TRILINEAR, "q8": vr.SAMPLE_TRILINEAR_Q8}.get(a.sampling, vr.SAMPLE_NEAREST)
    buf = torch.empty((W, W, 4), dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    res, spread, chosen = {}, {}, {}
    poses = [tuple(float(x) for x in q.split(",")) for q in a.pose.split(";") if q]
    for v in (range(len(poses)) if poses else [int(x) for x in a.views.split(",")]):
        view = vr.custom_view(W, W, len(poses[v]) > 3 and poses[v][3] != 0, poses[v][:3], 2.0) if poses else vr.benchmark_view(W, W, v)
        p = scene.frame_params(view, samp)
        for _ in range(4):                                   # builds the brick copy; records the tile costs / builds the launch order or the per-tile copy choice
            r.render_volume_device(p, buf.data_ptr(), stream)
        torch.cuda.synchronize()
        r.timing_reset()
        if a.each:                                           # every launch on its own: min / max of the kernel time
            each = []
            for _ in range(a.reps):
                r.render_volume_device(p, buf.data_ptr(), stream)
                torch.cuda.synchronize()
                each.append(round(r.timing().kernel_ms, 4))
            spread[v] = each
            res[v] = round(sum(each) / len(each), 4)
            continue
        for _ in range(a.reps):
            r.render_volume_device(p, buf.data_ptr(), stream)
        torch.cuda.synchronize()
        t = r.timing()
        res[v] = round(t.kernel_ms_sum / t.launches, 4)
        li = r.last_launch()
        chosen[v] = "L%d/p%d lanes %d shape %d phase %d,%d straddle %d" % (li["layout"], li["brick_plane"], li["lane_map"] & 3, li["lane_map"] >> 2, li["phase_x"], li["phase_y"], li["straddle_permille"])
    print(json.dumps({"volume": n, "viewport": W, "mode": a.mode, "sampling": a.sampling, "layout": a.layout, "light": a.light, "sched": a.sched,
                      "kernel_ms_per_view": res, "mean_ms": round(sum(res.values()) / len(res), 4), "minmax_ms": round(ms, 4),
                      "chosen": chosen, **({"each": spread} if a.each else {})}))


if __name__ == "__main__":
    main()
