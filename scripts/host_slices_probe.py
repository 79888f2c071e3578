#!/usr/bin/env python3
"""Prototype for the host-buffer entry point: the frame rendered as K contiguous row slices, each on its own
stream (descending priority), each slice copied to pinned host memory behind its kernel — how much of the 16 MiB D2H hides behind
the slices still rendering?  Wall ms per frame against the monolithic frame + one copy.  Tuning aid, run on the GPU box."""
import argparse
import copy
import importlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--volume", type=int, default=1024)
    ap.add_argument("--viewport", type=int, default=2048)
    ap.add_argument("--mode", default="nooptims")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    vr = importlib.import_module("volume-rendering_amd")
    r = vr.HipRenderer(0)
    n, W = a.volume, a.viewport
    r.generate_volume("shell", n, seed=1)
    mm, _, _, _ = r.volume_minmax()
    scene = vr.Scene().set_volume(dims=(n, n, n), minmax=mm)
    if a.mode == "nooptims":
        scene.set_modes(esl=False, ray_threshold=1.0)
    scene.set_modes(light_kd=0.6)
    r.set_transfer_fn(scene.tf, scene.esl)
    dev = torch.empty((W, W, 4), dtype=torch.uint8, device="cuda:0")
    host = torch.empty((W, W, 4), dtype=torch.uint8).pin_memory()
    lo, hi = (0, -1)
    views = [vr.benchmark_view(W, W, v) for v in range(8)]
    params = [vr.whole_frame(scene.frame_params(v, vr.SAMPLE_TRILINEAR)) for v in views]
    res = {"priority_range": [lo, hi]}

    def sliced(p, K, s):
        q = type(p).from_buffer_copy(p)
        rows = W // K
        q.out_rows = rows; q.band_rows = rows; q.band_stride = K; q.band_first = s
        return q

    for K, prio in ((1, False), (2, False), (2, True), (3, True), (4, True), (4, False), (8, True)):
        if prio:
            streams = [torch.cuda.Stream(priority=-1 if s < (K + 1) // 2 else 0) for s in range(K)]     # the first half of the slices ahead of the rest
        else:
            streams = [torch.cuda.Stream() for s in range(K)]
        rows = W // K
        ps = [[sliced(p, K, s) for s in range(K)] for p in params]
        per_view = []
        for rep in range(a.reps + 1):
            per_view = []
            for v in range(8):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for s in range(K):
                    r.render_volume_device(ps[v][s], dev[s * rows:].data_ptr(), streams[s].cuda_stream)
                for s in range(K):
                    with torch.cuda.stream(streams[s]):
                        host[s * rows:(s + 1) * rows].copy_(dev[s * rows:(s + 1) * rows], non_blocking=True)
                for s in range(K):
                    streams[s].synchronize()
                per_view.append((time.perf_counter() - t0) * 1e3)
        res[f"K{K}{'_prio' if prio else ''}"] = {"ms_mean": round(sum(per_view) / 8, 4), "per_view": [round(x, 3) for x in per_view]}
        print(f"K{K}{'_prio' if prio else ''}", res[f"K{K}{'_prio' if prio else ''}"], flush=True)
    # the kernel alone (no copy)
    per_view = []
    s0 = torch.cuda.Stream()
    for v in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.render_volume_device(params[v], dev.data_ptr(), s0.cuda_stream)
        s0.synchronize()
        per_view.append((time.perf_counter() - t0) * 1e3)
    res["kernel_only_wall"] = round(sum(per_view) / 8, 4)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
