#!/usr/bin/env python3
"""Does a rank of an N-rank run gain from two frames rendering CONCURRENTLY (two streams) instead of back to back (one stream)?
At N = 8 a rank's share of the 2048^2 frame is one full load of the chip (8192 waves): a single round of waves marching in lockstep.
Renders rank 0's band set of the 8 views, `frames` times, on one stream and alternating between two; wall ms per frame."""
import argparse
import importlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--mode", default="nooptims")
    ap.add_argument("--skip-streams", type=int, default=0)
    ap.add_argument("--streams", default="1,2", help="numbers of streams (= frames in flight rendering concurrently) to compare")
    ap.add_argument("--band-rows", type=int, default=0, help="rows per interleaved band (0 = the default rule)")
    a = ap.parse_args()
    vr = importlib.import_module("volume-rendering_amd")
    dmod = importlib.import_module("volume-rendering_amd.distributed")
    n, W = 1024, 2048
    r = vr.HipRenderer(0)
    r.generate_volume("shell", n, seed=1)
    scene = vr.Scene().set_volume(dims=(n, n, n), minmax=r.volume_minmax()[0])
    if a.mode == "nooptims":
        scene.set_modes(esl=False, ray_threshold=1.0)
    r.set_transfer_fn(scene.tf, scene.esl)
    views = [vr.benchmark_view(W, W, i) for i in range(8)]
    out = {}
    spare = [torch.cuda.Stream() for _ in range(a.skip_streams)]      # other streams of the process that hold a hardware queue
    for sp in spare:
        with torch.cuda.stream(sp):
            torch.zeros(16, device="cuda:0").add_(1)
    torch.cuda.synchronize()
    counts = [int(x) for x in a.streams.split(",")]
    pair = [torch.cuda.Stream() for _ in range(max(counts))]
    for world in [int(x) for x in a.ranks.split(",")]:
        split = dmod.FrameSplit(W, W, world, 0, a.band_rows or None)
        ps = [split.apply(scene.frame_params(v, vr.SAMPLE_TRILINEAR)) for v in views]
        bufs = [split.local_buffer("cuda:0") for _ in range(max(counts))]
        streams = pair                                         # ONE pair of streams for every measurement of the run
        for p in ps:
            for _ in range(3):
                r.render_volume_device(p, bufs[0].data_ptr(), streams[0].cuda_stream)
        torch.cuda.synchronize()
        res = {f"streams_{c}": [] for c in counts}
        for rep in range(3):                                   # interleaved repetitions: one, two, one, two, ...
            for label, nstreams in ((f"streams_{c}", c) for c in counts):
                t0 = time.perf_counter()
                for i in range(a.frames):
                    s = i % nstreams
                    r.render_volume_device(ps[i % 8], bufs[s].data_ptr(), streams[s].cuda_stream)
                torch.cuda.synchronize()
                res[label].append(round((time.perf_counter() - t0) / a.frames * 1e3, 4))
        out[f"n{world}"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
