#!/usr/bin/env python3
"""V_touched for the reference's optimised modes at the headline size: bytes of the distinct 128-byte lines
of the voxel array that a frame's sample set reads, counted by the CPU restatement's instrumentation
(oracle/vr_oracle.c `touch`, vro_render(count_lines=1)).  TEST INFRASTRUCTURE, build container only.

  shell 1024^3 u8 (seed 1) @ 2048x2048, the 8 benchmark views, light 0.6
  modes: default (ESL on, threshold 0.95) and ertonly (ESL off, threshold 0.95)  — VolR.cpp:288-294
  sampling: nearest (CPURenderer semantics) and trilinear (GPURenderer4 semantics)

Writes tests/golden/vtouched.json: per key "<mode>_<sampling>_<n>_<W>" the per-view byte counts and their mean, which
bench.py uses as the algorithmic bytes of those modes (B_alg = V_touched + 4*W*H).  The full march touches every voxel.
"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "tests", "golden", "vtouched.json")


def main():
    from helpers import Oracle
    vr = importlib.import_module("volume-rendering_amd")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    oracle = Oracle()
    vox = oracle.generate_volume("shell", n, 1)
    scene = vr.Scene().set_volume(voxels=vox)
    views = [vr.benchmark_view(W, W, i) for i in range(8)]
    res = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            res = json.load(f)
    t0 = time.time()
    for mode, esl in (("default", True), ("ertonly", False)):
        scene.set_modes(esl=esl, ray_threshold=0.95)
        for sname, samp in (("nearest", vr.SAMPLE_NEAREST), ("trilinear", vr.SAMPLE_TRILINEAR)):
            per_view, samples = [], []
            for v in views:
                _, st = oracle.render(scene.frame_params(v, samp), vox, scene.tf, scene.esl, threads=int(os.environ.get("GEN_THREADS", "8")),
                                      stats=True, count_lines=True)
                per_view.append(int(st.lines_touched) * 128)
                samples.append(int(st.samples))
                print(f"[{time.time() - t0:5.0f}s] {mode} {sname} view {len(per_view) - 1}: {per_view[-1] / 2**20:.1f} MiB, "
                      f"{samples[-1] / 1e6:.1f} M samples", file=sys.stderr, flush=True)
            res[f"{mode}_{sname}_{n}_{W}"] = {"per_view_bytes": per_view, "mean_bytes": sum(per_view) // len(per_view),
                                              "per_view_samples": samples, "volume_bytes": n ** 3}
    res["generator"] = "oracle/gen_vtouched.py (oracle/vr_oracle.c line instrumentation, 128-byte lines of the linear voxel array)"
    with open(OUT, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
