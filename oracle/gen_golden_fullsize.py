#!/usr/bin/env python3
"""Whole-frame goldens at the BASELINE sizes, rendered by the REFERENCE'S OWN CPU path (oracle/_ref/libvolr_ref.so =
CPURenderer.cpp:43-53 compiled from /root/reference by oracle/Makefile).  TEST INFRASTRUCTURE, build container only.

  C2: shell 256^3 (seed 1) @ 1024x1024    C3: shell 512^3  (seed 1) @ 1920x1080      C4: shell 1024^3 (seed 1) @ 2048x2048
  8 benchmark views (VolR.cpp:232-248) x {default: ESL on, threshold 0.95; nooptims: ESL off, threshold 1.0}, light 0.6

Only hashes travel (tests/golden/golden_fullsize.json): FNV-1a32 of the RGBA8 frame + the number of pixels with
non-zero alpha, like the c2_* cases of golden.json.  `-m gpu` tests compare the HIP NEAREST whole frame with them
(tests/test_gpu_fullsize.py).  The CPU renderer is serial, so one process per frame is used (8 at a time).

It also records, per C4 view, how long the reference took on one core (the `cpu_baseline` sanity figure of DESIGN.md).
"""
import ctypes as C
import importlib
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libvolr_ref.so")
OUT = os.path.join(ROOT, "tests", "golden", "golden_fullsize.json")

CONFIGS = {"c2": (256, 1024, 1024), "c3": (512, 1920, 1080), "c4": (1024, 2048, 2048)}
MODES = {"default": (1, 0.95), "nooptims": (0, 1.0)}


def render_one(task):
    cfg, view_index, mode = task
    n, W, H = CONFIGS[cfg]
    from helpers import Oracle                    # the restatement's generator + FNV (test infrastructure)
    vr = importlib.import_module("volume-rendering_amd")
    oracle = Oracle()
    vox = oracle.generate_volume("shell", n, 1)
    L = C.CDLL(REF_SO)
    devnull = os.open(os.devnull, os.O_WRONLY)    # the reference's Logger prints to stdout
    os.dup2(devnull, 1)
    L.volr_ref_init()
    assert L.volr_ref_set_volume(vox.ctypes.data_as(C.POINTER(C.c_ubyte)), n, n, n) == 0
    f6, i2 = np.zeros(6, np.float32), np.zeros(2, np.uint32)
    L.volr_ref_get_params(f6.ctypes.data_as(C.POINTER(C.c_float)), i2.ctypes.data_as(C.POINTER(C.c_uint)))
    esl, thr = MODES[mode]
    L.volr_ref_set_params(C.c_float(float(f6[0])), C.c_float(thr), C.c_float(float(f6[2])), esl)
    v = vr.benchmark_view(W, H, view_index)
    v15 = np.array(list(v.origin) + list(v.direction) + list(v.right_plane) + list(v.up_plane) + list(v.light_pos), np.float32)
    out = np.zeros((H, W, 4), np.uint8)
    secs = C.c_double()
    rc = L.volr_ref_render(W, H, v15.ctypes.data_as(C.POINTER(C.c_float)), int(v.perspective),
                           out.ctypes.data_as(C.POINTER(C.c_ubyte)), C.byref(secs))
    assert rc == 0
    fnv = "%08x" % oracle.L.vro_fnv1a32(out.ctypes.data_as(C.c_void_p), C.c_uint64(out.size))
    return {"config": cfg, "volume": n, "width": W, "height": H, "view": view_index, "mode": mode,
            "fnv": fnv, "nonzero_alpha": int((out[..., 3] != 0).sum()), "ref_seconds_1core": round(secs.value, 2),
            "ray_step": float(f6[0]), "light_kd": float(f6[2])}


def main():
    only = sys.argv[1:] or list(CONFIGS)
    tasks = [(cfg, v, mode) for cfg in only for mode in MODES for v in range(8)]
    tasks.sort(key=lambda t: (t[0] != "c4", t[2] != "nooptims"))       # longest first
    t0 = time.time()
    with mp.get_context("spawn").Pool(int(os.environ.get("GEN_PROCS", "8"))) as pool:
        cases = []
        for r in pool.imap_unordered(render_one, tasks):
            cases.append(r)
            print(f"[{time.time() - t0:6.0f}s] {r['config']} view {r['view']} {r['mode']:8s} {r['fnv']} "
                  f"nonzero {r['nonzero_alpha']} ({r['ref_seconds_1core']} s)", file=sys.stderr, flush=True)
    cases.sort(key=lambda r: (r["config"], r["mode"], r["view"]))
    old = []
    if os.path.exists(OUT):
        with open(OUT) as f:
            old = [c for c in json.load(f)["cases"] if c["config"] not in only]
    with open(OUT, "w") as f:
        json.dump({"generator": "oracle/gen_golden_fullsize.py", "renderer": "reference CPURenderer (oracle/_ref), NEAREST",
                   "cases": sorted(old + cases, key=lambda r: (r["config"], r["mode"], r["view"]))}, f, indent=1)
    print(f"wrote {OUT}: {len(old + cases)} cases", file=sys.stderr)


if __name__ == "__main__":
    main()
