#!/usr/bin/env python3
"""Whole-frame goldens of the TRILINEAR sampling modes at the BASELINE sizes, rendered by the CPU RESTATEMENT
(oracle/libvr_oracle.so, OpenMP).  TEST INFRASTRUCTURE, build container only.

GPURenderer4.cu (the reference's trilinear renderer, :53-87) needs nvcc + texture hardware and cannot run here, so these frames
pin HIP == restatement (bit for bit), not HIP == reference: "parity unpinned for the trilinear sampler" stays as DESIGN.md says.
What they add over the 16-row band checks of tests/test_gpu_fullsize.py: every one of the 8 benchmark views (VolR.cpp:232-248)
— i.e. every brick copy, tile phase and lane order the per-view policy of vr_hip_api.cpp picks — is compared over the WHOLE
frame at the sizes where the 32-bit byte offsets of the quad copy (exactly 2^32 bytes at 1024^3) and the 64-bit run tables are
actually reached.

  C2: shell 256^3 @ 1024x1024    C3: shell 512^3  (seed 1) @ 1920x1080      C4: shell 1024^3 (seed 1) @ 2048x2048
  8 views x {default: ESL on, threshold 0.95; nooptims: ESL off, threshold 1.0} x {TRILINEAR, TRILINEAR_Q8}, light 0.6
  C5: shell 2048^3 uint16 @ 4096x4096 (C5_PLAN below: default mode TRILINEAR + NEAREST for all 8 views, full march for views 0, 1, 5)

Only hashes travel (tests/golden/golden_fullsize_trilinear.json): FNV-1a32 of the RGBA8 frame + the number of pixels with
non-zero alpha.  The file is rewritten after every frame, so an interrupted run resumes where it stopped.
usage: gen_golden_fullsize_trilinear.py [c2] [c3] [c4] [c5]
"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "tests", "golden", "golden_fullsize_trilinear.json")

CONFIGS = {"c2": (256, 1024, 1024, 1), "c3": (512, 1920, 1080, 1), "c4": (1024, 2048, 2048, 1), "c5": (2048, 4096, 4096, 2)}
MODES = {"default": (True, 0.95), "nooptims": (False, 1.0)}
# C5 (2048^3 uint16 @ 4096^2, beyond the reference's 32-bit index arithmetic, ModelBase.h:12 / ModelBase.cpp:95-98): restatement only,
# NEAREST included; the full march (34 G samples per frame) for three views — 0 (along an axis, oct bricks), 1 (oblique, oct bricks),
# 5 (oblique perspective, quad bricks) — the default mode for all eight.
C5_PLAN = [("trilinear", "default", v) for v in range(8)] + [("nearest", "default", v) for v in range(8)] + \
          [("trilinear", "nooptims", v) for v in (0, 1, 5)]


def key(c):
    return (c["config"], c["sampling"], c["mode"], c["view"])


def main():
    from helpers import Oracle, fnv1a32
    vr = importlib.import_module("volume-rendering_amd")
    only = [a for a in sys.argv[1:] if a in CONFIGS] or list(CONFIGS)
    threads = int(os.environ.get("GEN_THREADS", "8"))
    cases = []
    if os.path.exists(OUT):
        with open(OUT) as f:
            cases = json.load(f)["cases"]
    done = {key(c) for c in cases}
    oracle = Oracle()
    t0 = time.time()
    codes = {"trilinear": vr.SAMPLE_TRILINEAR, "trilinear_q8": vr.SAMPLE_TRILINEAR_Q8, "nearest": vr.SAMPLE_NEAREST}
    for cfg in only:
        n, W, H, bpv = CONFIGS[cfg]
        plan = C5_PLAN if cfg == "c5" else [(s, m, v) for s in ("trilinear", "trilinear_q8") for m in MODES for v in range(8)]
        plan = [t for t in plan if (cfg,) + t not in done]
        if not plan:
            continue
        vox = oracle.generate_volume("shell", n, 1, bytes_per_voxel=bpv)
        if bpv == 1:
            scene = vr.Scene().set_volume(voxels=vox)
        else:                                       # 2-byte voxels: the ESL grid works on the high byte (the host mirror takes min/max pairs)
            scene = vr.Scene().set_volume(dims=(n, n, n), minmax=oracle.volume_minmax(vox)[0])
        for sname, mode, view in plan:
            samp = codes[sname]
            esl, thr = MODES[mode]
            scene.set_modes(esl=esl, ray_threshold=thr)
            t1 = time.time()
            out = oracle.render(scene.frame_params(vr.benchmark_view(W, H, view), samp), vox, scene.tf, scene.esl, threads=threads)
            c = {"config": cfg, "volume": n, "width": W, "height": H, "view": view, "mode": mode, "sampling": sname,
                 "fnv": fnv1a32(out), "nonzero_alpha": int((out[..., 3] != 0).sum()),
                 "ray_step": float(scene.params.ray_step), "light_kd": float(scene.params.light_kd)}
            cases.append(c)
            print(f"[{time.time() - t0:6.0f}s] {cfg} {sname:12s} {mode:8s} view {view} {c['fnv']} nonzero {c['nonzero_alpha']} "
                  f"({time.time() - t1:.0f} s)", file=sys.stderr, flush=True)
            with open(OUT + ".tmp", "w") as f:
                json.dump({"generator": "oracle/gen_golden_fullsize_trilinear.py",
                           "renderer": "CPU restatement (oracle/vr_oracle.c), TRILINEAR / TRILINEAR_Q8 — not the reference: GPURenderer4.cu cannot run here",
                           "cases": sorted(cases, key=key)}, f, indent=1)
            os.replace(OUT + ".tmp", OUT)
        del vox
    print(f"wrote {OUT}: {len(cases)} cases", file=sys.stderr)


if __name__ == "__main__":
    main()
