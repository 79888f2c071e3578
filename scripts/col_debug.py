#!/usr/bin/env python3
"""Debug aid: the poses of tests/test_gpu_parity.py::test_column_march_matches_oracle one by one, printing before every render."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import Golden, Oracle
vr = importlib.import_module("volume-rendering_amd")
golden, oracle = Golden(), Oracle()
gpu = vr.HipRenderer(0)
poses = ((0.0, 0.0, 0.0), (90.0, 0.0, 0.0), (180.0, 90.0, 0.0), (0.0, 90.0, 0.0), (270.0, 0.0, 0.0), (0.0, 180.0, 0.0), (90.0, 90.0, 0.0), (0.0, 0.0, 90.0), (0.02, 0.0, 0.0), (90.0, 0.013, 0.0), (-45.0, -45.0, 0.0), (1.5, 2.5, 0.0))
bad = 0
for name, label, sizes in (("bucky", "bench64_view1_default", ((256, 256), (130, 67))), ("blob_40x24x56", "view1_default", ((192, 160),))):
    st = golden.volume_state(name)
    gpu.set_transfer_fn(st["tf"], st["esl"]); gpu.set_volume(golden.voxels(name))
    case = [c for c in golden.cases(True) if c["label"] == label and c["volume"] == name][0]
    for (w, h) in sizes:
        gpu.set_window_buffer(w, h)
        for angles in poses:
            for plane in (-1, 8):
                p = golden.params(case, vr.SAMPLE_TRILINEAR)
                v = vr.custom_view(w, h, False, angles, 2.0)
                for f in ("origin", "direction", "right_plane", "up_plane"):
                    for j in range(3):
                        getattr(p.view, f)[j] = getattr(v, f)[j]
                p.view.width, p.view.height, p.view.perspective = w, h, 0
                p = vr.whole_frame(p)
                p.esl, p.ray_threshold = 0, 1.0
                want = oracle.render(p, golden.voxels(name), st["tf"], st["esl"])
                gpu.set_brick_plane(plane)
                print("render", name, (w, h), angles, plane, flush=True)
                out = gpu.render_volume(p)
                torch.cuda.synchronize()
                d = int((out != want).any(axis=-1).sum())
                print("   layout", gpu.last_launch()["layout"], "diff px", d, flush=True)
                bad += d != 0
print("bad", bad)
