#!/usr/bin/env python3
"""Frame time of the host-buffer entry point vr_hip_render (kernel + 16 MiB D2H, reference renderer ids 0-2) beside the
device-buffer entry point, headline workload.  Run on the GPU box."""
import importlib, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
vr = importlib.import_module("volume-rendering_amd")
r = vr.HipRenderer(0)
n, W = 1024, 2048
r.generate_volume("shell", n, seed=1)
mm = r.volume_minmax()[0]
scene = vr.Scene().set_volume(dims=(n, n, n), minmax=mm)
scene.set_modes(esl=False, ray_threshold=1.0)
r.set_transfer_fn(scene.tf, scene.esl)
ps = [scene.frame_params(vr.benchmark_view(W, W, v), vr.SAMPLE_TRILINEAR) for v in range(8)]
r.set_window_buffer(W, W)
for p in ps:
    r.render_volume(p)
t0 = time.perf_counter()
for _ in range(2):
    for p in ps:
        r.render_volume(p)
host_ms = (time.perf_counter() - t0) / 16 * 1e3
buf = torch.empty((W, W, 4), dtype=torch.uint8, device="cuda:0")
s = torch.cuda.current_stream().cuda_stream
for p in ps:
    r.render_volume_device(p, buf.data_ptr(), s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2):
    for p in ps:
        r.render_volume_device(p, buf.data_ptr(), s)
torch.cuda.synchronize()
dev_ms = (time.perf_counter() - t0) / 16 * 1e3
print(json.dumps({"host_buffer_ms_per_frame": round(host_ms, 3), "device_buffer_ms_per_frame": round(dev_ms, 3),
                  "Mrays_per_s_host_buffer": round(W * W / host_ms / 1e3, 1)}))
