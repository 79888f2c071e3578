import importlib, sys, os, json
import numpy as np, torch
sys.path.insert(0, os.getcwd())
vr = importlib.import_module("volume-rendering_amd")
n, W = 1024, 2048
r = vr.HipRenderer(0)
r.generate_volume("shell", n, seed=1)
scene = vr.Scene().set_volume(dims=(n, n, n), minmax=r.volume_minmax()[0])
r.set_transfer_fn(scene.tf, scene.esl)
buf = torch.empty((W, W, 4), dtype=torch.uint8, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
out = {}
for v in (5, 3, 0):
    p = scene.frame_params(vr.benchmark_view(W, W, v), vr.SAMPLE_TRILINEAR)
    r.set_tile_scheduling(2)
    maps = []
    for _ in range(4):
        r.render_volume_device(p, buf.data_ptr(), stream); torch.cuda.synchronize()
        maps.append(r.tile_costs().astype(np.float64).ravel())
    m = maps[-1]
    q = np.percentile(m, [50, 90, 99, 99.9, 100])
    corr = float(np.corrcoef(maps[-1], maps[-2])[0, 1])
    out[v] = {"pct_50_90_99_999_max": [int(x) for x in q], "mean": int(m.mean()), "sum": int(m.sum()), "corr_between_frames": round(corr, 3),
              "top16_share_of_sum": round(float(np.sort(m)[-16:].sum() / m.sum()), 4), "tiles": int(m.size)}
print(json.dumps(out))
