import importlib, sys, os, json
sys.path.insert(0, os.getcwd())
vr = importlib.import_module("volume-rendering_amd")
r = vr.HipRenderer(0)
out = {}
for bpv in (1, 2):
    r.generate_volume("shell", 1024, seed=1, bytes_per_voxel=bpv)
    for _ in range(3):
        h, ms = r.volume_histogram()
    mm = r.volume_minmax(); mm = r.volume_minmax()
    out[f"u{8*bpv}"] = {"histogram_ms": round(ms, 4), "histogram_GBs": round(1024**3 * bpv / ms / 1e6, 1), "minmax_ms": round(mm[3], 4), "minmax_GBs": round(1024**3 * bpv / mm[3] / 1e6, 1), "hist_sum": int(sum(h))}
print(json.dumps(out))
