"""Legs of bench.py that are NOT the timed headline (rank 0, one GPU): the other BASELINE configurations, the linear layout, the
reference's own timed region (host buffer, D2H included), the rank simulation behind `scale_model`, and the host overhead of the
single-process several-device path.  Everything here goes through the product's C ABI; nothing touches oracle/."""
import time

import torch

HBM_PEAK_GBS = 8000.0

# BASELINE.json configs: name -> (cube edge, width, height, bytes per voxel)
CONFIGS = {"c2": (256, 1024, 1024, 1), "c3": (512, 1920, 1080, 1), "c4": (1024, 2048, 2048, 1), "c5": (2048, 4096, 4096, 2)}


def set_mode(scene, mode):
    """The three configurations of the reference's optimisation benchmark, VolR.cpp:283-294."""
    if mode == "nooptims":
        scene.set_modes(esl=False, ray_threshold=1.0)
    elif mode == "ertonly":
        scene.set_modes(esl=False, ray_threshold=0.95)
    else:
        scene.set_modes(esl=True, ray_threshold=0.95)


def time_views(r, params, buf, stream, sync, warm=4, reps=3):
    """hipEvent kernel time of `reps` launches per parameter set after `warm` untimed ones (the first builds the brick copy the
    view reads, the first two record / build the measured-cost tile order, the first four the per-tile copy choice of the views
    that are not along an axis): per-view mean, overall mean and max."""
    per_view, worst = [], 0.0
    for p in params:
        for _ in range(warm):
            r.render_volume_device(p, buf.data_ptr(), stream)
        sync()
        r.timing_reset()
        for _ in range(reps):
            r.render_volume_device(p, buf.data_ptr(), stream)
        sync()
        t = r.timing()
        per_view.append(t.kernel_ms_sum / max(1, t.launches))
        worst = max(worst, t.kernel_ms_max)
    return per_view, sum(per_view) / len(per_view), worst


def first_visit_leg(vr, r, scene, W, H, buf, stream, sync, modes=("nooptims", "default"), samplings=("trilinear", "nearest")):
    """The regime the reference's own benchmark and its interactive loop run in (VolR.cpp:225-248: each of the 8 views ONCE per
    configuration; VolR.cpp:98-113 / UI.cpp:117-139: a new View every frame) beside the repeated-view regime of the headline.  Copies
    resident (set-up frames elsewhere have built them), then per mode and sampling:
      protocol : the 8 benchmark views, each rendered once right after the policy state was dropped (vr_hip_set_tile_scheduling resets it),
                 four rounds — kernel ms of those FIRST frames, mean / max / per view;
      moving   : 64 DISTINCT views (every benchmark pose turned by j * (0.4, 0.3, 0) degrees, j = 1..8: no parameter set ever repeats), rendered
                 back to back — ms per frame, mean / max;
      steady   : the same 8 + 64 views, each after three earlier identical frames (what the headline's timed region sees).
    Nothing here depends on an earlier frame with identical parameters except the `steady` figures."""
    poses = ((0.0, 0.0, 0.0), (-45.0, -45.0, 0.0), (90.0, 0.0, 0.0), (180.0, 90.0, 0.0))
    exact = [vr.benchmark_view(W, H, i) for i in range(8)]
    moved = [vr.custom_view(W, H, i >= 4, (poses[i & 3][0] + 0.4 * j, poses[i & 3][1] + 0.3 * j, poses[i & 3][2]), 2.0) for i in range(8) for j in range(1, 9)]
    code = {"trilinear": vr.SAMPLE_TRILINEAR, "nearest": vr.SAMPLE_NEAREST}

    def once(p):
        r.timing_reset()
        r.render_volume_device(p, buf.data_ptr(), stream)
        sync()
        return r.timing().kernel_ms

    out = {"what": "first frames (no earlier frame with the same parameters: the reference's one-frame-per-view benchmark protocol, and a camera "
                   "that moves every frame) against repeated frames; kernel ms (hipEvents); copies already resident.  In the leaping modes the first frame of a view launches in a PREDICTED order "
                   "(tile costs estimated from the ESL bit volume): tile_estimate_kernel + tile_order_kernel run in front of it, 0.035 ms together "
                   "(profiles/r04_first_visit_prepass.txt), which these kernel times do not include"}
    for mode in modes:
        set_mode(scene, mode)
        for sname in samplings:
            ps_exact = [scene.frame_params(v, code[sname]) for v in exact]
            ps_moved = [scene.frame_params(v, code[sname]) for v in moved]
            for p in ps_exact + ps_moved[::8]:                  # copies these views read exist before anything is timed
                r.render_volume_device(p, buf.data_ptr(), stream)
            sync()
            first = [[] for _ in range(8)]
            for _ in range(4):
                r.set_tile_scheduling(1)                        # forgets every remembered order / recording
                for i, p in enumerate(ps_exact):
                    first[i].append(once(p))
            steady = []
            for p in ps_exact:
                for _ in range(3):
                    r.render_volume_device(p, buf.data_ptr(), stream)
                sync()
                steady.append((once(p) + once(p)) / 2)
            r.set_tile_scheduling(1)
            mv = [once(p) for p in ps_moved]
            mv_steady = []
            for p in ps_moved[::4]:
                for _ in range(3):
                    r.render_volume_device(p, buf.data_ptr(), stream)
                sync()
                mv_steady.append(once(p))
            fm = [sum(x) / len(x) for x in first]
            out[f"{mode}_{sname}"] = {
                "protocol_first_ms": round(sum(fm) / 8, 4), "protocol_first_ms_max": round(max(max(x) for x in first), 4),
                "protocol_first_per_view_ms": [round(x, 4) for x in fm],
                "protocol_steady_ms": round(sum(steady) / 8, 4), "protocol_steady_per_view_ms": [round(x, 4) for x in steady],
                "protocol_first_over_steady": round(sum(fm) / sum(steady), 4),
                "moving_ms": round(sum(mv) / len(mv), 4), "moving_ms_max": round(max(mv), 4),
                "moving_steady_ms": round(sum(mv_steady) / len(mv_steady), 4),
                "moving_over_steady": round((sum(mv) / len(mv)) / (sum(mv_steady) / len(mv_steady)), 4)}
    return out


def roofline(alg_bytes, kernel_ms):
    ach = alg_bytes / (kernel_ms * 1e-3) / 1e9
    return {"bound": "hbm", "algorithmic_bytes_per_launch": int(alg_bytes), "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 5)}


def run_config(vr, name, device_index=0, layout=None, modes=("nooptims", "default"), sampling=None, reps=3):
    """One BASELINE configuration in a context of its own: shell volume generated in HBM, the reference's 8 benchmark views,
    TRILINEAR; kernel ms (mean / max over the timed launches) per mode, and the full march's roofline over X*Y*Z*bpv + 4*W*H."""
    n, W, H, bpv = CONFIGS[name]
    r = vr.HipRenderer(device_index)
    try:
        if layout is not None:
            r.set_layout(layout)
        t0 = time.perf_counter()
        r.generate_volume("shell", n, seed=1, bytes_per_voxel=bpv)
        minmax = r.volume_minmax()[0]
        scene = vr.Scene().set_volume(dims=(n, n, n), minmax=minmax)
        r.set_transfer_fn(scene.tf, scene.esl)
        samp = vr.SAMPLE_TRILINEAR if sampling is None else sampling
        views = [vr.benchmark_view(W, H, i) for i in range(8)]
        buf = torch.empty((H, W, 4), dtype=torch.uint8, device=f"cuda:{device_index}")
        s = torch.cuda.Stream(torch.device("cuda", device_index))
        out = {"volume": [n, n, n], "viewport": [W, H], "bytes_per_voxel": bpv, "ray_step": float(scene.params.ray_step)}
        with torch.cuda.stream(s):
            for mode in modes:
                set_mode(scene, mode)
                ps = [scene.frame_params(v, samp) for v in views]
                per_view, mean, worst = time_views(r, ps, buf, s.cuda_stream, s.synchronize, reps=reps)
                e = {"kernel_ms": round(mean, 4), "kernel_ms_max": round(worst, 4), "per_view_kernel_ms": [round(x, 4) for x in per_view],
                     "Mrays_per_s": round(W * H / (mean * 1e-3) / 1e6, 1)}
                if mode == "nooptims":
                    e["roofline"] = roofline(n ** 3 * bpv + 4 * W * H, mean)
                out[mode] = e
        info = r.volume_info()
        out["hbm_bytes"] = int(info.linear_bytes + info.bricked_bytes)
        out["copies_built"] = [vr.COPY_NAMES[k] for k in range(vr.COPY_KINDS) if (info.copies >> k) & 1]
        out["setup_s"] = round(time.perf_counter() - t0, 2)
        return out
    finally:
        r.close()


def host_buffer_leg(vr, r, scene, views, sampling):
    """The reference's timed region (VolR.cpp:109-111 around GPURenderer1.cu:107-110): clear + kernel + copy-out into a HOST buffer,
    through vr_hip_render; mean and max over the 8 views like Profiler.cpp:69-72.  PCIe-inclusive: never the headline value."""
    W, H = views[0].width, views[0].height
    r.set_window_buffer(W, H)
    totals = []
    for v in views:
        p = scene.frame_params(v, sampling)
        r.render_volume(p)                          # warm (copy / order already built by the timed region)
        t0 = time.perf_counter()
        r.render_volume(p)
        totals.append((time.perf_counter() - t0) * 1e3)
    return {"ms_mean": round(sum(totals) / len(totals), 4), "ms_max": round(max(totals), 4),
            "what": "vr_hip_render: fused clear + kernel + D2H of W*H*4 bytes into pageable host memory, wall clock per call"}


def scale_model(vr, dmod, r, scene, views, W, H, sampling, buf, stream, sync, ranks=(2, 4, 8), modes=("nooptims", "default")):
    """What ONE GPU can say about N: every rank's band set (the same interleaved bands the N-rank run uses) rendered alone, per
    view; predicted efficiency = mean over views of t_1 / (N * max_r t_r) — load balance only, no gather, no host overhead."""
    out = {"simulation": "ONE GPU renders what each of N ranks would (no second device, no gather): a prediction, not a measurement at N",
           "which_number_is_ms_per_frame": "latency_ms_per_frame = one frame, nothing else in flight (interactive viewer: north_star's ms/frame); "
                                           "throughput_ms_per_frame_3_in_flight = three frames of a rank in flight (+2 frames of latency); the older keys "
                                           "predicted_ms_per_frame / pipelined_ms_per_frame hold the same two figures",
           "what": "each rank's bands rendered alone on this GPU (kernel ms, hipEvents); efficiency = mean_v t1(v) / (N * max_r t_r(v)); "
                   "load balance only — the gather (16 MiB / N per rank over xGMI) and host overhead are not in it.  pipelined_*: rank 0's bands of "
                   "consecutive frames rendered CONCURRENTLY on three streams (three frames in flight), as the N-rank run does (wall ms per frame over 48 frames): a rank's share "
                   "of a frame fills the chip only briefly, two of them side by side keep it busy"}
    three = [torch.cuda.Stream() for _ in range(3)]
    bufs3 = [buf, torch.empty_like(buf), torch.empty_like(buf)]
    for mode in modes:
        set_mode(scene, mode)
        whole = [vr.whole_frame(scene.frame_params(v, sampling)) for v in views]
        t1, _, _ = time_views(r, whole, buf, stream, sync, reps=2)
        res = {"t1_per_view_ms": [round(x, 4) for x in t1]}
        for n in ranks:
            band_rows = dmod.default_band_rows(H, n)
            per_rank_view = []
            for rank in range(n):
                split = dmod.FrameSplit(W, H, n, rank, band_rows)
                ps = [split.apply(scene.frame_params(v, sampling)) for v in views]
                per_view, _, _ = time_views(r, ps, buf, stream, sync, reps=2)
                per_rank_view.append(per_view)
            per_rank = [sum(x) / len(x) for x in per_rank_view]
            eff = [t1[v] / (n * max(per_rank_view[k][v] for k in range(n))) for v in range(len(views))]
            split0 = dmod.FrameSplit(W, H, n, 0, band_rows)
            ps0 = [split0.apply(scene.frame_params(v, sampling)) for v in views]
            piped = None
            for attempt in range(2):                         # the first pass warms the second stream's queue up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(48):
                    r.render_volume_device(ps0[i % len(ps0)], bufs3[i % 3].data_ptr(), three[i % 3].cuda_stream)
                torch.cuda.synchronize()
                piped = (time.perf_counter() - t0) / 48 * 1e3
            # the LATENCY of one frame at N ranks = the slowest rank's band set rendered alone (what an interactive viewer waits for, before
            # the gather); finer bands even the mix of long and short rays a rank gets: 64- and 32-row bands beside the default
            finer = {}
            for rows in (64, 32):
                if rows >= band_rows or H // rows < 2 * n:
                    continue
                worst = [0.0] * len(views)
                for rank in range(n):
                    sp = dmod.FrameSplit(W, H, n, rank, rows)
                    per_view, _, _ = time_views(r, [sp.apply(scene.frame_params(v, sampling)) for v in views], buf, stream, sync, reps=2)
                    worst = [max(a_, b_) for a_, b_ in zip(worst, per_view)]
                finer[str(rows)] = round(sum(worst) / len(worst), 4)
            res[f"n{n}"] = {"band_rows": band_rows, "per_rank_kernel_ms": [round(x, 4) for x in per_rank],
                            "latency_ms_per_frame": round(sum(max(per_rank_view[k][v] for k in range(n)) for v in range(len(views))) / len(views), 4),
                            "latency_efficiency": round(sum(eff) / len(eff), 4),
                            "latency_ms_per_frame_by_band_rows": finer,
                            "throughput_ms_per_frame_3_in_flight": round(piped, 4), "throughput_efficiency_3_in_flight": round(sum(t1) / len(t1) / (n * piped), 4),
                            "pipelined_ms_per_frame": round(piped, 4), "predicted_efficiency_pipelined": round(sum(t1) / len(t1) / (n * piped), 4),
                            "max_over_mean": round(max(per_rank) / (sum(per_rank) / n), 4),
                            "predicted_efficiency": round(sum(eff) / len(eff), 4),
                            "predicted_ms_per_frame": round(sum(max(per_rank_view[k][v] for k in range(n)) for v in range(len(views))) / len(views), 4)}
        out[mode] = res
    return out


def multi_overhead(vr, device_index=0, n_volume=256, W=2048, H=2048, lists=(1, 2, 4, 8), frames=32):
    """Host overhead of the single-process several-device path (vr_hip_multi_*) measured on ONE GPU: device lists that repeat
    device 0.  Small volume (256^3: kernel time is a fraction of a millisecond), the headline viewport (the band copies and the
    assemble kernel move the real 16 MiB): wall ms per frame, synchronous call against the two-frames-in-flight pipeline."""
    out = {"what": f"vr_hip_multi_* with device lists [0]*N on one GPU, shell {n_volume}^3 @ {W}x{H}, full march TRILINEAR, view 1; wall ms per frame "
                   f"over {frames} frames: synchronous calls / async pipeline (3 frames in flight); kernel_ms_sum = sum of the N band kernels of a frame"}
    r0 = vr.HipRenderer(device_index)                       # the scene (TF, ESL, ray step) of this volume, by the feeders
    try:
        r0.generate_volume("shell", n_volume, seed=1)
        scene = vr.Scene().set_volume(dims=(n_volume,) * 3, minmax=r0.volume_minmax()[0])
    finally:
        r0.close()
    set_mode(scene, "nooptims")
    p = scene.frame_params(vr.benchmark_view(W, H, 1), vr.SAMPLE_TRILINEAR)
    for n in lists:
        m = vr.MultiRenderer([device_index] * n)
        try:
            m.set_window_buffer(W, H)
            m.generate_volume("shell", n_volume, seed=1)
            m.set_transfer_fn(scene.tf, scene.esl)
            bufs = [torch.empty((H, W, 4), dtype=torch.uint8, device=f"cuda:{device_index}") for _ in range(3)]
            torch.cuda.synchronize()
            for _ in range(3):
                m.render_volume_device(p, bufs[0].data_ptr())
            t0 = time.perf_counter()
            for i in range(frames):
                m.render_volume_device(p, bufs[i % 3].data_ptr())
            sync_ms = (time.perf_counter() - t0) / frames * 1e3
            per, _ = m.timing()
            t0 = time.perf_counter()
            for i in range(frames):
                m.render_volume_device_async(p, bufs[i % 3].data_ptr())
            m.sync()
            async_ms = (time.perf_counter() - t0) / frames * 1e3
            out[f"n{n}"] = {"transport": m.transport, "sync_ms_per_frame": round(sync_ms, 4), "pipelined_ms_per_frame": round(async_ms, 4),
                            "kernel_ms_sum": round(sum(per), 4)}
        finally:
            m.close()
    return out


def concurrent_frames_leg(vr, r, params, buf, frames=48):
    """Throughput with TWO whole frames rendering concurrently (one stream each) against back to back on one stream, same views, wall ms
    per frame: what an N = 1 caller with frames in flight can have.  The headline keeps one stream — there a launch's duration IS the kernel's."""
    pair = [torch.cuda.Stream(), torch.cuda.Stream()]
    bufs = [buf, torch.empty_like(buf)]
    res = {}
    for label, nstreams in (("one_stream", 1), ("two_streams", 2), ("one_stream_again", 1), ("two_streams_again", 2)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(frames):
            s = i % nstreams
            r.render_volume_device(params[i % len(params)], bufs[s].data_ptr(), pair[s].cuda_stream)
        torch.cuda.synchronize()
        res[label] = round((time.perf_counter() - t0) / frames * 1e3, 4)
    return {"one_stream_ms_per_frame": min(res["one_stream"], res["one_stream_again"]), "two_streams_ms_per_frame": min(res["two_streams"], res["two_streams_again"]),
            "what": "wall ms per frame over %d frames of the timed workload (8 views cycled): back to back on one stream / alternating between two streams" % frames}
