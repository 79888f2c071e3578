#!/usr/bin/env python3
"""bench.py — Mrays/s + ms/frame of the ray-march hot path on BASELINE.json's headline configuration:
synthetic "shell" 1024^3 u8 volume (1 GiB, resident in HBM) rendered at 2048 x 2048, one frame per step, cycling
through the reference's 8 benchmark views (2 projections x 4 poses at distance 2, VolR.cpp:225-253).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = one whole frame: every rank (one process per GPU) renders its interleaved 16-row bands of the frame with the
hand-written gfx950 kernel (replicated volume), then the RGBA8 bands are gathered on rank 0 over RCCL (xGMI).  The work
per frame is fixed as N grows => "scaling": "strong".  Default mode is the reference's "no optims" configuration
(VolR.cpp:283-287: empty-space leaping off, early-ray-termination threshold 1.0, light on) — the full march, the only
mode whose algorithmic bytes are view independent (260 B/ray) — in TRILINEAR sampling (GPURenderer4
semantics, the heavier mode).  The reference's default mode (ESL + ERT) and NEAREST sampling are timed as extras.

Rank 0 prints ONE JSON line.  With --gpus 1 it also times the reference's own CPURenderer (oracle/_ref, built from the
reference sources in the build container) on a bounded sample of the same workload on one host core.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--volume", type=int, default=1024, help="cube edge of the synthetic shell volume")
    ap.add_argument("--viewport", type=int, default=2048)
    ap.add_argument("--mode", choices=("nooptims", "default"), default="nooptims")
    ap.add_argument("--sampling", choices=("trilinear", "nearest"), default="trilinear")
    ap.add_argument("--band-rows", type=int, default=0, help="rows per interleaved band (0 = 16, or the whole frame at N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--cpu-band-rows", type=int, default=96, help="rows per view of the CPU-baseline sample")
    return ap.parse_args()


def cpu_baseline(vr, renderer, scene, views, n, width, height, band_rows):
    """The reference's CPURenderer::render_volume (compiled from the reference sources into oracle/_ref) on one host core,
    on a bounded sample: one band of `band_rows` rows per benchmark view (8 bands spread over the frame height), same
    volume / TF / mode; NEAREST sampling because that is what the reference's CPU renderer does."""
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libvolr_ref.so")
    vox = renderer.download_volume()
    if os.path.exists(ref_so):
        L = C.CDLL(ref_so)
        devnull = os.open(os.devnull, os.O_WRONLY)      # the reference's Logger prints to stdout: keep our JSON line alone
        saved = os.dup(1)
        sys.stdout.flush()
        os.dup2(devnull, 1)
        try:
            L.volr_ref_init()
            assert L.volr_ref_set_volume(vox.ctypes.data_as(C.POINTER(C.c_ubyte)), n, n, n) == 0
            L.volr_ref_set_params(C.c_float(scene.params.ray_step), C.c_float(scene.params.ray_threshold),
                                  C.c_float(scene.params.light_kd), int(scene.params.esl))
            rays, secs = 0, 0.0
            out = np.zeros((band_rows, width, 4), np.uint8)
            for i, v in enumerate(views):
                y0 = int((i + 0.5) * height / len(views)) - band_rows // 2
                shift = float(y0 - height // 2 + band_rows // 2)     # get_ray centres rows on dims.y/2 (ViewBase.h:26-33)
                o = np.array(list(v.origin), np.float32)
                d = np.array(list(v.direction), np.float32)
                up = np.array(list(v.up_plane), np.float32)
                if v.perspective:
                    d = d + up * np.float32(shift)
                else:
                    o = o + up * np.float32(shift)
                v15 = np.concatenate([o, d, np.array(list(v.right_plane), np.float32), up,
                                      np.array(list(v.light_pos), np.float32)]).astype(np.float32)
                s = C.c_double()
                rc = L.volr_ref_render(width, band_rows, v15.ctypes.data_as(C.POINTER(C.c_float)), int(v.perspective),
                                       out.ctypes.data_as(C.POINTER(C.c_ubyte)), C.byref(s))
                assert rc == 0
                rays += width * band_rows
                secs += s.value
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(devnull)
            os.close(saved)
        kind = "reference"
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import Oracle                       # cpu_baseline leg: the oracle only as the thing timed beside us
        oracle = Oracle()
        rays, secs = 0, 0.0
        for i, v in enumerate(views):
            p = scene.frame_params(v, vr.SAMPLE_NEAREST)
            nb = height // band_rows
            p.out_rows, p.band_rows, p.band_stride = band_rows, band_rows, nb
            p.band_first = min(nb - 1, int((i + 0.5) * nb / len(views)))
            t0 = time.perf_counter()
            oracle.render(p, vox, scene.tf, scene.esl, threads=1)
            secs += time.perf_counter() - t0
            rays += width * band_rows
        kind = "port"
    return {"value": round(rays / secs / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": kind,
            "sample": f"{len(views)} bands x {band_rows} rows (one per benchmark view) of the {width}x{height} frame = "
                      f"{rays} rays, {secs:.1f} s; NEAREST sampling (CPURenderer.cpp semantics), same volume/TF/mode",
            "ms_per_frame_extrapolated": round(secs / rays * width * height * 1e3, 1)}


def recorded_traffic(key):
    """HBM bytes per launch from the PMC pass of the SAME command (profiles/r01_traffic.json, written by
    scripts/profile_bench.sh from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes).
    None when no matching profile has been recorded — PMC counters cannot be read from inside this process."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        return rec.get(key)
    except (OSError, ValueError):
        return None


def main():
    # Rank 0 must print exactly ONE line on stdout.  Native libraries in this process write there too (RCCL prints a version
    # banner on init, the reference's Logger prints on init), so file descriptor 1 is pointed at stderr for the whole run
    # and the JSON line goes to the saved descriptor at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    distributed = world > 1 or "RANK" in os.environ            # under torch.distributed.run even N=1 goes through RCCL
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)      # "nccl" IS RCCL on ROCm

    vr = importlib.import_module("volume-rendering_amd")
    dmod = importlib.import_module("volume-rendering_amd.distributed")
    r = vr.HipRenderer(local_rank)
    n, W, H = a.volume, a.viewport, a.viewport

    # -- scene: volume generated in HBM, ESL min/max by the streaming reduction, TF/ESL/ray_step by the host mirror
    r.generate_volume("shell", n, seed=1)
    minmax, _, _, minmax_ms = r.volume_minmax()
    _, minmax_ms = r.volume_minmax()[0], r.volume_minmax()[3]       # second run: warm clocks
    scene = vr.Scene().set_volume(dims=(n, n, n), minmax=minmax)
    if a.mode == "nooptims":
        scene.set_modes(esl=False, ray_threshold=1.0)               # VolR.cpp:285-286
    r.set_transfer_fn(scene.tf, scene.esl)
    sampling = vr.SAMPLE_TRILINEAR if a.sampling == "trilinear" else vr.SAMPLE_NEAREST
    views = [vr.benchmark_view(W, H, i) for i in range(8)]

    band_rows = a.band_rows or dmod.default_band_rows(H, world)
    split = dmod.FrameSplit(W, H, world, rank, band_rows)
    # two frames in flight: frame i+1 is rendered while the bands of frame i travel to rank 0 on RCCL's stream
    local = [split.local_buffer(device) for _ in range(2)]
    staging = [split.staging_buffer(device) if (rank == 0 and distributed) else None for _ in range(2)]
    pending = [None, None]
    params = [split.apply(scene.frame_params(v, sampling)) for v in views]
    stream = torch.cuda.current_stream().cuda_stream

    # the assembled frame (what a display or an encoder would consume) lives on rank 0
    final = torch.empty((H, W, 4), dtype=torch.uint8, device=device) if (rank == 0 and distributed) else None

    def retire(slot):
        """Frame in `slot` has been gathered: order the current stream after the transfer and de-interleave the bands
        into the final frame (one strided copy kernel on rank 0)."""
        if pending[slot] is None:
            return
        work, finish = pending[slot]
        work.wait()
        frame = finish()
        if final is not None and frame is not None:
            final.copy_(frame)
        pending[slot] = None

    def step(i):
        slot = i & 1
        retire(slot)                                # the buffer pair of frame i-2 is free again
        r.render_volume_device(params[i % 8], local[slot].data_ptr(), stream)
        work, finish = split.gather_async(local[slot], staging[slot])
        if work is not None:
            pending[slot] = (work, finish)

    def fence():
        retire(0)
        retire(1)
        if distributed:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    fence()
    r.timing_reset()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    tm = r.timing()
    per_rank_kernel_ms = [tm.kernel_ms_sum / max(1, tm.launches)]
    if distributed:
        t = torch.tensor([elapsed, per_rank_kernel_ms[0]], dtype=torch.float64, device=device)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)                   # scaling is set by load balance — report every rank's kernel time
        per_rank_kernel_ms = [float(e[1]) for e in every]
        elapsed, kernel_ms = max(float(e[0]) for e in every), max(per_rank_kernel_ms)
    else:
        kernel_ms = per_rank_kernel_ms[0]

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        mrays = W * H / (elapsed / a.steps) / 1e6
        # ALGORITHMIC bytes per launch: compulsory HBM traffic = every voxel once + the RGBA8 framebuffer,
        # 260 B/ray at 1024^3 @ 2048^2 in the full march; one launch covers 1/world of the frame and of the voxel rows.
        alg_bytes = (n ** 3 * 1 + 4 * W * H) / world
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "Mrays/s (W*H / t_frame), 1024^3 volume @ 2048^2 viewport", "value": round(mrays, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"shell {n}^3 u8 (seed 1) @ {W}x{H}, reference's 8 benchmark views cycled, "
                                   f"mode={a.mode} ({'ESL off, threshold 1.0' if a.mode == 'nooptims' else 'ESL on, threshold 0.95'}, "
                                   f"light_kd 0.6), sampling={a.sampling}",
                       "volume": [n, n, n], "viewport": [W, H], "bytes_per_voxel": 1, "ray_step": float(scene.params.ray_step),
                       "partition": f"{world} rank(s) x interleaved {band_rows}-row bands, RCCL gather to rank 0, 2 frames in flight" if world > 1 else "single GPU, whole frame"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": recorded_traffic(f"{a.mode}_{a.sampling}_{n}_{W}_n{world}"), "kernel": "vr::raymarch_kernel",
                         "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes),
                         "per_rank_kernel_ms": [round(x, 4) for x in per_rank_kernel_ms],
                         "kernel_imbalance_max_over_mean": round(max(per_rank_kernel_ms) / (sum(per_rank_kernel_ms) / len(per_rank_kernel_ms)), 4),
                         "note": "full march is gather/VALU-issue bound, not HBM bound (SURVEY §8d 'honest ceiling')"},
            "minmax_feeder": {"kernel": "vr::minmax_kernel", "kernel_ms": round(minmax_ms, 4),
                              "achieved_GBs": round(n ** 3 / (minmax_ms * 1e-3) / 1e9, 1),
                              "frac_of_hbm_peak": round(n ** 3 / (minmax_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if not a.no_extras and world == 1:
            extras = {}
            for label, mode, samp in (("nooptims_nearest", "nooptims", vr.SAMPLE_NEAREST),
                                      ("default_trilinear", "default", vr.SAMPLE_TRILINEAR),
                                      ("default_nearest", "default", vr.SAMPLE_NEAREST)):
                if mode == "nooptims":
                    scene.set_modes(esl=False, ray_threshold=1.0)
                else:
                    scene.set_modes(esl=True, ray_threshold=0.95)
                ps = [split.apply(scene.frame_params(v, samp)) for v in views]
                for p in ps:
                    r.render_volume_device(p, local[0].data_ptr(), stream)
                torch.cuda.synchronize()
                r.timing_reset()
                t1 = time.perf_counter()
                for p in ps:
                    r.render_volume_device(p, local[0].data_ptr(), stream)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / 8
                extras[label] = {"ms_per_frame": round(dt * 1e3, 4), "Mrays_per_s": round(W * H / dt / 1e6, 1)}
            out["extras"] = extras
            if a.mode == "nooptims":
                scene.set_modes(esl=False, ray_threshold=1.0)
            else:
                scene.set_modes(esl=True, ray_threshold=0.95)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(vr, r, scene, views, n, W, H, a.cpu_band_rows)
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
