#!/usr/bin/env python3
"""bench.py — Mrays/s + ms/frame of the ray-march hot path on BASELINE.json's headline configuration:
synthetic "shell" 1024^3 u8 volume (1 GiB, resident in HBM) rendered at 2048 x 2048, one frame per step, cycling
through the reference's 8 benchmark views (2 projections x 4 poses at distance 2, VolR.cpp:225-253).

    python bench.py [--gpus N --steps K --warmup W] [--config c2|c3|c4|c5] [--simulate-ranks 2,4,8]

--config selects another BASELINE.json configuration for the timed region (c2 256^3 @ 1024^2, c3 512^3 @ 1920x1080, c4 = the default
headline, c5 2048^3 uint16 @ 4096^2); at N = 1 the default run also reports c2 / c3 / c5, the linear layout on the headline workload,
the reference's own host-buffer timed region and the N-rank load-balance model under `extras` / `scale_model`.

With N > 1 and no torch.distributed environment the script starts N ranks of itself (one per GPU) through
`python -m torch.distributed.run` and relays rank 0's JSON line; launched under torch.distributed.run directly
(RANK / LOCAL_RANK / WORLD_SIZE set) it runs as one of the ranks.  The parent process never touches a GPU.

A step = one whole frame: every rank renders its interleaved bands of the frame with the hand-written gfx950 kernel
(replicated volume), then the RGBA8 bands are gathered on rank 0 over RCCL (xGMI).  The work per frame is fixed as N
grows => "scaling": "strong".  Default mode is the reference's "no optims" configuration (VolR.cpp:283-287:
empty-space leaping off, early-ray-termination threshold 1.0, light on) — the full march, the only mode whose
algorithmic bytes are view independent (260 B/ray) — in TRILINEAR sampling (GPURenderer4 semantics, the
heavier mode).  The reference-pinned NEAREST sampling, the reference's default mode (ESL + ERT) and its ERT-only mode
(VolR.cpp:288-290) are timed as extras, the latter two with a roofline over the bytes their sample sets touch.

Rank 0 prints ONE JSON line.  With --gpus 1 it also times the reference's own CPURenderer (oracle/_ref, built from the
reference sources in the build container) on a bounded sample of the same workload on one host core, and the OpenMP
leg of the CPU restatement on all host cores.

--dry-run (CPU, gloo): no rendering — every rank fills its bands with a row pattern and the launcher / rendezvous /
band split / pipelined gather / JSON plumbing run exactly as in the real thing (tests/test_bench_launcher.py).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", choices=("c2", "c3", "c4", "c5"), default="c4", help="BASELINE.json configuration (c4 = the headline)")
    ap.add_argument("--volume", type=int, default=0, help="cube edge of the synthetic shell volume (default: the configuration's)")
    ap.add_argument("--viewport", type=int, default=0, help="viewport width (and height, unless --height is given)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--bytes-per-voxel", type=int, default=0, choices=(0, 1, 2))
    ap.add_argument("--simulate-ranks", default="2,4,8", help="N = 1: rank counts of the load-balance model (scale_model); '' switches it off")
    ap.add_argument("--mode", choices=("nooptims", "default", "ertonly"), default="nooptims")
    ap.add_argument("--sampling", choices=("trilinear", "nearest"), default="trilinear")
    ap.add_argument("--band-rows", type=int, default=0, help="rows per interleaved band (0 = automatic)")
    ap.add_argument("--frames-in-flight", type=int, default=0, choices=(0, 1, 2, 3),
                    help="frames a rank keeps in flight: 1 = interactive (every frame is finished before the next starts: ms_per_step IS the frame latency), "
                         "3 = throughput (default for N >= 2, each on a stream of its own), 0 = automatic (N = 1: two on one stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--extras", default="modes,first,host,scale,configs,linear,multi,builders",
                    help="N = 1: which extra legs run after the timed region (other modes, host-buffer region, rank simulation, other configs, linear layout, several-device host overhead)")
    ap.add_argument("--cpu-band-rows", type=int, default=96, help="rows per view of the 1-core CPU-baseline sample")
    ap.add_argument("--dry-run", action="store_true", help="CPU / gloo rehearsal of the multi-rank plumbing, nothing is rendered")
    ap.add_argument("--force-launcher", action="store_true", help="start the ranks as child processes even for --gpus 1")
    return ap.parse_args()


# ---- launcher ------------------------------------------------------------------------------------------------------------

def launch_ranks(a):
    """Parent of an N-rank run: starts `python -m torch.distributed.run ... bench.py <same flags>` and relays rank 0's
    JSON line and the exit code.  Nothing in this process initialises HIP."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = [x for x in sys.argv[1:] if x != "--force-launcher"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        out = out.strip()
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 3
    return rc


# ---- CPU baselines (rank 0, N = 1 only) ----------------------------------------------------------------------------------

def cpu_baseline(vr, renderer, scene, views, n, width, height, band_rows):
    """The reference's CPURenderer::render_volume (compiled from the reference sources into oracle/_ref) on ONE host core
    — the reference's loop is serial, CPURenderer.cpp:48-51 — on a bounded sample: one band of `band_rows` rows per
    benchmark view (8 bands spread over the frame height), same volume / TF / mode; NEAREST sampling because that is what
    the reference's CPU renderer does.  Beside it, `all_cores`: the OpenMP leg of the CPU restatement (oracle/vr_oracle.c,
    rows in parallel) on every host core, on a sample four times as large."""
    import ctypes as C
    import numpy as np

    ref_so = os.path.join(ROOT, "oracle", "_ref", "libvolr_ref.so")
    vox = renderer.download_volume()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import Oracle                       # cpu_baseline leg: the oracle only as the thing timed beside us
    oracle = Oracle()

    def port_leg(rows, threads):
        rays, secs = 0, 0.0
        for i, v in enumerate(views):
            p = scene.frame_params(v, vr.SAMPLE_NEAREST)
            nb = height // rows
            p.out_rows, p.band_rows, p.band_stride = rows, rows, nb
            p.band_first = min(nb - 1, int((i + 0.5) * nb / len(views)))
            t0 = time.perf_counter()
            oracle.render(p, vox, scene.tf, scene.esl, threads=threads)
            secs += time.perf_counter() - t0
            rays += width * rows
        return rays, secs

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
        rays, secs = port_leg(band_rows, 1)
        kind = "port"
    cores = usable_cores()
    all_rows = min(height // len(views), 4 * band_rows)
    a_rays, a_secs = port_leg(all_rows, cores)
    return {"value": round(rays / secs / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": kind,
            "sample": f"{len(views)} bands x {band_rows} rows (one per benchmark view) of the {width}x{height} frame = "
                      f"{rays} rays, {secs:.1f} s; NEAREST sampling (CPURenderer.cpp semantics), same volume/TF/mode",
            "ms_per_frame_extrapolated": round(secs / rays * width * height * 1e3, 1),
            "all_cores": {"value": round(a_rays / a_secs / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                          "sample": f"OpenMP over rows (oracle/vr_oracle.c), {len(views)} bands x {all_rows} rows = {a_rays} rays, "
                                    f"{a_secs:.1f} s", "ms_per_frame_extrapolated": round(a_secs / a_rays * width * height * 1e3, 1)}}


def usable_cores():
    """Host cores this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a share of its
    cores to every tenant; running the OpenMP leg on every core the mask shows would only oversubscribe that share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            quota, period = parse(open(path).read())
            if quota != "max" and int(quota) > 0:
                n = min(n, max(1, int(quota) // int(period)))
            break
        except (OSError, ValueError):
            continue
    return n


def recorded(path, key):
    try:
        with open(os.path.join(ROOT, path)) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


# ---- one rank ------------------------------------------------------------------------------------------------------------

def run_rank(a):
    # Rank 0 must print exactly ONE line on stdout.  Native libraries in this process write there too (RCCL prints a version
    # banner on init, the reference's Logger prints on init), so file descriptor 1 is pointed at stderr for the whole run
    # and the JSON line goes to the saved descriptor at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # The HIP runtime gives a process 4 hardware queues per device by default and lets further streams SHARE them (a stream takes its
    # queue at first use).  A rank uses the slot streams below, torch's default stream and the RCCL stream: with 4 queues two slot
    # streams can land on one queue and their frames serialise (measured, scripts/gpu_r03_x.sh: two slot streams beside two other
    # busy streams 0.565 ms per frame = no overlap at all, against 0.356 with queues of their own).  Read by the runtime when it
    # initialises, so it is set before torch is imported.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import importlib
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    distributed = world > 1 or "RANK" in os.environ            # under torch.distributed.run even N=1 goes through the collective
    dmod = importlib.import_module("volume-rendering_amd.distributed")
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bench_extras as bx
    cn, cW, cH, cbpv = bx.CONFIGS[a.config]
    n = a.volume or cn
    W = a.viewport or cW
    H = a.height or (a.viewport or cH)
    bpv = a.bytes_per_voxel or cbpv
    band_rows = a.band_rows or dmod.default_band_rows(H, world)
    split = dmod.FrameSplit(W, H, world, rank, band_rows)

    if a.dry_run:
        device = torch.device("cpu")
        if distributed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        vr = r = scene = None
        views = list(range(8))
        params = [None] * 8
        rows = torch.arange(split.local_rows)
        frame_rows = ((rows // band_rows) * world + rank) * band_rows + rows % band_rows      # frame row of every local row
        stream_ctx, stream = None, None
    else:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        if distributed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)      # "nccl" IS RCCL on ROCm
        vr = importlib.import_module("volume-rendering_amd")
        r = vr.HipRenderer(local_rank)
        # -- scene: volume generated in HBM, ESL min/max by the streaming reduction, TF/ESL/ray_step by the host mirror
        t_setup = time.perf_counter()
        r.generate_volume("shell", n, seed=1, bytes_per_voxel=bpv)
        minmax = r.volume_minmax()[0]
        minmax_ms = r.volume_minmax()[3]                        # second run: warm clocks
        scene = vr.Scene().set_volume(dims=(n, n, n), minmax=minmax)
        set_mode(scene, a.mode)
        r.set_transfer_fn(scene.tf, scene.esl)
        sampling = vr.SAMPLE_TRILINEAR if a.sampling == "trilinear" else vr.SAMPLE_NEAREST
        views = [vr.benchmark_view(W, H, i) for i in range(8)]
        params = [split.apply(scene.frame_params(v, sampling)) for v in views]
        # Brick copies are built by the first frame that reads them: the warm-up frames (one per view) do that, outside the timed
        # region; `set_volume` in the JSON line reports what the upload / generation and every copy cost.
        # Rendering, the RCCL gather and the de-interleave copy are all ordered through ONE stream torch knows about: the
        # kernel is launched on it (the C ABI takes the raw hipStream_t), dist.gather() makes RCCL's stream wait for it, and
        # work.wait() makes it wait for RCCL before the buffer pair is rendered into again.
        render_stream = torch.cuda.Stream(device)
        stream_ctx, stream = torch.cuda.stream(render_stream), render_stream.cuda_stream
        # N >= 2: a rank's share of the frame no longer fills the chip for long (at N = 8 it is exactly one load of 8192 waves: the
        # launch lasts as long as its longest wave while the mean wave is much shorter), so a rank keeps THREE frames in flight and
        # renders them concurrently, each slot on a stream of its own — the next frames' workgroups fill the tail of the one before.
        # Measured on one GPU with the band sets of an N-rank run (scripts/overlap_probe.py, scale_model.pipelined_*), ms per frame with
        # 1 / 2 / 3 / 4 streams: N = 2: 1.40 / 1.18 / 1.16 / 1.20, N = 4: 0.86 / 0.60 / 0.59 / 0.65, N = 8: 0.57 / 0.36 / 0.29 / 0.35.
        # At N = 1 one frame fills the chip (2.39 -> 2.29 with two) and the two slots share one stream, so that kernel_ms is the
        # kernel's own duration.
        many_streams = world >= 2 or os.environ.get("VR_BENCH_TWO_STREAMS") == "1"     # the env switch rehearses the pipeline on one GPU
        slot_streams = [render_stream] + [torch.cuda.Stream(device) for _ in range(2)] if many_streams else [render_stream, render_stream]
        if a.frames_in_flight:
            slot_streams = slot_streams[:a.frames_in_flight]

    import contextlib
    if a.dry_run:
        slot_streams = [None] * (a.frames_in_flight or (3 if world >= 2 else 2))
    slots = len(slot_streams)

    def on_slot(slot):
        return contextlib.nullcontext() if slot_streams[slot] is None else torch.cuda.stream(slot_streams[slot])

    # `slots` frames in flight: frame i+1 is rendered while the bands of frame i travel to rank 0 on the backend's stream
    local = [split.local_buffer(device) for _ in range(slots)]
    staging = [split.staging_buffer(device) if (rank == 0 and distributed) else None for _ in range(slots)]
    pending = [None] * slots
    # the assembled frame (what a display or an encoder would consume) lives on rank 0, one per slot
    final = [torch.empty((H, W, 4), dtype=torch.uint8, device=device) if rank == 0 else None for _ in range(slots)]

    def render(i, slot):
        if a.dry_run:
            local[slot].copy_(((frame_rows + i) % 251).to(torch.uint8).view(-1, 1, 1).expand(-1, W, 4))
        else:
            r.render_volume_device(params[i % 8], local[slot].data_ptr(), slot_streams[slot].cuda_stream)

    def retire(slot):
        """Frame in `slot` has been gathered: order the render stream after the transfer and de-interleave the bands
        into the final frame (one strided copy kernel on rank 0)."""
        if pending[slot] is None:
            return
        work, finish = pending[slot]
        if work is not None:
            work.wait()
        frame = finish()
        if final[slot] is not None and frame is not None:
            final[slot].copy_(frame)
        pending[slot] = None

    def step(i):
        slot = i % slots
        with on_slot(slot):                         # render, gather and de-interleave of a slot are ordered through the slot's stream
            retire(slot)                            # the buffers of frame i - slots are free again
            render(i, slot)
            pending[slot] = split.gather_async(local[slot], staging[slot])

    def fence():
        for slot in range(slots):
            with on_slot(slot):
                retire(slot)
        if distributed:
            dist.barrier(device_ids=None if a.dry_run else [local_rank])
        if not a.dry_run:
            torch.cuda.synchronize()

    def timed_region():
        # set-up, not a step: ONE frame per view, whatever --warmup is — it builds the brick copy the view reads (copies are built by the
        # first frame that wants them; `set_volume.copy_build_ms` lists what each cost) and touches it once.  Nothing else is learned
        # from earlier frames in the headline's mode: the full march keeps no history (the per-tile choice between the two run copies is
        # analytic since round 4); in the modes with leaping / early termination a frame launches its tiles in the order the previous
        # finished frame of that view direction measured (`extras.first_visit` reports what a first frame costs).
        if not a.dry_run:
            for i in range(8):
                render(i, 0)
            torch.cuda.synchronize()
        for i in range(a.warmup):
            step(i)
        fence()
        if r is not None:
            r.timing_reset()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(i)
        fence()
        return time.perf_counter() - t0

    def check_frame():
        """One more frame through the N-rank path, compared on rank 0 with rank 0's own whole-frame render of that view."""
        i = a.warmup + a.steps
        step(i)
        fence()
        if rank != 0:
            return None
        if a.dry_run:
            want = ((torch.arange(H) + i) % 251).to(torch.uint8).view(-1, 1, 1).expand(-1, W, 4)
            return "ok" if torch.equal(final[i % slots], want) else "MISMATCH"
        whole = torch.empty((H, W, 4), dtype=torch.uint8, device=device)
        r.render_volume_device(vr.whole_frame(scene.frame_params(views[i % 8], sampling)), whole.data_ptr(), stream)
        torch.cuda.synchronize()
        return "ok" if torch.equal(final[i % slots], whole) else "MISMATCH"

    def latency_region(frames=16):
        """north_star's "ms/frame" for an interactive viewer: ONE frame at a time through the same N-rank path (render, gather,
        de-interleave), finished before the next one starts — no frame hides behind another.  Outside the timed region."""
        fence()
        t0 = time.perf_counter()
        for i in range(frames):
            step(i)
            fence()
        return (time.perf_counter() - t0) / frames

    if stream_ctx is not None:
        with stream_ctx:
            elapsed = timed_region()
            tm = r.timing()
            frame_check = check_frame()
            latency = latency_region()
    else:
        elapsed = timed_region()
        tm = None
        frame_check = check_frame()
        latency = latency_region()

    per_rank_kernel_ms = [tm.kernel_ms_sum / max(1, tm.launches)] if tm is not None else [0.0]
    kernel_ms_max = float(tm.kernel_ms_max) if tm is not None else 0.0       # the longest single launch (Profiler.cpp:69-72 keeps sum and max)
    if distributed:
        t = torch.tensor([elapsed, per_rank_kernel_ms[0], kernel_ms_max, latency], dtype=torch.float64, device=device)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)                   # scaling is set by load balance — report every rank's kernel time
        per_rank_kernel_ms = [float(e[1]) for e in every]
        elapsed, kernel_ms, kernel_ms_max = max(float(e[0]) for e in every), max(per_rank_kernel_ms), max(float(e[2]) for e in every)
        latency = max(float(e[3]) for e in every)
    else:
        kernel_ms = per_rank_kernel_ms[0]

    rc = 0
    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        mrays = W * H / (elapsed / a.steps) / 1e6
        partition = (f"{world} rank(s) x interleaved {band_rows}-row bands, "
                     f"{'gloo' if a.dry_run else 'RCCL'} gather to rank 0, {slots} frames in flight"
                     f"{' rendered concurrently (one stream per slot)' if slots > 1 and slot_streams[0] is not slot_streams[1] else ''}") if distributed else "single GPU, whole frame"
        out = {
            "metric": f"Mrays/s (W*H / t_frame), {n}^3 volume @ {W}x{H} viewport" if (n, W, H) != (1024, 2048, 2048) else
                      "Mrays/s (W*H / t_frame), 1024^3 volume @ 2048^2 viewport", "value": round(mrays, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "frame_check": frame_check,
            # north_star asks for "Mrays/s + ms/frame": `value` / `ms_per_step` are THROUGHPUT with `frames_in_flight` frames of a rank in
            # flight (a renderer that feeds an encoder); `ms_per_frame_latency` is ONE frame through the same path with nothing else in
            # flight (an interactive viewer), measured over 16 synchronous frames after the timed region.  At N = 1 both are the kernel.
            "frames_in_flight": slots, "ms_per_frame_latency": round(latency * 1e3, 4), "Mrays_per_s_latency": round(W * H / latency / 1e6, 2) if latency > 0 else None,
        }
        if frame_check != "ok":
            rc = 4
        if a.dry_run:
            out.update({"dry_run": True, "value": 0.0,
                        "config": {"workload": "DRY RUN: no rendering, row-pattern bands through the real split / gather / launcher code",
                                   "viewport": [W, H], "partition": partition}})
        else:
            mode_txt = {"nooptims": "ESL off, threshold 1.0", "default": "ESL on, threshold 0.95", "ertonly": "ESL off, threshold 0.95"}[a.mode]
            out["config"] = {"workload": f"BASELINE {a.config}: shell {n}^3 u{8 * bpv} (seed 1) @ {W}x{H}, reference's 8 benchmark views cycled, "
                                         f"mode={a.mode} ({mode_txt}, light_kd 0.6), sampling={a.sampling}; repeated-view regime (each view was "
                                         f"rendered before: copies resident) — extras.first_visit holds the one-frame-per-view / moving-camera figures",
                             "volume": [n, n, n], "viewport": [W, H], "bytes_per_voxel": bpv, "ray_step": float(scene.params.ray_step),
                             "partition": partition}
            # ALGORITHMIC bytes per launch: compulsory HBM traffic = every voxel once + the RGBA8 framebuffer,
            # 260 B/ray at 1024^3 @ 2048^2 in the full march; one launch covers 1/world of the frame and of the voxel rows.
            key = f"{a.mode}_{a.sampling}_{n}_{W}"
            if a.mode == "nooptims":
                alg_bytes = (n ** 3 * bpv + 4 * W * H) / world
            else:
                touched = recorded("tests/golden/vtouched.json", key)
                alg_bytes = ((touched["mean_bytes"] if touched else n ** 3) + 4 * W * H) / world
            achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
            traffic = recorded("profiles/r04_traffic.json", f"{key}_n{world}") or recorded("profiles/r03_traffic.json", f"{key}_n{world}") or recorded("profiles/r02_traffic.json", f"{key}_n{world}") or {}
            out["roofline"] = {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic.get("bytes_per_launch"), "traffic_commit": traffic.get("commit"),
                "traffic_kernel_ms": traffic.get("kernel_ms"),
                "traffic_GBs": (round(traffic["bytes_per_launch"] / (traffic["kernel_ms"] * 1e-3) / 1e9, 1) if traffic.get("bytes_per_launch") and traffic.get("kernel_ms") else None),
                # what the kernel actually moves against the HBM peak (the copies it reads are 4 - 4.5 bytes per voxel and fetched 1 - 2.4 times)
                "traffic_frac_of_peak": (round(traffic["bytes_per_launch"] / (traffic["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic.get("bytes_per_launch") and traffic.get("kernel_ms") else None),
                # the other bound: vector-issue utilisation per benchmark view, from the counters of the same evidence run (profiles/, not live)
                "valu_busy_per_view": traffic.get("valu_busy_per_view"),
                "kernel": "vr::raymarch_kernel + vr::colmarch_kernel", "kernel_ms": round(kernel_ms, 4), "kernel_ms_max": round(kernel_ms_max, 4),
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "kernel_instantiations": "the frame's ONE launch is colmarch_kernel<sampling, axis, flips> (NEAREST: colmarch_nearest_kernel<axis, flips>) for frames without leaping of orthogonal views along a volume axis "
                                         "(column windows: three of the eight benchmark views), else raymarch_kernel<sampling,1,0,L>: L = 1 quad bricks, 2 / 3 run bricks along z / y, "
                                         "6 both run copies chosen per block of tiles (orthogonal views that are not along an axis), 4 voxel bricks (NEAREST); "
                                         "kernel_ms = hipEvent mean over ALL timed launches (the views cycle)",
                "per_rank_kernel_ms": [round(x, 4) for x in per_rank_kernel_ms],
                "kernel_ms_note": ("N >= 2: the three frames a rank has in flight render concurrently (one stream per slot), so kernel_ms is the duration of a "
                                   "launch that shares the chip with its neighbour — longer than the kernel alone; `value` (frames per second over all ranks) is the figure "
                                   "that counts, `scale_model` at N = 1 holds the per-rank kernel times without overlap") if slot_streams[0] is not slot_streams[1] else None,
                "kernel_imbalance_max_over_mean": round(max(per_rank_kernel_ms) / (sum(per_rank_kernel_ms) / len(per_rank_kernel_ms)), 4),
                "note": "the lit march is bound by vector-instruction issue as much as by memory (valu_busy_per_view 0.5 - 0.9; the column-window views stream 5.7 GB of copy per frame at 4.3 TB/s "
                        "when unlit), the run-brick views also by the gather rate of the L1 (SURVEY §8d 'honest ceiling'); every voxel is still fetched — the "
                        "exact per-wave shortcuts (transparent samples, rays whose accumulated alpha is exactly 1) skip arithmetic only"}
            out["minmax_feeder"] = {"kernel": "vr::minmax_kernel", "kernel_ms": round(minmax_ms, 4),
                                    "achieved_GBs": round(n ** 3 / (minmax_ms * 1e-3) / 1e9, 1),
                                    "frac_of_hbm_peak": round(n ** 3 / (minmax_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            if world == 1:
                # one launch per benchmark view, outside the timed region: the per-view kernel times behind the mean above
                per_view = []
                for v in views:
                    p_v = split.apply(scene.frame_params(v, sampling))
                    for _ in range(2):
                        r.render_volume_device(p_v, local[0].data_ptr(), stream)
                    render_stream.synchronize()
                    r.timing_reset()
                    for _ in range(4):
                        r.render_volume_device(p_v, local[0].data_ptr(), stream)
                    render_stream.synchronize()
                    tv = r.timing()
                    per_view.append(round(tv.kernel_ms_sum / max(1, tv.launches), 4))
                out["roofline"]["per_view_kernel_ms"] = per_view
            if world == 1:
                info = r.volume_info()
                out["set_volume"] = {"upload_or_generate_ms": round(info.upload_ms, 3),
                                     "copy_build_ms": {vr.COPY_NAMES[k]: round(info.build_ms[k], 3) for k in range(vr.COPY_KINDS) if (info.copies >> k) & 1},
                                     "hbm_bytes": int(info.linear_bytes + info.bricked_bytes),
                                     "note": "brick copies are built by the first frame that reads them (vr_hip_prepare builds them ahead of time): "
                                             "what is listed is what this run's views and modes asked for so far"}
            legs = set() if a.no_extras or world != 1 else set(x for x in a.extras.split(",") if x)
            if legs:
                out["extras"] = extras(vr, r, scene, views, split, local[0], stream, render_stream, n, W, H, bpv) if "modes" in legs else {}
                set_mode(scene, a.mode)
                if "modes" in legs:      # before the legs that create many streams: a process has 8 hardware queues, later streams share them
                    out["extras"]["two_frames_concurrent"] = bx.concurrent_frames_leg(vr, r, params, local[0])
                with torch.cuda.stream(render_stream):
                    if "first" in legs:
                        out["extras"]["first_visit"] = bx.first_visit_leg(vr, r, scene, W, H, local[0], stream, render_stream.synchronize)
                        set_mode(scene, a.mode)
                    if "host" in legs:
                        out["extras"]["host_buffer"] = bx.host_buffer_leg(vr, r, scene, views, sampling)
                        out["host_buffer_ms"] = out["extras"]["host_buffer"]["ms_mean"]
                    if a.simulate_ranks and "scale" in legs:
                        out["scale_model"] = bx.scale_model(vr, dmod, r, scene, views, W, H, sampling, local[0], stream, render_stream.synchronize,
                                                            ranks=tuple(int(x) for x in a.simulate_ranks.split(",")))
                        set_mode(scene, a.mode)
                # the other BASELINE configurations and the linear layout, each in a context of its own (this one keeps its copies)
                others = {}
                for name in ("c2", "c3", "c5", "c4"):
                    if name == a.config or "configs" not in legs:
                        continue
                    free, _ = torch.cuda.mem_get_info(device)
                    need = bx.CONFIGS[name][0] ** 3 * bx.CONFIGS[name][3] * 6
                    if need > free * 0.8:
                        others[name] = {"skipped": f"needs about {need >> 30} GiB of HBM, {free >> 30} GiB free"}
                        continue
                    others[name] = bx.run_config(vr, name, local_rank)
                if others:
                    out["extras"]["configs"] = others
                lin = bx.run_config(vr, a.config, local_rank, layout=vr.LAYOUT_LINEAR, modes=("nooptims",), reps=2) if "linear" in legs else None
                if lin:
                    out["extras"]["linear_layout"] = {"kernel_ms": lin["nooptims"]["kernel_ms"], "per_view_kernel_ms": lin["nooptims"]["per_view_kernel_ms"],
                                                  "roofline": lin["nooptims"]["roofline"], "hbm_bytes": lin["hbm_bytes"],
                                                  "what": "the same workload with vr_hip_set_layout(VR_LAYOUT_LINEAR): the reference's x-fastest array, "
                                                          "4 two-voxel loads per sample — north_star's literal layout, timed beside the product's brick copies"}
                if "multi" in legs:
                    out["extras"]["multi_overhead"] = bx.multi_overhead(vr, local_rank)
                if "builders" in legs:
                    # what set_volume costs, kernel by kernel, each with its own HBM roofline (bytes = linear array read once + copy written once;
                    # the generator: the array written once) — in a context of its own: generation, then every copy the policy has at this size
                    import copy_build_probe as cbp
                    r2 = vr.HipRenderer(local_rank)
                    try:
                        out["set_volume"]["builders"] = cbp.probe(vr, r2, n, bpv, reps=2)
                    finally:
                        r2.close()
            if world == 1 and not a.no_cpu_baseline and bpv == 1:
                out["cpu_baseline"] = cpu_baseline(vr, r, scene, views, n, W, H, a.cpu_band_rows)
                # the comparison north_star asks for, labelled — vs_baseline itself stays null (BASELINE.md has no published number)
                out["vs_cpu_baseline"] = {"ratio": round(mrays / out["cpu_baseline"]["value"], 1),
                                          "what": f"value / cpu_baseline.value: this GPU line ({a.sampling}) against the reference's CPURenderer (NEAREST) on 1 host core"}
            else:
                out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.barrier(device_ids=None if a.dry_run else [local_rank])
        dist.destroy_process_group()
    if r is not None:
        r.close()
    return rc


def set_mode(scene, mode):
    """The three configurations of the reference's optimisation benchmark, VolR.cpp:283-294."""
    if mode == "nooptims":
        scene.set_modes(esl=False, ray_threshold=1.0)
    elif mode == "ertonly":
        scene.set_modes(esl=False, ray_threshold=0.95)
    else:
        scene.set_modes(esl=True, ray_threshold=0.95)


def extras(vr, r, scene, views, split, buf, stream, render_stream, n, W, H, bpv=1):
    """Same volume and views in the other sampling mode and in the reference's two optimised configurations; for those the
    roofline uses V_touched, the bytes of the distinct 128-byte voxel lines the frame's sample set reads, counted by the
    CPU restatement in the build container (oracle/gen_vtouched.py -> tests/golden/vtouched.json)."""
    res = {}
    for label, mode, samp in (("nooptims_nearest", "nooptims", vr.SAMPLE_NEAREST),
                              ("default_trilinear", "default", vr.SAMPLE_TRILINEAR),
                              ("default_nearest", "default", vr.SAMPLE_NEAREST),
                              ("ertonly_trilinear", "ertonly", vr.SAMPLE_TRILINEAR),
                              ("ertonly_nearest", "ertonly", vr.SAMPLE_NEAREST)):
        set_mode(scene, mode)
        ps = [split.apply(scene.frame_params(v, samp)) for v in views]
        for _ in range(4):                          # builds the copies these views read; records / builds the measured-cost tile order
            for p in ps:
                r.render_volume_device(p, buf.data_ptr(), stream)
        render_stream.synchronize()
        r.timing_reset()
        t1 = time.perf_counter()
        for p in ps:
            r.render_volume_device(p, buf.data_ptr(), stream)
        render_stream.synchronize()
        dt = (time.perf_counter() - t1) / 8
        tm = r.timing()
        kernel_ms = tm.kernel_ms_sum / max(1, tm.launches)
        e = {"ms_per_frame": round(dt * 1e3, 4), "Mrays_per_s": round(W * H / dt / 1e6, 1), "kernel_ms": round(kernel_ms, 4),
             "kernel_ms_max": round(tm.kernel_ms_max, 4)}
        if mode == "nooptims":
            alg = n ** 3 * bpv + 4 * W * H
        else:
            touched = recorded("tests/golden/vtouched.json", f"{mode}_{'trilinear' if samp == vr.SAMPLE_TRILINEAR else 'nearest'}_{n}_{W}")
            alg = (touched["mean_bytes"] + 4 * W * H) if touched else None
        if alg:
            ach = alg / (kernel_ms * 1e-3) / 1e9
            e["roofline"] = {"bound": "hbm", "algorithmic_bytes_per_launch": int(alg), "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5)}
        res[label] = e
    return res


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.force_launcher):
        return launch_ranks(a)
    return run_rank(a)


if __name__ == "__main__":
    sys.exit(main())
