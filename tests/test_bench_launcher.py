"""bench.py started the way the driver starts it for N > 1 — plain `python bench.py --gpus N`, no torch.distributed
environment: the script must launch its own ranks, relay rank 0's single JSON line and exit 0.  Run here on CPU in
--dry-run mode (gloo, nothing is rendered: every rank fills its bands with a row pattern and rank 0 checks the assembled
frame), so launcher, rendezvous, band split, pipelined gather and JSON plumbing are the real code."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600, env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return p.returncode, lines, p.stderr


@pytest.mark.parametrize("gpus,viewport,band", [(2, 128, 0), (3, 200, 8)])
def test_bench_launches_its_own_ranks(gpus, viewport, band):
    rc, lines, err = _run(["--gpus", str(gpus), "--dry-run", "--steps", "4", "--warmup", "2", "--viewport", str(viewport),
                           "--band-rows", str(band)])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines                     # exactly ONE line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == gpus and out["steps"] == 4 and out["warmup"] == 2
    assert out["dry_run"] is True and out["frame_check"] == "ok"
    assert out["scaling"] == "strong" and out["higher_is_better"] is True
    # north_star's "Mrays/s + ms/frame": throughput with the frames of a rank in flight AND the latency of one synchronous frame
    assert out["frames_in_flight"] == 3 and out["ms_per_frame_latency"] > 0


def test_frames_in_flight_is_a_parameter():
    """--frames-in-flight 1 = the interactive configuration: every frame is finished before the next starts."""
    rc, lines, err = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--viewport", "96", "--frames-in-flight", "1"])
    assert rc == 0, err[-2000:]
    out = json.loads(lines[0])
    assert out["frames_in_flight"] == 1 and out["frame_check"] == "ok" and "1 frames in flight" in out["config"]["partition"]


def test_bench_rejects_mismatched_world():
    env_rc, lines, err = None, None, None
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 2 and p.stdout.strip() == ""
