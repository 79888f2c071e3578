"""The headless C++ driver (volume-rendering_amd/volr_bench): whole host stack in C++ —
ModelBase::load_model (PVM/DDS decode) -> RaycasterBase -> ViewBase -> HipRenderer -> C ABI -> kernel."""
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN_DIR, ROOT

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "volume-rendering_amd", "volr_bench")


def test_single_frame_ppm_equals_reference_frame(golden, tmp_path):
    """Bucky.pvm decoded by our codec, pose (-45,-45,0) at distance 2, NEAREST == the frame the reference's CPURenderer made."""
    ppm = tmp_path / "frame.ppm"
    out = subprocess.run([EXE, "-f", os.path.join(GOLDEN_DIR, "Bucky.pvm"), "-r", "0", "-s", "256", "256",
                          "-pose", "-45", "-45", "0", "2", "-o", str(ppm)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    data = ppm.read_bytes()
    assert data.startswith(b"P6\n256 256\n255\n")
    rgb = np.frombuffer(data[len(b"P6\n256 256\n255\n"):], np.uint8).reshape(256, 256, 3)[::-1]     # PPM is top row first
    case = [c for c in golden.cases(True) if c["label"] == "bench256_view1_default"][0]
    assert np.array_equal(rgb, golden.frame(case)[..., :3])
    # perspective flag + renderer 1 run too
    out = subprocess.run([EXE, "-f", os.path.join(GOLDEN_DIR, "Bucky.pvm"), "-r", "1", "-s", "128", "128", "-persp"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "HIP MI355X trilinear: 128x128 frame" in out.stdout


def test_reference_binding_on_the_gpu():
    """oracle/_ref/binding_check (built in the container from INTEGRATION.md's binding + the reference's own Renderer.h,
    RaycasterBase, ModelBase and CPURenderer object code): the HIPRenderer subclass and the reference's CPURenderer render the
    same orthogonal and perspective frame through `renderers[i]->render_volume(buffer, raycaster)` — byte-identical."""
    exe = os.path.join(ROOT, "oracle", "_ref", "binding_check")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/binding_check was not built (needs /root/reference in the build container)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "orthogonal: 0 of 7680 pixels differ" in out.stdout and "perspective: 0 of 7680 pixels differ" in out.stdout


def test_devices_option_splits_the_frame(golden, tmp_path):
    """volr_bench -devices a,b: the reference's `renderers[id]->render_volume()` call (VolR.cpp:110) drives several per-device
    contexts through HipRenderer's device-list constructor; on this one-GPU box the list names device 0 once and twice."""
    case = [c for c in golden.cases(True) if c["label"] == "bench256_view1_default"][0]
    for devices, transport in (("0", "single"), ("0,0", "peer-copy")):
        ppm = tmp_path / f"frame_{devices.replace(',', '_')}.ppm"
        out = subprocess.run([EXE, "-f", os.path.join(GOLDEN_DIR, "Bucky.pvm"), "-r", "0", "-s", "256", "256", "-devices", devices,
                              "-pose", "-45", "-45", "0", "2", "-o", str(ppm)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert f"bands gathered by {transport}" in out.stdout
        data = ppm.read_bytes()
        rgb = np.frombuffer(data[len(b"P6\n256 256\n255\n"):], np.uint8).reshape(256, 256, 3)[::-1]
        assert np.array_equal(rgb, golden.frame(case)[..., :3]), devices


def test_benchmark_matrix_output(tmp_path):
    """-b: the reference's configuration matrix (VolR.cpp:270-321); datasets that are not shipped are skipped like the
    reference skips a missing file, the option / scale / ray-step studies run on a synthetic stand-in for Foot."""
    out = subprocess.run([EXE, "-b", "-dir", GOLDEN_DIR, "-synthetic", "96", "-s", "256", "256"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    text = out.stdout
    assert "Summary profiler report:" in text and " Rend. 0: HIP MI355X nearest" in text and " Rend. 1: HIP MI355X trilinear" in text
    rows = {m.group(1).strip(): (m.group(2), m.group(3)) for m in
            re.finditer(r"^\s*([A-Za-z0-9:+*. ]+), Avg\(ms\),\s*([0-9.]+|N/A),\s*([0-9.]+|N/A)\s*$", text.split("Summary profiler report:")[1], re.M)}
    assert rows["Bucky"][0] != "N/A" and rows["Bucky"][1] != "N/A"          # 8 samples per renderer -> averages printed
    assert rows["Daisy"] == ("N/A", "N/A")                                   # dataset not shipped: skipped
    for name in ("Foot: No optims", "F: ERT on", "F: ERT+ESL on", "Scale 0.9", "Scale 0.3", "Ray step *1.1", "Ray step *1.7"):
        assert name in rows and rows[name][0] != "N/A", name
    assert float(rows["Foot: No optims"][1]) >= float(rows["F: ERT+ESL on"][1]) * 0.5
    assert "Resolution: 230x230" in text and "Resolution: 76x76" in text    # 256 * 0.9 ... 256 * 0.3 (ushort truncation)
