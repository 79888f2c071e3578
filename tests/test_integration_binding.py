"""INTEGRATION.md's reference-side binding (oracle/integration/HIPRenderer.h) compiled against the reference's REAL headers and
object code (oracle/Makefile `binding`, host compiler, -lvr_hip) and driven like VolR.cpp drives its renderers.  In the build
container there is no GPU: the constructor must log and survive, every virtual must be a safe no-op and render_volume() /
set_volume() must return 1 (the reference's failure convention, CPURenderer.cpp:44-45).  The GPU-side run of the same binary
(frames of HIPRenderer == frames of the reference's CPURenderer) is tests/test_gpu_driver.py::test_reference_binding_on_the_gpu."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "binding_check")


def build_binding():
    if os.path.isdir("/root/reference/VolumeRendering"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "volume-rendering_amd", "csrc")], stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "binding"], stdout=subprocess.DEVNULL)
    return os.path.exists(BIN)


def test_binding_builds_against_the_reference_headers_and_fails_safely_without_a_gpu():
    if not build_binding():
        pytest.skip("neither /root/reference nor a prebuilt oracle/_ref/binding_check")
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the -m gpu run of the same binary")
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "HIP renderer: " in out.stdout                                   # the constructor logged the create error
    assert "render_volume returned 1, set_volume returned 1" in out.stdout
    assert "binding check passed" in out.stdout
