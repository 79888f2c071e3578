"""The CPU side under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on this pool):
`make -C oracle asan` builds oracle/vr_oracle.c and the product's host sources (volume I/O, feeders, camera, frame statistics) with
-fsanitize=address,undefined -fno-sanitize-recover; oracle/asan_driver.cpp drives them — the reference's Bucky.pvm, ~600 truncated /
bit-flipped / garbage-header variants of it, RAW + 16->8 bit quantisation, the feeders at their limits, the camera, and 96 frames of the
restatement in every sampling mode.  Any finding aborts the binary."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_sources_and_restatement_under_asan_ubsan(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([os.path.join(ROOT, "oracle", "asan_driver"), os.path.join(ROOT, "tests", "golden", "Bucky.pvm"), str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "asan_driver ok" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    variants = int(r.stdout.split("ok: ")[1].split(" file variants")[0])
    assert variants >= 600
