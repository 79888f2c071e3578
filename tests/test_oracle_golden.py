"""The CPU restatement (oracle/vr_oracle.c) against frames rendered by the reference's own CPURenderer.cpp
(tests/golden, made by oracle/gen_golden.py from oracle/_ref).  Bar: bit-exact in NEAREST mode."""
import numpy as np
import pytest

from helpers import compare_frames, fnv1a32


def test_bucky_fixture_matches_survey_hash(golden):
    import os
    from helpers import GOLDEN_DIR
    raw = np.fromfile(os.path.join(GOLDEN_DIR, "bucky_32.raw"), dtype=np.uint8)
    assert raw.size == 32768
    assert fnv1a32(raw) == "70f1ecd5"                        # decoded Bucky.pvm
    assert raw.min() == 0 and raw.max() == 255 and int(raw.sum()) == 1174726
    assert list(raw[:16]) == [1, 1, 1, 1, 2, 4, 9, 22, 50, 92, 124, 107, 70, 51, 50, 72]
    assert np.array_equal(raw.reshape(32, 32, 32), golden.voxels("bucky"))


def test_oracle_nearest_bit_exact_on_every_golden_frame(oracle, golden):
    checked = 0
    for case in golden.cases(with_frames_only=True):
        st = golden.volume_state(case["volume"])
        out = oracle.render(golden.params(case, sampling=0), golden.voxels(case["volume"]), st["tf"], st["esl"], threads=4)
        ndiff, maxd = compare_frames(out, golden.frame(case))
        assert ndiff == 0, f"{case['label']}: {ndiff} pixels differ (max delta {maxd})"
        assert fnv1a32(out) == case["frame_fnv1a32"]
        checked += 1
    assert checked >= 40


def test_oracle_serial_equals_threaded(oracle, golden):
    case = golden.cases(True)[3]
    st = golden.volume_state(case["volume"])
    p = golden.params(case)
    a = oracle.render(p, golden.voxels(case["volume"]), st["tf"], st["esl"], threads=1)
    b = oracle.render(p, golden.voxels(case["volume"]), st["tf"], st["esl"], threads=8)
    assert np.array_equal(a, b)


def test_oracle_config2_hashes(oracle, golden):
    """BASELINE config 2 (shell 256^3 @ 1024^2): hashes of the reference's frames, default and no-optims."""
    vox = golden.voxels("shell256")
    assert fnv1a32(vox) == "6d5baf38"
    st = golden.volume_state("shell256")
    for case in golden.cases():
        if case["volume"] != "shell256":
            continue
        out = oracle.render(golden.params(case), vox, st["tf"], st["esl"], threads=8)
        assert fnv1a32(out) == case["frame_fnv1a32"], case["label"]
        assert int((out[..., 3] != 0).sum()) == case["nonzero_alpha"]


def test_esl_is_lossless(oracle, golden):
    """ESL on/off give identical images (NEAREST mode)."""
    for case in golden.cases(True):
        if case["volume"] != "shell48" or "default" not in case["label"]:
            continue
        st = golden.volume_state("shell48")
        p = golden.params(case)
        on = oracle.render(p, golden.voxels("shell48"), st["tf"], st["esl"])
        p.esl = 0
        off = oracle.render(p, golden.voxels("shell48"), st["tf"], st["esl"])
        assert np.array_equal(on, off)


def test_trilinear_close_to_nearest(oracle, golden):
    """TRILINEAR (GPURenderer4 semantics) is a different sampling model: informational bound against NEAREST only."""
    case = [c for c in golden.cases(True) if c["label"] == "bench64_view1_default"][0]
    st = golden.volume_state("bucky")
    near = oracle.render(golden.params(case, 0), golden.voxels("bucky"), st["tf"], st["esl"])
    tri = oracle.render(golden.params(case, 1), golden.voxels("bucky"), st["tf"], st["esl"])
    assert (near[..., 3] != 0).sum() > 1000 and (tri[..., 3] != 0).sum() > 1000
    mean_delta = np.abs(near.astype(np.float32) - tri.astype(np.float32)).mean()
    assert mean_delta < 12.0


def test_null_arguments_return_1(oracle, golden):
    import ctypes as C
    case = golden.cases(True)[0]
    p = golden.params(case)
    assert oracle.L.vro_render(C.byref(p), None, None, 1, None, None, None, 1, None, 0) == 1   # CPURenderer.cpp:44-45
