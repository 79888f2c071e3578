"""What pins TRILINEAR mode (GPURenderer4 semantics; the reference's renderer 4 needs nvcc + texture hardware and cannot run here).

Stated tolerances, every one asserted here on all golden cases, per channel of the RGBA8 frame:
  T1  fp32 restatement (oracle VR_SAMPLE_TRILINEAR == the HIP kernel bit for bit, tests/test_gpu_parity.py) against the same
      published model evaluated in DOUBLE precision (oracle VRO_SAMPLE_TRILINEAR_F64): at most 0.05 % of the pixels differ,
      by at most 4 (1 in all but one case) — the arithmetic is the model up to fp32 rounding;
  T2  8-bit filter weights (VR_SAMPLE_TRILINEAR_Q8, the texture unit's published weight precision) against fp32 weights:
      mean |delta| <= 0.10, at most 25 % of the pixels differ, max <= 40 (isolated pixels on transfer-function edges);
  T3  against the reference's own CPURenderer frames (NEAREST sampling — a different sampling MODEL, so this bounds the
      model difference, not an error): mean |delta| <= 21 on Bucky 32^3, <= 17 on shell48, <= 7.5 on the 40x24x56 blob;
      both renderers cover the same pixels up to 8 % of the covered area (views from outside the cube).
The GPU-side twins of T1-T3 (HIP frames instead of the restatement's) are in tests/test_gpu_parity.py."""
import numpy as np
import pytest

from helpers import VRO_SAMPLE_TRILINEAR_F64, frame_delta

T1_MAX_DIFFERING, T1_MAX_DELTA = 0.0005, 4
T2_MAX_MEAN, T2_MAX_DIFFERING, T2_MAX_DELTA = 0.10, 0.25, 40
T3_MAX_MEAN = {"bucky": 21.0, "shell48": 17.0, "blob_40x24x56": 7.5}
T3_MAX_COVERAGE_DIFF = 0.08


def _frames(vr, oracle, golden, case, modes):
    st = golden.volume_state(case["volume"])
    vox = golden.voxels(case["volume"])
    return [oracle.render(golden.params(case, m), vox, st["tf"], st["esl"]) for m in modes]


def test_t1_fp32_restatement_is_the_double_precision_model(vr, oracle, golden):
    for case in golden.cases(True):
        f32, f64 = _frames(vr, oracle, golden, case, (vr.SAMPLE_TRILINEAR, VRO_SAMPLE_TRILINEAR_F64))
        mean, differing, maxd = frame_delta(f32, f64)
        assert differing <= T1_MAX_DIFFERING and maxd <= T1_MAX_DELTA, (case["label"], differing, maxd)


def test_t2_eight_bit_weights_against_fp32_weights(vr, oracle, golden):
    worst = 0.0
    for case in golden.cases(True):
        f32, q8 = _frames(vr, oracle, golden, case, (vr.SAMPLE_TRILINEAR, vr.SAMPLE_TRILINEAR_Q8))
        mean, differing, maxd = frame_delta(f32, q8)
        assert mean <= T2_MAX_MEAN and differing <= T2_MAX_DIFFERING and maxd <= T2_MAX_DELTA, (case["label"], mean, differing, maxd)
        worst = max(worst, mean)
    assert worst > 0.0                                   # the two variants are NOT the same arithmetic


def test_t3_model_difference_against_reference_frames(vr, oracle, golden):
    for case in golden.cases(True):
        tri, = _frames(vr, oracle, golden, case, (vr.SAMPLE_TRILINEAR,))
        ref = golden.frame(case)
        mean, _, _ = frame_delta(tri, ref)
        assert mean <= T3_MAX_MEAN[case["volume"]], (case["label"], mean)
        if case["label"].startswith("inside"):
            continue        # camera inside the cube: the first samples sit in low-density voxels, where the two models differ most
        cov_t, cov_r = int((tri[..., 3] != 0).sum()), int((ref[..., 3] != 0).sum())
        assert abs(cov_t - cov_r) <= T3_MAX_COVERAGE_DIFF * max(cov_r, 1) + 16, (case["label"], cov_t, cov_r)
