"""Host mirror of the reference's scene-state managers (ViewBase / RaycasterBase in libvr_hip.so) against values that came
out of the reference's own object code (tests/golden) and against the CPU oracle."""
import ctypes as C

import numpy as np


def test_transfer_fn_esl_raystep_match_reference(vr, golden, oracle):
    for name in ("bucky", "shell48", "blob_40x24x56"):
        vox, st = golden.voxels(name), golden.volume_state(name)
        scene = vr.Scene().set_volume(voxels=vox)
        if st["base_tf"] is not None:
            scene.set_base_transfer_fn(st["base_tf"])
        assert np.array_equal(scene.tf, st["tf"]), name                       # premultiplied TF, bit for bit
        assert np.array_equal(scene.esl, st["esl"]), name                     # ESL bit-volume
        assert np.float32(scene.params.ray_step) == st["ray_step"]
        assert scene.params.esl_block_dims == st["esl_block_dims"]
        assert np.array_equal(np.array(list(scene.params.esl_block_size), np.float32), st["esl_block_size"])
        assert np.float32(scene.params.ray_threshold) == np.float32(0.95) and scene.params.esl == 1
        assert np.float32(scene.params.light_kd) == np.float32(0.6)
        # the oracle's restatement of the same feeders agrees as well
        base = oracle.default_base_tf() if st["base_tf"] is None else st["base_tf"]
        mm, bd, bs = oracle.volume_minmax(vox)
        tf, esl = oracle.update_transfer_fn(base, mm)
        assert np.array_equal(tf, st["tf"]) and np.array_equal(esl, st["esl"]) and bd == st["esl_block_dims"]
        assert np.array_equal(mm, scene.minmax)
        z, y, x = vox.shape
        assert oracle.default_ray_step((x, y, z)) == st["ray_step"]


def test_survey_captured_values(vr, golden):
    """values printed by the reference for Bucky."""
    st = golden.volume_state("bucky")
    assert sum(bin(int(w)).count("1") for w in st["esl"]) == 32704
    words = [int(w) for w in st["esl"]]
    assert set(words) == {0xFFFFFFF0, 0xFFFFFFFF} and words.count(0xFFFFFFF0) == 16   # 4x4x4 occupied blocks of 8^3
    np.testing.assert_allclose(st["tf"][13], [0.0309448, 0, 0, 0.101562], rtol=1e-5)
    np.testing.assert_allclose(st["tf"][64], [0, 0.257812, 0, 0.5], rtol=1e-5)
    np.testing.assert_allclose(st["tf"][127], [0, 0, 0.999939, 0.992188], rtol=1e-5)
    assert st["ray_step"] == np.float32(0.060546875)
    assert golden.index["volumes"]["shell256"]["esl_popcount"] == 24329


def test_benchmark_views(vr, golden):
    """The camera mirror regenerates the golden views exactly (those views reproduce the reference's frames hash for hash,
    see oracle/gen_golden.py); one view is also pinned numerically by SURVEY §8(c)."""
    a = golden.arrays
    for case in golden.cases():
        label = case["label"]
        if not label.startswith("bench256_view"):
            continue
        i = int(label[len("bench256_view")])
        v = vr.benchmark_view(256, 256, i)
        got = np.array(list(v.origin) + list(v.direction) + list(v.right_plane) + list(v.up_plane) + list(v.light_pos), np.float32)
        assert np.array_equal(got, a[f"case{case['id']}_view"]), label
        assert v.perspective == (1 if i >= 4 else 0)
    v = vr.benchmark_view(256, 256, 1)
    np.testing.assert_allclose(list(v.origin), [1, -1.414214, 1], rtol=2e-6)
    np.testing.assert_allclose(list(v.direction), [-0.5, 0.707107, -0.5], rtol=2e-6)
    np.testing.assert_allclose(list(v.right_plane), [0.00552427, 0, -0.00552427], rtol=2e-6, atol=1e-12)
    np.testing.assert_allclose(list(v.up_plane), [0.00390625, 0.00552427, 0.00390625], rtol=2e-6)
    # perspective: virtual view size 1.5 (ViewBase.cpp:100-105) -> step 1.5 / min(W,H); origin at distance 2
    v = vr.benchmark_view(2048, 1024, 4)
    assert list(v.origin) == [0.0, 0.0, 2.0] and abs(v.right_plane[0] - 1.5 / 1024) < 1e-9 and v.up_plane[1] == v.right_plane[0]


def test_clamped_setters(vr, golden):
    """RaycasterBase.cpp:26-44: ray_step clamps to [default/3, default*1.666], threshold to [0.5, 1], light to [0, 2]."""
    scene = vr.Scene().set_volume(voxels=golden.voxels("bucky"))
    d = np.float32(scene.params.ray_step)
    scene.set_modes(ray_step=10.0)
    assert np.float32(scene.params.ray_step) == np.float32(d * np.float32(1.666))
    scene.set_modes(ray_step=0.0)
    assert np.float32(scene.params.ray_step) == np.float32(d / np.float32(3))
    scene.set_modes(ray_threshold=0.1, light_kd=5.0, esl=False)
    assert scene.params.ray_threshold == 0.5 and scene.params.light_kd == 2.0 and scene.params.esl == 0
    scene.set_modes(ray_threshold=1.0, light_kd=-1.0, esl=True)
    assert scene.params.ray_threshold == 1.0 and scene.params.light_kd == 0.0 and scene.params.esl == 1


def test_band_partition_fields(vr):
    p = vr.VrParams()
    p.view.width, p.view.height = 199, 178
    p, per_rank = vr.band_partition(p, rank=2, world=3, band_rows=16)
    assert (p.out_width, p.out_rows, p.band_rows, p.band_stride, p.band_first, per_rank) == (199, 64, 16, 3, 2, 4)
    p = vr.whole_frame(p)
    assert (p.out_rows, p.band_rows, p.band_stride, p.band_first) == (178, 178, 1, 0)
