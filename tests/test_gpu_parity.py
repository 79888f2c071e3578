"""Parity tests proper: the HIP path, called through the C ABI, against (a) frames rendered by the reference's own
CPURenderer (tests/golden) and (b) the CPU oracle on the same inputs.

Bar: NEAREST mode bit-exact (integer RGBA8 output of identical fp32 arithmetic); TRILINEAR mode bit-exact against the
oracle's restatement of GPURenderer4's texture semantics (tolerance in the test: 0)."""
import ctypes as C

import numpy as np
import pytest

from helpers import compare_frames, fnv1a32

pytestmark = pytest.mark.gpu


def load_volume(gpu, golden, name):
    st = golden.volume_state(name)
    gpu.set_transfer_fn(st["tf"], st["esl"])
    gpu.set_volume(golden.voxels(name))
    return st


def test_native_library_is_the_one_loaded(vr, gpu):
    import os
    with open("/proc/self/maps") as f:
        assert os.path.basename(vr.library_path()) in f.read()      # in-tree libvr_hip.so (or the VR_HIP_LIB build under A/B)
    name, cus, mem = gpu.device_info()
    assert "gfx950" in name and cus == 256, (name, cus)


def test_nearest_bit_exact_vs_reference_frames(vr, gpu, golden):
    """Every golden frame (Bucky 8 benchmark views default/no-optims at 256^2 and 64^2, odd windows, camera inside the
    cube, light off, ray-step limits, synthetic shell, non-cubic volume with an edited TF) — NEAREST, bit for bit."""
    gpu.set_window_buffer(256, 256)
    current = None
    for case in golden.cases(with_frames_only=True):
        if case["volume"] != current:
            load_volume(gpu, golden, case["volume"])
            current = case["volume"]
        out = gpu.render_volume(golden.params(case, vr.SAMPLE_NEAREST))
        ndiff, maxd = compare_frames(out, golden.frame(case))
        assert ndiff == 0, f"{case['label']}: {ndiff} pixels differ, max delta {maxd}"


def test_nearest_address_chain_variants_agree(vr, gpu, golden):
    """Bucky is 32^3 (power-of-two edges): NEAREST marches in the scaled domain by default; + 8 switches that off, + 4 clamps every
    fetch coordinate.  All three are the reference's frame, bit for bit."""
    gpu.set_window_buffer(256, 256)
    load_volume(gpu, golden, "bucky")
    try:
        for case in [c for c in golden.cases(True) if c["volume"] == "bucky"][::3]:
            for force in (0, 8, 4, 12):
                for plane in (-1, 0):            # -1: the voxel bricks (one voxel per element); 0: the (x,y) quad copy
                    gpu.set_wide_addressing(force)
                    gpu.set_brick_plane(plane)
                    out = gpu.render_volume(golden.params(case, vr.SAMPLE_NEAREST))
                    assert compare_frames(out, golden.frame(case)) == (0, 0), (case["label"], force, plane)
    finally:
        gpu.set_wide_addressing(False)
        gpu.set_brick_plane(-1)


def test_config2_shell256_hashes(vr, gpu, golden):
    """BASELINE config 2: shell 256^3 generated on the GPU, 1024x1024, ortho poses 0/1, default and no-optims."""
    gpu.generate_volume("shell", 256, seed=1)
    assert fnv1a32(gpu.download_volume()) == "6d5baf38"
    st = golden.volume_state("shell256")
    gpu.set_transfer_fn(st["tf"], st["esl"])
    gpu.set_window_buffer(1024, 1024)
    for case in golden.cases():
        if case["volume"] != "shell256":
            continue
        out = gpu.render_volume(golden.params(case, vr.SAMPLE_NEAREST))
        assert fnv1a32(out) == case["frame_fnv1a32"], case["label"]
        assert int((out[..., 3] != 0).sum()) == case["nonzero_alpha"]


def test_trilinear_bit_exact_vs_oracle(vr, gpu, golden, oracle):
    gpu.set_window_buffer(256, 256)
    current, checked = None, 0
    for case in golden.cases(with_frames_only=True):
        if "bench256" in case["label"] and "view1" not in case["label"] and "view6" not in case["label"]:
            continue                                            # keep the CPU side of this test short
        if case["volume"] != current:
            st = load_volume(gpu, golden, case["volume"])
            current = case["volume"]
        p = golden.params(case, vr.SAMPLE_TRILINEAR)
        out = gpu.render_volume(p)
        ref = oracle.render(p, golden.voxels(case["volume"]), st["tf"], st["esl"], threads=16)
        ndiff, maxd = compare_frames(out, ref)
        assert ndiff == 0, f"{case['label']}: {ndiff} pixels differ, max delta {maxd}"
        checked += 1
    assert checked >= 30


def test_trilinear_q8_bit_exact_vs_oracle(vr, gpu, golden, oracle):
    """VR_SAMPLE_TRILINEAR_Q8 (8-bit filter weights, the texture unit's published precision): HIP == restatement, bit for bit."""
    gpu.set_window_buffer(256, 256)
    current, checked = None, 0
    for case in golden.cases(with_frames_only=True):
        if "bench256" in case["label"] and "view5" not in case["label"] and "view2" not in case["label"]:
            continue
        if case["volume"] != current:
            st = load_volume(gpu, golden, case["volume"])
            current = case["volume"]
        p = golden.params(case, vr.SAMPLE_TRILINEAR_Q8)
        out = gpu.render_volume(p)
        ref = oracle.render(p, golden.voxels(case["volume"]), st["tf"], st["esl"], threads=16)
        ndiff, maxd = compare_frames(out, ref)
        assert ndiff == 0, f"{case['label']}: {ndiff} pixels differ, max delta {maxd}"
        checked += 1
    assert checked >= 30


def test_trilinear_stated_tolerances_on_the_gpu(vr, gpu, golden, oracle):
    """The tolerances of tests/test_trilinear_pinning.py with the HIP frames themselves: T1 against the double-precision
    model, T2 8-bit against fp32 weights, T3 against the reference's CPURenderer frames (all 46 golden cases)."""
    import test_trilinear_pinning as tp
    from helpers import VRO_SAMPLE_TRILINEAR_F64, frame_delta
    gpu.set_window_buffer(256, 256)
    current = None
    for case in golden.cases(with_frames_only=True):
        if case["volume"] != current:
            st = load_volume(gpu, golden, case["volume"])
            current = case["volume"]
        tri = gpu.render_volume(golden.params(case, vr.SAMPLE_TRILINEAR))
        q8 = gpu.render_volume(golden.params(case, vr.SAMPLE_TRILINEAR_Q8))
        f64 = oracle.render(golden.params(case, VRO_SAMPLE_TRILINEAR_F64), golden.voxels(case["volume"]), st["tf"], st["esl"], threads=16)
        _, differing, maxd = frame_delta(tri, f64)
        assert differing <= tp.T1_MAX_DIFFERING and maxd <= tp.T1_MAX_DELTA, (case["label"], differing, maxd)
        mean, differing, maxd = frame_delta(tri, q8)
        assert mean <= tp.T2_MAX_MEAN and differing <= tp.T2_MAX_DIFFERING and maxd <= tp.T2_MAX_DELTA, (case["label"], mean, differing, maxd)
        mean, _, _ = frame_delta(tri, golden.frame(case))
        assert mean <= tp.T3_MAX_MEAN[case["volume"]], (case["label"], mean)


def test_trilinear_layouts_agree(vr, gpu, golden, oracle):
    """VR_LAYOUT_LINEAR and the brick copies of VR_LAYOUT_BRICKED hold the same voxels: identical images (all equal the oracle)."""
    for name, labels in (("bucky", ("bench64_view1_default", "bench64_view6_default", "inside_persp")),
                         ("blob_40x24x56", ("view1_default", "view7_default", "view3_esl_off"))):   # dims not multiples of 8
        st = load_volume(gpu, golden, name)
        gpu.set_window_buffer(256, 256)
        for label in labels:
            case = [c for c in golden.cases(True) if c["label"] == label and c["volume"] == name][0]
            p = golden.params(case, vr.SAMPLE_TRILINEAR)
            ref = oracle.render(p, golden.voxels(name), st["tf"], st["esl"])
            gpu.set_layout(vr.LAYOUT_LINEAR)
            assert np.array_equal(gpu.render_volume(p), ref), (name, label, "linear")
            gpu.set_layout(vr.LAYOUT_BRICKED)
            for plane in (-1, 0, 1, 2, 3, 4):    # brick copy per view, then each chunk plane and both run-brick copies forced (vr_hip_set_brick_plane)
                gpu.set_brick_plane(plane)
                assert np.array_equal(gpu.render_volume(p), ref), (name, label, plane)
            # both run copies in ONE launch, per tile (full-march frames only): 7 = alternating tiles, 6 = measured — frames 0-3 run on
            # one copy each (2 and 3 record the tile costs), frames 4 and 5 read the per-tile choice; fp32 and 8-bit weights, lit and unlit
            for samp in (vr.SAMPLE_TRILINEAR, vr.SAMPLE_TRILINEAR_Q8):
                for kd in (0.6, 0.0):
                    pf = golden.params(case, samp)
                    pf.esl, pf.ray_threshold, pf.light_kd = 0, 1.0, kd
                    want = oracle.render(pf, golden.voxels(name), st["tf"], st["esl"])
                    gpu.set_brick_plane(7)
                    assert np.array_equal(gpu.render_volume(pf), want), (name, label, samp, kd, "alternating tiles")
                    clamped = gpu.last_launch()["clamp_fetch"] != 0      # the clamping instantiation never takes the per-tile choice
                    assert gpu.last_launch()["layout"] == (2 if clamped else 6) or gpu.last_launch()["layout"] == 3
                    gpu.set_brick_plane(6)
                    seen = []
                    for frame in range(6):
                        assert np.array_equal(gpu.render_volume(pf), want), (name, label, samp, kd, "measured", frame)
                        seen.append(gpu.last_launch()["layout"])
                    assert clamped or seen == [2, 3, 2, 3, 6, 6], seen
                    # the product's rule: every tile picks its copy from the entry face of its block's centre ray, first frame included
                    gpu.set_brick_plane(-1)
                    assert np.array_equal(gpu.render_volume(pf), want), (name, label, samp, kd, "analytic")
            gpu.set_brick_plane(-1)


def test_column_march_matches_oracle(vr, gpu, golden, oracle):
    """kLayoutColumn (colmarch_kernel): full-march TRILINEAR frames of orthogonal views along a volume axis read the column windows and march
    on wave-uniform state; poses whose direction carries rounding noise make lanes flip cell columns mid-march (careful windows), windows
    that do not hold three cells end the volume (edges 32, 40, 24, 56), tile phases shift the waves over the columns.  Forced
    (vr_hip_set_brick_plane(8)) every orthogonal full-march frame takes the kernel, oblique ones mostly through its per-lane march."""
    poses = ((0.0, 0.0, 0.0), (90.0, 0.0, 0.0), (180.0, 90.0, 0.0), (0.0, 90.0, 0.0), (270.0, 0.0, 0.0), (0.0, 180.0, 0.0), (90.0, 90.0, 0.0), (0.0, 0.0, 90.0))
    for name, label, sizes in (("bucky", "bench64_view1_default", ((256, 256), (130, 67))), ("blob_40x24x56", "view1_default", ((192, 160),))):
        st = load_volume(gpu, golden, name)
        case = [c for c in golden.cases(True) if c["label"] == label and c["volume"] == name][0]
        for (w, h) in sizes:
            gpu.set_window_buffer(w, h)
            for angles in poses + ((0.02, 0.0, 0.0), (90.0, 0.013, 0.0), (-45.0, -45.0, 0.0), (1.5, 2.5, 0.0)):
                axis_aligned = angles in poses
                for samp, kd, step_scale in ((vr.SAMPLE_TRILINEAR, 0.6, 1.0), (vr.SAMPLE_TRILINEAR_Q8, 0.0, 1.0), (vr.SAMPLE_TRILINEAR, 0.6, 0.37),
                                             (vr.SAMPLE_NEAREST, 0.6, 1.0), (vr.SAMPLE_NEAREST, 0.0, 0.37)):       # NEAREST: windows of 16 voxels (colmarch_nearest_kernel)
                    p = golden.params(case, samp)
                    v = vr.custom_view(w, h, False, angles, 2.0)
                    for f in ("origin", "direction", "right_plane", "up_plane"):
                        for j in range(3):
                            getattr(p.view, f)[j] = getattr(v, f)[j]
                    p.view.width, p.view.height, p.view.perspective = w, h, 0
                    p = vr.whole_frame(p)
                    p.esl, p.ray_threshold, p.light_kd = 0, 1.0, kd
                    p.ray_step = float(np.float32(p.ray_step) * np.float32(step_scale))
                    want = oracle.render(p, golden.voxels(name), st["tf"], st["esl"])
                    gpu.set_brick_plane(-1)
                    assert np.array_equal(gpu.render_volume(p), want), (name, angles, samp, kd, step_scale, "automatic")
                    if axis_aligned:
                        assert gpu.last_launch()["layout"] == 7, (name, angles, gpu.last_launch())
                    gpu.set_brick_plane(8)
                    assert np.array_equal(gpu.render_volume(p), want), (name, angles, samp, kd, step_scale, "forced")
                    assert gpu.last_launch()["layout"] == 7, (name, angles, gpu.last_launch())
                    gpu.set_brick_plane(9)
                    assert np.array_equal(gpu.render_volume(p), want), (name, angles, samp, kd, step_scale, "never")
                    assert gpu.last_launch()["layout"] != 7
            # a band partition and forced tile phases move the waves over the cell columns
            p = golden.params(case, vr.SAMPLE_TRILINEAR)
            v = vr.custom_view(w, h, False, (180.0, 90.0, 0.0), 2.0)
            for f in ("origin", "direction", "right_plane", "up_plane"):
                for j in range(3):
                    getattr(p.view, f)[j] = getattr(v, f)[j]
            p.view.width, p.view.height, p.view.perspective = w, h, 0
            p = vr.whole_frame(p)
            p.esl, p.ray_threshold = 0, 1.0
            want = oracle.render(p, golden.voxels(name), st["tf"], st["esl"])
            gpu.set_brick_plane(-1)
            for lane_map in (0, 1, 2):
                for ph in ((0, 0), (3, 5), (7, 1)):
                    gpu.set_tile_mapping(lane_map, *ph)
                    assert np.array_equal(gpu.render_volume(p), want), (name, lane_map, ph)
                    assert gpu.last_launch()["layout"] == 7
            gpu.set_tile_mapping(-1)
            rows = []
            for rank in range(3):
                pb, _ = vr.band_partition(p.copy(), rank, 3, 16)
                rows.append((pb, gpu.render_volume(pb)))
            for rank, (pb, img) in enumerate(rows):
                assert np.array_equal(img, oracle.render(pb, golden.voxels(name), st["tf"], st["esl"])), (name, "band", rank)
    gpu.set_brick_plane(-1)


def test_u16_volume_matches_oracle(vr, gpu, golden, oracle):
    """Build-side extension (the reference quantises 16-bit data to 8 bit on load, ModelBase.cpp:95-98)."""
    vox16 = golden.voxels("bucky").astype(np.uint16) * 257
    st = golden.volume_state("bucky")
    gpu.set_transfer_fn(st["tf"], st["esl"])
    gpu.set_volume(vox16)
    gpu.set_window_buffer(96, 96)
    for label in ("nolight_view1", "nolight_view5", "bench64_view3_default", "bench64_view6_default"):
        case = [c for c in golden.cases(True) if c["label"] == label][0]
        for mode in (vr.SAMPLE_NEAREST, vr.SAMPLE_TRILINEAR):
            p = golden.params(case, mode)
            out = gpu.render_volume(p)
            ref = oracle.render(p, vox16, st["tf"], st["esl"])
            assert compare_frames(out, ref) == (0, 0), (label, mode)
    # orthogonal TRILINEAR views read the oct bricks (one 16-byte element per cell), perspective ones the quad bricks, NEAREST the voxel bricks
    info = gpu.volume_info()
    assert info.copies == vr.COPY_OCT | vr.COPY_QUAD_XY | vr.COPY_VOXEL and info.bricked_bytes == (16 + 8 + 2) * 32 ** 3, (info.copies, info.bricked_bytes)
    # the three layouts a 2-byte TRILINEAR frame can read agree: oct bricks (forced for every view), quad bricks (forced plane), the linear array
    for label in ("bench64_view1_default", "bench64_view6_default"):
        case = [c for c in golden.cases(True) if c["label"] == label][0]
        for mode in (vr.SAMPLE_TRILINEAR, vr.SAMPLE_TRILINEAR_Q8):
            p = golden.params(case, mode)
            gpu.set_brick_plane(5)
            oct_frame = gpu.render_volume(p)
            gpu.set_brick_plane(0)
            quad_frame = gpu.render_volume(p)
            gpu.set_brick_plane(-1)
            assert np.array_equal(gpu.render_volume(p), oct_frame)
            gpu.set_layout(vr.LAYOUT_LINEAR)
            linear_frame = gpu.render_volume(p)
            gpu.set_layout(vr.LAYOUT_BRICKED)
            assert np.array_equal(oct_frame, quad_frame) and np.array_equal(oct_frame, linear_frame), (label, mode)
            assert np.array_equal(oct_frame, oracle.render(p, vox16, st["tf"], st["esl"])), (label, mode)
    # u8 * 257 in NEAREST mode is the same picture as the u8 volume except for the /65535 vs /255 shading scale
    case = [c for c in golden.cases(True) if c["label"] == "nolight_view1"][0]
    out16 = gpu.render_volume(golden.params(case, vr.SAMPLE_NEAREST))
    assert np.array_equal(out16, golden.frame(case))


def test_wide_addressing_path(vr, gpu, golden, oracle):
    """The 64-bit index path of volumes beyond 1024^3 (BASELINE config 5), forced on small volumes: same images.
    u8 and u16, both layouts, both sampling modes."""
    vox8 = golden.voxels("blob_40x24x56")
    st = golden.volume_state("blob_40x24x56")
    gpu.set_transfer_fn(st["tf"], st["esl"])
    gpu.set_window_buffer(256, 256)
    cases = [c for c in golden.cases(True) if c["volume"] == "blob_40x24x56"][:3]
    try:
        for vox in (vox8, vox8.astype(np.uint16) * 257):
            gpu.set_volume(vox)
            for layout in (vr.LAYOUT_BRICKED, vr.LAYOUT_LINEAR):
                gpu.set_layout(layout)
                for case in cases:
                    for mode in (vr.SAMPLE_NEAREST, vr.SAMPLE_TRILINEAR):
                        p = golden.params(case, mode)
                        gpu.set_wide_addressing(False)
                        narrow = gpu.render_volume(p)
                        for force in (1, 2, 4, 6):  # arithmetic 64-bit path, table path with 64-bit z offsets, + clamped fetch
                            gpu.set_wide_addressing(force)
                            wide = gpu.render_volume(p)
                            assert np.array_equal(narrow, wide), (vox.dtype, layout, case["label"], mode, force)
                        assert np.array_equal(wide, oracle.render(p, vox, st["tf"], st["esl"])), (vox.dtype, layout, case["label"], mode)
    finally:
        gpu.set_wide_addressing(False)
        gpu.set_layout(vr.LAYOUT_BRICKED)


def test_tile_mapping_never_changes_the_image(vr, gpu, golden, oracle):
    """Lane order inside the 4x4-pixel blocks and the tile-grid phase are scheduling only (vr_hip_set_tile_mapping): every
    forced combination, and the automatic choice, give the oracle's image — whole frame, ragged window, and a banded partition."""
    vox8 = golden.voxels("blob_40x24x56")
    st = golden.volume_state("blob_40x24x56")
    gpu.set_volume(vox8)
    gpu.set_transfer_fn(st["tf"], st["esl"])
    cases = [c for c in golden.cases(True) if c["volume"] == "blob_40x24x56"][:2]
    try:
        for case in cases:
            for mode in (vr.SAMPLE_NEAREST, vr.SAMPLE_TRILINEAR):
                p = golden.params(case, mode)
                gpu.set_window_buffer(p.view.width, p.view.height)
                ref = oracle.render(p, vox8, st["tf"], st["esl"])
                gpu.set_tile_mapping(-1)
                assert np.array_equal(gpu.render_volume(p), ref), (case["label"], mode, "auto")
                for lane_map in (0, 1, 2, 4, 5, 6, 8, 9, 10):                 # lane order + 4 * wave shape (8x8, 16x4, 4x16)
                    for px, py in ((0, 0), (1, 0), (3, 2), (7, 7), (5, 0)) if lane_map < 3 else ((0, 0), (3, 6)):
                        gpu.set_tile_mapping(lane_map, px, py)
                        assert np.array_equal(gpu.render_volume(p), ref), (case["label"], mode, lane_map, px, py)
        # banded partition (rank 1 of 3, 16-row bands) with a shifted grid
        p = golden.params(cases[0], vr.SAMPLE_TRILINEAR)
        whole = oracle.render(p, vox8, st["tf"], st["esl"])
        part, _ = vr.band_partition(golden.params(cases[0], vr.SAMPLE_TRILINEAR), 1, 3, 16)
        ref_part = oracle.render(part, vox8, st["tf"], st["esl"])
        for lane_map, px, py in ((-1, 0, 0), (2, 3, 1), (1, 6, 5)):
            gpu.set_tile_mapping(lane_map, px, py)
            assert np.array_equal(gpu.render_volume(part), ref_part), (lane_map, px, py)
        assert whole.shape[0] >= ref_part.shape[0]
    finally:
        gpu.set_tile_mapping(-1)


def test_partition_concatenates_to_the_whole_frame(vr, gpu, golden):
    """SURVEY §8e correctness check: n ranks' bands assembled == 1-rank frame, byte for byte."""
    from importlib import import_module
    dist_mod = import_module("volume-rendering_amd.distributed")
    import torch
    load_volume(gpu, golden, "bucky")
    gpu.set_window_buffer(199, 178)
    case = [c for c in golden.cases(True) if c["label"] == "window_199x178_view1"][0]
    for mode in (vr.SAMPLE_NEAREST, vr.SAMPLE_TRILINEAR):
        whole = gpu.render_volume(golden.params(case, mode))
        for world, band in ((2, 16), (3, 16), (4, 8), (2, 89), (8, 16), (5, 7)):
            parts = []
            for rank in range(world):
                split = dist_mod.FrameSplit(199, 178, world, rank, band)
                parts.append(torch.from_numpy(gpu.render_volume(split.apply(golden.params(case, mode)))))
            frame = split.assemble(torch.stack(parts)).numpy()
            assert np.array_equal(frame, whole), (mode, world, band)
    # a column window (x0 / out_width) of the frame
    whole = gpu.render_volume(golden.params(case, vr.SAMPLE_NEAREST))
    p = golden.params(case, vr.SAMPLE_NEAREST)
    p.x0, p.out_width = 50, 64
    assert np.array_equal(gpu.render_volume(p), whole[:, 50:114])


def test_device_buffer_path_with_torch_stream(vr, gpu, golden):
    """vr_hip_render_device into a torch tensor on torch's current stream == the host-buffer path."""
    import torch
    load_volume(gpu, golden, "bucky")
    gpu.set_window_buffer(256, 256)
    case = [c for c in golden.cases(True) if c["label"] == "bench256_view5_default"][0]
    p = golden.params(case, vr.SAMPLE_NEAREST)
    # A stream torch owns (non-zero handle): the fill, the kernel and the copy back are ordered on it.  Handle 0 / NULL would
    # mean "the context's own non-blocking stream" to the C ABI, which is NOT ordered against torch's default stream.
    s = torch.cuda.Stream()
    assert s.cuda_stream != 0
    with torch.cuda.stream(s):
        dev = torch.full((256, 256, 4), 77, dtype=torch.uint8, device="cuda:0")     # the kernel must overwrite every byte
        gpu.render_volume_device(p, dev.data_ptr(), s.cuda_stream)
        host = dev.cpu()
    s.synchronize()
    assert np.array_equal(host.numpy(), golden.frame(case))
    t = gpu.timing()
    assert t.launches >= 1 and t.kernel_ms > 0


def test_set_volume_from_device_memory(vr, gpu, golden):
    """vr_hip_set_volume_device: the voxels come from a buffer that is already in HBM (a torch tensor here)."""
    import torch
    st = load_volume(gpu, golden, "blob_40x24x56")
    gpu.set_window_buffer(256, 256)
    case = [c for c in golden.cases(True) if c["volume"] == "blob_40x24x56"][0]
    vox = golden.voxels("blob_40x24x56")
    dev = torch.from_numpy(vox.copy()).to("cuda:0")
    torch.cuda.synchronize()
    for mode in (vr.SAMPLE_NEAREST, vr.SAMPLE_TRILINEAR):
        p = golden.params(case, mode)
        host_path = gpu.render_volume(p)
        gpu.set_volume_device(dev.data_ptr(), (vox.shape[2], vox.shape[1], vox.shape[0]), 1)
        assert np.array_equal(gpu.render_volume(p), host_path)
        assert np.array_equal(gpu.download_volume(), vox)
        gpu.set_volume(vox)
    assert np.array_equal(gpu.render_volume(golden.params(case, vr.SAMPLE_NEAREST)), golden.frame(case))


def test_tile_scheduling_is_placement_only(vr, gpu, oracle):
    """Measured-cost launch order (vr_hip_set_tile_scheduling 1, the default): the first frame with a set of parameters records the
    cost of every tile, later frames start their most expensive tiles first.  Every frame — recording, ordered, ordered from
    another stream — equals the frame rendered in plain workgroup order and the oracle's, in both sampling modes, for a frame
    with a ragged tile grid; and the full march (no ESL, threshold 1) never uses an order."""
    import torch
    vox = oracle.generate_volume("shell", 64, 1)
    tf, esl, bd, bs, step = oracle.scene_for(vox)
    gpu.set_transfer_fn(tf, esl)
    gpu.set_volume(vox)
    W, H = 500, 333                                       # 16 x 21 tiles of 32x16 pixels, ragged on both sides
    gpu.set_window_buffer(W, H)
    try:
        for view_i in (1, 3, 6):
            for samp in (vr.SAMPLE_NEAREST, vr.SAMPLE_TRILINEAR):
                p = vr.VrParams()
                p.view = vr.benchmark_view(W, H, view_i)
                p.ray_step, p.ray_threshold, p.esl, p.esl_block_dims, p.light_kd, p.sampling = float(step), 0.95, 1, bd, 0.6, samp
                for j in range(3):
                    p.esl_block_size[j] = float(bs[j])
                vr.whole_frame(p)
                want = oracle.render(p, vox, tf, esl)
                gpu.set_tile_scheduling(0)
                assert np.array_equal(gpu.render_volume(p), want)
                gpu.set_tile_scheduling(1)
                frames = [gpu.render_volume(p) for _ in range(3)]          # recording frame, then two ordered frames
                assert all(np.array_equal(f, want) for f in frames), (view_i, samp)
                # an ordered frame on ANOTHER stream than the one the order was built on
                dev = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0")
                other = torch.cuda.Stream()
                torch.cuda.synchronize()
                gpu.render_volume_device(p, dev.data_ptr(), other.cuda_stream)
                other.synchronize()
                assert np.array_equal(dev.cpu().numpy(), want), (view_i, samp)
                # mode 2: workgroup order + the cost map of the frame (profiling): same image, one cost per tile of the kernel's grid
                gpu.set_tile_scheduling(2)
                assert np.array_equal(gpu.render_volume(p), want), (view_i, samp)
                info, costs = gpu.last_launch(), gpu.tile_costs()
                assert costs.shape == (info["tiles_y"], info["tiles_x"]) and info["ordered"] == 0
                assert info["tiles_x"] >= (W + 31) // 32 and info["tiles_y"] >= (H + 15) // 16
                assert (costs[: H // 16, : W // 32] > 0).all()          # every tile that lies inside the frame ran and was timed
                assert info["layout"] in (1, 2, 3, 4) and info["lane_map"] < 12 and info["phase_x"] < 8 and info["phase_y"] < 8
    finally:
        gpu.set_tile_scheduling(1)


def test_orders_and_recordings_with_frames_in_flight_on_two_streams(vr, gpu, oracle):
    """buffers of the measured-cost orders must never be rewritten under a frame that is still reading them.  Ordered frames are
    queued on stream A while stream B — with no host synchronisation in between — renders more new parameter sets than the order cache
    holds entries (every one records and builds an order; a moving camera), the scheduling state is reset and a copy is prepared on the way.
    Every frame of both streams equals the oracle's (a permutation read while it is rewritten would leave pixels unwritten)."""
    import torch
    vox = oracle.generate_volume("shell", 64, 1)
    tf, esl, bd, bs, step = oracle.scene_for(vox)
    gpu.set_transfer_fn(tf, esl)
    gpu.set_volume(vox)
    W, H = 500, 333
    gpu.set_window_buffer(W, H)

    def params(view):
        p = vr.VrParams()
        p.view = view
        p.ray_step, p.ray_threshold, p.esl, p.esl_block_dims, p.light_kd, p.sampling = float(step), 0.95, 1, bd, 0.6, vr.SAMPLE_TRILINEAR
        for j in range(3):
            p.esl_block_size[j] = float(bs[j])
        return vr.whole_frame(p)

    p0 = params(vr.benchmark_view(W, H, 1))
    want0 = oracle.render(p0, vox, tf, esl)
    gpu.set_tile_scheduling(1)
    for _ in range(3):
        assert np.array_equal(gpu.render_volume(p0), want0)                # recordings, then an ordered frame
    assert gpu.last_launch()["ordered"] == 1
    moving = [params(vr.custom_view(W, H, j % 2 == 1, (-45.0 + 9.0 * j, -45.0 + 5.0 * j, 3.0 * j), 2.0)) for j in range(1, 25)]
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    out_a = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0") for _ in range(12)]
    out_b = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda:0") for _ in moving]
    torch.cuda.synchronize()
    for i, pj in enumerate(moving):
        if i % 2 == 0:
            gpu.render_volume_device(p0, out_a[i // 2].data_ptr(), a.cuda_stream)
        gpu.render_volume_device(pj, out_b[i].data_ptr(), b.cuda_stream)
        if i == 9:
            gpu.set_tile_scheduling(1)                                      # drops every remembered order with frames of both streams in flight
        if i == 15:
            gpu.prepare(vr.COPY_QUAD_XZ)
    torch.cuda.synchronize()
    for i, t in enumerate(out_a):
        assert np.array_equal(t.cpu().numpy(), want0), ("stream A frame", i)
    for i, (t, pj) in enumerate(zip(out_b, moving)):
        assert np.array_equal(t.cpu().numpy(), oracle.render(pj, vox, tf, esl)), ("stream B frame", i)


def test_multi_device_frame_equals_single_device(vr, gpu, golden):
    """vr_hip_multi_*: one call, several per-device contexts, interleaved bands gathered on devices[0] and de-interleaved there.
    On a one-GPU box the list names device 0 once (the single path), twice and three times (band split + peer-copy gather +
    assemble kernel on real hardware); with distinct devices the gather is RCCL send/recv.  Image == the single-device image."""
    import torch
    for label, sampling in (("bench256_view5_default", vr.SAMPLE_NEAREST), ("window_199x178_view1", vr.SAMPLE_TRILINEAR)):
        case = [c for c in golden.cases(True) if c["label"] == label][0]
        st = load_volume(gpu, golden, case["volume"])
        p = golden.params(case, sampling)
        gpu.set_window_buffer(p.view.width, p.view.height)
        want = gpu.render_volume(p)
        for devices in ([0], [0, 0], [0, 0, 0]):
            m = vr.MultiRenderer(devices)
            try:
                assert m.transport == ("single" if len(devices) == 1 else "peer-copy")
                m.set_window_buffer(p.view.width, p.view.height)
                m.set_transfer_fn(st["tf"], st["esl"])
                m.set_volume(golden.voxels(case["volume"]))
                assert np.array_equal(m.render_volume(p), want), (label, devices)
                dev = torch.full((p.view.height, p.view.width, 4), 9, dtype=torch.uint8, device="cuda:0")
                torch.cuda.synchronize()
                m.render_volume_device(p, dev.data_ptr())
                assert np.array_equal(dev.cpu().numpy(), want), (label, devices)
                per, total = m.timing()
                assert len(per) == len(devices) and all(x > 0 for x in per) and total > 0
            finally:
                m.close()


def test_multi_device_pipeline_rccl_self_and_selfcheck(vr, gpu, golden, monkeypatch):
    """The parts of the several-GPU path that one GPU can execute:
    * three frames in flight (vr_hip_multi_render_device_async / _sync): 8 different frames queued back to back into 8 buffers —
      every one equals the single-device frame, so no band buffer of frame i was overwritten by frames i+1 … i+3;
    * VR_MULTI_TRANSPORT=rccl-self: the bands travel through ncclSend / ncclRecv (one communicator, peer = self) — dlopen of
      librccl, ncclCommInitAll, the grouped calls and their stream ordering run on hardware;
    * VR_MULTI_SELFCHECK=1: the first-frame self-check (device 0's own render of the other ranks' bands against what arrived)
      passes on a correct gather."""
    import torch
    case = [c for c in golden.cases(True) if c["label"] == "bench256_view5_default"][0]
    st = load_volume(gpu, golden, case["volume"])
    W = H = 256
    gpu.set_window_buffer(W, H)
    scene_params = [golden.params(case, s) for s in (vr.SAMPLE_NEAREST, vr.SAMPLE_TRILINEAR)]
    frames = []
    for i in range(8):                                    # 8 different frames: 4 views x 2 sampling modes of the same scene
        p = scene_params[i & 1].copy()
        p.view = vr.benchmark_view(W, H, (i >> 1) + 4 * (i & 1))
        frames.append((p, gpu.render_volume(vr.whole_frame(p.copy()))))
    for transport, selfcheck, devices in (("peer", "1", [0, 0, 0]), ("rccl-self", "1", [0, 0]), ("rccl-self", "0", [0, 0, 0, 0])):
        monkeypatch.setenv("VR_MULTI_TRANSPORT", transport)
        monkeypatch.setenv("VR_MULTI_SELFCHECK", selfcheck)
        m = vr.MultiRenderer(devices)
        try:
            assert m.transport == ("peer-copy" if transport == "peer" else "rccl-self")
            m.set_window_buffer(W, H)
            m.set_transfer_fn(st["tf"], st["esl"])
            m.set_volume(golden.voxels(case["volume"]))
            m.prepare()
            outs = [torch.full((H, W, 4), 7, dtype=torch.uint8, device="cuda:0") for _ in frames]
            torch.cuda.synchronize()
            consumer = torch.cuda.Stream()
            for (p, _), out in zip(frames, outs):
                m.render_volume_device_async(p, out.data_ptr(), consumer.cuda_stream)
            with torch.cuda.stream(consumer):             # ordered behind the LAST frame by consumer_stream alone, no host sync yet
                last = outs[-1].clone()
            m.sync()
            consumer.synchronize()
            for i, ((_, want), out) in enumerate(zip(frames, outs)):
                assert np.array_equal(out.cpu().numpy(), want), (transport, devices, i)
            assert np.array_equal(last.cpu().numpy(), frames[-1][1])
            per, total = m.timing()
            assert all(x > 0 for x in per) and total > 0
        finally:
            m.close()


def test_volume_info_and_release_of_the_linear_copy(vr, golden):
    """Brick copies are built on first use (or by vr_hip_prepare) and vr_hip_volume_info reports them; after
    vr_hip_release_linear_copy rendering is unchanged and everything that needs the linear array says so instead of reading
    freed memory — including a frame that no resident copy can serve (NEAREST on the index-arithmetic path)."""
    r = vr.HipRenderer(0)
    try:
        st = golden.volume_state("bucky")
        r.set_transfer_fn(st["tf"], st["esl"])
        r.set_volume(golden.voxels("bucky"))
        r.set_window_buffer(256, 256)
        info = r.volume_info()
        assert (info.dim_x, info.dim_y, info.dim_z, info.bytes_per_voxel) == (32, 32, 32, 1)
        # nothing but the linear array after set_volume; the policy has all six copies at this size
        assert info.layout == vr.LAYOUT_BRICKED and info.copies == 0 and info.bricked_bytes == 0 and info.copies_in_policy == vr.COPY_ALL & ~vr.COPY_OCT
        assert info.brick_copies == 0 and info.brick_copies_wanted == 3 and info.linear_resident == 1 and info.linear_bytes >= 32 ** 3
        with pytest.raises(vr.VrError) as e:
            r.release_linear_copy()                                            # no brick copy resident yet
        assert e.value.code == 1
        case = [c for c in golden.cases(True) if c["label"] == "bench256_view1_default"][0]
        # a NEAREST frame builds the voxel bricks and nothing else (a NEAREST-only session)
        near = r.render_volume(golden.params(case, vr.SAMPLE_NEAREST))
        info = r.volume_info()
        assert info.copies == vr.COPY_VOXEL and info.run_copy == 4 and info.bricked_bytes == 32 ** 3 and info.build_ms[5] > 0
        # view 1 is oblique: a TRILINEAR frame builds one run copy, not the quad planes
        tri = r.render_volume(golden.params(case, vr.SAMPLE_TRILINEAR))
        info = r.volume_info()
        assert info.copies in (vr.COPY_VOXEL | vr.COPY_RUN_Z, vr.COPY_VOXEL | vr.COPY_RUN_Y) and info.brick_copies == 0
        r.prepare()                                                            # everything the policy has
        info = r.volume_info()
        assert info.copies == vr.COPY_ALL & ~vr.COPY_OCT and info.brick_copies == info.brick_copies_wanted == 3 and info.brick_planes == 7 and info.run_copy == 7
        assert info.bricked_bytes == 3 * 4 * 32 ** 3 + 2 * (4 * 4 * 4 * 2304 + 16) + 32 ** 3 + 3 * (8 * 8 * 11 * 256) + 3 * (8 * 8 * 2 * 256)    # + the column-window copies (quad elements, voxels)
        assert all(ms > 0 for ms in list(info.build_ms)[:6]) and info.build_ms[6] == 0 and all(ms > 0 for ms in list(info.build_ms)[7:13])
        assert info.copies_refused == 0 and info.upload_ms > 0
        before = [near, tri]
        r.release_linear_copy()
        info = r.volume_info()
        assert info.linear_resident == 0 and info.linear_bytes == 0 and info.brick_copies == 3
        after = [r.render_volume(golden.params(case, m)) for m in (vr.SAMPLE_NEAREST, vr.SAMPLE_TRILINEAR)]
        assert all(np.array_equal(a, b) for a, b in zip(before, after))
        assert np.array_equal(after[0], golden.frame(case))
        for call in (r.volume_minmax, r.volume_histogram, r.download_volume, lambda: r.set_layout(vr.LAYOUT_LINEAR),
                     lambda: r.set_wide_addressing(1)):
            with pytest.raises(vr.VrError) as e:
                call()
            assert e.value.code == 5 and "released" in str(e.value)            # VR_ERR_NOT_READY
        r.set_wide_addressing(2)                                               # the 64-bit TABLE path reads brick copies: still fine
        assert np.array_equal(r.render_volume(golden.params(case, vr.SAMPLE_NEAREST)), near)
        r.set_wide_addressing(0)
        # a context that released the linear array after preparing ONLY the TRILINEAR copies: NEAREST falls back to the quad copy
        r.set_volume(golden.voxels("bucky"))
        r.prepare(vr.COPY_QUAD_XY | vr.COPY_RUN_Z)
        r.release_linear_copy()
        assert np.array_equal(r.render_volume(golden.params(case, vr.SAMPLE_NEAREST)), near)
        assert np.array_equal(r.render_volume(golden.params(case, vr.SAMPLE_TRILINEAR)), tri)
        assert r.volume_info().copies == vr.COPY_QUAD_XY | vr.COPY_RUN_Z
        with pytest.raises(vr.VrError) as e:
            r.prepare(vr.COPY_VOXEL)
        assert e.value.code == 5
        # ... and one that holds only the voxel bricks cannot serve TRILINEAR: refused, not a NULL volume pointer in a kernel
        r.set_volume(golden.voxels("bucky"))
        r.prepare(vr.COPY_VOXEL)
        r.release_linear_copy()
        assert np.array_equal(r.render_volume(golden.params(case, vr.SAMPLE_NEAREST)), near)
        with pytest.raises(vr.VrError) as e:
            r.render_volume(golden.params(case, vr.SAMPLE_TRILINEAR))
        assert e.value.code == 5 and "released" in str(e.value)
        with pytest.raises(vr.VrError) as e:
            r.set_wide_addressing(1)
        assert e.value.code == 5
        r.set_volume(golden.voxels("bucky"))                                   # a new volume brings everything back
        assert r.volume_info().linear_resident == 1 and r.volume_minmax()[1] == 8
        r.set_wide_addressing(1)
        r.prepare(vr.COPY_VOXEL)
        with pytest.raises(vr.VrError) as e:
            r.release_linear_copy()                                            # the forced index-arithmetic path reads the linear array
        assert e.value.code == 1
        r.set_wide_addressing(0)
        r.set_layout(vr.LAYOUT_LINEAR)
        with pytest.raises(vr.VrError) as e:
            r.release_linear_copy()                                            # the linear array is the only copy now
        assert e.value.code == 1
    finally:
        r.close()


def test_long_thin_volume_axis_aligned_zero_direction(vr, gpu, oracle):
    """An exactly-zero direction component makes intersect() substitute 1e-5 (RaycasterBase.h:33-35): rays whose origin lies up
    to ky * 1e-5 outside a face still count as hits and march at that out-of-cube coordinate — with an edge of 40000 voxels that
    is more than one texel, so the host must clamp the fetch coordinates (clamp_fetch).  GPU == oracle, both sampling modes."""
    n = (40000, 8, 8)
    rng = np.random.default_rng(7)
    vox = rng.integers(0, 256, size=(n[2], n[1], n[0]), dtype=np.uint8)
    tf, esl, bd, bs, step = oracle.scene_for(vox)
    gpu.set_transfer_fn(tf, esl)
    gpu.set_volume(vox)
    gpu.set_window_buffer(96, 64)
    for persp in (0, 1):
        for angles in ((0.0, 0.0, 0.0), (90.0, 0.0, 0.0), (0.0, 90.0, 0.0)):
            v = vr.scene.custom_view(96, 64, persp, angles, 2.5)
            if not persp and angles == (0.0, 0.0, 0.0):
                # put pixel column 10 just OUTSIDE the x = -1 face, by 3.2e-5 < ky * 1e-5 = 3.5e-5: reported as a hit for the last
                # steps of the march, at texel coordinate -1.14 along the 40000-voxel edge
                assert v.direction[0] == 0.0 and v.right_plane[1] == 0.0
                v.origin[0] = float(np.float32(-1.0 - 3.2e-5) - np.float32(v.right_plane[0]) * np.float32(10 - 48))
            for samp in (vr.SAMPLE_NEAREST, vr.SAMPLE_TRILINEAR):
                p = vr.VrParams()
                p.view = v
                p.ray_step, p.ray_threshold, p.light_kd = float(step) * 400.0, 0.95, 0.6
                p.esl, p.esl_block_dims, p.sampling = 1, bd, samp
                for j in range(3):
                    p.esl_block_size[j] = float(bs[j])
                p = vr.whole_frame(p)
                out = gpu.render_volume(p)
                ref = oracle.render(p, vox, tf, esl, threads=16)
                assert compare_frames(out, ref) == (0, 0), (persp, angles, samp)


def test_cpp_renderer_mirror(vr, gpu, golden):
    """The host C++ mirror: RaycasterBase::reset_transfer_fn/set_volume -> HipRenderer(raycaster).render_volume()."""
    vox = golden.voxels("bucky")
    scene = vr.Scene().set_volume(voxels=vox)
    scene.set_modes(esl=True, ray_threshold=0.95, light_kd=0.6)     # RaycasterBase state is process-global, like the reference's
    st = golden.volume_state("bucky")
    assert np.array_equal(scene.tf, st["tf"]) and np.array_equal(scene.esl, st["esl"])
    for label in ("bench256_view2_default", "bench256_view7_default"):
        case = [c for c in golden.cases(True) if c["label"] == label][0]
        p = golden.params(case)
        out = np.zeros((256, 256, 4), np.uint8)
        rc = vr.lib().vr_host_render_frame(0, vr.SAMPLE_NEAREST, C.byref(p.view), out.ctypes.data)
        assert rc == 0
        assert np.array_equal(out, golden.frame(case)), label


def test_error_conventions(vr, golden):
    """0 ok / non-zero failure, never exit (reference: CPURenderer.cpp:44-45, GPURenderer1.cu:91-95,101-102)."""
    L = vr.lib()
    r = vr.HipRenderer(0)
    case = golden.cases(True)[0]
    p = golden.params(case)
    buf = np.zeros((p.out_rows, p.out_width, 4), np.uint8)
    assert L.vr_hip_render(r._ctx, C.byref(p), buf.ctypes.data) == 5           # VR_ERR_NOT_READY: nothing set yet
    assert L.vr_hip_set_volume(r._ctx, None, 32, 32, 32, 1) == 1               # NULL data -> 1 like the reference
    assert L.vr_hip_set_volume(r._ctx, buf.ctypes.data, 0, 32, 32, 1) == 1
    assert L.vr_hip_set_volume(r._ctx, buf.ctypes.data, 32, 32, 32, 3) == 1
    assert L.vr_hip_set_transfer_fn(r._ctx, None, None) == 1
    st = golden.volume_state("bucky")
    r.set_transfer_fn(st["tf"], st["esl"])
    r.set_volume(golden.voxels("bucky"))
    assert L.vr_hip_render(r._ctx, C.byref(p), None) == 1                      # NULL buffer -> 1
    assert L.vr_hip_render(r._ctx, None, buf.ctypes.data) == 1
    assert L.vr_hip_render(r._ctx, C.byref(p), buf.ctypes.data) == 5           # window buffer not set
    r.set_window_buffer(p.view.width, p.view.height)
    bad = p.copy(); bad.ray_step = 0.0
    assert L.vr_hip_render(r._ctx, C.byref(bad), buf.ctypes.data) == 1         # would never terminate
    bad = p.copy(); bad.view.origin[0] = float("nan")
    assert L.vr_hip_render(r._ctx, C.byref(bad), buf.ctypes.data) == 1
    bad = p.copy(); bad.sampling = 9
    assert L.vr_hip_render(r._ctx, C.byref(bad), buf.ctypes.data) == 1
    bad = p.copy(); bad.band_rows = 0
    assert L.vr_hip_render(r._ctx, C.byref(bad), buf.ctypes.data) == 1
    assert b"band" in L.vr_hip_last_error(r._ctx)
    assert L.vr_hip_render(r._ctx, C.byref(p), buf.ctypes.data) == 0           # and the context still works
    assert np.array_equal(buf, golden.frame(case))
    # policy switches and the profiling read-back: out-of-range values and premature calls are refused, the context keeps working
    tx, ty = C.c_uint32(7), C.c_uint32(7)
    assert L.vr_hip_read_tile_costs(r._ctx, None, 0, C.byref(tx), C.byref(ty)) == 5      # no frame with the cost map on yet
    assert L.vr_hip_set_tile_scheduling(r._ctx, 3) == 1 and L.vr_hip_set_brick_plane(r._ctx, 10) == 1 and L.vr_hip_set_brick_plane(r._ctx, -2) == 1
    assert L.vr_hip_set_tile_mapping(r._ctx, 16, 0, 0) == 1 and L.vr_hip_set_tile_mapping(r._ctx, 3, 0, 0) == 1 and L.vr_hip_set_tile_mapping(r._ctx, 0, 8, 0) == 1
    assert L.vr_hip_last_launch(r._ctx, None) == 1
    assert L.vr_hip_set_tile_scheduling(r._ctx, 2) == 0
    assert L.vr_hip_render(r._ctx, C.byref(p), buf.ctypes.data) == 0 and np.array_equal(buf, golden.frame(case))
    assert L.vr_hip_read_tile_costs(r._ctx, None, 0, C.byref(tx), C.byref(ty)) == 0 and tx.value * ty.value >= 1
    small = np.zeros(max(1, tx.value * ty.value - 1), np.uint32)
    assert L.vr_hip_read_tile_costs(r._ctx, small.ctypes.data, small.size if tx.value * ty.value > 1 else 0, C.byref(tx), C.byref(ty)) == 1   # buffer too small
    assert L.vr_hip_set_tile_scheduling(r._ctx, 1) == 0
    ctx = C.c_void_p()
    assert L.vr_hip_create(99, C.byref(ctx)) == 2                              # VR_ERR_NO_DEVICE
    r.close()


def test_feeders_minmax_histogram_generate(vr, gpu, golden, oracle):
    """GPU feeders against the CPU restatement of RaycasterBase::set_volume / compute_histogram."""
    for name in ("bucky", "blob_40x24x56", "shell48"):
        vox = golden.voxels(name)
        gpu.set_volume(vox)
        mm, bd, bs, _ = gpu.volume_minmax()
        omm, obd, obs = oracle.volume_minmax(vox)
        assert bd == obd and np.array_equal(np.array(bs, np.float32), obs)
        assert np.array_equal(mm, omm), name
        h, _ = gpu.volume_histogram()
        assert np.array_equal(h, oracle.histogram(vox)), name
        # TF/ESL built by the host mirror from the GPU min/max == what the reference built (golden)
        st = golden.volume_state(name)
        scene = vr.Scene().set_volume(dims=(vox.shape[2], vox.shape[1], vox.shape[0]), minmax=mm)
        if st["base_tf"] is not None:
            scene.set_base_transfer_fn(st["base_tf"])
        assert np.array_equal(scene.esl, st["esl"]) and np.array_equal(scene.tf, st["tf"]), name
        assert scene.params.esl_block_dims == st["esl_block_dims"]
        assert np.float32(scene.params.ray_step) == st["ray_step"]
    # streaming (16-byte) path of the reduction: 512^3 -> block 16; u16 too
    for n, bpv, kind in ((512, 1, "shell"), (256, 2, "noise"), (320, 1, "noise")):
        gpu.generate_volume(kind, n, seed=3, bytes_per_voxel=bpv)
        vox = gpu.download_volume()
        assert np.array_equal(vox, oracle.generate_volume(kind, n, 3, bpv)), (n, bpv, kind)
        mm, bd, bs, _ = gpu.volume_minmax()
        omm, obd, obs = oracle.volume_minmax(vox)
        assert bd == obd and np.array_equal(mm, omm), (n, bpv, kind)
        h, _ = gpu.volume_histogram()
        assert np.array_equal(h, oracle.histogram(vox))


def test_host_buffer_slices_equal_one_launch(vr):
    """vr_hip_render (the reference's host-buffer entry point) renders a frame of unpartitioned rows as two row slices on two streams so
    that the first slice's copy to the host overlaps the second slice (VR_HOST_SLICES=1: one launch + one copy).  Same bytes either way,
    for a frame whose rows are no multiple of the tile height, in a leaping and in the full-march mode, from an oblique and an axis-aligned
    pose (general and column kernels), and equal to the device-pointer entry point."""
    import os
    import subprocess
    import sys
    from helpers import ROOT
    child = r"""
import importlib, hashlib, os, sys
sys.path.insert(0, %(root)r)
import torch
vr = importlib.import_module("volume-rendering_amd")
r = vr.HipRenderer(0)
r.generate_volume("shell", 192, seed=3)
scene = vr.Scene().set_volume(dims=(192, 192, 192), minmax=r.volume_minmax()[0])
r.set_transfer_fn(scene.tf, scene.esl)
W, H = 1000, 777
r.set_window_buffer(W, H)
dev = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda:0")
for mode in ("default", "nooptims"):
    scene.set_modes(esl=(mode == "default"), ray_threshold=(0.95 if mode == "default" else 1.0), light_kd=0.6)
    for view in (0, 1, 5):
        for samp in (vr.SAMPLE_TRILINEAR, vr.SAMPLE_NEAREST):
            p = scene.frame_params(vr.benchmark_view(W, H, view), samp)
            host = r.render_volume(p)
            r.render_volume_device(p, dev.data_ptr())
            torch.cuda.synchronize()
            assert (dev.cpu().numpy() == host).all(), (mode, view, samp)
            print(mode, view, samp, hashlib.sha1(host.tobytes()).hexdigest(), int((host[..., 3] != 0).sum()))
"""
    outs = []
    for slices in ("1", "2"):
        r = subprocess.run([sys.executable, "-c", child % {"root": ROOT}], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, VR_HOST_SLICES=slices))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs.append(r.stdout)
    assert outs[0] == outs[1] and outs[0].count("\n") == 12
    assert all(int(line.split()[-1]) > 1000 for line in outs[0].strip().splitlines())     # the frames are not empty
