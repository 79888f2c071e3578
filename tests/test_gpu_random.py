"""Randomised parity: random volumes (odd dimensions, u8 / u16), random transfer functions, random views (camera inside
and outside the cube, axis-aligned directions with exact zeros, orthogonal and perspective), random ray step / threshold /
light / ESL — the HIP path must equal the CPU oracle bit for bit in all three sampling modes and both layouts."""
import os

import numpy as np
import pytest

from helpers import compare_frames

pytestmark = pytest.mark.gpu


def random_scene(rng, oracle, vr):
    dims = [int(rng.integers(3, 40)) for _ in range(3)]                      # x, y, z
    if rng.random() < 0.3:
        dims[int(rng.integers(0, 3))] = int(rng.choice([1, 2, 8, 16, 33]))
    x, y, z = dims
    zz, yy, xx = np.mgrid[0:z, 0:y, 0:x]
    field = 255.0 * np.exp(-(((xx - x * rng.random()) / (0.2 + x * 0.4)) ** 2 + ((yy - y * rng.random()) / (0.2 + y * 0.4)) ** 2 +
                             ((zz - z * rng.random()) / (0.2 + z * 0.4)) ** 2))
    vox = np.clip(field + rng.integers(0, 20, field.shape), 0, 255).astype(np.uint8)
    if rng.random() < 0.3:
        vox = vox.astype(np.uint16) * 257 + rng.integers(0, 200, field.shape).astype(np.uint16)
    base = oracle.default_base_tf()
    if rng.random() < 0.6:
        base = rng.random((128, 4)).astype(np.float32)
        base[:, 3] *= (rng.random(128) < 0.7)                                 # holes of zero opacity
        base[: int(rng.integers(0, 40)), 3] = 0
    tf, esl, bd, bs, ray_step = oracle.scene_for(vox, base)
    return vox, tf, esl, bd, bs, ray_step


def random_params(rng, vr, bd, bs, ray_step, sampling):
    p = vr.VrParams()
    w, h = int(rng.integers(1, 70)), int(rng.integers(1, 50))
    p.view.width, p.view.height = w, h
    persp = int(rng.random() < 0.5)
    p.view.perspective = persp
    kind = rng.integers(0, 4)
    if kind == 0:                                   # axis-aligned, exact zeros in the direction
        axis, sign = int(rng.integers(0, 3)), float(rng.choice([-1.0, 1.0]))
        d = np.zeros(3, np.float32); d[axis] = sign
        o = (-d * np.float32(rng.uniform(0.0, 3.0))).astype(np.float32)
        r = np.zeros(3, np.float32); r[(axis + 1) % 3] = 1
        u = np.zeros(3, np.float32); u[(axis + 2) % 3] = 1
    else:
        d = rng.normal(size=3).astype(np.float32); d /= np.linalg.norm(d)
        o = (-d * np.float32(rng.uniform(0.05, 3.0)) + rng.normal(scale=0.2, size=3)).astype(np.float32)
        r = np.cross(d, rng.normal(size=3)).astype(np.float32); r /= np.linalg.norm(r)
        u = np.cross(r, d).astype(np.float32)
    pitch = np.float32(rng.uniform(1.0, 3.0) / min(w, h))
    for j in range(3):
        p.view.origin[j], p.view.direction[j] = float(o[j]), float(d[j])
        p.view.right_plane[j], p.view.up_plane[j] = float(r[j] * pitch), float(u[j] * pitch)
        p.view.light_pos[j] = float(rng.normal(scale=2.0))
    p.ray_step = float(np.float32(ray_step) * np.float32(rng.uniform(0.34, 1.66)))
    if rng.random() < 0.12:
        p.ray_step *= 4.0                            # two steps leave the padding of the address tables: the clamping variant must take over
    p.ray_threshold = float(rng.choice([0.5, 0.8, 0.95, 1.0]))
    p.esl = int(rng.random() < 0.6)
    p.esl_block_dims = bd
    for j in range(3):
        p.esl_block_size[j] = float(bs[j])
    p.light_kd = float(rng.choice([0.0, 0.6, 1.3, 2.0]))
    p.sampling = sampling
    return vr.whole_frame(p)


def test_random_scenes_match_oracle(vr, gpu, oracle):
    rng = np.random.default_rng(int(os.environ.get("VR_TEST_SEED", "20261004")))     # other seeds: VR_TEST_SEED=... pytest -m gpu -k random
    gpu.set_window_buffer(70, 50)
    checked = nonempty = dual_frames = 0
    try:
        for scene_i in range(14):
            vox, tf, esl, bd, bs, ray_step = random_scene(rng, oracle, vr)
            gpu.set_layout(vr.LAYOUT_BRICKED if scene_i % 3 else vr.LAYOUT_LINEAR)
            gpu.set_wide_addressing({4: 1, 3: 2, 2: 8}.get(scene_i % 5, 0))     # + 8: no scaled-domain NEAREST
            gpu.set_transfer_fn(tf, esl)
            gpu.set_volume(vox)
            for frame_i in range(6):
                for sampling in (vr.SAMPLE_NEAREST, vr.SAMPLE_TRILINEAR if frame_i % 2 == 0 else vr.SAMPLE_TRILINEAR_Q8):
                    p = random_params(rng, vr, bd, bs, ray_step, sampling)
                    out = gpu.render_volume(p)
                    ref = oracle.render(p, vox, tf, esl, threads=4)
                    ndiff, maxd = compare_frames(out, ref)
                    assert ndiff == 0, (f"scene {scene_i} dims {vox.shape} {vox.dtype} sampling {sampling} persp {p.view.perspective} "
                                        f"esl {p.esl}: {ndiff} px differ, max delta {maxd}")
                    checked += 1
                    nonempty += int((out[..., 3] != 0).any())
            # the full march of an ORTHOGONAL view that is not along an axis: six frames with the same parameters take the launch through
            # the run copy along z, along y, the two recording frames and the per-block choice between the two copies (1-byte voxels,
            # bricked layout, table addressing; otherwise the frames simply repeat)
            p = random_params(rng, vr, bd, bs, ray_step, vr.SAMPLE_TRILINEAR if scene_i % 2 else vr.SAMPLE_TRILINEAR_Q8)
            for _ in range(20):
                if not p.view.perspective and min(abs(p.view.direction[j]) for j in range(3)) > 0.05:
                    break
                p = random_params(rng, vr, bd, bs, ray_step, p.sampling)
            p.esl, p.ray_threshold = 0, 1.0
            p.ray_step = float(np.float32(ray_step))                                   # (a four-fold step would select the clamping variant)
            p.view.width, p.view.height = 64, 48                                     # 2 x 3 workgroup tiles at least
            pitch = np.float32(2.0 / 48)
            for j in range(3):
                n_r = float(np.sqrt(sum(p.view.right_plane[i] ** 2 for i in range(3)))) or 1.0
                n_u = float(np.sqrt(sum(p.view.up_plane[i] ** 2 for i in range(3)))) or 1.0
                p.view.right_plane[j] = float(p.view.right_plane[j] / n_r * pitch)
                p.view.up_plane[j] = float(p.view.up_plane[j] / n_u * pitch)
            vr.whole_frame(p)
            if vox.dtype == np.uint8:
                gpu.set_layout(vr.LAYOUT_BRICKED)
                gpu.set_wide_addressing(0)
            ref = oracle.render(p, vox, tf, esl, threads=4)
            layouts = []
            for frame_i in range(6):
                ndiff, maxd = compare_frames(gpu.render_volume(p), ref)
                layouts.append(gpu.last_launch()["layout"])
                assert ndiff == 0, f"scene {scene_i} dims {vox.shape} {vox.dtype} repeated frame {frame_i} layouts {layouts}: {ndiff} px differ, max delta {maxd}"
            dual_frames += int(layouts[-1] == 6)
    finally:
        gpu.set_layout(vr.LAYOUT_BRICKED)
        gpu.set_wide_addressing(False)
    assert checked == 14 * 12
    assert dual_frames >= 4, dual_frames               # the 1-byte scenes did reach the per-block copy choice
    assert nonempty >= checked // 2, nonempty          # the random views do look at the volume
