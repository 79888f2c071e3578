"""Volume file I/O of the host mirror (ModelBase::load_model, PVM / DDS decode, 16 -> 8 bit quantise)
against what the reference's own loader produced from the same files (tests/golden/golden_io.npz, made by
oracle/gen_golden_io.py) and against the reference's dataset Bucky.pvm (decoded voxels FNV-1a32 70f1ecd5)."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR, fnv1a32


def load(vr, path):
    dims = (C.c_uint32 * 3)()
    rc = vr.lib().vr_host_load_model(path.encode(), dims)
    if rc:
        return rc, None, None
    x, y, z = dims
    p = vr.lib().vr_host_model_voxels()
    vox = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(z, y, x)).copy()
    hist = np.zeros(256, np.float32)
    vr.lib().vr_host_model_histogram(hist.ctypes.data_as(C.POINTER(C.c_float)))
    return 0, vox, hist


def test_bucky_pvm_decodes_to_the_reference_voxels(vr, golden):
    rc, vox, hist = load(vr, os.path.join(GOLDEN_DIR, "Bucky.pvm"))
    assert rc == 0 and vox.shape == (32, 32, 32)
    assert fnv1a32(vox) == "70f1ecd5"
    assert np.array_equal(vox, golden.voxels("bucky"))
    assert hist.max() == 1.0 and (hist >= 0).all()


def test_pvm_variants_match_reference_loader(vr, tmp_path):
    io = np.load(os.path.join(GOLDEN_DIR, "golden_io.npz"), allow_pickle=False)
    n = int(io["count"][0])
    assert n >= 5
    for i in range(n):
        name = bytes(io[f"io{i}_name"]).decode()
        path = tmp_path / (name + ".pvm")
        path.write_bytes(bytes(io[f"io{i}_file"]))
        rc, vox, hist = load(vr, str(path))
        assert rc == 0, name
        assert np.array_equal(vox, io[f"io{i}_voxels"]), name          # incl. the non-linear 16 -> 8 bit quantisation
        assert np.array_equal(hist, io[f"io{i}_hist"]), name


def test_quantize_linear(vr):
    io = np.load(os.path.join(GOLDEN_DIR, "golden_io.npz"), allow_pickle=False)
    src = np.ascontiguousarray(io["quant_linear_in"])
    out = np.zeros(24 * 18 * 10, np.uint8)
    assert vr.lib().vr_host_quantize(src.ctypes.data, 24, 18, 10, 1, out.ctypes.data) == 0
    assert np.array_equal(out, io["quant_linear_out"])


def test_raw_and_error_paths(vr, tmp_path):
    vol = (np.arange(6 * 5 * 4) % 251).astype(np.uint8)
    p = tmp_path / "v.raw"
    p.write_bytes(vol.tobytes())
    vr.lib().vr_host_set_raw_dims(6, 5, 4, 1)
    rc, vox, _ = load(vr, str(p))
    assert rc == 0 and np.array_equal(vox.reshape(-1), vol)
    vr.lib().vr_host_set_raw_dims(6, 5, 5, 1)                   # "Incorrect RAW file volume parameters" -> 1
    assert load(vr, str(p))[0] == 1
    assert load(vr, str(tmp_path / "missing.pvm"))[0] == 1       # file not found -> 1 (ModelBase.cpp:63-66)
    bad = tmp_path / "v.txt"
    bad.write_bytes(b"x")
    assert load(vr, str(bad))[0] == 1                            # unsupported extension -> 1 (ModelBase.cpp:38-41)
    junk = tmp_path / "junk.pvm"
    junk.write_bytes(b"not a volume at all")
    assert load(vr, str(junk))[0] == 1
    three = tmp_path / "three.pvm"
    three.write_bytes(b"PVM\n2 2 2\n3\n" + bytes(24))
    assert load(vr, str(three))[0] == 1                          # components > 2 unsupported (ModelBase.cpp:90-94)
