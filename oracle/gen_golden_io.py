#!/usr/bin/env python3
"""Generates tests/golden/golden_io.npz: volume FILES (PVM v1/v2/v3, DDS v3d/v3e compressed or plain, 8 and 16 bit) with
the voxels and histogram the REFERENCE's own loader (ModelBase::load_model -> ddsbase.cpp, oracle/_ref) produces from
them.  TEST INFRASTRUCTURE; run in the build container only.  The DDS encoder below exists only to make test inputs (the
reference ships a decoder only); it is validated by the reference decoding its output back to the source volume."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden import Ref, fnv1a32      # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


class BitWriter:
    def __init__(self):
        self.bits = []

    def write(self, value, n):
        for i in range(n - 1, -1, -1):
            self.bits.append((value >> i) & 1)

    def tobytes(self):
        b = self.bits + [0] * (-len(self.bits) % 8)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


def dds_encode(data, skip, strip, ident=b"DDS v3d\n"):
    d = np.frombuffer(bytes(data), np.uint8)
    d = np.concatenate([d[i::skip] for i in range(skip)]).astype(np.int64)       # de-interleave (whole stream)
    w = BitWriter()
    w.write(skip - 1, 2)
    w.write(strip - 1, 16)
    deltas, act = [], 0
    for cnt, v in enumerate(d):
        pred = act if (strip == 1 or cnt <= strip) else act + d[cnt - strip] - d[cnt - strip - 1]
        deltas.append(int((v - pred + 128) % 256) - 128)
        act = int(v)
    i = 0
    while i < len(deltas):
        run = deltas[i:i + 127]
        for b in (0, 2, 3, 4, 5, 6, 7, 8):
            half = (1 << b) // 2
            if all(0 <= x + half < (1 << b) or (b == 0 and x == 0) for x in run):
                break
        w.write(len(run), 7)
        w.write(0 if b == 0 else b - 1, 3)
        for x in run:
            w.write(x + (1 << b) // 2, b)
        i += len(run)
    w.write(0, 7)
    return ident + w.tobytes()


def blob(w, h, d, seed, sixteen=False):
    rng = np.random.RandomState(seed)
    z, y, x = np.mgrid[0:d, 0:h, 0:w]
    f = np.exp(-(((x - w * 0.45) / (w * 0.3)) ** 2 + ((y - h * 0.5) / (h * 0.35)) ** 2 + ((z - d * 0.55) / (d * 0.3)) ** 2))
    if sixteen:
        v = np.clip(f * 3800 + rng.randint(0, 40, f.shape), 0, 4095).astype(">u2")      # 12-bit CT-like data, big-endian
        return v
    return np.clip(f * 250 + rng.randint(0, 6, f.shape), 0, 255).astype(np.uint8)


def main():
    ref = Ref()
    ref.L.volr_ref_get_histogram.argtypes = [C.c_void_p]
    arrays, n = {}, 0

    def add(name, file_bytes, expect_dims):
        nonlocal n
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, name + ".pvm")
            with open(path, "wb") as f:
                f.write(file_bytes)
            vox = ref.load_model(path)
        assert vox.shape == (expect_dims[2], expect_dims[1], expect_dims[0]), (vox.shape, expect_dims)
        hist = np.zeros(256, np.float32)
        ref.L.volr_ref_get_histogram(hist.ctypes.data)
        arrays[f"io{n}_file"] = np.frombuffer(file_bytes, np.uint8)
        arrays[f"io{n}_voxels"] = vox
        arrays[f"io{n}_hist"] = hist
        arrays[f"io{n}_name"] = np.frombuffer(name.encode(), np.uint8)
        n += 1
        return vox

    v = blob(20, 12, 9, 1)
    body = b"PVM\n# a comment line\n20 12 9\n1\n" + v.tobytes()
    got = add("v1_dds_strip_w", dds_encode(body, 1, 20), (20, 12, 9))
    assert np.array_equal(got, v)                              # validates the test encoder through the reference decoder

    v = blob(17, 23, 5, 2)
    body = b"PVM3\n17 23 5\n1 1.5 2\n1\n" + v.tobytes() + b"Synthetic blob\0nobody\0none\0made for tests\0"
    got = add("v3_dds_strip1", dds_encode(body, 1, 1), (17, 23, 5))
    assert np.array_equal(got, v)

    v = blob(16, 16, 16, 3)
    body = b"PVM2\n16 16 16\n1 1 1\n1\n" + v.tobytes()
    got = add("v2_plain", body, (16, 16, 16))
    assert np.array_equal(got, v)

    v16 = blob(24, 18, 10, 4, sixteen=True)
    body = b"PVM3\n24 18 10\n1 1 1\n2\n" + v16.tobytes() + b"\0\0\0\0"
    add("v3_16bit_dds_v3e_skip2", dds_encode(body, 2, 48, ident=b"DDS v3e\n"), (24, 18, 10))   # load_model quantises to 8 bit

    body = b"PVM\n24 18 10\n2\n" + v16.tobytes()
    add("v1_16bit_plain", body, (24, 18, 10))

    # quantize(), linear flavour, straight on the 16-bit samples
    raw = np.frombuffer(v16.tobytes(), np.uint8).copy()
    q = np.zeros(24 * 18 * 10, np.uint8)
    ref.L.volr_ref_quantize(raw.ctypes.data_as(C.c_void_p), 24, 18, 10, 1, q.ctypes.data_as(C.c_void_p))
    arrays["quant_linear_in"] = raw
    arrays["quant_linear_out"] = q
    arrays["count"] = np.array([n], np.int32)
    np.savez_compressed(os.path.join(OUT, "golden_io.npz"), **arrays)
    print("wrote", n, "file cases;", os.path.getsize(os.path.join(OUT, "golden_io.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
