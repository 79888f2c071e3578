"""Every brick-copy builder (vr_kernels.hip brick_strip_kernel: quad bricks per chunk plane, voxel bricks, oct bricks, run bricks
along z / y) against a host-side numpy construction of the layout vr_device.h defines, byte for byte — power-of-two and ragged
edges, 1- and 2-byte voxels.  The images only ever read elements inside the volume; this test also pins the zero fill outside it
and the clamped +1 neighbours at the upper faces."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# vr_device.h brick_bit: position of coordinate bit k of axis a inside the 9-bit element offset
BRICK_BITS = {("u8", 0): (0, 2, 5, 1, 3, 6, 4, 7, 8), ("u8", 1): (0, 2, 5, 4, 7, 8, 1, 3, 6), ("u8", 2): (4, 7, 8, 0, 2, 5, 1, 3, 6),
              ("u16", 0): (2, 5, 8, 1, 4, 7, 0, 3, 6)}


def _local_xyz(bits):
    o = np.arange(512)
    def collect(axis):
        return sum(((o >> bits[3 * axis + k]) & 1) << k for k in range(3))
    return collect(0), collect(1), collect(2)


def _padded(vox, pad):
    """volume with `pad` extra cells per axis: index clamped at the upper faces"""
    z, y, x = vox.shape
    iz = np.minimum(np.arange(z + pad), z - 1); iy = np.minimum(np.arange(y + pad), y - 1); ix = np.minimum(np.arange(x + pad), x - 1)
    return vox[np.ix_(iz, iy, ix)]


def _brick_grid(vox):
    z, y, x = vox.shape
    return (x + 7) // 8, (y + 7) // 8, (z + 7) // 8


def quad_copy(vox, plane):
    """[brick][local] -> the 2x2 (x,y) neighbourhood of slice z, 4 voxels (u8: one dword; u16: 8 bytes)"""
    z, y, x = vox.shape
    nbx, nby, nbz = _brick_grid(vox)
    bits = BRICK_BITS[("u8" if vox.dtype == np.uint8 else "u16", plane if vox.dtype == np.uint8 else 0)]
    lx, ly, lz = _local_xyz(bits)
    p = _padded(vox, 9)
    out = np.zeros((nbz, nby, nbx, 512, 4), vox.dtype)
    bz, by, bx = np.meshgrid(np.arange(nbz), np.arange(nby), np.arange(nbx), indexing="ij")
    X = bx[..., None] * 8 + lx; Y = by[..., None] * 8 + ly; Z = bz[..., None] * 8 + lz
    inside = (X < x) & (Y < y) & (Z < z)
    Xc, Yc, Zc = np.minimum(X, x + 7), np.minimum(Y, y + 7), np.minimum(Z, z + 7)
    for i, (dx, dy) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        out[..., i] = np.where(inside, p[Zc, Yc + dy, Xc + dx], 0)
    return out.reshape(-1).view(np.uint8)


def voxel_copy(vox):
    z, y, x = vox.shape
    nbx, nby, nbz = _brick_grid(vox)
    lx, ly, lz = _local_xyz(BRICK_BITS[("u8" if vox.dtype == np.uint8 else "u16", 0)])
    p = _padded(vox, 9)
    bz, by, bx = np.meshgrid(np.arange(nbz), np.arange(nby), np.arange(nbx), indexing="ij")
    X = bx[..., None] * 8 + lx; Y = by[..., None] * 8 + ly; Z = bz[..., None] * 8 + lz
    inside = (X < x) & (Y < y) & (Z < z)
    out = np.where(inside, p[np.minimum(Z, z + 7), np.minimum(Y, y + 7), np.minimum(X, x + 7)], 0).astype(vox.dtype)
    return out.reshape(-1).view(np.uint8)


def oct_copy(vox):
    z, y, x = vox.shape
    nbx, nby, nbz = _brick_grid(vox)
    lx, ly, lz = _local_xyz(BRICK_BITS[("u16", 0)])
    p = _padded(vox, 9)
    out = np.zeros((nbz, nby, nbx, 512, 8), vox.dtype)
    bz, by, bx = np.meshgrid(np.arange(nbz), np.arange(nby), np.arange(nbx), indexing="ij")
    X = bx[..., None] * 8 + lx; Y = by[..., None] * 8 + ly; Z = bz[..., None] * 8 + lz
    inside = (X < x) & (Y < y) & (Z < z)
    Xc, Yc, Zc = np.minimum(X, x + 7), np.minimum(Y, y + 7), np.minimum(Z, z + 7)
    for i, (dx, dy, dz) in enumerate(((0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1))):
        out[..., i] = np.where(inside, p[Zc + dz, Yc + dy, Xc + dx], 0)
    return out.reshape(-1).view(np.uint8)


def run_copy(vox, along_y):
    """[brick][cell column (2-D Morton over x and the other axis)][k = 0..8 along the run axis] -> quad element, 4 bytes"""
    z, y, x = vox.shape
    nbx, nby, nbz = _brick_grid(vox)
    cell = np.arange(64)
    lx = (cell & 1) | ((cell >> 1) & 2) | ((cell >> 2) & 4)
    lo = ((cell >> 1) & 1) | ((cell >> 2) & 2) | ((cell >> 3) & 4)
    k = np.arange(9)
    p = _padded(vox, 10)
    nbo, nbr = (nbz, nby) if along_y else (nby, nbz)
    dim_o, dim_r = (z, y) if along_y else (y, z)
    out = np.zeros((nbr, nbo, nbx, 64, 9, 4), np.uint8)
    br, bo, bx = np.meshgrid(np.arange(nbr), np.arange(nbo), np.arange(nbx), indexing="ij")
    X = (bx[..., None, None] * 8 + lx[:, None]) + 0 * k
    O = (bo[..., None, None] * 8 + lo[:, None]) + 0 * k
    R = np.minimum(br[..., None, None] * 8 + k + 0 * lx[:, None], dim_r - 1)
    inside = (X < x) & (O < dim_o)
    Xc, Oc = np.minimum(X, x + 8), np.minimum(O, dim_o + 8)
    for i, (dx, do) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        v = p[Oc + do, R, Xc + dx] if along_y else p[R, Oc + do, Xc + dx]
        out[..., i] = np.where(inside, v, 0)
    return out.reshape(-1)


def column_copy(vox, m):
    """column windows along axis m (vr_device.h kLayoutColumn): [block row bv][block bu][window w][column (v & 3, u & 3)] -> four quad elements
    along m (march index 3w .. 3w+3, clamped at Nm - 1), each the 2x2 (u,v) neighbourhood with the +1 neighbours clamped at the upper faces;
    (u, v) = the two other axes in increasing order.  Columns beyond the volume hold the clamped edge columns."""
    dims = vox.shape[::-1]                                   # (x, y, z)
    ua, va = (1 if m == 0 else 0), (1 if m == 2 else 2)
    nu, nv, nm = dims[ua], dims[va], dims[m]
    nbu, nbv, nw = (nu + 3) // 4, (nv + 3) // 4, (nm + 2) // 3
    bv, bu, w, cv, cu, j = np.meshgrid(np.arange(nbv), np.arange(nbu), np.arange(nw), np.arange(4), np.arange(4), np.arange(4), indexing="ij")
    u, v, e = bu * 4 + cu, bv * 4 + cv, np.minimum(w * 3 + j, nm - 1)
    out = np.zeros(u.shape + (4,), np.uint8)
    for i, (du, dv) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        idx = [None, None, None]
        idx[ua], idx[va], idx[m] = np.minimum(u + du, nu - 1), np.minimum(v + dv, nv - 1), e
        out[..., i] = vox[idx[2], idx[1], idx[0]]
    return out.reshape(-1)


def column_voxel_copy(vox, m):
    """NEAREST column windows along axis m: [block row bv][block bu][window w][column (v & 3, u & 3)] -> 16 consecutive voxels of the column
    (march index 16w .. 16w+15, clamped at Nm - 1); columns beyond the volume hold the clamped edge columns."""
    dims = vox.shape[::-1]
    ua, va = (1 if m == 0 else 0), (1 if m == 2 else 2)
    nu, nv, nm = dims[ua], dims[va], dims[m]
    nbu, nbv, nw = (nu + 3) // 4, (nv + 3) // 4, (nm + 15) // 16
    bv, bu, w, cv, cu, j = np.meshgrid(np.arange(nbv), np.arange(nbu), np.arange(nw), np.arange(4), np.arange(4), np.arange(16), indexing="ij")
    idx = [None, None, None]
    idx[ua], idx[va], idx[m] = np.minimum(bu * 4 + cu, nu - 1), np.minimum(bv * 4 + cv, nv - 1), np.minimum(w * 16 + j, nm - 1)
    return vox[idx[2], idx[1], idx[0]].astype(np.uint8).reshape(-1)


def _volume(shape, dtype, seed):
    rng = np.random.default_rng(seed)
    hi = 256 if dtype == np.uint8 else 65536
    return rng.integers(0, hi, size=shape, dtype=dtype)


@pytest.mark.parametrize("shape", [(64, 64, 64), (40, 24, 56), (9, 130, 17), (8, 8, 264), (3, 301, 2)])      # (z, y, x)
def test_u8_copies_equal_the_host_construction(vr, gpu, shape):
    vox = _volume(shape, np.uint8, 7)
    gpu.set_layout(vr.LAYOUT_BRICKED)
    gpu.set_volume(vox)
    gpu.prepare(vr.COPY_QUAD_XY | vr.COPY_QUAD_XZ | vr.COPY_QUAD_YZ | vr.COPY_RUN_Z | vr.COPY_RUN_Y | vr.COPY_VOXEL | vr.COPY_COL_X | vr.COPY_COL_Y | vr.COPY_COL_Z | vr.COPY_COLV_X | vr.COPY_COLV_Y | vr.COPY_COLV_Z)
    for plane in range(3):
        assert np.array_equal(gpu.download_copy(plane), quad_copy(vox, plane)), ("quad", plane)
    assert np.array_equal(gpu.download_copy(3), run_copy(vox, False)), "run z"
    assert np.array_equal(gpu.download_copy(4), run_copy(vox, True)), "run y"
    assert np.array_equal(gpu.download_copy(5), voxel_copy(vox)), "voxel"
    for m in range(3):
        assert np.array_equal(gpu.download_copy(7 + m), column_copy(vox, m)), ("column windows along", "xyz"[m])
        assert np.array_equal(gpu.download_copy(10 + m), column_voxel_copy(vox, m)), ("voxel column windows along", "xyz"[m])


@pytest.mark.parametrize("shape", [(32, 32, 32), (17, 40, 137)])
def test_u16_copies_equal_the_host_construction(vr, gpu, shape):
    vox = _volume(shape, np.uint16, 11)
    gpu.set_layout(vr.LAYOUT_BRICKED)
    gpu.set_volume(vox)
    gpu.prepare(vr.COPY_QUAD_XY | vr.COPY_VOXEL | vr.COPY_OCT)
    assert np.array_equal(gpu.download_copy(0), quad_copy(vox, 0)), "quad u16"
    assert np.array_equal(gpu.download_copy(5), voxel_copy(vox)), "voxel u16"
    assert np.array_equal(gpu.download_copy(6), oct_copy(vox)), "oct"


def test_generated_volumes_equal_the_oracle_generator(vr, gpu, oracle):
    """generate_kernel (16-byte stores, quotient by reciprocal + correction) == the integer-only definition of SURVEY §8d, incl. edges
    that are not a multiple of the 16-byte chunk and the 2-byte variant."""
    for n, bpv in ((64, 1), (37, 1), (129, 1), (50, 2), (33, 2)):
        for kind in ("shell", "noise"):
            gpu.generate_volume(kind, n, seed=3, bytes_per_voxel=bpv)
            assert np.array_equal(gpu.download_volume(), oracle.generate_volume(kind, n, 3, bytes_per_voxel=bpv)), (kind, n, bpv)
