"""Scene state feeding the renderer: python face of the host mirror (ViewBase / RaycasterBase, include/vr_host.h).

Reference call order (VolR.cpp:412-417): load volume -> reset_transfer_fn -> set_volume; per frame
(VolR.cpp:98-113): set_view -> render_volume.
"""
import ctypes as C

import numpy as np

from .binding import VrParams, VrView, VrError, lib, SAMPLE_TRILINEAR

BENCH_POSES = ((0.0, 0.0, 0.0), (-45.0, -45.0, 0.0), (90.0, 0.0, 0.0), (180.0, 90.0, 0.0))   # VolR.cpp:233-246


def benchmark_view(width, height, index):
    """View `index` (0-3 orthogonal, 4-7 perspective) of the reference's benchmark loop, VolR.cpp:225-253."""
    v = VrView()
    rc = lib().vr_host_benchmark_view_index(width, height, index, C.byref(v))
    if rc:
        raise VrError(rc, "vr_host_benchmark_view_index")
    return v


def custom_view(width, height, perspective, angles_deg, distance):
    v = VrView()
    a = (C.c_float * 3)(*angles_deg)
    rc = lib().vr_host_benchmark_view(width, height, int(bool(perspective)), a, float(distance), C.byref(v))
    if rc:
        raise VrError(rc, "vr_host_benchmark_view")
    return v


def whole_frame(params):
    """Partition fields for rendering the whole frame into one buffer."""
    p = params
    p.x0 = 0
    p.out_width = p.view.width
    p.out_rows = p.view.height
    p.band_rows = max(1, p.view.height)
    p.band_stride = 1
    p.band_first = 0
    return p


def band_partition(params, rank, world, band_rows):
    """Screen-space split: rank `rank` of `world` renders the bands b with b % world == rank, each
    `band_rows` rows high, into a buffer of out_rows = bands_per_rank * band_rows rows.  Returns (params, bands_per_rank).
    band_rows = ceil(height / world) gives contiguous strips; small band_rows interleaves for load balance."""
    p = params
    h = p.view.height
    nbands = -(-h // band_rows)
    per_rank = -(-nbands // world)
    p.x0 = 0
    p.out_width = p.view.width
    p.out_rows = per_rank * band_rows
    p.band_rows = band_rows
    p.band_stride = world
    p.band_first = rank
    return p, per_rank


class Scene:
    """Everything `Renderer::set_*` and `render_volume` are fed with, produced by the host mirror of RaycasterBase."""

    def __init__(self):
        self.dims = None
        self.tf = np.zeros((128, 4), dtype=np.float32)
        self.esl = np.zeros(1024, dtype=np.uint32)
        self.minmax = np.zeros((32 * 32 * 32, 2), dtype=np.uint8)
        self.base_tf = np.zeros((128, 4), dtype=np.float32)
        self.params = VrParams()
        self.params.sampling = SAMPLE_TRILINEAR
        self._voxels = None

    def _refresh(self):
        L = lib()
        rc = L.vr_host_raycaster_get(C.byref(self.params), self.tf.ctypes.data, self.esl.ctypes.data,
                                     self.minmax.ctypes.data, self.base_tf.ctypes.data)
        if rc:
            raise VrError(rc, "vr_host_raycaster_get")

    def set_volume(self, voxels=None, dims=None, minmax=None):
        """reset_transfer_fn + RaycasterBase::set_volume.  Either host `voxels` (u8, shape z,y,x) for the serial scan,
        or `dims` + `minmax` pairs computed on the GPU (vr_hip_volume_minmax)."""
        L = lib()
        if voxels is not None:
            voxels = np.ascontiguousarray(voxels, dtype=np.uint8)
            dims = (voxels.shape[2], voxels.shape[1], voxels.shape[0])
            self._voxels = voxels
        mm = None
        if minmax is not None:
            mm = np.ascontiguousarray(minmax, dtype=np.uint8)
        rc = L.vr_host_raycaster_set_volume(None if voxels is None else voxels.ctypes.data, dims[0], dims[1], dims[2],
                                            None if mm is None else mm.ctypes.data)
        if rc:
            raise VrError(rc, "vr_host_raycaster_set_volume")
        self.dims = tuple(int(d) for d in dims)
        self._refresh()
        return self

    def set_base_transfer_fn(self, base_rgba):
        b = np.ascontiguousarray(base_rgba, dtype=np.float32).reshape(128, 4)
        rc = lib().vr_host_raycaster_set_base_transfer_fn(b.ctypes.data)
        if rc:
            raise VrError(rc, "vr_host_raycaster_set_base_transfer_fn")
        self._refresh()

    def reset_transfer_fn(self):
        lib().vr_host_raycaster_reset_transfer_fn()
        self._refresh()

    def set_modes(self, esl=None, ray_threshold=None, light_kd=None, ray_step=None):
        """Clamped setters of RaycasterBase (RaycasterBase.cpp:26-44)."""
        L = lib()
        if esl is not None:
            L.vr_host_raycaster_set_esl(int(bool(esl)))
        if ray_threshold is not None:
            L.vr_host_raycaster_change_ray_threshold(float(ray_threshold), 1)
        if light_kd is not None:
            L.vr_host_raycaster_change_light_intensity(float(light_kd), 1)
        if ray_step is not None:
            L.vr_host_raycaster_change_ray_step(float(ray_step), 1)
        self._refresh()

    def frame_params(self, view, sampling=None):
        p = self.params.copy()
        p.view = view
        if sampling is not None:
            p.sampling = sampling
        return whole_frame(p)
