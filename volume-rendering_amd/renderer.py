"""HipRenderer — python face of the C ABI context (include/vr_hip.h); method names follow the reference's Renderer
interface (VolumeRendering/Renderer.h:13-28)."""
import ctypes as C

import numpy as np

from .binding import VrError, VrLaunchInfo, VrParams, VrTiming, lib


class HipRenderer:
    def __init__(self, device=0):
        self._L = lib()
        self._ctx = C.c_void_p()
        rc = self._L.vr_hip_create(int(device), C.byref(self._ctx))
        if rc:
            msg = self._L.vr_hip_last_error(self._ctx).decode() if self._ctx else "no usable HIP device; there is no CPU fallback"
            if self._ctx:
                self._L.vr_hip_destroy(self._ctx)
                self._ctx = C.c_void_p()
            raise VrError(rc, msg)
        self.device = int(device)
        self.dims = None
        self.bytes_per_voxel = None

    # -- lifetime
    def close(self):
        if self._ctx:
            self._L.vr_hip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc:
            raise VrError(rc, f"{what}: {self._L.vr_hip_last_error(self._ctx).decode()}")

    def get_name(self):
        return "HIP MI355X"

    # -- Renderer::set_*
    def set_window_buffer(self, width, height):
        self._check(self._L.vr_hip_set_window(self._ctx, width, height), "set_window_buffer")

    def set_transfer_fn(self, tf_premult, esl_bits):
        tf = np.ascontiguousarray(tf_premult, dtype=np.float32)
        esl = np.ascontiguousarray(esl_bits, dtype=np.uint32)
        if tf.size != 512 or esl.size != 1024:
            raise VrError(1, "transfer_fn must be 128x4 floats and esl 1024 words")
        self._check(self._L.vr_hip_set_transfer_fn(self._ctx, tf.ctypes.data, esl.ctypes.data), "set_transfer_fn")

    def set_volume(self, voxels):
        """voxels: numpy array shaped (z, y, x), uint8 or uint16."""
        v = np.ascontiguousarray(voxels)
        if v.dtype not in (np.uint8, np.uint16) or v.ndim != 3:
            raise VrError(1, "volume must be a 3-D uint8 / uint16 array")
        z, y, x = v.shape
        self._check(self._L.vr_hip_set_volume(self._ctx, v.ctypes.data, x, y, z, v.dtype.itemsize), "set_volume")
        self.dims, self.bytes_per_voxel = (x, y, z), v.dtype.itemsize

    def set_volume_device(self, dev_ptr, dims, bytes_per_voxel):
        self._check(self._L.vr_hip_set_volume_device(self._ctx, C.c_void_p(dev_ptr), dims[0], dims[1], dims[2], bytes_per_voxel),
                    "set_volume_device")
        self.dims, self.bytes_per_voxel = tuple(dims), bytes_per_voxel

    def set_layout(self, layout):
        """LAYOUT_LINEAR | LAYOUT_BRICKED for the TRILINEAR copy of the volume (images are identical, speed is not)."""
        self._check(self._L.vr_hip_set_layout(self._ctx, int(layout)), "set_layout")

    def set_wide_addressing(self, force):
        """Testing aid: 1 = arithmetic 64-bit path, 2 = 64-bit table path (what volumes > 1024^3 use), 0/False = automatic."""
        self._check(self._L.vr_hip_set_wide_addressing(self._ctx, int(force)), "set_wide_addressing")

    def set_brick_plane(self, plane=-1):
        """Brick copy for the TRILINEAR fetch: -1 per view (default), 0 / 1 / 2 = chunk plane (x,y) / (x,z) / (y,z); speed only."""
        self._check(self._L.vr_hip_set_brick_plane(self._ctx, int(plane)), "set_brick_plane")

    def set_tile_mapping(self, lane_map=-1, phase_x=0, phase_y=0):
        """Lane order inside a 4x4-pixel block (-1 automatic, 0 rows, 1 columns, 2 2x2 blocks) and tile-grid phase; speed only."""
        self._check(self._L.vr_hip_set_tile_mapping(self._ctx, int(lane_map), int(phase_x), int(phase_y)), "set_tile_mapping")

    def set_tile_scheduling(self, mode=1):
        """0: tile = workgroup id; 1 (default): measured-cost order for frames with ESL / ERT on (most expensive tiles first); speed only."""
        self._check(self._L.vr_hip_set_tile_scheduling(self._ctx, int(mode)), "set_tile_scheduling")

    def last_launch(self):
        """What the last render launched: dict(layout, brick_plane, lane_map, phase_x, phase_y, clamp_fetch, tiles_x, tiles_y, ordered, straddle_permille)."""
        info = VrLaunchInfo()
        self._check(self._L.vr_hip_last_launch(self._ctx, C.byref(info)), "last_launch")
        return {n: int(getattr(info, n)) for n, _ in VrLaunchInfo._fields_}

    def tile_costs(self):
        """Cost map of the last frame rendered with set_tile_scheduling(2): [tiles_y, tiles_x] uint32, 64-cycle units per workgroup tile."""
        import numpy as np
        tx, ty = C.c_uint32(0), C.c_uint32(0)
        self._check(self._L.vr_hip_read_tile_costs(self._ctx, None, 0, C.byref(tx), C.byref(ty)), "read_tile_costs")
        out = np.zeros((ty.value, tx.value), dtype=np.uint32)
        self._check(self._L.vr_hip_read_tile_costs(self._ctx, out.ctypes.data, out.size, C.byref(tx), C.byref(ty)), "read_tile_costs")
        return out

    def generate_volume(self, kind, n, seed=1, bytes_per_voxel=1):
        """Synthetic benchmark volume ('shell' | 'noise') generated straight into HBM."""
        k = {"shell": 0, "noise": 1}[kind]
        self._check(self._L.vr_hip_generate_volume(self._ctx, k, n, seed, bytes_per_voxel), "generate_volume")
        self.dims, self.bytes_per_voxel = (n, n, n), bytes_per_voxel

    def download_volume(self):
        x, y, z = self.dims
        out = np.empty((z, y, x), dtype=np.uint8 if self.bytes_per_voxel == 1 else np.uint16)
        self._check(self._L.vr_hip_download_volume(self._ctx, out.ctypes.data, out.nbytes), "download_volume")
        return out

    # -- Renderer::render_volume
    def render_volume(self, params):
        """Host-buffer flavour (reference renderer ids 0-2): returns an (out_rows, out_width, 4) uint8 array."""
        out = np.empty((params.out_rows, params.out_width, 4), dtype=np.uint8)
        self._check(self._L.vr_hip_render(self._ctx, C.byref(params), out.ctypes.data), "render_volume")
        return out

    def render_volume_device(self, params, dev_ptr, stream=None):
        """Device-buffer flavour (ids 3-4): asynchronous launch into `dev_ptr` on `stream` (raw hipStream_t or None)."""
        self._check(self._L.vr_hip_render_device(self._ctx, C.byref(params), C.c_void_p(dev_ptr),
                                                 C.c_void_p(stream) if stream else None), "render_volume_device")

    # -- timing (Profiler.cpp:46-67)
    def timing(self):
        t = VrTiming()
        self._check(self._L.vr_hip_timing(self._ctx, C.byref(t)), "timing")
        return t

    def timing_reset(self):
        self._check(self._L.vr_hip_timing_reset(self._ctx), "timing_reset")

    # -- feeders on the GPU
    def volume_minmax(self):
        mm = np.empty((32 * 32 * 32, 2), dtype=np.uint8)
        bd = C.c_uint32()
        bs = (C.c_float * 3)()
        ms = C.c_float()
        self._check(self._L.vr_hip_volume_minmax(self._ctx, mm.ctypes.data, C.byref(bd), bs, C.byref(ms)), "volume_minmax")
        return mm, int(bd.value), tuple(float(v) for v in bs), float(ms.value)

    def volume_histogram(self):
        h = np.empty(256, dtype=np.uint64)
        ms = C.c_float()
        self._check(self._L.vr_hip_volume_histogram(self._ctx, h.ctypes.data, C.byref(ms)), "volume_histogram")
        return h, float(ms.value)

    def volume_info(self):
        """vr_volume_info: what the resident volume occupies (how many brick copies were actually built, linear array present)."""
        from .binding import VrVolumeInfo
        info = VrVolumeInfo()
        self._check(self._L.vr_hip_volume_info(self._ctx, C.byref(info)), "volume_info")
        return info

    def prepare(self, copies=8191):
        """vr_hip_prepare: build the brick copies named by the VR_COPY_* bits now instead of on first use (default: all the
        layout policy has at this size).  A refused copy (HBM guard) raises VrError(VR_ERR_ALLOC); frames then read the next best."""
        self._check(self._L.vr_hip_prepare(self._ctx, int(copies)), "prepare")

    def download_copy(self, kind):
        """vr_hip_download_copy (testing aid): the raw bytes of resident brick copy `kind` (bit index of its VR_COPY_* flag)."""
        n = C.c_uint64()
        self._check(self._L.vr_hip_download_copy(self._ctx, int(kind), None, 0, C.byref(n)), "download_copy")
        out = np.empty(n.value, dtype=np.uint8)
        self._check(self._L.vr_hip_download_copy(self._ctx, int(kind), out.ctypes.data, out.nbytes, None), "download_copy")
        return out

    def release_linear_copy(self):
        """Frees the linear array (feeders / download / layout changes then need a new set_volume); rendering is unaffected."""
        self._check(self._L.vr_hip_release_linear_copy(self._ctx), "release_linear_copy")

    def device_info(self):
        name = C.create_string_buffer(256)
        cus = C.c_uint32()
        mem = C.c_uint64()
        self._check(self._L.vr_hip_device_info(self._ctx, name, 256, C.byref(cus), C.byref(mem)), "device_info")
        return name.value.decode(), int(cus.value), int(mem.value)


class MultiRenderer:
    """Several GPUs of ONE process behind one render call (vr_hip_multi_* of include/vr_hip.h): interleaved bands per device,
    gathered on devices[0] over xGMI (RCCL send/recv, or peer copies), de-interleaved there.  `params` describe the whole frame."""

    def __init__(self, devices):
        self._L = lib()
        self._m = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        rc = self._L.vr_hip_multi_create(len(devices), devs, C.byref(self._m))
        if rc:
            msg = self._L.vr_hip_multi_last_error(self._m).decode() if self._m else "no usable HIP device; there is no CPU fallback"
            if self._m:
                self._L.vr_hip_multi_destroy(self._m)
                self._m = C.c_void_p()
            raise VrError(rc, msg)
        self.devices = list(devices)

    def close(self):
        if self._m:
            self._L.vr_hip_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc:
            raise VrError(rc, f"{what}: {self._L.vr_hip_multi_last_error(self._m).decode()}")

    @property
    def transport(self):
        return self._L.vr_hip_multi_transport(self._m).decode()

    def set_window_buffer(self, width, height):
        self._check(self._L.vr_hip_multi_set_window(self._m, width, height), "set_window_buffer")

    def set_transfer_fn(self, tf_premult, esl_bits):
        tf = np.ascontiguousarray(tf_premult, dtype=np.float32)
        esl = np.ascontiguousarray(esl_bits, dtype=np.uint32)
        self._check(self._L.vr_hip_multi_set_transfer_fn(self._m, tf.ctypes.data, esl.ctypes.data), "set_transfer_fn")

    def set_volume(self, voxels):
        v = np.ascontiguousarray(voxels)
        z, y, x = v.shape
        self._check(self._L.vr_hip_multi_set_volume(self._m, v.ctypes.data, x, y, z, v.dtype.itemsize), "set_volume")

    def generate_volume(self, kind, n, seed=1, bytes_per_voxel=1):
        self._check(self._L.vr_hip_multi_generate_volume(self._m, {"shell": 0, "noise": 1}[kind], n, seed, bytes_per_voxel), "generate_volume")

    def render_volume(self, params):
        out = np.empty((params.view.height, params.view.width, 4), dtype=np.uint8)
        self._check(self._L.vr_hip_multi_render(self._m, C.byref(params), out.ctypes.data), "render_volume")
        return out

    def render_volume_device(self, params, dev_ptr):
        self._check(self._L.vr_hip_multi_render_device(self._m, C.byref(params), C.c_void_p(dev_ptr)), "render_volume_device")

    def render_volume_device_async(self, params, dev_ptr, consumer_stream=None):
        """Queues a frame (up to THREE in flight: kFrames in vr_multi.cpp; frames in flight must not share `dev_ptr`); `consumer_stream`
        (raw hipStream_t on devices[0]) waits for the assembled frame."""
        self._check(self._L.vr_hip_multi_render_device_async(self._m, C.byref(params), C.c_void_p(dev_ptr),
                                                             C.c_void_p(consumer_stream) if consumer_stream else None), "render_volume_device_async")

    def sync(self):
        self._check(self._L.vr_hip_multi_sync(self._m), "sync")

    def prepare(self, copies=8191):
        self._check(self._L.vr_hip_multi_prepare(self._m, int(copies)), "prepare")

    def context_timing(self, rank):
        """vr_timing of one device's context (kernel_ms_sum / launches since its last reset)."""
        from .binding import VrTiming
        t = VrTiming()
        rc = self._L.vr_hip_timing(C.c_void_p(self._L.vr_hip_multi_context(self._m, rank)), C.byref(t))
        if rc:
            raise VrError(rc, "vr_hip_timing")
        return t

    def timing(self):
        per = (C.c_float * len(self.devices))()
        total = C.c_float()
        self._check(self._L.vr_hip_multi_timing(self._m, per, C.byref(total)), "timing")
        return [float(x) for x in per], float(total.value)
