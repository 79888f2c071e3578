"""ctypes declarations of the C ABI in include/vr_hip.h and include/vr_host.h (struct layouts must match)."""
import ctypes as C
import os

SAMPLE_NEAREST = 0      # vr_sampling.VR_SAMPLE_NEAREST   — CPURenderer / GPURenderer1-3 semantics
SAMPLE_TRILINEAR = 1    # vr_sampling.VR_SAMPLE_TRILINEAR — GPURenderer4 semantics, fp32 filter weights
SAMPLE_TRILINEAR_Q8 = 2  # vr_sampling.VR_SAMPLE_TRILINEAR_Q8 — the same with 8-bit filter weights (the texture unit's definition)
LAYOUT_LINEAR = 0        # vr_layout
LAYOUT_BRICKED = 1
# VR_COPY_* bits (vr_hip_prepare / vr_volume_info.copies): quad bricks per chunk plane, run bricks along z / y, voxel bricks
COPY_QUAD_XY, COPY_QUAD_XZ, COPY_QUAD_YZ, COPY_RUN_Z, COPY_RUN_Y, COPY_VOXEL, COPY_OCT, COPY_COL_X, COPY_COL_Y, COPY_COL_Z, COPY_ALL = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 8191
COPY_COLV_X, COPY_COLV_Y, COPY_COLV_Z = 1024, 2048, 4096
COPY_NAMES = ("quad_xy", "quad_xz", "quad_yz", "run_z", "run_y", "voxel", "oct", "col_x", "col_y", "col_z", "colv_x", "colv_y", "colv_z")
COPY_KINDS = 13
TF_SIZE = 128
ESL_VOLUME_SIZE = 1024

_STATUS = {0: "VR_OK", 1: "VR_ERR_INVALID", 2: "VR_ERR_NO_DEVICE", 3: "VR_ERR_ALLOC", 4: "VR_ERR_HIP", 5: "VR_ERR_NOT_READY"}


class VrError(RuntimeError):
    def __init__(self, code, message=""):
        self.code = code
        super().__init__(f"{_STATUS.get(code, code)}: {message}" if message else _STATUS.get(code, str(code)))


class VrView(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("origin", C.c_float * 3), ("direction", C.c_float * 3), ("right_plane", C.c_float * 3),
        ("up_plane", C.c_float * 3), ("light_pos", C.c_float * 3), ("perspective", C.c_uint32),
    ]


class VrParams(C.Structure):
    _fields_ = [
        ("view", VrView),
        ("ray_step", C.c_float), ("ray_threshold", C.c_float), ("esl", C.c_uint32), ("esl_block_dims", C.c_uint32),
        ("esl_block_size", C.c_float * 3), ("light_kd", C.c_float), ("sampling", C.c_uint32),
        ("x0", C.c_uint32), ("out_width", C.c_uint32), ("out_rows", C.c_uint32),
        ("band_rows", C.c_uint32), ("band_stride", C.c_uint32), ("band_first", C.c_uint32),
    ]

    def copy(self):
        other = VrParams()
        C.memmove(C.byref(other), C.byref(self), C.sizeof(VrParams))
        return other


class VrVolumeInfo(C.Structure):
    _fields_ = [("dim_x", C.c_uint32), ("dim_y", C.c_uint32), ("dim_z", C.c_uint32), ("bytes_per_voxel", C.c_uint32),
                ("layout", C.c_uint32), ("brick_copies", C.c_uint32), ("brick_copies_wanted", C.c_uint32), ("brick_planes", C.c_uint32),
                ("linear_resident", C.c_uint32), ("run_copy", C.c_uint32), ("linear_bytes", C.c_uint64), ("bricked_bytes", C.c_uint64),
                ("copies", C.c_uint32), ("copies_in_policy", C.c_uint32), ("copies_refused", C.c_uint32),
                ("build_ms", C.c_float * 13), ("upload_ms", C.c_float)]


class VrTiming(C.Structure):
    _fields_ = [("kernel_ms", C.c_float), ("total_ms", C.c_float), ("launches", C.c_uint64), ("kernel_ms_sum", C.c_double),
                ("kernel_ms_max", C.c_float), ("total_ms_max", C.c_float)]


class VrLaunchInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("layout", "brick_plane", "lane_map", "phase_x", "phase_y", "clamp_fetch", "tiles_x", "tiles_y",
                                          "ordered", "straddle_permille")]


def library_path():
    """In-tree libvr_hip.so; VR_HIP_LIB selects another build of the same library (A/B runs of kernel variants)."""
    return os.environ.get("VR_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvr_hip.so")


_lib = None


def lib():
    """Loads libvr_hip.so once.  Raises (never falls back) when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  Loaded first, the
    # dynamic linker resolves libvr_hip.so's dependency on that SONAME to the SAME object, so device pointers, streams
    # and events are shared with torch.  Loaded the other way round the process ends up with two runtimes and the
    # second one finds no device.
    import torch  # noqa: F401
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing — build it with `make -C volume-rendering_amd/csrc` "
                          f"(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(path)
    P = C.POINTER
    vp, u32, u64, f32p = C.c_void_p, C.c_uint32, C.c_uint64, P(C.c_float)
    sig = {
        "vr_hip_version": (C.c_char_p, []),
        "vr_hip_create": (C.c_int, [C.c_int, P(vp)]),
        "vr_hip_destroy": (None, [vp]),
        "vr_hip_last_error": (C.c_char_p, [vp]),
        "vr_hip_set_window": (C.c_int, [vp, u32, u32]),
        "vr_hip_set_transfer_fn": (C.c_int, [vp, vp, vp]),
        "vr_hip_set_volume": (C.c_int, [vp, vp, u32, u32, u32, u32]),
        "vr_hip_set_volume_device": (C.c_int, [vp, vp, u32, u32, u32, u32]),
        "vr_hip_set_layout": (C.c_int, [vp, u32]),
        "vr_hip_set_wide_addressing": (C.c_int, [vp, u32]),
        "vr_hip_set_tile_mapping": (C.c_int, [vp, C.c_int32, u32, u32]),
        "vr_hip_set_brick_plane": (C.c_int, [vp, C.c_int32]),
        "vr_hip_set_tile_scheduling": (C.c_int, [vp, u32]),
        "vr_hip_last_launch": (C.c_int, [vp, P(VrLaunchInfo)]),
        "vr_hip_read_tile_costs": (C.c_int, [vp, vp, u32, P(u32), P(u32)]),
        "vr_hip_render": (C.c_int, [vp, P(VrParams), vp]),
        "vr_hip_render_device": (C.c_int, [vp, P(VrParams), vp, vp]),
        "vr_hip_timing": (C.c_int, [vp, P(VrTiming)]),
        "vr_hip_timing_reset": (C.c_int, [vp]),
        "vr_hip_volume_minmax": (C.c_int, [vp, vp, P(u32), f32p, f32p]),
        "vr_hip_volume_histogram": (C.c_int, [vp, vp, f32p]),
        "vr_hip_generate_volume": (C.c_int, [vp, u32, u32, u32, u32]),
        "vr_hip_download_volume": (C.c_int, [vp, vp, u64]),
        "vr_hip_device_info": (C.c_int, [vp, C.c_char_p, C.c_size_t, P(u32), P(u64)]),
        "vr_hip_volume_info": (C.c_int, [vp, P(VrVolumeInfo)]),
        "vr_hip_release_linear_copy": (C.c_int, [vp]),
        "vr_hip_prepare": (C.c_int, [vp, u32]),
        "vr_hip_download_copy": (C.c_int, [vp, u32, vp, u64, vp]),
        "vr_hip_multi_create": (C.c_int, [C.c_int, P(C.c_int), P(vp)]),
        "vr_hip_multi_destroy": (None, [vp]),
        "vr_hip_multi_last_error": (C.c_char_p, [vp]),
        "vr_hip_multi_count": (C.c_int, [vp]),
        "vr_hip_multi_context": (vp, [vp, C.c_int]),
        "vr_hip_multi_transport": (C.c_char_p, [vp]),
        "vr_hip_multi_set_window": (C.c_int, [vp, u32, u32]),
        "vr_hip_multi_set_transfer_fn": (C.c_int, [vp, vp, vp]),
        "vr_hip_multi_set_volume": (C.c_int, [vp, vp, u32, u32, u32, u32]),
        "vr_hip_multi_generate_volume": (C.c_int, [vp, u32, u32, u32, u32]),
        "vr_hip_multi_render": (C.c_int, [vp, P(VrParams), vp]),
        "vr_hip_multi_render_device": (C.c_int, [vp, P(VrParams), vp]),
        "vr_hip_multi_timing": (C.c_int, [vp, f32p, f32p]),
        "vr_hip_multi_render_device_async": (C.c_int, [vp, P(VrParams), vp, vp]),
        "vr_hip_multi_sync": (C.c_int, [vp]),
        "vr_hip_multi_prepare": (C.c_int, [vp, u32]),
        "vr_hip_multi_band_map": (None, [u32, u32, u32, P(u32), P(u32)]),
        "vr_hip_multi_default_band_rows": (u32, [u32, u32]),
        "vr_host_benchmark_view": (C.c_int, [u32, u32, u32, f32p, C.c_float, P(VrView)]),
        "vr_host_benchmark_view_index": (C.c_int, [u32, u32, u32, P(VrView)]),
        "vr_host_raycaster_set_volume": (C.c_int, [vp, u32, u32, u32, vp]),
        "vr_host_raycaster_reset_transfer_fn": (None, []),
        "vr_host_raycaster_set_base_transfer_fn": (C.c_int, [vp]),
        "vr_host_raycaster_change_ray_step": (None, [C.c_float, C.c_int]),
        "vr_host_raycaster_change_ray_threshold": (None, [C.c_float, C.c_int]),
        "vr_host_raycaster_change_light_intensity": (None, [C.c_float, C.c_int]),
        "vr_host_raycaster_set_esl": (None, [C.c_int]),
        "vr_host_raycaster_reset_ray_step": (None, []),
        "vr_host_raycaster_get": (C.c_int, [P(VrParams), vp, vp, vp, vp]),
        "vr_host_render_frame": (C.c_int, [C.c_int, u32, P(VrView), vp]),
        "vr_host_load_model": (C.c_int, [C.c_char_p, P(u32)]),
        "vr_host_set_raw_dims": (None, [u32, u32, u32, u32]),
        "vr_host_model_voxels": (vp, []),
        "vr_host_model_histogram": (None, [f32p]),
        "vr_host_quantize": (C.c_int, [vp, u32, u32, u32, C.c_int, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)      # AttributeError here = the library does not export what the headers declare
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTED_SYMBOLS = None  # filled lazily by tests from the headers
