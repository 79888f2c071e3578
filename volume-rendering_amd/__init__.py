"""MI355X-native volume raycaster — python plumbing over the C ABI (include/vr_hip.h, include/vr_host.h).

The product is libvr_hip.so (hand-written gfx950 HIP kernels behind a C ABI that mirrors the reference's
``Renderer`` plug-in interface, VolumeRendering/Renderer.h:13-28).  This package only loads that library with
ctypes and adds what python is here for: device buffers / streams (torch) and the multi-GPU frame split
(torch.distributed over RCCL).  It never imports anything from ``oracle/`` and has no CPU fallback: without the
built library, or without a GPU, it fails loudly.

The directory name contains a hyphen, so import it with
``importlib.import_module("volume-rendering_amd")`` (``__graft_entry__.load_package()`` does that).
"""
from .binding import (  # noqa: F401
    VrError, VrParams, VrView, VrTiming, lib, library_path,
    SAMPLE_NEAREST, SAMPLE_TRILINEAR, SAMPLE_TRILINEAR_Q8, TF_SIZE, ESL_VOLUME_SIZE, LAYOUT_LINEAR, LAYOUT_BRICKED,
    COPY_QUAD_XY, COPY_QUAD_XZ, COPY_QUAD_YZ, COPY_RUN_Z, COPY_RUN_Y, COPY_VOXEL, COPY_OCT, COPY_COL_X, COPY_COL_Y, COPY_COL_Z, COPY_COLV_X, COPY_COLV_Y, COPY_COLV_Z, COPY_ALL, COPY_NAMES, COPY_KINDS,
)
from .scene import Scene, benchmark_view, custom_view, whole_frame, band_partition  # noqa: F401
from .renderer import HipRenderer, MultiRenderer  # noqa: F401

__all__ = [
    "VrError", "VrParams", "VrView", "VrTiming", "lib", "library_path", "SAMPLE_NEAREST", "SAMPLE_TRILINEAR", "SAMPLE_TRILINEAR_Q8",
    "TF_SIZE", "ESL_VOLUME_SIZE", "LAYOUT_LINEAR", "LAYOUT_BRICKED", "COPY_QUAD_XY", "COPY_QUAD_XZ", "COPY_QUAD_YZ", "COPY_RUN_Z", "COPY_RUN_Y",
    "COPY_VOXEL", "COPY_OCT", "COPY_COL_X", "COPY_COL_Y", "COPY_COL_Z", "COPY_COLV_X", "COPY_COLV_Y", "COPY_COLV_Z", "COPY_ALL", "COPY_NAMES", "COPY_KINDS", "Scene", "benchmark_view", "custom_view", "whole_frame", "band_partition", "HipRenderer", "MultiRenderer",
]
