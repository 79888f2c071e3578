/*
 * vr_hip.h — C ABI of the MI355X-native volume raycaster (libvr_hip.so).
 *
 * This is the drop-in boundary for ONE path of MiroBeno/Volume-Rendering: what happens below
 * `Renderer::render_volume()` and the three `Renderer::set_*()` calls (reference VolumeRendering/Renderer.h:13-28),
 * i.e. the work the reference does in GPURenderer1.cu:65-112, GPURenderer23.cu:55-81 and GPURenderer4.cu:89-153.
 * The host C++ mirror of the reference interface (volume-rendering_amd/csrc/host/Renderer.h, class HipRenderer)
 * is the intended caller; INTEGRATION.md shows the few lines a maintainer of the reference adds.
 *
 * Conventions
 *   - plain C, plain pointers and sizes, no C++/torch types; every entry point returns 0 on success and a
 *     non-zero vr_status otherwise; nothing exits the process or throws (the reference's cuda_safe_call
 *     exit(EXIT_FAILURE) policy, cuda_utils.h:21-31, is deliberately NOT reproduced below the ABI);
 *   - one opaque vr_ctx per GPU, no globals (the reference keeps everything in statics, Renderer.h:39-43);
 *   - a context is not thread-safe (neither is the reference);
 *   - there is NO CPU fallback: without a usable HIP device vr_hip_create() fails with VR_ERR_NO_DEVICE.
 */
#ifndef VR_HIP_H
#define VR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Reference constants, RaycasterBase.h:12-16 */
#define VR_TF_SIZE          128   /* TF_SIZE: entries of the transfer function (float4 each, premultiplied) */
#define VR_TF_RATIO         2     /* TF_RATIO = 256 / TF_SIZE */
#define VR_ESL_VOLUME_DIMS  32    /* ESL_VOLUME_DIMS: empty-space-leaping grid is 32^3 blocks */
#define VR_ESL_VOLUME_SIZE  1024  /* ESL_VOLUME_SIZE: 32^3 bits = 1024 uint32 words */
#define VR_ESL_MIN_BLOCK    8     /* ESL_MIN_BLOCK_SIZE */

typedef enum vr_status {
	VR_OK               = 0,
	VR_ERR_INVALID      = 1,  /* NULL / out-of-range argument; same value the reference returns (CPURenderer.cpp:44-45) */
	VR_ERR_NO_DEVICE    = 2,
	VR_ERR_ALLOC        = 3,  /* device allocation failed (reference: GPURenderer1.cu:91-95 returns 1) */
	VR_ERR_HIP          = 4,  /* any other HIP runtime error; see vr_hip_last_error() */
	VR_ERR_NOT_READY    = 5   /* render before set_window / set_transfer_fn / set_volume */
} vr_status;

typedef enum vr_sampling {
	VR_SAMPLE_NEAREST   = 0,  /* Model::sample_data (ModelBase.h:17-23) + transfer_fn[sample / TF_RATIO] (CPURenderer.cpp:31):
	                             semantics of CPURenderer and GPURenderer1/2/3 */
	VR_SAMPLE_TRILINEAR = 1,  /* GPURenderer4 semantics (GPURenderer4.cu:76-77,91-99,136-141): trilinear volume fetch with
	                             normalised coordinates + clamp addressing, linearly filtered transfer function;
	                             interpolation weights in full fp32 */
	VR_SAMPLE_TRILINEAR_Q8 = 2 /* the same, with the three volume weights and the transfer-function weight rounded to 8
	                             fractional bits (rint(w * 256) / 256) before use — the published definition of the linear
	                             filtering GPURenderer4's tex3D / tex1D calls run on ("9-bit fixed point with 8 bits of
	                             fractional value"): what the texture unit of renderer 4 computes, as far as it is specified */
} vr_sampling;

/* How the render paths keep the volume in HBM.  VR_LAYOUT_LINEAR: every sampling mode reads the reference's linear array.
 * VR_LAYOUT_BRICKED (default): TRILINEAR reads quad or run bricks, NEAREST reads voxel bricks (one voxel per element in brick
 * order) — copies built from the linear array on first use (vr_hip_prepare builds them ahead of time).
 * All layouts give bit-identical images; the choice is speed only. */
typedef enum vr_layout {
	VR_LAYOUT_LINEAR  = 0,    /* x-fastest linear array exactly as Model::data (ModelBase.h:18-22) */
	VR_LAYOUT_BRICKED = 1     /* default: "quad bricks" — every element packs the 2x2 (x,y) voxel neighbourhood of a slice into
	                             one aligned word, stored in 8x8x8-element bricks: a trilinear sample is two aligned loads
	                             and the cache-line footprint no longer depends on the view direction
	                             (4x the voxel bytes in HBM per copy, up to three copies for 1-byte voxels — vr_hip_set_brick_plane;
	                             volume-rendering_amd/csrc/vr_device.h) */
} vr_layout;

/* struct View, ViewBase.h:14-21 (dims widened to 32 bit, bool -> uint32) */
typedef struct vr_view {
	uint32_t width, height;     /* View::dims */
	float origin[3];
	float direction[3];
	float right_plane[3];
	float up_plane[3];
	float light_pos[3];
	uint32_t perspective;       /* 0 = orthogonal, 1 = perspective */
} vr_view;

/* The by-value part of struct Raycaster (RaycasterBase.h:20-30) that travels with every render_volume() call,
 * plus the sampling mode and the screen-space partition used for multi-GPU rendering. */
typedef struct vr_params {
	vr_view  view;
	float    ray_step;          /* Raycaster::ray_step */
	float    ray_threshold;     /* Raycaster::ray_threshold (early ray termination) */
	uint32_t esl;               /* Raycaster::esl (empty space leaping on/off) */
	uint32_t esl_block_dims;    /* Raycaster::esl_block_dims (voxels per ESL block edge) */
	float    esl_block_size[3]; /* Raycaster::esl_block_size (block edge in model space) */
	float    light_kd;          /* Raycaster::light_kd */
	uint32_t sampling;          /* vr_sampling */
	/* Screen-space partition. The output buffer holds out_rows rows of out_width pixels (RGBA8, row-major, y-up):
	 *   out[ly * out_width + lx]  <-  frame pixel (x0 + lx, gy)
	 *   gy = ((ly / band_rows) * band_stride + band_first) * band_rows + ly % band_rows
	 * Whole frame: x0 = 0, out_width = width, out_rows = height, band_rows = height, band_stride = 1, band_first = 0.
	 * Rank r of n (interleaved bands of b rows): band_rows = b, band_stride = n, band_first = r.
	 * Pixels with gy >= height stay cleared. */
	uint32_t x0, out_width, out_rows;
	uint32_t band_rows, band_stride, band_first;
} vr_params;

typedef struct vr_ctx vr_ctx;

/* Per-frame statistics filled in by vr_hip_timing(). */
typedef struct vr_timing {
	float    kernel_ms;     /* hipEvent time of the ray-march kernel of the last render (on the render's stream) */
	float    total_ms;      /* clear + parameter upload + kernel (+ D2H copy for vr_hip_render), the reference's timed region
	                           (VolR.cpp:109-111, GPURenderer1.cu:107-110) */
	uint64_t launches;      /* ray-march launches since the last vr_hip_timing_reset() */
	double   kernel_ms_sum; /* sum of kernel_ms over those launches */
	float    kernel_ms_max; /* the longest of those launches (the reference's Profiler keeps sum and max, Profiler.cpp:69-72) */
	float    total_ms_max;  /* the longest vr_hip_render call (kernel + D2H) since the reset */
} vr_timing;

/* ---- lifetime: replaces GPURenderer1::GPURenderer1 / ~GPURenderer1 (GPURenderer1.cu:17-28) ---- */
int  vr_hip_create(int device, vr_ctx **out);
void vr_hip_destroy(vr_ctx *ctx);
const char *vr_hip_last_error(const vr_ctx *ctx);           /* never NULL */

/* ---- Renderer::set_window_buffer(View) — GPURenderer1.cu:74-84: (re)allocates the device framebuffer ---- */
int vr_hip_set_window(vr_ctx *ctx, uint32_t width, uint32_t height);

/* ---- Renderer::set_transfer_fn(Raycaster) — GPURenderer1.cu:65-72: uploads the premultiplied TF (128 x float4)
 *      AND the ESL bit-volume (1024 words; bit set = block empty) ---- */
int vr_hip_set_transfer_fn(vr_ctx *ctx, const float *tf_premult_rgba, const uint32_t *esl_bits);

/* ---- Renderer::set_volume(Model) — GPURenderer1.cu:86-97: copies the host voxels to HBM.
 *      bytes_per_voxel 1 (reference) or 2 (build-side extension, little-endian u16); sizes are 64-bit inside. ---- */
int vr_hip_set_volume(vr_ctx *ctx, const void *host_voxels, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z,
                      uint32_t bytes_per_voxel);
/* same, source already in device memory (x-fastest, unpadded) */
int vr_hip_set_volume_device(vr_ctx *ctx, const void *dev_voxels, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z,
                             uint32_t bytes_per_voxel);

/* Layout policy for the TRILINEAR copy of the volume (default VR_LAYOUT_BRICKED).  Takes effect immediately for the
 * resident volume (rebuilds or drops the bricked copy) and for later set_volume calls.  No reference counterpart. */
int vr_hip_set_layout(vr_ctx *ctx, uint32_t layout);

/* Testing aid: makes every later frame use one of the two 64-bit addressing paths that volumes beyond 1024^3 / 4 GiB need
 * (BASELINE config 5), so that they can be parity-tested on small volumes: 1 = index arithmetic without tables (dims above
 * 2048), 2 = address tables with 64-bit z offsets (dims up to 2048), 0 = automatic; + 4 = clamp the fetch coordinates of
 * every sample, which only views very far from the volume or very long ray steps need; + 8 = NEAREST does not use the
 * scaled-domain address chain that volumes with power-of-two edges get.  Images are identical either way. */
int vr_hip_set_wide_addressing(vr_ctx *ctx, uint32_t force);

/* Which of the up to three brick copies the TRILINEAR fetch reads — they differ in the plane, (x,y) / (x,z) / (y,z), that the
 * 16-byte chunks of the brick order cover (volume-rendering_amd/csrc/vr_device.h): -1 = per view, the plane perpendicular to
 * the view's dominant axis (default); 0, 1, 2 = always that plane (where the copy exists: 1-byte voxels, edges up to 1024,
 * else (x,y)); 3 / 4 = the "run bricks", copies in which the two slices of a sample along z / along y are 8 adjacent bytes (one
 * gather per sample), which per-view selection uses for every view whose lane quads cannot be chunk-aligned; 5 = 2-byte volumes read
 * their oct bricks (one 16-byte element per cell) for EVERY view — per-view selection uses them for orthogonal views with at most
 * one cell per pixel and the quad bricks otherwise; 6 = BOTH run copies in one launch, chosen per screen tile — what per-view
 * selection does for full-march frames of views that are not along a volume axis: which copy is cheaper depends on the cube face
 * a tile's rays enter through, so frames 0 - 3 of a parameter set run on one copy each (twice, the second time recording what every
 * tile cost) and from frame 4 on every tile reads the copy that was cheaper for it — here for every view; 7 = both copies on
 * alternating tiles (no measurement; parity tests); 8 = the column windows (VR_COPY_COL_*) along the view's major axis for EVERY orthogonal
 * full-march frame — per-view selection reads them only when the view runs along a volume axis; 9 = never (the round-3 choice for
 * those views).  Speed only; testing and tuning aid.  No reference counterpart. */
int vr_hip_set_brick_plane(vr_ctx *ctx, int32_t plane);

/* Which pixels of a 4x4-pixel block share a lane quad, and where the tile grid starts: speed only, images are identical.
 * lane_map -1 = chosen per frame from the view (default); else (lane order) + 4 * (wave shape): order 0 = 4 pixels along screen x,
 * 1 = along screen y, 2 = 2x2-pixel blocks; wave shape 0 = 8x8 pixels per wavefront, 1 = 16 wide x 4 high, 2 = 4 wide x 16 high;
 * phase_x / phase_y (0..7) shift the tile grid left / down (ignored when lane_map is -1).  Testing and tuning aid
 * (tests force every combination and compare the images).  No reference counterpart. */
int vr_hip_set_tile_mapping(vr_ctx *ctx, int32_t lane_map, uint32_t phase_x, uint32_t phase_y);

/* In which order the screen tiles of a frame are started: 1 (default) = measured-cost order — the first frame with a given set of
 * parameters records what every tile cost (its longest wavefront), a small kernel behind it sorts the tiles, and later frames with
 * the same parameters start their most expensive tiles first, so that a few long tiles (rays that probe along a block face, deep
 * rays next to early-terminated ones) no longer form the tail of the frame; applied only while empty-space leaping or early ray
 * termination is on (the full march has no tail).  0 = tile number = workgroup id, always.  Placement only: images are identical.
 * 2 = as 0, and every frame leaves a COST MAP behind (profiling): vr_hip_read_tile_costs copies the last frame's tiles_x * tiles_y
 * values — duration of the tile's workgroup, from its start to the end of its last wavefront, in units of 64 shader-clock cycles, row
 * major from the bottom-left tile of the kernel's tile grid (32x16-pixel tiles; 32x32 on the 1024-thread paths); `host_out` NULL
 * only reports the grid.  Waits for the context's frames.
 * No reference counterpart (its 16x16 blocks are started in grid order, GPURenderer1.cu:81-82,108). */
int vr_hip_set_tile_scheduling(vr_ctx *ctx, uint32_t mode);
int vr_hip_read_tile_costs(vr_ctx *ctx, uint32_t *host_out, uint32_t capacity, uint32_t *tiles_x, uint32_t *tiles_y);

/* ---- Renderer::render_volume(uchar4 *buffer, Raycaster r) ----
 * vr_hip_render: `host_rgba` is a HOST pointer of out_width*out_rows*4 bytes (renderer ids 0-2 in the reference,
 *   VolR.cpp:76-87; GPURenderer1.cu:107-110 = clear + kernel + D2H).  Synchronous.  A frame of unpartitioned rows (band_stride 1) of
 *   at least 1 MiB is rendered as two row slices on two streams of the context, so that the first slice's copy to the host runs while
 *   the second still renders; the bytes written are the same (VR_HOST_SLICES=1: one launch + one copy).
 * vr_hip_render_device: `dev_rgba` is a DEVICE pointer (renderer ids 3-4, GPURenderer23.cu:72-81) — clear + kernel on
 *   `stream` (a hipStream_t).  NULL means the context's OWN non-blocking stream, which is not ordered against the legacy
 *   default stream: a caller that fills or reads `dev_rgba` on another stream must pass that stream (or synchronise itself).
 *   Asynchronous with respect to the host.
 * Return 0 ok / VR_ERR_INVALID on NULL arguments like the reference (GPURenderer1.cu:101-102). */
int vr_hip_render(vr_ctx *ctx, const vr_params *params, uint8_t *host_rgba);
int vr_hip_render_device(vr_ctx *ctx, const vr_params *params, void *dev_rgba, void *stream);

/* What the last vr_hip_render* call of this context launched (tuning aid and test hook; no reference counterpart): the volume copy,
 * the lane order / wave shape / tile phase that were chosen (or forced), and the kernel's tile grid. */
typedef struct vr_launch_info {
	uint32_t layout;        /* 0 linear array, 1 quad bricks, 2 / 3 run bricks along z / y, 4 voxel bricks, 5 oct bricks, 6 both run copies (per tile),
	                           7 column windows (brick_plane then holds the march axis 0 x, 1 y, 2 z) */
	uint32_t brick_plane;   /* chunk plane of a quad copy: 0 (x,y), 1 (x,z), 2 (y,z) */
	uint32_t lane_map;      /* (lane order) + 4 * (wave shape), as in vr_hip_set_tile_mapping */
	uint32_t phase_x, phase_y;
	uint32_t clamp_fetch;   /* 1: the coordinate-clamping instantiation was needed */
	uint32_t tiles_x, tiles_y;
	uint32_t ordered;       /* 1: the frame ran in a measured-cost tile order */
	uint32_t straddle_permille;   /* orthogonal views along an axis: lane groups that still straddle cells under the chosen phase */
} vr_launch_info;
int vr_hip_last_launch(vr_ctx *ctx, vr_launch_info *out);

/* ---- timing: replaces the cudaEvent pair of Profiler.cpp:46-67 ---- */
int vr_hip_timing(vr_ctx *ctx, vr_timing *out);             /* synchronises the pending events */
int vr_hip_timing_reset(vr_ctx *ctx);

/* ---- feeders of the path on the GPU ----
 * Per-ESL-block min/max scan of RaycasterBase::set_volume (RaycasterBase.cpp:101-117) as an HBM-streaming reduction
 * over the resident volume.  minmax_out: 32*32*32 pairs {min,max} (uint8; u16 volumes use the high byte),
 * index z*1024 + y*32 + x; unused blocks keep {255, 0}.  esl_block_dims_out / esl_block_size_out follow
 * RaycasterBase.cpp:97-99,118-122.  kernel_ms_out (optional) = hipEvent time of the reduction kernel. */
int vr_hip_volume_minmax(vr_ctx *ctx, uint8_t *minmax_out, uint32_t *esl_block_dims_out, float *esl_block_size_out,
                         float *kernel_ms_out);
/* 256-bin histogram of the resident volume (ModelBase.cpp:19-33 raw counts; u16: high byte). */
int vr_hip_volume_histogram(vr_ctx *ctx, uint64_t *hist256_out, float *kernel_ms_out);

/* ---- synthetic benchmark volumes generated straight into HBM (SURVEY §8d "shell" / "noise") ----
 * kind 0 = shell, 1 = noise.  Replaces the resident volume (cube n^3). */
int vr_hip_generate_volume(vr_ctx *ctx, uint32_t kind, uint32_t n, uint32_t seed, uint32_t bytes_per_voxel);
/* copy the resident volume back (unpadded, x-fastest) — for checksums and for feeding the CPU baseline */
int vr_hip_download_volume(vr_ctx *ctx, void *host_out, uint64_t bytes);

/* ---- several MI355X behind one call (the reference drives one device, VolR.cpp:141-172) ----
 * One process, one context + stream per device, volume / TF / ESL replicated; a frame is cut into interleaved bands of rows
 * (band b -> devices[b mod n]), every device renders its bands, the RGBA8 bands travel to devices[0] over xGMI — RCCL
 * ncclSend / ncclRecv (communicators from ncclCommInitAll; librccl is loaded on demand) or peer copies when RCCL is not
 * available or a device is listed twice — and a copy kernel there de-interleaves them.  `params` describe the WHOLE frame
 * (partition fields are ignored); the image equals the single-device image byte for byte.
 * vr_hip_multi_render / _render_device are synchronous, like every renderer call of the reference.  _render_device_async queues
 * a frame and returns: up to three frames are in flight (band buffers, staging, events and per-device streams exist three times; nothing
 * is created per frame; the frames in flight must not share `dev_rgba`), frame i+1 renders while the bands of frame i travel; `consumer_stream` (hipStream_t on devices[0], may be NULL) is made to wait
 * for the assembled frame; vr_hip_multi_sync waits for everything queued.
 * A one-entry list is the single-device path.  The reference's `renderers[id]->render_volume()` reaches this through
 * volr::HipRenderer's device-list constructor (volume-rendering_amd/csrc/host/Renderer.h).
 * VERIFICATION STATE: with DISTINCT devices the RCCL / peer-copy transfer has not run on hardware yet (one-GPU build box); the
 * split, the gather by copies, the assemble kernel and the pipeline run with lists that repeat device 0, and the RCCL calls run
 * with VR_MULTI_TRANSPORT=rccl-self (one communicator, peer = self).  Therefore the first frame after every set_window on
 * distinct devices is self-checked on devices[0] (its own render of the other ranks' bands against what arrived; VR_ERR_HIP
 * "gather self-check failed" on a mismatch; VR_MULTI_SELFCHECK=0 disables, =1 forces it for any list). */
typedef struct vr_multi vr_multi;
int  vr_hip_multi_create(int n, const int *devices, vr_multi **out);
void vr_hip_multi_destroy(vr_multi *m);
const char *vr_hip_multi_last_error(const vr_multi *m);
int  vr_hip_multi_count(const vr_multi *m);
vr_ctx *vr_hip_multi_context(vr_multi *m, int rank);               /* the per-device context (feeders, layout knobs, timing) */
const char *vr_hip_multi_transport(const vr_multi *m);             /* "rccl" | "peer-copy" | "single" */
int  vr_hip_multi_set_window(vr_multi *m, uint32_t width, uint32_t height);
int  vr_hip_multi_set_transfer_fn(vr_multi *m, const float *tf_premult_rgba, const uint32_t *esl_bits);
int  vr_hip_multi_set_volume(vr_multi *m, const void *host_voxels, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, uint32_t bytes_per_voxel);
int  vr_hip_multi_generate_volume(vr_multi *m, uint32_t kind, uint32_t n, uint32_t seed, uint32_t bytes_per_voxel);
int  vr_hip_multi_render(vr_multi *m, const vr_params *params, uint8_t *host_rgba);           /* whole frame -> host buffer */
int  vr_hip_multi_render_device(vr_multi *m, const vr_params *params, void *dev_rgba);        /* whole frame -> buffer on devices[0] */
int  vr_hip_multi_render_device_async(vr_multi *m, const vr_params *params, void *dev_rgba, void *consumer_stream);
int  vr_hip_multi_sync(vr_multi *m);
int  vr_hip_multi_prepare(vr_multi *m, uint32_t copies);                                      /* vr_hip_prepare on every device */
int  vr_hip_multi_timing(vr_multi *m, float *per_device_kernel_ms, float *total_ms);
/* the partition itself: frame row y belongs to device *rank_out and is row *local_row_out of that device's band buffer */
void vr_hip_multi_band_map(uint32_t n, uint32_t band_rows, uint32_t y, uint32_t *rank_out, uint32_t *local_row_out);
uint32_t vr_hip_multi_default_band_rows(uint32_t height, uint32_t n);

/* ---- brick copies: built lazily, or ahead of time ----
 * With VR_LAYOUT_BRICKED a frame reads one of up to six copies of the volume, chosen per frame from the sampling mode and the
 * view (volume-rendering_amd/csrc/vr_device.h).  set_volume only uploads the linear array; a copy is built (one kernel, 6-8 ms at
 * 1024^3, the context's stream, synchronous) by the first frame that wants it, so a NEAREST-only session holds the linear array
 * + the voxel bricks and nothing else.  vr_hip_prepare builds the copies named by `copies` (VR_COPY_* bits) now — what a
 * benchmark, or a caller about to vr_hip_release_linear_copy, does.  Copies the layout policy does not have at this volume
 * size (the second / third quad plane and the run bricks need 1-byte voxels and edges <= 1024; voxel bricks edges <= 2048; oct
 * bricks 2-byte voxels) are
 * skipped silently; every copy but the first quad copy is refused (VR_ERR_ALLOC) unless half of the HBM stays free — frames then
 * read the next best copy, see vr_volume_info::copies_refused.  No reference counterpart (GPURenderer4 builds its one cudaArray
 * in set_volume, GPURenderer4.cu:123-141). */
#define VR_COPY_QUAD_XY  (1u << 0)   /* quad bricks, 16-byte chunks in the (x,y) plane: TRILINEAR views along z (and every fallback) */
#define VR_COPY_QUAD_XZ  (1u << 1)   /* ... (x,z) plane: orthogonal views along y */
#define VR_COPY_QUAD_YZ  (1u << 2)   /* ... (y,z) plane: orthogonal views along x */
#define VR_COPY_RUN_Z    (1u << 3)   /* run bricks along z: TRILINEAR views that cannot be chunk-aligned, marching mostly along x or y */
#define VR_COPY_RUN_Y    (1u << 4)   /* run bricks along y: the same, marching mostly along z */
#define VR_COPY_VOXEL    (1u << 5)   /* voxel bricks: NEAREST */
#define VR_COPY_OCT      (1u << 6)   /* oct bricks (2-byte voxels only): one 16-byte element per cell = the whole 2x2x2 neighbourhood; what
                                        TRILINEAR reads for 2-byte voxels — one gather per sample instead of the quad bricks' two */
#define VR_COPY_COL_X    (1u << 7)   /* column windows along x / y / z (1-byte voxels, edges <= 2048): per cell column, four consecutive quad */
#define VR_COPY_COL_Y    (1u << 8)   /* elements along the axis in ONE aligned 16-byte word; what TRILINEAR reads for full-march frames of */
#define VR_COPY_COL_Z    (1u << 9)   /* ORTHOGONAL views along that axis: one gather and one transparency test per ~3 samples (vr_device.h) */
#define VR_COPY_COLV_X   (1u << 10)  /* the same for NEAREST: 16 consecutive VOXELS of a cell column in one aligned 16-byte word (1 byte per voxel): */
#define VR_COPY_COLV_Y   (1u << 11)  /* one gather and one transparency test per sixteen samples */
#define VR_COPY_COLV_Z   (1u << 12)
#define VR_COPY_ALL      0x1fffu
#define VR_COPY_KINDS    13
int vr_hip_prepare(vr_ctx *ctx, uint32_t copies);
/* Testing aid: the raw bytes of one resident brick copy (kind = bit index of its VR_COPY_* flag), so that a test can hold every copy
 * builder against a host-side construction of the layout, byte for byte.  *bytes_out (optional) = size of the copy; host_out may be
 * NULL for a size query.  VR_ERR_NOT_READY if the copy is not resident, VR_ERR_INVALID if capacity is too small. */
int vr_hip_download_copy(vr_ctx *ctx, uint32_t kind, void *host_out, uint64_t capacity, uint64_t *bytes_out);

/* ---- what the resident volume occupies in HBM, and giving some of it back ----
 * The bricked layout keeps the reference's linear array (feeders, download, building copies) next to the brick copies built so
 * far; every copy but the first is only built while at least half of the device memory stays free, so `brick_copies` may stay
 * smaller than `brick_copies_wanted` — a speed cliff this call makes visible (axis-aligned views along x / y then read the
 * (x,y) copy).  vr_hip_release_linear_copy frees the linear array once nothing will need it again: afterwards frames read the
 * copies that are resident (no further copy can be built: prepare first), while vr_hip_volume_minmax / _histogram /
 * vr_hip_download_volume / vr_hip_set_layout / vr_hip_prepare return VR_ERR_NOT_READY until the next set_volume, and so does a
 * frame that no resident copy can serve.  Refused (VR_ERR_INVALID) when no brick copy is resident yet, for edges above 2048,
 * and while the index-arithmetic path is forced (vr_hip_set_wide_addressing 1). */
typedef struct vr_volume_info {
	uint32_t dim_x, dim_y, dim_z, bytes_per_voxel;
	uint32_t layout;                /* vr_layout actually in use */
	uint32_t brick_copies;          /* brick copies resident (0..3) */
	uint32_t brick_copies_wanted;   /* copies the layout policy asks for at this size (3: u8 with edges <= 1024, else 1; 0: linear) */
	uint32_t brick_planes;          /* bit i set = the copy with chunk plane i (0 (x,y), 1 (x,z), 2 (y,z)) is resident */
	uint32_t linear_resident;       /* 1 while the linear array is in HBM */
	uint32_t run_copy;              /* further copies resident: bit 0 run bricks along z, bit 1 run bricks along y (one 8-byte gather per
	                                   TRILINEAR sample), bit 2 voxel bricks (one voxel per element, what NEAREST reads) */
	uint64_t linear_bytes, bricked_bytes;
	uint32_t copies;                /* VR_COPY_* bits of the copies resident now */
	uint32_t copies_in_policy;      /* VR_COPY_* bits the layout policy has at this volume size (what frames may still build) */
	uint32_t copies_refused;        /* VR_COPY_* bits whose build was refused (HBM guard / allocation); not retried until vr_hip_prepare or set_volume */
	float    build_ms[VR_COPY_KINDS]; /* hipEvent time of the kernel that built copy i (0 if never built) */
	float    upload_ms;             /* wall time of the upload / generation of the linear array in the last set_volume */
} vr_volume_info;
int vr_hip_volume_info(vr_ctx *ctx, vr_volume_info *out);
int vr_hip_release_linear_copy(vr_ctx *ctx);

/* ---- introspection ---- */
int vr_hip_device_info(vr_ctx *ctx, char *name_out, size_t name_cap, uint32_t *compute_units, uint64_t *hbm_bytes);
const char *vr_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VR_HIP_H */
