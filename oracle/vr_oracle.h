/*
 * vr_oracle.h — CPU restatement of the reference's ray-march path.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle; the product
 * (libvr_hip.so, the host mirror, the python plumbing) never does and has no CPU fallback.
 *
 * Parity status: PINNED.  NEAREST mode is checked bit-for-bit against the reference's own CPURenderer
 * compiled here from /root/reference (oracle/_ref, see oracle/Makefile) and against the committed
 * fixtures in tests/golden/ generated from it (oracle/gen_golden.py, oracle/gen_golden_fullsize.py).
 * The TRILINEAR modes restate GPURenderer4.cu's texture semantics, which cannot be compiled here (no nvcc):
 * against the reference's renderer 4 itself they stay "parity unpinned"; what IS pinned, with stated
 * tolerances asserted in tests/test_trilinear_pinning.py: the fp32 restatement against the
 * published linear-filtering model evaluated in double precision (VRO_SAMPLE_TRILINEAR_F64 below), the 8-bit
 * weight variant against the fp32 one, and the model difference against the reference's CPURenderer frames.
 */
#ifndef VR_ORACLE_H
#define VR_ORACLE_H

#include <stdint.h>
#include "../include/vr_hip.h"   /* POD parameter blocks only (vr_view, vr_params) */

#ifdef __cplusplus
extern "C" {
#endif

/* vr_params.sampling value understood by the ORACLE ONLY (the product rejects it): the TRILINEAR model in double precision */
#define VRO_SAMPLE_TRILINEAR_F64 100u

typedef struct vro_stats {
	uint64_t rays_hit;        /* rays that intersect the cube */
	uint64_t esl_probes;      /* iterations of the empty-space-leaping loop */
	uint64_t samples;         /* iterations of the colour-accumulation loop */
	uint64_t shade_fetches;   /* extra light-direction fetches */
	uint64_t lines_touched;   /* distinct 128-byte lines of the voxel array read (only if count_lines) */
} vro_stats;

/* CPURenderer::render_volume (CPURenderer.cpp:43-53).  Output layout as vr_params describes (partition aware).
 * threads <= 1: serial row/col loop like the reference; > 1: OpenMP over rows.
 * stats may be NULL; count_lines != 0 additionally tracks distinct 128-B lines (slow). Returns 0 / 1. */
int vro_render(const vr_params *p, const void *voxels, const uint32_t dims[3], uint32_t bytes_per_voxel,
               const float *tf_premult, const uint32_t *esl_bits, uint8_t *rgba_out, int threads,
               vro_stats *stats, int count_lines);

/* RaycasterBase::reset_transfer_fn (RaycasterBase.cpp:76-84): default base (non-premultiplied) TF, 128 x rgba */
void vro_default_base_tf(float *base_rgba);
/* RaycasterBase::update_transfer_fn (RaycasterBase.cpp:46-74): premultiply + ESL bits from block min/max */
void vro_update_transfer_fn(const float *base_rgba, const uint8_t *minmax32k, float *tf_premult_out, uint32_t *esl_bits_out);
/* RaycasterBase::set_volume (RaycasterBase.cpp:94-125) without the TF update: block dims, min/max scan, block size */
void vro_volume_minmax(const void *voxels, const uint32_t dims[3], uint32_t bytes_per_voxel,
                       uint8_t *minmax32k_out, uint32_t *esl_block_dims_out, float *esl_block_size_out);
/* RaycasterBase::reset_ray_step (RaycasterBase.cpp:86-92) */
float vro_default_ray_step(const uint32_t dims[3]);
/* ModelBase::compute_histogram raw counts (ModelBase.cpp:19-33) */
void vro_histogram(const void *voxels, uint64_t count, uint32_t bytes_per_voxel, uint64_t *hist256_out);

/* SURVEY §8(d) synthetic volumes: kind 0 "shell", 1 "noise"; n^3 voxels, bytes_per_voxel 1 or 2 */
void vro_generate_volume(uint32_t kind, uint32_t n, uint32_t seed, uint32_t bytes_per_voxel, void *out);

uint32_t vro_fnv1a32(const void *data, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif
