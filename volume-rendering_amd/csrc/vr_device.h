// vr_device.h — launch interface between the C ABI (vr_hip_api.cpp) and the gfx950 kernels (vr_kernels.hip).
// Internal to libvr_hip.so; the public boundary is include/vr_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vr_hip.h"

namespace vr {

// Hang guard: no ray of any reference view needs more than ~3.5 / ray_step iterations (k spans at most the cube
// diagonal, 2*sqrt(3), in units of |direction| >= 1).  A ray that would exceed this many iterations of either loop
// is cut off instead of stalling the GPU (k += ray_step stops advancing once k >= 2^24 * ray_step).
constexpr uint32_t kMaxRaySteps = 1u << 22;

// Everything the ray-march kernel reads that is not an array: passed BY VALUE as the kernel argument (the reference
// does the same with its Raycaster POD, GPURenderer1.cu:30,108) so it lands in SGPRs via s_load from the kernarg segment.
struct RayKernelArgs {
	vr_params p;
	uint32_t dim_x, dim_y, dim_z;      // Model::dims (ModelBase.h:13), widened
	uint32_t tiles_x, tiles_y;         // 16x16-pixel workgroup tiles over the out_width x out_rows output
	uint64_t stride_y, stride_z;       // voxel strides (elements): dim_x, dim_x*dim_y
	float    half_x, half_y, half_z;   // 0.5f * dim  (TRILINEAR coordinate: xb = fma(pos, half, half - 0.5))
	float    off_x,  off_y,  off_z;    // 0.5f * dim - 0.5f
	float    max_x,  max_y,  max_z;    // dim - 1 as float (clamp addressing)
};

// Volume resident in HBM: the reference's linear layout (x fastest, then y, then z — ModelBase.h:18-22) followed by
// kTailSlack zeroed elements, so the +1 neighbours of a trilinear fetch at the upper faces (weight exactly 0) stay
// inside the allocation.
inline uint64_t volume_tail_slack(uint32_t dim_x, uint32_t dim_y) { return (uint64_t) dim_x * dim_y + dim_x + 2; }

hipError_t launch_raymarch(const RayKernelArgs &a, const void *volume, uint32_t bytes_per_voxel,
                           const float *tf_premult /* 128 x float4 */, const uint32_t *esl_bits /* 1024 */,
                           void *out_rgba, hipStream_t stream);

hipError_t launch_minmax(const void *volume, uint32_t bytes_per_voxel, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z,
                         uint32_t esl_block_dims, uint8_t *minmax_dev /* 32768 x {min,max} */, hipStream_t stream);

hipError_t launch_histogram(const void *volume, uint32_t bytes_per_voxel, uint64_t voxels,
                            unsigned long long *hist256_dev, hipStream_t stream);

hipError_t launch_generate(void *volume, uint32_t kind, uint32_t n, uint32_t seed, uint32_t bytes_per_voxel,
                           hipStream_t stream);

}  // namespace vr
