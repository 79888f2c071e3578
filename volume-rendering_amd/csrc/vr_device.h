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

// Software pipeline of the march: the gathers of sample i + kDepth are issued before sample i is consumed; the loop body is
// instantiated once per fetch slot (kDepth + 1 of them, rotating roles — no register copies) and iteration.  A finished lane has
// up to kDepth speculative fetches beyond its exit point, which the kLutPad repeated edge entries of the address tables absorb:
// the host checks kOverrunSteps * (cells per step) + 1.5 <= kLutPad per frame and otherwise runs the coordinate-clamping variant.
#ifndef VR_DEPTH
#define VR_DEPTH 5
#endif
constexpr int kDepth = VR_DEPTH, kSlots = kDepth + 1;
// Lazy exit test (unclamped march): whether a lane has left its ray segment is tested once per rotation of the fetch slots (and by
// every sample that composites), not per sample — a finished lane then marches up to kSlots further steps before its step becomes 0.
#ifndef VR_LAZY_EXIT
#define VR_LAZY_EXIT 1
#endif
// prefetch depth of the run-brick variants (one 8-byte gather = two registers per slot)
#ifndef VR_RUN_DEPTH
#define VR_RUN_DEPTH VR_DEPTH
#endif
constexpr int kDepthTwoByte = 3;        // TRILINEAR with 2-byte voxels: four registers per slot — three samples ahead keep 64 VGPRs
constexpr int kRunDepth = VR_RUN_DEPTH, kMaxDepth = kRunDepth > kDepth ? kRunDepth : kDepth;
constexpr int kOverrunSteps = kMaxDepth + (VR_LAZY_EXIT ? kMaxDepth + 1 : 0);     // steps a speculative fetch may lie beyond a ray's exit point
constexpr int kLutPad = (kOverrunSteps * 5 + 2) / 3 + 2;                          // >= kOverrunSteps * 1.666 (the reference's longest step, in cells) + 1.5
static_assert(kDepth >= 1 && kDepth <= 6 && kRunDepth >= 1 && kRunDepth <= 12, "prefetch depth");

// Everything the ray-march kernel reads that is not an array: passed BY VALUE as the kernel argument (the reference
// does the same with its Raycaster POD, GPURenderer1.cu:30,108) so it lands in SGPRs via s_load from the kernarg segment.
struct RayKernelArgs {
	vr_params p;
	uint32_t dim_x, dim_y, dim_z;      // Model::dims (ModelBase.h:13), widened
	uint32_t tiles_x, tiles_y;         // workgroup tiles (32x16 or 32x32 pixels) over the output; filled in at launch
	uint32_t phase_x, phase_y;         // the tile grid starts at pixel (-phase_x, -phase_y) of the output buffer (0..7)
	uint32_t lane_map;                 // order of the 16 lanes of a 4x4-pixel group: kLaneRows (4 consecutive lanes = 4 pixels along
	                                   // screen x), kLaneColumns (along screen y), kLaneBlocks (2x2-pixel blocks)
	uint64_t stride_y, stride_z;       // voxel strides (elements): dim_x, dim_x*dim_y
	float    half_x, half_y, half_z;   // 0.5f * dim  (TRILINEAR coordinate: xb = fma(pos, half, half - 0.5))
	float    off_x,  off_y,  off_z;    // 0.5f * dim - 0.5f
	float    max_x,  max_y,  max_z;    // dim - 1 as float (clamp addressing)
	float    lh_x,   lh_y,   lh_z;     // 0.01f * half: texel-space length of the shading offset (GPURenderer4.cu:43-46)
	float    tf_scale;                 // 128/255 (u8) or 128/65535 (u16): raw interpolated voxel -> TF texel coordinate + 0.5
	float    kd_scaled;                // light_kd / 255 (u8) or / 65535 (u16)
	float    tf_zero_below;            // entries 0..tf_zero_below of the premultiplied TF are exactly (0,0,0,0); -1 if entry 0 is not
	uint32_t skip_mask;                // per-voxel bit mask ~(skip_below - 1), replicated over the packed word: all 8 corners below the
	                                   // power of two skip_below => TF coordinate <= tf_zero_below
	uint32_t skip_cmp;                 // the corner test is (corners & skip_mask) != skip_cmp: 0 normally; a TF with no leading zero entries
	                                   // (no sample may take the shortcut) sets skip_mask = 0, skip_cmp = 1 — the test then always holds
	uint32_t clamp_fetch;              // 1: clamp the fetch coordinates of every sample (views whose fp32 coordinates may leave (-1, N),
	                                   // ray steps so long that two of them leave the table padding)
	uint32_t near_scaled;              // 1: every edge is a power of two — NEAREST may march in the scaled domain (sample_nearest_scaled)
	uint32_t esl_div_magic, esl_div_shift;   // n / esl_block_dims: magic != 0 ? mulhi(n, magic) : n >> shift
	uint32_t layout;                   // vr_layout in use for this launch
	uint32_t brick_plane;              // chunk plane of the brick copy handed to the kernel (kPlaneXY ...)
	uint32_t force_wide;               // testing aid: 1 = arithmetic 64-bit path, 2 = 64-bit table path, even for small volumes
	uint32_t nbx, nby, nbz;            // bricks per axis (bricked layout)
	uint64_t alt_copy;                 // kLayoutRunDual: address of the run copy along y (the kernel's `vol` argument is the copy along z)
	uint32_t dual_analytic;            // kLayoutRunDual: 1 = every tile picks its run copy from dual_bits (no launch-order entry needed)
	uint32_t dual_shift;               // ... one bit per group of (1 << dual_shift) consecutive tile NUMBERS (>= 6: 64 numbers = one 8x8-tile block of the numbering)
	uint32_t dual_bits[32];            // ... bit set = the block's tiles read the copy with runs along y (vr_hip_api.cpp dual_choice_bits)
	uint32_t col_axis;                 // kLayoutColumn: the march axis m (0 x, 1 y, 2 z) of the window copy handed to the kernel
	// kLayoutColumn: what the DENSE path of the column kernels reads back from the kernel-argument segment (it cannot keep these ~25 scalars
	// live across the march), grouped so that ONE scalar load fetches everything a step of that path needs: the path is a dependent chain
	// per wave, and every separate s_load + s_waitcnt on it is a scalar-cache round trip (seven of them per shaded sample before the grouping)
	struct ColDenseSample {            // every composited sample
		float ax, ay, az;              // direction * half (the texel-space direction A of coordinate = fma(k, A, B)), the device's own fp32 products
		float tf_scale;
		float max_x, max_y, max_z;
		float tf_zero_below;
		float light_kd, ray_threshold;
		uint32_t pad[2];
	} col_sample;
	struct ColDenseShade {             // samples that are shaded
		float dir[3];  float kd_scaled;
		float light[3]; uint32_t nbu;  // lateral blocks along u (col_blocks(dim_u))
		float lh[3];   uint32_t nw;    // windows per column (col_windows(dim_m))
		uint32_t dim[3], pad;          // Model::dims (colmarch_nearest_kernel: map_float_int of the shading position)
	} col_shade;
#ifdef VR_BOUNDS_CHECK
	// `make EXTRA=-DVR_BOUNDS_CHECK` (debug build, not the product): every gather address of the march is held against the array it must
	// lie in, every address-table index against its padded table, the tile-cost slot against its buffer; the first violation is recorded
	// in bc_fault[0..5] = { code, workgroup, thread, value lo, value hi, limit } and the access is redirected to the start of the array,
	// so the frame completes and the launch returns VR_ERR_HIP instead of faulting the GPU.
	uint64_t bc_base, bc_bytes;        // the array `vol` points at (brick copy or linear array incl. tail slack)
	uint64_t bc_alt_bytes;             // kLayoutRunDual: size of the copy at alt_copy
	uint32_t *bc_fault;
	uint32_t bc_ntiles;
#endif
};
// Tiles are numbered in VR_TILE_ORDER x VR_TILE_ORDER blocks (blocks row-major, tiles column by column inside a block — VR_XCD_MODE
// below —, the ragged right / bottom margins after them): consecutive workgroups — which the hardware spreads over the eight XCDs — are screen neighbours.
#ifndef VR_TILE_ORDER
#define VR_TILE_ORDER 8
#endif
// Which tiles of an 8x8-tile block share an XCD (vr_kernels.hip tile_to_xy, measurements there): 5 = the tiles of a block are numbered
// column by column, so that in workgroup order XCD x renders ROW x of every block.
#ifndef VR_XCD_MODE
#define VR_XCD_MODE 5
#endif
// tile number -> tile column / row; what raymarch_kernel computes (without its other experiment switches), for the host (cost map)
inline void tile_number_to_xy(uint32_t tile, uint32_t tiles_x, uint32_t tiles_y, uint32_t *x, uint32_t *y) {
	constexpr uint32_t B = VR_TILE_ORDER > 1 ? VR_TILE_ORDER : 1;
	const uint32_t full_cols = tiles_x / B, full_rows = tiles_y / B, nblocked = full_cols * full_rows * B * B;
	if (tile < nblocked) {
		const uint32_t blk = tile / (B * B), in = tile - blk * (B * B), by = blk / full_cols, bx = blk - by * full_cols;
		if (B == 8 && VR_XCD_MODE == 5) { *x = bx * B + in / B; *y = by * B + in % B; }
		else { *x = bx * B + in % B; *y = by * B + in / B; }
		return;
	}
	uint32_t rest = tile - nblocked;
	const uint32_t right_w = tiles_x - full_cols * B, right_n = right_w * full_rows * B;
	if (rest < right_n) { *y = rest / right_w; *x = full_cols * B + rest % right_w; }
	else { rest -= right_n; *y = full_rows * B + rest / tiles_x; *x = rest % tiles_x; }
}

// measured-cost launch order of a frame (vr_kernels.hip tile_order_kernel): order = workgroup id -> tile number or NULL (identity);
// cost = per tile, the longest wave of the tile in 64-cycle units, or NULL (not recorded)
struct TileSchedule { const uint32_t *order = nullptr; uint32_t *cost = nullptr; };

// TRILINEAR volume layouts.
//   kLayoutLinear : the reference's x-fastest array (+ zeroed tail slack); a sample = 4 two-voxel loads at VOXEL
//                   alignment.  Measured on MI355X (scripts/ubench/vmem_rate.hip): a 2-byte load at an odd address costs
//                   4x an aligned one, and a wave that touches n cache lines pays ~n cycles in the L1 tag pipe.
//   kLayoutBricked: "quad bricks".  Element (x,y,z) packs the 2x2 (x,y) neighbourhood {v(x,y), v(x+1,y), v(x,y+1),
//                   v(x+1,y+1)} of slice z (indices clamped at the upper faces, where the weight is exactly 0) into one
//                   naturally ALIGNED 4-byte (u8) / 8-byte (u16) word.  Elements are stored in bricks of 8x8x8 positions;
//                   the order inside a brick is kBrickBits below.  A trilinear sample is TWO aligned loads (slices z and
//                   z+1); the vector memory pipeline serves a gather one lane quad at a time and is fastest when the
//                   quad's four addresses share an aligned 16-byte chunk (scripts/ubench/tcp_coalesce.hip), which is
//                   what the brick order and the lane order of the ray-march kernel are chosen for.  Costs 4x the voxel
//                   bytes in HBM.
//   kLayoutRun    : "run bricks" — the same quad elements, but the 8 elements of a cell column (x,y) inside a brick are
//                   CONTIGUOUS along z and followed by a duplicate of the next brick's first one (9 x 4 = 36 bytes), so that the
//                   slices z and z+1 of ANY sample are 8 adjacent bytes: ONE global_load_dwordx2 (4-byte aligned) per sample.
//                   Measured (scripts/ubench/tcp_gather64.hip): for lane quads that straddle chunks — every view that is not
//                   along a volume axis — one such gather costs 8-9 ns per wave against 2 x 7-9 ns for the two 4-byte gathers;
//                   for the chunk-aligned quads of axis-aligned views the two 4-byte gathers are cheaper (2 x 1.9 ns against
//                   6.8), so this copy is read by oblique views only.  9/8 of a quad copy, 64-bit addresses (4.5 GiB at 1024^3).
//   kLayoutRunY   : run bricks whose runs lie along Y: the element is the 2x2 neighbourhood in the (x,z) plane of row y, the 9-element
//                   runs hold rows y .. y+8 of a cell column (x,z).  Same single 8-byte gather; read by the views that march
//                   mostly along z, for which the runs of kLayoutRun would lie ALONG the march (measured: perspective along z
//                   2.99 ms with runs along z against 2.47 ms with the quad copy) — runs should lie across it.
//   kLayoutVoxel  : "voxel bricks" for NEAREST sampling — ONE voxel per element in the brick order of the (x,y) quad copy.  NEAREST
//                   needs one voxel per sample; reading it out of a 4-byte quad element moves 4x the bytes and makes a 16-byte
//                   chunk cover 2x2 cells.  With 1-byte elements a chunk covers 4x4 cells of a slice and a 128-byte line 8x8x2
//                   voxels: the lane quads of views that are not aligned stay inside one chunk far more often, and the
//                   compulsory traffic of a frame is the volume itself (1 GiB at 1024^3).
//   kLayoutOct    : "oct bricks" for TRILINEAR sampling of 2-BYTE voxels — element (x,y,z) is the whole 2x2x2 neighbourhood, eight
//                   2-byte voxels = ONE aligned 16-byte word, in the brick order of the 2-byte quad copy.  Measured
//                   (scripts/ubench/tcp_gather64.hip, column "u16 oct"): an aligned 16-byte gather costs what ONE 8-byte gather costs
//                   (6.7-10 ns per wave in every lane pattern), and the quad bricks need two of those per sample.  Twice the bytes of
//                   the quad copy (16 per voxel: 128 GiB for 2048^3); for 1-byte voxels the same idea (8-byte octets) lost against
//                   the run bricks, which share their elements along z.
//   kLayoutRunDual: BOTH run copies in one launch, chosen per screen tile (bit 31 of the tile's entry in the launch order: 0 = runs
//                   along z, 1 = runs along y).  For a view that is not along an axis the better copy depends on the cube face a tile's
//                   rays enter through: all lanes of a wave start ON that face and march in lockstep, so the wave's 64 samples lie in a
//                   plane parallel to it — with runs perpendicular to that plane no two lanes share a run (measured per tile on the
//                   oblique benchmark pose: the upper half of the screen is 25 % cheaper with runs along y, the lower half with runs
//                   along z).  The host measures both copies per tile on the first two frames of a parameter set (vr_hip_api.cpp).
//   kLayoutColumn : "column windows" (round 4) for ORTHOGONAL views along a volume axis m, full march.  Every ray of such a view stays in
//                   one cell column (u,v) for its whole march (but for at most one cell flip per lateral axis when the direction carries
//                   rounding noise), and all rays share one k sequence, so the cell ALONG the march is the same for the whole wave at
//                   every sample.  The copy holds, per cell column, WINDOWS of four consecutive quad elements along m (element e =
//                   the 2x2 (u,v) neighbourhood at march index min(e, Nm-1); window w = elements 3w .. 3w+3 = the three cells
//                   3w .. 3w+2) as ONE aligned 16-byte word; the 4x4 columns of a lateral block are contiguous (256 bytes) and a
//                   block's windows follow each other along m: address = ((bv * nbu + bu) * nw + w) * 256 + (v & 3) * 64 + (u & 3) * 16.
//                   One 16-byte gather serves the ~3 samples of a window, its transparency test is done once for the window, and the
//                   address chain of a sample shrinks to the wave-uniform cell along m (colmarch_kernel).  16/3 bytes per voxel.
enum : uint32_t { kLayoutLinear = 0, kLayoutBricked = 1, kLayoutRun = 2, kLayoutRunY = 3, kLayoutVoxel = 4, kLayoutOct = 5, kLayoutRunDual = 6, kLayoutColumn = 7 };
__host__ __device__ constexpr bool is_run_layout(int layout) { return layout == (int) kLayoutRun || layout == (int) kLayoutRunY || layout == (int) kLayoutRunDual; }
constexpr uint32_t kDualWords = 32;
constexpr uint32_t kTileAltBit = 0x80000000u;       // kLayoutRunDual: set in a launch-order entry = this tile reads the copy with runs along y
__host__ __device__ constexpr bool is_brick_table_layout(int layout) { return layout == (int) kLayoutBricked || layout == (int) kLayoutVoxel; }
constexpr uint32_t kRunLen = 9, kRunBytes = kRunLen * 4, kRunBrickBytes = 64 * kRunBytes;       // 8x8 cell columns per brick
// byte offset of cell column (x & 7, y & 7) inside a run brick: 2-D Morton order, 36-byte runs
__host__ __device__ inline uint32_t run_cell_spread(uint32_t axis, uint32_t v) {
	return (((v & 1u) << axis) | (((v >> 1) & 1u) << (2 + axis)) | (((v >> 2) & 1u) << (4 + axis))) * kRunBytes;
}
inline uint64_t run_copy_bytes(uint32_t dim_x, uint32_t dim_y, uint32_t dim_z) {
	return (uint64_t) ((dim_x + 7) / 8) * ((dim_y + 7) / 8) * ((dim_z + 7) / 8) * kRunBrickBytes + 16;
}
enum : uint32_t { kLaneRows = 0, kLaneColumns = 1, kLaneBlocks = 2 };
constexpr uint32_t kBrickEdge = 8, kBrickPitch = 512;

// Where each coordinate bit lands inside the 9-bit element offset of a brick: positions of x0 x1 x2, y0 y1 y2, z0 z1 z2.
//  * 1-byte voxels (4-byte elements): a0 b0 | a1 b1 | c0 | a2 b2 | c1 c2 for a "chunk plane" (a,b) — every aligned 16-byte
//    chunk (the unit the vector memory pipeline serves a lane quad from) is a 2x2 (a,b) block of elements, every 64 bytes a
//    4x4 block, every 128-byte line a 4x4x2 block.  Up to three copies exist, one per chunk plane (x,y) / (x,z) / (y,z);
//    the host picks the plane perpendicular to the view's dominant axis (vr_hip_api.cpp), NEAREST always reads (x,y);
//  * 2-byte voxels (8-byte elements, served at one lane quad per step whatever the order): plain Z-order with z in the
//    lowest, y in the middle and x in the top slot — measured 10 % faster than the order above on 1024^3 u16.
// All measured against the alternatives with scripts/gpu_variants.sh / gpu_planes.sh.
enum : uint32_t { kPlaneXY = 0, kPlaneXZ = 1, kPlaneYZ = 2, kPlanes = 3 };
// the brick copies a context may hold (bit i of vr_hip_prepare's mask / vr_volume_info::copies = copy i): quad bricks per chunk
// plane, run bricks along z / y, voxel bricks
enum : uint32_t { kCopyQuadXY = 0, kCopyQuadXZ = 1, kCopyQuadYZ = 2, kCopyRunZ = 3, kCopyRunY = 4, kCopyVoxel = 5, kCopyOct = 6,
                  kCopyColX = 7, kCopyColY = 8, kCopyColZ = 9,        // column windows along x / y / z (kLayoutColumn), quad elements: TRILINEAR
                  kCopyColVoxX = 10, kCopyColVoxY = 11, kCopyColVoxZ = 12,      // ... of plain voxels: NEAREST
                  kCopyKinds = 13 };
static_assert(kCopyKinds == VR_COPY_KINDS, "include/vr_hip.h VR_COPY_KINDS");
// column windows (kLayoutColumn): lateral axes (u, v) of march axis m are the two other axes in increasing order
// A lateral block is kColEdge x kColEdge cell columns: its windows are kColBlockBytes each and follow each other along m.  4 x 4 columns
// = what one wave's 8 x 8 pixels cover at the reference's zoom, 256 bytes per window.  Measured against 8 x 8 columns (1 KiB per window
// of a block, -DVR_COL_EDGE_LOG2=3; full march, the three axis-aligned poses): 1.79 / 1.84 / 2.30 ms against 1.82 / 1.88 / 2.59 ms.
#ifndef VR_COL_EDGE_LOG2
#define VR_COL_EDGE_LOG2 2
#endif
constexpr uint32_t kColEdgeLog2 = VR_COL_EDGE_LOG2, kColEdge = 1u << kColEdgeLog2, kColEdgeMask = kColEdge - 1u;
constexpr uint32_t kColCells = 3, kColWindowBytes = 16, kColRowBytes = kColEdge * kColWindowBytes, kColBlockBytes = kColEdge * kColRowBytes;       // cells per window; edge x edge columns x 16 bytes
// NEAREST reads ONE voxel per sample: its column windows hold 16 consecutive voxels of the column (cells 16w .. 16w+15, the index
// clamped at Nm - 1) in the same block / window geometry — one 16-byte gather and one transparency test per SIXTEEN samples, 1 byte per voxel.
constexpr uint32_t kColVoxCells = 16;
__host__ __device__ constexpr uint32_t col_blocks(uint32_t n) { return (n + kColEdgeMask) >> kColEdgeLog2; }
__host__ __device__ constexpr uint32_t col_axis_u(uint32_t m) { return m == 0u ? 1u : 0u; }
__host__ __device__ constexpr uint32_t col_axis_v(uint32_t m) { return m == 2u ? 1u : 2u; }
__host__ __device__ inline uint32_t col_windows(uint32_t nm, uint32_t cells = kColCells) { return (nm + cells - 1u) / cells; }
// The march prefetches windows past a ray's exit and addresses them by a running pointer without clamping the window index: inside
// the copy that reads a neighbouring block's windows (never used: those samples lie outside every segment), at its two ends it reads
// this much zeroed padding (64 windows).  The kernel bounds its window count by the windows left in march direction + kColSlots + 2 (colmarch_kernel).
constexpr uint32_t kColPadBytes = 64u * kColBlockBytes;
inline uint64_t col_copy_bytes(const uint32_t dim[3], uint32_t m, bool voxels = false) {      // without the padding
	return (uint64_t) col_blocks(dim[col_axis_u(m)]) * col_blocks(dim[col_axis_v(m)]) * col_windows(dim[m], voxels ? kColVoxCells : kColCells) * kColBlockBytes;
}
// bit position of coordinate bit k (0..2) of axis (0 = x, 1 = y, 2 = z)
__host__ __device__ inline uint32_t brick_bit(uint32_t bytes_per_voxel, uint32_t plane, uint32_t axis, uint32_t k) {
	constexpr uint8_t table[4][9] = {
		{ 0, 2, 5, 1, 3, 6, 4, 7, 8 },      // chunk plane (x,y)
		{ 0, 2, 5, 4, 7, 8, 1, 3, 6 },      // chunk plane (x,z)
		{ 4, 7, 8, 0, 2, 5, 1, 3, 6 },      // chunk plane (y,z)
		{ 2, 5, 8, 1, 4, 7, 0, 3, 6 },      // 2-byte voxels: Z-order
	};
	return table[bytes_per_voxel == 2 ? 3u : plane][3 * axis + k];
}
// spread the three low bits of a coordinate to their positions / collect them again
__host__ __device__ inline uint32_t brick_spread(uint32_t bpv, uint32_t plane, uint32_t axis, uint32_t v) {
	return ((v & 1u) << brick_bit(bpv, plane, axis, 0)) | (((v >> 1) & 1u) << brick_bit(bpv, plane, axis, 1)) | (((v >> 2) & 1u) << brick_bit(bpv, plane, axis, 2));
}
__host__ __device__ inline uint32_t brick_collect(uint32_t bpv, uint32_t plane, uint32_t axis, uint32_t local) {
	return ((local >> brick_bit(bpv, plane, axis, 0)) & 1u) | (((local >> brick_bit(bpv, plane, axis, 1)) & 1u) << 1) | (((local >> brick_bit(bpv, plane, axis, 2)) & 1u) << 2);
}

// Volume resident in HBM: the reference's linear layout (x fastest, then y, then z — ModelBase.h:18-22) followed by
// kTailSlack zeroed elements, so the +1 neighbours of a trilinear fetch at the upper faces (weight exactly 0) stay
// inside the allocation.
inline uint64_t volume_tail_slack(uint32_t dim_x, uint32_t dim_y) { return (uint64_t) dim_x * dim_y + dim_x + 2; }

// linear -> quad-brick copy
hipError_t launch_brickify(const void *linear, void *bricked, uint32_t bytes_per_voxel, uint32_t plane, uint32_t dim_x, uint32_t dim_y,
                           uint32_t dim_z, hipStream_t stream);
// linear -> voxel bricks (NEAREST): element (x,y,z) = the voxel, brick order of the (x,y) quad copy
hipError_t launch_brickify_voxel(const void *linear, void *voxel_bricks, uint32_t bytes_per_voxel, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z,
                                 hipStream_t stream);
// linear -> oct bricks (2-byte voxels): element = the 2x2x2 neighbourhood, 16 bytes
hipError_t launch_brickify_oct(const void *linear, void *oct_bricks, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, hipStream_t stream);
// linear -> run bricks (1-byte voxels)
hipError_t launch_brickify_run(const void *linear, void *run_copy, uint32_t run_layout /* kLayoutRun | kLayoutRunY */, uint32_t dim_x, uint32_t dim_y,
                               uint32_t dim_z, hipStream_t stream);
// linear -> column windows along axis m (1-byte voxels)
hipError_t launch_build_column(const void *linear, void *col_copy, uint32_t axis, bool voxels /* NEAREST windows */, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, hipStream_t stream);
// number of quad elements (each 4 * bytes_per_voxel bytes)
inline uint64_t bricked_elems(uint32_t dim_x, uint32_t dim_y, uint32_t dim_z) {
	return (uint64_t) ((dim_x + kBrickEdge - 1) / kBrickEdge) * ((dim_y + kBrickEdge - 1) / kBrickEdge) *
	       ((dim_z + kBrickEdge - 1) / kBrickEdge) * kBrickPitch;
}

// `linear` is the reference's array; `bricked` the brick copy RayKernelArgs::layout names (quad bricks of plane
// brick_plane, or the run bricks) or NULL (VR_LAYOUT_LINEAR).
hipError_t launch_raymarch(const RayKernelArgs &a, const void *linear, const void *bricked, uint32_t bytes_per_voxel,
                           const float *tf_premult /* 128 x float4 */, const uint32_t *esl_bits /* 1024 */,
                           void *out_rgba, TileSchedule sched, hipStream_t stream);

// what launch_raymarch will do with these arguments: whether the variant reads the LINEAR array (refused once that was released),
// and its grid of workgroup tiles (32x16 pixels, 32x32 for the 64-bit address tables)
struct RaymarchPlan { bool reads_linear; uint32_t tiles_x, tiles_y; uint32_t tile_h = 16; /* rows of a workgroup tile (32 pixels wide) */ };
RaymarchPlan plan_raymarch(const RayKernelArgs &a, bool have_bricked, uint32_t bytes_per_voxel);
// cost[ntiles] (recorded by a frame) -> order[ntiles] for the next frame with the same parameters; clears cost
hipError_t launch_tile_order(uint32_t *cost, uint32_t *order, uint32_t ntiles, hipStream_t stream);
// predicted cost[ntiles] for a frame that leaps and has no recording yet (a.tiles_x / tiles_y / phase filled in; tile_h = rows of a workgroup tile)
hipError_t launch_tile_estimate(const RayKernelArgs &a, uint32_t tile_h, const uint32_t *esl, uint32_t *cost, uint32_t ntiles, hipStream_t stream);
// kLayoutRunDual: choice[t] = t | (cost_y[t] < cost_z[t] ? kTileAltBit : 0); both costs NULL = alternating tiles (testing)
hipError_t launch_tile_choice(const uint32_t *cost_z, const uint32_t *cost_y, uint32_t *choice, uint32_t ntiles, hipStream_t stream);

hipError_t launch_minmax(const void *volume, uint32_t bytes_per_voxel, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z,
                         uint32_t esl_block_dims, uint8_t *minmax_dev /* 32768 x {min,max} */, hipStream_t stream);

hipError_t launch_histogram(const void *volume, uint32_t bytes_per_voxel, uint64_t voxels,
                            unsigned long long *hist256_dev, hipStream_t stream);

hipError_t launch_generate(void *volume, uint32_t kind, uint32_t n, uint32_t seed, uint32_t bytes_per_voxel,
                           hipStream_t stream);

}  // namespace vr
