// vr_kernels.hip — gfx950 (CDNA4) kernels of the volume raycaster.  Written for MI355X only.
//
// The hot path: per-pixel ray generation, cube intersection, empty-space leaping, ray-march with NEAREST or manual
// TRILINEAR sampling of a volume held in linear HBM, transfer-function lookup from an LDS-staged table, optional
// diffuse shading, front-to-back compositing with a wavefront-ballot early-ray-termination test, RGBA8 store.
// What it computes is the reference's render_ray (CPURenderer.cpp:11-41 for NEAREST, GPURenderer4.cu:53-87 for
// TRILINEAR); how it is laid out is not:
//   * one 64-lane wavefront owns one 8x8-pixel screen tile; a workgroup is 8 waves = 32x16 pixels (16 waves = 32x32 for the
//     64-bit address tables) and stages one copy of the tables per workgroup;
//   * the transfer function (+ per-entry deltas for the filtered lookup), the ESL bit-volume and per-axis address tables live in LDS;
//   * liveness is one scalar 64-bit wave mask updated with v_cmp results; the loop runs while it is non-zero (the wave's vote);
//   * workgroup id = tile number inside 8x8-tile blocks: every block is spread over all eight XCDs (plain interleave — measured
//     faster than one screen region per XCD, see the tile-map comment in the kernel);
//   * the frame clear is fused: every pixel of the output is written exactly once (misses write 0), there is no
//     separate memset pass over the framebuffer (the reference clears first, CPURenderer.cpp:47).
//
// Numerics.  This file is compiled with -ffp-contract=off.  NEAREST mode keeps the reference's float operation order
// expression by expression (IEEE divide / sqrt, no fused ops), so its output is bit-identical to the reference's CPU
// renderer.  TRILINEAR mode is defined with explicit fused multiply-adds (oracle/vr_oracle.c states the same sequence).
#include "vr_device.h"
#include <initializer_list>

namespace vr {

#define VR_FMA(a, b, c) __builtin_fmaf((a), (b), (c))

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }
// common.h:88-96 flmin/flmax — written as the reference's ternaries (NaN behaviour included)
__device__ __forceinline__ float flmin(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float flmax(float a, float b) { return a > b ? a : b; }

// common.h:105-110 map_float_int
__device__ __forceinline__ uint32_t map_float_int(float f, uint32_t n) {
	int i = (int) (f * (float) n);
	if (i >= (int) n) i = (int) n - 1;
	if (i < 0) i = 0;
	return (uint32_t) i;
}

// LDS image, one per workgroup
struct __attribute__((aligned(16))) LdsTables {
	f4 tf[VR_TF_SIZE + 1];         // premultiplied TF; entry 128 duplicates 127 (clamp addressing of the filtered lookup)
	f4 dtf[VR_TF_SIZE + 1];        // dtf[i] = tf[i+1] - tf[i] (same fp32 subtraction the lerp would do per sample)
	uint32_t esl[VR_ESL_VOLUME_SIZE];
	float unit[256];               // NEAREST, 1-byte voxels: unit[s] = (float) s / 255.0f, the quotient Raycaster::shade forms twice per shaded sample
};

// How voxel addresses are formed (template parameter ADDR):
//   kAddr32   : 32-bit BYTE offsets from a scalar base (global_load ... v_off, s[base:base+1]); brick copy <= 4 GiB, dims <=
//               1024; per-axis offset tables in LDS, 512-thread workgroups (8 waves = 32x16 pixels)
//   kAddrLut64: dims <= 2048 and any size (BASELINE config 5: 2048^3 u16 = 64 GiB of bricks): the z table holds 64-bit byte
//               offsets, x and y 32-bit offsets inside one brick slab; 1024-thread workgroups (16 waves = 32x32 pixels) so
//               that two workgroups per CU still reach the 32-wave limit next to 56 KiB of tables each
//   kAddrWide : full 64-bit index arithmetic, no tables (anything larger; also the linear layout beyond 4 GiB)
enum : int { kAddr32 = 0, kAddrLut64 = 1, kAddrWide = 2 };


// Brick address tables at FIXED LDS positions, so a lookup is one shift + one ds_read with an immediate offset:
// z entries first ({offset(z), offset(min(z+1, Z-1))} pairs: one ds_read_b64 / b128 serves both slices), then x, then y.
// Every table has kLutPad (vr_device.h) extra entries on both sides that repeat the edge entry (clamp addressing): a speculative
// fetch up to kLutPad cells outside the volume still reads a valid address, so the march needs neither a coordinate clamp nor a
// min(k, ky) per sample (the host checks that kDepth ray steps plus rounding stay below kLutPad cells, else the clamping variant runs).
template <int ADDR> struct LutCfg          { static constexpr uint32_t max_dim = 0,    z_words = 0, x_at = 0,    y_at = 0,     words = 4,     threads = 512; };
template <> struct LutCfg<kAddr32>         { static constexpr uint32_t max_dim = 1024, z_words = 2, x_at = (1024 + 2 * kLutPad) * 2, y_at = x_at + 1024 + 2 * kLutPad,
                                                                       words = y_at + 1024 + 2 * kLutPad, threads = 512; };
template <> struct LutCfg<kAddrLut64>      { static constexpr uint32_t max_dim = 2048, z_words = 4, x_at = (2048 + 2 * kLutPad) * 4, y_at = x_at + 2048 + 2 * kLutPad,
                                                                       words = y_at + 2048 + 2 * kLutPad, threads = 1024; };

template <int BPV> struct VoxelT;
template <> struct VoxelT<1> { typedef uint8_t type; };
template <> struct VoxelT<2> { typedef uint16_t type; };

// ---- bounds-checked debug build (make EXTRA=-DVR_BOUNDS_CHECK; see RayKernelArgs) -----------------------------------------------
#ifdef VR_BOUNDS_CHECK
enum : uint32_t { kBcTableIndex = 1, kBcOffset = 2, kBcAddress = 3, kBcCostSlot = 4 };
__shared__ uint32_t bc_table_entries[3];       // entries of the x / y / z address tables as staged by this workgroup (dim + 2 * kLutPad)
__device__ __forceinline__ void bc_report(const RayKernelArgs &a, uint32_t code, uint64_t value, uint64_t limit) {
	if (atomicCAS(a.bc_fault, 0u, code) == 0u) {
		a.bc_fault[1] = blockIdx.x; a.bc_fault[2] = threadIdx.x; a.bc_fault[3] = (uint32_t) value; a.bc_fault[4] = (uint32_t) (value >> 32); a.bc_fault[5] = (uint32_t) limit;
	}
}
// table index i (cell coordinate, -kLutPad .. dim - 1 + kLutPad) of table `axis` (0 x, 1 y, 2 z as STAGED: the run axis is "z")
__device__ __forceinline__ int bc_index(const RayKernelArgs &a, uint32_t axis, int i) {
	const uint32_t entry = (uint32_t) (i + kLutPad);
	if (entry < bc_table_entries[axis]) return i;
	bc_report(a, kBcTableIndex + (axis << 8), (uint64_t) (int64_t) i, bc_table_entries[axis]);
	return 0;
}
__device__ __forceinline__ uint32_t bc_offset(const RayKernelArgs &a, uint32_t offset, uint32_t bytes) {
	if ((uint64_t) offset + bytes <= a.bc_bytes) return offset;
	bc_report(a, kBcOffset, offset, a.bc_bytes);
	return 0u;
}
__device__ __forceinline__ uint64_t bc_address(const RayKernelArgs &a, uint64_t address, uint32_t bytes) {
	if (address >= a.bc_base && address + bytes <= a.bc_base + a.bc_bytes) return address;
	if (a.alt_copy != 0ull && address >= a.alt_copy && address + bytes <= a.alt_copy + a.bc_alt_bytes) return address;
	bc_report(a, kBcAddress, address, a.bc_bytes);
	return a.bc_base;
}
#define VR_BC_INDEX(a, axis, i) bc_index((a), (axis), (i))
#define VR_BC_OFFSET(a, offset, bytes) bc_offset((a), (offset), (bytes))
#define VR_BC_ADDRESS(a, address, bytes) bc_address((a), (uint64_t) (address), (bytes))
#define VR_BC_POINTER(a, T, pointer, bytes) ((T) (uintptr_t) bc_address((a), (uint64_t) (uintptr_t) (pointer), (bytes)))
#else
#define VR_BC_INDEX(a, axis, i) (i)
#define VR_BC_OFFSET(a, offset, bytes) (offset)
#define VR_BC_ADDRESS(a, address, bytes) (address)
#define VR_BC_POINTER(a, T, pointer, bytes) (pointer)
#endif

// ---- volume fetch --------------------------------------------------------------------------------------------

// "Managed" gathers of the software-pipelined march: issued through inline asm, so the compiler's s_waitcnt insertion does not
// know them and the ray loop waits for exactly the loads it is about to read (s_waitcnt vmcnt(N), N = the loads issued since).
// Left to the compiler, the waits at the loop's control-flow joins are merged conservatively (vmcnt(1) / vmcnt(0) where vmcnt(4)
// would do) and the prefetch distance collapses to one sample — the march then runs at memory latency, not at issue rate.
// Only the hot instantiations use them (1-byte voxels, 32-bit table addressing, the quad or run bricks); the loop drains them
// with s_waitcnt vmcnt(0) before it lets go of the destination registers.
template <int BPV, int ADDR, int LAYOUT> struct Managed {
	static constexpr bool value = BPV == 1 && ADDR == kAddr32 && (is_brick_table_layout(LAYOUT) || is_run_layout(LAYOUT));
};
__device__ __forceinline__ void managed_load32(uint32_t &dst, uint32_t byte_offset, const void *base) {
	asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(byte_offset), "s"(base));
}
__device__ __forceinline__ void managed_load8(uint32_t &dst, uint32_t byte_offset, const void *base) {       // zero-extended byte
	asm volatile("global_load_ubyte %0, %1, %2" : "=v"(dst) : "v"(byte_offset), "s"(base));
}
// TRILINEAR with 2-byte voxels: the two 8-byte elements of a quad-brick sample, by 64-bit address (copies beyond 4 GiB included)
template <int BPV, int ADDR, int LAYOUT> struct ManagedTri {
	static constexpr bool value = Managed<BPV, ADDR, LAYOUT>::value || (BPV == 2 && (LAYOUT == kLayoutBricked || LAYOUT == kLayoutOct) && (ADDR == kAddr32 || ADDR == kAddrLut64));
};
__device__ __forceinline__ void managed_load64(uint64_t &dst, uint64_t address) {      // split into halves only AFTER the wait
	asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(dst) : "v"(address));
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void managed_load128(u32x4 &dst, uint64_t address) {        // oct bricks: the 2x2x2 neighbourhood of 2-byte voxels
	asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(address));
}

// Single voxel (NEAREST).  LINEAR: the reference's array.  BRICKED: component 0 of the quad element (x,y,z) IS v(x,y,z), so
// NEAREST reads the same Z-ordered bricks as TRILINEAR with one aligned element load and keeps their view-independent
// cache-line footprint; the value — hence the image — is identical.
template <int BPV, int ADDR, int LAYOUT, bool MANAGED = false>
__device__ __forceinline__ uint32_t fetch_voxel(const void *vol, const RayKernelArgs &a, const uint32_t *lut,
                                                int ix, int iy, int iz) {
	typedef typename VoxelT<BPV>::type V;
	if (is_brick_table_layout(LAYOUT)) {
		typedef LutCfg<ADDR> L;                      // table lookups take indices -kLutPad .. dim - 1 + kLutPad
		ix = VR_BC_INDEX(a, 0, ix); iy = VR_BC_INDEX(a, 1, iy); iz = VR_BC_INDEX(a, 2, iz);
		const uint32_t exy = lut[(int) L::x_at + kLutPad + ix] + lut[(int) L::y_at + kLutPad + iy];
		const uint8_t *q;
		constexpr uint32_t kBytes = LAYOUT == kLayoutVoxel ? BPV : 4u;       // what the load below reads
		(void) kBytes;
		if (ADDR == kAddr32) {
			if (MANAGED && Managed<BPV, ADDR, LAYOUT>::value) {
				uint32_t word;
				if (LAYOUT == kLayoutVoxel) managed_load8(word, VR_BC_OFFSET(a, exy + lut[(int) L::z_words * (iz + kLutPad)], kBytes), vol);
				else managed_load32(word, VR_BC_OFFSET(a, exy + lut[(int) L::z_words * (iz + kLutPad)], kBytes), vol);
				return word;
			}
			q = (const uint8_t *) vol + (exy + lut[(int) L::z_words * (iz + kLutPad)]);
		} else {
			const uint2 z = *(const uint2 *) (lut + (int) L::z_words * (iz + kLutPad));
			q = (const uint8_t *) vol + ((((uint64_t) z.y) << 32 | z.x) + exy);
		}
		q = VR_BC_POINTER(a, const uint8_t *, q, kBytes);
		// the RAW element word: the voxel is its low byte / half (voxel_of).  Masking here would hand the compiler an operation on the
		// loaded value that it hoists to the loop latch of the software-pipelined march — behind an s_waitcnt vmcnt(0) that drains
		// every prefetch once per iteration (measured: the NEAREST full march was latency bound because of it).
		if (LAYOUT == kLayoutVoxel) return *(const V *) q;           // voxel bricks: the element IS the voxel
		return *(const uint32_t *) q;
	}
	if (ADDR == kAddrWide) {
		uint64_t idx = ((uint64_t) iz * a.dim_y + iy) * a.dim_x + ix;
		return *VR_BC_POINTER(a, const V *, (const V *) vol + idx, (uint32_t) sizeof(V));
	} else {
		uint32_t idx = (iz * a.dim_y + iy) * a.dim_x + ix;
		return *VR_BC_POINTER(a, const V *, (const V *) vol + idx, (uint32_t) sizeof(V));
	}
}

// the voxel inside what fetch_voxel returned (bricked layouts return the whole quad element)
template <int BPV, int LAYOUT> __device__ __forceinline__ uint32_t voxel_of(uint32_t fetched) {
	return LAYOUT == kLayoutBricked ? (BPV == 1 ? fetched & 0xffu : fetched & 0xffffu) : fetched;
}

// ModelBase.h:17-23 Model::sample_data
template <int BPV, int ADDR, int LAYOUT, bool MANAGED = false>
__device__ __forceinline__ uint32_t sample_nearest(const void *vol, const RayKernelArgs &a, const uint32_t *lut, f3 pos) {
	uint32_t iz = map_float_int((pos.z + 1) * 0.5f, a.dim_z);
	uint32_t iy = map_float_int((pos.y + 1) * 0.5f, a.dim_y);
	uint32_t ix = map_float_int((pos.x + 1) * 0.5f, a.dim_x);
	return fetch_voxel<BPV, ADDR, LAYOUT, MANAGED>(vol, a, lut, (int) ix, (int) iy, (int) iz);
}

// The same voxel for a position INSIDE the cube, or at most kLutPad cells outside it when the layout has (padded) tables:
// truncation alone gives the cell — a fraction above -1 truncates to 0 like the lower clamp, the upper clamp is the repeated
// edge entry of the table — and ((pos + 1) * 0.5f) * n == (pos + 1) * (0.5f * n) bit for bit (both scalings by 0.5 are exact).
template <int BPV, int ADDR, int LAYOUT, bool MANAGED = false>
__device__ __forceinline__ uint32_t sample_nearest_incube(const void *vol, const RayKernelArgs &a, const uint32_t *lut, f3 pos) {
	int iz = (int) ((pos.z + 1) * a.half_z), iy = (int) ((pos.y + 1) * a.half_y), ix = (int) ((pos.x + 1) * a.half_x);
	if (!(is_brick_table_layout(LAYOUT) && ADDR != kAddrWide)) {        // no tables: clamp the index at the upper face
		const int mz = (int) a.dim_z - 1, my = (int) a.dim_y - 1, mx = (int) a.dim_x - 1;
		ix = ix < mx ? ix : mx; iy = iy < my ? iy : my; iz = iz < mz ? iz : mz;
	}
	return fetch_voxel<BPV, ADDR, LAYOUT, MANAGED>(vol, a, lut, ix, iy, iz);
}

// NEAREST in the SCALED domain, for volumes whose edges are powers of two: ps = origin * half + (direction * half) * k, and the
// cell is (int)(ps + half).  Scaling by a power of two commutes with every fp32 rounding of the reference's sequence
// t = dir * k; p = origin + t; q = p + 1; cell = (int)(q * half)   (q * half is exact), so the cell is the same bit for bit while
// one multiplication per axis disappears from the per-sample address chain.
template <int BPV, int ADDR, int LAYOUT, bool MANAGED = false>
__device__ __forceinline__ uint32_t sample_nearest_scaled(const void *vol, const RayKernelArgs &a, const uint32_t *lut, f3 ps) {
	return fetch_voxel<BPV, ADDR, LAYOUT, MANAGED>(vol, a, lut, (int) (ps.x + a.half_x), (int) (ps.y + a.half_y), (int) (ps.z + a.half_z));
}

__device__ __forceinline__ float lerp(float a, float b, float t) { return VR_FMA(t, b - a, a); }

// ---- manual trilinear fetch, split in two so that the ray-march loop can software-pipeline it -----------------------
//
// tri_issue():   texel-space coordinates -> clamp -> cell index + fractions -> address -> ISSUE the loads.
// tri_resolve(): unpack the returned words and do the 7 lerps.
// Semantics = tex3D with normalised coordinates, linear filter, clamp addressing (GPURenderer4.cu:76,136-141).
// Clamping the COORDINATE to [0, N-1] is equivalent to clamping the two neighbour indices: outside that range both
// neighbours clamp to the same voxel and lerp(a, a, t) == a exactly; at N-1 the weight of the upper neighbour is exactly
// 0.  Coordinates are >= 0 after the clamp, so float->int truncation is floor() and v_fract_f32 is x - floor(x).
// Because of the clamp every address is in bounds for ANY coordinate, so loads may be issued speculatively.
template <int BPV, int LAYOUT> struct TriFetch {
	// bricked: slice z quad, slice z+1 quad (u8: one dword each, u16: two dwords each);
	// linear : the four x-pairs (y,z) (y+1,z) (y,z+1) (y+1,z+1)
	uint32_t w0, w1, w2, w3;
	uint64_t q;                                      // run bricks, managed load: both slices as ONE 64-bit destination (w0 = low, w1 = high)
	uint64_t q2;                                     // 2-byte voxels, managed loads: q = the element of slice z (w0, w1), q2 = of slice z+1 (w2, w3)
	u32x4 o;                                         // oct bricks, managed load: the whole 16-byte element (w0 .. w3)
};

// `clamp` (wave-uniform) = false is allowed for positions INSIDE the volume's cube, i.e. coordinates in (-1, N): there
// truncation toward zero already yields the clamped cell (x in (-1, 0) -> 0 like clamp-to-0; x in (N-1, N) -> N-1 like
// clamp-to-N-1), so the three v_med3 are only needed for the interpolation weights, and those are computed in
// tri_resolve, which most samples of a sparse volume never reach (transparent shortcut of the ray loop).
template <int BPV, int ADDR, int LAYOUT, bool MANAGED = false>
__device__ __forceinline__ TriFetch<BPV, LAYOUT> tri_issue(const void *vol, const RayKernelArgs &a, const uint32_t *lut,
                                                           float xb, float yb, float zb, bool clamp) {
	TriFetch<BPV, LAYOUT> f;
	f.q = 0; f.q2 = 0; f.o = (u32x4) (0u);
	if (clamp) {
		xb = __builtin_amdgcn_fmed3f(xb, 0.0f, a.max_x);
		yb = __builtin_amdgcn_fmed3f(yb, 0.0f, a.max_y);
		zb = __builtin_amdgcn_fmed3f(zb, 0.0f, a.max_z);
	}
	int ix = (int) xb, iy = (int) yb, iz = (int) zb;                // table layouts: -kLutPad .. dim - 1 + kLutPad are valid
	if (LAYOUT != kLayoutLinear && ADDR != kAddrWide) {             // (debug build: each index against the table it is about to address)
		const bool run_y = LAYOUT == kLayoutRunY;
		(void) run_y;
		ix = VR_BC_INDEX(a, 0, ix); iy = VR_BC_INDEX(a, run_y ? 2 : 1, iy); iz = VR_BC_INDEX(a, run_y ? 1 : 2, iz);
	}
	f.w0 = f.w1 = f.w2 = f.w3 = 0;
	if (is_run_layout(LAYOUT)) {
		// run bricks: two tables hold the cell column's offset, the third the ABSOLUTE 64-bit address of (brick slab, run coordinate
		// & 7); the two slices along the run axis are 8 adjacent bytes (the ninth element of a run duplicates the next brick's first).
		// kLayoutRun: runs along z, columns (x,y); kLayoutRunY: runs along y, columns (x,z) — the table regions swap roles.
		typedef LutCfg<kAddr32> L;
		const int irun = LAYOUT == kLayoutRunY ? iy : iz, iother = LAYOUT == kLayoutRunY ? iz : iy;
		const uint32_t exy = lut[(int) L::x_at + kLutPad + ix] + lut[(int) L::y_at + kLutPad + iother];
		const uint2 zz = *(const uint2 *) (lut + 2 * (irun + kLutPad));
		const uint64_t address = VR_BC_ADDRESS(a, (((uint64_t) zz.y) << 32 | zz.x) + exy, 8u);
		if (MANAGED && Managed<BPV, ADDR, LAYOUT>::value) {
			managed_load64(f.q, address);
		} else {
			const uint2 both = *(const uint2 *) address;                                     // global_load_dwordx2, 4-byte aligned
			f.w0 = both.x; f.w1 = both.y;
		}
	} else if (LAYOUT == kLayoutOct) {
		// oct bricks (2-byte voxels): ONE aligned 16-byte element holds both slices; tables as for the quad bricks (the z + 1 entry is unused)
		typedef LutCfg<ADDR> L;
		const uint32_t exy = lut[(int) L::x_at + kLutPad + ix] + lut[(int) L::y_at + kLutPad + iy];
		uint64_t address;
		if (ADDR == kAddr32) address = (uint64_t) (uintptr_t) vol + (uint64_t) (exy + lut[(int) L::z_words * (iz + kLutPad)]);
		else { const uint2 zz = *(const uint2 *) (lut + (int) L::z_words * (iz + kLutPad)); address = (uint64_t) (uintptr_t) vol + ((((uint64_t) zz.y) << 32 | zz.x) + exy); }
		address = VR_BC_ADDRESS(a, address, 16u);
		if (MANAGED && ManagedTri<BPV, ADDR, LAYOUT>::value) managed_load128(f.o, address);
		else { const uint4 v = *(const uint4 *) address; f.w0 = v.x; f.w1 = v.y; f.w2 = v.z; f.w3 = v.w; }
	} else if (LAYOUT == kLayoutBricked) {
		constexpr uint32_t kElem = 4 * BPV;
		const uint8_t *q0, *q1;
		if (ADDR == kAddrWide) {
			const uint32_t iz1 = (uint32_t) iz + 1 < a.dim_z ? iz + 1 : iz;
			const uint64_t bxy = (uint64_t) (iy >> 3) * a.nbx + (ix >> 3), slab = (uint64_t) a.nbx * a.nby;
			const uint32_t lxy = brick_spread(BPV, a.brick_plane, 0, ix & 7u) | brick_spread(BPV, a.brick_plane, 1, iy & 7u);
			q0 = (const uint8_t *) vol + (((iz >> 3) * slab + bxy) * kBrickPitch + (lxy | brick_spread(BPV, a.brick_plane, 2, iz & 7u))) * kElem;
			q1 = (const uint8_t *) vol + (((iz1 >> 3) * slab + bxy) * kBrickPitch + (lxy | brick_spread(BPV, a.brick_plane, 2, iz1 & 7u))) * kElem;
		} else {
			// per-axis byte-offset tables in LDS (LutCfg): brick base + Morton-dilated in-brick offset, split by axis
			typedef LutCfg<ADDR> L;
			const uint32_t exy = lut[(int) L::x_at + kLutPad + ix] + lut[(int) L::y_at + kLutPad + iy];
			if (ADDR == kAddr32) {
				const uint2 zz = *(const uint2 *) (lut + (int) L::z_words * (iz + kLutPad));
				q0 = (const uint8_t *) vol + (exy + zz.x);
				q1 = (const uint8_t *) vol + (exy + zz.y);
			} else {
				const uint4 zz = *(const uint4 *) (lut + (int) L::z_words * (iz + kLutPad));
				q0 = (const uint8_t *) vol + ((((uint64_t) zz.y) << 32 | zz.x) + exy);
				q1 = (const uint8_t *) vol + ((((uint64_t) zz.w) << 32 | zz.z) + exy);
			}
		}
		q0 = VR_BC_POINTER(a, const uint8_t *, q0, kElem); q1 = VR_BC_POINTER(a, const uint8_t *, q1, kElem);
		if (BPV == 1 && MANAGED && Managed<BPV, ADDR, LAYOUT>::value) {
			managed_load32(f.w0, (uint32_t) (q0 - (const uint8_t *) vol), vol);
			managed_load32(f.w1, (uint32_t) (q1 - (const uint8_t *) vol), vol);
		} else if (BPV == 1) {                           // 2 x global_load_dword, 4-byte aligned
			f.w0 = *(const uint32_t *) q0;
			f.w1 = *(const uint32_t *) q1;
		} else if (MANAGED && ManagedTri<BPV, ADDR, LAYOUT>::value) {
			managed_load64(f.q, (uint64_t) (uintptr_t) q0);
			managed_load64(f.q2, (uint64_t) (uintptr_t) q1);
		} else {                                         // 2 x global_load_dwordx2, 8-byte aligned
			const uint2 lo = *(const uint2 *) q0, hi = *(const uint2 *) q1;
			f.w0 = lo.x; f.w1 = lo.y; f.w2 = hi.x; f.w3 = hi.y;
		}
	} else {
		// LINEAR layout: one load per x-pair at VOXEL alignment (slow when the address is odd, see vr_device.h)
		const uint8_t *p00, *p10, *p01, *p11;
		if (ADDR == kAddrWide) {
			const uint64_t e = (((uint64_t) iz * a.dim_y + iy) * a.dim_x + ix) * BPV;
			p00 = (const uint8_t *) vol + e;
			p10 = p00 + a.stride_y * BPV; p01 = p00 + a.stride_z * BPV; p11 = p01 + a.stride_y * BPV;
		} else {
			const uint32_t e = ((iz * a.dim_y + iy) * a.dim_x + ix) * (uint32_t) BPV;
			const uint32_t sy = (uint32_t) a.stride_y * BPV, sz = (uint32_t) a.stride_z * BPV;
			p00 = (const uint8_t *) vol + e;
			p10 = (const uint8_t *) vol + (e + sy); p01 = (const uint8_t *) vol + (e + sz); p11 = (const uint8_t *) vol + (e + sz + sy);
		}
		p00 = VR_BC_POINTER(a, const uint8_t *, p00, 2u * BPV); p10 = VR_BC_POINTER(a, const uint8_t *, p10, 2u * BPV);
		p01 = VR_BC_POINTER(a, const uint8_t *, p01, 2u * BPV); p11 = VR_BC_POINTER(a, const uint8_t *, p11, 2u * BPV);
		if (BPV == 1) {
			uint16_t h0, h1, h2, h3;
			__builtin_memcpy(&h0, p00, 2); __builtin_memcpy(&h1, p10, 2); __builtin_memcpy(&h2, p01, 2); __builtin_memcpy(&h3, p11, 2);
			f.w0 = h0; f.w1 = h1; f.w2 = h2; f.w3 = h3;
		} else {
			__builtin_memcpy(&f.w0, p00, 4); __builtin_memcpy(&f.w1, p10, 4); __builtin_memcpy(&f.w2, p01, 4); __builtin_memcpy(&f.w3, p11, 4);
		}
	}
	return f;
}

// returns the interpolated RAW voxel value
// VR_SAMPLE_TRILINEAR_Q8: an interpolation weight in 9-bit fixed point with 8 fractional bits, rint(w * 256) / 256 (v_rndne_f32)
template <bool Q8> __device__ __forceinline__ float filter_weight(float w) {
	return Q8 ? __builtin_rintf(w * 256.0f) * (1.0f / 256.0f) : w;
}

// (xb, yb, zb): the texel-space coordinates the words were fetched at; the fetch slots of the march do not carry them — the few
// samples that get this far recompute them from the sample's k (three fused multiply-adds, the same values bit for bit)
// `along_y` (wave-uniform, kLayoutRunDual only): the words came from the copy with runs along y
template <int BPV, int LAYOUT, bool Q8>
__device__ __forceinline__ float tri_resolve(const TriFetch<BPV, LAYOUT> &f, const RayKernelArgs &a, float xb, float yb, float zb, bool along_y = false) {
	const float ax = filter_weight<Q8>(__builtin_amdgcn_fractf(__builtin_amdgcn_fmed3f(xb, 0.0f, a.max_x)));
	const float ay = filter_weight<Q8>(__builtin_amdgcn_fractf(__builtin_amdgcn_fmed3f(yb, 0.0f, a.max_y)));
	const float az = filter_weight<Q8>(__builtin_amdgcn_fractf(__builtin_amdgcn_fmed3f(zb, 0.0f, a.max_z)));
	float v000, v100, v010, v110, v001, v101, v011, v111;
	if (LAYOUT == kLayoutRunY) {                         // elements are (x,z) neighbourhoods, the two words are rows y and y+1
		v000 = (float) (f.w0 & 0xffu); v100 = (float) ((f.w0 >> 8) & 0xffu); v001 = (float) ((f.w0 >> 16) & 0xffu); v101 = (float) (f.w0 >> 24);
		v010 = (float) (f.w1 & 0xffu); v110 = (float) ((f.w1 >> 8) & 0xffu); v011 = (float) ((f.w1 >> 16) & 0xffu); v111 = (float) (f.w1 >> 24);
	} else if (LAYOUT == kLayoutRunDual) {               // either of the two: bytes 2, 3 of word 0 and bytes 0, 1 of word 1 change places
		const float t2 = (float) ((f.w0 >> 16) & 0xffu), t3 = (float) (f.w0 >> 24), t4 = (float) (f.w1 & 0xffu), t5 = (float) ((f.w1 >> 8) & 0xffu);
		v000 = (float) (f.w0 & 0xffu); v100 = (float) ((f.w0 >> 8) & 0xffu); v011 = (float) ((f.w1 >> 16) & 0xffu); v111 = (float) (f.w1 >> 24);
		v010 = along_y ? t4 : t2; v110 = along_y ? t5 : t3; v001 = along_y ? t2 : t4; v101 = along_y ? t3 : t5;
	} else if (LAYOUT != kLayoutLinear) {
		if (BPV == 1) {                                  // v_cvt_f32_ubyte0..3
			v000 = (float) (f.w0 & 0xffu); v100 = (float) ((f.w0 >> 8) & 0xffu); v010 = (float) ((f.w0 >> 16) & 0xffu); v110 = (float) (f.w0 >> 24);
			v001 = (float) (f.w1 & 0xffu); v101 = (float) ((f.w1 >> 8) & 0xffu); v011 = (float) ((f.w1 >> 16) & 0xffu); v111 = (float) (f.w1 >> 24);
		} else {
			v000 = (float) (f.w0 & 0xffffu); v100 = (float) (f.w0 >> 16); v010 = (float) (f.w1 & 0xffffu); v110 = (float) (f.w1 >> 16);
			v001 = (float) (f.w2 & 0xffffu); v101 = (float) (f.w2 >> 16); v011 = (float) (f.w3 & 0xffffu); v111 = (float) (f.w3 >> 16);
		}
	} else {
		if (BPV == 1) {
			v000 = (float) (f.w0 & 0xffu); v100 = (float) (f.w0 >> 8); v010 = (float) (f.w1 & 0xffu); v110 = (float) (f.w1 >> 8);
			v001 = (float) (f.w2 & 0xffu); v101 = (float) (f.w2 >> 8); v011 = (float) (f.w3 & 0xffu); v111 = (float) (f.w3 >> 8);
		} else {
			v000 = (float) (f.w0 & 0xffffu); v100 = (float) (f.w0 >> 16); v010 = (float) (f.w1 & 0xffffu); v110 = (float) (f.w1 >> 16);
			v001 = (float) (f.w2 & 0xffffu); v101 = (float) (f.w2 >> 16); v011 = (float) (f.w3 & 0xffffu); v111 = (float) (f.w3 >> 16);
		}
	}
	const float c00 = lerp(v000, v100, ax), c10 = lerp(v010, v110, ax);
	const float c01 = lerp(v001, v101, ax), c11 = lerp(v011, v111, ax);
	const float c0 = lerp(c00, c10, ay), c1 = lerp(c01, c11, ay);
	return lerp(c0, c1, az);
}

// x where the wave mask has the lane's bit set, 0 elsewhere: one v_cndmask with the mask taken straight from SGPRs
__device__ __forceinline__ float select_lanes(uint64_t mask, float x) {
	float r;
	asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(mask));
	return r;
}
enum : int { kFcmpOGT = 2, kFcmpOGE = 3, kFcmpOLE = 5, kFcmpUNE = 14, kIcmpNE = 33, kIcmpSGT = 38 };   // LLVM fcmp / icmp predicate codes for __builtin_amdgcn_fcmpf / sicmp

// Exact saturation shortcut, decided per wave.  A sample is composited with weight t = 1 - acc.w (CPURenderer.cpp:34); once a
// ray's accumulated alpha is EXACTLY 1.0f that weight is exactly 0 and every later sample leaves all four channels bit for
// bit unchanged (acc + c * 0 == acc, fma(c, 0, acc) == acc for finite c), whatever the early-termination threshold — with the
// reference's "no optims" threshold of 1.0 its own test `acc.w > threshold` never fires.  When no live lane of the wave has
// acc.w != 1.0 the wave therefore skips interpolation, transfer function, shading and compositing of the sample; the march
// itself (k, the fetches, the exit test) goes on unchanged.  Lanes the mask calls open: live and acc.w != 1 (NaN counts as open).
#ifdef VR_NO_SAT_SHORTCUT
#define VR_OPEN_LANES(acc_w, live) (live)
#else
#define VR_OPEN_LANES(acc_w, live) (__builtin_amdgcn_fcmpf((acc_w), 1.0f, kFcmpUNE) & (live))
#endif

// 1/sqrt(x) of the light vector in TRILINEAR mode: integer seed + three Newton steps in plain IEEE fp32 operations,
// identical on CPU and GPU (oracle/vr_oracle.c rsqrt_nr); relative error < 2e-7.
__device__ __forceinline__ float rsqrt_nr(float x) {
	float y = __uint_as_float(0x5f3759dfu - (__float_as_uint(x) >> 1));
	const float h = 0.5f * x;
	y = y * VR_FMA(-(h * y), y, 1.5f);
	y = y * VR_FMA(-(h * y), y, 1.5f);
	y = y * VR_FMA(-(h * y), y, 1.5f);
	return y;
}

// ---- per-ray helpers (reference order of operations) ------------------------------------------------------------

// RaycasterBase.h:32-42 Raycaster::intersect; min_bound = (-1,-1,-1) (ModelBase.cpp:10-14)
__device__ __forceinline__ bool intersect(f3 pt, f3 dir, float &kx, float &ky) {
	if (dir.x == 0) dir.x = 0.00001f;
	if (dir.y == 0) dir.y = 0.00001f;
	if (dir.z == 0) dir.z = 0.00001f;
	const float mb = -1.0f, nb = 1.0f;
	f3 k1 = mk3((mb - pt.x) / dir.x, (mb - pt.y) / dir.y, (mb - pt.z) / dir.z);
	f3 k2 = mk3((nb - pt.x) / dir.x, (nb - pt.y) / dir.y, (nb - pt.z) / dir.z);
	kx = flmax(flmax(flmin(k1.x, k2.x), flmin(k1.y, k2.y)), flmin(k1.z, k2.z));
	ky = flmin(flmin(flmax(k1.x, k2.x), flmax(k1.y, k2.y)), flmax(k1.z, k2.z));
	kx = flmax(kx, 0);
	return (kx < ky) && (ky > 0);
}

// n / esl_block_dims for n < 65536 without an integer divide: shift when the block edge is a power of two, otherwise the
// high half of n * (floor(2^32 / d) + 1), which is exact for n * d < 2^32 (host: RayKernelArgs::esl_div_*).
__device__ __forceinline__ uint32_t block_of(const RayKernelArgs &a, uint32_t n) {
	return a.esl_div_magic ? __umulhi(n, a.esl_div_magic) : (n >> a.esl_div_shift);
}

struct BlockIdx { uint32_t x, y, z; };
// block coordinates of a position: map_float_int((p + 1) / 2, dim) / esl_block_dims per axis (RaycasterBase.h:59-63,69-73)
__device__ __forceinline__ BlockIdx block_index(const RayKernelArgs &a, f3 pos) {
	BlockIdx b;
	b.x = block_of(a, map_float_int((pos.x + 1) * 0.5f, a.dim_x));
	b.y = block_of(a, map_float_int((pos.y + 1) * 0.5f, a.dim_y));
	b.z = block_of(a, map_float_int((pos.z + 1) * 0.5f, a.dim_z));
	return b;
}

// RaycasterBase.h:52-65 Raycaster::sample_data_esl — bit set = block is empty; table read from LDS
__device__ __forceinline__ bool block_empty(const LdsTables &t, BlockIdx b) {
	const uint32_t index = (b.z * VR_ESL_VOLUME_DIMS + b.y) & 0xffffu;          // `unsigned short index` in the reference
	const uint32_t word = t.esl[index & (VR_ESL_VOLUME_SIZE - 1)];
	return (word & (1u << (b.x & 31u))) != 0;
}

// RaycasterBase.h:67-85 Raycaster::leap_empty_space
__device__ __forceinline__ float leap_empty_space(const RayKernelArgs &a, BlockIdx b, f3 pt, f3 dir) {
	uint32_t ix = b.x, iy = b.y, iz = b.z;
	if (dir.x > 0) ix++;
	if (dir.y > 0) iy++;
	if (dir.z > 0) iz++;
	const f3 num = mk3((-1.0f + a.p.esl_block_size[0] * (float) ix) - pt.x, (-1.0f + a.p.esl_block_size[1] * (float) iy) - pt.y,
	                   (-1.0f + a.p.esl_block_size[2] * (float) iz) - pt.z);
	// Exact shortcut: a quotient num / dir is <= 0 when num is 0 (and dir is not) or when the signs differ, and one
	// non-positive quotient makes dk = max(min(..), 0) = 0, i.e. a leap of floor(0 / step) * step = 0 — no division needed.
	// That is the steady state of a ray that runs exactly along a block face (axis-aligned views): it probes every step.
	// The sign test is the product num * dir < 0 (a product that underflows to 0 just takes the division path), kept in
	// VGPR arithmetic: per-axis lane masks would cost SGPRs, and above 80 of them a SIMD holds 7 waves instead of 8.
	{
		const float sx = num.x == 0 ? -__builtin_fabsf(dir.x) : num.x * dir.x;
		const float sy = num.y == 0 ? -__builtin_fabsf(dir.y) : num.y * dir.y;
		const float sz = num.z == 0 ? -__builtin_fabsf(dir.z) : num.z * dir.z;
		if (__builtin_fminf(__builtin_fminf(sx, sy), sz) < 0)
			return 0.0f;
	}
	f3 kp = mk3(num.x / dir.x, num.y / dir.y, num.z / dir.z);
	if (dir.x == 0) kp.x = 100;
	if (dir.y == 0) kp.y = 100;
	if (dir.z == 0) kp.z = 100;
	float dk = flmin(kp.x, kp.y);
	dk = flmin(dk, kp.z);
	dk = flmax(dk, 0);
	return __builtin_floorf(dk / a.p.ray_step) * a.p.ray_step;
}

template <int SAMPLING>
__device__ __forceinline__ f3 march_point(f3 origin, f3 dir, float k) {
	if (SAMPLING == VR_SAMPLE_NEAREST)       // CPURenderer.cpp:17,24,38: origin + (direction * k), two roundings
		return mk3(origin.x + dir.x * k, origin.y + dir.y * k, origin.z + dir.z * k);
	return mk3(VR_FMA(dir.x, k, origin.x), VR_FMA(dir.y, k, origin.y), VR_FMA(dir.z, k, origin.z));
}

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F &&body) {
	if constexpr (I < N) { body(std::integral_constant<int, I>()); static_for<I + 1, N>(body); }
}
// keeps a value in its register across this point (an inline-asm operand must not be a lambda capture, hence the functions)
__device__ __forceinline__ void pin(uint32_t &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(uint32_t &x, uint32_t &y) { asm volatile("" : "+v"(x), "+v"(y)); }
__device__ __forceinline__ void pin(uint32_t &x, uint32_t &y, uint32_t &z, uint32_t &w) { asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)); }
__device__ __forceinline__ void pin(uint64_t &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(uint64_t &x, uint64_t &y) { asm volatile("" : "+v"(x), "+v"(y)); }
__device__ __forceinline__ void pin(u32x4 &x) { asm volatile("" : "+v"(x)); }
template <int I> __device__ __forceinline__ void managed_wait() {       // s_waitcnt vmcnt(I): all but the I youngest gathers have landed
	static_assert(I >= 0 && I <= 63, "vmcnt is a 6-bit field on gfx9");
	asm volatile("s_waitcnt vmcnt(%0)" : : "n"(I));
}

// -- workgroup -> tile map, chosen by measurement on the 8-XCD chip (scripts/gpu_variants.sh, lit full march, 8-view
//    mean).  Tiles are numbered in BxB-tile blocks (B = 8: 256x128 pixels), so the ~1000 workgroups in flight at any time
//    cover a compact screen region and share bricks in both screen directions (row-major numbering: +3..7 %).  Workgroups
//    are dealt round-robin over the XCDs (b and b + 8 share an L2); three assignments of tiles to XCDs were measured:
//      0  tile = workgroup id — every block is spread over all eight XCDs (XCD x renders column x of each block)   4.66 ms
//      1  each XCD owns one contiguous eighth of the tile list (one screen band per L2)                            6.25 ms
//      2  each XCD owns whole blocks, interleaved over the frame                                                   6.06 ms
//    Concentrating a compact brick region on ONE L2 (1, 2) is markedly slower than letting all eight L2s serve it —
//    the reuse between neighbouring tiles is small (the quad elements already carry the +1 neighbours) and a compact
//    region exercises few L2 channels.  Round 4 (C4 full march, 8 views): WHICH tiles of a block share an XCD matters a little —
//      3  pairs of x-neighbours   2.43 ms (views 1 / 4 / 5)      4  2x2 quads   2.42      0  columns   2.43
//      5  XCD x renders ROW x of each block: 2.38 on those views, 2.069 against 2.093 over all eight (x-neighbours read
//         neighbouring bricks of the x-fastest brick order); frames that launch in a measured-cost order: 0.599 against 0.594.
//    Placement affects speed only.  5 is the numbering of the tiles inside a block (column-major), the same for every frame, and
//    the host's tile_number_to_xy (vr_device.h, where VR_XCD_MODE is defined) follows it.
#ifndef VR_COL_XCD_MODE
#define VR_COL_XCD_MODE 0              // the column kernels' own choice
#endif
constexpr uint32_t kTileBlock = VR_TILE_ORDER > 1 ? VR_TILE_ORDER : 1;
// tile number (the launch-order entry, or the workgroup id `bid`) -> workgroup tile column / row
template <int XCD_MODE = VR_XCD_MODE>
__device__ __forceinline__ void tile_to_xy(uint32_t tiles_x, uint32_t tiles_y, uint32_t tile, uint32_t bid, uint32_t &tile_x, uint32_t &tile_y) {
	constexpr uint32_t B = kTileBlock;
	const uint32_t ntiles = tiles_x * tiles_y;
	const uint32_t full_cols = tiles_x / B, full_rows = tiles_y / B;
	const uint32_t nblocked = full_cols * full_rows * B * B;          // tiles that lie in complete BxB blocks
	if (XCD_MODE == 1) {                                           // contiguous chunk of the tile list per XCD
		const uint32_t xcd = bid & 7u, slot = bid >> 3, q = ntiles >> 3, r = ntiles & 7u;
		tile = xcd * q + (xcd < r ? xcd : r) + slot;
	} else if (XCD_MODE == 2) {                                    // whole blocks per XCD, interleaved over the frame
		const uint32_t covered = (nblocked / (8u * B * B)) * (8u * B * B);
		if (bid < covered) {
			const uint32_t set = bid / (8u * B * B), within = bid - set * (8u * B * B);
			tile = (set * 8u + (within & 7u)) * (B * B) + (within >> 3);
		}
	}
	tile_y = tile / tiles_x; tile_x = tile - tile_y * tiles_x;
	if (B > 1) {
		if (tile < nblocked) {
			const uint32_t blk = tile / (B * B), in = tile - blk * (B * B);
			uint32_t by = blk / full_cols, bx = blk - by * full_cols;
#ifdef VR_CENTER_FIRST
			// blocks from the middle of the frame outwards: the long / opaque rays of a centred object start first
			by = (by & 1u) ? full_rows / 2u - 1u - (by >> 1) : full_rows / 2u + (by >> 1);
			bx = (bx & 1u) ? full_cols / 2u - 1u - (bx >> 1) : full_cols / 2u + (bx >> 1);
#endif
			uint32_t ix = in % B, iy = in / B;
			if (B == 8 && XCD_MODE >= 3) {                // which tiles of a block share an XCD (= in & 7 in workgroup order): 3 pairs along x, 4 2x2 quads, 5 rows
				const uint32_t xcd = in & 7u, slot = in >> 3;
				if (XCD_MODE == 3) { ix = ((xcd & 3u) << 1) | (slot & 1u); iy = ((slot >> 1) << 1) | (xcd >> 2); }
				else if (XCD_MODE == 4) { ix = ((xcd & 3u) << 1) | (slot & 1u); iy = ((slot >> 1) & 1u) | ((xcd >> 2) << 1) | ((slot >> 2) << 2); }
				else { ix = slot; iy = xcd; }
			}
			tile_x = bx * B + ix; tile_y = by * B + iy;
		} else {                                      // ragged right / bottom margins: leftover tiles, row-major
			uint32_t rest = tile - nblocked;
			const uint32_t right_w = tiles_x - full_cols * B, right_n = right_w * full_rows * B;
			if (rest < right_n) { tile_y = rest / right_w; tile_x = full_cols * B + rest % right_w; }
			else { rest -= right_n; tile_y = full_rows * B + rest / tiles_x; tile_x = rest % tiles_x; }
		}
	}

}

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); }
__device__ __forceinline__ float rlane(float v, int lane) { return __uint_as_float((uint32_t) __builtin_amdgcn_readlane((int) __float_as_uint(v), lane)); }
// the values must be in scalar registers HERE: scalar loads that produce them are issued together before this point and waited for once
template <typename T> __device__ __forceinline__ void hold_scalar(const T &v) { asm volatile("" :: "s"(v)); }
template <typename... T> __device__ __forceinline__ void hold_scalars(const T &...v) { (void) std::initializer_list<int>{ (hold_scalar(v), 0)... }; }

// ---- the ray-march kernel ------------------------------------------------------------------------------------------

template <int SAMPLING, int BPV, int ADDR, int LAYOUT>
__global__ __launch_bounds__(LutCfg<(LAYOUT != kLayoutLinear ? ADDR : kAddrWide)>::threads)
void raymarch_kernel(const RayKernelArgs a, const void *__restrict__ vol, const float *__restrict__ tf_g,
                     const uint32_t *__restrict__ esl_g, uint32_t *__restrict__ out,
                     const uint32_t *__restrict__ tile_order, uint32_t *__restrict__ tile_cost) {
	// tile_order: workgroup id -> tile number (measured-cost launch order), or NULL: identity.  tile_cost: per tile, the longest wave of
	// the tile in 64-cycle units (atomicMax), or NULL: not recorded.  Both are consumed FIRST, before the tables are staged: the hot
	// variants sit at the 80-SGPR limit of 8 waves per SIMD and their peak is the staging code, so nothing of the schedule may be live
	// there or during the march — the workgroup's start time and the address of its tile's cost word wait in LDS (one record per
	// workgroup, written by thread 0 before the staging barrier) for the end of every wave.  (A per-wave record indexed by the wave's
	// hardware slot, HW_ID, was tried and is WRONG: a wave that is context-switched out and back — several queues share the GPU —
	// comes back in another slot, reads a record nobody wrote, and the atomic below goes to a wild address.)
	__shared__ uint32_t group_sched[4];
	const uint32_t order_entry = tile_order ? tile_order[blockIdx.x] : blockIdx.x;
	// kLayoutRunDual: bit 31 of the entry selects the tile's copy (runs along y instead of z); it waits in LDS like the rest of the record
	const uint32_t tile_of_group = LAYOUT == kLayoutRunDual ? (order_entry & ~kTileAltBit) : order_entry;
	typedef LutCfg<(LAYOUT != kLayoutLinear ? ADDR : kAddrWide)> L;
	constexpr uint32_t kThreads = L::threads;
	// kLayoutRunDual: which of the two run copies this tile reads.  From the launch-order entry (measured choice, testing aid), or —
	// the product's rule, no history needed — from the bit the HOST set for the tile's group of 64 consecutive tile numbers, i.e. its
	// 8x8-tile block (RayKernelArgs::dual_bits: the analytic entry-face rule of vr_hip_api.cpp dual_choice_bits).  Read through a laundered pointer to the kernel-argument segment
	// (`a` is the first argument: offset 0) so that nothing of it stays in scalar registers across the staging code.
	bool alt_tile = LAYOUT == kLayoutRunDual && (order_entry & kTileAltBit) != 0u;
	if (LAYOUT == kLayoutRunDual && a.dual_analytic != 0u) {
		typedef const RayKernelArgs __attribute__((address_space(4))) *ConstArgs;      // constant address space: scalar loads
		ConstArgs q = (ConstArgs) __builtin_amdgcn_kernarg_segment_ptr();
		asm volatile("" : "+s"(q));
		const uint32_t bit = tile_of_group >> q->dual_shift;        // groups of consecutive tile NUMBERS: 64 = one 8x8-tile block of the numbering
		const uint32_t word = q->dual_bits[(bit >> 5) & (kDualWords - 1u)];
		alt_tile = __builtin_amdgcn_readfirstlane((int) ((word >> (bit & 31u)) & 1u)) != 0;       // uniform by construction (kernel arguments and the tile number only)
	}
	if (threadIdx.x == 0) {
#ifdef VR_BOUNDS_CHECK
		if (tile_of_group >= a.bc_ntiles) { bc_report(a, kBcCostSlot, tile_of_group, a.bc_ntiles); tile_cost = nullptr; }
#endif
		const uint64_t slot = tile_cost ? (uint64_t) (uintptr_t) (tile_cost + tile_of_group) : 0ull;
		group_sched[0] = (uint32_t) (__builtin_readcyclecounter() >> 6); group_sched[1] = (uint32_t) slot; group_sched[2] = (uint32_t) (slot >> 32);
		group_sched[3] = alt_tile ? 1u : 0u;
	}
	constexpr bool kQ8 = SAMPLING == VR_SAMPLE_TRILINEAR_Q8;        // 8-bit filter weights; everything else as TRILINEAR
	constexpr bool kUseLut = L::max_dim != 0;
	__shared__ LdsTables lds;
	__shared__ __attribute__((aligned(16))) uint32_t lut[L::words];
#ifdef VR_LDS_PAD          // tuning aid: occupy extra LDS to lower the number of resident workgroups per CU
	__shared__ uint32_t lds_pad[VR_LDS_PAD / 4];
	if (a.dim_x == 0xffffffffu) lds_pad[threadIdx.x] = 1;
#endif

	// -- stage TF (+ deltas), the ESL bit-volume and the brick address tables in LDS
	{
		const uint32_t t = threadIdx.x;
		// entry j of a table belongs to cell clamp(j - kLutPad, 0, dim - 1): the pad entries repeat the edge cells
		auto cell_of = [](uint32_t j, uint32_t n) { const int c = (int) j - kLutPad; return (uint32_t) (c < 0 ? 0 : (c > (int) n - 1 ? (int) n - 1 : c)); };
		if (kUseLut && is_run_layout(LAYOUT)) {
			// r = the run axis (z, or y for kLayoutRunY), o = the other column axis (y, or z); bricks: x fastest, then o, then r.
			// kLayoutRunDual: a tile that reads the copy along y builds ITS tables exactly like kLayoutRunY and marches with the
			// y and z components of its texel-space ray exchanged — the table regions then meet the coordinates they were built for.
			const bool along_y = LAYOUT == kLayoutRunY || alt_tile;
			const uint32_t nx = a.dim_x, nr = along_y ? a.dim_y : a.dim_z, no = along_y ? a.dim_z : a.dim_y;
			const uint32_t nbo = along_y ? a.nbz : a.nby;
			const uint64_t slab = (uint64_t) a.nbx * nbo * kRunBrickBytes;
			const uint64_t copy_base = alt_tile ? a.alt_copy : (uint64_t) (uintptr_t) vol;
#ifdef VR_BOUNDS_CHECK
			if (t == 0) { bc_table_entries[0] = nx + 2 * kLutPad; bc_table_entries[1] = no + 2 * kLutPad; bc_table_entries[2] = nr + 2 * kLutPad; }
#endif
			for (uint32_t j = t; j < nr + 2 * kLutPad; j += kThreads) {
				const uint32_t i = cell_of(j, nr);
				const uint64_t z0 = copy_base + (i >> 3) * slab + (i & 7u) * 4u;
				lut[2 * j] = (uint32_t) z0; lut[2 * j + 1] = (uint32_t) (z0 >> 32);
			}
			for (uint32_t j = t; j < nx + 2 * kLutPad; j += kThreads) { const uint32_t i = cell_of(j, nx); lut[L::x_at + j] = (i >> 3) * kRunBrickBytes + run_cell_spread(0, i & 7u); }
			for (uint32_t j = t; j < no + 2 * kLutPad; j += kThreads) { const uint32_t i = cell_of(j, no); lut[L::y_at + j] = (i >> 3) * a.nbx * kRunBrickBytes + run_cell_spread(1, i & 7u); }
		} else if (kUseLut) {
			const uint32_t nx = a.dim_x, ny = a.dim_y, nz = a.dim_z;
			const uint32_t elem = LAYOUT == kLayoutVoxel ? BPV : (LAYOUT == kLayoutOct ? 8u * BPV : 4u * BPV);   // bytes per element: one voxel, a quad, or the 2x2x2 neighbourhood
#ifdef VR_BOUNDS_CHECK
			if (t == 0) { bc_table_entries[0] = nx + 2 * kLutPad; bc_table_entries[1] = ny + 2 * kLutPad; bc_table_entries[2] = nz + 2 * kLutPad; }
#endif
			const uint32_t row = a.nbx * kBrickPitch;                        // elements per brick row / slab
			const uint64_t slab = (uint64_t) a.nby * row;
			for (uint32_t jj = t; jj < nz + 2 * kLutPad; jj += kThreads) {
				const uint32_t i = cell_of(jj, nz);
				const uint32_t j = i + 1 < nz ? i + 1 : i;
				const uint64_t z0 = ((i >> 3) * slab + brick_spread(BPV, a.brick_plane, 2, i & 7u)) * elem;
				const uint64_t z1 = ((j >> 3) * slab + brick_spread(BPV, a.brick_plane, 2, j & 7u)) * elem;
				if (ADDR == kAddr32) {
					lut[2 * jj] = (uint32_t) z0; lut[2 * jj + 1] = (uint32_t) z1;
				} else {
					lut[4 * jj] = (uint32_t) z0; lut[4 * jj + 1] = (uint32_t) (z0 >> 32);
					lut[4 * jj + 2] = (uint32_t) z1; lut[4 * jj + 3] = (uint32_t) (z1 >> 32);
				}
			}
			for (uint32_t j = t; j < nx + 2 * kLutPad; j += kThreads) { const uint32_t i = cell_of(j, nx); lut[L::x_at + j] = ((i >> 3) * kBrickPitch + brick_spread(BPV, a.brick_plane, 0, i & 7u)) * elem; }
			for (uint32_t j = t; j < ny + 2 * kLutPad; j += kThreads) { const uint32_t i = cell_of(j, ny); lut[L::y_at + j] = ((i >> 3) * row + brick_spread(BPV, a.brick_plane, 1, i & 7u)) * elem; }
		}
		if (t <= VR_TF_SIZE) {
			const f4 *tf4 = (const f4 *) tf_g;
			uint32_t i0 = t < VR_TF_SIZE ? t : VR_TF_SIZE - 1;
			uint32_t i1 = t + 1 < VR_TF_SIZE ? t + 1 : VR_TF_SIZE - 1;
			f4 c0 = tf4[i0], c1 = tf4[i1];
			lds.tf[t] = c0;
			f4 d; d.x = c1.x - c0.x; d.y = c1.y - c0.y; d.z = c1.z - c0.z; d.w = c1.w - c0.w;
			lds.dtf[t] = d;
		}
		for (uint32_t i = t; i < VR_ESL_VOLUME_SIZE; i += kThreads) lds.esl[i] = esl_g[i];
		if (SAMPLING == VR_SAMPLE_NEAREST && BPV == 1 && t < 256u) lds.unit[t] = (float) t / 255.0f;   // the same IEEE division, once
	}
	__syncthreads();

	uint32_t tile_x, tile_y;
	tile_to_xy(a.tiles_x, a.tiles_y, tile_of_group, blockIdx.x, tile_x, tile_y);

	// -- one wavefront = one 8x8 pixel tile; 8 waves = 32x16 pixels, 16 waves = 32x32.  Inside the wave each group of 16
	//    consecutive lanes is a 4x4-pixel block (not two 8-pixel rows): a compact block keeps the group's samples inside the
	//    fewest cache sectors whatever the view direction.
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t qd = lane >> 4;
	// Order of the 16 lanes inside the group, picked per frame by the host (vr_hip_api.cpp choose_tile_mapping): the vector
	// memory pipeline handles 4 consecutive lanes together and is fastest when their addresses share one aligned 16-byte
	// chunk, so the 4 lanes should be the 4 pixels whose samples lie closest together in the brick order.
	uint32_t gu = lane & 3u, gv = (lane >> 2) & 3u;                             // kLaneRows: lanes run along screen x
	const uint32_t order = a.lane_map & 3u, shape = kThreads == 512u ? (a.lane_map >> 2) : 0u;
	if (order == kLaneBlocks) { gu = ((lane >> 1) & 2u) | (lane & 1u); gv = ((lane >> 2) & 2u) | ((lane >> 1) & 1u); }
	else if (order == kLaneColumns) { const uint32_t t = gu; gu = gv; gv = t; }
	// Shape of the wave's pixel tile inside the 32x16-pixel workgroup tile (bits 2.. of lane_map): 0 = 8x8 (four 4x4 groups as 2x2),
	// 1 = 16 wide x 4 high (the groups side by side; the 8 waves 2 across x 4 down), 2 = 4 wide x 16 high (8 waves across).
	uint32_t wx, wy, ox, oy;
	if (shape == 1u) { wx = qd * 4u + gu; wy = gv; ox = (wave & 1u) * 16u; oy = (wave >> 1) * 4u; }
	else if (shape == 2u) { wx = gu; wy = qd * 4u + gv; ox = wave * 4u; oy = 0u; }
	else { wx = (qd & 1u) * 4u + gu; wy = (qd >> 1) * 4u + gv; ox = (wave & 3u) * 8u; oy = (wave >> 2) * 8u; }
	const uint32_t lx = tile_x * 32u + ox + wx - a.phase_x;                      // wraps for the pixels left of / below the buffer
	const uint32_t ly = tile_y * (kThreads / 32u) + oy + wy - a.phase_y;
	if (lx >= a.p.out_width || ly >= a.p.out_rows)
		return;                                     // no barrier below this point
	const uint32_t band = ly / a.p.band_rows;
	const uint32_t gy = (band * a.p.band_stride + a.p.band_first) * a.p.band_rows + (ly - band * a.p.band_rows);
	const uint32_t gx = a.p.x0 + lx;
	uint32_t *out_px = out + (size_t) ly * a.p.out_width + lx;

	// -- View::get_ray (ViewBase.h:23-35)
	f3 origin, dir;
	bool alive = gx < a.p.view.width && gy < a.p.view.height;
	{
		const f3 vo = ld3(a.p.view.origin), vd = ld3(a.p.view.direction);
		const f3 vr_ = ld3(a.p.view.right_plane), vu = ld3(a.p.view.up_plane);
		const float fx = (float) ((int) gx - (int) (a.p.view.width / 2u));
		const float fy = (float) ((int) gy - (int) (a.p.view.height / 2u));
		if (a.p.view.perspective) {
			origin = vo;
			dir = mk3(vd.x + vr_.x * fx, vd.y + vr_.y * fx, vd.z + vr_.z * fx);
			dir = mk3(dir.x + vu.x * fy, dir.y + vu.y * fy, dir.z + vu.z * fy);
		} else {
			dir = vd;
			origin = mk3(vo.x + vr_.x * fx, vo.y + vr_.y * fx, vo.z + vr_.z * fx);
			origin = mk3(origin.x + vu.x * fy, origin.y + vu.y * fy, origin.z + vu.z * fy);
		}
	}
	float kx = 0, ky = 0;
	alive = alive && intersect(origin, dir, kx, ky);
	const float step = a.p.ray_step;
	// Termination guard, once per ray instead of a counter per sample: k advances by `step` every iteration as long as
	// ky + step != ky (fp32 spacing grows with magnitude, so that holds for every k <= ky), and the march is cut after
	// kMaxRaySteps steps.  Neither condition can trigger for a view the reference can produce (k spans <= 2*sqrt(3)).
	alive = alive && (ky + step > ky);
	ky = flmin(ky, kx + step * (float) kMaxRaySteps);
	const bool hit = alive;
	f3 pt = march_point<SAMPLING>(origin, dir, kx);

	// -- empty space leaping loop (CPURenderer.cpp:18-25)
	// Cooperative look-ahead for zero-leap chains (round 4).  A ray that runs exactly along a block face (rows / columns of pixels of the
	// axis-aligned views; a few rays of every view) finds its block empty but its distance to the exit plane 0 at EVERY sample: it leaps by
	// floor(0 / step) * step = 0 and probes again one step on — up to ~1000 dependent probes by one or two lanes of a wave that is alone
	// on its SIMD at the end of the frame (the default mode's tail).  While at most VR_ESL_COOP_LANES lanes still probe and one of them
	// has just leapt by exactly 0, the whole wave evaluates that ray's next positions — the j-th active lane the position j rounds of
	// "+= 0; += step" on, formed EXACTLY (inside a binade fl(k + step) = k + round_u(step): an arithmetic progression, see colmarch_kernel;
	// else by the sequential additions) — and the ray jumps to the first position whose probe is not again "empty block, zero leap",
	// which the ordinary step below then evaluates.  Exact by construction.
#ifndef VR_ESL_COOP_LANES
#define VR_ESL_COOP_LANES 8
#endif
	if (a.p.esl) {
		bool probing = alive;
		uint64_t zero_leap = 0ull, pm;                                  // lanes whose last probe leapt by exactly 0
		int coop_pause = 0;                                             // a look-ahead that skipped fewer than four positions cost more than it saved: pause (perspective views: their chains are short)
		while ((pm = __builtin_amdgcn_ballot_w64(probing)) != 0ull) {
			const uint64_t chain = pm & zero_leap;
			if (coop_pause > 0) coop_pause--;
			else if (VR_ESL_COOP_LANES > 0 && chain != 0ull && __builtin_popcountll(pm) <= VR_ESL_COOP_LANES) {
				const int leader = __builtin_ctzll(chain);
				const f3 lo = mk3(rlane(origin.x, leader), rlane(origin.y, leader), rlane(origin.z, leader)), ld = mk3(rlane(dir.x, leader), rlane(dir.y, leader), rlane(dir.z, leader));
				const float k0 = rlane(kx, leader), kend = rlane(ky, leader);
				// (pixels outside the buffer have left the kernel: position j of the chain lives in the j-th ACTIVE lane)
				const uint64_t here = __builtin_amdgcn_ballot_w64(true);
				const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (here >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) here, 0u));
				float mine;
				{
					const float k1 = k0 + step, delta = k1 - k0, low = step - delta;
					const uint32_t e = __float_as_uint(k0) >> 23;
					const float half_ulp = __uint_as_float((e > 24u ? e - 24u : 1u) << 23), klast = VR_FMA(63.0f, delta, k0);
					const bool fast = rfl((e > 24u && (__float_as_uint(klast) >> 23) == e && __builtin_fabsf(low) != half_ulp && delta > 0.0f) ? 1u : 0u) != 0u;
					if (fast) mine = VR_FMA((float) rank, delta, k0);
					else {
						float run = k0;
						mine = k0;
						#pragma nounroll
						for (uint32_t j = 1; j < 64u; j++) { run = run + step; mine = rank >= j ? run : mine; }
					}
				}
				const f3 lp = march_point<SAMPLING>(lo, ld, mine);
				const BlockIdx lb = block_index(a, lp);
				bool goes_on = mine <= kend && block_empty(lds, lb);
				if (goes_on) goes_on = leap_empty_space(a, lb, lp, ld) == 0.0f;
				const uint64_t stops = here & ~__builtin_amdgcn_ballot_w64(goes_on);
				const int first = stops != 0ull ? __builtin_ctzll(stops) : 63 - __builtin_clzll(here);      // the first position that does not go on (else the last: the step below re-evaluates it)
				const float k_new = rlane(mine, first);
				if ((int) lane == leader) { kx = k_new; pt = march_point<SAMPLING>(origin, dir, kx); }
				if (__builtin_popcountll(here & ((1ull << first) - 1ull)) < 4) coop_pause = 24;
			}
			bool zl = false;
			if (probing) {
				const BlockIdx blk = block_index(a, pt);
				if (kx <= ky && block_empty(lds, blk)) {
					const float leap = leap_empty_space(a, blk, pt, dir);
					zl = leap == 0.0f;
					kx += leap;
					kx += step;
					pt = march_point<SAMPLING>(origin, dir, kx);
				} else {
					probing = false;
				}
			}
			zero_leap = __builtin_amdgcn_ballot_w64(zl);
		}
	}
	alive = alive && (kx <= ky);                    // CPURenderer.cpp:26-27: fully empty ray — pixel keeps the clear value
	const bool visible = alive;

	// -- colour accumulation loop (CPURenderer.cpp:29-39 / GPURenderer4.cu:75-86), front to back, premultiplied
	f4 acc; acc.x = acc.y = acc.z = acc.w = 0.0f;
	const f3 light = ld3(a.p.view.light_pos);
	const float kd = a.p.light_kd;
	const bool lit = kd > 0.01f;
	const float threshold = a.p.ray_threshold;
	if (SAMPLING == VR_SAMPLE_NEAREST) {
		// Same loop shape as the TRILINEAR branch below (prefetch of sample i+1, finished lanes composited with weight 0,
		// per-wave transparent-sample shortcut, two samples per exit vote); the arithmetic is the reference's, unfused:
		// acc + cur * 0 == acc exactly, and map_float_int clamps every index, so speculative fetches stay in bounds.
		// In-bounds speculative fetches without clamping to 0: every fetch position is taken at min(k, ky), on the ray's own
		// segment inside the cube (for live lanes that IS the sample position); lanes without a segment march position 0.
		// clamp_fetch (far-away views, see TRILINEAR) falls back to the reference's two-sided clamp.
		const int tf_zero_idx = (int) a.tf_zero_below;                 // entries 0..tf_zero_idx are (0,0,0,0)
		const int opaque_above = (tf_zero_idx + 1) * VR_TF_RATIO * (BPV == 1 ? 1 : 256) - 1;
		uint64_t live = __builtin_amdgcn_ballot_w64(alive);            // liveness as one scalar wave mask (see TRILINEAR)
		if (!alive) { kx = 0.0f; ky = 0.0f; origin = mk3(0.0f, 0.0f, 0.0f); dir = origin; pt = origin; }
		constexpr bool kTables = is_brick_table_layout(LAYOUT) && ADDR != kAddrWide;  // padded address tables (kLutPad)
		auto march = [&](auto clamp_tag, auto scaled_tag) {
		constexpr bool kClamp = decltype(clamp_tag)::value;
		// kFree: fetch positions are NOT pulled back to the ray's segment — a finished lane's k simply stops (its step becomes 0, a
		// wave-uniform branch when a lane finishes), so a speculative position lies at most two steps past the exit point, inside
		// the table padding.  Otherwise (no tables, or the clamping variant) every fetch position is taken at min(k, ky) / clamped.
		constexpr bool kFree = !kClamp && kTables;
		constexpr bool kScaled = kFree && decltype(scaled_tag)::value;               // sample_nearest_scaled
		constexpr bool kLazy = kFree && VR_LAZY_EXIT;
		const f3 so = kScaled ? mk3(origin.x * a.half_x, origin.y * a.half_y, origin.z * a.half_z) : origin;
		const f3 sd = kScaled ? mk3(dir.x * a.half_x, dir.y * a.half_y, dir.z * a.half_z) : dir;
		float step_v = kFree ? select_lanes(live, step) : step;
		auto fetch_at = [&](float k) {
			if (!kClamp && !kFree) k = __builtin_fminf(k, ky);
			const f3 p = mk3(so.x + sd.x * k, so.y + sd.y * k, so.z + sd.z * k);      // CPURenderer.cpp:17,24,38: two roundings per axis
			if (kClamp) return sample_nearest<BPV, ADDR, LAYOUT, true>(vol, a, lut, p);
			if (kScaled) return sample_nearest_scaled<BPV, ADDR, LAYOUT, true>(vol, a, lut, p);
			return sample_nearest_incube<BPV, ADDR, LAYOUT, true>(vol, a, lut, p);
		};
		constexpr bool kManaged = Managed<BPV, ADDR, LAYOUT>::value;
		// kDepth samples ahead: slot j carries the fetched word and the k of its sample
		uint32_t word[kSlots]; float ks[kSlots];
		ks[0] = kx; word[kSlots - 1] = 0;
		static_for<0, kDepth>([&](auto j) {
			if constexpr (j.value > 0) ks[j.value] = ks[j.value - 1] + step_v;
			word[j.value] = fetch_at(ks[j.value]);
		});
		auto step_sample = [&](auto jc) {
			constexpr int c = decltype(jc)::value, n = (c + kDepth) % kSlots, nx = (c + 1) % kSlots;
			ks[n] = ks[(n + kSlots - 1) % kSlots] + step_v;
			word[n] = fetch_at(ks[n]);
			__builtin_amdgcn_sched_barrier(0);
			kx = ks[c];
			const float kn = ks[nx];
			(void) kn;
			// Nothing that reads the fetched word may move above this point: the compiler otherwise hoists such work to the loop latch,
			// behind an s_waitcnt vmcnt(0) that drains the prefetches in flight once per iteration.
			if (kManaged) { pin(word[c]); managed_wait<kDepth>(); pin(word[c]); }
			else pin(word[c]);
			const uint32_t s = voxel_of<BPV, LAYOUT>(word[c]);
			// transfer_fn[sample / TF_RATIO] (CPURenderer.cpp:31) is (0,0,0,0) for index <= tf_zero_idx, i.e. for
			// s <= opaque_above = (tf_zero_idx + 1) * TF_RATIO * (1 or 256) - 1: tested on the voxel itself, the index is only formed
			// by the few samples that get past the test
			if ((__builtin_amdgcn_sicmp((int) s, opaque_above, kIcmpSGT) & live) != 0ull && VR_OPEN_LANES(acc.w, live) != 0ull) {
				if (kLazy) {                                                                      // this sample's own segment test (see the loop)
					const uint64_t inside = __builtin_amdgcn_fcmpf(kx, ky, kFcmpOLE);
					if ((live & ~inside) != 0ull) { live &= inside; step_v = select_lanes(live, step); }       // a lane that is dropped here stops HERE:
				}                                                                                 // the rotation test only looks at lanes still marked live
				uint32_t idx = (BPV == 1 ? s : (s >> 8)) / VR_TF_RATIO;
				asm volatile("" : "+v"(idx));                                                       // keep the index arithmetic inside the branch
				f4 cur = lds.tf[idx];
				const uint64_t shaded = lit ? (__builtin_amdgcn_fcmpf(cur.w, 0.05f, kFcmpOGT) & live) : 0ull;
				if (shaded != 0ull) {                                                             // RaycasterBase.h:87-98 shade
					const float raw = BPV == 1 ? 255.0f : 65535.0f;
					pt = march_point<SAMPLING>(origin, dir, kx);                                  // the sample's own position
					f3 d = mk3(light.x - pt.x, light.y - pt.y, light.z - pt.z);
					float inv = 1.0f / __builtin_sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
					f3 l = mk3(d.x * inv, d.y * inv, d.z * inv);
					f3 ps = mk3(pt.x + l.x * 0.01f, pt.y + l.y * 0.01f, pt.z + l.z * 0.01f);
					const uint32_t s_l = voxel_of<BPV, LAYOUT>(sample_nearest<BPV, ADDR, LAYOUT>(vol, a, lut, ps));
					const float sl = BPV == 1 ? lds.unit[s_l] : (float) s_l / raw;                  // RaycasterBase.h:93-96
					const float sc = BPV == 1 ? lds.unit[s] : (float) s / raw;
					const float diffuse = select_lanes(shaded, (sl - sc) * kd);                   // x + 0 == x: unshaded lanes unchanged
					cur.x += diffuse; cur.y += diffuse; cur.z += diffuse;
				}
				const float t = select_lanes(live, 1 - acc.w);                                    // CPURenderer.cpp:34
				acc.x = acc.x + cur.x * t; acc.y = acc.y + cur.y * t;
				acc.z = acc.z + cur.z * t; acc.w = acc.w + cur.w * t;
				live &= ~__builtin_amdgcn_fcmpf(acc.w, threshold, kFcmpOGT);                      // CPURenderer.cpp:35-36
				if (kFree) step_v = select_lanes(live, step);                                     // terminated rays stop marching
			}
			if (!kLazy) {
				const uint64_t still = __builtin_amdgcn_fcmpf(kn, ky, kFcmpOLE);
				if (kFree && (live & ~still) != 0ull) step_v = select_lanes(live & still, step);      // a lane has just left its segment
				live &= still;
			}
		};
		// kLazy: `live` is brought up to date once per rotation of the slots (the k of the next sample against the end of the segment),
		// and by every sample that composites, for itself.  In between a finished lane still counts as live: its fetches lie in the
		// table padding (kOverrunSteps), a transparent sample does nothing with it, a compositing sample tests it first.
		while (live != 0ull) {
			static_for<0, kSlots>(step_sample);
			if (kLazy) {
				const uint64_t still = __builtin_amdgcn_fcmpf(ks[0], ky, kFcmpOLE);
				if ((live & ~still) != 0ull) step_v = select_lanes(live & still, step);
				live &= still;
			}
		}
		if (kManaged) {                                              // nothing in flight into registers we release
			static_for<0, kSlots>([&](auto j) { pin(word[j.value]); });
			managed_wait<0>();
			static_for<0, kSlots>([&](auto j) { pin(word[j.value]); });
		}
		};
		if (a.clamp_fetch) march(std::true_type(), std::false_type());
		else if (kTables && a.near_scaled) march(std::false_type(), std::true_type());
		else march(std::false_type(), std::false_type());
	} else {
		// texel-space ray: coordinate = fma(k, A, B) (see oracle/vr_oracle.c axis_setup)
		f3 A = mk3(dir.x * a.half_x, dir.y * a.half_y, dir.z * a.half_z);
		f3 B = mk3(VR_FMA(origin.x, a.half_x, a.off_x), VR_FMA(origin.y, a.half_y, a.off_y), VR_FMA(origin.z, a.half_z, a.off_z));
		// Lanes that are finished keep executing an in-bounds fetch with a zero weight instead of being masked off:
		// acc = fma(cur, 0, acc) leaves them bit-for-bit unchanged, and the loop body needs no per-lane control flow
		// except the shading block.  The wave leaves when no lane is alive.
		// In bounds without clamping three coordinates per sample: every fetch position is taken at min(k, ky), i.e. on the
		// ray's own segment inside the cube, where truncation alone gives the clamped cell (tri_issue); lanes that never
		// had a segment march the constant position 0.  The host switches the coordinate clamp back on (clamp_fetch) for
		// views so far from the volume that fp32 rounding of the coordinates could leave (-1, N).
		if (!alive) { kx = 0.0f; ky = 0.0f; A = mk3(0.0f, 0.0f, 0.0f); B = A; }
		// kLayoutRunDual, tile on the copy along y: the texel-space ray is kept with y and z EXCHANGED — that is what the tile's
		// address tables index (staging above); the few samples that are filtered put the two coordinates back (same fma, same bits)
		if (LAYOUT == kLayoutRunDual && group_sched[3] != 0u) { float t = A.y; A.y = A.z; A.z = t; t = B.y; B.y = B.z; B.z = t; }
		constexpr bool kTables = LAYOUT != kLayoutLinear && ADDR != kAddrWide;        // padded address tables (kLutPad)
		auto march = [&](auto clamp_tag) {                    // instantiated for both settings: no per-sample test of the flag
			constexpr bool kClamp = decltype(clamp_tag)::value;
			constexpr bool kFree = !kClamp && kTables;        // see the NEAREST loop: no min(k, ky), finished lanes stop instead
			constexpr bool kLazy = kFree && VR_LAZY_EXIT;     // exit test once per rotation of the slots (see the NEAREST loop)
			// Software pipeline: the loads of sample i+2 are issued before sample i is
			// unpacked, filtered and composited, so memory round trips overlap the arithmetic inside every wave (on top of the
			// 8 waves per SIMD).  The body is written once (`step_sample`) and instantiated once per fetch slot and iteration
			// with the slots rotated: no register copies, one exit vote per three samples (a finished wave at worst composites
			// two more weight-0 samples).
			auto issue = [&](float k) {
				if (!kClamp && !kFree) k = __builtin_fminf(k, ky);
				return tri_issue<BPV, ADDR, LAYOUT, true>(vol, a, lut, VR_FMA(k, A.x, B.x), VR_FMA(k, A.y, B.y), VR_FMA(k, A.z, B.z), kClamp);
			};
			constexpr bool kManaged = ManagedTri<BPV, ADDR, LAYOUT>::value;
			// Lane liveness is kept as ONE 64-bit wave mask in scalar registers (`live`), updated with v_cmp results
			// (__builtin_amdgcn_fcmpf returns the wave's compare mask) — no per-lane control flow, no mask <-> VGPR round trips:
			// the body is straight-line code with two wave-uniform branches (transparent shortcut, shading block).
			uint64_t live = __builtin_amdgcn_ballot_w64(alive);
			float step_v = kFree ? select_lanes(live, step) : step;       // per-lane step: 0 once the lane is finished
			// kDepth samples ahead (three for 2-byte voxels, whose slots hold four words: 64 VGPRs keep 8 waves per SIMD):
			// slot j carries the fetched words and the k of its sample
			constexpr int kDepth = is_run_layout(LAYOUT) ? kRunDepth : (BPV == 1 ? vr::kDepth : (vr::kDepth > kDepthTwoByte ? kDepthTwoByte : vr::kDepth)), kSlots = kDepth + 1;
			TriFetch<BPV, LAYOUT> f[kSlots]; float ks[kSlots];
			ks[0] = kx;
			f[kSlots - 1].w0 = f[kSlots - 1].w1 = f[kSlots - 1].w2 = f[kSlots - 1].w3 = 0; f[kSlots - 1].q = 0; f[kSlots - 1].q2 = 0; f[kSlots - 1].o = (u32x4) (0u);
			static_for<0, kDepth>([&](auto j) {
				if constexpr (j.value > 0) ks[j.value] = ks[j.value - 1] + step_v;
				f[j.value] = issue(ks[j.value]);
			});
			auto step_sample = [&](auto jc) {
				constexpr int c = decltype(jc)::value, n = (c + kDepth) % kSlots, nx = (c + 1) % kSlots;
				ks[n] = ks[(n + kSlots - 1) % kSlots] + step_v;
				f[n] = issue(ks[n]);
#ifndef VR_NO_SCHED_BARRIER
				__builtin_amdgcn_sched_barrier(0);
#endif
				TriFetch<BPV, LAYOUT> &cur = f[c];
				kx = ks[c];
				const float kn = ks[nx];
				(void) kn;
			(void) kn;
				// Exact shortcuts, decided per wave.  Entries 0..tf_zero_below of the premultiplied TF are all zero (the reference's
				// default TF is zero below 10 % density), so a sample whose TF coordinate tb is <= tf_zero_below has colour
				// (0,0,0,0), is never shaded (alpha 0 <= 0.05) and leaves acc bit-for-bit unchanged.
				//  (1) before any arithmetic: if all 8 corner voxels of every live lane are below the power of two `skip_below`
				//      (a bit test on the packed words), the interpolated value is too — a lerp never leaves [min, max] of its
				//      operands, fp32 rounding included — and skip_below was chosen on the host so that tb <= tf_zero_below
				//      follows: the wave skips unpacking, the 7 lerps and everything after them;
				//  (2) after the interpolation: the same test on tb itself skips the LDS lookups, the shading test and the composite.
				// pinned below the issue (see the NEAREST loop), on the slot's own registers: waits only for the slot's loads
				if (kManaged && is_run_layout(LAYOUT)) {               // one 8-byte gather per slot: kDepth younger ones may be in flight
					pin(cur.q); managed_wait<kDepth>(); pin(cur.q);
					cur.w0 = (uint32_t) cur.q; cur.w1 = (uint32_t) (cur.q >> 32);
				} else if (kManaged && LAYOUT == kLayoutOct) {         // one 16-byte gather per slot (2-byte voxels, oct bricks)
					pin(cur.o); managed_wait<kDepth>(); pin(cur.o);
					cur.w0 = cur.o.x; cur.w1 = cur.o.y; cur.w2 = cur.o.z; cur.w3 = cur.o.w;
				} else if (kManaged && BPV == 2) {                     // two 8-byte gathers per slot (2-byte voxels)
					pin(cur.q, cur.q2); managed_wait<2 * kDepth>(); pin(cur.q, cur.q2);
					cur.w0 = (uint32_t) cur.q; cur.w1 = (uint32_t) (cur.q >> 32); cur.w2 = (uint32_t) cur.q2; cur.w3 = (uint32_t) (cur.q2 >> 32);
				} else if (kManaged) {                                 // two 4-byte gathers per slot: 2 * kDepth younger ones
					pin(cur.w0, cur.w1); managed_wait<2 * kDepth>(); pin(cur.w0, cur.w1);
				} else if (LAYOUT != kLayoutLinear && BPV == 1) pin(cur.w0, cur.w1);
				else pin(cur.w0, cur.w1, cur.w2, cur.w3);
				const TriFetch<BPV, LAYOUT> &now = cur;
				uint32_t corners;
				if (LAYOUT != kLayoutLinear) corners = BPV == 1 ? (now.w0 | now.w1) : (now.w0 | now.w1 | now.w2 | now.w3);
				else                          corners = now.w0 | now.w1 | now.w2 | now.w3;
				// skip_cmp is 0; a TF without leading zero entries (nothing may be skipped) comes with skip_mask 0 and skip_cmp 1: 0 != 1 always
				if ((__builtin_amdgcn_uicmp(corners & a.skip_mask, a.skip_cmp, kIcmpNE) & live) != 0ull && VR_OPEN_LANES(acc.w, live) != 0ull) {
				if (kLazy) {
					const uint64_t inside = __builtin_amdgcn_fcmpf(kx, ky, kFcmpOLE);
					if ((live & ~inside) != 0ull) { live &= inside; step_v = select_lanes(live, step); }
				}
				const float xb = VR_FMA(kx, A.x, B.x);                                                            // where the words were fetched
				float yb = VR_FMA(kx, A.y, B.y), zb = VR_FMA(kx, A.z, B.z);
				// kLayoutRunDual: read again from LDS here, so that the flag occupies no register across the march — through an index the
				// compiler cannot see through, or it hoists the read out of the loop (a volatile access would become a FLAT load with a
				// vmcnt(0) wait behind it: the whole prefetch pipeline drained per filtered sample)
				bool along_y = false;
				if (LAYOUT == kLayoutRunDual) {
					uint32_t opaque_zero;
					asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
					along_y = group_sched[3u + opaque_zero] != 0u;
				}
				if (LAYOUT == kLayoutRunDual && along_y) { const float t = yb; yb = zb; zb = t; }                 // the true coordinates again
				const float raw = tri_resolve<BPV, LAYOUT, kQ8>(now, a, xb, yb, zb, along_y);                 // GPURenderer4.cu:76
				// GPURenderer4.cu:77 filtered TF: texel coordinate tb, entries floor(tb) and floor(tb)+1
				const float tb = __builtin_amdgcn_fmed3f(VR_FMA(raw, a.tf_scale, -0.5f), 0.0f, (float) (VR_TF_SIZE - 1));
				if ((__builtin_amdgcn_fcmpf(tb, a.tf_zero_below, kFcmpOGE) & live) != 0ull) {
					f4 c;
					{
						const uint32_t i = (uint32_t) (int) tb;
						const float w = filter_weight<kQ8>(__builtin_amdgcn_fractf(tb));
						const f4 c0 = lds.tf[i], dc = lds.dtf[i];
						c.x = VR_FMA(w, dc.x, c0.x); c.y = VR_FMA(w, dc.y, c0.y);
						c.z = VR_FMA(w, dc.z, c0.z); c.w = VR_FMA(w, dc.w, c0.w);
					}
					const uint64_t shaded = lit ? (__builtin_amdgcn_fcmpf(c.w, 0.05f, kFcmpOGT) & live) : 0ull;   // GPURenderer4.cu:78
					if (shaded != 0ull) {                                                              // GPURenderer4.cu:41-51 shade_texture
						const f3 p3 = march_point<SAMPLING>(origin, dir, kx);
						const f3 d = mk3(light.x - p3.x, light.y - p3.y, light.z - p3.z);
						const float inv = rsqrt_nr(VR_FMA(d.z, d.z, VR_FMA(d.y, d.y, d.x * d.x)));
						const float lx = VR_FMA(d.x * inv, a.lh_x, xb), ly = VR_FMA(d.y * inv, a.lh_y, yb), lz = VR_FMA(d.z * inv, a.lh_z, zb);
						TriFetch<BPV, LAYOUT> lf;
						if (LAYOUT == kLayoutRunDual) {                     // clamp with the true bounds, then hand the tile's table order over
							const float cy = __builtin_amdgcn_fmed3f(ly, 0.0f, a.max_y), cz = __builtin_amdgcn_fmed3f(lz, 0.0f, a.max_z);
							lf = tri_issue<BPV, ADDR, LAYOUT>(vol, a, lut, __builtin_amdgcn_fmed3f(lx, 0.0f, a.max_x), along_y ? cz : cy, along_y ? cy : cz, false);
						} else lf = tri_issue<BPV, ADDR, LAYOUT>(vol, a, lut, lx, ly, lz, true);
						const float raw_l = tri_resolve<BPV, LAYOUT, kQ8>(lf, a, lx, ly, lz, along_y);
						const float diffuse = select_lanes(shaded, (raw_l - raw) * a.kd_scaled);       // 0 for lanes that are not shaded
						c.x += diffuse; c.y += diffuse; c.z += diffuse;
					}
					const float t = select_lanes(live, 1 - acc.w);                                     // finished lanes: weight 0
					acc.x = VR_FMA(c.x, t, acc.x); acc.y = VR_FMA(c.y, t, acc.y);
					acc.z = VR_FMA(c.z, t, acc.z); acc.w = VR_FMA(c.w, t, acc.w);
					live &= ~__builtin_amdgcn_fcmpf(acc.w, threshold, kFcmpOGT);                        // ERT (CPURenderer.cpp:35-36)
					if (kFree) step_v = select_lanes(live, step);                                       // terminated rays stop marching
				}
				}
				if (!kLazy) {
					const uint64_t still = __builtin_amdgcn_fcmpf(kn, ky, kFcmpOLE);                    // the loop condition
					if (kFree && (live & ~still) != 0ull) step_v = select_lanes(live & still, step);    // a lane has just left its segment
					live &= still;
				}
			};
			while (live != 0ull) {
				static_for<0, kSlots>(step_sample);
				if (kLazy) {
					const uint64_t still = __builtin_amdgcn_fcmpf(ks[0], ky, kFcmpOLE);
					if ((live & ~still) != 0ull) step_v = select_lanes(live & still, step);
					live &= still;
				}
			}
			if (kManaged) {                                            // nothing in flight into registers we release
				auto pin_slot = [&](auto j) { if (is_run_layout(LAYOUT)) pin(f[j.value].q); else if (LAYOUT == kLayoutOct) pin(f[j.value].o); else if (BPV == 2) pin(f[j.value].q, f[j.value].q2); else pin(f[j.value].w0, f[j.value].w1); };
				static_for<0, kSlots>(pin_slot);
				managed_wait<0>();
				static_for<0, kSlots>(pin_slot);
			}
		};
		if (a.clamp_fetch) march(std::true_type()); else march(std::false_type());
	}

	// -- RaycasterBase.h:44-50 write_color (+ the fused clear: misses and fully-empty rays store 0)
	uint32_t rgba = 0;
	if (hit && visible) {
		rgba = map_float_int(acc.x, 256) | (map_float_int(acc.y, 256) << 8) |
		       (map_float_int(acc.z, 256) << 16) | (map_float_int(acc.w, 256) << 24);
	}
	*out_px = rgba;
	// cost of the tile = the end of its last wave after the start of the workgroup, in 64-cycle units (at least 1); every wave
	// reports, through the first of its lanes that is still here
	const uint32_t lane_id = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
	if (lane_id == (uint32_t) __builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) {
		const uint32_t *w = group_sched;
		const uint64_t slot = ((uint64_t) w[2] << 32) | w[1];
		if (slot != 0ull) atomicMax((uint32_t *) (uintptr_t) slot, ((uint32_t) (__builtin_readcyclecounter() >> 6) - w[0]) | 1u);
	}
}

// ---- the column march (kLayoutColumn, round 4) -----------------------------------------------------------------------------------
//
// Orthogonal views along a volume axis m, full march (no leaping, no early termination).  What makes them special (measured on the
// benchmark poses, scripts: every 8x8-pixel tile of the three axis-aligned views): all rays of a wave share kx bit for bit, so they
// share the whole k sequence, and with the direction's lateral components at most rounding noise (4e-8) a ray stays in ONE cell column
// (u,v) but for at most one cell flip per lateral axis.  The march therefore runs on wave-uniform state:
//   * the cell along m of a sample is the same for all 64 lanes: k += step, fma, float -> int, v_readfirstlane — four vector
//     instructions per sample instead of the 15 + 3 LDS lookups of the general address chain;
//   * ONE aligned 16-byte gather per lane and WINDOW (four consecutive quad elements of the lane's column = three cells, vr_device.h),
//     prefetched kColDepth windows ahead with exact s_waitcnt vmcnt accounting (one gather per window: static);
//   * the transparency test is done ONCE per window on all four elements: a window that is transparent for every live lane lets its
//     ~3 samples pass with the uniform chain alone (exact: every pair of elements the samples would test is part of the window);
//   * a lane whose column flips (monotone coordinate: at most once per axis, at the smallest float t with cell(t) != cell(kx), found by
//     bisection once per ray) reads the column of the state PREDICTED for a window's first sample; a window whose samples may see a
//     different state (t inside the window's k range, or between prediction and truth) takes the careful path: explicit per-sample
//     fetches from each lane's true column.
// Waves whose live lanes do not share kx and the coordinate along m (none on pose (0,0,0) and (90,0,0), 256 of 65536 on (180,90,0)),
// whose columns would flip by more than one cell, or whose lateral spread leaves the 32-bit offset range, march per lane with explicit
// fetches from the same copy (exact, unpipelined).  Arithmetic per composited sample is the general kernel's, expression by expression.
#ifndef VR_COL_DEPTH
#define VR_COL_DEPTH 3
#endif
constexpr int kColDepth = VR_COL_DEPTH, kColSlots = kColDepth + 1;

__device__ __forceinline__ void managed_load128_s(u32x4 &dst, uint32_t byte_offset, uint64_t base) {     // window gather: scalar base + per-lane offset
	asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(byte_offset), "s"(base));
}
template <int I> __device__ __forceinline__ float comp3(const f3 &v) { return I == 0 ? v.x : (I == 1 ? v.y : v.z); }

// the eight corner voxels of a sample out of the two quad elements along m (w0 = march index i, w1 = i + 1) -> trilinear value;
// element bytes are (u,v), (u+1,v), (u,v+1), (u+1,v+1) with (u,v) the lateral axes of m in increasing order; lerps in x, y, z order
template <int M, bool Q8>
__device__ __forceinline__ float col_resolve(uint32_t w0, uint32_t w1, float max_x, float max_y, float max_z, float xb, float yb, float zb) {
	const float ax = filter_weight<Q8>(__builtin_amdgcn_fractf(__builtin_amdgcn_fmed3f(xb, 0.0f, max_x)));
	const float ay = filter_weight<Q8>(__builtin_amdgcn_fractf(__builtin_amdgcn_fmed3f(yb, 0.0f, max_y)));
	const float az = filter_weight<Q8>(__builtin_amdgcn_fractf(__builtin_amdgcn_fmed3f(zb, 0.0f, max_z)));
	const float p0 = (float) (w0 & 0xffu), p1 = (float) ((w0 >> 8) & 0xffu), p2 = (float) ((w0 >> 16) & 0xffu), p3 = (float) (w0 >> 24);
	const float q0 = (float) (w1 & 0xffu), q1 = (float) ((w1 >> 8) & 0xffu), q2 = (float) ((w1 >> 16) & 0xffu), q3 = (float) (w1 >> 24);
	float v000, v100, v010, v110, v001, v101, v011, v111;
	if (M == 2)      { v000 = p0; v100 = p1; v010 = p2; v110 = p3; v001 = q0; v101 = q1; v011 = q2; v111 = q3; }      // (u,v) = (x,y), pair along z
	else if (M == 1) { v000 = p0; v100 = p1; v001 = p2; v101 = p3; v010 = q0; v110 = q1; v011 = q2; v111 = q3; }      // (u,v) = (x,z), pair along y
	else             { v000 = p0; v010 = p1; v001 = p2; v011 = p3; v100 = q0; v110 = q1; v101 = q2; v111 = q3; }      // (u,v) = (y,z), pair along x
	const float c00 = lerp(v000, v100, ax), c10 = lerp(v010, v110, ax);
	const float c01 = lerp(v001, v101, ax), c11 = lerp(v011, v111, ax);
	const float c0 = lerp(c00, c10, ay), c1 = lerp(c01, c11, ay);
	return lerp(c0, c1, az);
}

// FLIPS: the instantiation that follows lanes through a change of their cell column (views whose direction carries rounding noise
// in its lateral components); without it (lateral components exactly 0: no lane can flip — the host decides) a wave that does flip
// marches per lane.  Two kernels rather than two loops in one: each stays inside 64 VGPRs / 80 SGPRs without spilling.
template <int SAMPLING, int M, bool FLIPS>
#ifndef VR_COL_WAVES
#define VR_COL_WAVES 4
#endif
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(VR_COL_WAVES, 8)))       // 8 waves per SIMD wanted (64 VGPRs, 80 SGPRs), never by spilling: tests/test_abi.py checks the built kernels
void colmarch_kernel(const RayKernelArgs a, const uint8_t *__restrict__ copy, const float *__restrict__ tf_g, uint32_t *__restrict__ out) {
	constexpr bool kQ8 = SAMPLING == VR_SAMPLE_TRILINEAR_Q8;
	constexpr int U = M == 0 ? 1 : 0, V = M == 2 ? 1 : 2;
	typedef const RayKernelArgs __attribute__((address_space(4))) *ConstArgs;
	__shared__ f4 tf_l[VR_TF_SIZE + 1], dtf_l[VR_TF_SIZE + 1];
	__shared__ f4 org_l[512];                                           // every thread's ray origin (see origin_again)
	{
		const uint32_t t = threadIdx.x;
		if (t <= VR_TF_SIZE) {
			const f4 *tf4 = (const f4 *) tf_g;
			const uint32_t i0 = t < VR_TF_SIZE ? t : VR_TF_SIZE - 1, i1 = t + 1 < VR_TF_SIZE ? t + 1 : VR_TF_SIZE - 1;
			const f4 c0 = tf4[i0], c1 = tf4[i1];
			tf_l[t] = c0;
			f4 d; d.x = c1.x - c0.x; d.y = c1.y - c0.y; d.z = c1.z - c0.z; d.w = c1.w - c0.w;
			dtf_l[t] = d;
		}
	}
	__syncthreads();
	uint32_t tile_x, tile_y;
	tile_to_xy<VR_COL_XCD_MODE>(a.tiles_x, a.tiles_y, blockIdx.x, blockIdx.x, tile_x, tile_y);
	// pixel of this lane: the general kernel's mapping (lane order, 8x8 waves, tile phase)
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, qd = lane >> 4;
	uint32_t gu = lane & 3u, gv = (lane >> 2) & 3u;
	const uint32_t order = a.lane_map & 3u;
	if (order == kLaneBlocks) { gu = ((lane >> 1) & 2u) | (lane & 1u); gv = ((lane >> 2) & 2u) | ((lane >> 1) & 1u); }
	else if (order == kLaneColumns) { const uint32_t t = gu; gu = gv; gv = t; }
	const uint32_t wx = (qd & 1u) * 4u + gu, wy = (qd >> 1) * 4u + gv, ox = (wave & 3u) * 8u, oy = (wave >> 2) * 8u;
	const uint32_t lx = tile_x * 32u + ox + wx - a.phase_x, ly = tile_y * 16u + oy + wy - a.phase_y;
	// lanes outside the buffer stay in the wave (the batched sample sequence below lives in all 64 lanes): no segment, no store
	const bool in_frame = lx < a.p.out_width && ly < a.p.out_rows;
	const uint32_t band = ly / a.p.band_rows;
	const uint32_t gy = (band * a.p.band_stride + a.p.band_first) * a.p.band_rows + (ly - band * a.p.band_rows);
	const uint32_t gx = a.p.x0 + lx;
	// where the pixel goes: 0xffffffff for lanes outside the buffer (frames are at most 65535 x 65535 pixels, validate_params: every real index is smaller)
	uint32_t out_index = in_frame ? ly * a.p.out_width + lx : 0xffffffffu;

	// -- View::get_ray (ViewBase.h:23-35), orthogonal branch only (the host never launches this kernel for a perspective view)
	bool alive = in_frame && gx < a.p.view.width && gy < a.p.view.height;
	const f3 dir = ld3(a.p.view.direction);
	// (the march does not keep the ray origin in three registers: the samples that are shaded — the dense path is bound by its vector
	// instructions — read it back from the thread's own LDS slot, one ds_read_b128; `org_slot` is the slot's byte offset)
	f3 origin;
	{
		const float fx = (float) ((int) gx - (int) (a.p.view.width / 2u)), fy = (float) ((int) gy - (int) (a.p.view.height / 2u));
		const f3 o = mk3(a.p.view.origin[0] + a.p.view.right_plane[0] * fx, a.p.view.origin[1] + a.p.view.right_plane[1] * fx, a.p.view.origin[2] + a.p.view.right_plane[2] * fx);
		origin = mk3(o.x + a.p.view.up_plane[0] * fy, o.y + a.p.view.up_plane[1] * fy, o.z + a.p.view.up_plane[2] * fy);
	}
	uint32_t org_slot = threadIdx.x * (uint32_t) sizeof(f4);
	{ f4 o4; o4.x = origin.x; o4.y = origin.y; o4.z = origin.z; o4.w = 0.0f; org_l[threadIdx.x] = o4; }      // read by this thread only: no barrier
	auto origin_again = [&]() { pin(org_slot); const f4 o4 = *(const f4 *) ((const char *) org_l + org_slot); return mk3(o4.x, o4.y, o4.z); };
	float kx = 0, ky = 0;
	alive = alive && intersect(origin, dir, kx, ky);
	const float step = a.p.ray_step;
	alive = alive && (ky + step > ky);                                       // termination guard (see raymarch_kernel)
	ky = flmin(ky, kx + step * (float) kMaxRaySteps);
	const uint64_t alive_mask = __builtin_amdgcn_ballot_w64(alive);
	if (alive_mask == 0ull) { if (in_frame) out[out_index] = 0u; return; }
	if (!alive) ky = -1.0f;                                                  // (also what the final store reads "no segment" from)

	// texel-space ray (oracle/vr_oracle.c axis_setup): coordinate = fma(k, A, B); A is wave-uniform (orthogonal view)
	// (the same value in every lane, formed by the vector unit: moved to scalar registers so that it does not occupy three VGPRs for the whole march)
	auto uni = [](float v) { return __uint_as_float(rfl(__float_as_uint(v))); };
	const f3 A = mk3(uni(dir.x * a.half_x), uni(dir.y * a.half_y), uni(dir.z * a.half_z));
	const f3 B = mk3(VR_FMA(origin.x, a.half_x, a.off_x), VR_FMA(origin.y, a.half_y, a.off_y), VR_FMA(origin.z, a.half_z, a.off_z));
	const float Am = comp3<M>(A), Au = comp3<U>(A), Av = comp3<V>(A);
	const float Bm = comp3<M>(B), Bu = comp3<U>(B), Bv = comp3<V>(B);
	const uint32_t dim_u = U == 0 ? a.dim_x : a.dim_y, dim_m = M == 0 ? a.dim_x : (M == 1 ? a.dim_y : a.dim_z);
	const float max_u = U == 0 ? a.max_x : a.max_y, max_v = V == 1 ? a.max_y : a.max_z;
	const uint32_t nbu = col_blocks(dim_u), nw = col_windows(dim_m);
	const uint64_t stride_u = (uint64_t) nw * kColBlockBytes, stride_v = (uint64_t) nbu * stride_u;       // bytes between lateral blocks
	auto f_u = [&](int c) { return (uint64_t) ((uint32_t) c >> kColEdgeLog2) * stride_u + ((uint32_t) c & kColEdgeMask) * kColWindowBytes; };
	auto f_v = [&](int c) { return (uint64_t) ((uint32_t) c >> kColEdgeLog2) * stride_v + ((uint32_t) c & kColEdgeMask) * kColRowBytes; };

	f4 acc; acc.x = acc.y = acc.z = acc.w = 0.0f;
	uint64_t live = alive_mask;
	float k = kx;                                   // the sample being processed (wave-uniform on the column path, per lane on the fallback)

	// The transparent march runs on a dozen scalars; everything else the DENSE path needs (clamp bounds, shading constants, the light,
	// the view direction) is read again from the kernel-argument segment where it is used, through a pointer the compiler cannot see
	// through — kept live across the march those ~25 scalars push the kernel past the 80 SGPRs that 8 waves per SIMD allow, and the
	// compiler then spills scalars into VGPR lanes inside the window loop (RayKernelArgs is the first argument: offset 0).
	auto dense_args = []() { ConstArgs q = (ConstArgs) __builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(q)); return q; };
	struct KernelArguments { RayKernelArgs a; const uint8_t *copy; const float *tf_g; uint32_t *out; };      // the kernel's parameter list as it lies in that segment
	typedef const KernelArguments __attribute__((address_space(4))) *ConstKernelArguments;
	// explicit fetch of the element pair (march index i, i + 1) of a texel-space position, clamp addressing: any position is in bounds
	auto pair_address = [&](const uint8_t *copy_p, float mx, float my, float mz, uint32_t blocks_u, uint32_t windows, float xb, float yb, float zb) {
		const int ix = (int) __builtin_amdgcn_fmed3f(xb, 0.0f, mx), iy = (int) __builtin_amdgcn_fmed3f(yb, 0.0f, my), iz = (int) __builtin_amdgcn_fmed3f(zb, 0.0f, mz);
		const uint32_t iu = (uint32_t) (U == 0 ? ix : iy), iv = (uint32_t) (V == 1 ? iy : iz), im = (uint32_t) (M == 0 ? ix : (M == 1 ? iy : iz));
		const uint32_t wq = __umulhi(im, 0xAAAAAAABu) >> 1, sub = im - wq * 3u;
		const uint32_t block = ((iv >> kColEdgeLog2) * blocks_u + (iu >> kColEdgeLog2)) * windows + wq;
		const uint32_t in_block = (iv & kColEdgeMask) * kColRowBytes + (iu & kColEdgeMask) * kColWindowBytes + sub * 4u;      // < 256: summed in 32 bits
		const uint8_t *p = copy_p + ((uint64_t) block * kColBlockBytes + in_block);
		return VR_BC_POINTER(a, const uint8_t *, p, 8u);
	};
	auto coords = [&](ConstArgs q, float kk, float &xb, float &yb, float &zb) {          // fma(k, A, B) with A = direction * N/2 (col_sample: the same fp32 products, formed by the host)
		xb = VR_FMA(kk, q->col_sample.ax, B.x); yb = VR_FMA(kk, q->col_sample.ay, B.y); zb = VR_FMA(kk, q->col_sample.az, B.z);
	};
	auto fetch_at = [&](float kk, uint32_t &w0, uint32_t &w1) {                           // the element pair of the sample at kk, from each lane's true column
		ConstArgs q = dense_args();
		float xb, yb, zb;
		coords(q, kk, xb, yb, zb);
		const uint2 both = *(const uint2 *) pair_address(((ConstKernelArguments) q)->copy, q->col_sample.max_x, q->col_sample.max_y, q->col_sample.max_z, q->col_shade.nbu, q->col_shade.nw, xb, yb, zb);
		w0 = both.x; w1 = both.y;
	};
	// the same as a MANAGED gather (the compiler does not see it: a load it knows to be in flight across the window loop's back edge makes
	// it put s_waitcnt vmcnt(0) in front of every window gather, and the prefetch pipeline is gone): wait with managed_wait<0>() before use
	auto fetch_at_managed = [&](float kk, uint64_t &both) {
		ConstArgs q = dense_args();
		float xb, yb, zb;
		coords(q, kk, xb, yb, zb);
		managed_load64(both, (uint64_t) (uintptr_t) pair_address(((ConstKernelArguments) q)->copy, q->col_sample.max_x, q->col_sample.max_y, q->col_sample.max_z, q->col_shade.nbu, q->col_shade.nw, xb, yb, zb));
	};
	// one sample at `k` whose element pair is (w0, w1): the general kernel's body from the transparency test on
	auto sample = [&](uint32_t w0, uint32_t w1) {
		if ((__builtin_amdgcn_uicmp((w0 | w1) & a.skip_mask, a.skip_cmp, kIcmpNE) & live) != 0ull && VR_OPEN_LANES(acc.w, live) != 0ull) {
			ConstArgs q = dense_args();
			// everything this sample needs from the argument segment in ONE scalar load (held: the compiler would otherwise load each value
			// where it is first used, one scalar-cache round trip after the other on a path that is a dependent chain)
			RayKernelArgs::ColDenseSample ds;
			ds.ax = q->col_sample.ax; ds.ay = q->col_sample.ay; ds.az = q->col_sample.az; ds.tf_scale = q->col_sample.tf_scale;
			ds.max_x = q->col_sample.max_x; ds.max_y = q->col_sample.max_y; ds.max_z = q->col_sample.max_z; ds.tf_zero_below = q->col_sample.tf_zero_below;
			ds.light_kd = q->col_sample.light_kd; ds.ray_threshold = q->col_sample.ray_threshold;
			hold_scalars(ds.ax, ds.ay, ds.az, ds.tf_scale, ds.max_x, ds.max_y, ds.max_z, ds.tf_zero_below, ds.light_kd, ds.ray_threshold);
			live &= __builtin_amdgcn_fcmpf(k, ky, kFcmpOLE);                                              // the sample's own segment test
			const float xb = VR_FMA(k, ds.ax, B.x), yb = VR_FMA(k, ds.ay, B.y), zb = VR_FMA(k, ds.az, B.z);
			const float raw = col_resolve<M, kQ8>(w0, w1, ds.max_x, ds.max_y, ds.max_z, xb, yb, zb);     // GPURenderer4.cu:76
			const float tb = __builtin_amdgcn_fmed3f(VR_FMA(raw, ds.tf_scale, -0.5f), 0.0f, (float) (VR_TF_SIZE - 1));
			if ((__builtin_amdgcn_fcmpf(tb, ds.tf_zero_below, kFcmpOGE) & live) != 0ull) {
				f4 c;
				{
					const uint32_t i = (uint32_t) (int) tb;
					const float w = filter_weight<kQ8>(__builtin_amdgcn_fractf(tb));
					const f4 c0 = tf_l[i], dc = dtf_l[i];
					c.x = VR_FMA(w, dc.x, c0.x); c.y = VR_FMA(w, dc.y, c0.y); c.z = VR_FMA(w, dc.z, c0.z); c.w = VR_FMA(w, dc.w, c0.w);
				}
				const uint64_t shaded = ds.light_kd > 0.01f ? (__builtin_amdgcn_fcmpf(c.w, 0.05f, kFcmpOGT) & live) : 0ull;   // GPURenderer4.cu:78
				if (shaded != 0ull) {                                                                  // GPURenderer4.cu:41-51 shade_texture
					const f3 org = origin_again();                                                     // (its LDS read is in flight with the scalar load below)
					RayKernelArgs::ColDenseShade dh;                                                   // (one scalar load again)
					for (int i = 0; i < 3; i++) { dh.dir[i] = q->col_shade.dir[i]; dh.light[i] = q->col_shade.light[i]; dh.lh[i] = q->col_shade.lh[i]; }
					dh.kd_scaled = q->col_shade.kd_scaled; dh.nbu = q->col_shade.nbu; dh.nw = q->col_shade.nw;
					const uint8_t *const copy_p = ((ConstKernelArguments) q)->copy;
					hold_scalars(dh.dir[0], dh.dir[1], dh.dir[2], dh.kd_scaled, dh.light[0], dh.light[1], dh.light[2], dh.lh[0], dh.lh[1], dh.lh[2]);
					hold_scalars(dh.nbu, dh.nw, (uint64_t) (uintptr_t) copy_p);
					const f3 p3 = march_point<SAMPLING>(org, mk3(dh.dir[0], dh.dir[1], dh.dir[2]), k);
					const f3 d = mk3(dh.light[0] - p3.x, dh.light[1] - p3.y, dh.light[2] - p3.z);
					const float inv = rsqrt_nr(VR_FMA(d.z, d.z, VR_FMA(d.y, d.y, d.x * d.x)));
					const float sx = VR_FMA(d.x * inv, dh.lh[0], xb), sy = VR_FMA(d.y * inv, dh.lh[1], yb), sz = VR_FMA(d.z * inv, dh.lh[2], zb);
					uint32_t l0, l1;
#ifdef VR_COL_EXP_NO_SHADE_FETCH      // timing-only experiment: the shading sample costs no memory round trip
					l0 = w0 ^ (uint32_t) (int) sx; l1 = w1 ^ (uint32_t) (int) sy;
#else
					{ const uint2 both = *(const uint2 *) pair_address(copy_p, ds.max_x, ds.max_y, ds.max_z, dh.nbu, dh.nw, sx, sy, sz); l0 = both.x; l1 = both.y; }
#endif
					const float raw_l = col_resolve<M, kQ8>(l0, l1, ds.max_x, ds.max_y, ds.max_z, sx, sy, sz);
					const float diffuse = select_lanes(shaded, (raw_l - raw) * dh.kd_scaled);
					c.x += diffuse; c.y += diffuse; c.z += diffuse;
				}
				const float t = select_lanes(live, 1 - acc.w);
				acc.x = VR_FMA(c.x, t, acc.x); acc.y = VR_FMA(c.y, t, acc.y); acc.z = VR_FMA(c.z, t, acc.z); acc.w = VR_FMA(c.w, t, acc.w);
				live &= ~__builtin_amdgcn_fcmpf(acc.w, ds.ray_threshold, kFcmpOGT);                    // ERT (CPURenderer.cpp:35-36)
			}
		}
	};

	// -- can this wave take the column path?
	const int leader = __builtin_ctzll(alive_mask);
	const float kx_l = rlane(kx, leader), Bm_l = rlane(Bm, leader);
	bool ok = __builtin_amdgcn_ballot_w64(alive && (__float_as_uint(kx) != __float_as_uint(kx_l) || __float_as_uint(Bm) != __float_as_uint(Bm_l))) == 0ull;
	const float advance = __builtin_fabsf(Am) * step;                       // cells along m per sample: a window is never skipped, and holds at most ~200 samples
	ok = ok && __builtin_amdgcn_ballot_w64(!(advance >= (1.0f / 64.0f) && advance <= 1.0f)) == 0ull;
	// lateral cells at the two ends of the segment (the coordinate is monotone in k, so is its cell)
	auto cell = [&](float kk, float Ac, float Bc, float maxc) { return (int) __builtin_amdgcn_fmed3f(VR_FMA(kk, Ac, Bc), 0.0f, maxc); };
	int cu0 = cell(kx, Au, Bu, max_u), cv0 = cell(kx, Av, Bv, max_v);
	int cu1 = cell(ky, Au, Bu, max_u), cv1 = cell(ky, Av, Bv, max_v);
	{   // lanes without a segment ride along in the leader's column
		const int lu = __builtin_amdgcn_readlane(cu0, leader), lv = __builtin_amdgcn_readlane(cv0, leader);
		if (!alive) { cu0 = cu1 = lu; cv0 = cv1 = lv; ky = -1.0f; }
	}
	const uint64_t flips_u = __builtin_amdgcn_ballot_w64(cu0 != cu1), flips_v = __builtin_amdgcn_ballot_w64(cv0 != cv1);
	ok = ok && __builtin_amdgcn_ballot_w64((cu1 - cu0) * (cu1 - cu0) > 1 || (cv1 - cv0) * (cv1 - cv0) > 1) == 0ull;
	// offsets relative to the leader's column, biased by 2^30 so that they are unsigned 32-bit VGPR offsets of one scalar base
	const int64_t ref = (int64_t) (f_u(__builtin_amdgcn_readlane(cu0, leader)) + f_v(__builtin_amdgcn_readlane(cv0, leader)));
	const int64_t rel0 = (int64_t) (f_u(cu0) + f_v(cv0)) - ref;
	const int64_t du64 = (int64_t) f_u(cu1) - (int64_t) f_u(cu0), dv64 = (int64_t) f_v(cv1) - (int64_t) f_v(cv0);
	{
		const int64_t lim = 1ll << 28;
		ok = ok && __builtin_amdgcn_ballot_w64(rel0 <= -lim || rel0 >= lim || du64 <= -lim || du64 >= lim || dv64 <= -lim || dv64 >= lim) == 0ull;
	}

	const bool has_flips = (flips_u | flips_v) != 0ull;
	// per-lane march with explicit fetches (exact, unpipelined): the few waves that straddle two kx values, and forced testing
	auto per_lane_march = [&]() {
		while (live != 0ull) {
			uint32_t w0, w1;
			fetch_at(k, w0, w1);
			sample(w0, w1);
			k += step;
			live &= __builtin_amdgcn_fcmpf(k, ky, kFcmpOLE);
		}
	};
	if (ok) {
		const uint32_t voff0 = (uint32_t) (rel0 + (1ll << 30));
		uint64_t s_base;
		{
			const uint64_t b = (uint64_t) (uintptr_t) copy + (uint64_t) ref - (1ull << 30);
			s_base = ((uint64_t) rfl((uint32_t) (b >> 32)) << 32) | rfl((uint32_t) b);
		}
		const int dsign = (__float_as_uint(comp3<M>(dir)) >> 31) != 0u ? -1 : 1;     // march direction along m, from the kernel argument's bits: stays scalar
		// window index = floor(cell / 3), also for the cells below 0 a ray reaches after its exit (biased by a multiple of 3; a cell index
		// must never stick to a window: the loops below end when the samples have moved on)
		auto window_of = [](int lc) { return (int) (__umulhi((uint32_t) lc + 0x30000000u, 0xAAAAAAABu) >> 1) - 0x10000000; };
		// Column flips.  A lane's lateral cell changes at the smallest float t in (kx, ky] with cell(t) != cell(kx) (bisection over the
		// positive float bit patterns, once per ray).  What the march needs of t is only the WINDOW Wt the position k = t lies in: the
		// coordinate along m and its cell are monotone in k, so every sample in a window before Wt (in march order) has k < t — the lane
		// still reads its first column — and every sample in a window after Wt has k >= t — its second column; only window Wt itself
		// can hold samples of both kinds (a "careful" window: every sample fetches from each lane's true column, explicitly).  Windows are
		// compared through keys that grow by one per window in march order: key = dsign * window.
		constexpr int kNoEvent = 0x7fffffff;
		int key_u = kNoEvent, key_v = kNoEvent;
		auto bisect = [&](bool flipping, int c0, float Ac, float Bc, float maxc) {
			uint32_t lo = __float_as_uint(kx), hi = __float_as_uint(ky);
			if (!flipping) hi = lo;
			for (int it = 0; it < 34 && __builtin_amdgcn_ballot_w64(hi - lo > 1u) != 0ull; it++) {
				const uint32_t mid = lo + ((hi - lo) >> 1);
				const bool same = cell(__uint_as_float(mid), Ac, Bc, maxc) == c0;
				if (hi - lo > 1u) { if (same) lo = mid; else hi = mid; }
			}
			return flipping ? dsign * window_of((int) VR_FMA(__uint_as_float(hi), Am, Bm_l)) : kNoEvent;
		};
		if (FLIPS && flips_u != 0ull) key_u = bisect(cu0 != cu1, cu0, Au, Bu, max_u);
		if (FLIPS && flips_v != 0ull) key_v = bisect(cv0 != cv1, cv0, Av, Bv, max_v);
		// the wave's distinct event keys in march order: lane i of `events` holds the i-th (kNoEvent beyond the last), so that the march
		// compares the window it issues / consumes with ONE scalar and touches the lanes only where something happens
		int events = kNoEvent;
		bool events_ok = true;
		if (FLIPS && has_flips) {
			const uint32_t lane_i = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
			int last = (int) 0x80000000, count = 0;
			#pragma nounroll
			for (; count < 64; count++) {
				int cand = key_u > last ? key_u : kNoEvent;
				if (key_v > last && key_v < cand) cand = key_v;
				#pragma unroll
				for (int d = 32; d >= 1; d >>= 1) { const int other = __shfl_xor(cand, d, 64); cand = other < cand ? other : cand; }
				cand = (int) rfl((uint32_t) cand);
				if (cand == kNoEvent) break;
				events = lane_i == (uint32_t) count ? cand : events;
				last = cand;
			}
			events_ok = count < 64;                                         // (more distinct events than lanes: cannot happen with <= 64 rows and columns per wave; marched per lane if it does)
		}
		// What a lane needs to know at its events, packed into ONE register for the march (the flips instantiation has to stay at 64
		// VGPRs too): bits 0-11 / 12-23 its event keys + 1024 (0xfff: none; |key| <= 683 + 1 for edges up to 2048), bit 24 / 25 set =
		// the step to its second column crosses a block edge, bit 28 / 29 set = the column index goes up.  The byte delta to the second
		// column: +-16 (+-64 along v) inside a block, +-(block stride - 3 * 16) (- 3 * 64) across a block edge.
		uint32_t flipinfo = 0x00ffffffu;
		if (FLIPS && has_flips) {
			const uint32_t pu = key_u == kNoEvent ? 0xfffu : (uint32_t) (key_u + 1024) & 0xfffu, pv = key_v == kNoEvent ? 0xfffu : (uint32_t) (key_v + 1024) & 0xfffu;
			const bool up_u = cu1 > cu0, up_v = cv1 > cv0;
			const bool cross_u = ((uint32_t) cu0 & kColEdgeMask) == (up_u ? kColEdgeMask : 0u), cross_v = ((uint32_t) cv0 & kColEdgeMask) == (up_v ? kColEdgeMask : 0u);
			flipinfo = pu | (pv << 12) | (cross_u ? 1u << 24 : 0u) | (cross_v ? 1u << 25 : 0u) | (up_u ? 1u << 28 : 0u) | (up_v ? 1u << 29 : 0u);
			events_ok = events_ok && __builtin_amdgcn_ballot_w64((key_u != kNoEvent && (key_u < -1023 || key_u > 1023)) || (key_v != kNoEvent && (key_v < -1023 || key_v > 1023))) == 0ull;
		}

		// -- the wave-uniform sample sequence, 64 samples at a time: lane j of `kvec` holds k of sample n + j, `lcvec` its (logical) cell
		// along m, `wvec` the window that cell lies in.  The reference forms k by repeated fp32 additions k += step.  Inside one binade
		// [2^e, 2^(e+1)) every k is a multiple of u = 2^(e-23), and fl(k + step) = k + round_u(step) whatever k is — unless step's part
		// below u is exactly u / 2 (a tie, broken by k's parity) — so the sequence is an EXACT arithmetic progression there: k_(n+j) =
		// fma(j, delta, k_n) with delta = fl(k_n + step) - k_n, no rounding (every term is a multiple of u inside the binade).  A batch that
		// would cross a binade, a tie, or k <= 2^-102 is formed by the 64 sequential additions instead (lane j keeps the j-th sum): a handful
		// of batches per ray.
		float kvec = 0.0f, knext = kx_l;
		int wvec = 0;
		auto refill = [&]() {
			const uint32_t lane_i = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));     // 0 .. 63
			const float kbase = knext;
			const float k1 = kbase + step, delta = k1 - kbase, low = step - delta;           // delta = round_u(step); low = what the rounding dropped (both exact)
			const uint32_t e = __float_as_uint(kbase) >> 23;                                  // kbase >= 0: the biased exponent
			const float half_ulp = __uint_as_float((e > 24u ? e - 24u : 1u) << 23);
			const float kend = VR_FMA(64.0f, delta, kbase);
			const bool fast = rfl((e > 24u && (__float_as_uint(kend) >> 23) == e && __builtin_fabsf(low) != half_ulp && delta > 0.0f) ? 1u : 0u) != 0u;
			if (fast) {
				kvec = VR_FMA((float) lane_i, delta, kbase);
				knext = uni(kend);
			} else {
				float kc = kbase;
				#pragma nounroll
				for (uint32_t j = 0; j < 64u; j++) { kvec = lane_i == j ? kc : kvec; kc = kc + step; }
				knext = uni(kc);
			}
			wvec = window_of((int) VR_FMA(kvec, Am, Bm_l));                                  // (cell by truncation; may leave 0 .. Nm-1 past the exit)
		};
		refill();
		int pos = 0;                                                        // next sample of the batch
		int cur = __builtin_amdgcn_readlane(wvec, 0);                       // the window being consumed (0 .. nw-1 for the first sample of a live ray)
		cur = cur < 0 ? 0 : (cur > (int) nw - 1 ? (int) nw - 1 : cur);
		// Hang / bounds guard: a ray cannot need more windows than lie ahead of its first one in march direction; past the last of them
		// every k exceeds every ky and `live` empties at the next rotation, i.e. at most 2 * kColSlots windows later.  Those — and the
		// kColDepth prefetched beyond — are read without a clamp: neighbouring blocks' windows, or the kColPadBytes of zeroes at both
		// ends of the copy (64 windows).
		static_assert(2 * kColSlots + kColDepth + 4 <= 64, "kColPadBytes");
		int guard = (dsign > 0 ? (int) nw - cur : cur + 1) + 2 * kColSlots;
		int woff = cur * (int) kColBlockBytes;                              // byte offset, inside a block's run of windows, of the window being ISSUED (may leave 0 .. nw * 256: see the guard)
		auto march = [&](auto flips_tag) {
			constexpr bool kFlips = decltype(flips_tag)::value;
			// flips: `vo` follows the issue frontier — when it passes an event window, the lanes that flip there move on to their second column
			uint32_t vo = voff0;
			int issue_key = dsign * cur, issue_at = 0, issue_event = kFlips ? __builtin_amdgcn_readlane(events, 0) : kNoEvent;
			int cons_at = 0, cons_event = issue_event;
			// byte delta to their second column for the lanes that flip in the event window `event` (0 for the others): integer arithmetic
			// only (no lane masks: they would cost scalar registers in every window step)
			auto event_delta = [&](int event) {
				ConstArgs q = dense_args();
				const uint32_t qdim_u = U == 0 ? q->dim_x : q->dim_y, qdim_m = M == 0 ? q->dim_x : (M == 1 ? q->dim_y : q->dim_z);
				const uint32_t stride_u32 = col_windows(qdim_m) * kColBlockBytes, stride_v32 = col_blocks(qdim_u) * stride_u32;      // < 2^28 (checked through du64 / dv64 for every lane that flips)
				const uint32_t want = (uint32_t) (event + 1024) & 0xfffu;
				const uint32_t hit_u = (uint32_t) ((int) (((flipinfo ^ want) & 0xfffu) - 1u) >> 31), hit_v = (uint32_t) ((int) ((((flipinfo >> 12) ^ want) & 0xfffu) - 1u) >> 31);   // all ones where the key matches
				const uint32_t mag_u = kColWindowBytes + ((flipinfo >> 24) & 1u) * (stride_u32 - kColRowBytes), mag_v = kColRowBytes + ((flipinfo >> 25) & 1u) * (stride_v32 - kColBlockBytes);
				const uint32_t neg_u = ((flipinfo >> 28) & 1u) - 1u, neg_v = ((flipinfo >> 29) & 1u) - 1u;             // all ones = the column index goes down
				return (((mag_u ^ neg_u) - neg_u) & hit_u) + (((mag_v ^ neg_v) - neg_v) & hit_v);                      // two's complement deltas: the 32-bit sums wrap back into range
			};
			auto issue = [&](u32x4 &dst) {
				if (kFlips) {
					while (issue_key > issue_event) {                        // (rare: a handful of events per ray)
						vo += event_delta(issue_event);
						issue_at++;
						issue_event = __builtin_amdgcn_readlane(events, issue_at & 63);
					}
					issue_key++;
				}
				const uint32_t lane_offset = vo + (uint32_t) woff;            // 2^30 - 2^28 - padding < lane_offset < 2^30 + 2^29: an unsigned 32-bit offset of the one scalar base
#if defined(VR_BOUNDS_CHECK)           // debug build: the address is held against the copy (incl. its padding) and redirected if it leaves it
				managed_load128(dst, VR_BC_ADDRESS(a, s_base + lane_offset, 16u));
#elif defined(VR_COL_EXP_NO_LOAD)      // timing-only experiment: no gathers
				dst = (u32x4) (lane_offset & 0u);
#else
				managed_load128_s(dst, lane_offset, s_base);
#endif
				woff += dsign * (int) kColBlockBytes;
			};
			u32x4 slot[kColSlots];
			slot[kColSlots - 1] = (u32x4) (0u);
			static_for<0, kColDepth>([&](auto j) { issue(slot[j.value]); });
			auto window_step = [&](auto jc) {
				constexpr int c = decltype(jc)::value, n = (c + kColDepth) % kColSlots;
				issue(slot[n]);
				__builtin_amdgcn_sched_barrier(0);
				pin(slot[c]); managed_wait<kColDepth>(); pin(slot[c]);
				const u32x4 o = slot[c];
				if (c == 0) live &= __builtin_amdgcn_fcmpf(rlane(kvec, pos), ky, kFcmpOLE);      // lazy exit test: once per rotation of the slots (and by every sample that composites)
				bool careful = false;                                       // an event window: some lane changes its column somewhere inside
				if (kFlips) {
					const int key = issue_key - (kColDepth + 1);               // = dsign * cur: the issue frontier is kColDepth windows ahead and has just moved on
					while (key > cons_event) { cons_at++; cons_event = __builtin_amdgcn_readlane(events, cons_at & 63); }
					careful = key == cons_event;
#ifdef VR_COL_EXP_NO_CAREFUL          // timing-only experiment: event windows are treated like any other (wrong pixels where a ray changes its column)
					careful = false;
#endif
				}
				const uint32_t all4 = (o.x | o.y | o.z | o.w) & a.skip_mask;
				bool dense = careful;
#ifndef VR_COL_EXP_NO_DENSE           // (timing-only experiment: every window counts as transparent)
				if ((__builtin_amdgcn_uicmp(all4, a.skip_cmp, kIcmpNE) & live) != 0ull) dense = dense || VR_OPEN_LANES(acc.w, live) != 0ull;
#endif
				if (!dense) {
					// a transparent window: its samples — consecutive lanes of the batch, from pos — just pass.  A window holds at most 3 * 64 + 1
					// samples (a sample advances >= 1/64 cell, checked above), i.e. it ends within four batches: the bound is a hang guard
					for (int batches = 0; batches < 5; batches++) {
						pos += __builtin_popcountll(__builtin_amdgcn_ballot_w64(wvec == cur));
						if (pos < 64) break;
						refill(); pos = 0;                                 // the window may go on in the next batch
					}
					pos = pos < 64 ? pos : 63;
				} else {
					const uint32_t first = (uint32_t) (cur * 3);
					for (int batches = 0; batches < 5; batches++) {
						const int cnt = __builtin_popcountll(__builtin_amdgcn_ballot_w64(wvec == cur));
						if (kFlips && careful) {
							// every sample from each lane's true column, explicitly (one exposed memory round trip per sample: a few windows per ray)
							for (int i = pos; i < pos + cnt; i++) {
								k = rlane(kvec, i);
								uint64_t both;
								fetch_at_managed(k, both);
								pin(both); managed_wait<0>(); pin(both);               // (also the window gathers in flight: a careful window is rare)
								sample((uint32_t) both, (uint32_t) (both >> 32));
							}
						} else {
							for (int i = pos; i < pos + cnt; i++) {
								k = rlane(kvec, i);
								const uint32_t sub = rfl((uint32_t) (int) VR_FMA(k, Am, Bm_l)) - first;        // the sample's cell inside the window (the batch keeps only its window)
								sample(sub == 0u ? o.x : (sub == 1u ? o.y : o.z), sub == 0u ? o.y : (sub == 1u ? o.z : o.w));
							}
						}
						pos += cnt;
						if (pos < 64) break;
						refill(); pos = 0;
					}
					pos = pos < 64 ? pos : 63;
				}
				cur += dsign;
			};
			while (live != 0ull && guard > 0) {
				static_for<0, kColSlots>(window_step);
				guard -= kColSlots;
			}
			static_for<0, kColSlots>([&](auto j) { pin(slot[j.value]); });
			managed_wait<0>();
			static_for<0, kColSlots>([&](auto j) { pin(slot[j.value]); });
		};
		if (!events_ok || (has_flips && !FLIPS)) per_lane_march();
		else march(std::integral_constant<bool, FLIPS>());
	} else per_lane_march();
	// what the final store needs is read off two vector registers the march keeps anyway (lane masks held across it would cost scalar
	// registers): a lane has a segment iff its ky is positive (lanes without one were given -1), and is inside the buffer iff it has an index
	uint32_t ky_bits = __float_as_uint(ky);
	pin(ky_bits, out_index);
	uint32_t rgba = 0;
	if (__uint_as_float(ky_bits) > 0.0f) rgba = map_float_int(acc.x, 256) | (map_float_int(acc.y, 256) << 8) | (map_float_int(acc.z, 256) << 16) | (map_float_int(acc.w, 256) << 24);
	if (out_index != 0xffffffffu) ((ConstKernelArguments) dense_args())->out[out_index] = rgba;
}

// ---- the column march for NEAREST sampling (round 4) — the mode that is bit-exact against the reference's own CPURenderer ------------
//
// The same march as colmarch_kernel (wave-uniform k in 64-sample batches, one managed 16-byte gather per lane and window, event windows
// for lanes that change their column), with Model::sample_data's arithmetic (ModelBase.h:17-23, CPURenderer.cpp:17,24,38): position =
// origin + direction * k (two roundings), cell = map_float_int((position + 1) / 2, dim).  A sample needs ONE voxel, so a window is 16
// consecutive voxels of the lane's column (vr_device.h kColVoxCells): one gather and one transparency test per SIXTEEN samples, and the
// copy is 1 byte per voxel.  The cell along m is tracked UNclamped (a clamped index would stick to the last window and the window loops
// rely on the samples moving on), except for the one value Nm that map_float_int folds onto Nm - 1 for positions on the far face.
__device__ __forceinline__ void managed_load8_at(uint32_t &dst, uint64_t address) {       // zero-extended byte by 64-bit address
	asm volatile("global_load_ubyte %0, %1, off" : "=&v"(dst) : "v"(address));
}

template <int M, bool FLIPS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(VR_COL_WAVES, 8)))
void colmarch_nearest_kernel(const RayKernelArgs a, const uint8_t *__restrict__ copy, const float *__restrict__ tf_g, uint32_t *__restrict__ out) {
	constexpr int U = M == 0 ? 1 : 0, V = M == 2 ? 1 : 2;
	constexpr int kCells = (int) kColVoxCells;
	typedef const RayKernelArgs __attribute__((address_space(4))) *ConstArgs;
	__shared__ f4 tf_l[VR_TF_SIZE];
	__shared__ float unit_l[256];                                       // unit[s] = (float) s / 255.0f, the quotient Raycaster::shade forms twice per shaded sample
	__shared__ f4 org_l[512];                                           // every thread's ray origin
	{
		const uint32_t t = threadIdx.x;
		if (t < VR_TF_SIZE) tf_l[t] = ((const f4 *) tf_g)[t];
		if (t < 256u) unit_l[t] = (float) t / 255.0f;
	}
	__syncthreads();
	uint32_t tile_x, tile_y;
	tile_to_xy<VR_COL_XCD_MODE>(a.tiles_x, a.tiles_y, blockIdx.x, blockIdx.x, tile_x, tile_y);
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, qd = lane >> 4;
	uint32_t gu = lane & 3u, gv = (lane >> 2) & 3u;
	const uint32_t order = a.lane_map & 3u;
	if (order == kLaneBlocks) { gu = ((lane >> 1) & 2u) | (lane & 1u); gv = ((lane >> 2) & 2u) | ((lane >> 1) & 1u); }
	else if (order == kLaneColumns) { const uint32_t t = gu; gu = gv; gv = t; }
	const uint32_t wx = (qd & 1u) * 4u + gu, wy = (qd >> 1) * 4u + gv, ox = (wave & 3u) * 8u, oy = (wave >> 2) * 8u;
	const uint32_t lx = tile_x * 32u + ox + wx - a.phase_x, ly = tile_y * 16u + oy + wy - a.phase_y;
	const bool in_frame = lx < a.p.out_width && ly < a.p.out_rows;
	const uint32_t band = ly / a.p.band_rows;
	const uint32_t gy = (band * a.p.band_stride + a.p.band_first) * a.p.band_rows + (ly - band * a.p.band_rows);
	const uint32_t gx = a.p.x0 + lx;
	uint32_t out_index = in_frame ? ly * a.p.out_width + lx : 0xffffffffu;

	bool alive = in_frame && gx < a.p.view.width && gy < a.p.view.height;
	const f3 dir = ld3(a.p.view.direction);
	f3 origin;                                                          // kept in the thread's LDS slot for the samples that are shaded (see colmarch_kernel)
	{
		const float fx = (float) ((int) gx - (int) (a.p.view.width / 2u)), fy = (float) ((int) gy - (int) (a.p.view.height / 2u));
		const f3 o = mk3(a.p.view.origin[0] + a.p.view.right_plane[0] * fx, a.p.view.origin[1] + a.p.view.right_plane[1] * fx, a.p.view.origin[2] + a.p.view.right_plane[2] * fx);
		origin = mk3(o.x + a.p.view.up_plane[0] * fy, o.y + a.p.view.up_plane[1] * fy, o.z + a.p.view.up_plane[2] * fy);
	}
	uint32_t org_slot = threadIdx.x * (uint32_t) sizeof(f4);
	{ f4 o4; o4.x = origin.x; o4.y = origin.y; o4.z = origin.z; o4.w = 0.0f; org_l[threadIdx.x] = o4; }
	auto origin_again = [&]() { pin(org_slot); const f4 o4 = *(const f4 *) ((const char *) org_l + org_slot); return mk3(o4.x, o4.y, o4.z); };
	float kx = 0, ky = 0;
	alive = alive && intersect(origin, dir, kx, ky);
	const float step = a.p.ray_step;
	alive = alive && (ky + step > ky);
	ky = flmin(ky, kx + step * (float) kMaxRaySteps);
	const uint64_t alive_mask = __builtin_amdgcn_ballot_w64(alive);
	if (alive_mask == 0ull) { if (in_frame) out[out_index] = 0u; return; }
	if (!alive) ky = -1.0f;

	auto uni = [](float v) { return __uint_as_float(rfl(__float_as_uint(v))); };
	const float dm = comp3<M>(dir), du = comp3<U>(dir), dv = comp3<V>(dir);            // kernel arguments: scalar
	const float om = comp3<M>(origin), ou = comp3<U>(origin), ov = comp3<V>(origin);
	const uint32_t dim_u = U == 0 ? a.dim_x : a.dim_y, dim_v = V == 1 ? a.dim_y : a.dim_z, dim_m = M == 0 ? a.dim_x : (M == 1 ? a.dim_y : a.dim_z);
	const float half_m = M == 0 ? a.half_x : (M == 1 ? a.half_y : a.half_z);
	const uint32_t nbu = col_blocks(dim_u), nw = col_windows(dim_m, kColVoxCells);
	const uint64_t stride_u = (uint64_t) nw * kColBlockBytes, stride_v = (uint64_t) nbu * stride_u;
	auto f_u = [&](int c) { return (uint64_t) ((uint32_t) c >> kColEdgeLog2) * stride_u + ((uint32_t) c & kColEdgeMask) * kColWindowBytes; };
	auto f_v = [&](int c) { return (uint64_t) ((uint32_t) c >> kColEdgeLog2) * stride_v + ((uint32_t) c & kColEdgeMask) * kColRowBytes; };

	f4 acc; acc.x = acc.y = acc.z = acc.w = 0.0f;
	uint64_t live = alive_mask;
	float k = kx;
	// transfer_fn[sample / TF_RATIO] (CPURenderer.cpp:31) is (0,0,0,0) for sample <= opaque_above; the per-WINDOW test is the weaker
	// "every voxel below the largest power of two <= opaque_above + 1" (a mask on the packed bytes; windows that fail it test per sample)
	const int opaque_above = ((int) a.tf_zero_below + 1) * VR_TF_RATIO - 1;
	uint32_t near_mask = 0u, near_cmp = 1u;                             // nothing may be skipped: 0 != 1 always
	if (opaque_above >= 0) { const uint32_t p2 = 1u << (31 - __builtin_clz((uint32_t) opaque_above + 1u)); near_mask = (0xffu & ~(p2 - 1u)) * 0x01010101u; near_cmp = 0u; }

	auto dense_args = []() { ConstArgs q = (ConstArgs) __builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(q)); return q; };
	struct KernelArguments { RayKernelArgs a; const uint8_t *copy; const float *tf_g; uint32_t *out; };
	typedef const KernelArguments __attribute__((address_space(4))) *ConstKernelArguments;
	// address of the voxel Model::sample_data reads for a position (every index clamped: any position is in bounds)
	auto voxel_address = [&](ConstArgs q, f3 pos) {
		const uint32_t ix = map_float_int((pos.x + 1) * 0.5f, q->dim_x), iy = map_float_int((pos.y + 1) * 0.5f, q->dim_y), iz = map_float_int((pos.z + 1) * 0.5f, q->dim_z);
		const uint32_t iu = U == 0 ? ix : iy, iv = V == 1 ? iy : iz, im = M == 0 ? ix : (M == 1 ? iy : iz);
		const uint32_t qdim_u = U == 0 ? q->dim_x : q->dim_y, qdim_m = M == 0 ? q->dim_x : (M == 1 ? q->dim_y : q->dim_z);
		const uint32_t block = ((iv >> kColEdgeLog2) * col_blocks(qdim_u) + (iu >> kColEdgeLog2)) * col_windows(qdim_m, kColVoxCells) + (im >> 4);
		const uint32_t in_block = (iv & kColEdgeMask) * kColRowBytes + (iu & kColEdgeMask) * kColWindowBytes + (im & 15u);      // < 256: summed in 32 bits
		const uint8_t *p = ((ConstKernelArguments) q)->copy + ((uint64_t) block * kColBlockBytes + in_block);
		return VR_BC_POINTER(a, const uint8_t *, p, 1u);
	};
	auto position = [&](ConstArgs q, float kk) {                         // CPURenderer.cpp:17,24,38: origin + direction * k, two roundings per axis
		const f3 o = origin_again();
		return mk3(o.x + q->p.view.direction[0] * kk, o.y + q->p.view.direction[1] * kk, o.z + q->p.view.direction[2] * kk);
	};
	// one sample at `k` whose voxel is s: the general kernel's NEAREST body from the transparency test on (CPURenderer.cpp:29-39)
	auto sample = [&](uint32_t s) {
		if ((__builtin_amdgcn_sicmp((int) s, opaque_above, kIcmpSGT) & live) != 0ull && VR_OPEN_LANES(acc.w, live) != 0ull) {
			ConstArgs q = dense_args();
			live &= __builtin_amdgcn_fcmpf(k, ky, kFcmpOLE);
			uint32_t idx = s / VR_TF_RATIO;
			asm volatile("" : "+v"(idx));
			f4 cur = tf_l[idx & (VR_TF_SIZE - 1u)];
			const float kd = q->col_sample.light_kd, threshold = q->col_sample.ray_threshold;      // (adjacent: one scalar load)
			hold_scalars(kd, threshold);
			const uint64_t shaded = kd > 0.01f ? (__builtin_amdgcn_fcmpf(cur.w, 0.05f, kFcmpOGT) & live) : 0ull;
			if (shaded != 0ull) {                                                             // RaycasterBase.h:87-98 shade
				// everything the shading needs from the argument segment in ONE scalar load (see colmarch_kernel), beside the LDS read of the origin
				const f3 o = origin_again();
				RayKernelArgs::ColDenseShade dh;
				for (int i = 0; i < 3; i++) { dh.dir[i] = q->col_shade.dir[i]; dh.light[i] = q->col_shade.light[i]; dh.dim[i] = q->col_shade.dim[i]; }
				dh.nbu = q->col_shade.nbu; dh.nw = q->col_shade.nw;
				const uint8_t *const copy_p = ((ConstKernelArguments) q)->copy;
				hold_scalars(dh.dir[0], dh.dir[1], dh.dir[2], dh.light[0], dh.light[1], dh.light[2], dh.dim[0], dh.dim[1], dh.dim[2], dh.nbu, dh.nw, (uint64_t) (uintptr_t) copy_p);
				const f3 pt = mk3(o.x + dh.dir[0] * k, o.y + dh.dir[1] * k, o.z + dh.dir[2] * k);        // position(q, k)
				const f3 d = mk3(dh.light[0] - pt.x, dh.light[1] - pt.y, dh.light[2] - pt.z);
				const float inv = 1.0f / __builtin_sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
				const f3 l = mk3(d.x * inv, d.y * inv, d.z * inv);
				const f3 ps = mk3(pt.x + l.x * 0.01f, pt.y + l.y * 0.01f, pt.z + l.z * 0.01f);
				uint32_t s_l;
				{   // voxel_address(q, ps) with the constants held above
					const uint32_t ix = map_float_int((ps.x + 1) * 0.5f, dh.dim[0]), iy = map_float_int((ps.y + 1) * 0.5f, dh.dim[1]), iz = map_float_int((ps.z + 1) * 0.5f, dh.dim[2]);
					const uint32_t iu = U == 0 ? ix : iy, iv = V == 1 ? iy : iz, im = M == 0 ? ix : (M == 1 ? iy : iz);
					const uint32_t block = ((iv >> kColEdgeLog2) * dh.nbu + (iu >> kColEdgeLog2)) * dh.nw + (im >> 4);
					const uint32_t in_block = (iv & kColEdgeMask) * kColRowBytes + (iu & kColEdgeMask) * kColWindowBytes + (im & 15u);
					const uint8_t *p = copy_p + ((uint64_t) block * kColBlockBytes + in_block);
					s_l = *VR_BC_POINTER(a, const uint8_t *, p, 1u);
				}
				const float sl = unit_l[s_l], sc = unit_l[s & 255u];                          // RaycasterBase.h:93-96
				const float diffuse = select_lanes(shaded, (sl - sc) * kd);
				cur.x += diffuse; cur.y += diffuse; cur.z += diffuse;
			}
			const float t = select_lanes(live, 1 - acc.w);                                    // CPURenderer.cpp:34
			acc.x = acc.x + cur.x * t; acc.y = acc.y + cur.y * t;
			acc.z = acc.z + cur.z * t; acc.w = acc.w + cur.w * t;
			live &= ~__builtin_amdgcn_fcmpf(acc.w, threshold, kFcmpOGT);                      // CPURenderer.cpp:35-36
		}
	};

	// -- can this wave take the column path?  (all live lanes share kx and the origin's component along m: one k sequence, one cell along m)
	const int leader = __builtin_ctzll(alive_mask);
	const float kx_l = rlane(kx, leader), om_l = rlane(om, leader);
	bool ok = __builtin_amdgcn_ballot_w64(alive && (__float_as_uint(kx) != __float_as_uint(kx_l) || __float_as_uint(om) != __float_as_uint(om_l))) == 0ull;
	const float advance = __builtin_fabsf(dm * half_m) * step;
	ok = ok && __builtin_amdgcn_ballot_w64(!(advance >= (1.0f / 64.0f) && advance <= 1.0f)) == 0ull;
	// the cell along m of the wave's sample at kk: map_float_int's product by truncation, NOT clamped, but for the value Nm (positions on
	// the far face, folded onto Nm - 1 like map_float_int does)
	auto cell_m = [&](float kk) { const int c = (int) (((om_l + dm * kk) + 1.0f) * half_m); return c == (int) dim_m ? (int) dim_m - 1 : c; };
	auto cell_lat = [&](float kk, float oc, float dc, uint32_t n) { return (int) map_float_int(((oc + dc * kk) + 1) * 0.5f, n); };
	int cu0 = cell_lat(kx, ou, du, dim_u), cv0 = cell_lat(kx, ov, dv, dim_v);
	int cu1 = cell_lat(ky, ou, du, dim_u), cv1 = cell_lat(ky, ov, dv, dim_v);
	{
		const int lu = __builtin_amdgcn_readlane(cu0, leader), lv = __builtin_amdgcn_readlane(cv0, leader);
		if (!alive) { cu0 = cu1 = lu; cv0 = cv1 = lv; }
	}
	const uint64_t flips_u = __builtin_amdgcn_ballot_w64(cu0 != cu1), flips_v = __builtin_amdgcn_ballot_w64(cv0 != cv1);
	ok = ok && __builtin_amdgcn_ballot_w64((cu1 - cu0) * (cu1 - cu0) > 1 || (cv1 - cv0) * (cv1 - cv0) > 1) == 0ull;
	const int64_t ref = (int64_t) (f_u(__builtin_amdgcn_readlane(cu0, leader)) + f_v(__builtin_amdgcn_readlane(cv0, leader)));
	const int64_t rel0 = (int64_t) (f_u(cu0) + f_v(cv0)) - ref;
	const int64_t du64 = (int64_t) f_u(cu1) - (int64_t) f_u(cu0), dv64 = (int64_t) f_v(cv1) - (int64_t) f_v(cv0);
	{
		const int64_t lim = 1ll << 28;
		ok = ok && __builtin_amdgcn_ballot_w64(rel0 <= -lim || rel0 >= lim || du64 <= -lim || du64 >= lim || dv64 <= -lim || dv64 >= lim) == 0ull;
	}
	const bool has_flips = (flips_u | flips_v) != 0ull;
	auto per_lane_march = [&]() {                                        // exact, unpipelined: waves that straddle two kx values, forced testing
		while (live != 0ull) {
			ConstArgs q = dense_args();
			const uint32_t s = *voxel_address(q, position(q, k));
			sample(s);
			k += step;
			live &= __builtin_amdgcn_fcmpf(k, ky, kFcmpOLE);
		}
	};
	if (ok) {
		const uint32_t voff0 = (uint32_t) (rel0 + (1ll << 30));
		uint64_t s_base;
		{
			const uint64_t b = (uint64_t) (uintptr_t) copy + (uint64_t) ref - (1ull << 30);
			s_base = ((uint64_t) rfl((uint32_t) (b >> 32)) << 32) | rfl((uint32_t) b);
		}
		const int dsign = (__float_as_uint(dm) >> 31) != 0u ? -1 : 1;
		auto window_of = [](int lc) { return lc >> 4; };                  // floor(cell / 16), also below 0
		constexpr int kNoEvent = 0x7fffffff;
		int key_u = kNoEvent, key_v = kNoEvent;
		auto bisect = [&](bool flipping, int c0, float oc, float dc, uint32_t n) {
			uint32_t lo = __float_as_uint(kx), hi = __float_as_uint(ky);
			if (!flipping) hi = lo;
			for (int it = 0; it < 34 && __builtin_amdgcn_ballot_w64(hi - lo > 1u) != 0ull; it++) {
				const uint32_t mid = lo + ((hi - lo) >> 1);
				const bool same = cell_lat(__uint_as_float(mid), oc, dc, n) == c0;
				if (hi - lo > 1u) { if (same) lo = mid; else hi = mid; }
			}
			return flipping ? dsign * window_of(cell_m(__uint_as_float(hi))) : kNoEvent;
		};
		if (FLIPS && flips_u != 0ull) key_u = bisect(cu0 != cu1, cu0, ou, du, dim_u);
		if (FLIPS && flips_v != 0ull) key_v = bisect(cv0 != cv1, cv0, ov, dv, dim_v);
		int events = kNoEvent;
		bool events_ok = true;
		if (FLIPS && has_flips) {
			const uint32_t lane_i = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
			int last = (int) 0x80000000, count = 0;
			#pragma nounroll
			for (; count < 64; count++) {
				int cand = key_u > last ? key_u : kNoEvent;
				if (key_v > last && key_v < cand) cand = key_v;
				#pragma unroll
				for (int d = 32; d >= 1; d >>= 1) { const int other = __shfl_xor(cand, d, 64); cand = other < cand ? other : cand; }
				cand = (int) rfl((uint32_t) cand);
				if (cand == kNoEvent) break;
				events = lane_i == (uint32_t) count ? cand : events;
				last = cand;
			}
			events_ok = count < 64;
		}
		uint32_t flipinfo = 0x00ffffffu;                                  // colmarch_kernel's packing: keys + 1024, block-edge bits, directions
		if (FLIPS && has_flips) {
			const uint32_t pu = key_u == kNoEvent ? 0xfffu : (uint32_t) (key_u + 1024) & 0xfffu, pv = key_v == kNoEvent ? 0xfffu : (uint32_t) (key_v + 1024) & 0xfffu;
			const bool up_u = cu1 > cu0, up_v = cv1 > cv0;
			const bool cross_u = ((uint32_t) cu0 & kColEdgeMask) == (up_u ? kColEdgeMask : 0u), cross_v = ((uint32_t) cv0 & kColEdgeMask) == (up_v ? kColEdgeMask : 0u);
			flipinfo = pu | (pv << 12) | (cross_u ? 1u << 24 : 0u) | (cross_v ? 1u << 25 : 0u) | (up_u ? 1u << 28 : 0u) | (up_v ? 1u << 29 : 0u);
			events_ok = events_ok && __builtin_amdgcn_ballot_w64((key_u != kNoEvent && (key_u < -1023 || key_u > 1023)) || (key_v != kNoEvent && (key_v < -1023 || key_v > 1023))) == 0ull;
		}
		// the wave-uniform sample sequence, 64 samples at a time (see colmarch_kernel: exact arithmetic progression inside a binade)
		float kvec = 0.0f, knext = kx_l;
		int wvec = 0;
		auto refill = [&]() {
			const uint32_t lane_i = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
			const float kbase = knext;
			const float k1 = kbase + step, delta = k1 - kbase, low = step - delta;
			const uint32_t e = __float_as_uint(kbase) >> 23;
			const float half_ulp = __uint_as_float((e > 24u ? e - 24u : 1u) << 23);
			const float kend = VR_FMA(64.0f, delta, kbase);
			const bool fast = rfl((e > 24u && (__float_as_uint(kend) >> 23) == e && __builtin_fabsf(low) != half_ulp && delta > 0.0f) ? 1u : 0u) != 0u;
			if (fast) { kvec = VR_FMA((float) lane_i, delta, kbase); knext = uni(kend); }
			else {
				float kc = kbase;
				#pragma nounroll
				for (uint32_t j = 0; j < 64u; j++) { kvec = lane_i == j ? kc : kvec; kc = kc + step; }
				knext = uni(kc);
			}
			wvec = window_of(cell_m(kvec));
		};
		refill();
		int pos = 0;
		int cur = __builtin_amdgcn_readlane(wvec, 0);
		cur = cur < 0 ? 0 : (cur > (int) nw - 1 ? (int) nw - 1 : cur);
		int guard = (dsign > 0 ? (int) nw - cur : cur + 1) + 2 * kColSlots;          // hang / bounds guard, as in colmarch_kernel
		int woff = cur * (int) kColBlockBytes;
		auto march = [&](auto flips_tag) {
			constexpr bool kFlips = decltype(flips_tag)::value;
			uint32_t vo = voff0;
			int issue_key = dsign * cur, issue_at = 0, issue_event = kFlips ? __builtin_amdgcn_readlane(events, 0) : kNoEvent;
			int cons_at = 0, cons_event = issue_event;
			auto event_delta = [&](int event) {
				ConstArgs q = dense_args();
				const uint32_t qdim_u = U == 0 ? q->dim_x : q->dim_y, qdim_m = M == 0 ? q->dim_x : (M == 1 ? q->dim_y : q->dim_z);
				const uint32_t stride_u32 = col_windows(qdim_m, kColVoxCells) * kColBlockBytes, stride_v32 = col_blocks(qdim_u) * stride_u32;
				const uint32_t want = (uint32_t) (event + 1024) & 0xfffu;
				const uint32_t hit_u = (uint32_t) ((int) (((flipinfo ^ want) & 0xfffu) - 1u) >> 31), hit_v = (uint32_t) ((int) ((((flipinfo >> 12) ^ want) & 0xfffu) - 1u) >> 31);
				const uint32_t mag_u = kColWindowBytes + ((flipinfo >> 24) & 1u) * (stride_u32 - kColRowBytes), mag_v = kColRowBytes + ((flipinfo >> 25) & 1u) * (stride_v32 - kColBlockBytes);
				const uint32_t neg_u = ((flipinfo >> 28) & 1u) - 1u, neg_v = ((flipinfo >> 29) & 1u) - 1u;
				return (((mag_u ^ neg_u) - neg_u) & hit_u) + (((mag_v ^ neg_v) - neg_v) & hit_v);
			};
			auto issue = [&](u32x4 &dst) {
				if (kFlips) {
					while (issue_key > issue_event) { vo += event_delta(issue_event); issue_at++; issue_event = __builtin_amdgcn_readlane(events, issue_at & 63); }
					issue_key++;
				}
				const uint32_t lane_offset = vo + (uint32_t) woff;
#if defined(VR_BOUNDS_CHECK)
				managed_load128(dst, VR_BC_ADDRESS(a, s_base + lane_offset, 16u));
#else
				managed_load128_s(dst, lane_offset, s_base);
#endif
				woff += dsign * (int) kColBlockBytes;
			};
			u32x4 slot[kColSlots];
			slot[kColSlots - 1] = (u32x4) (0u);
			static_for<0, kColDepth>([&](auto j) { issue(slot[j.value]); });
			auto window_step = [&](auto jc) {
				constexpr int c = decltype(jc)::value, n = (c + kColDepth) % kColSlots;
				issue(slot[n]);
				__builtin_amdgcn_sched_barrier(0);
				pin(slot[c]); managed_wait<kColDepth>(); pin(slot[c]);
				const u32x4 o = slot[c];
				if (c == 0) live &= __builtin_amdgcn_fcmpf(rlane(kvec, pos), ky, kFcmpOLE);
				bool careful = false;
				if (kFlips) {
					const int key = issue_key - (kColDepth + 1);
					while (key > cons_event) { cons_at++; cons_event = __builtin_amdgcn_readlane(events, cons_at & 63); }
					careful = key == cons_event;
				}
				const uint32_t all16 = (o.x | o.y | o.z | o.w) & near_mask;
				bool dense = careful;
				if ((__builtin_amdgcn_uicmp(all16, near_cmp, kIcmpNE) & live) != 0ull) dense = dense || VR_OPEN_LANES(acc.w, live) != 0ull;
				// a window holds at most 16 * 64 + 1 samples (a sample advances >= 1/64 cell): it ends within kCells + 2 batches — a hang guard
				if (!dense) {
					for (int batches = 0; batches < kCells + 2; batches++) {
						pos += __builtin_popcountll(__builtin_amdgcn_ballot_w64(wvec == cur));
						if (pos < 64) break;
						refill(); pos = 0;
					}
					pos = pos < 64 ? pos : 63;
				} else {
					const int first = cur * kCells;
					for (int batches = 0; batches < kCells + 2; batches++) {
						const int cnt = __builtin_popcountll(__builtin_amdgcn_ballot_w64(wvec == cur));
						for (int i = pos; i < pos + cnt; i++) {
							k = rlane(kvec, i);
							if (kFlips && careful) {                            // the voxel of each lane's true column, explicitly
								ConstArgs q = dense_args();
								uint32_t s;
								managed_load8_at(s, (uint64_t) (uintptr_t) voxel_address(q, position(q, k)));
								pin(s); managed_wait<0>(); pin(s);
								sample(s);
							} else {
								const uint32_t sub = (uint32_t) ((int) rfl((uint32_t) cell_m(k)) - first) & 15u;      // the sample's voxel inside the window (uniform)
								const uint32_t word = (sub >> 2) == 0u ? o.x : ((sub >> 2) == 1u ? o.y : ((sub >> 2) == 2u ? o.z : o.w));
								sample((word >> ((sub & 3u) * 8u)) & 0xffu);
							}
						}
						pos += cnt;
						if (pos < 64) break;
						refill(); pos = 0;
					}
					pos = pos < 64 ? pos : 63;
				}
				cur += dsign;
			};
			while (live != 0ull && guard > 0) {
				static_for<0, kColSlots>(window_step);
				guard -= kColSlots;
			}
			static_for<0, kColSlots>([&](auto j) { pin(slot[j.value]); });
			managed_wait<0>();
			static_for<0, kColSlots>([&](auto j) { pin(slot[j.value]); });
		};
		if (!events_ok || (has_flips && !FLIPS)) per_lane_march();
		else march(std::integral_constant<bool, FLIPS>());
	} else per_lane_march();
	uint32_t ky_bits = __float_as_uint(ky);
	pin(ky_bits, out_index);
	uint32_t rgba = 0;
	if (__uint_as_float(ky_bits) > 0.0f) rgba = map_float_int(acc.x, 256) | (map_float_int(acc.y, 256) << 8) | (map_float_int(acc.z, 256) << 16) | (map_float_int(acc.w, 256) << 24);
	if (out_index != 0xffffffffu) ((ConstKernelArguments) dense_args())->out[out_index] = rgba;
}

// Which instantiation a frame runs: ONE selector, visited by the launcher and by the host's questions about the launch (does it read
// the linear array?  how many workgroup tiles?), so the answers cannot drift from what is launched.  `visit` is called with four
// std::integral_constant tags <SAMPLING, BPV, ADDR, LAYOUT> and a bool: true = the variant reads `linear`, false = the brick copy.
template <int SAMPLING, int BPV, class F>
static auto select_sampling(const RayKernelArgs &a, bool have_bricked, F &&visit) {
	constexpr bool nearest = SAMPLING == VR_SAMPLE_NEAREST;
	typedef std::integral_constant<int, SAMPLING> S;
	typedef std::integral_constant<int, BPV> V;
	const uint32_t max_dim = a.dim_x > a.dim_y ? (a.dim_x > a.dim_z ? a.dim_x : a.dim_z) : (a.dim_y > a.dim_z ? a.dim_y : a.dim_z);
	if constexpr (!nearest && BPV == 1) {
		if (have_bricked && a.layout == kLayoutRun)
			return visit(S(), V(), std::integral_constant<int, kAddr32>(), std::integral_constant<int, kLayoutRun>(), false);
		if (have_bricked && a.layout == kLayoutRunY)
			return visit(S(), V(), std::integral_constant<int, kAddr32>(), std::integral_constant<int, kLayoutRunY>(), false);
		if (have_bricked && a.layout == kLayoutRunDual)
			return visit(S(), V(), std::integral_constant<int, kAddr32>(), std::integral_constant<int, kLayoutRunDual>(), false);
	}
	if constexpr (nearest) {
		if (have_bricked && a.layout == kLayoutVoxel) {
			if (max_dim <= LutCfg<kAddr32>::max_dim && a.force_wide != 2)
				return visit(S(), V(), std::integral_constant<int, kAddr32>(), std::integral_constant<int, kLayoutVoxel>(), false);
			return visit(S(), V(), std::integral_constant<int, kAddrLut64>(), std::integral_constant<int, kLayoutVoxel>(), false);
		}
	}
	if constexpr (!nearest && BPV == 2) {
		if (have_bricked && a.layout == kLayoutOct) {
			const uint64_t bytes = bricked_elems(a.dim_x, a.dim_y, a.dim_z) * 8 * BPV;
			if (!a.force_wide && max_dim <= LutCfg<kAddr32>::max_dim && bytes <= (1ull << 32))
				return visit(S(), V(), std::integral_constant<int, kAddr32>(), std::integral_constant<int, kLayoutOct>(), false);
			return visit(S(), V(), std::integral_constant<int, kAddrLut64>(), std::integral_constant<int, kLayoutOct>(), false);
		}
	}
	if (have_bricked && a.layout == kLayoutBricked) {
		const uint64_t bytes = bricked_elems(a.dim_x, a.dim_y, a.dim_z) * 4 * BPV;
		if (!a.force_wide && max_dim <= LutCfg<kAddr32>::max_dim && bytes <= (1ull << 32))
			return visit(S(), V(), std::integral_constant<int, kAddr32>(), std::integral_constant<int, kLayoutBricked>(), false);
		if (a.force_wide != 1 && max_dim <= LutCfg<kAddrLut64>::max_dim)
			return visit(S(), V(), std::integral_constant<int, kAddrLut64>(), std::integral_constant<int, kLayoutBricked>(), false);
		if constexpr (!nearest)
			return visit(S(), V(), std::integral_constant<int, kAddrWide>(), std::integral_constant<int, kLayoutBricked>(), false);
	}
	// the reference's linear array; 32-bit byte offsets cover every volume the reference can express (ModelBase.h:12)
	const bool wide = a.force_wide || ((uint64_t) a.dim_x * a.dim_y * a.dim_z + volume_tail_slack(a.dim_x, a.dim_y)) * BPV >= (1ull << 32);
	return wide ? visit(S(), V(), std::integral_constant<int, kAddrWide>(), std::integral_constant<int, kLayoutLinear>(), true)
	            : visit(S(), V(), std::integral_constant<int, kAddr32>(), std::integral_constant<int, kLayoutLinear>(), true);
}

template <class F>
static auto select_variant(const RayKernelArgs &a, bool have_bricked, uint32_t bpv, F &&visit) {
	if (bpv == 1) {
		if (a.p.sampling == VR_SAMPLE_NEAREST) return select_sampling<VR_SAMPLE_NEAREST, 1>(a, have_bricked, visit);
		if (a.p.sampling == VR_SAMPLE_TRILINEAR_Q8) return select_sampling<VR_SAMPLE_TRILINEAR_Q8, 1>(a, have_bricked, visit);
		return select_sampling<VR_SAMPLE_TRILINEAR, 1>(a, have_bricked, visit);
	}
	if (a.p.sampling == VR_SAMPLE_NEAREST) return select_sampling<VR_SAMPLE_NEAREST, 2>(a, have_bricked, visit);
	if (a.p.sampling == VR_SAMPLE_TRILINEAR_Q8) return select_sampling<VR_SAMPLE_TRILINEAR_Q8, 2>(a, have_bricked, visit);
	return select_sampling<VR_SAMPLE_TRILINEAR, 2>(a, have_bricked, visit);
}

template <int ADDR, int LAYOUT> constexpr uint32_t variant_threads() { return LutCfg<(LAYOUT != kLayoutLinear ? ADDR : kAddrWide)>::threads; }

// what launch_raymarch will do with these arguments (launch_frame asks before it launches)
RaymarchPlan plan_raymarch(const RayKernelArgs &a, bool have_bricked, uint32_t bpv) {
	if (have_bricked && a.layout == kLayoutColumn) {                     // colmarch_kernel: 512 threads = 32x16 pixels
		RaymarchPlan plan;
		plan.reads_linear = false;
		plan.tiles_x = (a.p.out_width + a.phase_x + 31u) / 32u; plan.tiles_y = (a.p.out_rows + a.phase_y + 15u) / 16u;
		return plan;
	}
	return select_variant(a, have_bricked, bpv, [&](auto, auto, auto addr, auto layout, bool reads_linear) {
		constexpr uint32_t threads = variant_threads<decltype(addr)::value, decltype(layout)::value>();
		RaymarchPlan plan;
		plan.reads_linear = reads_linear;
		plan.tiles_x = (a.p.out_width + a.phase_x + 31u) / 32u;
		plan.tiles_y = (a.p.out_rows + a.phase_y + threads / 32u - 1u) / (threads / 32u);
		plan.tile_h = threads / 32u;
		return plan;
	});
}

hipError_t launch_raymarch(const RayKernelArgs &args, const void *linear, const void *bricked, uint32_t bpv, const float *tf,
                           const uint32_t *esl, void *out, TileSchedule sched, hipStream_t stream) {
	if (bricked != nullptr && args.layout == kLayoutColumn) {            // orthogonal view along args.col_axis, full march, TRILINEAR, 1-byte voxels (launch_frame)
		RayKernelArgs a = args;
		a.tiles_x = (a.p.out_width + a.phase_x + 31u) / 32u; a.tiles_y = (a.p.out_rows + a.phase_y + 15u) / 16u;
		const dim3 grid(a.tiles_x * a.tiles_y), block(512);
		const bool q8 = a.p.sampling == VR_SAMPLE_TRILINEAR_Q8;
		auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, block, 0, stream, a, (const uint8_t *) bricked, tf, (uint32_t *) out); };
		// lateral direction components exactly 0: no lane can change its column — the kernel without the flip logic
		const uint32_t m = a.col_axis;
		const bool flips = a.p.view.direction[m == 0u ? 1 : 0] != 0.0f || a.p.view.direction[m == 2u ? 1 : 2] != 0.0f;
		if (a.p.sampling == VR_SAMPLE_NEAREST) {                          // voxel windows (kCopyColVoxX ..)
			if (m == 0u) { if (flips) go(colmarch_nearest_kernel<0, true>); else go(colmarch_nearest_kernel<0, false>); }
			else if (m == 1u) { if (flips) go(colmarch_nearest_kernel<1, true>); else go(colmarch_nearest_kernel<1, false>); }
			else { if (flips) go(colmarch_nearest_kernel<2, true>); else go(colmarch_nearest_kernel<2, false>); }
			return hipGetLastError();
		}
		auto pick = [&](auto sampling, auto axis) {
			constexpr int S = decltype(sampling)::value, AX = decltype(axis)::value;
			if (flips) go(colmarch_kernel<S, AX, true>); else go(colmarch_kernel<S, AX, false>);
		};
		auto pick_axis = [&](auto sampling) {
			if (m == 0u) pick(sampling, std::integral_constant<int, 0>()); else if (m == 1u) pick(sampling, std::integral_constant<int, 1>()); else pick(sampling, std::integral_constant<int, 2>());
		};
		if (q8) pick_axis(std::integral_constant<int, VR_SAMPLE_TRILINEAR_Q8>()); else pick_axis(std::integral_constant<int, VR_SAMPLE_TRILINEAR>());
		return hipGetLastError();
	}
	return select_variant(args, bricked != nullptr, bpv, [&](auto sampling, auto voxel, auto addr, auto layout, bool reads_linear) {
		constexpr int SAMPLING = decltype(sampling)::value, BPV = decltype(voxel)::value, ADDR = decltype(addr)::value, LAYOUT = decltype(layout)::value;
		constexpr uint32_t threads = variant_threads<ADDR, LAYOUT>();
		RayKernelArgs a = args;
		a.tiles_x = (a.p.out_width + a.phase_x + 31u) / 32u;
		a.tiles_y = (a.p.out_rows + a.phase_y + threads / 32u - 1u) / (threads / 32u);
		// Run-brick frames are launched with 16 KiB of unused dynamic LDS: 3 instead of 4 workgroups per CU (24 waves).  Their waves
		// touch ~10 cache lines per step, 32 of them overflow the 256 lines of the 32 KiB L1 between two steps and the L2 catches only a
		// quarter of that reuse (measured: fabric requests -11 %, frame time -2 ... -5 % on those views; the VALU-bound quad-brick views
		// need all 32 waves and lose 10 % with the same padding).  Not with empty-space leaping: those rays are short, the frame time is
		// the tail of the few waves that probe a whole row of blocks, and fewer resident workgroups lengthen it (view 3: 1.53 -> 2.16 ms).
		// VR_RUN_LDS_PAD=0 builds without it (A/B).
#ifndef VR_RUN_LDS_PAD
#define VR_RUN_LDS_PAD 16384
#endif
#ifndef VR_PAD_LAYOUTS
#define VR_PAD_LAYOUTS ((1u << kLayoutRun) | (1u << kLayoutRunY) | (1u << kLayoutRunDual))
#endif
		const uint32_t dynamic_lds = ((VR_PAD_LAYOUTS >> LAYOUT) & 1u) && !a.p.esl ? VR_RUN_LDS_PAD : 0;
		hipLaunchKernelGGL((raymarch_kernel<SAMPLING, BPV, ADDR, LAYOUT>), dim3(a.tiles_x * a.tiles_y), dim3(threads), dynamic_lds, stream,
		                   a, reads_linear ? linear : bricked, tf, esl, (uint32_t *) out, sched.order, sched.cost);
		return hipGetLastError();
	});
}

// ---- measured-cost tile order -------------------------------------------------------------------------------------------------
//
// The hardware starts workgroups in id order as slots free up; rays of very different length (empty-space leaping, early
// termination, rays that probe along a block face) make some tiles 10-50x longer than others, and a long tile that starts late
// IS the tail of the frame.  A frame can record what every tile cost (tile_cost: the longest wave of the tile, in 64-cycle units);
// this kernel turns that into a launch order for the next frame with the same parameters: tiles binned by cost into kOrderBins
// bins, most expensive bin first, original tile order inside a bin (neighbouring tiles of similar cost stay neighbours: they share
// cache lines).  One workgroup, a stable counting sort through LDS; clears the costs for the next recording.  Placement only.
#ifndef VR_ORDER_BINS
#define VR_ORDER_BINS 16
#endif
constexpr uint32_t kOrderBins = VR_ORDER_BINS, kOrderThreads = 512;

__global__ __launch_bounds__(kOrderThreads)
void tile_order_kernel(uint32_t *__restrict__ cost, uint32_t *__restrict__ order, uint32_t ntiles) {
	__shared__ uint32_t wave_total[kOrderBins][kOrderThreads / 64u], bin_total[kOrderBins];
	__shared__ uint32_t vmax;
	// up to kOrderCached tiles (a 2048^2 frame has 8192) the costs are read ONCE, coalesced, into LDS and the three passes below run on that
	// copy (each thread owns a contiguous run of tiles, i.e. strided global reads otherwise); larger frames read them from memory
	constexpr uint32_t kOrderCached = 8192;
	__shared__ uint32_t cached[kOrderCached];
	const uint32_t t = threadIdx.x;
	const uint32_t chunk = (ntiles + kOrderThreads - 1) / kOrderThreads, lo = t * chunk < ntiles ? t * chunk : ntiles, hi = lo + chunk < ntiles ? lo + chunk : ntiles;
	const bool in_lds = ntiles <= kOrderCached;
	if (in_lds) for (uint32_t i = t; i < ntiles; i += kOrderThreads) { cached[i] = cost[i]; cost[i] = 0; }      // (cleared for the next recording on the way)
	if (t == 0) vmax = 0;
	__syncthreads();
	auto cost_of = [&](uint32_t i) { return in_lds ? cached[i] : cost[i]; };
	uint32_t m = 0;
	for (uint32_t i = lo; i < hi; i++) { const uint32_t c = cost_of(i); m = c > m ? c : m; }
	atomicMax(&vmax, m);
	__syncthreads();
	// (bins by a float product: the 64-bit division the first version used here, twice per tile, was most of the kernel's 25 us; any
	// monotone function does as long as both passes use the same one)
	const float scale = (float) kOrderBins / ((float) vmax + 1.0f);
	auto bin_of = [&](uint32_t c) { const uint32_t q = (uint32_t) ((float) c * scale); return kOrderBins - 1u - (q < kOrderBins ? q : kOrderBins - 1u); };    // 0 = most expensive
	uint32_t mine[kOrderBins];
	for (uint32_t b = 0; b < kOrderBins; b++) mine[b] = 0;
	for (uint32_t i = lo; i < hi; i++) mine[bin_of(cost_of(i))]++;
	// exclusive scan of every bin's per-thread counts over the 512 threads: inside a wave by shuffles, across the 8 waves through LDS
	// (the first version scanned each bin serially in one thread: 22 of the kernel's 26 us)
	const uint32_t lane = t & 63u, wave = t >> 6;
	uint32_t before[kOrderBins];
	for (uint32_t b = 0; b < kOrderBins; b++) {
		uint32_t v = mine[b];
		#pragma unroll
		for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t n = __shfl_up(v, d, 64); if (lane >= d) v += n; }
		before[b] = v - mine[b];
		if (lane == 63u) wave_total[b][wave] = v;
	}
	__syncthreads();
	if (t < kOrderBins) {
		uint32_t run = 0;
		for (uint32_t w = 0; w < kOrderThreads / 64u; w++) { const uint32_t c = wave_total[t][w]; wave_total[t][w] = run; run += c; }
		bin_total[t] = run;
	}
	__syncthreads();
	uint32_t base = 0, pos[kOrderBins];
	for (uint32_t b = 0; b < kOrderBins; b++) { pos[b] = base + wave_total[b][wave] + before[b]; base += bin_total[b]; }
	for (uint32_t i = lo; i < hi; i++) { order[pos[bin_of(cost_of(i))]++] = i; }
	if (!in_lds) {
		__syncthreads();
		for (uint32_t i = lo; i < hi; i++) cost[i] = 0;
	}
}

// ---- a launch order for the FIRST frame of a policy key (round 4) ------------------------------------------------------------------
// A frame that leaps has no recorded costs yet when its view is new (the reference's benchmark renders every view once): this kernel
// predicts them.  Eight lanes per workgroup tile walk one ray each (the centres of the tile's eighths) through the 32^3 ESL bit volume in
// half-block strides and count the strides that lie in non-empty blocks: cost = the largest such count in samples + one per stride
// probed (what the ESL loop pays in empty space).  Early ray termination is not modelled (an upper estimate for opaque regions).
// Feeds tile_order_kernel like a recording does.  Placement only: the image does not depend on it.
__global__ __launch_bounds__(256)
void tile_estimate_kernel(const RayKernelArgs a, uint32_t tile_h, const uint32_t *__restrict__ esl_g, uint32_t *__restrict__ cost, uint32_t ntiles) {
	const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
	const uint32_t t = gid >> 3, probe = gid & 7u;                      // eight lanes per tile, one ray each
	uint32_t tile_x = 0, tile_y = 0;
	if (t < ntiles) tile_to_xy(a.tiles_x, a.tiles_y, t, t, tile_x, tile_y);
	const f3 vo = ld3(a.p.view.origin), vd = ld3(a.p.view.direction), vr_ = ld3(a.p.view.right_plane), vu = ld3(a.p.view.up_plane);
	const float edge = flmin(flmin(a.p.esl_block_size[0], a.p.esl_block_size[1]), a.p.esl_block_size[2]);
	uint32_t best = 0;
	do {
		if (t >= ntiles) break;
		// the centres of the tile's 4 x 2 eighths (one 8x8-pixel wave each in the 32x16 tile)
		const uint32_t px = (probe & 3u) * 8u + 4u, py = (probe >> 2) * (tile_h / 2u) + tile_h / 4u;
		const uint32_t lx = tile_x * 32u + px - a.phase_x, ly = tile_y * tile_h + py - a.phase_y;
		if (lx >= a.p.out_width || ly >= a.p.out_rows) continue;
		const uint32_t band = ly / a.p.band_rows;
		const uint32_t gy = (band * a.p.band_stride + a.p.band_first) * a.p.band_rows + (ly - band * a.p.band_rows), gx = a.p.x0 + lx;
		if (gx >= a.p.view.width || gy >= a.p.view.height) continue;
		const float fx = (float) ((int) gx - (int) (a.p.view.width / 2u)), fy = (float) ((int) gy - (int) (a.p.view.height / 2u));
		f3 origin = vo, dir = vd;
		if (a.p.view.perspective) dir = mk3(vd.x + vr_.x * fx + vu.x * fy, vd.y + vr_.y * fx + vu.y * fy, vd.z + vr_.z * fx + vu.z * fy);
		else origin = mk3(vo.x + vr_.x * fx + vu.x * fy, vo.y + vr_.y * fx + vu.y * fy, vo.z + vr_.z * fx + vu.z * fy);
		float kx, ky;
		if (!intersect(origin, dir, kx, ky)) continue;
		const float longest = flmax(flmax(__builtin_fabsf(dir.x), __builtin_fabsf(dir.y)), __builtin_fabsf(dir.z));
		const float dk = 0.5f * edge / flmax(longest, 1e-6f);
		if (!(dk > 0.0f)) continue;
		const float strides_f = (ky - kx) / dk;
		const uint32_t strides = strides_f < 1.0f ? 1u : (strides_f > 400.0f ? 400u : (uint32_t) strides_f);
		uint32_t full = 0;
		for (uint32_t i = 0; i < strides; i++) {
			const float k = kx + ((float) i + 0.5f) * dk;
			const BlockIdx b = block_index(a, mk3(origin.x + dir.x * k, origin.y + dir.y * k, origin.z + dir.z * k));
			const uint32_t index = (b.z * VR_ESL_VOLUME_DIMS + b.y) & 0xffffu;
			if ((esl_g[index & (VR_ESL_VOLUME_SIZE - 1)] & (1u << (b.x & 31u))) == 0u) full++;
		}
		const float samples_per_stride = dk / flmax(a.p.ray_step, 1e-9f);
		const float est = (float) full * flmin(samples_per_stride, 4096.0f) + (float) strides;
		best = est > 4.0e9f ? 4000000000u : (uint32_t) est;
	} while (false);
	#pragma unroll
	for (uint32_t d = 1; d < 8u; d <<= 1) { const uint32_t o = __shfl_xor(best, d, 64); best = o > best ? o : best; }
	if (probe == 0u && t < ntiles) cost[t] = best;
}

hipError_t launch_tile_estimate(const RayKernelArgs &a, uint32_t tile_h, const uint32_t *esl, uint32_t *cost, uint32_t ntiles, hipStream_t stream) {
	hipLaunchKernelGGL(tile_estimate_kernel, dim3((ntiles * 8u + 255u) / 256u), dim3(256), 0, stream, a, tile_h, esl, cost, ntiles);
	return hipGetLastError();
}

hipError_t launch_tile_order(uint32_t *cost, uint32_t *order, uint32_t ntiles, hipStream_t stream) {
	hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(kOrderThreads), 0, stream, cost, order, ntiles);
	return hipGetLastError();
}

// ---- choice between the two run copies per block of tiles (kLayoutRunDual) ------------------------------------------------------
// choice[t] = t, with kTileAltBit set where the frame recorded on the copy along y was cheaper than the frame recorded on the copy
// along z.  Decided per group of 64 consecutive tile numbers — one 8x8-tile block of the numbering, 256x128 pixels — from the SUMS of
// the tile costs, and for the copy along y only if it wins by 5 % (VR_DUAL_KEEP_PERCENT, tuning aid): single tile costs are noisy (they depend on what else ran on the
// CU), and tiles that read different copies share no cache lines — neighbours must agree (measured: alternating tiles +20 % frame
// time, a per-tile choice +12 % on the perspective oblique pose, where the two copies are nearly level).  Both costs NULL:
// alternating tiles (testing aid: the two copies meet at tile boundaries all over the frame).  Placement only.
__global__ __launch_bounds__(64)
void tile_choice_kernel(const uint32_t *__restrict__ cost_z, const uint32_t *__restrict__ cost_y, uint32_t *__restrict__ choice, uint32_t ntiles, uint32_t keep_percent) {
	const uint32_t t = blockIdx.x * 64u + threadIdx.x;
	bool alt;
	if (cost_z != nullptr && cost_y != nullptr) {
		uint64_t z = t < ntiles ? cost_z[t] : 0u, y = t < ntiles ? cost_y[t] : 0u;
		for (int d = 32; d >= 1; d >>= 1) { z += __shfl_xor(z, d, 64); y += __shfl_xor(y, d, 64); }
		alt = y * 100u < z * keep_percent;
	} else alt = ((t ^ (t >> 3)) & 1u) != 0u;
	if (t < ntiles) choice[t] = t | (alt ? kTileAltBit : 0u);
}

hipError_t launch_tile_choice(const uint32_t *cost_z, const uint32_t *cost_y, uint32_t *choice, uint32_t ntiles, hipStream_t stream) {
	static const uint32_t keep_percent = [] { const char *e = getenv("VR_DUAL_KEEP_PERCENT"); return e ? (uint32_t) atoi(e) : 95u; }();
	hipLaunchKernelGGL(tile_choice_kernel, dim3((ntiles + 63u) / 64u), dim3(64), 0, stream, cost_z, cost_y, choice, ntiles, keep_percent);
	return hipGetLastError();
}

// ---- linear -> brick copies: LDS-tiled streaming transposes ---------------------------------------------------------------------
//
// Every copy (quad bricks per chunk plane, voxel bricks, oct bricks, run bricks along z / y) is built by ONE kernel shape: a workgroup
// owns a STRIP of kStripBricks bricks along x — contiguous in the copy, because bricks are stored x fastest — stages the voxel rows the
// strip's elements are made of (8 x-bricks + 1 voxel wide, 8 or 9 rows x 8 or 9 slices: the +1 neighbours, indices clamped at the upper
// faces, where the interpolation weight is exactly 0) with aligned 16-byte loads into LDS, builds the elements from LDS and writes the
// strip with full 16-byte stores in copy order (256 threads x 16 bytes = 4 KiB contiguous per pass).  HBM sees the linear array about
// once (the y+1 / z+1 rows of the neighbouring strips mostly hit the L2) and the copy exactly once: bound = HBM, bytes = linear + copy.
// (Before: one thread per 4-byte element, four scattered byte loads and four byte stores each — 0.06-0.16 of the HBM peak.)
enum : int { kBuildQuad = 0, kBuildVoxel = 1, kBuildOct = 2, kBuildRunZ = 3, kBuildRunY = 4 };
constexpr uint32_t kStripBricks = 16, kStripThreads = 256;

template <int BPV, int KIND> struct StripCfg {
	static constexpr uint32_t ny = KIND == kBuildVoxel ? 8u : 9u;                                            // staged rows along y
	static constexpr uint32_t nz = (KIND == kBuildQuad || KIND == kBuildVoxel) ? 8u : 9u;                    // staged slices along z
	static constexpr uint32_t row_voxels = kStripBricks * 8u + 1u;                                           // + the x+1 neighbour of the last cell
	static constexpr uint32_t pitch_words = (row_voxels * BPV + 3u) / 4u + (((row_voxels * BPV + 3u) / 4u) % 2u == 0u ? 1u : 0u);   // odd: rows spread over the banks
	static constexpr uint32_t brick_bytes = KIND == kBuildQuad ? 512u * 4u * BPV : KIND == kBuildVoxel ? 512u * BPV : KIND == kBuildOct ? 512u * 8u * BPV : kRunBrickBytes;
	static constexpr uint32_t chunks_per_brick = brick_bytes / 16u;
};

template <int BPV, int KIND, int PLANE>
__global__ __launch_bounds__(kStripThreads)
void brick_strip_kernel(const void *__restrict__ lin, uint4 *__restrict__ out, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, uint32_t nbx) {
	typedef StripCfg<BPV, KIND> S;
	typedef typename VoxelT<BPV>::type V;
	__shared__ uint32_t rows[S::ny * S::nz * S::pitch_words];
	// brick order: x fastest, then the "other" axis, then the outer axis (quad / voxel / oct / runs along z: y then z; runs along y: z then y)
	const uint32_t bx0 = blockIdx.x * kStripBricks, mid = blockIdx.y, outer = blockIdx.z;
	const uint32_t y0 = (KIND == kBuildRunY ? outer : mid) * 8u, z0 = (KIND == kBuildRunY ? mid : outer) * 8u, x0 = bx0 * 8u;
	const uint32_t t = threadIdx.x;
	// -- stage: row (dy, dz) = voxels x0 .. x0 + 128 of line (min(y0 + dy, Y-1), min(z0 + dz, Z-1)), x clamped to X-1
	{
		const bool fast = ((uint64_t) dim_x * BPV) % 16u == 0u && (uint64_t) x0 + kStripBricks * 8u <= dim_x && ((uintptr_t) lin & 15u) == 0u;
		constexpr uint32_t vec_per_row = (kStripBricks * 8u * BPV) / 16u;                 // whole 16-byte chunks of a row (the +1 voxel comes separately)
		if (fast) {
			for (uint32_t i = t; i < S::ny * S::nz * vec_per_row; i += kStripThreads) {
				const uint32_t r = i / vec_per_row, cx = i - r * vec_per_row, dy = r % S::ny, dz = r / S::ny;
				const uint32_t y = y0 + dy < dim_y ? y0 + dy : dim_y - 1u, z = z0 + dz < dim_z ? z0 + dz : dim_z - 1u;
				const uint4 v = *(const uint4 *) ((const uint8_t *) lin + (((uint64_t) z * dim_y + y) * dim_x + x0) * BPV + (uint64_t) cx * 16u);
				uint32_t *dst = rows + r * S::pitch_words + cx * 4u;
				dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
			}
			for (uint32_t r = t; r < S::ny * S::nz; r += kStripThreads) {                 // the x+1 neighbour of the strip's last cell (clamped at the face)
				const uint32_t dy = r % S::ny, dz = r / S::ny;
				const uint32_t y = y0 + dy < dim_y ? y0 + dy : dim_y - 1u, z = z0 + dz < dim_z ? z0 + dz : dim_z - 1u;
				const uint32_t x = x0 + kStripBricks * 8u < dim_x ? x0 + kStripBricks * 8u : dim_x - 1u;
				((V *) (rows + r * S::pitch_words))[kStripBricks * 8u] = ((const V *) lin)[((uint64_t) z * dim_y + y) * dim_x + x];
			}
		} else {
			for (uint32_t i = t; i < S::ny * S::nz * S::row_voxels; i += kStripThreads) {
				const uint32_t r = i / S::row_voxels, lx = i - r * S::row_voxels, dy = r % S::ny, dz = r / S::ny;
				const uint32_t y = y0 + dy < dim_y ? y0 + dy : dim_y - 1u, z = z0 + dz < dim_z ? z0 + dz : dim_z - 1u;
				const uint32_t x = x0 + lx < dim_x ? x0 + lx : dim_x - 1u;
				((V *) (rows + r * S::pitch_words))[lx] = ((const V *) lin)[((uint64_t) z * dim_y + y) * dim_x + x];
			}
		}
	}
	__syncthreads();
	auto vox = [&](uint32_t lx, uint32_t dy, uint32_t dz) -> uint32_t { return ((const V *) (rows + (dz * S::ny + dy) * S::pitch_words))[lx]; };
	// one 32-bit word of quad element (lx, ly, lz) of slice lz: 1-byte voxels: the whole element; 2-byte: half h (0: row y, 1: row y+1)
	auto quad_word = [&](uint32_t lx, uint32_t ly, uint32_t lz, uint32_t h) -> uint32_t {
		if (BPV == 1) return vox(lx, ly, lz) | (vox(lx + 1u, ly, lz) << 8) | (vox(lx, ly + 1u, lz) << 16) | (vox(lx + 1u, ly + 1u, lz) << 24);
		return vox(lx, ly + h, lz) | (vox(lx + 1u, ly + h, lz) << 16);
	};
	const uint32_t bricks_here = nbx - bx0 < kStripBricks ? nbx - bx0 : kStripBricks;
	const uint64_t first_brick = ((uint64_t) outer * gridDim.y + mid) * nbx + bx0;
	uint4 *dst = out + first_brick * S::chunks_per_brick;
	for (uint32_t c = t; c < bricks_here * S::chunks_per_brick; c += kStripThreads) {
		const uint32_t b = c / S::chunks_per_brick, in = c - b * S::chunks_per_brick, xb = b * 8u;      // brick of the strip, chunk inside it
		uint32_t w[4];
		#pragma unroll
		for (uint32_t i = 0; i < 4u; i++) {
			uint32_t word = 0u;
			if (KIND == kBuildQuad || KIND == kBuildOct) {
				// element index inside the brick and which word of it: quad u8: 1 word per element; quad u16: 2; oct (u16): 4
				constexpr uint32_t words_per_elem = KIND == kBuildOct ? 4u : (uint32_t) BPV;
				const uint32_t local = (in * 4u + i) / words_per_elem, part = (in * 4u + i) % words_per_elem;
				const uint32_t lx = brick_collect(BPV, PLANE, 0, local), ly = brick_collect(BPV, PLANE, 1, local), lz = brick_collect(BPV, PLANE, 2, local);
				if (x0 + xb + lx < dim_x && y0 + ly < dim_y && z0 + lz < dim_z)
					word = KIND == kBuildOct ? quad_word(xb + lx, ly, lz + (part >> 1), part & 1u) : quad_word(xb + lx, ly, lz, part);
			} else if (KIND == kBuildVoxel) {
				constexpr uint32_t per_word = 4u / BPV;
				#pragma unroll
				for (uint32_t j = 0; j < per_word; j++) {
					const uint32_t local = (in * 4u + i) * per_word + j;
					const uint32_t lx = brick_collect(BPV, kPlaneXY, 0, local), ly = brick_collect(BPV, kPlaneXY, 1, local), lz = brick_collect(BPV, kPlaneXY, 2, local);
					if (x0 + xb + lx < dim_x && y0 + ly < dim_y && z0 + lz < dim_z) word |= vox(xb + lx, ly, lz) << (8u * BPV * j);
				}
			} else {
				// run bricks: 64 cell columns (2-D Morton over x and the other axis) x 9 elements along the run axis; element 8 = the next brick's first
				const uint32_t e = in * 4u + i, cell = e / kRunLen, k = e - cell * kRunLen;
				const uint32_t lx = (cell & 1u) | ((cell >> 1) & 2u) | ((cell >> 2) & 4u), lo = ((cell >> 1) & 1u) | ((cell >> 2) & 2u) | ((cell >> 3) & 4u);
				if (KIND == kBuildRunZ) {
					if (x0 + xb + lx < dim_x && y0 + lo < dim_y) word = quad_word(xb + lx, lo, k, 0u);
				} else if (x0 + xb + lx < dim_x && z0 + lo < dim_z) {           // element = the (x,z) neighbourhood of row y0 + k
					word = vox(xb + lx, k, lo) | (vox(xb + lx + 1u, k, lo) << 8) | (vox(xb + lx, k, lo + 1u) << 16) | (vox(xb + lx + 1u, k, lo + 1u) << 24);
				}
			}
			w[i] = word;
		}
		dst[c] = make_uint4(w[0], w[1], w[2], w[3]);
	}
}

template <int BPV, int KIND, int PLANE>
static hipError_t launch_strip(const void *linear, void *copy, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, hipStream_t stream) {
	const uint32_t nbx = (dim_x + 7u) / 8u, nby = (dim_y + 7u) / 8u, nbz = (dim_z + 7u) / 8u;
	const dim3 grid((nbx + kStripBricks - 1u) / kStripBricks, KIND == kBuildRunY ? nbz : nby, KIND == kBuildRunY ? nby : nbz);
	hipLaunchKernelGGL((brick_strip_kernel<BPV, KIND, PLANE>), grid, dim3(kStripThreads), 0, stream, linear, (uint4 *) copy, dim_x, dim_y, dim_z, nbx);
	return hipGetLastError();
}

hipError_t launch_brickify(const void *linear, void *bricked, uint32_t bpv, uint32_t plane, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z,
                           hipStream_t stream) {
	if (bpv == 2) return launch_strip<2, kBuildQuad, kPlaneXY>(linear, bricked, dim_x, dim_y, dim_z, stream);       // 2-byte voxels: one order (Z-order)
	if (plane == kPlaneXZ) return launch_strip<1, kBuildQuad, kPlaneXZ>(linear, bricked, dim_x, dim_y, dim_z, stream);
	if (plane == kPlaneYZ) return launch_strip<1, kBuildQuad, kPlaneYZ>(linear, bricked, dim_x, dim_y, dim_z, stream);
	return launch_strip<1, kBuildQuad, kPlaneXY>(linear, bricked, dim_x, dim_y, dim_z, stream);
}

// linear -> oct bricks (2-byte voxels): element o of the 2-byte brick order holds the 2x2x2 neighbourhood of its cell, 16 bytes
hipError_t launch_brickify_oct(const void *linear, void *oct_bricks, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, hipStream_t stream) {
	return launch_strip<2, kBuildOct, kPlaneXY>(linear, oct_bricks, dim_x, dim_y, dim_z, stream);
}

// linear -> voxel bricks: element o of the (x,y)-plane brick order holds the voxel itself (zero outside the volume)
hipError_t launch_brickify_voxel(const void *linear, void *voxel_bricks, uint32_t bpv, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, hipStream_t stream) {
	if (bpv == 1) return launch_strip<1, kBuildVoxel, kPlaneXY>(linear, voxel_bricks, dim_x, dim_y, dim_z, stream);
	return launch_strip<2, kBuildVoxel, kPlaneXY>(linear, voxel_bricks, dim_x, dim_y, dim_z, stream);
}

// linear -> run bricks (1-byte voxels): element k = 8 of a run is the first element of the next brick along the run axis (index clamped at
// the upper face, where the interpolation weight is exactly 0).  Runs along z: element = (x,y) neighbourhood of slice z; runs along y:
// element = (x,z) neighbourhood of row y.
hipError_t launch_brickify_run(const void *linear, void *run_copy, uint32_t run_layout, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, hipStream_t stream) {
	if (run_layout == kLayoutRunY) return launch_strip<1, kBuildRunY, kPlaneXY>(linear, run_copy, dim_x, dim_y, dim_z, stream);
	return launch_strip<1, kBuildRunZ, kPlaneXY>(linear, run_copy, dim_x, dim_y, dim_z, stream);
}

// ---- linear -> column windows (kLayoutColumn): LDS-tiled like the brick strips --------------------------------------------------------
// A workgroup owns NBU lateral blocks (4x4 cell columns each) side by side along u, one block row along v, and NW consecutive windows along
// the march axis m.  It stages the (4 NBU + 1) x 5 x (3 NW + 1) voxels those windows are made of (+1 neighbours, every index clamped at
// the upper faces, where the interpolation weight is exactly 0) with loads that are contiguous along x — x is u for m = y, z and the march
// axis itself for m = x, hence the two tile shapes — then writes the windows with 16-byte stores in copy order: thread t -> (block, window,
// column), 256 contiguous bytes per (block, window), a block's windows back to back.  Bound: HBM, bytes = linear + copy.
// VOX: the NEAREST windows — 16 consecutive voxels of the column itself (no +1 neighbours), cells 16w .. 16w+15, index clamped at Nm - 1.
template <int M, bool VOX> struct ColBuildCfg {
	static constexpr uint32_t cells = VOX ? kColVoxCells : kColCells;
	static constexpr uint32_t nbu = (M == 0 ? 16u : 128u) / kColEdge, nwin = M == 0 ? (VOX ? 16u : 85u) : (VOX ? 2u : 8u);        // 16 / 128 columns along u per workgroup
	static constexpr uint32_t tu = kColEdge * nbu + (VOX ? 0u : 1u), tv = kColEdge + (VOX ? 0u : 1u), te = cells * nwin + (VOX ? 0u : 1u);
	static constexpr uint32_t tx = M == 0 ? te : tu;                           // tile extent along x (the contiguous axis of the linear array)
	static constexpr uint32_t pitch = (tx + 3u) / 4u * 4u + 4u;               // bytes per staged x-row (multiple of 4, rows shifted over the banks)
	static constexpr uint32_t rows = M == 0 ? tu * tv : tv * te;              // staged rows
};

template <int M, bool VOX>
__global__ __launch_bounds__(256)
void column_build_kernel(const uint8_t *__restrict__ lin, uint4 *__restrict__ out, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z) {
	typedef ColBuildCfg<M, VOX> S;
	constexpr int U = M == 0 ? 1 : 0;
	__shared__ __attribute__((aligned(16))) uint8_t tile[S::rows * S::pitch];
	const uint32_t dim[3] = { dim_x, dim_y, dim_z };
	const uint32_t nbu = col_blocks(dim[U]), nw = col_windows(dim[M], S::cells);
	const uint32_t bu0 = blockIdx.x * S::nbu, bv = blockIdx.y, w0 = blockIdx.z * S::nwin;
	const uint32_t u0 = bu0 * kColEdge, v0 = bv * kColEdge, e0 = w0 * S::cells;
	const uint32_t t = threadIdx.x;
	// row r of the tile: m = y, z: r = dv * te + de holds u = u0 ..; m = x: r = dv * tu + du holds e = e0 ..  (x runs along the row either way)
	auto row_of = [&](uint32_t du, uint32_t dv, uint32_t de) { return M == 0 ? dv * S::tu + du : dv * S::te + de; };
	auto at = [&](uint32_t du, uint32_t dv, uint32_t de) -> uint32_t { return tile[row_of(du, dv, de) * S::pitch + (M == 0 ? de : du)]; };
	{
		const uint32_t x0 = M == 0 ? e0 : u0;
		const bool words = dim_x % 4u == 0u && ((uintptr_t) lin & 3u) == 0u;
		constexpr uint32_t wpr = (S::tx + 3u) / 4u;                           // dwords per row (the last one may be partial)
		for (uint32_t i = t; i < S::rows * wpr; i += 256u) {
			const uint32_t r = i / wpr, cw = i - r * wpr;
			uint32_t y, z;                                                    // the row's two coordinates, clamped at the upper faces
			if (M == 2) { y = v0 + r / S::te; z = e0 + r % S::te; }
			else if (M == 1) { z = v0 + r / S::te; y = e0 + r % S::te; }
			else { z = v0 + r / S::tu; y = u0 + r % S::tu; }
			y = y < dim_y ? y : dim_y - 1u; z = z < dim_z ? z : dim_z - 1u;
			const uint8_t *src = lin + ((uint64_t) z * dim_y + y) * dim_x;
			const uint32_t x = x0 + cw * 4u;
			uint32_t word;
			if (words && x + 3u < dim_x) word = *(const uint32_t *) (src + x);
			else {
				word = 0u;
				for (uint32_t j = 0; j < 4u; j++) { const uint32_t xx = x + j < dim_x ? x + j : dim_x - 1u; word |= (uint32_t) src[xx] << (8u * j); }
			}
			*(uint32_t *) (tile + r * S::pitch + cw * 4u) = word;
		}
	}
	__syncthreads();
	const uint32_t blocks_here = nbu - bu0 < S::nbu ? nbu - bu0 : S::nbu, wins_here = nw - w0 < S::nwin ? nw - w0 : S::nwin;
	constexpr uint32_t kCols = kColEdge * kColEdge;
	for (uint32_t i = t; i < blocks_here * wins_here * kCols; i += 256u) {
		const uint32_t col = i & (kCols - 1u), bw = i / kCols, w = bw % wins_here, b = bw / wins_here;
		const uint32_t du = b * kColEdge + (col & kColEdgeMask), dv = col >> kColEdgeLog2;
		uint32_t word[4];
		#pragma unroll
		for (uint32_t j = 0; j < 4u; j++) {
			if (VOX) {                                                       // dword j = voxels 16w + 4j .. + 3 of the column
				word[j] = 0u;
				#pragma unroll
				for (uint32_t b4 = 0; b4 < 4u; b4++) {
					uint32_t e = e0 + w * S::cells + j * 4u + b4;
					if (e > dim[M] - 1u) e = dim[M] - 1u;
					word[j] |= at(du, dv, e - e0) << (8u * b4);
				}
			} else {
				// element 3w + j, march index clamped at Nm - 1 (tile-relative: the staged index of the clamped element)
				uint32_t e = e0 + w * S::cells + j;
				if (e > dim[M] - 1u) e = dim[M] - 1u;
				const uint32_t de = e - e0;
				word[j] = at(du, dv, de) | (at(du + 1u, dv, de) << 8) | (at(du, dv + 1u, de) << 16) | (at(du + 1u, dv + 1u, de) << 24);
			}
		}
		out[((uint64_t) ((uint64_t) bv * nbu + bu0 + b) * nw + w0 + w) * kCols + col] = make_uint4(word[0], word[1], word[2], word[3]);
	}
}

hipError_t launch_build_column(const void *linear, void *col_copy, uint32_t axis, bool voxels, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, hipStream_t stream) {
	const uint32_t dim[3] = { dim_x, dim_y, dim_z };
	const uint32_t nbu = col_blocks(dim[col_axis_u(axis)]), nbv = col_blocks(dim[col_axis_v(axis)]), nw = col_windows(dim[axis], voxels ? kColVoxCells : kColCells);
	auto go = [&](auto kernel, uint32_t per_u, uint32_t per_w) {
		hipLaunchKernelGGL(kernel, dim3((nbu + per_u - 1u) / per_u, nbv, (nw + per_w - 1u) / per_w), dim3(256), 0, stream, (const uint8_t *) linear, (uint4 *) col_copy, dim_x, dim_y, dim_z);
	};
	if (voxels) {
		if (axis == 0) go(column_build_kernel<0, true>, ColBuildCfg<0, true>::nbu, ColBuildCfg<0, true>::nwin);
		else if (axis == 1) go(column_build_kernel<1, true>, ColBuildCfg<1, true>::nbu, ColBuildCfg<1, true>::nwin);
		else go(column_build_kernel<2, true>, ColBuildCfg<2, true>::nbu, ColBuildCfg<2, true>::nwin);
	} else {
		if (axis == 0) go(column_build_kernel<0, false>, ColBuildCfg<0, false>::nbu, ColBuildCfg<0, false>::nwin);
		else if (axis == 1) go(column_build_kernel<1, false>, ColBuildCfg<1, false>::nbu, ColBuildCfg<1, false>::nwin);
		else go(column_build_kernel<2, false>, ColBuildCfg<2, false>::nbu, ColBuildCfg<2, false>::nwin);
	}
	return hipGetLastError();
}

// ---- feeders: per-ESL-block min/max (RaycasterBase.cpp:101-117) as an HBM-streaming reduction --------------------------
//
// One workgroup per (y-block, z-block) pair: it streams block_dims^2 rows of dim_x voxels with 16-byte loads and keeps
// the 32 x-block minima/maxima in LDS.  min/max are order independent, so the result equals the serial scan exactly.

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// running min / max of the 8-bit samples of one dword, two at a time in packed 16-bit lanes (v_pk_min_u16 / v_pk_max_u16)
template <int BPV>
__device__ __forceinline__ void minmax_word(uint32_t w, us2 &mn, us2 &mx) {
	if (BPV == 1) {
		const uint32_t even = w & 0x00ff00ffu, odd = (w >> 8) & 0x00ff00ffu;
		const us2 e = __builtin_bit_cast(us2, even), o = __builtin_bit_cast(us2, odd);
		mn = __builtin_elementwise_min(mn, __builtin_elementwise_min(e, o));
		mx = __builtin_elementwise_max(mx, __builtin_elementwise_max(e, o));
	} else {                                             // u16 volumes: the ESL grid works on the high byte
		const us2 h = __builtin_bit_cast(us2, (w >> 8) & 0x00ff00ffu);
		mn = __builtin_elementwise_min(mn, h);
		mx = __builtin_elementwise_max(mx, h);
	}
}

template <int BPV>
__device__ __forceinline__ void minmax_chunk(uint4 v, us2 &mn, us2 &mx) {
	minmax_word<BPV>(v.x, mn, mx); minmax_word<BPV>(v.y, mn, mx); minmax_word<BPV>(v.z, mn, mx); minmax_word<BPV>(v.w, mn, mx);
}

// One workgroup per (y-block, z-block) pair of the 32^3 ESL grid: it streams block_dims^2 rows of dim_x voxels and keeps the
// 32 x-block minima / maxima in LDS.  Three paths, same result (min / max are order independent):
//   streaming: a row is 1..256 16-byte chunks (a power of two) and every chunk lies inside one x-block — each thread owns
//              one chunk COLUMN, walks the rows with 8 independent 16-byte loads in flight, reduces in registers and touches
//              LDS once at the end.  This is the HBM-bound path (1024^3: 64 chunks per row, 4 rows per pass);
//   chunked  : 16-byte chunks inside one x-block, any row length;
//   generic  : one voxel at a time (odd dimensions, block edges that are not a multiple of the chunk).
template <int BPV>
__global__ __launch_bounds__(256)
void minmax_kernel(const void *__restrict__ vol, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, uint32_t bd,
                   uint8_t *__restrict__ minmax) {
	__shared__ uint32_t smin[VR_ESL_VOLUME_DIMS], smax[VR_ESL_VOLUME_DIMS];
	const uint32_t yb = blockIdx.x, zb = blockIdx.y;
	if (threadIdx.x < VR_ESL_VOLUME_DIMS) { smin[threadIdx.x] = 255u; smax[threadIdx.x] = 0u; }
	__syncthreads();
	const uint32_t y0 = yb * bd, z0 = zb * bd;
	const uint32_t ny = min(bd, dim_y - y0), nz = min(bd, dim_z - z0);
	const uint32_t rows = ny * nz;
	const uint64_t row_bytes = (uint64_t) dim_x * BPV;
	const uint32_t chunk_voxels = 16 / BPV;
	const bool chunked = (dim_x % chunk_voxels == 0) && (bd % chunk_voxels == 0);
	const uint32_t cpr = chunked ? dim_x / chunk_voxels : 0;          // chunks per row
	if (chunked && cpr <= 256 && (cpr & (cpr - 1)) == 0) {
		const uint32_t rows_per_pass = 256 / cpr;
		const uint32_t cx = threadIdx.x & (cpr - 1), r0 = threadIdx.x / cpr;
		us2 mn = { 255, 255 }, mx = { 0, 0 };
		auto row_ptr = [&](uint32_t row) {
			const uint32_t z = z0 + row / ny, y = y0 + row - (row / ny) * ny;
			return (const uint4 *) ((const uint8_t *) vol + ((uint64_t) z * dim_y + y) * row_bytes + (uint64_t) cx * 16);
		};
		uint32_t row = r0;
		for (; row + 7 * rows_per_pass < rows; row += 8 * rows_per_pass) {
			uint4 v[8];
			#pragma unroll
			for (int u = 0; u < 8; u++) v[u] = *row_ptr(row + u * rows_per_pass);
			#pragma unroll
			for (int u = 0; u < 8; u++) minmax_chunk<BPV>(v[u], mn, mx);
		}
		for (; row < rows; row += rows_per_pass) minmax_chunk<BPV>(*row_ptr(row), mn, mx);
		const uint32_t xb = (cx * chunk_voxels) / bd;
		atomicMin(&smin[xb], (uint32_t) min(mn.x, mn.y));
		atomicMax(&smax[xb], (uint32_t) max(mx.x, mx.y));
	} else if (chunked) {
		const uint32_t total = rows * cpr;                            // < 2^32: rows <= 2^16 * 2^16 / ... bounded by the slab size
		for (uint32_t c = threadIdx.x; c < total; c += 256) {
			const uint32_t row = c / cpr, cx = c - row * cpr;
			const uint32_t y = y0 + row % ny, z = z0 + row / ny;
			const uint8_t *p = (const uint8_t *) vol + ((uint64_t) z * dim_y + y) * row_bytes + (uint64_t) cx * 16;
			us2 mn = { 255, 255 }, mx = { 0, 0 };
			minmax_chunk<BPV>(*(const uint4 *) p, mn, mx);
			const uint32_t xb = (cx * chunk_voxels) / bd;
			atomicMin(&smin[xb], (uint32_t) min(mn.x, mn.y));
			atomicMax(&smax[xb], (uint32_t) max(mx.x, mx.y));
		}
	} else {
		const uint64_t total = (uint64_t) rows * dim_x;
		for (uint64_t i = threadIdx.x; i < total; i += 256) {
			const uint32_t row = (uint32_t) (i / dim_x), x = (uint32_t) (i - (uint64_t) row * dim_x);
			const uint32_t y = y0 + row % ny, z = z0 + row / ny;
			const uint64_t e = ((uint64_t) z * dim_y + y) * dim_x + x;
			const uint32_t s = BPV == 1 ? ((const uint8_t *) vol)[e] : (uint32_t) (((const uint16_t *) vol)[e] >> 8);
			atomicMin(&smin[x / bd], s);
			atomicMax(&smax[x / bd], s);
		}
	}
	__syncthreads();
	const uint32_t nxb = (dim_x + bd - 1) / bd;
	if (threadIdx.x < nxb && threadIdx.x < VR_ESL_VOLUME_DIMS) {
		const uint32_t e = zb * VR_ESL_VOLUME_DIMS * VR_ESL_VOLUME_DIMS + yb * VR_ESL_VOLUME_DIMS + threadIdx.x;
		minmax[2 * e] = (uint8_t) smin[threadIdx.x];
		minmax[2 * e + 1] = (uint8_t) smax[threadIdx.x];
	}
}

__global__ void minmax_init_kernel(uint8_t *minmax) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < 32u * 32u * 32u) { minmax[2 * i] = 255; minmax[2 * i + 1] = 0; }   // RaycasterBase.cpp:101-104
}

hipError_t launch_minmax(const void *volume, uint32_t bpv, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z,
                         uint32_t bd, uint8_t *minmax_dev, hipStream_t stream) {
	hipLaunchKernelGGL(minmax_init_kernel, dim3(128), dim3(256), 0, stream, minmax_dev);
	const dim3 grid((dim_y + bd - 1) / bd, (dim_z + bd - 1) / bd);
	if (bpv == 1) hipLaunchKernelGGL(minmax_kernel<1>, grid, dim3(256), 0, stream, volume, dim_x, dim_y, dim_z, bd, minmax_dev);
	else          hipLaunchKernelGGL(minmax_kernel<2>, grid, dim3(256), 0, stream, volume, dim_x, dim_y, dim_z, bd, minmax_dev);
	return hipGetLastError();
}

// ---- feeders: 256-bin histogram (ModelBase.cpp:19-26) ----------------------------------------------------------------------

// Each wave keeps kHistCopies interleaved copies of the 256 bins in LDS (bin b of copy c at b * kHistCopies + c, c = lane % 8):
// real volumes are dominated by a few values (air), and lanes that hit the same bin in one ds_add serialise — spreading them
// over 8 copies in 8 different banks cuts that 8-fold.  16-byte loads, 4 in flight per thread.
// LDS histograms per wave: lanes that count the same bin in the same instruction serialise, so every wave keeps several copies (lane & (copies - 1)).
// Measured on 1024^3 (scripts/feeder_probe.py): 1-byte voxels 3.3 / 4.0 / 4.6 / 3.1 TB/s with 4 / 8 / 16 / 32 copies, 2-byte voxels 4.7 / 4.8 / 4.4 / 4.0.
template <int BPV> struct HistCopies { static constexpr uint32_t value = BPV == 1 ? 16u : 8u; };

template <int BPV>
__global__ __launch_bounds__(256)
void histogram_kernel(const void *__restrict__ vol, uint64_t voxels, unsigned long long *__restrict__ hist) {
	constexpr uint32_t kHistCopies = HistCopies<BPV>::value;
	__shared__ uint32_t sh[4][256 * kHistCopies];
	for (uint32_t i = threadIdx.x; i < 4 * 256 * kHistCopies; i += 256) ((uint32_t *) sh)[i] = 0;
	__syncthreads();
	uint32_t *mine = sh[threadIdx.x >> 6] + (threadIdx.x & (kHistCopies - 1));
	auto count = [&](uint32_t bin) { atomicAdd(&mine[bin * kHistCopies], 1u); };
	auto chunk = [&](uint4 v) {
		const uint32_t w[4] = { v.x, v.y, v.z, v.w };
		#pragma unroll
		for (int i = 0; i < 4; i++) {
			if (BPV == 1) { count(w[i] & 0xffu); count((w[i] >> 8) & 0xffu); count((w[i] >> 16) & 0xffu); count(w[i] >> 24); }
			else          { count((w[i] >> 8) & 0xffu); count(w[i] >> 24); }          // u16: high byte
		}
	};
	const uint64_t stride = (uint64_t) gridDim.x * 256;
	const uint64_t vec = voxels * BPV / 16;              // whole 16-byte chunks
	uint64_t c = (uint64_t) blockIdx.x * 256 + threadIdx.x;
	for (; c + 3 * stride < vec; c += 4 * stride) {
		const uint4 v0 = ((const uint4 *) vol)[c], v1 = ((const uint4 *) vol)[c + stride];
		const uint4 v2 = ((const uint4 *) vol)[c + 2 * stride], v3 = ((const uint4 *) vol)[c + 3 * stride];
		chunk(v0); chunk(v1); chunk(v2); chunk(v3);
	}
	for (; c < vec; c += stride) chunk(((const uint4 *) vol)[c]);
	if (blockIdx.x == 0) {                               // tail (fewer than 16 bytes)
		const uint64_t done = vec * 16 / BPV;
		for (uint64_t i = done + threadIdx.x; i < voxels; i += 256)
			count(BPV == 1 ? ((const uint8_t *) vol)[i] : (uint32_t) (((const uint16_t *) vol)[i] >> 8));
	}
	__syncthreads();
	const uint32_t b = threadIdx.x;
	unsigned long long sum = 0;
	for (uint32_t w = 0; w < 4; w++)
		for (uint32_t cp = 0; cp < kHistCopies; cp++) sum += sh[w][b * kHistCopies + cp];
	if (sum) atomicAdd(&hist[b], sum);
}

hipError_t launch_histogram(const void *volume, uint32_t bpv, uint64_t voxels, unsigned long long *hist, hipStream_t stream) {
	hipError_t e = hipMemsetAsync(hist, 0, 256 * sizeof(unsigned long long), stream);
	if (e != hipSuccess) return e;
	// each workgroup may add at most 2^32-1 per bin into its LDS counters: bound the voxels per workgroup
	uint64_t blocks = (voxels + (1ull << 24) - 1) >> 24;
	if (blocks < 2048) blocks = 2048;
	if (bpv == 1) hipLaunchKernelGGL(histogram_kernel<1>, dim3((uint32_t) blocks), dim3(256), 0, stream, volume, voxels, hist);
	else          hipLaunchKernelGGL(histogram_kernel<2>, dim3((uint32_t) blocks), dim3(256), 0, stream, volume, voxels, hist);
	return hipGetLastError();
}

// ---- synthetic benchmark volumes, integer-only, generated straight into HBM --------------------------------

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
	h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
	return h;
}

// One thread per 16-byte chunk of the array (16 or 8 consecutive voxels along x, wrapping into the next row / slice), one 16-byte store.
// The shell's 1000 * d2 / (N * N) is an exact integer quotient (< 12000): a shift where N is a power of two, else formed in DOUBLE
// precision (every quantity is an integer below 2^53 for n <= 65535, so the fused remainder r = num - q * N^2 is exact and the estimate
// floor(num / N^2) is corrected by its sign) — the 64-bit integer division it replaces was what the old one-voxel-per-thread kernel
// spent its time in.  What remains is the murmur finaliser per voxel: the kernel is bound by integer issue, not by HBM.
// `shift` != 0: n is a power of two, n * n = 1 << shift, and the quotient is a 64-bit multiply and a shift (the benchmark sizes).
// The shell is zero wherever 1000 * d2 / n^2 lies outside (120, 600): |q - 360| >= 240.
template <int BPV>
__device__ __forceinline__ uint32_t synthetic_voxel(uint32_t kind, uint64_t idx, int ax, uint64_t ayz2, double nn, double inv_nn, uint32_t shift, uint32_t seed) {
	const uint32_t h = fmix32((uint32_t) (idx ^ (idx >> 32)) + seed * 0x9E3779B9u);
	if (kind != 0) return h & 255u;
	const uint64_t num = 1000ull * ((uint64_t) ((int64_t) ax * ax) + ayz2);   // < 2^53 for every n <= 65535
	int q;
	if (shift != 0u) q = (int) (num >> shift);
	else {
		double e = __builtin_floor((double) num * inv_nn);
		const double r = __builtin_fma(-e, nn, (double) num);        // exact remainder of the estimate (every quantity is an integer < 2^53)
		if (r < 0.0) e -= 1.0; else if (r >= nn) e += 1.0;
		q = (int) e;
	}
	int t = q - 360;
	if (t < 0) t = -t;
	int shell = 255 - (int) ((uint32_t) t * 255u / 240u);
	if (shell < 0) shell = 0;
	const uint32_t v = (uint32_t) shell + (h & 15u);
	return v > 255u ? 255u : v;
}

template <int BPV>
__global__ __launch_bounds__(256)
void generate_kernel(void *__restrict__ vol, uint32_t kind, uint32_t n, uint32_t seed) {
	constexpr uint32_t kPerChunk = 16u / BPV;
	const int N = (int) n;
	const double nn = (double) n * (double) n, inv_nn = 1.0 / nn;
	const uint32_t shift = (n & (n - 1u)) == 0u ? 2u * (uint32_t) __builtin_ctz(n) : 0u;
	const uint64_t total = (uint64_t) n * n * n, chunks = total / kPerChunk;
	const uint64_t stride = (uint64_t) gridDim.x * 256;
	for (uint64_t c = (uint64_t) blockIdx.x * 256 + threadIdx.x; c < chunks; c += stride) {
		uint64_t idx = c * kPerChunk;
		const uint64_t row = idx / n;
		uint32_t x = (uint32_t) (idx - row * n), y = (uint32_t) (row % n), z = (uint32_t) (row / n);
		int64_t ay = 2 * (int64_t) y + 1 - N, az = 2 * (int64_t) z + 1 - N;
		uint64_t ayz2 = (uint64_t) (ay * ay + az * az);
		uint32_t w[4] = { 0u, 0u, 0u, 0u };
		#pragma unroll
		for (uint32_t j = 0; j < kPerChunk; j++) {
			const uint32_t v = synthetic_voxel<BPV>(kind, idx, 2 * (int) x + 1 - N, ayz2, nn, inv_nn, shift, seed);
			if (BPV == 1) w[j / 4u] |= v << (8u * (j % 4u)); else w[j / 2u] |= (v * 257u) << (16u * (j % 2u));
			idx++;
			if (++x == n) {                                       // next row (and slice)
				x = 0;
				if (++y == n) { y = 0; z++; az = 2 * (int64_t) z + 1 - N; }
				ay = 2 * (int64_t) y + 1 - N; ayz2 = (uint64_t) (ay * ay + az * az);
			}
		}
		((uint4 *) vol)[c] = make_uint4(w[0], w[1], w[2], w[3]);
	}
	if (blockIdx.x == 0) {                                           // fewer than 16 bytes left over
		for (uint64_t idx = chunks * kPerChunk + threadIdx.x; idx < total; idx += 256) {
			const uint64_t row = idx / n;
			const int x = (int) (idx - row * n);
			const int64_t ay = 2 * (int64_t) (row % n) + 1 - N, az = 2 * (int64_t) (row / n) + 1 - N;
			const uint32_t v = synthetic_voxel<BPV>(kind, idx, 2 * x + 1 - N, (uint64_t) (ay * ay + az * az), nn, inv_nn, shift, seed);
			if (BPV == 1) ((uint8_t *) vol)[idx] = (uint8_t) v; else ((uint16_t *) vol)[idx] = (uint16_t) (v * 257u);
		}
	}
}

hipError_t launch_generate(void *volume, uint32_t kind, uint32_t n, uint32_t seed, uint32_t bpv, hipStream_t stream) {
	if (bpv == 1) hipLaunchKernelGGL(generate_kernel<1>, dim3(8192), dim3(256), 0, stream, volume, kind, n, seed);
	else          hipLaunchKernelGGL(generate_kernel<2>, dim3(8192), dim3(256), 0, stream, volume, kind, n, seed);
	return hipGetLastError();
}

}  // namespace vr
