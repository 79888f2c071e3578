// vr_hip_api.cpp — the C ABI of include/vr_hip.h: one opaque context per GPU that owns the device copies of the
// volume / transfer function / ESL bit-volume / framebuffer and launches the gfx950 kernels of vr_kernels.hip.
//
// It replaces the device management the reference spreads over GPURenderer1.cu:17-28,65-112, GPURenderer23.cu:55-81
// and GPURenderer4.cu:89-153 (static globals, cudaMalloc/cudaMemcpy per set_*, cuda_safe_call -> exit).  Differences
// by design: no globals, no exit(), errors are return codes + vr_hip_last_error(); the per-frame parameter block is
// the kernel argument itself (no cudaMemcpyToSymbol per frame, GPURenderer23.cu:65,77); the frame clear is fused into
// the kernel (no cudaMemset per frame, GPURenderer1.cu:107).  There is no CPU fallback.
#include "vr_device.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace vr;

#ifndef VR_ORDER_ALWAYS
#define VR_ORDER_ALWAYS 0      // 1: the measured-cost tile order also for the full march (A/B: TRILINEAR +2 %, NEAREST -2 %)
#endif

namespace {

constexpr int kEventRing = 256;

struct EventPair { hipEvent_t start = nullptr, stop = nullptr; bool pending = false; };

}  // namespace

struct vr_ctx {
	int device = -1;
	hipStream_t stream = nullptr;           // context-owned stream for the synchronous entry points
	hipStream_t stream_first = nullptr;     // vr_hip_render: the first row slice of a frame runs here, AHEAD of the rest (highest stream priority)
	// window
	uint32_t win_w = 0, win_h = 0;
	void *fb = nullptr; size_t fb_bytes = 0;
	// transfer function + ESL
	float *tf = nullptr; uint32_t *esl = nullptr; bool tf_set = false;
	float tf_zero_below = -1.0f;            // leading all-zero entries of the resident TF (exact transparent-sample shortcut)
	// volume
	void *vol = nullptr; uint64_t vol_elems = 0; uint32_t dim[3] = { 0, 0, 0 }; uint32_t bpv = 0;
	// Brick copies of the resident volume (vr_device.h), built ON FIRST USE by the frame that wants them (or ahead of time by
	// vr_hip_prepare): a NEAREST-only session never pays for the quad / run copies, a session that only looks along z never
	// builds the (x,z) / (y,z) planes.  copy[kCopyQuadXY..YZ] = quad bricks per chunk plane, kCopyRunZ / kCopyRunY = run bricks,
	// kCopyVoxel = voxel bricks (what NEAREST reads), kCopyOct = oct bricks (what TRILINEAR reads for 2-byte voxels).  copy_failed: a build was refused (HBM guard / allocation) — not retried
	// until the next set_volume, so a frame never stalls twice on the same refusal.
	void *copy[kCopyKinds] = {};            // kCopyColX..Z = column windows along x / y / z (orthogonal full-march frames along that axis)
	float copy_build_ms[kCopyKinds] = {};
	bool copy_failed[kCopyKinds] = {};
	int32_t column_force = 0;               // vr_hip_set_brick_plane: 0 = per view (views along a volume axis), 1 (plane 8) = every orthogonal full-march frame, -1 (plane 9) = never
	float upload_ms = 0;                    // host -> HBM copy (or generation) of the linear array in the last set_volume
	int32_t brick_plane_force = -1;         // -1 = per view (plane perpendicular to the dominant view axis; run bricks for oblique views),
	                                        // 0..2 = that chunk plane, 3 = the run bricks (testing)
	uint32_t layout = VR_LAYOUT_BRICKED;
	uint32_t force_wide = 0;
	uint32_t force_clamp_fetch = 0;              // testing aid (vr_hip_set_wide_addressing bit 2)
	int32_t  tile_lane_map = -1;                 // -1 = choose per frame (choose_tile_mapping), else forced
	uint32_t tile_phase_x = 0, tile_phase_y = 0;
	// the last few automatic tile mappings (lane order, tile phase), keyed by the frame parameters and the volume size (a benchmark
	// cycles 8 views).  cost / order / dual_state / order_ready / order_stream serve only the MEASURED run-copy choice
	// (vr_hip_set_brick_plane(6), the validator of the analytic rule: frames 0 / 1 of a parameter set run on the copy along z / y,
	// frames 2 / 3 do so again and record their tile costs, the choice kernel behind frame 3 fills order[], frames 4.. read it).
	struct MapEntry { vr_params p; uint32_t dim[3]; uint32_t lane_map, phase_x, phase_y, straddle_permille;
	                  uint32_t *cost = nullptr, *order = nullptr; uint32_t capacity = 0, order_tiles = 0;
	                  uint32_t dual_state = 0;
	                  hipEvent_t order_ready = nullptr; hipStream_t order_stream = nullptr; };
	MapEntry map_cache[32]; uint32_t map_cached = 0, map_next = 0;      // (8 benchmark views x the two row slices of vr_hip_render, and the whole frames beside them)
	// Measured-cost launch orders (vr_kernels.hip tile_order_kernel), keyed by what the cost of a tile DEPENDS on — sampling mode,
	// leaping / termination on or off, projection, the view's major axis, the band partition, the tile grid and the copy read — not by
	// the byte image of the parameters: a camera that moves keeps its entry, every frame launches its tiles in the order the most recent
	// FINISHED recording gives and records its own costs for the next one; a frame repeated with identical parameters stops recording
	// after two (the first may have run on cold caches).  Three slots per entry so that frames in flight on several streams never
	// share a buffer that is being rewritten: a slot is recycled only once every frame that read or recorded it is known to have
	// finished (the event ring below) and its order kernel has run (`ready`) — ADVICE r3: no buffer of a frame in flight is touched.
	struct SchedSlot { uint32_t *cost = nullptr, *order = nullptr; uint32_t capacity = 0, ntiles = 0; hipEvent_t ready = nullptr; hipStream_t stream = nullptr;
	                   uint64_t last_seq = 0, issue_seq = 0; bool valid = false; };
	struct OrderKey { int32_t direction_q[3]; uint32_t sampling, esl, ert, perspective, major_axis, layout, view_w, view_h, x0, out_width, out_rows, band_rows, band_stride, band_first,
	                  dim[3], tiles_x, tiles_y; };
	struct OrderEntry { OrderKey key; bool used = false; SchedSlot slot[3]; vr_params last; bool has_last = false; uint32_t repeats = 0; uint64_t lru = 0; };
	OrderEntry order_cache[32];
	uint64_t seq_next = 1, completed_seq = 0;      // frames launched so far + 1; every frame <= completed_seq is known to have finished
	uint64_t ring_seq[kEventRing] = {};            // the frame whose events sit in ring entry i
	uint32_t tile_scheduling = 1;           // vr_hip_set_tile_scheduling: 0 = tile = workgroup id, 1 = measured-cost order, 2 = cost map
	vr_launch_info last_launch = {};        // vr_hip_last_launch
	uint32_t *cost_map = nullptr;           // mode 2: what every tile of the LAST frame cost (vr_hip_read_tile_costs)
	uint32_t cost_map_capacity = 0, cost_map_tiles_x = 0, cost_map_tiles_y = 0;
	bool oct_always = false;                // vr_hip_set_brick_plane(5): 2-byte voxels read the oct bricks for every view (testing)
	// feeders scratch
	uint8_t *minmax = nullptr; unsigned long long *hist = nullptr;
#ifdef VR_BOUNDS_CHECK
	uint32_t *bc_fault = nullptr;           // debug build: first out-of-bounds access of a frame (RayKernelArgs::bc_fault)
#endif
	// timing
	EventPair ring[kEventRing]; int ring_head = 0;
	hipEvent_t aux_start = nullptr, aux_stop = nullptr;
	float last_kernel_ms = 0, last_total_ms = 0; uint64_t launches = 0; double kernel_ms_sum = 0; float kernel_ms_max = 0, total_ms_max = 0;
	std::string err;
};

namespace {

int fail(vr_ctx *c, int code, const char *what, hipError_t e = hipSuccess) {
	if (c) {
		char buf[512];
		if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s (%d)", what, hipGetErrorString(e), (int) e);
		else snprintf(buf, sizeof buf, "%s", what);
		c->err = buf;
	}
	return code;
}

#define VR_TRY(c, expr)                                                                              \
	do {                                                                                             \
		hipError_t e_ = (expr);                                                                      \
		if (e_ != hipSuccess) {                                                                      \
			(void) hipGetLastError();                                                                \
			return fail((c), e_ == hipErrorOutOfMemory ? VR_ERR_ALLOC : VR_ERR_HIP, #expr, e_);      \
		}                                                                                            \
	} while (0)

bool copy_possible(const vr_ctx *c, uint32_t kind);
uint64_t copy_bytes(const vr_ctx *c, uint32_t kind);
hipError_t drain(vr_ctx *c);
const void *copy_for(vr_ctx *c, uint32_t kind);

bool finite3(const float *v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }

// folds one finished event pair into the statistics
void harvest(vr_ctx *c, EventPair &p) {
	if (!p.pending) return;
	float ms = 0;
	if (hipEventSynchronize(p.stop) == hipSuccess && hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
		c->last_kernel_ms = ms; c->kernel_ms_sum += ms; c->launches++;
		if (ms > c->kernel_ms_max) c->kernel_ms_max = ms;
	}
	p.pending = false;
}

int validate_params(vr_ctx *c, const vr_params *p) {
	if (p == nullptr) return fail(c, VR_ERR_INVALID, "params is NULL");
	const vr_view &v = p->view;
	if (v.width == 0 || v.height == 0 || v.width > 65535u || v.height > 65535u)      // View::dims is ushort2
		return fail(c, VR_ERR_INVALID, "view dims out of range (1..65535)");
	if (!finite3(v.origin) || !finite3(v.direction) || !finite3(v.right_plane) || !finite3(v.up_plane) || !finite3(v.light_pos))
		return fail(c, VR_ERR_INVALID, "view vectors must be finite");
	if (!(p->ray_step >= 1e-6f) || !std::isfinite(p->ray_step))
		return fail(c, VR_ERR_INVALID, "ray_step must be finite and >= 1e-6");
	if (!std::isfinite(p->ray_threshold) || !std::isfinite(p->light_kd))
		return fail(c, VR_ERR_INVALID, "ray_threshold / light_kd must be finite");
	// The per-wave shortcuts for transparent samples skip the `acc.w > threshold` test, which is exact only if that test cannot
	// newly fire at an unchanged acc.w >= 0, i.e. for threshold >= 0 (the reference's setter keeps it in [0.5, 1], RaycasterBase.cpp:31-33)
	if (p->ray_threshold < 0.0f)
		return fail(c, VR_ERR_INVALID, "ray_threshold must be >= 0");
	if (p->esl && (p->esl_block_dims == 0 || p->esl_block_dims > 65535u || !finite3(p->esl_block_size)))   // unsigned short in the reference
		return fail(c, VR_ERR_INVALID, "esl_block_dims must be in 1..65535 and esl_block_size finite when esl is on");
	if (p->sampling != VR_SAMPLE_NEAREST && p->sampling != VR_SAMPLE_TRILINEAR && p->sampling != VR_SAMPLE_TRILINEAR_Q8)
		return fail(c, VR_ERR_INVALID, "unknown sampling mode");
	if (p->out_width == 0 || p->out_rows == 0 || p->out_width > 65535u || p->out_rows > 65535u)
		return fail(c, VR_ERR_INVALID, "out_width / out_rows out of range");
	if (p->band_rows == 0 || p->band_stride == 0 || p->band_first >= p->band_stride)
		return fail(c, VR_ERR_INVALID, "band partition invalid (band_rows > 0, band_first < band_stride)");
	return VR_OK;
}

// Which pixels share a lane quad, and where the tile grid starts — speed only, the image does not depend on it.
// Measured on MI355X (scripts/ubench/tcp_coalesce.hip): the vector memory pipeline handles a wave's
// gather 4 consecutive lanes at a time and runs ~3.5x faster when the 4 addresses share one aligned 16-byte chunk (= the
// 2x2 (x,y) element block of the brick order).  So:
//  * a view along a volume axis puts 2x2-pixel blocks into a quad (kLaneBlocks): at the usual >= 2 pixels per voxel the four
//    samples fall into one or two cells;
//  * any other view keeps 4x1-pixel quads, along the screen axis whose image in the volume stays closest to the x/y plane;
//  * orthogonal views also shift the tile grid by 0..3 pixels so that 4-pixel groups start on even cells (when the pixel
//    pitch is commensurate with the voxel grid — the reference's default zoom — every quad then reads a single chunk).
// Returns, for orthogonal views, how many of the sampled 4-pixel groups still straddle cells under the best phase (per mille);
// 1000 for views whose quads cannot be aligned at all (perspective, or not along an axis).
uint32_t choose_tile_mapping(RayKernelArgs &a) {
	const vr_view &v = a.p.view;
	// voxel-cell coordinate of a position p: TRILINEAR p * N/2 + N/2 - 1/2 (texel space), NEAREST (p + 1) / 2 * N (ModelBase.h:17-23)
	const bool nearest = a.p.sampling == VR_SAMPLE_NEAREST;
	const float half[3] = { a.half_x, a.half_y, a.half_z };
	const float off[3] = { nearest ? a.half_x : a.off_x, nearest ? a.half_y : a.off_y, nearest ? a.half_z : a.off_z };
	float d[3], sx[3], sy[3];
	for (int i = 0; i < 3; i++) { d[i] = v.direction[i] * half[i]; sx[i] = v.right_plane[i] * half[i]; sy[i] = v.up_plane[i] * half[i]; }
	auto norm = [](const float *u) { return std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]); };
	auto major = [](const float *u) { int m = 0; for (int i = 1; i < 3; i++) if (std::fabs(u[i]) > std::fabs(u[m])) m = i; return m; };
	a.lane_map = kLaneRows; a.phase_x = a.phase_y = 0;
	const float dn = norm(d);
	if (!(dn > 0.0f)) return 1000u;
	if (std::fabs(d[major(d)]) > 0.98f * dn) a.lane_map = kLaneBlocks;
	else {
		// 4x1-pixel quads along the screen axis that leaves the chunk plane least: `out` is the axis not in the plane
		// (oblique views always read the (x,y) copy; forced planes are a testing aid)
		const int out = a.brick_plane == kPlaneXY ? 2 : (a.brick_plane == kPlaneXZ ? 1 : 0);
		const float nx = norm(sx), ny = norm(sy);
		if (nx > 0.0f && ny > 0.0f && std::fabs(sy[out]) * nx < std::fabs(sx[out]) * ny) a.lane_map = kLaneColumns;
	}
	if (v.perspective) return 1000u;
	const bool axis_aligned = a.lane_map == kLaneBlocks;
	// orthogonal view: every ray has the same direction; [k_in, k_out] = the central ray's path through the cube
	float k_in = -1e30f, k_out = 1e30f;
	for (int i = 0; i < 3; i++) {
		if (v.direction[i] == 0.0f) continue;
		const float k1 = (-1.0f - v.origin[i]) / v.direction[i], k2 = (1.0f - v.origin[i]) / v.direction[i];
		k_in = std::fmax(k_in, std::fmin(k1, k2)); k_out = std::fmin(k_out, std::fmax(k1, k2));
	}
	if (!(k_in < k_out) || !(k_out < 1e29f)) return 1000u;
	k_in = std::fmax(k_in, 0.0f);
	// Voxel cell of frame pixel (gx, gy) at depth k along volume axis `ax`, with the KERNEL'S OWN fp32 operations (get_ray,
	// then fma(k, A, B) / (pos + 1) * half): pixels of an axis-aligned view at the reference's default zoom sit exactly on
	// cell boundaries, so which neighbour a boundary pixel joins is decided by the last bit — and may change along the ray
	// when the direction carries rounding noise (pose (180,90,0): components of 4e-8), hence several depths are sampled.
	auto cell_of = [&](int ax, long gx, long gy, float k) -> long {
		const float fx = (float) ((int) gx - (int) (v.width / 2u)), fy = (float) ((int) gy - (int) (v.height / 2u));
		float o = v.origin[ax] + v.right_plane[ax] * fx;
		o = o + v.up_plane[ax] * fy;
		float t;
		if (nearest) t = ((o + v.direction[ax] * k) + 1.0f) * half[ax];
		else         t = std::fmaf(k, v.direction[ax] * half[ax], std::fmaf(o, half[ax], off[ax]));
		return (long) std::floor(t);
	};
	// Phase 0..7 of the 8-pixel wave columns / rows along one screen direction.  Primary cost (what the vector memory pipeline
	// charges): 4-pixel groups that leave their aligned cell pair, pixel pairs that straddle two cells.  Secondary: 8-pixel wave
	// edges that are not 4-cell (128-byte line) boundaries — a wave whose 4x4 cells sit inside one line column touches a
	// quarter of the lines, and workgroup footprints that end on line boundaries do not fetch their border lines twice
	// (measured before: 2.0x the compulsory bytes at L2 with 4-pixel alignment only).
	long groups_seen[2] = { 0, 0 }, groups_straddling[2] = { 0, 0 };      // of the winning phase, per screen direction (0 = x, 1 = y)
	auto best_phase = [&](bool horizontal, uint32_t first, uint32_t count, uint32_t other_centre) {
		const float *sdir = horizontal ? sx : sy;
		const int ax = major(sdir);
		const long hi = (long) (2.0f * half[ax]) - 1;
		uint32_t best = 0; long best_cost = -1, best_seen = 0, best_bad = 0;
		for (uint32_t ph = 0; ph < 8; ph++) {
			long cost = 0, seen = 0, bad = 0;
			for (int depth = 0; depth < 4; depth++) {
				const float k = k_in + (k_out - k_in) * (0.125f + 0.25f * (float) depth);
				for (long g0 = -(long) ph; g0 < (long) count; g0 += 8 * 7) {      // every seventh wave column is plenty
					long cell[8]; bool in[8];
					for (int i = 0; i < 8; i++) {
						const long g = (long) first + g0 + i;
						cell[i] = horizontal ? cell_of(ax, g, other_centre, k) : cell_of(ax, other_centre, g, k);
						in[i] = g0 + i >= 0 && g0 + i < (long) count && cell[i] >= 0 && cell[i] <= hi;
					}
					for (int q = 0; q < 8; q += 4) {
						const long before = cost;
						if (in[q] && in[q + 3] && (cell[q] >> 1) != (cell[q + 3] >> 1)) cost += 32;          // the group leaves its aligned cell pair
						if (in[q] && in[q + 1] && cell[q] != cell[q + 1]) cost += 16;                        // a pixel pair straddles two cells
						if (in[q + 2] && in[q + 3] && cell[q + 2] != cell[q + 3]) cost += 16;
						if (in[q] && in[q + 3]) { seen++; if (cost != before) bad++; }
					}
					if (in[0] && in[7] && (cell[0] >> 2) != (cell[7] >> 2)) cost += 1;                       // the wave leaves its line column
				}
			}
			if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ph; best_seen = seen; best_bad = bad; }
		}
		groups_seen[horizontal ? 0 : 1] = best_seen; groups_straddling[horizontal ? 0 : 1] = best_bad;
		return best;
	};
	// rows: local row ly maps to frame row gy; with bands of a multiple of 8 rows (or one band) gy = ly + const (mod 8)
	const uint32_t gy0 = a.p.band_first * a.p.band_rows;
	a.phase_x = best_phase(true, a.p.x0, a.p.out_width, gy0 + std::min(a.p.out_rows, a.p.band_rows) / 2u);
	a.phase_y = best_phase(false, gy0, a.p.out_rows < a.p.band_rows ? a.p.out_rows : a.p.band_rows, a.p.x0 + a.p.out_width / 2u);
	if (!axis_aligned || groups_seen[0] + groups_seen[1] == 0) return 1000u;
	// One screen direction clean, the other not (pose (90,0,0): the rows of pixels sit on cell boundaries of the axis that carries the
	// 4e-8 of rounding noise, 9 % of the 2x2 blocks straddle; the columns are clean): TRILINEAR puts the four lanes of a quad along the
	// CLEAN direction — measured on that pose 2.49 -> 2.24 ms (scripts/phase_grid_probe.py); with both directions clean the 2x2 blocks
	// stay (2.11 against 2.16 ms on pose (0,0,0)).  What is returned is the share of straddling groups of the lane order chosen.
	if (!nearest && groups_seen[0] > 0 && groups_seen[1] > 0) {
		const bool dirty_x = groups_straddling[0] * 50 > groups_seen[0], dirty_y = groups_straddling[1] * 50 > groups_seen[1];     // > 2 %
		if (dirty_y && !dirty_x) { a.lane_map = kLaneRows; return (uint32_t) (groups_straddling[0] * 1000 / groups_seen[0]); }
		if (dirty_x && !dirty_y) { a.lane_map = kLaneColumns; return (uint32_t) (groups_straddling[1] * 1000 / groups_seen[1]); }
	}
	return (uint32_t) ((groups_straddling[0] + groups_straddling[1]) * 1000 / (groups_seen[0] + groups_seen[1]));
}

// Moves c->completed_seq forward over every frame whose stop event has been reached (frames complete out of order across streams:
// the mark only passes a frame once all earlier ones are done too — conservative, which is what recycling a buffer needs).
void advance_completed(vr_ctx *c) {
	while (c->completed_seq + 1 < c->seq_next) {
		const uint64_t next = c->completed_seq + 1;
		const int slot = (int) ((c->ring_head + kEventRing - (int) ((c->seq_next - next) % kEventRing)) % kEventRing);
		if (c->seq_next - next < (uint64_t) kEventRing && c->ring_seq[slot] == next && c->ring[slot].pending) {
			const hipError_t q = hipEventQuery(c->ring[slot].stop);
			if (q != hipSuccess) { (void) hipGetLastError(); break; }
		}                                            // (an entry that was overwritten or harvested since belongs to a finished frame)
		c->completed_seq = next;
	}
}

// kLayoutRunDual, the product's rule (no history needed): which run copy the tiles of every block of workgroup tiles read, from the
// cube face the block's centre ray enters through.  All lanes of a wave start ON that face and march in lockstep, so a wave's samples
// lie in a plane parallel to it: the runs should lie IN that plane (lanes that are neighbours along the run axis then read the same
// 36-byte run, and the wave touches fewer lines per step) — entry through a z face -> runs along y, through a y face -> runs along z,
// through an x face -> runs along y.  Measured (scripts/gpu_r04_c.sh, full march, four oblique orthogonal poses, ms): this rule 2.76 /
// 2.88 / 2.82 / 2.99; the round-3 MEASURED per-block choice (four set-up frames per parameter set) 2.80 / 2.94 / 2.88 / 2.96; one copy
// for the whole frame 3.18-3.54; the other seven face -> copy tables 2.85-4.0 (x face -> runs along z: 2.85 / 2.86 / 3.05 / 2.95).
// vr_hip_set_brick_plane(6) keeps the measured choice as the validator.  Where a block's centre ray misses the cube other tiles of
// the block are tried; a block that misses everywhere keeps the frame's default.  Blocks are groups of 64 consecutive tile numbers =
// the 8x8-tile blocks of the tile numbering (larger groups for huge frames: at most 1024 bits travel in the kernel argument).
void dual_choice_bits(RayKernelArgs &a, const RaymarchPlan &plan, bool default_alt) {
	const vr_view &v = a.p.view;
	const uint32_t ntiles = plan.tiles_x * plan.tiles_y;
	uint32_t shift = 6;                                            // 64 consecutive tile numbers = one 8x8-tile block (tile_number_to_xy)
	while (((ntiles + (1u << shift) - 1u) >> shift) > 32u * kDualWords) shift++;
	const uint32_t groups = (ntiles + (1u << shift) - 1u) >> shift;
	const float tile_w = 32.0f, tile_h = 16.0f;                     // the 512-thread workgroup tile of the table-addressed variants
	memset(a.dual_bits, 0, sizeof a.dual_bits);
	// tuning aid: VR_DUAL_RULE = 3-bit mask, bit f set = tiles entered through a face of axis f (0 x, 1 y, 2 z) read the copy along y
	static const int rule = [] { const char *e = getenv("VR_DUAL_RULE"); return e ? atoi(e) : -1; }();
	auto entry_choice = [&](float px, float py, bool &alt) -> bool {          // pixel of the tile grid -> does its ray hit, and which copy
		const float lxf = std::fmin(std::fmax(px - (float) a.phase_x, 0.0f), (float) (a.p.out_width - 1u));
		const float lyf = std::fmin(std::fmax(py - (float) a.phase_y, 0.0f), (float) (a.p.out_rows - 1u));
		const uint32_t ly = (uint32_t) lyf, band = ly / a.p.band_rows;
		const uint32_t gy = (band * a.p.band_stride + a.p.band_first) * a.p.band_rows + (ly - band * a.p.band_rows);
		const float fx = (float) ((int) (a.p.x0 + (uint32_t) lxf) - (int) (v.width / 2u)), fy = (float) ((int) gy - (int) (v.height / 2u));
		float o[3], d[3], k_in[3], k_out[3];
		for (int i = 0; i < 3; i++) {
			o[i] = v.perspective ? v.origin[i] : v.origin[i] + v.right_plane[i] * fx + v.up_plane[i] * fy;
			d[i] = v.perspective ? v.direction[i] + v.right_plane[i] * fx + v.up_plane[i] * fy : v.direction[i];
			if (d[i] != 0.0f) { const float k1 = (-1.0f - o[i]) / d[i], k2 = (1.0f - o[i]) / d[i]; k_in[i] = std::fmin(k1, k2); k_out[i] = std::fmax(k1, k2); }
			else { k_in[i] = -3.0e38f; k_out[i] = std::fabs(o[i]) <= 1.0f ? 3.0e38f : -3.0e38f; }
		}
		const float kin = std::fmax(std::fmax(k_in[0], k_in[1]), k_in[2]), kout = std::fmin(std::fmin(k_out[0], k_out[1]), k_out[2]);
		if (!(kin < kout && kout > 0.0f)) return false;
		const int face = (k_in[2] >= k_in[0] && k_in[2] >= k_in[1]) ? 2 : (k_in[1] >= k_in[0] ? 1 : 0);
		alt = rule >= 0 ? ((rule >> face) & 1) != 0 : face != 1;
		return true;
	};
	for (uint32_t g = 0; g < groups; g++) {
		// the centre of the group's tiles first (tile 27 of 64 = column 3, row 3 of an 8x8 block: next to the block's centre), then a spread of the others
		const uint32_t first = g << shift, count = std::min(1u << shift, ntiles - first);
		static const uint32_t probe[9] = { 27, 36, 9, 14, 49, 54, 0, 63, 31 };
		bool alt = default_alt;
		for (int i = 0; i < 9; i++) {
			const uint32_t t = first + (uint32_t) (((uint64_t) probe[i] * count) >> 6);
			uint32_t tx = 0, ty = 0;
			tile_number_to_xy(t, plan.tiles_x, plan.tiles_y, &tx, &ty);
			const float cx = ((float) tx + (probe[i] == 27 ? 1.0f : 0.5f)) * tile_w, cy = ((float) ty + (probe[i] == 27 ? 1.0f : 0.5f)) * tile_h;
			bool a_i;
			if (entry_choice(cx, cy, a_i)) { alt = a_i; break; }
		}
		if (alt) a.dual_bits[g >> 5] |= 1u << (g & 31u);
	}
	a.dual_analytic = 1u; a.dual_shift = shift;
}

int launch_frame(vr_ctx *c, const vr_params *p, void *dev_rgba, hipStream_t stream) {
	RayKernelArgs a;
	memset(&a, 0, sizeof a);
	a.p = *p;
	a.dim_x = c->dim[0]; a.dim_y = c->dim[1]; a.dim_z = c->dim[2];
	a.stride_y = c->dim[0]; a.stride_z = (uint64_t) c->dim[0] * c->dim[1];
	a.half_x = 0.5f * (float) c->dim[0]; a.half_y = 0.5f * (float) c->dim[1]; a.half_z = 0.5f * (float) c->dim[2];
	a.off_x = a.half_x - 0.5f; a.off_y = a.half_y - 0.5f; a.off_z = a.half_z - 0.5f;
	a.max_x = (float) (c->dim[0] - 1); a.max_y = (float) (c->dim[1] - 1); a.max_z = (float) (c->dim[2] - 1);
	a.lh_x = 0.01f * a.half_x; a.lh_y = 0.01f * a.half_y; a.lh_z = 0.01f * a.half_z;
	a.tf_scale = c->bpv == 1 ? (float) VR_TF_SIZE / 255.0f : (float) VR_TF_SIZE / 65535.0f;
	a.kd_scaled = p->light_kd * (c->bpv == 1 ? (1.0f / 255.0f) : (1.0f / 65535.0f));
	a.tf_zero_below = c->tf_zero_below;
	{   // largest power of two P with fma(P, tf_scale, -0.5) <= tf_zero_below: raw < P implies a transparent sample
		uint32_t below = 0;
		const uint32_t top = c->bpv == 1 ? 256u : 65536u;
		for (uint32_t P = 1; P <= top; P <<= 1)
			if (c->tf_zero_below >= 0.0f && std::fmaf((float) P, a.tf_scale, -0.5f) <= c->tf_zero_below) below = P;
		a.skip_cmp = below == 0 ? 1u : 0u;
		if (below == 0) a.skip_mask = 0u;
		else if (c->bpv == 1) a.skip_mask = ((0xffu & ~(below - 1u)) * 0x01010101u);
		else a.skip_mask = ((0xffffu & ~(below - 1u)) * 0x00010001u);
	}
	{   // In-cube sample coordinates are exact to ~2^-23 * (1 + 2 max|origin|) * N/2 texels; the unclamped fetch (vr_kernels.hip)
		// needs them inside (-1, N), i.e. an error below 1/2.  Keep a factor-4 margin, else clamp every sample.
		const float omax = std::fmax(std::fabs(p->view.origin[0]), std::fmax(std::fabs(p->view.origin[1]), std::fabs(p->view.origin[2])));
		const float nmax = (float) std::max(c->dim[0], std::max(c->dim[1], c->dim[2]));
		a.clamp_fetch = (1.0f + 2.0f * omax) * nmax < 1048576.0f ? 0u : 1u;
		// intersect() replaces a direction component that is exactly 0 by 1e-5 (RaycasterBase.h:33-35): a ray whose origin lies
		// up to ky * 1e-5 outside a face of the cube is then still reported as a hit and marches at that constant out-of-cube
		// coordinate, ky * 1e-5 * N/2 texels beyond the face.  ky <= sqrt(3) * (omax + 1) for |direction| >= 1 (orthogonal: unit
		// direction; perspective: a unit vector plus in-plane offsets).  Keep that below 1/8 texel, else clamp every sample.
		if (1.7321f * (omax + 1.0f) * 1e-5f * nmax * 0.5f >= 0.125f) a.clamp_fetch = 1u;
		// The unclamped march fetches up to kDepth steps past a ray's exit point (software pipeline): that must stay inside the
		// kLutPad repeated edge entries of the address tables.  The reference's longest step is 1.666 cells (RaycasterBase.cpp:90).
		// A step moves a ray by ray_step * |direction component| * N/2 cells along an axis; perspective directions are not
		// normalised (direction + right * fx + up * fy, ViewBase.h:29-31), so the bound is taken over the whole frame.
		{
			const float half[3] = { a.half_x, a.half_y, a.half_z };
			float advance = 0.0f;
			for (int i = 0; i < 3; i++) {
				float d = std::fabs(p->view.direction[i]);
				if (p->view.perspective)
					d += std::fabs(p->view.right_plane[i]) * (0.5f * (float) p->view.width + 1.0f) + std::fabs(p->view.up_plane[i]) * (0.5f * (float) p->view.height + 1.0f);
				advance = std::fmax(advance, d * half[i]);
			}
			if (!((float) kOverrunSteps * p->ray_step * advance + 1.5f <= (float) kLutPad)) a.clamp_fetch = 1u;
		}
		if (c->force_clamp_fetch & 1u) a.clamp_fetch = 1u;
		auto pow2 = [](uint32_t n) { return n != 0 && (n & (n - 1)) == 0; };
		a.near_scaled = (pow2(c->dim[0]) && pow2(c->dim[1]) && pow2(c->dim[2]) && !(c->force_clamp_fetch & 2u)) ? 1u : 0u;
	}
	a.force_wide = c->force_wide;
	const bool bricked = c->layout == VR_LAYOUT_BRICKED;
	a.layout = bricked ? kLayoutBricked : kLayoutLinear;
	bool run_candidate = false, run_if_unaligned = false, dual_candidate = false;
	uint32_t run_layout = kLayoutRun;        // which run copy a run-brick frame reads: runs along z unless the view marches along z
	// Which brick copy: the one whose 16-byte chunks lie in the plane perpendicular to the view's dominant axis, so that the
	// pixels of a lane quad — neighbours on the screen — are neighbours inside a chunk (TRILINEAR; NEAREST keeps (x,y)).
	// Copies are built on first use (copy_for, further down); copy_possible asks without building.
	a.brick_plane = kPlaneXY;
	if (p->sampling != VR_SAMPLE_NEAREST && bricked && !c->force_wide) {
		uint32_t plane = kPlaneXY;
		if (c->brick_plane_force >= 0) plane = (uint32_t) c->brick_plane_force;
		else {
			// only for views (or, in perspective, central directions) along a volume axis: measured on the oblique benchmark pose the
			// (x,y) copy is as good as any (4.1 ms against 4.1 - 5.1 ms), along an axis the perpendicular plane wins by 8 - 16 %
			const float dx = std::fabs(p->view.direction[0] * a.half_x), dy = std::fabs(p->view.direction[1] * a.half_y),
			            dz = std::fabs(p->view.direction[2] * a.half_z);
			const float dmax = std::fmax(dx, std::fmax(dy, dz));
			if (dz >= dx && dz >= dy && copy_possible(c, kCopyRunY)) run_layout = kLayoutRunY;     // runs across the march, never along it
			if (dmax > 0.98f * std::sqrt(dx * dx + dy * dy + dz * dz)) {
				plane = dz >= dx && dz >= dy ? kPlaneXY : (dy >= dx ? kPlaneXZ : kPlaneYZ);
				// Perspective along x or y: the pixel pitch grows from 0.4 to 1.1 cells along the march, most lane quads straddle
				// chunks, and the run bricks — whose runs (along z) then lie across the march — are faster (measured 2.08 / 2.18 ms
				// against 2.49 / 2.51 ms on the benchmark poses).  Along z the runs lie along the march and the quad copy wins
				// (2.48 against 2.99 ms).  Orthogonal views along an axis are decided below, from how well their quads can be aligned.
				if (p->view.perspective && (plane != kPlaneXY || run_layout == kLayoutRunY)) run_candidate = true;
				else if (!p->view.perspective) run_if_unaligned = true;
			} else {
				run_candidate = true;       // not along an axis: lane quads straddle chunks whatever the plane -> one 8-byte gather
				// ... and which run copy is better depends on the cube face a tile's rays enter through (kLayoutRunDual).  Orthogonal
				// views only: on the oblique perspective pose the two copies are level almost everywhere (per-tile best of both -3 %)
				// and mixing them costs that much again (2.60 -> 2.62-2.64 ms); on the orthogonal one 3.20 -> 2.81 ms.
				dual_candidate = !p->view.perspective;
			}
		}
		if (plane == kPlanes + 3 || plane == kPlanes + 4) { run_candidate = true; dual_candidate = true; }     // forced: 6 = measured per-tile choice, 7 = alternating tiles
		if (plane < kPlanes && copy_possible(c, kCopyQuadXY + plane)) a.brick_plane = plane;
		else if (plane == kPlanes && copy_possible(c, kCopyRunZ)) a.layout = kLayoutRun;               // forced: 3 = runs along z, 4 = runs along y
		else if (plane == kPlanes + 1 && copy_possible(c, kCopyRunY)) a.layout = kLayoutRunY;
		if (run_candidate && copy_possible(c, run_layout == kLayoutRunY ? kCopyRunY : kCopyRunZ)) a.layout = run_layout;
	}
	// NEAREST: the voxel bricks (one voxel per element) unless a quad copy is forced (testing) or a 64-bit path is
	if (p->sampling == VR_SAMPLE_NEAREST && bricked && copy_possible(c, kCopyVoxel) && c->brick_plane_force < 0 && c->force_wide != 1)
		a.layout = kLayoutVoxel;
	// TRILINEAR with 2-byte voxels: the oct bricks (one 16-byte gather per sample instead of two 8-byte ones) for ORTHOGONAL views
	// with at most one cell per pixel — there the pixels of a lane quad share their elements and the halved gather count wins
	// (1024^3: 4.0 / 5.8 / 4.0 / 4.2 ms against 5.1 / 6.0 / 5.9 / 5.6 with the quad bricks; 2048^3: 28-44 against 40-48 ms); the
	// rays of a perspective view diverge to more than a cell per pixel, every lane then pulls 16 bytes it uses once, and the doubled
	// traffic loses (4.2-5.5 against 3.5-4.4 ms): those keep the quad bricks.  Not when a quad copy is forced (testing) or the
	// index-arithmetic path is.
	if (p->sampling != VR_SAMPLE_NEAREST && c->bpv == 2 && bricked && copy_possible(c, kCopyOct) && c->brick_plane_force < 0 && c->force_wide != 1 &&
	    (!p->view.perspective || c->oct_always)) {
		bool dense = true;
		const float half[3] = { a.half_x, a.half_y, a.half_z };
		for (int ax = 0; ax < 3; ax++)
			if ((std::fabs(p->view.right_plane[ax]) + std::fabs(p->view.up_plane[ax])) * half[ax] > 1.0f) dense = false;
		if (dense || c->oct_always) a.layout = kLayoutOct;
	}
	a.nbx = (c->dim[0] + kBrickEdge - 1) / kBrickEdge; a.nby = (c->dim[1] + kBrickEdge - 1) / kBrickEdge;
	a.nbz = (c->dim[2] + kBrickEdge - 1) / kBrickEdge;
	{   // RaycasterBase.h:59-63: index / esl_block_dims, prepared as shift or multiply-high
		const uint32_t bd = p->esl_block_dims ? p->esl_block_dims : 1u;
		if ((bd & (bd - 1)) == 0) { a.esl_div_magic = 0; a.esl_div_shift = (uint32_t) __builtin_ctz(bd); }
		else { a.esl_div_magic = (uint32_t) ((1ull << 32) / bd + 1); a.esl_div_shift = 0; }
	}

	vr_ctx::MapEntry *hit = nullptr;
	if (c->tile_lane_map >= 0) { a.lane_map = (uint32_t) c->tile_lane_map; a.phase_x = c->tile_phase_x; a.phase_y = c->tile_phase_y; }
	else {
		for (uint32_t i = 0; i < c->map_cached && hit == nullptr; i++)
			if (memcmp(&c->map_cache[i].p, p, sizeof *p) == 0 && memcmp(c->map_cache[i].dim, c->dim, sizeof c->dim) == 0) hit = &c->map_cache[i];
		if (hit == nullptr) {
			const uint32_t straddle = choose_tile_mapping(a);
			hit = &c->map_cache[c->map_next];
			hit->straddle_permille = straddle;
			// the recycled entry's measured copy choice belonged to other parameters; frames that read or record its buffers may still be
			// in flight on other streams — the validator is a testing aid, so it simply waits for them
			if (hit->dual_state != 0 || hit->order_tiles != 0) VR_TRY(c, drain(c));
			hit->dual_state = 0; hit->order_tiles = 0;
			constexpr uint32_t kMaps = sizeof c->map_cache / sizeof c->map_cache[0];
			c->map_next = (c->map_next + 1) % kMaps;
			if (c->map_cached < kMaps) c->map_cached++;
			hit->p = *p; memcpy(hit->dim, c->dim, sizeof c->dim);
			hit->lane_map = a.lane_map; hit->phase_x = a.phase_x; hit->phase_y = a.phase_y;
		}
		a.lane_map = hit->lane_map; a.phase_x = hit->phase_x; a.phase_y = hit->phase_y;
		// An orthogonal view along an axis whose pixels sit exactly on cell boundaries can carry rounding noise in its direction
		// (pose (180,90,0): components of 4e-8) that moves boundary pixels to the other neighbour part-way along the ray and
		// differently across the frame: no phase aligns it.  Measured on that pose: 3.18 ms with the run bricks against 3.71 ms.
		if (run_if_unaligned && hit->straddle_permille > 150u && copy_possible(c, run_layout == kLayoutRunY ? kCopyRunY : kCopyRunZ)) a.layout = run_layout;
	}
	// Per-tile choice between the two run copies (kLayoutRunDual, vr_device.h): full-march frames only (with leaping or early
	// termination the rays are short and the tile order is what matters), never for the clamping instantiation (its clamp bounds are
	// per axis, and the tile's y / z exchange would have to reach them).  The product's rule is ANALYTIC and needs no earlier frame:
	// every tile picks its copy in the kernel from the cube face the centre ray of its block of tiles enters through (raymarch_kernel)
	// — the first frame of a new view already reads the right copies.  vr_hip_set_brick_plane(6) keeps the round-3 MEASURED choice
	// (frames 0-3 of a parameter set on one copy each, the last two recording tile costs, choice kernel, per-block choice from frame 4 on)
	// as the validator of that rule; 7 = alternating tiles (testing: the copies meet at tile boundaries all over the frame).
	bool dual_analytic = false;
	int dual_stage = -1;                                 // measured / alternating choice only: -1 no; 0..3: the four frames before; 4: per-tile choice in use
	const bool dual_test = c->brick_plane_force == (int) kPlanes + 4, dual_measured = c->brick_plane_force == (int) kPlanes + 3;
	if (dual_candidate && is_run_layout(a.layout) && !p->esl && p->ray_threshold >= 1.0f && !a.clamp_fetch && c->bpv == 1 &&
	    copy_possible(c, kCopyRunZ) && copy_possible(c, kCopyRunY)) {
		if (dual_test || dual_measured) {
			if (hit != nullptr && c->tile_scheduling == 1 && !VR_ORDER_ALWAYS) {
				dual_stage = dual_test ? 4 : (int) hit->dual_state;
				a.layout = dual_stage == 4 ? kLayoutRunDual : ((dual_stage & 1) ? kLayoutRunY : kLayoutRun);
			}
		} else {
			dual_analytic = true;
			a.layout = kLayoutRunDual;
		}
	}
	// Column windows (kLayoutColumn, vr_device.h): full-march TRILINEAR frames of ORTHOGONAL views along a volume axis — every ray stays in
	// one cell column (at most one cell flip per lateral axis: the direction's other components are rounding noise), all rays share the k
	// sequence, and colmarch_kernel marches on wave-uniform state: one 16-byte gather and one transparency test per ~3 samples.  What the
	// host checks: the lateral drift over the longest possible segment stays below half a cell, a sample advances between 1/64 and 1 cell
	// along the axis (the kernel re-checks both per wave and otherwise marches per lane), at most one cell per pixel (the 32-bit offsets
	// of a wave's columns), no clamping instantiation.  Measured on the benchmark poses (full march, lit): 2.12 / 2.13 / 2.86 ms before.
	// NEAREST takes the same march over windows of 16 plain voxels (colmarch_nearest_kernel, kCopyColVox*): one gather per sixteen samples.
	// Early ray termination alone is fine (the kernels clear a terminated lane's live bit like the general one; the k sequence stays shared);
	// such frames give up the measured-cost tile order, which the column kernels do not take.
	static const bool col_ert = [] { const char *e = getenv("VR_COL_ERT"); return e == nullptr || atoi(e) != 0; }();      // VR_COL_ERT=0: A/B
	if (bricked && c->bpv == 1 && !c->force_wide && !p->view.perspective && !p->esl && (p->ray_threshold >= 1.0f || col_ert) &&
	    !a.clamp_fetch && c->column_force >= 0 && (c->brick_plane_force < 0 || c->column_force > 0)) {
		const float half[3] = { a.half_x, a.half_y, a.half_z };
		float d[3];
		for (int i = 0; i < 3; i++) d[i] = std::fabs(p->view.direction[i] * half[i]);
		const int m = d[2] >= d[0] && d[2] >= d[1] ? 2 : (d[1] >= d[0] ? 1 : 0);
		const float advance = d[m] * p->ray_step;
		bool take = advance >= 1.0f / 64.0f && advance <= 1.0f;
		for (int i = 0; i < 3; i++) if ((std::fabs(p->view.right_plane[i]) + std::fabs(p->view.up_plane[i])) * half[i] > 1.0f) take = false;
		if (c->column_force == 0)                                          // per view: along the axis — 4 > 2 sqrt(3), the longest segment in k
			for (int i = 0; i < 3; i++) if (i != m && d[i] * 4.0f >= 0.45f) take = false;
		if (take && copy_possible(c, (p->sampling == VR_SAMPLE_NEAREST ? kCopyColVoxX : kCopyColX) + (uint32_t) m)) {
			a.layout = kLayoutColumn; a.col_axis = (uint32_t) m; a.brick_plane = (uint32_t) m;
			// the dense path's constants, grouped for one scalar load each (vr_device.h); the products are the device's own fp32 products
			const bool nearest = p->sampling == VR_SAMPLE_NEAREST;
			a.col_sample.ax = p->view.direction[0] * a.half_x; a.col_sample.ay = p->view.direction[1] * a.half_y; a.col_sample.az = p->view.direction[2] * a.half_z;
			a.col_sample.tf_scale = a.tf_scale; a.col_sample.tf_zero_below = a.tf_zero_below;
			a.col_sample.max_x = a.max_x; a.col_sample.max_y = a.max_y; a.col_sample.max_z = a.max_z;
			a.col_sample.light_kd = p->light_kd; a.col_sample.ray_threshold = p->ray_threshold;
			for (int i = 0; i < 3; i++) { a.col_shade.dir[i] = p->view.direction[i]; a.col_shade.light[i] = p->view.light_pos[i]; }
			a.col_shade.lh[0] = a.lh_x; a.col_shade.lh[1] = a.lh_y; a.col_shade.lh[2] = a.lh_z;
			a.col_shade.kd_scaled = a.kd_scaled;
			for (int i = 0; i < 3; i++) a.col_shade.dim[i] = c->dim[i];
			a.col_shade.nbu = col_blocks(c->dim[m == 0 ? 1 : 0]);
			a.col_shade.nw = col_windows(c->dim[m], nearest ? kColVoxCells : kColCells);
			dual_analytic = false; dual_stage = -1;
		}
	}
	// The copy this frame reads, built now if this is its first use.  A build that is refused (HBM guard, allocation, linear array
	// released) degrades to the next best resident copy; the image is the same.
	const void *brick_copy = nullptr;
	if (a.layout == kLayoutRunDual) {
		brick_copy = copy_for(c, kCopyRunZ);
		a.alt_copy = (uint64_t) (uintptr_t) copy_for(c, kCopyRunY);
		if (brick_copy == nullptr || a.alt_copy == 0) { a.layout = run_layout; dual_stage = -1; dual_analytic = false; brick_copy = nullptr; }
	}
	if (a.layout != kLayoutLinear && brick_copy == nullptr) {
		const uint32_t want = a.layout == kLayoutRun ? kCopyRunZ : a.layout == kLayoutRunY ? kCopyRunY : a.layout == kLayoutVoxel ? kCopyVoxel :
		                      a.layout == kLayoutOct ? kCopyOct : a.layout == kLayoutColumn ? (p->sampling == VR_SAMPLE_NEAREST ? kCopyColVoxX : kCopyColX) + a.col_axis : kCopyQuadXY + a.brick_plane;
		brick_copy = copy_for(c, want);
		if (brick_copy == nullptr && want != kCopyQuadXY) {
			a.layout = kLayoutBricked; a.brick_plane = kPlaneXY;
			c->map_cached = 0; c->map_next = 0;              // lane orders were chosen for the copy that could not be had
			brick_copy = copy_for(c, kCopyQuadXY);
		}
		if (brick_copy == nullptr) {                         // no quad copy either: any resident copy this sampling mode can read
			if (p->sampling == VR_SAMPLE_NEAREST && c->copy[kCopyVoxel] && c->force_wide != 1) { a.layout = kLayoutVoxel; brick_copy = c->copy[kCopyVoxel]; }
			else if (p->sampling != VR_SAMPLE_NEAREST && !c->force_wide && c->copy[kCopyRunZ]) { a.layout = kLayoutRun; brick_copy = c->copy[kCopyRunZ]; }
			else if (p->sampling != VR_SAMPLE_NEAREST && !c->force_wide && c->copy[kCopyRunY]) { a.layout = kLayoutRunY; brick_copy = c->copy[kCopyRunY]; }
			else if (p->sampling != VR_SAMPLE_NEAREST && c->bpv == 2 && c->force_wide != 1 && c->copy[kCopyOct]) { a.layout = kLayoutOct; brick_copy = c->copy[kCopyOct]; }
			else a.layout = kLayoutLinear;
		}
	}
	// Shape of a wave's pixel tile (8x8, 16x4 or 4x16 inside the 32x16-pixel workgroup tile): a perspective view along a volume axis
	// that reads run bricks gets its waves elongated along the screen direction the RUNS map to — the lanes of a wave then share the
	// 36-byte runs (and their cache lines) instead of spreading over twice as many cell columns.  Measured on the benchmark's
	// perspective poses: 2.07 -> 1.96 (along z, runs along y: tall), 2.04 -> 1.95 (along y, runs along z: tall), 2.05 -> 1.96 ms
	// (along x, runs along z: wide); oblique views and orthogonal views lose 3-9 % with either elongated shape and keep 8x8.
	if (c->tile_lane_map < 0 && is_run_layout(a.layout) && p->view.perspective && !p->esl) {       // (with leaping the rays are short: no difference measured)
		const float dx = std::fabs(p->view.direction[0] * a.half_x), dy = std::fabs(p->view.direction[1] * a.half_y), dz = std::fabs(p->view.direction[2] * a.half_z);
		if (std::fmax(dx, std::fmax(dy, dz)) > 0.98f * std::sqrt(dx * dx + dy * dy + dz * dz)) {
			const int run_axis = a.layout == kLayoutRunY ? 1 : 2;
			const float across = std::fabs(p->view.right_plane[run_axis]), down = std::fabs(p->view.up_plane[run_axis]);
			a.lane_map = (a.lane_map & 3u) | ((down > across ? 2u : 1u) << 2);
		}
	}
	// NEAREST on voxel bricks, perspective view along x or y: the four lanes of a quad are four pixels along the screen direction that
	// maps to z — four different slices, i.e. four different lines — not a 2x2-pixel block: byte gathers whose lanes fall into the same
	// dwords serialise (measured on the benchmark poses: 1.44 -> 1.38 ms along y, 1.46 -> 1.38 ms along x; along z the blocks stay: 1.23).
	if (c->tile_lane_map < 0 && a.layout == kLayoutVoxel && p->view.perspective && (a.lane_map & 3u) == kLaneBlocks) {
		const float dx = std::fabs(p->view.direction[0] * a.half_x), dy = std::fabs(p->view.direction[1] * a.half_y), dz = std::fabs(p->view.direction[2] * a.half_z);
		if (dz < dx || dz < dy) a.lane_map = std::fabs(p->view.up_plane[2]) > std::fabs(p->view.right_plane[2]) ? kLaneColumns : kLaneRows;
	}
	const RaymarchPlan plan = plan_raymarch(a, brick_copy != nullptr, c->bpv);
	if (plan.reads_linear && c->vol == nullptr)
		return fail(c, VR_ERR_NOT_READY, "this frame needs the linear array, which was released (vr_hip_release_linear_copy): no resident brick copy "
		                                 "serves this sampling mode / addressing path — prepare it before releasing, or set the volume again");

	// Measured-cost launch order: only where rays differ in length (empty-space leaping or early termination on) — the full
	// march has no tail to remove and keeps its cache-friendly tile numbering.
	const uint32_t ntiles = plan.tiles_x * plan.tiles_y;
	if (dual_analytic && a.layout == kLayoutRunDual) dual_choice_bits(a, plan, run_layout == kLayoutRunY);
	TileSchedule sched;
	vr_ctx::SchedSlot *read_slot = nullptr, *record_slot = nullptr;
	vr_ctx::OrderEntry *oe = nullptr;
	if (c->tile_scheduling == 1 && c->tile_lane_map < 0 && (VR_ORDER_ALWAYS || p->esl || p->ray_threshold < 1.0f) && ntiles >= 64 && ntiles <= (1u << 20)) {
		advance_completed(c);
		vr_ctx::OrderKey key;
		memset(&key, 0, sizeof key);
		key.sampling = p->sampling; key.esl = p->esl ? 1u : 0u; key.ert = p->ray_threshold < 1.0f ? 1u : 0u; key.perspective = p->view.perspective ? 1u : 0u;
		{
			const float dx = std::fabs(p->view.direction[0] * a.half_x), dy = std::fabs(p->view.direction[1] * a.half_y), dz = std::fabs(p->view.direction[2] * a.half_z);
			const int m = dz >= dx && dz >= dy ? 2 : (dy >= dx ? 1 : 0);
			key.major_axis = (uint32_t) m * 2u + (p->view.direction[m] < 0.0f ? 1u : 0u);
			// ... and the direction itself in steps of 1/8 per component (views within a few degrees share their costs, different poses do not)
			const float len = std::sqrt(p->view.direction[0] * p->view.direction[0] + p->view.direction[1] * p->view.direction[1] + p->view.direction[2] * p->view.direction[2]);
			for (int i = 0; i < 3; i++) key.direction_q[i] = len > 0.0f ? (int32_t) std::lrint(p->view.direction[i] / len * 8.0f) : 0;
		}
		key.layout = a.layout; key.view_w = p->view.width; key.view_h = p->view.height; key.x0 = p->x0; key.out_width = p->out_width; key.out_rows = p->out_rows;
		key.band_rows = p->band_rows; key.band_stride = p->band_stride; key.band_first = p->band_first;
		memcpy(key.dim, c->dim, sizeof key.dim); key.tiles_x = plan.tiles_x; key.tiles_y = plan.tiles_y;
		vr_ctx::OrderEntry *lru = &c->order_cache[0];        // an unused entry if there is one, else the least recently used
		for (auto &e : c->order_cache) {
			if (e.used && memcmp(&e.key, &key, sizeof key) == 0) { oe = &e; break; }
			if (lru->used && (!e.used || e.lru < lru->lru)) lru = &e;
		}
		if (oe == nullptr) {                         // new policy key: take the least recently used entry; its slots keep their buffers and their last-use marks
			oe = lru;
			oe->used = true; oe->key = key; oe->has_last = false; oe->repeats = 0;
			for (auto &sl : oe->slot) sl.valid = false;
		}
		oe->lru = c->seq_next;
		// the order this frame launches in: the most recently issued one that has FINISHED (or was issued on this very stream, which orders it
		// before this frame); only when there is none, one still being built on another stream, behind a device-side wait
		vr_ctx::SchedSlot *pending_elsewhere = nullptr;
		for (auto &sl : oe->slot) {
			if (!sl.valid || sl.ntiles != ntiles) continue;
			bool usable = sl.stream == stream;
			if (!usable) { const hipError_t q = hipEventQuery(sl.ready); if (q == hipSuccess) usable = true; else (void) hipGetLastError(); }
			if (usable) { if (read_slot == nullptr || sl.issue_seq > read_slot->issue_seq) read_slot = &sl; }
			else if (pending_elsewhere == nullptr || sl.issue_seq > pending_elsewhere->issue_seq) pending_elsewhere = &sl;
		}
		if (read_slot == nullptr && pending_elsewhere != nullptr) {
			VR_TRY(c, hipStreamWaitEvent(stream, pending_elsewhere->ready, 0));
			read_slot = pending_elsewhere;
		}
		// No order at all — the first frame under this policy key (a new view: the reference's benchmark renders every view once): predict
		// the tile costs from the ESL bit volume (tile_estimate_kernel) and order by them, on this stream, in front of the frame
		static const bool estimate = [] { const char *e = getenv("VR_TILE_ESTIMATE"); return e == nullptr || atoi(e) != 0; }();      // VR_TILE_ESTIMATE=0: A/B
		if (read_slot == nullptr && estimate && p->esl && a.layout != kLayoutColumn) {
			vr_ctx::SchedSlot *est = nullptr;
			for (auto &sl : oe->slot) {
				if (sl.last_seq > c->completed_seq) continue;                                          // a frame that may still run reads or records it
				if (sl.ready != nullptr && sl.issue_seq != 0) { const hipError_t q = hipEventQuery(sl.ready); if (q != hipSuccess) { (void) hipGetLastError(); continue; } }
				est = &sl; break;
			}
			if (est != nullptr && est->capacity < ntiles) {
				if (est->cost) { (void) hipFree(est->cost); (void) hipFree(est->order); est->cost = est->order = nullptr; est->capacity = 0; }
				if (hipMalloc((void **) &est->cost, (size_t) ntiles * 4) == hipSuccess && hipMalloc((void **) &est->order, (size_t) ntiles * 4) == hipSuccess) est->capacity = ntiles;
				else { (void) hipGetLastError(); if (est->cost) (void) hipFree(est->cost); est->cost = est->order = nullptr; est = nullptr; }
			}
			if (est != nullptr) {
				if (est->ready == nullptr) VR_TRY(c, hipEventCreateWithFlags(&est->ready, hipEventDisableTiming));
				RayKernelArgs ea = a;
				ea.tiles_x = plan.tiles_x; ea.tiles_y = plan.tiles_y;
				VR_TRY(c, launch_tile_estimate(ea, plan.tile_h, c->esl, est->cost, ntiles, stream));
				VR_TRY(c, launch_tile_order(est->cost, est->order, ntiles, stream));
				VR_TRY(c, hipEventRecord(est->ready, stream));
				est->valid = true; est->ntiles = ntiles; est->stream = stream; est->issue_seq = c->seq_next;
				read_slot = est;
			}
		}
		if (read_slot != nullptr) sched.order = read_slot->order;
		// record this frame's tile costs?  Always while the parameters keep changing (a moving camera: the next frame's order comes from
		// this one), twice for a frame that is repeated unchanged (the first recording of a view may have run on cold caches)
		const bool same = oe->has_last && memcmp(&oe->last, p, sizeof *p) == 0;
		if (!same) { oe->last = *p; oe->has_last = true; oe->repeats = 0; }
		if (oe->repeats < 2u) {
			for (auto &sl : oe->slot) {
				if (&sl == read_slot || sl.last_seq > c->completed_seq) continue;                     // in use by this frame / by a frame that may still run
				if (sl.ready != nullptr && sl.issue_seq != 0) { const hipError_t q = hipEventQuery(sl.ready); if (q != hipSuccess) { (void) hipGetLastError(); continue; } }   // its order kernel has not run yet
				if (record_slot == nullptr || (!sl.valid && record_slot->valid) || (sl.valid == record_slot->valid && sl.issue_seq < record_slot->issue_seq)) record_slot = &sl;
			}
			if (record_slot != nullptr && record_slot->capacity < ntiles) {                        // (free: nothing of it is in flight)
				if (record_slot->cost) { (void) hipFree(record_slot->cost); (void) hipFree(record_slot->order); record_slot->cost = record_slot->order = nullptr; record_slot->capacity = 0; }
				if (hipMalloc((void **) &record_slot->cost, (size_t) ntiles * 4) == hipSuccess && hipMalloc((void **) &record_slot->order, (size_t) ntiles * 4) == hipSuccess) record_slot->capacity = ntiles;
				else { (void) hipGetLastError(); if (record_slot->cost) (void) hipFree(record_slot->cost); record_slot->cost = record_slot->order = nullptr; record_slot = nullptr; }
			}
			if (record_slot != nullptr) {
				if (record_slot->ready == nullptr) VR_TRY(c, hipEventCreateWithFlags(&record_slot->ready, hipEventDisableTiming));
				record_slot->valid = false;
				VR_TRY(c, hipMemsetAsync(record_slot->cost, 0, (size_t) ntiles * 4, stream));
				sched.cost = record_slot->cost;
			}
		}
	}

	// kLayoutRunDual: frames 2 and 3 record, frame 3 is followed by the choice kernel, frame 4 onwards reads the choice.  Every step
	// is ordered behind the one before through the entry's event when it runs on another stream.
	bool dual_advance = false;
	if (dual_stage >= 0) {
		if (ntiles < 2u || ntiles > (1u << 20)) dual_stage = -1;      // (a frame that keeps kLayoutRunDual without a choice reads the copy along z)
		else if (hit->capacity < ntiles) {
			if (hit->cost) { VR_TRY(c, drain(c)); (void) hipFree(hit->cost); (void) hipFree(hit->order); hit->cost = hit->order = nullptr; hit->capacity = 0; }
			if (hipMalloc((void **) &hit->cost, (size_t) ntiles * 4) == hipSuccess && hipMalloc((void **) &hit->order, (size_t) ntiles * 8) == hipSuccess) hit->capacity = ntiles;
			else { (void) hipGetLastError(); if (hit->cost) (void) hipFree(hit->cost); hit->cost = hit->order = nullptr; dual_stage = -1; }
			if (!dual_test) { hit->dual_state = 0; if (dual_stage > 0) dual_stage = -1; }      // (cannot happen: the grid of a parameter set does not change)
		}
	}
	if (dual_stage >= 0) {
		if (hit->order_ready == nullptr) VR_TRY(c, hipEventCreateWithFlags(&hit->order_ready, hipEventDisableTiming));
		if (dual_stage > 0 && !(dual_test && hit->order_tiles != ntiles) && hit->order_stream != nullptr && stream != hit->order_stream)
			VR_TRY(c, hipStreamWaitEvent(stream, hit->order_ready, 0));
		if (dual_test) {                                   // alternating tiles, built once per grid size
			if (hit->order_tiles != ntiles) {
				VR_TRY(c, launch_tile_choice(nullptr, nullptr, hit->order, ntiles, stream));
				VR_TRY(c, hipEventRecord(hit->order_ready, stream));
				hit->order_tiles = ntiles; hit->order_stream = stream;
			}
			sched.order = hit->order;
		} else if (dual_stage == 2 || dual_stage == 3) {
			uint32_t *into = dual_stage == 2 ? hit->cost : hit->order + hit->capacity;
			VR_TRY(c, hipMemsetAsync(into, 0, (size_t) ntiles * 4, stream));
			sched.cost = into;
			dual_advance = true;
		} else if (dual_stage == 4) sched.order = hit->order;
		else dual_advance = true;
	}

	if (c->tile_scheduling == 2 && ntiles <= (1u << 20)) {     // profiling: the cost of every tile of this frame, tile = workgroup id
		if (c->cost_map_capacity < ntiles) {
			VR_TRY(c, drain(c));
			if (c->cost_map) (void) hipFree(c->cost_map);
			c->cost_map = nullptr; c->cost_map_capacity = 0;
			VR_TRY(c, hipMalloc((void **) &c->cost_map, (size_t) ntiles * 4));
			c->cost_map_capacity = ntiles;
		}
		VR_TRY(c, hipMemsetAsync(c->cost_map, 0, (size_t) ntiles * 4, stream));
		sched.cost = c->cost_map;
		c->cost_map_tiles_x = plan.tiles_x; c->cost_map_tiles_y = plan.tiles_y;
	}

	c->last_launch = vr_launch_info{ a.layout, a.brick_plane, a.lane_map, a.phase_x, a.phase_y, a.clamp_fetch, plan.tiles_x, plan.tiles_y,
	                                 sched.order != nullptr ? 1u : 0u, hit != nullptr ? hit->straddle_permille : 1000u };

#ifdef VR_BOUNDS_CHECK
	{   // debug build: what the frame's gathers must stay inside, and where the first violation is recorded
		if (c->bc_fault == nullptr) { VR_TRY(c, hipMalloc((void **) &c->bc_fault, 8 * sizeof(uint32_t))); VR_TRY(c, hipMemset(c->bc_fault, 0, 8 * sizeof(uint32_t))); }
		const void *array = plan.reads_linear ? c->vol : brick_copy;
		a.bc_base = (uint64_t) (uintptr_t) array - (a.layout == kLayoutColumn ? kColPadBytes : 0u);
		a.bc_bytes = plan.reads_linear ? (c->vol_elems + volume_tail_slack(c->dim[0], c->dim[1])) * c->bpv :
		             copy_bytes(c, a.layout == kLayoutRun || a.layout == kLayoutRunDual ? kCopyRunZ : a.layout == kLayoutRunY ? kCopyRunY : a.layout == kLayoutVoxel ? kCopyVoxel :
		                           a.layout == kLayoutOct ? kCopyOct : a.layout == kLayoutColumn ? (p->sampling == VR_SAMPLE_NEAREST ? kCopyColVoxX : kCopyColX) + a.col_axis : kCopyQuadXY + a.brick_plane);
		if (a.layout == kLayoutColumn) a.bc_bytes += 2ull * kColPadBytes;
		a.bc_alt_bytes = a.alt_copy ? copy_bytes(c, kCopyRunY) : 0;
		a.bc_fault = c->bc_fault; a.bc_ntiles = ntiles;
		// self-test of the net itself: VR_BC_SELFTEST=1 halves the size the checks hold the gathers against — a full-march frame must then fail
		if (const char *e = getenv("VR_BC_SELFTEST")) if (atoi(e) == 1) a.bc_bytes /= 2;
	}
#endif
	EventPair &ev = c->ring[c->ring_head];
	const uint64_t frame_seq = c->seq_next++;
	c->ring_seq[c->ring_head] = frame_seq;
	c->ring_head = (c->ring_head + 1) % kEventRing;
	harvest(c, ev);                              // only blocks if 256 launches are still in flight
	VR_TRY(c, hipEventRecord(ev.start, stream));
	VR_TRY(c, launch_raymarch(a, c->vol, brick_copy, c->bpv, c->tf, c->esl, dev_rgba, sched, stream));
	VR_TRY(c, hipEventRecord(ev.stop, stream));
	ev.pending = true;
#ifdef VR_BOUNDS_CHECK
	{
		uint32_t fault[6] = { 0, 0, 0, 0, 0, 0 };
		VR_TRY(c, hipStreamSynchronize(stream));
		VR_TRY(c, hipMemcpy(fault, c->bc_fault, sizeof fault, hipMemcpyDeviceToHost));
		if (fault[0] != 0u) {
			char msg[256];
			snprintf(msg, sizeof msg, "bounds check: code %u (1 table index [axis << 8], 2 offset, 3 address, 4 cost slot) workgroup %u thread %u value 0x%08x%08x limit %u, layout %u",
			         fault[0], fault[1], fault[2], fault[4], fault[3], fault[5], a.layout);
			(void) hipMemset(c->bc_fault, 0, 8 * sizeof(uint32_t));
			return fail(c, VR_ERR_HIP, msg);
		}
	}
#endif
	if (dual_advance) {                          // behind the frame, on its stream
		if (dual_stage == 3) VR_TRY(c, launch_tile_choice(hit->cost, hit->order + hit->capacity, hit->order, ntiles, stream));
		VR_TRY(c, hipEventRecord(hit->order_ready, stream));
		hit->order_stream = stream; hit->order_tiles = ntiles;
		hit->dual_state = (uint32_t) dual_stage + 1u;
	}
	if (read_slot != nullptr) read_slot->last_seq = frame_seq;
	if (record_slot != nullptr) {                // behind the frame, on its stream: the order for the next frame under this policy key
		VR_TRY(c, launch_tile_order(record_slot->cost, record_slot->order, ntiles, stream));
		VR_TRY(c, hipEventRecord(record_slot->ready, stream));
		record_slot->valid = true; record_slot->ntiles = ntiles; record_slot->stream = stream; record_slot->issue_seq = record_slot->last_seq = frame_seq;
		oe->repeats++;
	}
	return VR_OK;
}

int ready(vr_ctx *c) {
	bool any = c->vol != nullptr;
	for (uint32_t k = 0; k < kCopyKinds; k++) any = any || c->copy[k] != nullptr;
	if (!any) return fail(c, VR_ERR_NOT_READY, "render before set_volume");
	if (!c->tf_set) return fail(c, VR_ERR_NOT_READY, "render before set_transfer_fn");
	return VR_OK;
}

// Waits for every frame this context has queued (on its own or on callers' streams) and for its own stream: what the reference's
// synchronous set_* calls imply, without stalling other contexts or streams of the device (no hipDeviceSynchronize).
hipError_t drain(vr_ctx *c) {
	for (int i = 0; i < kEventRing; i++)
		if (c->ring[i].pending) { hipError_t e = hipEventSynchronize(c->ring[i].stop); if (e != hipSuccess) return e; }
	return hipStreamSynchronize(c->stream);
}

void free_bricks(vr_ctx *c) {
	for (uint32_t k = 0; k < kCopyKinds; k++) {
		if (c->copy[k]) { (void) hipFree(k >= kCopyColX && k <= kCopyColVoxZ ? (uint8_t *) c->copy[k] - kColPadBytes : (uint8_t *) c->copy[k]); c->copy[k] = nullptr; }
		c->copy_build_ms[k] = 0; c->copy_failed[k] = false;
	}
}

uint32_t max_dim_of(const vr_ctx *c) { return std::max(c->dim[0], std::max(c->dim[1], c->dim[2])); }

uint64_t copy_bytes(const vr_ctx *c, uint32_t kind) {
	if (kind <= kCopyQuadYZ) return bricked_elems(c->dim[0], c->dim[1], c->dim[2]) * 4 * c->bpv;
	if (kind == kCopyVoxel) return bricked_elems(c->dim[0], c->dim[1], c->dim[2]) * c->bpv;
	if (kind == kCopyOct) return bricked_elems(c->dim[0], c->dim[1], c->dim[2]) * 8 * c->bpv;
	if (kind >= kCopyColX && kind <= kCopyColZ) return col_copy_bytes(c->dim, kind - kCopyColX);
	if (kind >= kCopyColVoxX && kind <= kCopyColVoxZ) return col_copy_bytes(c->dim, kind - kCopyColVoxX, true);
	return run_copy_bytes(c->dim[0], c->dim[1], c->dim[2]);
}

// does the layout policy have this copy at this volume size at all?  (quad (x,y): always; the other planes and the run bricks:
// 1-byte voxels, edges the 32-bit tables cover; voxel bricks: wherever the address tables reach)
bool copy_in_policy(const vr_ctx *c, uint32_t kind) {
	if (c->layout != VR_LAYOUT_BRICKED || c->dim[0] == 0) return false;
	if (kind == kCopyQuadXY) return true;
	if (kind == kCopyVoxel) return max_dim_of(c) <= 2048u;
	if (kind == kCopyOct) return c->bpv == 2 && max_dim_of(c) <= 2048u;
	if (kind >= kCopyColX && kind <= kCopyColVoxZ) return c->bpv == 1 && max_dim_of(c) <= 2048u;
	if (kind == kCopyQuadXZ || kind == kCopyQuadYZ) return c->bpv == 1 && max_dim_of(c) <= 1024u && copy_bytes(c, kind) <= (1ull << 32);
	return c->bpv == 1 && max_dim_of(c) <= 1024u;
}

// resident, or buildable on first use (the policy has it, the linear array is still there, no earlier build was refused)
bool copy_possible(const vr_ctx *c, uint32_t kind) {
	if (c->copy[kind]) return true;
	return copy_in_policy(c, kind) && c->vol != nullptr && !c->copy_failed[kind];
}

// Builds copy `kind` from the linear array on the context's stream and waits for it (a one-time stall of the first frame that
// wants it: 6-8 ms per copy at 1024^3).  Every copy but the first quad copy is only built while half of the HBM stays free.
int build_copy(vr_ctx *c, uint32_t kind) {
	if (c->copy[kind]) return VR_OK;
	if (!copy_possible(c, kind)) return VR_ERR_NOT_READY;
	const uint64_t bytes = copy_bytes(c, kind);
	// the half-of-the-HBM rule exempts the ONE copy a sampling mode cannot do without: the first quad copy — for 2-byte voxels the oct
	// copy, which TRILINEAR reads instead (building both is left to a caller who forces the quad copy: testing)
	if (kind != kCopyQuadXY && !(kind == kCopyOct && c->copy[kCopyQuadXY] == nullptr)) {
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes >= free_b || free_b - bytes < total_b / 2) { (void) hipGetLastError(); c->copy_failed[kind] = true; return VR_ERR_ALLOC; }
	}
	void *dst = nullptr;
	const bool column = kind >= kCopyColX && kind <= kCopyColVoxZ;       // column windows: zeroed padding in front and behind (vr_device.h kColPadBytes)
	if (hipMalloc(&dst, bytes + (column ? 2ull * kColPadBytes : 0ull)) != hipSuccess) { (void) hipGetLastError(); c->copy_failed[kind] = true; return fail(c, VR_ERR_ALLOC, "brick copy allocation failed"); }
	hipError_t e = hipSuccess;
	if (column) {
		e = hipMemsetAsync(dst, 0, kColPadBytes, c->stream);
		if (e == hipSuccess) e = hipMemsetAsync((uint8_t *) dst + kColPadBytes + bytes, 0, kColPadBytes, c->stream);
		dst = (uint8_t *) dst + kColPadBytes;                            // what kernels and downloads see; free_bricks undoes the offset
	}
	if (e == hipSuccess) e = hipEventRecord(c->aux_start, c->stream);
	if (e == hipSuccess) {
		if (kind <= kCopyQuadYZ) e = launch_brickify(c->vol, dst, c->bpv, kind - kCopyQuadXY, c->dim[0], c->dim[1], c->dim[2], c->stream);
		else if (kind == kCopyVoxel) e = launch_brickify_voxel(c->vol, dst, c->bpv, c->dim[0], c->dim[1], c->dim[2], c->stream);
		else if (kind == kCopyOct) e = launch_brickify_oct(c->vol, dst, c->dim[0], c->dim[1], c->dim[2], c->stream);
		else if (kind >= kCopyColX && kind <= kCopyColZ) e = launch_build_column(c->vol, dst, kind - kCopyColX, false, c->dim[0], c->dim[1], c->dim[2], c->stream);
		else if (kind >= kCopyColVoxX && kind <= kCopyColVoxZ) e = launch_build_column(c->vol, dst, kind - kCopyColVoxX, true, c->dim[0], c->dim[1], c->dim[2], c->stream);
		else e = launch_brickify_run(c->vol, dst, kind == kCopyRunY ? kLayoutRunY : kLayoutRun, c->dim[0], c->dim[1], c->dim[2], c->stream);
	}
	if (e == hipSuccess) e = hipEventRecord(c->aux_stop, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (e != hipSuccess) { (void) hipGetLastError(); (void) hipFree(column ? (uint8_t *) dst - kColPadBytes : (uint8_t *) dst); c->copy_failed[kind] = true; return fail(c, VR_ERR_HIP, "brick copy build failed", e); }
	(void) hipEventElapsedTime(&c->copy_build_ms[kind], c->aux_start, c->aux_stop);
	c->copy[kind] = dst;
	return VR_OK;
}

const void *copy_for(vr_ctx *c, uint32_t kind) {
	if (c->copy[kind] == nullptr) (void) build_copy(c, kind);
	return c->copy[kind];
}

int alloc_volume(vr_ctx *c, uint32_t x, uint32_t y, uint32_t z, uint32_t bpv) {
	if (x == 0 || y == 0 || z == 0 || x > 65535u || y > 65535u || z > 65535u)       // Model::dims is ushort3
		return fail(c, VR_ERR_INVALID, "volume dims out of range (1..65535)");
	if (bpv != 1 && bpv != 2) return fail(c, VR_ERR_INVALID, "bytes_per_voxel must be 1 or 2");
	VR_TRY(c, drain(c));                         // frames still reading the volume we are about to free
	if (c->vol) { (void) hipFree(c->vol); c->vol = nullptr; }
	free_bricks(c);
	c->map_cached = 0; c->map_next = 0;          // cached tile mappings belong to the previous volume
	c->dim[0] = c->dim[1] = c->dim[2] = 0;
	const uint64_t elems = (uint64_t) x * y * z;
	const uint64_t slack = volume_tail_slack(x, y);
	VR_TRY(c, hipMalloc(&c->vol, (elems + slack) * bpv));
	VR_TRY(c, hipMemsetAsync((uint8_t *) c->vol + elems * bpv, 0, slack * bpv, c->stream));
	c->vol_elems = elems; c->dim[0] = x; c->dim[1] = y; c->dim[2] = z; c->bpv = bpv;
	return VR_OK;
}

}  // namespace

extern "C" {

const char *vr_hip_version(void) { return "vr_hip 0.1 (gfx950)"; }

int vr_hip_create(int device, vr_ctx **out) {
	if (out == nullptr) return VR_ERR_INVALID;
	*out = nullptr;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) {
		(void) hipGetLastError();
		return VR_ERR_NO_DEVICE;                 // no CPU fallback: the product path needs an MI355X
	}
	vr_ctx *c = new (std::nothrow) vr_ctx();
	if (c == nullptr) return VR_ERR_ALLOC;
	c->device = device;
	*out = c;                                    // handed out even on failure below so last_error stays readable
	VR_TRY(c, hipSetDevice(device));
	VR_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	{
		int least = 0, greatest = 0;
		VR_TRY(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
		VR_TRY(c, hipStreamCreateWithPriority(&c->stream_first, hipStreamNonBlocking, greatest));
	}
	VR_TRY(c, hipMalloc((void **) &c->tf, VR_TF_SIZE * 4 * sizeof(float)));
	VR_TRY(c, hipMalloc((void **) &c->esl, VR_ESL_VOLUME_SIZE * sizeof(uint32_t)));
	VR_TRY(c, hipMalloc((void **) &c->minmax, 32 * 32 * 32 * 2));
	VR_TRY(c, hipMalloc((void **) &c->hist, 256 * sizeof(unsigned long long)));
	for (int i = 0; i < kEventRing; i++) {
		VR_TRY(c, hipEventCreate(&c->ring[i].start));
		VR_TRY(c, hipEventCreate(&c->ring[i].stop));
	}
	VR_TRY(c, hipEventCreate(&c->aux_start));
	VR_TRY(c, hipEventCreate(&c->aux_stop));
	return VR_OK;
}

void vr_hip_destroy(vr_ctx *c) {
	if (c == nullptr) return;
	(void) hipSetDevice(c->device);
	if (c->stream) (void) hipStreamSynchronize(c->stream);
	for (int i = 0; i < kEventRing; i++) {
		if (c->ring[i].start) (void) hipEventDestroy(c->ring[i].start);
		if (c->ring[i].stop) (void) hipEventDestroy(c->ring[i].stop);
	}
	if (c->aux_start) (void) hipEventDestroy(c->aux_start);
	if (c->aux_stop) (void) hipEventDestroy(c->aux_stop);
	if (c->cost_map) (void) hipFree(c->cost_map);
#ifdef VR_BOUNDS_CHECK
	if (c->bc_fault) (void) hipFree(c->bc_fault);
#endif
	for (auto &e : c->map_cache) { if (e.cost) (void) hipFree(e.cost); if (e.order) (void) hipFree(e.order); if (e.order_ready) (void) hipEventDestroy(e.order_ready); }
	for (auto &e : c->order_cache) for (auto &sl : e.slot) { if (sl.cost) (void) hipFree(sl.cost); if (sl.order) (void) hipFree(sl.order); if (sl.ready) (void) hipEventDestroy(sl.ready); }
	if (c->fb) (void) hipFree(c->fb);
	if (c->tf) (void) hipFree(c->tf);
	if (c->esl) (void) hipFree(c->esl);
	if (c->vol) (void) hipFree(c->vol);
	free_bricks(c);
	if (c->minmax) (void) hipFree(c->minmax);
	if (c->hist) (void) hipFree(c->hist);
	if (c->stream_first) { (void) hipStreamSynchronize(c->stream_first); (void) hipStreamDestroy(c->stream_first); }
	if (c->stream) (void) hipStreamDestroy(c->stream);
	delete c;
}

const char *vr_hip_last_error(const vr_ctx *c) { return c ? c->err.c_str() : "no context"; }

int vr_hip_set_window(vr_ctx *c, uint32_t w, uint32_t h) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (w == 0 || h == 0 || w > 65535u || h > 65535u) return fail(c, VR_ERR_INVALID, "window dims out of range (1..65535)");
	VR_TRY(c, hipSetDevice(c->device));
	const size_t bytes = (size_t) w * h * 4;
	if (bytes != c->fb_bytes) {
		if (c->fb) { (void) hipFree(c->fb); c->fb = nullptr; c->fb_bytes = 0; }
		VR_TRY(c, hipMalloc(&c->fb, bytes));
		c->fb_bytes = bytes;
	}
	c->win_w = w; c->win_h = h;
	return VR_OK;
}

int vr_hip_set_transfer_fn(vr_ctx *c, const float *tf, const uint32_t *esl) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (tf == nullptr || esl == nullptr) return fail(c, VR_ERR_INVALID, "transfer_fn / esl_volume is NULL");
	VR_TRY(c, hipSetDevice(c->device));
	// Frames queued by vr_hip_render_device run asynchronously (context stream or a caller's stream): wait for THIS CONTEXT'S
	// frames (the event ring records every one of them) before the tables they read are rewritten, then upload on the context
	// stream and wait again — the reference is synchronous here too, and calls this on every mouse-motion event of its TF editor
	// (UI.cpp:52-61,317-341), so other contexts and streams of the device are not stalled (no hipDeviceSynchronize).
	VR_TRY(c, drain(c));
	VR_TRY(c, hipMemcpyAsync(c->tf, tf, VR_TF_SIZE * 4 * sizeof(float), hipMemcpyHostToDevice, c->stream));
	VR_TRY(c, hipMemcpyAsync(c->esl, esl, VR_ESL_VOLUME_SIZE * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	VR_TRY(c, hipStreamSynchronize(c->stream));
	int zero = -1;
	while (zero + 1 < VR_TF_SIZE && tf[4 * (zero + 1)] == 0.0f && tf[4 * (zero + 1) + 1] == 0.0f && tf[4 * (zero + 1) + 2] == 0.0f &&
	       tf[4 * (zero + 1) + 3] == 0.0f)
		zero++;
	c->tf_zero_below = (float) zero;
	c->tf_set = true;
	// tile orders recorded under the old transfer function describe other ray lengths (placement only; nothing is in flight here: drained above)
	for (auto &e : c->map_cache) { e.dual_state = 0; e.order_tiles = 0; }
	for (auto &e : c->order_cache) { e.has_last = false; e.repeats = 0; for (auto &sl : e.slot) sl.valid = false; }
	return VR_OK;
}

int vr_hip_set_volume(vr_ctx *c, const void *host, uint32_t x, uint32_t y, uint32_t z, uint32_t bpv) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (host == nullptr) return fail(c, VR_ERR_INVALID, "volume data is NULL");     // GPURenderer1.cu:91-92
	VR_TRY(c, hipSetDevice(c->device));
	int rc = alloc_volume(c, x, y, z, bpv);
	if (rc) return rc;
	const auto t0 = std::chrono::steady_clock::now();
	VR_TRY(c, hipMemcpy(c->vol, host, c->vol_elems * bpv, hipMemcpyHostToDevice));
	VR_TRY(c, hipStreamSynchronize(c->stream));
	c->upload_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
	return VR_OK;                                // brick copies are built by the first frame that reads them (or vr_hip_prepare)
}

int vr_hip_set_volume_device(vr_ctx *c, const void *dev, uint32_t x, uint32_t y, uint32_t z, uint32_t bpv) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (dev == nullptr) return fail(c, VR_ERR_INVALID, "volume data is NULL");
	VR_TRY(c, hipSetDevice(c->device));
	int rc = alloc_volume(c, x, y, z, bpv);
	if (rc) return rc;
	// on the context's own (non-blocking) stream: a device-to-device hipMemcpy on the null stream may return before it has
	// finished and would not be ordered before the brick builder below
	const auto t0 = std::chrono::steady_clock::now();
	VR_TRY(c, hipMemcpyAsync(c->vol, dev, c->vol_elems * bpv, hipMemcpyDeviceToDevice, c->stream));
	VR_TRY(c, hipStreamSynchronize(c->stream));
	c->upload_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
	return VR_OK;
}

int vr_hip_set_layout(vr_ctx *c, uint32_t layout) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (layout != VR_LAYOUT_LINEAR && layout != VR_LAYOUT_BRICKED) return fail(c, VR_ERR_INVALID, "unknown volume layout");
	VR_TRY(c, hipSetDevice(c->device));
	VR_TRY(c, drain(c));                         // a frame may still be reading the copies we are about to drop
	if (c->vol == nullptr && c->dim[0] != 0)
		return fail(c, VR_ERR_NOT_READY, "the linear copy was released (vr_hip_release_linear_copy): the brick copies cannot be rebuilt or dropped");
	c->layout = layout;
	free_bricks(c);
	c->map_cached = 0; c->map_next = 0;
	return VR_OK;
}

int vr_hip_set_wide_addressing(vr_ctx *c, uint32_t force) {
	if (c == nullptr) return VR_ERR_INVALID;
	// The index-arithmetic path (1) reads the linear array for NEAREST: refused once that array was released (the frame
	// would otherwise be launched with a NULL volume pointer).  launch_frame checks the same for every frame.
	if ((force & 3u) == 1u && c->vol == nullptr && c->dim[0] != 0)
		return fail(c, VR_ERR_NOT_READY, "the linear copy was released (vr_hip_release_linear_copy): the index-arithmetic path needs it");
	c->force_wide = force & 3u;                  // 0 auto, 1 arithmetic 64-bit path, 2 table path with 64-bit z offsets
	c->force_clamp_fetch = (force >> 2) & 3u;    // + 4: clamp the fetch coordinates of every sample (far-away views do that);
	                                             // + 8: NEAREST never marches in the scaled domain (volumes with power-of-two edges do)
	return VR_OK;
}

int vr_hip_set_brick_plane(vr_ctx *c, int32_t plane) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (plane == (int32_t) kPlanes + 5 || plane == (int32_t) kPlanes + 6) {      // 8 / 9: the column windows for every orthogonal full-march frame / never
		c->column_force = plane == (int32_t) kPlanes + 5 ? 1 : -1;
		c->oct_always = false; c->brick_plane_force = -1;
		c->map_cached = 0; c->map_next = 0;
		return VR_OK;
	}
	c->column_force = 0;
	if (plane < -1 || plane > (int32_t) kPlanes + 4)
		return fail(c, VR_ERR_INVALID, "plane must be -1 (per view), 0 (x,y), 1 (x,z), 2 (y,z), 3 (run bricks along z), 4 (run bricks along y), 5 (oct bricks for every view of a "
		                               "2-byte volume), 6 (both run copies, chosen per tile by measurement, for every full-march view), 7 (both run copies on alternating tiles), "
		                               "8 (column windows for every orthogonal full-march frame) or 9 (never the column windows)");
	c->oct_always = plane == (int32_t) kPlanes + 2;
	c->brick_plane_force = c->oct_always ? -1 : plane;
	c->map_cached = 0; c->map_next = 0;          // cached lane orders were chosen for another plane
	return VR_OK;
}

int vr_hip_set_tile_scheduling(vr_ctx *c, uint32_t mode) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (mode > 2u) return fail(c, VR_ERR_INVALID, "tile scheduling mode must be 0 (workgroup id), 1 (measured-cost order) or 2 (workgroup id + cost map)");
	c->tile_scheduling = mode;
	for (auto &e : c->map_cache) { e.dual_state = 0; e.order_tiles = 0; }
	for (auto &e : c->order_cache) { e.has_last = false; e.repeats = 0; for (auto &sl : e.slot) sl.valid = false; }      // (buffers and last-use marks stay: recycled only when free)
	return VR_OK;
}

int vr_hip_last_launch(vr_ctx *c, vr_launch_info *out) {
	if (c == nullptr || out == nullptr) return VR_ERR_INVALID;
	*out = c->last_launch;
	return VR_OK;
}

int vr_hip_read_tile_costs(vr_ctx *c, uint32_t *host_out, uint32_t capacity, uint32_t *tiles_x, uint32_t *tiles_y) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (tiles_x) *tiles_x = c->cost_map_tiles_x;
	if (tiles_y) *tiles_y = c->cost_map_tiles_y;
	const uint32_t n = c->cost_map_tiles_x * c->cost_map_tiles_y;
	if (c->cost_map == nullptr || n == 0) return fail(c, VR_ERR_NOT_READY, "no cost map: render a frame with vr_hip_set_tile_scheduling(ctx, 2) first");
	if (host_out == nullptr) return VR_OK;                   // size query
	if (capacity < n) return fail(c, VR_ERR_INVALID, "cost map buffer too small");
	VR_TRY(c, hipSetDevice(c->device));
	VR_TRY(c, drain(c));
	std::vector<uint32_t> by_number(n);
	VR_TRY(c, hipMemcpy(by_number.data(), c->cost_map, (size_t) n * 4, hipMemcpyDeviceToHost));
	for (uint32_t t = 0; t < n; t++) {                        // tile number (blocks of tiles) -> row-major
		uint32_t x = 0, y = 0;
		tile_number_to_xy(t, c->cost_map_tiles_x, c->cost_map_tiles_y, &x, &y);
		host_out[(size_t) y * c->cost_map_tiles_x + x] = by_number[t];
	}
	return VR_OK;
}

int vr_hip_set_tile_mapping(vr_ctx *c, int32_t lane_map, uint32_t phase_x, uint32_t phase_y) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (lane_map < -1 || (lane_map >= 0 && ((lane_map & 3) > (int32_t) kLaneBlocks || (lane_map >> 2) > 2)) || phase_x > 7u || phase_y > 7u)
		return fail(c, VR_ERR_INVALID, "lane_map must be -1 or (order 0..2) + 4 * (wave shape 0..2), and the phases 0..7");
	c->tile_lane_map = lane_map; c->tile_phase_x = phase_x; c->tile_phase_y = phase_y;
	return VR_OK;
}

int vr_hip_render_device(vr_ctx *c, const vr_params *p, void *dev_rgba, void *stream) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (dev_rgba == nullptr) return fail(c, VR_ERR_INVALID, "buffer is NULL");      // GPURenderer1.cu:101-102
	int rc = validate_params(c, p);
	if (rc) return rc;
	rc = ready(c);
	if (rc) return rc;
	VR_TRY(c, hipSetDevice(c->device));
	return launch_frame(c, p, dev_rgba, stream ? (hipStream_t) stream : c->stream);
}

int vr_hip_render(vr_ctx *c, const vr_params *p, uint8_t *host_rgba) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (host_rgba == nullptr) return fail(c, VR_ERR_INVALID, "buffer is NULL");
	int rc = validate_params(c, p);
	if (rc) return rc;
	rc = ready(c);
	if (rc) return rc;
	const size_t bytes = (size_t) p->out_width * p->out_rows * 4;
	if (c->fb == nullptr || bytes > c->fb_bytes)
		return fail(c, VR_ERR_NOT_READY, "output larger than the window buffer: call vr_hip_set_window first");
	VR_TRY(c, hipSetDevice(c->device));
	const auto t0 = std::chrono::steady_clock::now();
	// The reference's timed region ends with the frame in HOST memory (GPURenderer1.cu:107-110).  A frame of unpartitioned rows is
	// rendered as TWO row slices: the first on a stream of the highest priority, so that its workgroups are dispatched ahead of the
	// second slice's and it finishes about halfway through the frame; its copy to the host then runs while the second slice still
	// renders, and only the second half of the PCIe transfer is left behind the kernels (measured on the prototype,
	// scripts/host_slices_probe.py, C4: 2.32 ms per frame against 2.48 as one launch + one copy, the kernel alone 2.28 wall).  The
	// pixels are the same: a slice is a band partition (x0 / band fields of vr_params), the partition the multi-GPU path renders.
	static const int host_slices = [] { const char *e = getenv("VR_HOST_SLICES"); return e ? atoi(e) : 2; }();         // VR_HOST_SLICES=1: one launch + one copy (A/B)
	const uint32_t first_rows = ((p->out_rows + 1u) / 2u + 15u) & ~15u;      // whole workgroup tiles (32x16 pixels)
	if (host_slices >= 2 && p->band_stride == 1 && p->band_first == 0 && first_rows < p->out_rows && bytes >= (1u << 20)) {
		vr_params q = *p;
		q.band_rows = first_rows; q.band_stride = 2;                        // gy = band_first * first_rows + ly for ly < first_rows
		q.band_first = 0; q.out_rows = first_rows;
		rc = launch_frame(c, &q, c->fb, c->stream_first);
		if (rc) return rc;
		const size_t first_bytes = (size_t) first_rows * p->out_width * 4;
		q.band_first = 1; q.out_rows = p->out_rows - first_rows;
		rc = launch_frame(c, &q, (uint8_t *) c->fb + first_bytes, c->stream);
		if (rc) { (void) hipStreamSynchronize(c->stream_first); return rc; }
		VR_TRY(c, hipMemcpyAsync(host_rgba, c->fb, first_bytes, hipMemcpyDeviceToHost, c->stream_first));
		VR_TRY(c, hipMemcpyAsync(host_rgba + first_bytes, (uint8_t *) c->fb + first_bytes, bytes - first_bytes, hipMemcpyDeviceToHost, c->stream));
		VR_TRY(c, hipStreamSynchronize(c->stream_first));
		VR_TRY(c, hipStreamSynchronize(c->stream));
	} else {
		rc = launch_frame(c, p, c->fb, c->stream);
		if (rc) return rc;
		VR_TRY(c, hipMemcpyAsync(host_rgba, c->fb, bytes, hipMemcpyDeviceToHost, c->stream));
		VR_TRY(c, hipStreamSynchronize(c->stream));
	}
	const auto t1 = std::chrono::steady_clock::now();
	c->last_total_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
	if (c->last_total_ms > c->total_ms_max) c->total_ms_max = c->last_total_ms;
	return VR_OK;
}

int vr_hip_timing(vr_ctx *c, vr_timing *out) {
	if (c == nullptr || out == nullptr) return VR_ERR_INVALID;
	(void) hipSetDevice(c->device);
	for (int i = 0; i < kEventRing; i++)
		harvest(c, c->ring[(c->ring_head + i) % kEventRing]);    // oldest first
	out->kernel_ms = c->last_kernel_ms;
	out->total_ms = c->last_total_ms > 0 ? c->last_total_ms : c->last_kernel_ms;
	out->launches = c->launches;
	out->kernel_ms_sum = c->kernel_ms_sum;
	out->kernel_ms_max = c->kernel_ms_max;
	out->total_ms_max = c->total_ms_max;
	return VR_OK;
}

int vr_hip_timing_reset(vr_ctx *c) {
	if (c == nullptr) return VR_ERR_INVALID;
	vr_timing t;
	(void) vr_hip_timing(c, &t);
	c->launches = 0; c->kernel_ms_sum = 0; c->last_kernel_ms = 0; c->last_total_ms = 0; c->kernel_ms_max = 0; c->total_ms_max = 0;
	return VR_OK;
}

int vr_hip_volume_minmax(vr_ctx *c, uint8_t *minmax_out, uint32_t *bd_out, float *bs_out, float *kernel_ms_out) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (minmax_out == nullptr) return fail(c, VR_ERR_INVALID, "minmax_out is NULL");
	if (c->vol == nullptr) return fail(c, VR_ERR_NOT_READY, c->dim[0] ? "the linear copy was released (vr_hip_release_linear_copy): set the volume again" : "minmax before set_volume");
	VR_TRY(c, hipSetDevice(c->device));
	// RaycasterBase.cpp:97-99
	uint32_t max_dim = c->dim[0] > c->dim[1] ? c->dim[0] : c->dim[1];
	if (c->dim[2] > max_dim) max_dim = c->dim[2];
	uint32_t bd = (max_dim + VR_ESL_VOLUME_DIMS - 1) / VR_ESL_VOLUME_DIMS;
	if (bd < VR_ESL_MIN_BLOCK) bd = VR_ESL_MIN_BLOCK;
	VR_TRY(c, hipEventRecord(c->aux_start, c->stream));
	VR_TRY(c, launch_minmax(c->vol, c->bpv, c->dim[0], c->dim[1], c->dim[2], bd, c->minmax, c->stream));
	VR_TRY(c, hipEventRecord(c->aux_stop, c->stream));
	VR_TRY(c, hipMemcpyAsync(minmax_out, c->minmax, 32 * 32 * 32 * 2, hipMemcpyDeviceToHost, c->stream));
	VR_TRY(c, hipStreamSynchronize(c->stream));
	if (kernel_ms_out) VR_TRY(c, hipEventElapsedTime(kernel_ms_out, c->aux_start, c->aux_stop));
	if (bd_out) *bd_out = bd;
	if (bs_out) {                                // RaycasterBase.cpp:118-122
		bs_out[0] = 2.0f * (float) bd / (float) c->dim[0];
		bs_out[1] = 2.0f * (float) bd / (float) c->dim[1];
		bs_out[2] = 2.0f * (float) bd / (float) c->dim[2];
	}
	return VR_OK;
}

int vr_hip_volume_histogram(vr_ctx *c, uint64_t *hist_out, float *kernel_ms_out) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (hist_out == nullptr) return fail(c, VR_ERR_INVALID, "hist_out is NULL");
	if (c->vol == nullptr) return fail(c, VR_ERR_NOT_READY, c->dim[0] ? "the linear copy was released (vr_hip_release_linear_copy): set the volume again" : "histogram before set_volume");
	VR_TRY(c, hipSetDevice(c->device));
	VR_TRY(c, hipEventRecord(c->aux_start, c->stream));
	VR_TRY(c, launch_histogram(c->vol, c->bpv, c->vol_elems, c->hist, c->stream));
	VR_TRY(c, hipEventRecord(c->aux_stop, c->stream));
	VR_TRY(c, hipMemcpyAsync(hist_out, c->hist, 256 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
	VR_TRY(c, hipStreamSynchronize(c->stream));
	if (kernel_ms_out) VR_TRY(c, hipEventElapsedTime(kernel_ms_out, c->aux_start, c->aux_stop));
	return VR_OK;
}

int vr_hip_generate_volume(vr_ctx *c, uint32_t kind, uint32_t n, uint32_t seed, uint32_t bpv) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (kind > 1) return fail(c, VR_ERR_INVALID, "unknown synthetic volume kind");
	VR_TRY(c, hipSetDevice(c->device));
	int rc = alloc_volume(c, n, n, n, bpv);
	if (rc) return rc;
	const auto t0 = std::chrono::steady_clock::now();
	VR_TRY(c, launch_generate(c->vol, kind, n, seed, bpv, c->stream));
	VR_TRY(c, hipStreamSynchronize(c->stream));
	c->upload_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
	return VR_OK;
}

int vr_hip_download_volume(vr_ctx *c, void *host_out, uint64_t bytes) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (host_out == nullptr) return fail(c, VR_ERR_INVALID, "host_out is NULL");
	if (c->vol == nullptr) return fail(c, VR_ERR_NOT_READY, c->dim[0] ? "the linear copy was released (vr_hip_release_linear_copy): set the volume again" : "download before set_volume");
	if (bytes != c->vol_elems * c->bpv) return fail(c, VR_ERR_INVALID, "byte count does not match the resident volume");
	VR_TRY(c, hipSetDevice(c->device));
	VR_TRY(c, hipMemcpy(host_out, c->vol, bytes, hipMemcpyDeviceToHost));
	return VR_OK;
}

int vr_hip_volume_info(vr_ctx *c, vr_volume_info *out) {
	if (c == nullptr || out == nullptr) return VR_ERR_INVALID;
	memset(out, 0, sizeof *out);
	if (c->dim[0] == 0) return fail(c, VR_ERR_NOT_READY, "volume_info before set_volume");
	out->dim_x = c->dim[0]; out->dim_y = c->dim[1]; out->dim_z = c->dim[2]; out->bytes_per_voxel = c->bpv;
	out->layout = c->layout;
	out->linear_resident = c->vol != nullptr ? 1u : 0u;
	out->linear_bytes = c->vol != nullptr ? (c->vol_elems + volume_tail_slack(c->dim[0], c->dim[1])) * c->bpv : 0;
	for (uint32_t k = 0; k < kCopyKinds; k++) {
		out->build_ms[k] = c->copy_build_ms[k];
		if (copy_in_policy(c, k)) out->copies_in_policy |= 1u << k;
		if (c->copy_failed[k]) out->copies_refused |= 1u << k;
		if (c->copy[k] == nullptr) continue;
		out->copies |= 1u << k;
		out->bricked_bytes += copy_bytes(c, k);
		if (k <= kCopyQuadYZ) { out->brick_planes |= 1u << k; out->brick_copies++; }
	}
	if (c->copy[kCopyRunZ]) out->run_copy |= 1u;
	if (c->copy[kCopyRunY]) out->run_copy |= 2u;
	if (c->copy[kCopyVoxel]) out->run_copy |= 4u;
	out->brick_copies_wanted = (c->layout == VR_LAYOUT_BRICKED) ? (copy_in_policy(c, kCopyQuadXZ) ? 3u : 1u) : 0u;
	out->upload_ms = c->upload_ms;
	return VR_OK;
}

int vr_hip_prepare(vr_ctx *c, uint32_t copies) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (c->dim[0] == 0) return fail(c, VR_ERR_NOT_READY, "prepare before set_volume");
	if (copies & ~((1u << kCopyKinds) - 1u)) return fail(c, VR_ERR_INVALID, "unknown copy bits");
	VR_TRY(c, hipSetDevice(c->device));
	int worst = VR_OK;
	for (uint32_t k = 0; k < kCopyKinds; k++) {
		if (!((copies >> k) & 1u) || c->copy[k] || !copy_in_policy(c, k)) continue;      // copies the policy does not have at this size are skipped
		if (c->vol == nullptr) return fail(c, VR_ERR_NOT_READY, "the linear copy was released (vr_hip_release_linear_copy): set the volume again");
		c->copy_failed[k] = false;                   // an explicit request retries an earlier refusal
		const int rc = build_copy(c, k);
		if (rc != VR_OK && worst == VR_OK) { worst = rc; if (c->err.empty() || rc == VR_ERR_ALLOC) fail(c, rc, "a brick copy was not built (less than half of the HBM would stay free, or allocation failed)"); }
	}
	if (copies) { c->map_cached = 0; c->map_next = 0; }
	return worst;
}

int vr_hip_download_copy(vr_ctx *c, uint32_t kind, void *host_out, uint64_t capacity, uint64_t *bytes_out) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (kind >= kCopyKinds) return fail(c, VR_ERR_INVALID, "unknown copy kind");
	if (c->copy[kind] == nullptr) return fail(c, VR_ERR_NOT_READY, "this brick copy is not resident (render a frame that reads it, or vr_hip_prepare)");
	uint64_t bytes = copy_bytes(c, kind);
	if (kind == kCopyRunZ || kind == kCopyRunY) bytes -= 16;     // the allocation's tail slack is not part of the copy
	if (bytes_out) *bytes_out = bytes;
	if (host_out == nullptr) return VR_OK;
	if (capacity < bytes) return fail(c, VR_ERR_INVALID, "buffer too small for this copy");
	VR_TRY(c, hipSetDevice(c->device));
	VR_TRY(c, drain(c));
	VR_TRY(c, hipMemcpy(host_out, c->copy[kind], bytes, hipMemcpyDeviceToHost));
	return VR_OK;
}

int vr_hip_release_linear_copy(vr_ctx *c) {
	if (c == nullptr) return VR_ERR_INVALID;
	if (c->dim[0] == 0) return fail(c, VR_ERR_NOT_READY, "release_linear_copy before set_volume");
	if (c->vol == nullptr) return VR_OK;
	bool any = false;
	for (uint32_t k = 0; k < kCopyKinds; k++) any = any || c->copy[k] != nullptr;
	if (!any || max_dim_of(c) > 2048u)
		return fail(c, VR_ERR_INVALID, "the linear array is the only copy a render path can read (no brick copy resident: render a frame or call "
		                               "vr_hip_prepare first; linear layout; or an edge above 2048)");
	if (c->force_wide == 1u)
		return fail(c, VR_ERR_INVALID, "the index-arithmetic path (vr_hip_set_wide_addressing 1) reads the linear array");
	VR_TRY(c, hipSetDevice(c->device));
	VR_TRY(c, drain(c));
	(void) hipFree(c->vol);
	c->vol = nullptr;
	return VR_OK;
}

int vr_hip_device_info(vr_ctx *c, char *name_out, size_t name_cap, uint32_t *cus, uint64_t *hbm_bytes) {
	if (c == nullptr) return VR_ERR_INVALID;
	hipDeviceProp_t prop;
	VR_TRY(c, hipGetDeviceProperties(&prop, c->device));
	if (name_out && name_cap) { snprintf(name_out, name_cap, "%s (%s)", prop.name, prop.gcnArchName); }
	if (cus) *cus = (uint32_t) prop.multiProcessorCount;
	if (hbm_bytes) *hbm_bytes = (uint64_t) prop.totalGlobalMem;
	return VR_OK;
}

}  // extern "C"
