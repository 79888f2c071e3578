// RaycasterBase.cpp — feeders of the ray-march path, restated from the reference's behaviour
// (VolumeRendering/RaycasterBase.cpp); see RaycasterBase.h.
#include "RaycasterBase.h"

#include <string.h>

namespace volr {

namespace {
const int kBlocks = ESL_VOLUME_DIMS * ESL_VOLUME_DIMS * ESL_VOLUME_DIMS;

template <typename T> T clamp_to(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }

int largest_dim(const Model &m) {
	int d = m.dims.x > m.dims.y ? m.dims.x : m.dims.y;
	return d > m.dims.z ? d : m.dims.z;
}
}  // namespace

float4 RaycasterBase::tf_storage[TF_SIZE];
esl_type RaycasterBase::esl_storage[ESL_VOLUME_SIZE];
unsigned char RaycasterBase::esl_min_max[kBlocks * 2];
float4 RaycasterBase::base_transfer_fn[TF_SIZE];
float2 RaycasterBase::ray_step_limits = { 0.0f, 0.0f };

// Defaults of RaycasterBase.cpp:9-20: ray_step 0.06, threshold 0.95, ESL on, block 8, light_kd 0.6
Raycaster RaycasterBase::raycaster = {
	{ nullptr, 0, { 0, 0, 0 }, { -1.0f, -1.0f, -1.0f } },
	{ { 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, false },
	RaycasterBase::tf_storage,
	0.06f,
	0.95f,
	true,
	RaycasterBase::esl_storage,
	ESL_MIN_BLOCK_SIZE,
	{ 0, 0, 0 },
	0.6f
};

// RaycasterBase.cpp:26-44 — clamped setters (reset: absolute value, otherwise increment)
void RaycasterBase::change_ray_step(float step, bool reset) {
	raycaster.ray_step = clamp_to(reset ? step : raycaster.ray_step + step, ray_step_limits.x, ray_step_limits.y);
}

void RaycasterBase::change_ray_threshold(float threshold, bool reset) {
	raycaster.ray_threshold = clamp_to(reset ? threshold : raycaster.ray_threshold + threshold, 0.5f, 1.0f);
}

void RaycasterBase::change_light_intensity(float intensity, bool reset) {
	raycaster.light_kd = clamp_to(reset ? intensity : raycaster.light_kd + intensity, 0.0f, 2.0f);
}

void RaycasterBase::toggle_esl() {
	raycaster.esl = !raycaster.esl;
}

void RaycasterBase::set_view(View view) {
	raycaster.view = view;
}

// RaycasterBase.cpp:46-74: premultiply colour by opacity; then flag every ESL block whose whole [min,max] value range
// maps to zero opacity.  first_visible[x] = first TF index >= x with non-zero alpha; a block is empty iff
// first_visible[min / TF_RATIO] > max / TF_RATIO.  Bit (i % 32) of word (i / 32), i = z*1024 + y*32 + x.
void RaycasterBase::update_transfer_fn() {
	for (int i = 0; i < TF_SIZE; i++) {
		const float4 b = base_transfer_fn[i];
		raycaster.transfer_fn[i] = make_float4(b.x * b.w, b.y * b.w, b.z * b.w, b.w);
	}
	unsigned short first_visible[TF_SIZE];
	int next = TF_SIZE;                              // scan from the top: O(n) instead of the reference's O(n^2) loop
	for (int x = TF_SIZE - 1; x >= 0; x--) {
		if (raycaster.transfer_fn[x].w != 0)
			next = x;
		first_visible[x] = (unsigned short) next;
	}
	memset(raycaster.esl_volume, 0, ESL_VOLUME_SIZE * sizeof(esl_type));
	for (int i = 0; i < kBlocks; i++) {
		const unsigned mn = esl_min_max[2 * i], mx = esl_min_max[2 * i + 1];
		if (first_visible[mn / TF_RATIO] > mx / TF_RATIO)
			raycaster.esl_volume[i >> 5] |= 1u << (i & 31);
	}
}

// RaycasterBase.cpp:76-84: default TF — three consecutive colour ramps (r, g, b thirds), alpha = i/128 above the
// 10 % noise floor ((255 * 0.1) / TF_RATIO = 12.75), then update_transfer_fn()
void RaycasterBase::reset_transfer_fn() {
	const int third = TF_SIZE / 3;
	for (int i = 0; i < TF_SIZE; i++) {
		float r = 0.0f, g = 0.0f, b = 0.0f;
		if (i <= third)              r = (i * 3) / (float) TF_SIZE;
		else if (i <= third * 2)     g = ((i - third) * 3) / (float) TF_SIZE;
		else                         b = ((i - third * 2) * 3) / (float) TF_SIZE;
		const float a = i > ((255.0f * 0.1f) / TF_RATIO) ? i / (float) TF_SIZE : 0.0f;
		base_transfer_fn[i] = make_float4(r, g, b, a);
	}
	update_transfer_fn();
}

// RaycasterBase.cpp:86-92: one sample per voxel of the longest axis, minus one voxel's worth
void RaycasterBase::reset_ray_step() {
	const int max_dim = largest_dim(raycaster.volume);
	raycaster.ray_step = 2.0f / max_dim;
	raycaster.ray_step -= raycaster.ray_step / max_dim;
	ray_step_limits.x = raycaster.ray_step / 3;
	ray_step_limits.y = raycaster.ray_step * 1.666f;
}

// RaycasterBase.cpp:97-99,118-122
void RaycasterBase::set_block_geometry(Model volume) {
	raycaster.volume = volume;
	const int max_dim = largest_dim(volume);
	int bd = (max_dim + ESL_VOLUME_DIMS - 1) / ESL_VOLUME_DIMS;
	if (bd < ESL_MIN_BLOCK_SIZE) bd = ESL_MIN_BLOCK_SIZE;
	raycaster.esl_block_dims = (unsigned short) bd;
	raycaster.esl_block_size = make_float3(2.0f * raycaster.esl_block_dims / volume.dims.x,
	                                       2.0f * raycaster.esl_block_dims / volume.dims.y,
	                                       2.0f * raycaster.esl_block_dims / volume.dims.z);
}

// RaycasterBase.cpp:94-125: serial per-block min/max scan over the host voxels (blocks never touched keep {255, 0}
// and are therefore flagged empty), then TF/ESL update and ray-step reset
void RaycasterBase::set_volume(Model volume) {
	set_block_geometry(volume);
	const unsigned bd = raycaster.esl_block_dims;
	for (int i = 0; i < kBlocks; i++) { esl_min_max[2 * i] = 255; esl_min_max[2 * i + 1] = 0; }
	for (unsigned z = 0; z < volume.dims.z; z++)
		for (unsigned y = 0; y < volume.dims.y; y++) {
			const unsigned char *row = volume.data + ((size_t) z * volume.dims.y + y) * volume.dims.x;
			unsigned char *blocks = esl_min_max + 2 * ((z / bd) * ESL_VOLUME_DIMS * ESL_VOLUME_DIMS + (y / bd) * ESL_VOLUME_DIMS);
			for (unsigned x = 0; x < volume.dims.x; x++) {
				unsigned char *mm = blocks + 2 * (x / bd);
				if (mm[0] > row[x]) mm[0] = row[x];
				if (mm[1] < row[x]) mm[1] = row[x];
			}
		}
	update_transfer_fn();
	reset_ray_step();
}

void RaycasterBase::set_volume(Model volume, const unsigned char *minmax_pairs) {
	set_block_geometry(volume);
	memcpy(esl_min_max, minmax_pairs, sizeof esl_min_max);
	update_transfer_fn();
	reset_ray_step();
}

const unsigned char *RaycasterBase::block_min_max() { return esl_min_max; }

}  // namespace volr
