// RaycasterBase.h — host mirror of the reference's raycaster manager (VolumeRendering/RaycasterBase.h:102-118,
// RaycasterBase.cpp): owner of the global Raycaster parameter block and of the FEEDERS of the ray-march path — the
// premultiplied transfer function, the empty-space-leaping (ESL) bit-volume, the ESL block geometry and the default
// ray step.  These must reproduce the reference bit for bit, otherwise the renderer is fed different inputs.
#pragma once

#include "Renderer.h"

namespace volr {

class RaycasterBase {
	public:
		static Raycaster raycaster;
		static float4 base_transfer_fn[TF_SIZE];       // editable, NOT premultiplied
		static float2 ray_step_limits;
		static void change_ray_step(float step, bool reset);
		static void change_ray_threshold(float threshold, bool reset);
		static void change_light_intensity(float intensity, bool reset);
		static void toggle_esl();
		static void set_volume(Model volume);
		static void set_view(View view);
		static void update_transfer_fn();
		static void reset_transfer_fn();
		static void reset_ray_step();
		// Extension: same as set_volume(), but the per-block min/max come from the GPU reduction
		// (vr_hip_volume_minmax) instead of a serial scan over the host copy.
		static void set_volume(Model volume, const unsigned char *minmax_pairs);
		static const unsigned char *block_min_max();   // ESL_VOLUME_DIMS^3 pairs {min, max}
	private:
		static void set_block_geometry(Model volume);
		static unsigned char esl_min_max[ESL_VOLUME_DIMS * ESL_VOLUME_DIMS * ESL_VOLUME_DIMS * 2];
		static float4 tf_storage[TF_SIZE];
		static esl_type esl_storage[ESL_VOLUME_SIZE];
};

}  // namespace volr
