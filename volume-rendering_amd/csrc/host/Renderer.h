// Renderer.h — host-side C++ mirror of the reference's renderer plug-in interface, for the ONE back-end this
// project provides: HipRenderer (MI355X).  Same class / method / field names, argument meaning and return
// conventions as the reference (VolumeRendering/Renderer.h:13-28, RaycasterBase.h:20-30, ModelBase.h:11-15,
// ViewBase.h:14-21), so a call site written against the reference compiles against this header unchanged
// (inside namespace volr) and tests read like the reference's own call sequences.
//
// All GPU work happens below the C ABI of include/vr_hip.h; this file contains no HIP calls.
#pragma once

#include <hip/hip_vector_types.h>   // float3 / float4 / uchar4 / ushort2 / ushort3 / short2 PODs (HIP's own, not CUDA's)
#include "../../../include/vr_hip.h"

namespace volr {

typedef unsigned int esl_type;      // RaycasterBase.h:18

constexpr int TF_SIZE = VR_TF_SIZE;
constexpr int TF_RATIO = VR_TF_RATIO;
constexpr int ESL_VOLUME_DIMS = VR_ESL_VOLUME_DIMS;
constexpr int ESL_VOLUME_SIZE = VR_ESL_VOLUME_SIZE;
constexpr int ESL_MIN_BLOCK_SIZE = VR_ESL_MIN_BLOCK;

// ModelBase.h:11-15.  `data` stays owned by the caller; renderers copy it in set_volume().
struct Model {
	unsigned char *data;
	unsigned int size;
	ushort3 dims;
	float3 min_bound;               // always (-1,-1,-1) (ModelBase.cpp:10-14)
};

// ViewBase.h:14-21
struct View {
	ushort2 dims;
	float3 origin;
	float3 direction;
	float3 right_plane;
	float3 up_plane;
	float3 light_pos;
	bool perspective;
};

// RaycasterBase.h:20-30 (the per-ray methods of the reference struct live in the HIP kernel, not here)
struct Raycaster {
	Model volume;
	View view;
	float4 *transfer_fn;            // TF_SIZE premultiplied entries, owned by RaycasterBase
	float ray_step;
	float ray_threshold;
	bool esl;
	esl_type *esl_volume;           // ESL_VOLUME_SIZE words, bit set = block empty, owned by RaycasterBase
	unsigned short esl_block_dims;
	float3 esl_block_size;
	float light_kd;
};

// Renderer.h:13-28
class Renderer {
	public:
		virtual ~Renderer() {}
		virtual const char *get_name() { return "Default"; }
		virtual void set_window_buffer(View) {}
		virtual void set_transfer_fn(Raycaster) {}
		virtual int set_volume(Model) { return 0; }
		virtual int render_volume(uchar4 *buffer, Raycaster r) = 0;
};

// The MI355X back-end.  Construction performs the three set_* calls like every reference GPU renderer
// (GPURenderer1.cu:17-21).  `buffer` of render_volume() is a HOST pointer by default (reference renderer ids 0-2,
// VolR.cpp:76-87) or a DEVICE pointer when constructed with device_buffer = true (ids 3-4).
class HipRenderer : public Renderer {
	public:
		explicit HipRenderer(Raycaster r, int device = 0, vr_sampling sampling = VR_SAMPLE_TRILINEAR, bool device_buffer = false);
		// several GPUs behind the same five virtuals (include/vr_hip.h vr_hip_multi_*): the frame is split into interleaved bands,
		// gathered on devices[0]; with device_buffer the `buffer` of render_volume() is a device pointer on devices[0]
		HipRenderer(Raycaster r, const int *devices, int n_devices, vr_sampling sampling = VR_SAMPLE_TRILINEAR, bool device_buffer = false);
		virtual ~HipRenderer();
		virtual const char *get_name() { return sampling_ == VR_SAMPLE_NEAREST ? "HIP MI355X nearest" : (sampling_ == VR_SAMPLE_TRILINEAR_Q8 ? "HIP MI355X trilinear q8" : "HIP MI355X trilinear"); }
		virtual void set_window_buffer(View view);
		virtual void set_transfer_fn(Raycaster r);
		virtual int set_volume(Model volume);
		virtual int render_volume(uchar4 *buffer, Raycaster r);

		bool ok() const { return (ctx_ != nullptr || multi_ != nullptr) && create_status_ == 0; }
		const char *last_error() const;
		vr_ctx *context() { return multi_ ? vr_hip_multi_context(multi_, 0) : ctx_; }
		vr_multi *multi() { return multi_; }
		void set_sampling(vr_sampling s) { sampling_ = s; }
		// fills the by-value parameter block from a Raycaster (whole-frame partition)
		static void to_params(const Raycaster &r, vr_sampling sampling, vr_params *out);
	private:
		HipRenderer(const HipRenderer &);
		HipRenderer &operator=(const HipRenderer &);
		void prime(const Raycaster &r);
		vr_ctx *ctx_;
		vr_multi *multi_;
		int create_status_;
		vr_sampling sampling_;
		bool device_buffer_;
};

}  // namespace volr
