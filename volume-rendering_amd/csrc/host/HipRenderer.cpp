// HipRenderer.cpp — the Renderer back-end that forwards the reference's five virtuals to the C ABI (include/vr_hip.h).
// Return conventions follow the reference: 0 = ok, 1 = failure (CPURenderer.cpp:44-45, GPURenderer1.cu:91-95,101-102);
// nothing here exits the process (the reference's cuda_safe_call does, cuda_utils.h:25-31 — deliberately not mirrored).
#include "Renderer.h"

namespace volr {

static void copy3(float *dst, const float3 &v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

void HipRenderer::to_params(const Raycaster &r, vr_sampling sampling, vr_params *p) {
	p->view.width = r.view.dims.x;
	p->view.height = r.view.dims.y;
	copy3(p->view.origin, r.view.origin);
	copy3(p->view.direction, r.view.direction);
	copy3(p->view.right_plane, r.view.right_plane);
	copy3(p->view.up_plane, r.view.up_plane);
	copy3(p->view.light_pos, r.view.light_pos);
	p->view.perspective = r.view.perspective ? 1u : 0u;
	p->ray_step = r.ray_step;
	p->ray_threshold = r.ray_threshold;
	p->esl = r.esl ? 1u : 0u;
	p->esl_block_dims = r.esl_block_dims;
	copy3(p->esl_block_size, r.esl_block_size);
	p->light_kd = r.light_kd;
	p->sampling = (uint32_t) sampling;
	p->x0 = 0;
	p->out_width = r.view.dims.x;
	p->out_rows = r.view.dims.y;
	p->band_rows = r.view.dims.y ? r.view.dims.y : 1;
	p->band_stride = 1;
	p->band_first = 0;
}

HipRenderer::HipRenderer(Raycaster r, int device, vr_sampling sampling, bool device_buffer)
	: ctx_(nullptr), multi_(nullptr), create_status_(0), sampling_(sampling), device_buffer_(device_buffer) {
	create_status_ = vr_hip_create(device, &ctx_);
	if (create_status_ == 0)
		prime(r);
}

HipRenderer::HipRenderer(Raycaster r, const int *devices, int n_devices, vr_sampling sampling, bool device_buffer)
	: ctx_(nullptr), multi_(nullptr), create_status_(0), sampling_(sampling), device_buffer_(device_buffer) {
	create_status_ = vr_hip_multi_create(n_devices, devices, &multi_);
	if (create_status_ == 0)
		prime(r);
}

// GPURenderer1.cu:17-21: the constructor primes window, TF and volume from the Raycaster it is given
void HipRenderer::prime(const Raycaster &r) {
	if (r.view.dims.x != 0 && r.view.dims.y != 0)
		set_window_buffer(r.view);
	if (r.transfer_fn != nullptr && r.esl_volume != nullptr)
		set_transfer_fn(r);
	if (r.volume.data != nullptr)
		set_volume(r.volume);
}

HipRenderer::~HipRenderer() {
	vr_hip_destroy(ctx_);
	vr_hip_multi_destroy(multi_);
}

const char *HipRenderer::last_error() const {
	if (multi_ != nullptr)
		return vr_hip_multi_last_error(multi_);
	if (ctx_ == nullptr)
		return create_status_ == VR_ERR_NO_DEVICE ? "no usable HIP device (there is no CPU fallback)" : "context creation failed";
	return vr_hip_last_error(ctx_);
}

void HipRenderer::set_window_buffer(View view) {
	if (!ok())
		return;
	if (multi_) vr_hip_multi_set_window(multi_, view.dims.x, view.dims.y);
	else vr_hip_set_window(ctx_, view.dims.x, view.dims.y);
}

void HipRenderer::set_transfer_fn(Raycaster r) {
	if (!ok())
		return;
	if (multi_) vr_hip_multi_set_transfer_fn(multi_, (const float *) r.transfer_fn, r.esl_volume);
	else vr_hip_set_transfer_fn(ctx_, (const float *) r.transfer_fn, r.esl_volume);
}

int HipRenderer::set_volume(Model volume) {
	if (!ok())
		return 1;
	int rc = multi_ ? vr_hip_multi_set_volume(multi_, volume.data, volume.dims.x, volume.dims.y, volume.dims.z, 1)
	                : vr_hip_set_volume(ctx_, volume.data, volume.dims.x, volume.dims.y, volume.dims.z, 1);
	return rc == 0 ? 0 : 1;
}

int HipRenderer::render_volume(uchar4 *buffer, Raycaster r) {
	if (!ok() || buffer == nullptr)
		return 1;
	vr_params p;
	to_params(r, sampling_, &p);
	int rc;
	if (multi_) rc = device_buffer_ ? vr_hip_multi_render_device(multi_, &p, buffer) : vr_hip_multi_render(multi_, &p, (uint8_t *) buffer);
	else rc = device_buffer_ ? vr_hip_render_device(ctx_, &p, buffer, nullptr) : vr_hip_render(ctx_, &p, (uint8_t *) buffer);
	return rc == 0 ? 0 : 1;
}

}  // namespace volr
