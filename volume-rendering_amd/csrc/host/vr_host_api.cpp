// vr_host_api.cpp — extern "C" wrappers of include/vr_host.h around the host mirror classes.
#include "../../../include/vr_host.h"
#include "RaycasterBase.h"
#include "camera.h"
#include "ModelBase.h"

#include <string.h>

using namespace volr;

namespace {

void view_to_c(const View &v, vr_view *out) {
	out->width = v.dims.x; out->height = v.dims.y;
	out->origin[0] = v.origin.x; out->origin[1] = v.origin.y; out->origin[2] = v.origin.z;
	out->direction[0] = v.direction.x; out->direction[1] = v.direction.y; out->direction[2] = v.direction.z;
	out->right_plane[0] = v.right_plane.x; out->right_plane[1] = v.right_plane.y; out->right_plane[2] = v.right_plane.z;
	out->up_plane[0] = v.up_plane.x; out->up_plane[1] = v.up_plane.y; out->up_plane[2] = v.up_plane.z;
	out->light_pos[0] = v.light_pos.x; out->light_pos[1] = v.light_pos.y; out->light_pos[2] = v.light_pos.z;
	out->perspective = v.perspective ? 1u : 0u;
}

View view_from_c(const vr_view &v) {
	View r;
	r.dims = make_ushort2((unsigned short) v.width, (unsigned short) v.height);
	r.origin = make_float3(v.origin[0], v.origin[1], v.origin[2]);
	r.direction = make_float3(v.direction[0], v.direction[1], v.direction[2]);
	r.right_plane = make_float3(v.right_plane[0], v.right_plane[1], v.right_plane[2]);
	r.up_plane = make_float3(v.up_plane[0], v.up_plane[1], v.up_plane[2]);
	r.light_pos = make_float3(v.light_pos[0], v.light_pos[1], v.light_pos[2]);
	r.perspective = v.perspective != 0;
	return r;
}

const float kBenchPoses[4][3] = { { 0, 0, 0 }, { -45, -45, 0 }, { 90, 0, 0 }, { 180, 90, 0 } };   // VolR.cpp:233-246

}  // namespace

extern "C" {

int vr_host_benchmark_view(uint32_t w, uint32_t h, uint32_t perspective, const float angles[3], float distance, vr_view *out) {
	if (out == nullptr || angles == nullptr || w == 0 || h == 0 || w > 65535u || h > 65535u)
		return VR_ERR_INVALID;
	ViewBase::reset();
	ViewBase::set_viewport_dims(make_ushort2((unsigned short) w, (unsigned short) h));
	ViewBase::view.perspective = perspective != 0;          // VolR.cpp:231,247
	ViewBase::toggle_perspective(1);                         // VolR.cpp:233
	ViewBase::set_camera_position(make_float3(angles[0], angles[1], angles[2]), distance);
	view_to_c(ViewBase::view, out);
	return VR_OK;
}

int vr_host_benchmark_view_index(uint32_t w, uint32_t h, uint32_t index, vr_view *out) {
	if (index > 7) return VR_ERR_INVALID;
	return vr_host_benchmark_view(w, h, index >= 4, kBenchPoses[index & 3], 2.0f, out);
}

int vr_host_raycaster_set_volume(const uint8_t *voxels, uint32_t x, uint32_t y, uint32_t z, const uint8_t *minmax) {
	if ((voxels == nullptr && minmax == nullptr) || x == 0 || y == 0 || z == 0 || x > 65535u || y > 65535u || z > 65535u)
		return VR_ERR_INVALID;
	Model m;
	m.data = const_cast<unsigned char *>(voxels);
	m.size = (unsigned int) ((uint64_t) x * y * z);
	m.dims = make_ushort3((unsigned short) x, (unsigned short) y, (unsigned short) z);
	m.min_bound = make_float3(-1, -1, -1);
	RaycasterBase::reset_transfer_fn();
	if (minmax) RaycasterBase::set_volume(m, minmax);
	else        RaycasterBase::set_volume(m);
	return VR_OK;
}

void vr_host_raycaster_reset_transfer_fn(void) { RaycasterBase::reset_transfer_fn(); }

int vr_host_raycaster_set_base_transfer_fn(const float *base) {
	if (base == nullptr) return VR_ERR_INVALID;
	for (int i = 0; i < TF_SIZE; i++)
		RaycasterBase::base_transfer_fn[i] = make_float4(base[4 * i], base[4 * i + 1], base[4 * i + 2], base[4 * i + 3]);
	RaycasterBase::update_transfer_fn();
	return VR_OK;
}

void vr_host_raycaster_change_ray_step(float step, int reset) { RaycasterBase::change_ray_step(step, reset != 0); }
void vr_host_raycaster_change_ray_threshold(float t, int reset) { RaycasterBase::change_ray_threshold(t, reset != 0); }
void vr_host_raycaster_change_light_intensity(float i, int reset) { RaycasterBase::change_light_intensity(i, reset != 0); }
void vr_host_raycaster_set_esl(int on) { if (RaycasterBase::raycaster.esl != (on != 0)) RaycasterBase::toggle_esl(); }
void vr_host_raycaster_reset_ray_step(void) { RaycasterBase::reset_ray_step(); }

int vr_host_raycaster_get(vr_params *p, float *tf_out, uint32_t *esl_out, uint8_t *minmax_out, float *base_out) {
	const Raycaster &r = RaycasterBase::raycaster;
	if (p) {
		p->ray_step = r.ray_step; p->ray_threshold = r.ray_threshold; p->esl = r.esl ? 1u : 0u;
		p->esl_block_dims = r.esl_block_dims;
		p->esl_block_size[0] = r.esl_block_size.x; p->esl_block_size[1] = r.esl_block_size.y; p->esl_block_size[2] = r.esl_block_size.z;
		p->light_kd = r.light_kd;
	}
	if (tf_out) memcpy(tf_out, r.transfer_fn, TF_SIZE * sizeof(float4));
	if (esl_out) memcpy(esl_out, r.esl_volume, ESL_VOLUME_SIZE * sizeof(esl_type));
	if (minmax_out) memcpy(minmax_out, RaycasterBase::block_min_max(), ESL_VOLUME_DIMS * ESL_VOLUME_DIMS * ESL_VOLUME_DIMS * 2);
	if (base_out) memcpy(base_out, RaycasterBase::base_transfer_fn, TF_SIZE * sizeof(float4));
	return VR_OK;
}

int vr_host_load_model(const char *file_name, uint32_t dims_out[3]) {
	if (file_name == nullptr) return 1;
	const int rc = ModelBase::load_model(file_name);
	if (rc == 0 && dims_out) {
		dims_out[0] = ModelBase::volume.dims.x; dims_out[1] = ModelBase::volume.dims.y; dims_out[2] = ModelBase::volume.dims.z;
	}
	return rc;
}

void vr_host_set_raw_dims(uint32_t w, uint32_t h, uint32_t d, uint32_t components) { ModelBase::set_raw_dims(w, h, d, components); }
const uint8_t *vr_host_model_voxels(void) { return ModelBase::volume.data; }
void vr_host_model_histogram(float out[256]) { memcpy(out, ModelBase::histogram, sizeof ModelBase::histogram); }

int vr_host_quantize(const uint8_t *data16, uint32_t w, uint32_t h, uint32_t d, int linear, uint8_t *out8) {
	if (data16 == nullptr || out8 == nullptr || w == 0 || h == 0 || d == 0) return VR_ERR_INVALID;
	const std::vector<uint8_t> q = quantize_16_to_8(data16, w, h, d, linear != 0);
	memcpy(out8, q.data(), q.size());
	return VR_OK;
}

int vr_host_render_frame(int device, uint32_t sampling, const vr_view *view, uint8_t *host_rgba) {
	if (view == nullptr) return 1;
	RaycasterBase::set_view(view_from_c(*view));                                  // VolR.cpp:107
	HipRenderer renderer(RaycasterBase::raycaster, device, (vr_sampling) sampling);
	if (!renderer.ok()) return 1;
	return renderer.render_volume((uchar4 *) host_rgba, RaycasterBase::raycaster);  // VolR.cpp:110
}

}  // extern "C"
