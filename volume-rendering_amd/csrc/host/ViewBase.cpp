// ViewBase.cpp — camera state -> View, restated from the reference's behaviour (VolumeRendering/ViewBase.cpp) with
// the OpenGL matrix calls replaced by plain fp32 math; see ViewBase.h.
#include "ViewBase.h"

#include <math.h>
#include <string.h>

namespace volr {

namespace {

const float kIdentity[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };

int shorter_side(int w, int h) { return w < h ? w : h; }

// What glRotatef(angle, x, y, z) does to the current matrix M (column-major, m[col*4+row]): M = M * R with R the
// rotation by `angle` degrees about the normalised axis.  Only the upper-left 3x3 of M is ever non-trivial here.
void post_rotate(float m[16], float degrees, float ax, float ay, float az) {
	const float len = sqrtf(ax * ax + ay * ay + az * az);
	if (len == 0.0f)
		return;
	const float x = ax / len, y = ay / len, z = az / len;
	const float rad = degrees * 3.14159265358979323846f / 180.0f;
	const float c = cosf(rad), s = sinf(rad), t = 1.0f - c;
	const float r[3][3] = {                           // r[row][col]
		{ x * x * t + c,     x * y * t - z * s, x * z * t + y * s },
		{ y * x * t + z * s, y * y * t + c,     y * z * t - x * s },
		{ x * z * t - y * s, y * z * t + x * s, z * z * t + c     },
	};
	float out[16];
	memcpy(out, m, sizeof out);
	for (int row = 0; row < 3; row++)
		for (int col = 0; col < 3; col++) {
			float acc = 0.0f;
			for (int k = 0; k < 3; k++)
				acc = acc + m[k * 4 + row] * r[k][col];
			out[col * 4 + row] = acc;
		}
	memcpy(m, out, sizeof out);
}

float3 scaled(float3 v, float s) { return make_float3(v.x * s, v.y * s, v.z * s); }

}  // namespace

// start-up state, ViewBase.cpp:8-24
View ViewBase::view = {
	{ INT_WIN_WIDTH, INT_WIN_HEIGHT },
	{ 0, 0, 3 },
	{ 0, 0, -1 },
	{ 0, 0, -1.0f * (3.0f / INT_WIN_HEIGHT) },        // sic: the reference initialises right_plane along -z (ViewBase.cpp:11)
	{ 0, 1.0f * (3.0f / INT_WIN_HEIGHT), 0 },
	{ 0, 0, 3 },
	false
};
const float2 ViewBase::distance_limits = { 0.1f, 3.0f };
float4 ViewBase::cam_pos = { 0, 0, 3, 1 };
float4 ViewBase::light_pos = { 0, 0, 3, 1 };
float ViewBase::cam_matrix[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
float ViewBase::light_matrix[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
float ViewBase::pixel_ratio_rotation = 180.0f / INT_WIN_HEIGHT;
float ViewBase::pixel_ratio_translation = (3.0f - 0.1f) / (INT_WIN_HEIGHT / 2);
float ViewBase::virtual_view_size = 3.0f;

void ViewBase::reset() {
	view.dims = make_ushort2(INT_WIN_WIDTH, INT_WIN_HEIGHT);
	view.perspective = false;
	view.light_pos = make_float3(0, 0, 3);
	cam_pos = make_float4(0, 0, 3, 1);
	light_pos = make_float4(0, 0, 3, 1);
	memcpy(cam_matrix, kIdentity, sizeof kIdentity);
	memcpy(light_matrix, kIdentity, sizeof kIdentity);
	pixel_ratio_rotation = 180.0f / shorter_side(INT_WIN_WIDTH, INT_WIN_HEIGHT);
	pixel_ratio_translation = (distance_limits.y - distance_limits.x) / (INT_WIN_HEIGHT / 2);
	virtual_view_size = 3.0f;
	update_view();
}

// ViewBase.cpp:26-32: three dot products of v with consecutive quadruples of the matrix array
float3 ViewBase::vector_rotate(float4 v, const float m[16]) {
	float3 r;
	r.x = v.x * m[0] + v.y * m[1] + v.z * m[2]  + v.w * m[3];
	r.y = v.x * m[4] + v.y * m[5] + v.z * m[6]  + v.w * m[7];
	r.z = v.x * m[8] + v.y * m[9] + v.z * m[10] + v.w * m[11];
	return r;
}

// ViewBase.cpp:34-47: rotate about the matrix's own three axes; all three axes are read from the matrix as it was
// BEFORE the first rotation (the reference only reads the GL matrix back after the third glRotatef)
void ViewBase::matrix_rotate(float matrix[], float3 angles, bool reset) {
	if (reset)
		memcpy(matrix, kIdentity, sizeof kIdentity);
	float m[16];
	memcpy(m, matrix, sizeof m);
	post_rotate(m, angles.x, matrix[0], matrix[4], matrix[8]);
	post_rotate(m, angles.y, matrix[1], matrix[5], matrix[9]);
	post_rotate(m, angles.z, matrix[2], matrix[6], matrix[10]);
	memcpy(matrix, m, sizeof m);
}

// ViewBase.cpp:49-55
void ViewBase::update_view() {
	view.origin = vector_rotate(cam_pos, cam_matrix);
	const float3 n = make_float3(-view.origin.x, -view.origin.y, -view.origin.z);
	view.direction = scaled(n, 1.0f / sqrtf(n.x * n.x + n.y * n.y + n.z * n.z));
	const float step_px = virtual_view_size / shorter_side(view.dims.x, view.dims.y);
	view.right_plane = scaled(vector_rotate(make_float4(1, 0, 0, 0), cam_matrix), step_px);
	view.up_plane = scaled(vector_rotate(make_float4(0, 1, 0, 0), cam_matrix), step_px);
}

void ViewBase::camera_rotate(float3 angles, bool reset) {
	matrix_rotate(cam_matrix, angles, reset);
	update_view();
}

void ViewBase::camera_rotate(int2 pixels) {
	camera_rotate(make_float3(pixels.y * pixel_ratio_rotation, pixels.x * pixel_ratio_rotation, 0));
}

void ViewBase::camera_rotate(int3 pixels) {
	camera_rotate(make_float3(pixels.y * pixel_ratio_rotation, pixels.x * pixel_ratio_rotation, pixels.z * pixel_ratio_rotation));
}

// ViewBase.cpp:74-79: in orthogonal mode the zoom IS the size of the virtual window
void ViewBase::camera_zoom(float distance) {
	float z = cam_pos.z + distance;
	z = z < distance_limits.x ? distance_limits.x : (z > distance_limits.y ? distance_limits.y : z);
	cam_pos.z = z;
	if (!view.perspective)
		virtual_view_size = cam_pos.z;
	update_view();
}

void ViewBase::camera_zoom(int pixels) {
	camera_zoom(pixels * pixel_ratio_translation);
}

// ViewBase.cpp:85-89
void ViewBase::set_camera_position(float3 angles, float distance) {
	cam_pos.z = 0;
	camera_zoom(distance);
	camera_rotate(angles, true);
}

// ViewBase.cpp:91-98
void ViewBase::light_rotate(int2 pixels) {
	matrix_rotate(light_matrix, make_float3(pixels.y * pixel_ratio_rotation, pixels.x * pixel_ratio_rotation, 0), false);
	view.light_pos = vector_rotate(light_pos, light_matrix);
}

// ViewBase.cpp:100-105: update_mode != 0 only refreshes the derived state
void ViewBase::toggle_perspective(int update_mode) {
	if (!update_mode)
		view.perspective = !view.perspective;
	virtual_view_size = view.perspective ? 1.5f : cam_pos.z;
	update_view();
}

// ViewBase.cpp:107-113
void ViewBase::set_viewport_dims(ushort2 dims, float scale) {
	view.dims.x = (unsigned short) (dims.x * scale);
	view.dims.y = (unsigned short) (dims.y * scale);
	pixel_ratio_rotation = 180.0f / shorter_side(dims.x, dims.y);
	pixel_ratio_translation = (distance_limits.y - distance_limits.x) / (dims.y / 2);
	update_view();
}

}  // namespace volr
