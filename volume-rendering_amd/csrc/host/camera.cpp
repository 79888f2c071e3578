// camera.cpp — see camera.h.  Behaviour follows VolumeRendering/ViewBase.cpp; the arithmetic is fp32 in the exact order
// that reproduces the reference's frames (the views are inputs of the ray-march path).
#include "camera.h"

#include <math.h>
#include <string.h>

namespace volr {

namespace {

const float kMinDistance = 0.1f, kMaxDistance = 3.0f;                  // ViewBase.cpp:16 distance_limits

void set_identity(float m[16]) {
	for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
}

// m <- m * R(angle about axis): what one glRotatef does to the current matrix.  Column-major storage, only the 3x3
// rotation block is touched (the rest stays identity here).  Each element is a left-to-right fp32 sum of three products.
void post_multiply_rotation(float m[16], float degrees, float ax, float ay, float az) {
	const float norm = sqrtf(ax * ax + ay * ay + az * az);
	if (norm == 0.0f)
		return;
	const float x = ax / norm, y = ay / norm, z = az / norm;
	const float angle = degrees * 3.14159265358979323846f / 180.0f;
	const float c = cosf(angle), s = sinf(angle), t = 1.0f - c;
	const float rot[3][3] = {
		{ x * x * t + c,     x * y * t - z * s, x * z * t + y * s },
		{ y * x * t + z * s, y * y * t + c,     y * z * t - x * s },
		{ x * z * t - y * s, y * z * t + x * s, z * z * t + c     },
	};
	float product[3][3];
	for (int r = 0; r < 3; r++)
		for (int col = 0; col < 3; col++) {
			float sum = 0.0f;
			for (int k = 0; k < 3; k++)
				sum = sum + m[k * 4 + r] * rot[k][col];
			product[r][col] = sum;
		}
	for (int r = 0; r < 3; r++)
		for (int col = 0; col < 3; col++)
			m[col * 4 + r] = product[r][col];
}

// the reference's vector_rotate (ViewBase.cpp:26-32): dot products of a 4-vector with consecutive quadruples of the array
float3 apply(const float m[16], float x, float y, float z, float w) {
	return make_float3(x * m[0] + y * m[1] + z * m[2]  + w * m[3],
	                   x * m[4] + y * m[5] + z * m[6]  + w * m[7],
	                   x * m[8] + y * m[9] + z * m[10] + w * m[11]);
}

int short_side(const ushort2 &d) { return d.x < d.y ? d.x : d.y; }

}  // namespace

void Camera::restart() {
	set_identity(rotation);
	set_identity(light_rotation);
	distance_ = 3.0f;
	light_distance_ = 3.0f;
	window_size_ = 3.0f;
	view.dims = make_ushort2(INT_WIN_WIDTH, INT_WIN_HEIGHT);
	view.perspective = false;
	view.light_pos = make_float3(0, 0, 3);
	deg_per_pixel_ = 180.0f / short_side(view.dims);
	dist_per_pixel_ = (kMaxDistance - kMinDistance) / (INT_WIN_HEIGHT / 2);
	derive();
}

// ViewBase.cpp:49-55
void Camera::derive() {
	view.origin = apply(rotation, 0.0f, 0.0f, distance_, 1.0f);
	const float3 to_centre = make_float3(-view.origin.x, -view.origin.y, -view.origin.z);
	const float inv_len = 1.0f / sqrtf(to_centre.x * to_centre.x + to_centre.y * to_centre.y + to_centre.z * to_centre.z);
	view.direction = make_float3(to_centre.x * inv_len, to_centre.y * inv_len, to_centre.z * inv_len);
	const float pixel = window_size_ / short_side(view.dims);
	const float3 right = apply(rotation, 1, 0, 0, 0), up = apply(rotation, 0, 1, 0, 0);
	view.right_plane = make_float3(right.x * pixel, right.y * pixel, right.z * pixel);
	view.up_plane = make_float3(up.x * pixel, up.y * pixel, up.z * pixel);
}

// ViewBase.cpp:34-47: three successive rotations about the matrix's own axes — all three axes taken from the matrix as
// it was before the first of them
void Camera::turn(float m[16], const float3 &degrees, bool from_identity) {
	if (from_identity)
		set_identity(m);
	const float own_x[3] = { m[0], m[4], m[8] }, own_y[3] = { m[1], m[5], m[9] }, own_z[3] = { m[2], m[6], m[10] };
	post_multiply_rotation(m, degrees.x, own_x[0], own_x[1], own_x[2]);
	post_multiply_rotation(m, degrees.y, own_y[0], own_y[1], own_y[2]);
	post_multiply_rotation(m, degrees.z, own_z[0], own_z[1], own_z[2]);
}

void Camera::orbit(const float3 &degrees, bool from_identity) {
	turn(rotation, degrees, from_identity);
	derive();
}

void Camera::orbit_pixels(int dx, int dy, int dz) {
	orbit(make_float3(dy * deg_per_pixel_, dx * deg_per_pixel_, dz * deg_per_pixel_), false);   // ViewBase.cpp:62-72
}

// ViewBase.cpp:74-79: in the orthogonal projection zooming IS resizing the virtual window
void Camera::dolly(float delta) {
	float d = distance_ + delta;
	if (d < kMinDistance) d = kMinDistance;
	if (d > kMaxDistance) d = kMaxDistance;
	distance_ = d;
	if (!view.perspective)
		window_size_ = distance_;
	derive();
}

void Camera::dolly_pixels(int pixels) { dolly(pixels * dist_per_pixel_); }

// ViewBase.cpp:85-89
void Camera::place(const float3 &degrees, float distance) {
	distance_ = 0;
	dolly(distance);
	orbit(degrees, true);
}

// ViewBase.cpp:91-98
void Camera::orbit_light_pixels(int dx, int dy) {
	turn(light_rotation, make_float3(dy * deg_per_pixel_, dx * deg_per_pixel_, 0), false);
	view.light_pos = apply(light_rotation, 0.0f, 0.0f, light_distance_, 1.0f);
}

// ViewBase.cpp:100-105: perspective uses a fixed 1.5-unit window at the camera's unit distance
void Camera::flip_projection(bool only_refresh) {
	if (!only_refresh)
		view.perspective = !view.perspective;
	window_size_ = view.perspective ? 1.5f : distance_;
	derive();
}

// ViewBase.cpp:107-113
void Camera::resize(ushort2 dims, float scale) {
	view.dims.x = (unsigned short) (dims.x * scale);
	view.dims.y = (unsigned short) (dims.y * scale);
	deg_per_pixel_ = 180.0f / short_side(dims);
	dist_per_pixel_ = (kMaxDistance - kMinDistance) / (dims.y / 2);
	derive();
}

// ---- static facade ------------------------------------------------------------------------------------------------------

Camera &ViewBase::camera() {
	static Camera instance;
	return instance;
}

View &ViewBase::view = ViewBase::camera().view;
float (&ViewBase::cam_matrix)[16] = ViewBase::camera().rotation;
float (&ViewBase::light_matrix)[16] = ViewBase::camera().light_rotation;

}  // namespace volr
