// camera.h — projection / camera state of the scene, GL-free.
//
// `Camera` is this project's own orbit camera: a rotation about the origin kept as a column-major 4x4 float array (the
// layout the reference's GLUI rotation widgets bind to), a distance, a projection mode and a light on its own orbit; it
// derives the `View` parameter block the ray-march path consumes.  `ViewBase` is the static facade with the reference's
// interface (VolumeRendering/ViewBase.h:38-62) over one global Camera, so call sites written against the reference
// (VolR.cpp:232-248, UI.cpp) read the same.  The reference computes its rotations with OpenGL's matrix stack
// (ViewBase.cpp:34-47); here the same products are formed in fp32 by hand, and they reproduce the reference's benchmark
// frames hash for hash (tests/golden, oracle/gen_golden.py).
#pragma once

#include "Renderer.h"

namespace volr {

constexpr int INT_WIN_WIDTH = 799;      // ViewBase.h:11-12: start-up window
constexpr int INT_WIN_HEIGHT = 715;

class Camera {
	public:
		Camera() { restart(); }
		void restart();                                               // start-up state (ViewBase.cpp:8-24)
		void resize(ushort2 dims, float scale);                        // viewport + pixel->angle / pixel->distance ratios
		void orbit(const float3 &degrees, bool from_identity);         // turn about the camera's OWN x, y, z axes
		void orbit_pixels(int dx, int dy, int dz);
		void dolly(float delta);                                       // distance change, clamped to [0.1, 3]
		void dolly_pixels(int pixels);
		void place(const float3 &degrees, float distance);             // absolute pose
		void orbit_light_pixels(int dx, int dy);
		void flip_projection(bool only_refresh);
		void derive();                                                 // state -> view (origin, direction, image-plane steps)

		View view;
		float rotation[16];          // camera orientation, column-major
		float light_rotation[16];
	private:
		static void turn(float m[16], const float3 &degrees, bool from_identity);
		float distance_;             // camera sits at (0, 0, distance_) before rotation
		float light_distance_;
		float window_size_;          // edge of the virtual window in model units
		float deg_per_pixel_, dist_per_pixel_;
};

// The reference's static interface (ViewBase.h:38-62).
class ViewBase {
	public:
		static View &view;
		static float (&cam_matrix)[16];
		static float (&light_matrix)[16];
		static void update_view() { camera().derive(); }
		static void camera_rotate(float3 angles, bool reset = false) { camera().orbit(angles, reset); }
		static void camera_rotate(int3 pixels) { camera().orbit_pixels(pixels.x, pixels.y, pixels.z); }
		static void camera_rotate(int2 pixels) { camera().orbit_pixels(pixels.x, pixels.y, 0); }
		static void camera_zoom(float distance) { camera().dolly(distance); }
		static void camera_zoom(int pixels) { camera().dolly_pixels(pixels); }
		static void set_camera_position(float3 angles, float distance = 3.0f) { camera().place(angles, distance); }
		static void light_rotate(int2 pixels) { camera().orbit_light_pixels(pixels.x, pixels.y); }
		static void toggle_perspective(int update_mode) { camera().flip_projection(update_mode != 0); }
		static void set_viewport_dims(ushort2 dims, float scale = 1.0f) { camera().resize(dims, scale); }
		static void reset() { camera().restart(); }                    // extension: back to the start-up state
		static Camera &camera();
};

}  // namespace volr
