// ViewBase.h — host mirror of the reference's projection / camera manager (VolumeRendering/ViewBase.h:38-62,
// ViewBase.cpp).  The reference borrows OpenGL's matrix stack as its math library (ViewBase.cpp:34-47); this mirror
// carries its own fp32 rotation code instead (no GL anywhere in this project).  Its outputs — the View structs — are
// inputs of the ray-march path, so the arithmetic is fp32 throughout like the GL the reference frames were made with:
// it reproduces the reference's benchmark frames hash for hash (tests/golden, oracle/gen_golden.py).
#pragma once

#include "Renderer.h"

namespace volr {

// ViewBase.h:11-12
constexpr int INT_WIN_WIDTH = 799;
constexpr int INT_WIN_HEIGHT = 715;

class ViewBase {
	public:
		static View view;
		static float cam_matrix[16];
		static float light_matrix[16];
		static void update_view();
		static void camera_rotate(float3 angles, bool reset = false);
		static void camera_rotate(int3 pixels);
		static void camera_rotate(int2 pixels);
		static void camera_zoom(float distance);
		static void camera_zoom(int pixels);
		static void set_camera_position(float3 angles, float distance = 3.0f);
		static void light_rotate(int2 pixels);
		static void toggle_perspective(int update_mode);
		static void set_viewport_dims(ushort2 dims, float scale = 1.0f);
		static void reset();        // extension: back to the start-up state (the reference never needs it)
	private:
		static void matrix_rotate(float matrix[], float3 angles, bool reset);
		static float3 vector_rotate(float4 v, const float rot_matrix[16]);
		static const float2 distance_limits;
		static float4 cam_pos;
		static float4 light_pos;
		static float pixel_ratio_rotation;
		static float pixel_ratio_translation;
		static float virtual_view_size;
};

}  // namespace volr
