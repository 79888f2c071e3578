// HIPRenderer.h — THE BINDING A MAINTAINER OF MiroBeno/Volume-Rendering ADDS, verbatim.
//
// One more `Renderer` subclass (VolumeRendering/Renderer.h:13-28) that forwards the five virtuals to the C ABI of
// include/vr_hip.h.  This file is compiled here against the REFERENCE'S REAL HEADERS (oracle/Makefile target `binding`, host
// compiler only, linked with -lvr_hip) by oracle/integration/binding_check.cpp — test infrastructure: it proves the snippet
// of INTEGRATION.md builds and behaves, it is not part of the product.
#ifndef HIP_RENDERER_BINDING_H
#define HIP_RENDERER_BINDING_H

#include "Renderer.h"          // the reference's own header (-I/root/reference/VolumeRendering)
#include "vr_hip.h"            // this repository: include/vr_hip.h

class HIPRenderer : public Renderer {            // declared next to GPURenderer1..4 in the reference's Renderer.h
	vr_ctx *ctx;
	vr_sampling sampling;
	static void fill(vr_params *p, const Raycaster &r, vr_sampling s) {
		p->view.width = r.view.dims.x;  p->view.height = r.view.dims.y;
		const float3 *src[5] = { &r.view.origin, &r.view.direction, &r.view.right_plane, &r.view.up_plane, &r.view.light_pos };
		float *dst[5] = { p->view.origin, p->view.direction, p->view.right_plane, p->view.up_plane, p->view.light_pos };
		for (int i = 0; i < 5; i++) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
		p->view.perspective = r.view.perspective;
		p->ray_step = r.ray_step;  p->ray_threshold = r.ray_threshold;  p->esl = r.esl;
		p->esl_block_dims = r.esl_block_dims;
		p->esl_block_size[0] = r.esl_block_size.x; p->esl_block_size[1] = r.esl_block_size.y; p->esl_block_size[2] = r.esl_block_size.z;
		p->light_kd = r.light_kd;  p->sampling = s;
		p->x0 = 0; p->out_width = r.view.dims.x; p->out_rows = r.view.dims.y;         // whole frame
		p->band_rows = r.view.dims.y; p->band_stride = 1; p->band_first = 0;
	}
public:
	HIPRenderer(Raycaster r, vr_sampling s = VR_SAMPLE_TRILINEAR) : ctx(0), sampling(s) {
		if (vr_hip_create(0, &ctx) != 0) { Logger::log("HIP renderer: %s\n", vr_hip_last_error(ctx)); vr_hip_destroy(ctx); ctx = 0; return; }
		set_window_buffer(r.view); set_transfer_fn(r); set_volume(r.volume);           // like GPURenderer1.cu:17-21
	}
	virtual ~HIPRenderer() { vr_hip_destroy(ctx); }
	virtual const char *get_name() { return "HIP MI355X"; }
	virtual void set_window_buffer(View v) { vr_hip_set_window(ctx, v.dims.x, v.dims.y); }
	virtual void set_transfer_fn(Raycaster r) { vr_hip_set_transfer_fn(ctx, (const float *) r.transfer_fn, r.esl_volume); }
	virtual int set_volume(Model m) { return vr_hip_set_volume(ctx, m.data, m.dims.x, m.dims.y, m.dims.z, 1) ? 1 : 0; }
	virtual int render_volume(uchar4 *buffer, Raycaster r) {
		vr_params p; fill(&p, r, sampling);
		return vr_hip_render(ctx, &p, (uint8_t *) buffer) ? 1 : 0;    // host PBO pointer, renderer ids 0-2 (VolR.cpp:76-87)
		// device-mapped buffer (ids 3-4): vr_hip_render_device(ctx, &p, buffer, NULL)
	}
};

#endif
