// binding_check — TEST INFRASTRUCTURE.  Builds the INTEGRATION.md binding (HIPRenderer.h) against the reference's real
// Renderer.h / RaycasterBase / ModelBase / CPURenderer object code and drives it exactly like VolR.cpp does
// (VolR.cpp:412-417 init, :107-110 per frame): renderers[i]->render_volume(buffer, RaycasterBase::raycaster).
//   * without a usable GPU (the build container): the constructor logs the error, every virtual is a safe no-op and
//     render_volume() returns 1 — the reference's failure convention (CPURenderer.cpp:44-45); exit code 0 if so;
//   * with a GPU: the NEAREST frame of the binding must equal the frame of the reference's own CPURenderer byte for byte
//     (two views of a small synthetic volume); exit code 0 if so.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "HIPRenderer.h"

static void set_view(bool perspective, unsigned short w, unsigned short h) {
	View v;
	v.dims = make_ushort2(w, h);
	// benchmark pose (-45,-45,0) at distance 2 (VolR.cpp:237; values of SURVEY §8c), pixel step = view size / min(w, h)
	const float step = (perspective ? 1.5f : 2.0f) / (float) (w < h ? w : h);
	v.origin = make_float3(1.0f, -1.414214f, 1.0f);
	v.direction = make_float3(-0.5f, 0.707107f, -0.5f);
	v.right_plane = make_float3(0.707107f * step, 0.0f, -0.707107f * step);
	v.up_plane = make_float3(0.5f * step, 0.707107f * step, 0.5f * step);
	v.light_pos = make_float3(0.0f, 0.0f, 3.0f);
	v.perspective = perspective;
	RaycasterBase::set_view(v);
}

int main() {
	static char empty[1] = "";
	Logger::init(empty, 'n');
	const unsigned n = 40;
	unsigned char *vox = (unsigned char *) malloc(n * n * n);
	for (unsigned z = 0; z < n; z++) for (unsigned y = 0; y < n; y++) for (unsigned x = 0; x < n; x++) {
		const float dx = x - 19.5f, dy = y - 19.5f, dz = z - 19.5f;
		const float r = sqrtf(dx * dx + dy * dy + dz * dz);
		float v = 255.0f - fabsf(r - 12.0f) * 40.0f;
		vox[(z * n + y) * n + x] = (unsigned char) (v < 0 ? (x * 7 + y * 3 + z) % 20 : v);
	}
	ModelBase::volume.data = vox;
	ModelBase::volume.size = n * n * n;
	ModelBase::volume.dims = make_ushort3(n, n, n);
	RaycasterBase::reset_transfer_fn();
	RaycasterBase::set_volume(ModelBase::volume);
	const unsigned short W = 96, H = 80;
	set_view(false, W, H);

	Renderer *renderers[2];
	renderers[0] = new CPURenderer(RaycasterBase::raycaster);
	renderers[1] = new HIPRenderer(RaycasterBase::raycaster, VR_SAMPLE_NEAREST);
	printf("renderer 1: %s\n", renderers[1]->get_name());
	uchar4 *a = (uchar4 *) malloc(W * H * sizeof(uchar4)), *b = (uchar4 *) malloc(W * H * sizeof(uchar4));
	int rc = 0;
	vr_ctx *probe = NULL;
	const int have_gpu = vr_hip_create(0, &probe) == 0;
	vr_hip_destroy(probe);
	if (!have_gpu) {
		const int r = renderers[1]->render_volume(b, RaycasterBase::raycaster);
		const int s = renderers[1]->set_volume(RaycasterBase::raycaster.volume);
		printf("no usable GPU: render_volume returned %d, set_volume returned %d (expected 1, 1)\n", r, s);
		rc = (r == 1 && s == 1) ? 0 : 1;
	} else {
		for (int persp = 0; persp < 2; persp++) {
			set_view(persp != 0, W, H);
			if (renderers[0]->render_volume(a, RaycasterBase::raycaster) != 0 || renderers[1]->render_volume(b, RaycasterBase::raycaster) != 0) {
				printf("render_volume failed\n"); rc = 1; break;
			}
			unsigned differing = 0, covered = 0;
			for (unsigned i = 0; i < (unsigned) W * H; i++) {
				if (memcmp(&a[i], &b[i], 4) != 0) differing++;
				if (a[i].w != 0) covered++;
			}
			printf("%s: %u of %u pixels differ between CPURenderer and HIPRenderer (%u covered)\n", persp ? "perspective" : "orthogonal",
			       differing, (unsigned) W * H, covered);
			if (differing != 0 || covered < 500) rc = 1;
		}
	}
	delete renderers[1];
	delete renderers[0];
	printf(rc == 0 ? "binding check passed\n" : "binding check FAILED\n");
	return rc;
}
