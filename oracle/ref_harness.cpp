// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// extern "C" harness around the *reference's own* CPU path, compiled from the sources where they lie
// under /root/reference (see oracle/Makefile; output goes to oracle/_ref/ only).  Nothing in here
// re-implements the algorithm: every call lands in the reference's CPURenderer / RaycasterBase /
// ModelBase / ddsbase object code.  It is used
//   * to pin oracle/vr_oracle.c (the CPU restatement) and to generate tests/golden/ fixtures
//     (oracle/gen_golden.py), and
//   * as bench.py's cpu_baseline leg with kind "reference".
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the resulting library.
//
// The call sequence mirrors VolR.cpp:412-417 (init) and VolR.cpp:98-113 / 232-248 (per frame).
// ViewBase.cpp is NOT linked: it needs OpenGL's matrix stack (ViewBase.cpp:34-47), which this image
// lacks, and a GL stand-in is not allowed.  Views are therefore passed in explicitly by the caller.

#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "Renderer.h"   // reference header (-I/root/reference/VolumeRendering)

static CPURenderer *g_cpu = NULL;

extern "C" {

int volr_ref_init(void) {
	static char empty[1] = "";
	static bool done = false;
	if (!done) {
		Logger::init(empty, 'n');           // Logger.cpp:23-28: 'n' disables the log file
		done = true;
	}
	if (g_cpu == NULL)
		g_cpu = new CPURenderer(RaycasterBase::raycaster);
	return 0;
}

// ModelBase::load_model (ModelBase.cpp:35-109) -> readPVMvolume -> DDS decode; then the init sequence
// RaycasterBase::reset_transfer_fn / set_volume of VolR.cpp:416-417.
int volr_ref_load_model(const char *path) {
	int err = ModelBase::load_model(path);
	if (err) return err;
	RaycasterBase::reset_transfer_fn();
	RaycasterBase::set_volume(ModelBase::volume);
	return 0;
}

// Synthetic volumes: fill ModelBase::volume directly (the .raw loader prompts on stdin, ModelBase.cpp:78-88).
int volr_ref_set_volume(const unsigned char *voxels, unsigned int x, unsigned int y, unsigned int z) {
	unsigned int size = x * y * z;
	unsigned char *copy = (unsigned char *) malloc(size);
	if (copy == NULL) return 1;
	memcpy(copy, voxels, size);
	if (ModelBase::volume.data != NULL) free(ModelBase::volume.data);
	ModelBase::volume.data = copy;
	ModelBase::volume.size = size;
	ModelBase::volume.dims = make_ushort3(x, y, z);
	RaycasterBase::reset_transfer_fn();
	RaycasterBase::set_volume(ModelBase::volume);
	return 0;
}

void volr_ref_get_dims(unsigned int *x, unsigned int *y, unsigned int *z) {
	*x = ModelBase::volume.dims.x; *y = ModelBase::volume.dims.y; *z = ModelBase::volume.dims.z;
}

const unsigned char *volr_ref_get_voxels(void) { return ModelBase::volume.data; }

void volr_ref_reset_transfer_fn(void) { RaycasterBase::reset_transfer_fn(); }

// Install a custom (non-premultiplied) 128x4 transfer function, then RaycasterBase::update_transfer_fn.
void volr_ref_set_base_transfer_fn(const float *rgba128) {
	for (int i = 0; i < TF_SIZE; i++)
		RaycasterBase::base_transfer_fn[i] = make_float4(rgba128[4*i], rgba128[4*i+1], rgba128[4*i+2], rgba128[4*i+3]);
	RaycasterBase::update_transfer_fn();
}

void volr_ref_get_transfer_fn(float *out128x4) {
	memcpy(out128x4, RaycasterBase::raycaster.transfer_fn, TF_SIZE * sizeof(float4));
}

void volr_ref_get_esl(unsigned int *out1024) {
	memcpy(out1024, RaycasterBase::raycaster.esl_volume, ESL_VOLUME_SIZE * sizeof(esl_type));
}

// out: ray_step, ray_threshold, light_kd, esl_block_size.xyz ; iout: esl (0/1), esl_block_dims
void volr_ref_get_params(float *out6, unsigned int *iout2) {
	const Raycaster &r = RaycasterBase::raycaster;
	out6[0] = r.ray_step; out6[1] = r.ray_threshold; out6[2] = r.light_kd;
	out6[3] = r.esl_block_size.x; out6[4] = r.esl_block_size.y; out6[5] = r.esl_block_size.z;
	iout2[0] = r.esl ? 1 : 0; iout2[1] = r.esl_block_dims;
}

// Direct field writes (the GLUI panel binds live variables to these fields, UI.cpp:516-523,535).
void volr_ref_set_params(float ray_step, float ray_threshold, float light_kd, int esl) {
	Raycaster &r = RaycasterBase::raycaster;
	r.ray_step = ray_step; r.ray_threshold = ray_threshold; r.light_kd = light_kd; r.esl = esl != 0;
}

// One frame: RaycasterBase::set_view + CPURenderer::render_volume (VolR.cpp:107-110).
// view15 = origin, direction, right_plane, up_plane, light_pos (3 floats each).
// Returns the renderer's own return code; *seconds = wall-clock of render_volume only.
int volr_ref_render(unsigned int w, unsigned int h, const float *view15, int perspective,
                    unsigned char *rgba_out, double *seconds) {
	View v;
	v.dims = make_ushort2(w, h);
	v.origin      = make_float3(view15[0],  view15[1],  view15[2]);
	v.direction   = make_float3(view15[3],  view15[4],  view15[5]);
	v.right_plane = make_float3(view15[6],  view15[7],  view15[8]);
	v.up_plane    = make_float3(view15[9],  view15[10], view15[11]);
	v.light_pos   = make_float3(view15[12], view15[13], view15[14]);
	v.perspective = perspective != 0;
	RaycasterBase::set_view(v);
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	int rc = g_cpu->render_volume((uchar4 *) rgba_out, RaycasterBase::raycaster);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	if (seconds) *seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
	return rc;
}

// ModelBase::histogram (ModelBase.cpp:19-33) of the loaded model
void volr_ref_get_histogram(float *out256) { memcpy(out256, ModelBase::histogram, 256 * sizeof(float)); }

// ddsbase.cpp quantize(): 16-bit big-endian samples -> 8 bit; the reference frees its input unless nofree
void volr_ref_quantize(const unsigned char *data16, unsigned int w, unsigned int h, unsigned int d, int linear, unsigned char *out8) {
	unsigned char *q = quantize((unsigned char *) data16, w, h, d, linear != 0, TRUE);
	memcpy(out8, q, (size_t) w * h * d);
	free(q);
}

// sizeof checks for the record (160 / 32 / 68 on LP64)
void volr_ref_sizes(unsigned int *out3) {
	out3[0] = sizeof(Raycaster); out3[1] = sizeof(Model); out3[2] = sizeof(View);
}

}
