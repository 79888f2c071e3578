/*
 * vr_oracle.c — CPU restatement of the reference's ray-march path.  TEST INFRASTRUCTURE, NOT PRODUCT CODE
 * (see vr_oracle.h for who may use it and for the parity status).
 *
 * Every function cites the reference lines it follows; "VR/" = /root/reference/VolumeRendering/.
 * The float operation ORDER of the reference is kept expression by expression, and the file is built with
 * -ffp-contract=off: with that, NEAREST mode is bit-identical to the reference's CPURenderer (g++ -O2, SSE2).
 * The only fused operations are the explicit fmaf() calls of the TRILINEAR mode, which the reference's CPU
 * path does not have.
 */
#include "vr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;

/* VR/common.h:24-47 — component-wise helpers, same operand order */
static inline f3 f3_make(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f3 f3_scale(f3 a, float b) { return f3_make(a.x * b, a.y * b, a.z * b); }
static inline f3 f3_add(f3 a, f3 b) { return f3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 f3_sub(f3 a, f3 b) { return f3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 f3_div(f3 a, f3 b) { return f3_make(a.x / b.x, a.y / b.y, a.z / b.z); }
/* VR/common.h:88-96 */
static inline float flmin(float a, float b) { return a < b ? a : b; }
static inline float flmax(float a, float b) { return a > b ? a : b; }

/* VR/common.h:105-110 map_float_int: float <0,1> -> int <0,n-1>, truncation toward zero, clamped */
static inline uint32_t map_float_int(float f, uint32_t n) {
	int64_t i = (int64_t) (f * (float) n);
	if (i >= (int64_t) n) i = (int64_t) n - 1;
	if (i < 0) i = 0;
	return (uint32_t) i;
}

typedef struct {
	const vr_params *p;
	const uint8_t  *vox8;
	const uint16_t *vox16;
	uint32_t dx, dy, dz;
	uint32_t bpv;
	const f4 *tf;
	const uint32_t *esl;
	f3 min_bound;             /* VR/ModelBase.cpp:10-14: hard-coded (-1,-1,-1) */
	uint64_t *line_bits;      /* instrumentation only */
} scene;

static inline void touch(const scene *s, uint64_t voxel_index) {
	if (s->line_bits) {
		uint64_t line = (voxel_index * s->bpv) >> 7;
		__atomic_fetch_or(&s->line_bits[line >> 6], 1ull << (line & 63), __ATOMIC_RELAXED);
	}
}

static inline uint32_t fetch_raw(const scene *s, uint32_t ix, uint32_t iy, uint32_t iz) {
	uint64_t idx = ((uint64_t) iz * s->dy + iy) * s->dx + ix;
	touch(s, idx);
	return s->bpv == 1 ? s->vox8[idx] : s->vox16[idx];
}

/* VR/ModelBase.h:17-23 Model::sample_data — nearest voxel */
static inline uint32_t sample_nearest(const scene *s, f3 pos) {
	uint32_t iz = map_float_int((pos.z + 1) * 0.5f, s->dz);
	uint32_t iy = map_float_int((pos.y + 1) * 0.5f, s->dy);
	uint32_t ix = map_float_int((pos.x + 1) * 0.5f, s->dx);
	return fetch_raw(s, ix, iy, iz);
}

/* One axis of a CUDA linear-filtered, normalised, clamped texture fetch (semantics of the tex3D call at
 * VR/GPURenderer4.cu:76 with the texture set up at :136-141): xB = f*N - 0.5, i = floor(xB), a = frac(xB),
 * indices clamped to [0, N-1].  The caller supplies xB.  Deviation (documented in DESIGN.md): the weight keeps
 * full fp32 precision, the texture unit would quantise it to 8 fractional bits.
 *
 * TRILINEAR-mode arithmetic is DEFINED here with explicit fused multiply-adds (the GPU the reference's renderer 4
 * ran on contracts to FMA as well; no CPU run of it exists).  With f = (pos+1)/2 and pos = origin + dir*k:
 *     xB = f*N - 0.5 = pos*(N/2) + (N/2 - 0.5) = k * (dir*N/2) + fma(origin, N/2, N/2 - 0.5) = fma(k, A, B),
 * A and B computed once per ray.  The HIP kernel executes the same fmaf sequence, so the two agree bit for bit. */
static inline void axis_setup(float xb, uint32_t n, uint32_t *i0, uint32_t *i1, float *a, int q8) {
	float fl = floorf(xb);
	*a = xb - fl;
	if (q8)                        /* VR_SAMPLE_TRILINEAR_Q8: weight in 9-bit fixed point with 8 fractional bits (round to nearest even) */
		*a = rintf(*a * 256.0f) * (1.0f / 256.0f);
	int32_t i = (int32_t) fl;
	int32_t lo = i < 0 ? 0 : (i > (int32_t) n - 1 ? (int32_t) n - 1 : i);
	int32_t j = i + 1;
	int32_t hi = j < 0 ? 0 : (j > (int32_t) n - 1 ? (int32_t) n - 1 : j);
	*i0 = (uint32_t) lo; *i1 = (uint32_t) hi;
}

static inline float lerp(float a, float b, float t) { return fmaf(t, b - a, a); }

/* trilinear fetch at texel-space coordinates (xb,yb,zb); returns the interpolated RAW voxel value (0..255 or
 * 0..65535) — the normalisation of cudaReadModeNormalizedFloat (VR/GPURenderer4.cu:12) is folded into the constants
 * of its two consumers (transfer-function coordinate, shading difference). */
static inline float sample_trilinear_raw(const scene *s, float xb, float yb, float zb) {
	uint32_t x0, x1, y0, y1, z0, z1; float ax, ay, az;
	const int q8 = s->p->sampling == VR_SAMPLE_TRILINEAR_Q8;
	axis_setup(xb, s->dx, &x0, &x1, &ax, q8);
	axis_setup(yb, s->dy, &y0, &y1, &ay, q8);
	axis_setup(zb, s->dz, &z0, &z1, &az, q8);
	float v000 = (float) fetch_raw(s, x0, y0, z0), v100 = (float) fetch_raw(s, x1, y0, z0);
	float v010 = (float) fetch_raw(s, x0, y1, z0), v110 = (float) fetch_raw(s, x1, y1, z0);
	float v001 = (float) fetch_raw(s, x0, y0, z1), v101 = (float) fetch_raw(s, x1, y0, z1);
	float v011 = (float) fetch_raw(s, x0, y1, z1), v111 = (float) fetch_raw(s, x1, y1, z1);
	float c00 = lerp(v000, v100, ax), c10 = lerp(v010, v110, ax);
	float c01 = lerp(v001, v101, ax), c11 = lerp(v011, v111, ax);
	float c0 = lerp(c00, c10, ay), c1 = lerp(c01, c11, ay);
	return lerp(c0, c1, az);
}

/* linearly filtered TF fetch: tex1D(transfer_fn_texture, sample), VR/GPURenderer4.cu:77,91-99 (normalised
 * coordinate, clamp): xB = sample*128 - 0.5 with sample = raw/255  ==>  xB = fma(raw, 128/255, -0.5) */
static inline f4 tf_linear(const scene *s, float raw) {
	uint32_t i0, i1; float a;
	const float scale = s->bpv == 1 ? (float) VR_TF_SIZE / 255.0f : (float) VR_TF_SIZE / 65535.0f;
	axis_setup(fmaf(raw, scale, -0.5f), VR_TF_SIZE, &i0, &i1, &a, s->p->sampling == VR_SAMPLE_TRILINEAR_Q8);
	f4 c0 = s->tf[i0], c1 = s->tf[i1];
	f4 r = { lerp(c0.x, c1.x, a), lerp(c0.y, c1.y, a), lerp(c0.z, c1.z, a), lerp(c0.w, c1.w, a) };
	return r;
}

/* 1/sqrt(x) of the TRILINEAR-mode light vector: integer seed + three Newton steps, every operation a plain IEEE
 * fp32 op so that CPU and GPU agree bit for bit (relative error < 2e-7; the result only offsets the shading sample by
 * 0.01 units along the light direction, VR/GPURenderer4.cu:41-47). */
static inline float rsqrt_nr(float x) {
	union { float f; uint32_t u; } v;
	v.f = x;
	v.u = 0x5f3759dfu - (v.u >> 1);
	float y = v.f;
	const float h = 0.5f * x;
	y = y * fmaf(-(h * y), y, 1.5f);
	y = y * fmaf(-(h * y), y, 1.5f);
	y = y * fmaf(-(h * y), y, 1.5f);
	return y;
}

/* VR/ViewBase.h:23-35 View::get_ray */
static inline void get_ray(const vr_view *v, int px, int py, f3 *origin, f3 *direction) {
	f3 vo = f3_make(v->origin[0], v->origin[1], v->origin[2]);
	f3 vd = f3_make(v->direction[0], v->direction[1], v->direction[2]);
	f3 vr = f3_make(v->right_plane[0], v->right_plane[1], v->right_plane[2]);
	f3 vu = f3_make(v->up_plane[0], v->up_plane[1], v->up_plane[2]);
	float fx = (float) (px - (int) (v->width / 2));
	float fy = (float) (py - (int) (v->height / 2));
	if (v->perspective) {
		*origin = vo;
		*direction = f3_add(vd, f3_scale(vr, fx));
		*direction = f3_add(*direction, f3_scale(vu, fy));
	} else {
		*direction = vd;
		*origin = f3_add(vo, f3_scale(vr, fx));
		*origin = f3_add(*origin, f3_scale(vu, fy));
	}
}

/* VR/RaycasterBase.h:32-42 Raycaster::intersect */
static inline int intersect(const scene *s, f3 pt, f3 dir, float *kx, float *ky) {
	if (dir.x == 0) dir.x = 0.00001f;
	if (dir.y == 0) dir.y = 0.00001f;
	if (dir.z == 0) dir.z = 0.00001f;
	f3 k1 = f3_div(f3_sub(s->min_bound, pt), dir);
	f3 nb = f3_make(-s->min_bound.x, -s->min_bound.y, -s->min_bound.z);
	f3 k2 = f3_div(f3_sub(nb, pt), dir);
	*kx = flmax(flmax(flmin(k1.x, k2.x), flmin(k1.y, k2.y)), flmin(k1.z, k2.z));
	*ky = flmin(flmin(flmax(k1.x, k2.x), flmax(k1.y, k2.y)), flmax(k1.z, k2.z));
	*kx = flmax(*kx, 0);
	return (*kx < *ky) && (*ky > 0);
}

/* VR/RaycasterBase.h:52-65 Raycaster::sample_data_esl — bit set = block is empty */
static inline int sample_data_esl(const scene *s, f3 pos) {
	uint32_t bd = s->p->esl_block_dims;
	uint16_t index = (uint16_t) ((map_float_int((pos.z + 1) * 0.5f, s->dz) / bd) * VR_ESL_VOLUME_DIMS +
	                             (map_float_int((pos.y + 1) * 0.5f, s->dy) / bd));
	uint32_t sample = s->esl[index];
	index = (uint16_t) (map_float_int((pos.x + 1) * 0.5f, s->dx) / bd);
	return (sample & (1u << index)) != 0;
}

/* VR/RaycasterBase.h:67-85 Raycaster::leap_empty_space */
static inline void leap_empty_space(const scene *s, f3 pt, f3 dir, float *kx) {
	uint32_t bd = s->p->esl_block_dims;
	uint16_t ix = (uint16_t) (map_float_int((pt.x + 1) * 0.5f, s->dx) / bd);
	uint16_t iy = (uint16_t) (map_float_int((pt.y + 1) * 0.5f, s->dy) / bd);
	uint16_t iz = (uint16_t) (map_float_int((pt.z + 1) * 0.5f, s->dz) / bd);
	if (dir.x > 0) ix++;
	if (dir.y > 0) iy++;
	if (dir.z > 0) iz++;
	f3 bs = f3_make(s->p->esl_block_size[0], s->p->esl_block_size[1], s->p->esl_block_size[2]);
	f3 plane = f3_make(bs.x * (float) ix, bs.y * (float) iy, bs.z * (float) iz);   /* operator*(float3, ushort3) */
	f3 kp = f3_div(f3_sub(f3_add(s->min_bound, plane), pt), dir);
	if (dir.x == 0) kp.x = 100;
	if (dir.y == 0) kp.y = 100;
	if (dir.z == 0) kp.z = 100;
	float dk = flmin(kp.x, kp.y);
	dk = flmin(dk, kp.z);
	dk = flmax(dk, 0);
	dk = floorf(dk / s->p->ray_step) * s->p->ray_step;
	*kx += dk;
}

/* VR/common.h:62-69 vector_normalize = v * (1.0f / sqrtf(x*x + y*y + z*z)) */
static inline f3 vector_normalize(f3 v) {
	float len = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
	return f3_scale(v, 1.0f / len);
}

typedef struct { uint64_t rays_hit, esl_probes, samples, shade_fetches; } counters;

/* pt = origin + direction * k (VR/CPURenderer.cpp:17,24,38): two roundings in NEAREST mode like the reference's
 * CPU build, one fused rounding in TRILINEAR mode (see axis_setup) */
static inline f3 march_point(int fused, f3 origin, f3 direction, float k) {
	if (fused)
		return f3_make(fmaf(direction.x, k, origin.x), fmaf(direction.y, k, origin.y), fmaf(direction.z, k, origin.z));
	return f3_add(origin, f3_scale(direction, k));
}

/* ---- VRO_SAMPLE_TRILINEAR_F64: the TRILINEAR model evaluated in double precision -----------------------------------------
 * Same rays, same fp32 k sequence (k += ray_step in float, VR/CPURenderer.cpp:37), same cell / weight definition
 * (xB = pos * N/2 + N/2 - 1/2, clamp addressing, GPURenderer4.cu:76-77,136-141), but every product, lerp, square root and
 * composite in IEEE double.  Used by tests only, to bound the fp32 rounding error of the product's TRILINEAR arithmetic (and,
 * against VR_SAMPLE_TRILINEAR_Q8, the effect of 8-bit filter weights) independently of the restatement's own fmaf sequence. */
static inline double tri_f64(const scene *s, double xb, double yb, double zb) {
	double c[3] = { xb, yb, zb }, a[3];
	uint32_t lo[3], hi[3];
	const uint32_t n[3] = { s->dx, s->dy, s->dz };
	for (int i = 0; i < 3; i++) {
		double fl = floor(c[i]);
		a[i] = c[i] - fl;
		long j = (long) fl, m = (long) n[i] - 1;
		lo[i] = (uint32_t) (j < 0 ? 0 : (j > m ? m : j));
		hi[i] = (uint32_t) (j + 1 < 0 ? 0 : (j + 1 > m ? m : j + 1));
	}
	double v[2][2][2];
	for (int z = 0; z < 2; z++) for (int y = 0; y < 2; y++) for (int x = 0; x < 2; x++)
		v[z][y][x] = (double) fetch_raw(s, x ? hi[0] : lo[0], y ? hi[1] : lo[1], z ? hi[2] : lo[2]);
	double c00 = v[0][0][0] + a[0] * (v[0][0][1] - v[0][0][0]), c10 = v[0][1][0] + a[0] * (v[0][1][1] - v[0][1][0]);
	double c01 = v[1][0][0] + a[0] * (v[1][0][1] - v[1][0][0]), c11 = v[1][1][0] + a[0] * (v[1][1][1] - v[1][1][0]);
	double c0 = c00 + a[1] * (c10 - c00), c1 = c01 + a[1] * (c11 - c01);
	return c0 + a[2] * (c1 - c0);
}

static void render_ray_f64(const scene *s, int px, int py, uint8_t *out_px, counters *c) {
	const vr_params *p = s->p;
	f3 origin, direction;
	float kx, ky;
	get_ray(&p->view, px, py, &origin, &direction);
	if (!intersect(s, origin, direction, &kx, &ky))
		return;
	c->rays_hit++;
	while (kx <= ky) {                                   /* empty space leaping: the reference's fp32 loop, unchanged */
		f3 pt = march_point(1, origin, direction, kx);
		c->esl_probes++;
		if (p->esl && sample_data_esl(s, pt))
			leap_empty_space(s, pt, direction, &kx);
		else
			break;
		kx += p->ray_step;
	}
	if (kx > ky)
		return;
	const double o[3] = { origin.x, origin.y, origin.z }, d[3] = { direction.x, direction.y, direction.z };
	const double half[3] = { 0.5 * s->dx, 0.5 * s->dy, 0.5 * s->dz };
	const double lp[3] = { p->view.light_pos[0], p->view.light_pos[1], p->view.light_pos[2] };
	const double raw_scale = s->bpv == 1 ? 255.0 : 65535.0;
	double acc[4] = { 0, 0, 0, 0 };
	while (kx <= ky) {
		c->samples++;
		double pos[3], tb[3];
		for (int i = 0; i < 3; i++) { pos[i] = o[i] + d[i] * (double) kx; tb[i] = pos[i] * half[i] + half[i] - 0.5; }
		const double raw = tri_f64(s, tb[0], tb[1], tb[2]);
		double t = raw / raw_scale * VR_TF_SIZE - 0.5;
		if (t < 0) t = 0;
		if (t > VR_TF_SIZE - 1) t = VR_TF_SIZE - 1;
		const int i0 = (int) floor(t), i1 = i0 + 1 < VR_TF_SIZE ? i0 + 1 : i0;
		const double w = t - floor(t);
		const f4 t0 = s->tf[i0], t1 = s->tf[i1];
		double cur[4] = { t0.x + w * ((double) t1.x - t0.x), t0.y + w * ((double) t1.y - t0.y),
		                  t0.z + w * ((double) t1.z - t0.z), t0.w + w * ((double) t1.w - t0.w) };
		if (cur[3] > 0.05 && p->light_kd > 0.01f) {
			double l[3] = { lp[0] - pos[0], lp[1] - pos[1], lp[2] - pos[2] };
			const double inv = 1.0 / sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
			const double raw_l = tri_f64(s, tb[0] + l[0] * inv * 0.01 * half[0], tb[1] + l[1] * inv * 0.01 * half[1],
			                             tb[2] + l[2] * inv * 0.01 * half[2]);
			const double diffuse = (raw_l - raw) / raw_scale * (double) p->light_kd;
			cur[0] += diffuse; cur[1] += diffuse; cur[2] += diffuse;
			c->shade_fetches++;
		}
		const double tr = 1.0 - acc[3];
		for (int i = 0; i < 4; i++) acc[i] += cur[i] * tr;
		if (acc[3] > (double) p->ray_threshold)
			break;
		kx += p->ray_step;
	}
	for (int i = 0; i < 4; i++) {
		long q = (long) (acc[i] * 256.0);
		out_px[i] = (uint8_t) (q < 0 ? 0 : (q > 255 ? 255 : q));
	}
}

/* VR/CPURenderer.cpp:11-41 render_ray (NEAREST) and VR/GPURenderer4.cu:53-87 (TRILINEAR) */
static void render_ray(const scene *s, int px, int py, uint8_t *out_px, counters *c) {
	const vr_params *p = s->p;
	if (p->sampling == VRO_SAMPLE_TRILINEAR_F64) {
		render_ray_f64(s, px, py, out_px, c);
		return;
	}
	f3 origin, direction;
	float kx, ky;
	get_ray(&p->view, px, py, &origin, &direction);
	if (!intersect(s, origin, direction, &kx, &ky))
		return;
	c->rays_hit++;
	const int fused = p->sampling != VR_SAMPLE_NEAREST;
	f3 pt = march_point(fused, origin, direction, kx);
	while (kx <= ky) {                                   /* empty space leaping loop */
		c->esl_probes++;
		if (p->esl && sample_data_esl(s, pt))
			leap_empty_space(s, pt, direction, &kx);
		else
			break;
		kx += p->ray_step;
		pt = march_point(fused, origin, direction, kx);
	}
	if (kx > ky)
		return;
	f4 acc = { 0, 0, 0, 0 };
	const f3 half = f3_make(0.5f * (float) s->dx, 0.5f * (float) s->dy, 0.5f * (float) s->dz);
	const f3 A = f3_make(direction.x * half.x, direction.y * half.y, direction.z * half.z);
	const f3 B = f3_make(fmaf(origin.x, half.x, half.x - 0.5f), fmaf(origin.y, half.y, half.y - 0.5f),
	                     fmaf(origin.z, half.z, half.z - 0.5f));
	f3 light_pos = f3_make(p->view.light_pos[0], p->view.light_pos[1], p->view.light_pos[2]);
	const float raw_scale = s->bpv == 1 ? 255.0f : 65535.0f;
	while (kx <= ky) {                                   /* colour accumulation loop */
		c->samples++;
		f4 cur;
		if (p->sampling == VR_SAMPLE_NEAREST) {
			uint32_t sample = sample_nearest(s, pt);
			uint32_t s8 = s->bpv == 1 ? sample : (sample >> 8);
			cur = s->tf[s8 / VR_TF_RATIO];
			if (cur.w > 0.05f && p->light_kd > 0.01f) {  /* VR/RaycasterBase.h:87-98 Raycaster::shade */
				f3 light_dir = vector_normalize(f3_sub(light_pos, pt));
				float sample_l = (float) sample_nearest(s, f3_add(pt, f3_scale(light_dir, 0.01f))) / raw_scale;
				float diffuse = (sample_l - (float) sample / raw_scale) * p->light_kd;
				cur.x += diffuse; cur.y += diffuse; cur.z += diffuse;
				c->shade_fetches++;
			}
		} else {
			const float xb = fmaf(kx, A.x, B.x), yb = fmaf(kx, A.y, B.y), zb = fmaf(kx, A.z, B.z);
			float raw = sample_trilinear_raw(s, xb, yb, zb);  /* VR/GPURenderer4.cu:76 */
			cur = tf_linear(s, raw);                           /* VR/GPURenderer4.cu:77 */
			if (cur.w > 0.05f && p->light_kd > 0.01f) {        /* VR/GPURenderer4.cu:41-51 shade_texture */
				f3 d = f3_sub(light_pos, pt);
				float inv = rsqrt_nr(fmaf(d.z, d.z, fmaf(d.y, d.y, d.x * d.x)));
				f3 light_dir = f3_scale(d, inv);
				/* texel coordinate of pos + 0.01*light_dir = xB + light_dir * (0.01 * N/2) */
				float raw_l = sample_trilinear_raw(s, fmaf(light_dir.x, 0.01f * half.x, xb),
				                                      fmaf(light_dir.y, 0.01f * half.y, yb),
				                                      fmaf(light_dir.z, 0.01f * half.z, zb));
				float diffuse = (raw_l - raw) * (p->light_kd * (s->bpv == 1 ? (1.0f / 255.0f) : (1.0f / 65535.0f)));
				cur.x += diffuse; cur.y += diffuse; cur.z += diffuse;
				c->shade_fetches++;
			}
		}
		float t = 1 - acc.w;                             /* C_out = C_in + C * (1 - alpha_in) */
		if (fused) {
			acc.x = fmaf(cur.x, t, acc.x); acc.y = fmaf(cur.y, t, acc.y);
			acc.z = fmaf(cur.z, t, acc.z); acc.w = fmaf(cur.w, t, acc.w);
		} else {
			acc.x = acc.x + cur.x * t; acc.y = acc.y + cur.y * t;
			acc.z = acc.z + cur.z * t; acc.w = acc.w + cur.w * t;
		}
		if (acc.w > p->ray_threshold)                    /* early ray termination */
			break;
		kx += p->ray_step;
		pt = march_point(fused, origin, direction, kx);
	}
	/* VR/RaycasterBase.h:44-50 write_color */
	out_px[0] = (uint8_t) map_float_int(acc.x, 256);
	out_px[1] = (uint8_t) map_float_int(acc.y, 256);
	out_px[2] = (uint8_t) map_float_int(acc.z, 256);
	out_px[3] = (uint8_t) map_float_int(acc.w, 256);
}

/* VR/CPURenderer.cpp:43-53 CPURenderer::render_volume */
int vro_render(const vr_params *p, const void *voxels, const uint32_t dims[3], uint32_t bytes_per_voxel,
               const float *tf_premult, const uint32_t *esl_bits, uint8_t *rgba_out, int threads,
               vro_stats *stats, int count_lines) {
	if (p == NULL || voxels == NULL || tf_premult == NULL || esl_bits == NULL || rgba_out == NULL)
		return 1;
	if (bytes_per_voxel != 1 && bytes_per_voxel != 2)
		return 1;
	if (p->band_rows == 0 || p->band_stride == 0)
		return 1;
	scene s;
	s.p = p; s.vox8 = (const uint8_t *) voxels; s.vox16 = (const uint16_t *) voxels;
	s.dx = dims[0]; s.dy = dims[1]; s.dz = dims[2]; s.bpv = bytes_per_voxel;
	s.tf = (const f4 *) tf_premult; s.esl = esl_bits;
	s.min_bound = f3_make(-1, -1, -1);
	s.line_bits = NULL;
	uint64_t nlines = (((uint64_t) s.dx * s.dy * s.dz * s.bpv) >> 7) + 1;
	if (count_lines)
		s.line_bits = (uint64_t *) calloc((nlines >> 6) + 1, sizeof(uint64_t));

	memset(rgba_out, 0, (size_t) p->out_width * p->out_rows * 4);
	uint64_t rays_hit = 0, esl_probes = 0, samples = 0, shade_fetches = 0;
	const int rows = (int) p->out_rows;
	#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 1 ? threads : 1) \
	        reduction(+:rays_hit, esl_probes, samples, shade_fetches)
	for (int ly = 0; ly < rows; ly++) {
		uint32_t gy = ((uint32_t) ly / p->band_rows * p->band_stride + p->band_first) * p->band_rows + (uint32_t) ly % p->band_rows;
		if (gy >= p->view.height) continue;
		counters c = { 0, 0, 0, 0 };
		for (uint32_t lx = 0; lx < p->out_width; lx++) {
			uint32_t gx = p->x0 + lx;
			if (gx >= p->view.width) continue;
			render_ray(&s, (int) gx, (int) gy, rgba_out + ((size_t) ly * p->out_width + lx) * 4, &c);
		}
		rays_hit += c.rays_hit; esl_probes += c.esl_probes; samples += c.samples; shade_fetches += c.shade_fetches;
	}
	if (stats) {
		stats->rays_hit = rays_hit; stats->esl_probes = esl_probes; stats->samples = samples;
		stats->shade_fetches = shade_fetches; stats->lines_touched = 0;
		if (s.line_bits)
			for (uint64_t w = 0; w < (nlines >> 6) + 1; w++)
				stats->lines_touched += (uint64_t) __builtin_popcountll(s.line_bits[w]);
	}
	free(s.line_bits);
	return 0;
}

/* VR/RaycasterBase.cpp:76-84 RaycasterBase::reset_transfer_fn — the default base TF (before premultiplication) */
void vro_default_base_tf(float *base) {
	const int n = VR_TF_SIZE;
	for (int i = 0; i < n; i++) {
		base[4*i+0] = i <= n/3 ? (float) (i*3) / (float) n : 0.0f;
		base[4*i+1] = (i > n/3) && (i <= n/3*2) ? (float) ((i - n/3)*3) / (float) n : 0.0f;
		base[4*i+2] = i > n/3*2 ? (float) ((i - n/3*2)*3) / (float) n : 0.0f;
		base[4*i+3] = (float) i > ((255.0f * 0.1f) / VR_TF_RATIO) ? (float) i / (float) n : 0.0f;
	}
}

/* VR/RaycasterBase.cpp:46-74 RaycasterBase::update_transfer_fn */
void vro_update_transfer_fn(const float *base, const uint8_t *minmax, float *tf, uint32_t *esl) {
	for (int i = 0; i < VR_TF_SIZE; i++) {
		tf[4*i+0] = base[4*i+0] * base[4*i+3];
		tf[4*i+1] = base[4*i+1] * base[4*i+3];
		tf[4*i+2] = base[4*i+2] * base[4*i+3];
		tf[4*i+3] = base[4*i+3];
	}
	uint16_t first_visible[VR_TF_SIZE];     /* esl_temp_tf: first index >= x with non-zero opacity */
	for (int x = 0; x < VR_TF_SIZE; x++) {
		int y;
		for (y = x; y < VR_TF_SIZE; y++)
			if (tf[4*y+3] != 0) break;
		first_visible[x] = (uint16_t) y;
	}
	memset(esl, 0, VR_ESL_VOLUME_SIZE * sizeof(uint32_t));
	for (uint32_t i = 0; i < 32u * 32u * 32u; i++) {
		uint8_t mn = minmax[2*i], mx = minmax[2*i+1];
		if (first_visible[mn / VR_TF_RATIO] > mx / VR_TF_RATIO)
			esl[i / 32] |= 1u << (i % 32);
	}
}

/* VR/RaycasterBase.cpp:94-122 RaycasterBase::set_volume: block dims, per-block min/max, block size */
void vro_volume_minmax(const void *voxels, const uint32_t dims[3], uint32_t bpv,
                       uint8_t *minmax, uint32_t *block_dims_out, float *block_size_out) {
	uint32_t max_dim = dims[0] > dims[1] ? dims[0] : dims[1];
	if (dims[2] > max_dim) max_dim = dims[2];
	uint32_t bd = (max_dim + VR_ESL_VOLUME_DIMS - 1) / VR_ESL_VOLUME_DIMS;
	if (bd < VR_ESL_MIN_BLOCK) bd = VR_ESL_MIN_BLOCK;
	for (uint32_t i = 0; i < 32u * 32u * 32u; i++) { minmax[2*i] = 255; minmax[2*i+1] = 0; }
	const uint8_t *v8 = (const uint8_t *) voxels; const uint16_t *v16 = (const uint16_t *) voxels;
	for (uint32_t z = 0; z < dims[2]; z++)
		for (uint32_t y = 0; y < dims[1]; y++) {
			uint64_t row = ((uint64_t) z * dims[1] + y) * dims[0];
			uint32_t base = (z / bd) * 1024 + (y / bd) * 32;
			for (uint32_t x = 0; x < dims[0]; x++) {
				uint8_t sample = bpv == 1 ? v8[row + x] : (uint8_t) (v16[row + x] >> 8);
				uint32_t e = base + x / bd;
				if (minmax[2*e] > sample) minmax[2*e] = sample;
				if (minmax[2*e+1] < sample) minmax[2*e+1] = sample;
			}
		}
	*block_dims_out = bd;
	block_size_out[0] = 2.0f * (float) bd / (float) dims[0];
	block_size_out[1] = 2.0f * (float) bd / (float) dims[1];
	block_size_out[2] = 2.0f * (float) bd / (float) dims[2];
}

/* VR/RaycasterBase.cpp:86-92 RaycasterBase::reset_ray_step */
float vro_default_ray_step(const uint32_t dims[3]) {
	int max_dim = (int) (dims[0] > dims[1] ? dims[0] : dims[1]);
	if ((int) dims[2] > max_dim) max_dim = (int) dims[2];
	float ray_step = 2.0f / (float) max_dim;
	ray_step -= ray_step / (float) max_dim;
	return ray_step;
}

/* VR/ModelBase.cpp:19-26 raw histogram counts */
void vro_histogram(const void *voxels, uint64_t count, uint32_t bpv, uint64_t *hist) {
	memset(hist, 0, 256 * sizeof(uint64_t));
	const uint8_t *v8 = (const uint8_t *) voxels; const uint16_t *v16 = (const uint16_t *) voxels;
	for (uint64_t i = 0; i < count; i++)
		hist[bpv == 1 ? v8[i] : (v16[i] >> 8)]++;
}

/* murmur3 finaliser */
static inline uint32_t fmix32(uint32_t h) {
	h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
	return h;
}

/* deterministic integer-only synthetic volumes */
void vro_generate_volume(uint32_t kind, uint32_t n, uint32_t seed, uint32_t bpv, void *out) {
	uint8_t *o8 = (uint8_t *) out; uint16_t *o16 = (uint16_t *) out;
	const int64_t N = n;
	#pragma omp parallel for schedule(static)
	for (int64_t z = 0; z < N; z++)
		for (int64_t y = 0; y < N; y++)
			for (int64_t x = 0; x < N; x++) {
				uint64_t idx = ((uint64_t) z * (uint64_t) N + (uint64_t) y) * (uint64_t) N + (uint64_t) x;
				uint32_t h = fmix32((uint32_t) (idx ^ (idx >> 32)) + seed * 0x9E3779B9u);
				uint32_t v;
				if (kind == 0) {
					int64_t ax = 2 * x + 1 - N, ay = 2 * y + 1 - N, az = 2 * z + 1 - N;
					int64_t d2 = ax * ax + ay * ay + az * az;
					int64_t t = 1000 * d2 / (N * N) - 360;
					if (t < 0) t = -t;
					int64_t shell = 255 - t * 255 / 240;
					if (shell < 0) shell = 0;
					v = (uint32_t) shell + (h & 15);
					if (v > 255) v = 255;
				} else {
					v = h & 255;
				}
				if (bpv == 1) o8[idx] = (uint8_t) v; else o16[idx] = (uint16_t) (v * 257);
			}
}

uint32_t vro_fnv1a32(const void *data, uint64_t bytes) {
	const uint8_t *p = (const uint8_t *) data;
	uint32_t h = 2166136261u;
	for (uint64_t i = 0; i < bytes; i++) { h ^= p[i]; h *= 16777619u; }
	return h;
}
