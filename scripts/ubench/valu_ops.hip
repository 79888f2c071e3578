// Microbenchmark: issue cost of the individual VALU instructions of the ray-march loop on gfx950 (8 waves per SIMD,
// blocks of 16 independent instructions of one kind, inline asm so that nothing is folded away).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
	float r[16];
	for (int i = 0; i < 16; i++) r[i] = seed + threadIdx.x + i;
	const float m = 1.0000001f;
	const unsigned sh = 2;
	for (int it = 0; it < iters; it++) {
#define OP(i) \
		if (KIND == 0)  asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(m)); \
		if (KIND == 1)  asm volatile("v_med3_f32 %0, %0, %1, 0" : "+v"(r[i]) : "s"(1e30f)); \
		if (KIND == 2)  asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r[i])); \
		if (KIND == 3)  asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(r[i])); \
		if (KIND == 4)  asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(m)); \
		if (KIND == 5)  asm volatile("v_cmp_ne_u32 vcc, 0, %0" :: "v"(r[i]) : "vcc"); \
		if (KIND == 6)  asm volatile("v_or_b32 %0, %1, %0" : "+v"(r[i]) : "s"(sh)); \
		if (KIND == 7)  asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r[i])); \
		if (KIND == 8)  asm volatile("v_sub_u32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0" : "+v"(r[i])); \
		if (KIND == 9)  asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(r[i]) : "v"(m)); \
		if (KIND == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(m) : "vcc"); \
		if (KIND == 11) asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(m)); \
		if (KIND == 12) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r[i]) : "v"(m)); \
		if (KIND == 13) asm volatile("v_fract_f32 %0, %0" : "+v"(r[i])); \
		if (KIND == 14) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(m)); \
		if (KIND == 15) asm volatile("v_min_i32 %0, %0, %1" : "+v"(r[i]) : "v"(m)); \
		if (KIND == 16) asm volatile("v_bfe_u32 %0, %0, 1, 7" : "+v"(r[i])); \
		if (KIND == 17) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(m)); \
		if (KIND == 18) asm volatile("v_cmp_le_f32 vcc, %0, %1" :: "v"(r[i]), "v"(m) : "vcc"); \
		if (KIND == 19) asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0xc8" : "+v"(r[i]) : "s"(sh)); \
		if (KIND == 20) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[i]) : "v"(m)); \
		if (KIND == 21) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r[i])); \
		if (KIND == 22) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i]) : "v"(m));
		REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
	}
	float s = 0; for (int i = 0; i < 16; i++) s += r[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND> void run(const char *name, float *out) {
	const int iters = 4000, blocks = 256 * 8;          // 8 waves per SIMD
	k<KIND><<<blocks, 256>>>(out, 10, 1.0f);
	hipDeviceSynchronize();
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	k<KIND><<<blocks, 256>>>(out, iters, 1.0f);
	hipEventRecord(e1); hipDeviceSynchronize();
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double insts_per_simd = (double) iters * 64 * 8;
	printf("%-22s %6.3f ns per SIMD-instruction\n", name, ms * 1e6 / insts_per_simd);
}

int main() {
	float *out; hipMalloc(&out, 256 * 2048 * 4);
	run<0>("v_fma_f32", out); run<17>("v_add_f32", out); run<14>("v_mul_f32", out); run<20>("v_sub_f32", out);
	run<1>("v_med3_f32", out); run<2>("v_cvt_i32_f32", out); run<21>("v_cvt_f32_i32", out); run<13>("v_fract_f32", out);
	run<7>("v_cvt_f32_ubyte1", out); run<8>("v_sub_u32_sdwa", out);
	run<3>("v_lshlrev_b32", out); run<4>("v_add_u32", out); run<6>("v_or_b32", out); run<19>("v_bitop3_b32", out);
	run<15>("v_min_i32", out); run<16>("v_bfe_u32", out); run<22>("v_mov_b32", out);
	run<9>("v_mad_u32_u24", out); run<11>("v_add3_u32", out); run<12>("v_lshl_add_u32", out);
	run<10>("v_cndmask_b32", out); run<5>("v_cmp_ne_u32", out); run<18>("v_cmp_le_f32", out);
	return 0;
}
