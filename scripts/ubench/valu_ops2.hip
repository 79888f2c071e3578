// Microbenchmark (round 4): issue cost of the integer-multiply, 64-bit and lane-select instructions of the DENSE path of the march
// kernels on gfx950 (8 waves per SIMD, blocks of 16 independent instructions of one kind, inline asm).  ns per SIMD-instruction; the
// plain fp32 operations cost ~1.0-1.2, most other 32-bit operations ~1.7 (profiles/r01_ubench_valu_ops.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed, uint64_t mask) {
	float r[16];
	uint64_t w[16];
	for (int i = 0; i < 16; i++) { r[i] = seed + threadIdx.x + i; w[i] = (uint64_t) (threadIdx.x + i) * 77u; }
	const float m = 1.0000001f;
	uint32_t sdst = 0;
	for (int it = 0; it < iters; it++) {
#define OP(i) \
		if (KIND == 0)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(m) : "vcc"); \
		if (KIND == 1)  asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(m), "s"(mask)); \
		if (KIND == 2)  asm volatile("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(r[i]) : "s"(mask)); \
		if (KIND == 3)  asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(m)); \
		if (KIND == 4)  asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(m)); \
		if (KIND == 5)  asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(r[i]), "v"(m) : "vcc"); \
		if (KIND == 6)  asm volatile("v_lshlrev_b64 %0, 8, %0" : "+v"(w[i])); \
		if (KIND == 7)  asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 15])); \
		if (KIND == 8)  asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(m)); \
		if (KIND == 9)  asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sdst) : "v"(r[i])); \
		if (KIND == 10) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sdst) : "v"(r[i])); \
		if (KIND == 11) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(r[i]) : "s"(m)); \
		if (KIND == 12) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(m)); \
		if (KIND == 13) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mask) : "v"(r[i]), "v"(m)); \
		if (KIND == 14) asm volatile("v_rsq_f32 %0, %0" : "+v"(r[i])); \
		if (KIND == 15) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i])); \
		if (KIND == 16) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3fc00000" : "+v"(r[i]) : "v"(m)); \
		if (KIND == 17) asm volatile("v_mul_f32_e64 %0, %0, -%1" : "+v"(r[i]) : "v"(m)); \
		if (KIND == 18) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i]) : "s"(m)); \
		if (KIND == 19) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i]));
		REP16(OP) REP16(OP) REP16(OP) REP16(OP)
#undef OP
	}
	float s = (float) sdst + (float) (mask & 1u); for (int i = 0; i < 16; i++) s += r[i] + (float) w[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND> void run(const char *name, float *out) {
	const int iters = 2000, blocks = 256 * 8;          // 8 waves per SIMD
	k<KIND><<<blocks, 256>>>(out, 10, 1.0f, 0x5555aaaa5555aaaaull);
	hipDeviceSynchronize();
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	k<KIND><<<blocks, 256>>>(out, iters, 1.0f, 0x5555aaaa5555aaaaull);
	hipEventRecord(e1);
	if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); exit(3); }
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double insts_per_simd = (double) iters * 64 * 8;
	printf("%-28s %6.3f ns per SIMD-instruction\n", name, ms * 1e6 / insts_per_simd);
}

int main() {
	setvbuf(stdout, NULL, _IONBF, 0);
	float *out; hipMalloc(&out, 256 * 2048 * 4);
	run<11>("v_fma_f32 (sgpr operand)", out);
	run<0>("v_cndmask_b32 vcc", out);
	run<1>("v_cndmask_b32 sgpr mask", out);
	run<2>("v_cndmask_b32 0, v, sgpr", out);
	run<3>("v_mul_hi_u32", out);
	run<4>("v_mul_lo_u32", out);
	run<5>("v_mad_u64_u32", out);
	run<6>("v_lshlrev_b64", out);
	run<7>("v_lshl_add_u64", out);
	run<8>("v_mul_u32_u24", out);
	run<9>("v_readlane_b32", out);
	run<10>("v_readfirstlane_b32", out);
	run<12>("v_and_or_b32", out);
	run<13>("v_cmp_lt_f32 -> sgpr", out);
	run<14>("v_rsq_f32", out);
	run<15>("v_rcp_f32", out);
	run<19>("v_sqrt_f32", out);
	run<16>("v_fmaak_f32", out);
	run<17>("v_mul_f32 (neg modifier)", out);
	run<18>("v_mov_b32 from sgpr", out);
	return 0;
}
