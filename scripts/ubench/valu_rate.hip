// Microbenchmark: VALU issue rate per SIMD on gfx950 for the instruction kinds of the ray-march loop.
// Each wave runs N iterations of an unrolled block of independent ops; we report SIMD cycles per wave-instruction
// (s_memtime) at 1, 2, 4, 8 waves per SIMD.  Tuning aid only (not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float seed) {
	float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	const float m = 1.0000001f, c = 1e-7f;
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int u = 0; u < 8; u++) {
			if (KIND == 0) { // v_fma_f32
				a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
				a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
			} else if (KIND == 1) { // v_pk_fma_f32
				typedef float f2 __attribute__((ext_vector_type(2)));
				f2 x0 = {a0, a1}, x1 = {a2, a3}, x2 = {a4, a5}, x3 = {a6, a7}, mm = {m, m}, cc = {c, c};
				x0 = __builtin_elementwise_fma(x0, mm, cc); x1 = __builtin_elementwise_fma(x1, mm, cc);
				x2 = __builtin_elementwise_fma(x2, mm, cc); x3 = __builtin_elementwise_fma(x3, mm, cc);
				x0 = __builtin_elementwise_fma(x0, mm, cc); x1 = __builtin_elementwise_fma(x1, mm, cc);
				x2 = __builtin_elementwise_fma(x2, mm, cc); x3 = __builtin_elementwise_fma(x3, mm, cc);
				a0 = x0.x; a1 = x0.y; a2 = x1.x; a3 = x1.y; a4 = x2.x; a5 = x2.y; a6 = x3.x; a7 = x3.y;
			} else if (KIND == 2) { // cvt_i32 / fract / med3 mix
				a0 = __builtin_amdgcn_fractf(a0) + 0.f; a1 = __builtin_amdgcn_fmed3f(a1, 0.f, 1e30f); a2 = (float) (int) a2; a3 = __builtin_amdgcn_fractf(a3);
				a4 = __builtin_amdgcn_fmed3f(a4, 0.f, 1e30f); a5 = (float) (int) a5; a6 = __builtin_amdgcn_fractf(a6); a7 = __builtin_amdgcn_fmed3f(a7, 0.f, 1e30f);
			} else if (KIND == 3) { // v_add_f32
				a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c;
			} else if (KIND == 4) { // integer add3 / lshl_add
				unsigned b0 = __float_as_uint(a0), b1 = __float_as_uint(a1), b2 = __float_as_uint(a2), b3 = __float_as_uint(a3);
				b0 = (b0 << 2) + b1; b1 = (b1 << 2) + b2; b2 = (b2 << 2) + b3; b3 = (b3 << 2) + b0;
				b0 = b0 + b1 + b2; b1 = b1 + b2 + b3; b2 = b2 + b3 + b0; b3 = b3 + b0 + b1;
				a0 = __uint_as_float(b0); a1 = __uint_as_float(b1); a2 = __uint_as_float(b2); a3 = __uint_as_float(b3);
			}
		}
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
	if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND> void run(const char *name, int ops_per_iter) {
	float *out; unsigned long long *cyc;
	hipMalloc(&out, 256 * 2048 * 4); hipMalloc(&cyc, 2048 * 4 * 8);
	const int iters = 20000;
	for (int wps = 1; wps <= 8; wps *= 2) {            // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
		int blocks = 256 * wps;                         // 256 CUs x wps blocks/CU
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		k<KIND><<<blocks, 256>>>(out, cyc, 100, 1.0f);
		hipDeviceSynchronize();
		hipEventRecord(e0);
		k<KIND><<<blocks, 256>>>(out, cyc, iters, 1.0f);
		hipEventRecord(e1); hipDeviceSynchronize();
		float ms; hipEventElapsedTime(&ms, e0, e1);
		std::vector<unsigned long long> h(blocks * 4);
		hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
		double mean = 0; for (auto v : h) mean += v; mean /= h.size();
		double insts = (double) iters * ops_per_iter;   // per wave
		// s_memtime ticks at shader clock?  report both cycles/instr (per wave) and wall ns/instr per SIMD
		printf("%-14s waves/SIMD=%d  memtime ticks per wave-instr=%.3f  => per SIMD %.3f ticks/instr ; wall %.3f ns per SIMD-instr (%.2f ms)\n",
		       name, wps, mean / insts, mean / insts / wps, ms * 1e6 / (insts * wps), ms);
	}
	hipFree(out); hipFree(cyc);
}

int main() {
	run<0>("v_fma_f32", 64);
	run<1>("v_pk_fma_f32", 64);     // 64 pk instructions = 128 fmas
	run<2>("fract/med3/cvt", 64 + 16 + 8);
	run<3>("v_add_f32", 64);
	run<4>("int lshl_add/add3", 64);
	return 0;
}
