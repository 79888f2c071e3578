// Microbenchmark: what it costs a wave to read the 8 corner voxels of a trilinear sample out of an LDS window of 1-BYTE voxels
// laid out as cell columns of 36 bytes (33 slices of the march axis + pad): 4 x ds_read_u16 at byte alignment (the slice pair
// (m, m+1) of the columns (u,v), (u+1,v), (u,v+1), (u+1,v+1)), against ONE ds_read2_b32 of 4-byte quad elements.
// Loads are inline asm (nothing hoisted); the slice index advances every iteration, so odd and even addresses alternate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int SU = 28, SV = 16, COL = 36;

template <int VARIANT>
__global__ __launch_bounds__(512) void k(uint32_t *out, int iters, const uint32_t *lane_col) {
	__shared__ __attribute__((aligned(16))) uint8_t win[SU * SV * COL + 256];
	for (int i = threadIdx.x; i < (SU * SV * COL + 256) / 4; i += 512) ((uint32_t *) win)[i] = i * 2654435761u;
	__syncthreads();
	const int lane = threadIdx.x & 63;
	const uint32_t base = (uint32_t) (uintptr_t) win + lane_col[lane] * (VARIANT != 1 ? COL : 4);
	uint32_t acc = 0, m = 0;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int u = 0; u < 8; u++) {
			if (VARIANT >= 2) {         // 2: u16 at even addresses only; 3: b32 at byte alignment; 4: b32 at 4-byte alignment
				uint32_t a, b, c, d;
				const uint32_t addr = base + (VARIANT == 2 ? (m & ~1u) : (VARIANT == 4 ? (m & ~3u) : m));
				if (VARIANT == 2)
					asm volatile("ds_read_u16 %0, %4\n\tds_read_u16 %1, %4 offset:36\n\tds_read_u16 %2, %4 offset:1008\n\tds_read_u16 %3, %4 offset:1044\n\ts_waitcnt lgkmcnt(0)"
					             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr));
				else
					asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:36\n\tds_read_b32 %2, %4 offset:1008\n\tds_read_b32 %3, %4 offset:1044\n\ts_waitcnt lgkmcnt(0)"
					             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr));
				acc += (a | b | c | d);
			} else if (VARIANT == 0) {
				uint32_t a, b, c, d;
				const uint32_t addr = base + m;
				asm volatile("ds_read_u16 %0, %4\n\tds_read_u16 %1, %4 offset:36\n\tds_read_u16 %2, %4 offset:1008\n\tds_read_u16 %3, %4 offset:1044\n\ts_waitcnt lgkmcnt(0)"
				             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr));
				acc += (a | b | c | d);
			} else {
				uint64_t ab;
				const uint32_t addr = base + m * 4u * SU * SV / 8u;     // some slice-dependent dword offset
				asm volatile("ds_read2_b32 %0, %1 offset1:28\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ab) : "v"(addr));
				acc += (uint32_t) ab | (uint32_t) (ab >> 32);
			}
			m = (m + 1) & 31;
		}
	}
	out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int VARIANT> static double run(uint32_t *out, uint32_t *d_col) {
	const int iters = 4000, blocks = 256 * 4;
	k<VARIANT><<<blocks, 512>>>(out, 10, d_col);
	hipDeviceSynchronize();
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	k<VARIANT><<<blocks, 512>>>(out, iters, d_col);
	hipEventRecord(e1);
	if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); exit(3); }
	float ms; hipEventElapsedTime(&ms, e0, e1);
	return ms * 1e6 / ((double) iters * 8 * 8 * 4);        // ns per wave-sample per CU (4 workgroups of 8 waves per CU)
}

int main() {
	setvbuf(stdout, NULL, _IONBF, 0);
	uint32_t *out, *d_col, h[64];
	hipMalloc(&out, 256 * 4 * 512 * 4); hipMalloc(&d_col, 256);
	const float pitches[] = { 0.375f, 0.5f, 0.75f, 1.125f };
	printf("%-40s %14s %14s %14s %14s %14s\n", "lane pattern (ns per wave-sample per CU)", "4 x u16 bytes", "1 x read2_b32", "4 x u16 even", "4 x b32 bytes", "4 x b32 al4");
	for (float pitch : pitches) for (int skew = 0; skew < 2; skew++) {
		for (int l = 0; l < 64; l++) {
			const int qd = l >> 4, i = (qd & 1) * 4 + (l & 3), j = (qd >> 1) * 4 + ((l >> 2) & 3);
			const int u = (int) (0.3f + i * pitch + (skew ? 0.35f * j : 0.0f)), v = (int) (0.6f + j * pitch);
			h[l] = (uint32_t) (v * SU + u);
		}
		hipMemcpy(d_col, h, sizeof h, hipMemcpyHostToDevice);
		char name[64]; snprintf(name, sizeof name, "8x8 px, %.3f cells/px%s", pitch, skew ? ", skewed" : "");
		printf("%-40s %14.2f %14.2f %14.2f %14.2f %14.2f\n", name, run<0>(out, d_col), run<1>(out, d_col), run<2>(out, d_col), run<3>(out, d_col), run<4>(out, d_col));
	}
	return 0;
}
