// Microbenchmark for the column march (round 4): what ONE wave-gather costs the L1 / address path per CU when the four lanes of a
// lane quad read the SAME aligned 16-byte window (2x2 pixels of one cell column), against the shapes the product uses today.
//   "dword chunk"   : 4-byte loads, the quad's four addresses are the four dwords of one aligned 16-byte chunk (quad bricks, aligned view)
//   "x4 same"       : 16-byte loads, the quad's four lanes read the SAME aligned 16 bytes; the wave's 16 windows are contiguous (256 B)
//   "x4 straddle"   : 16-byte loads, lanes 0,1 of a quad one window, lanes 2,3 the next (quad straddles two columns)
//   "x4 distinct"   : 16-byte loads, 64 different windows, contiguous (1 KiB per wave-load)
//   "x4 col64"      : 16-byte loads, 64 different windows, each in its own 64-byte block (scattered columns)
//   "x2 same"       : 8-byte loads, the quad's four lanes read the same 8 bytes
//   "x2 run"        : 8-byte loads at 4-byte alignment, 36-byte runs, 2x2-px quads share a run (run bricks on an aligned view)
//   "dword same"    : 4-byte loads, the quad's lanes read the same dword
// Data is L1 / L2 resident (16 KiB window per wave); figure = ns per wave-instruction per CU with 32 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <csignal>

template <int W>
__global__ __launch_bounds__(256) void k(const uint8_t *buf, uint32_t *out, int iters, const uint32_t *lane_off) {
	const int lane = threadIdx.x & 63;
	uint32_t off = lane_off[lane] + (threadIdx.x >> 6) * 16384 + (blockIdx.x & 7) * 65536;
	uint32_t acc = 0, walk = 0;
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int u = 0; u < 8; u++) {
			const uint8_t *p = buf + off + walk + u * 128 * 1024;
			if (W == 4) acc ^= *(const uint32_t *) p;
			else if (W == 16) { const uint4 v = *(const uint4 *) p; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
			else { const uint2 v = *(const uint2 *) p; acc ^= v.x ^ v.y; }
		}
		walk = (walk + 4096) & 8191;
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int W>
static double run(const uint8_t *buf, uint32_t *out, uint32_t *d_off, const uint32_t *h_off) {
	hipMemcpy(d_off, h_off, 64 * 4, hipMemcpyHostToDevice);
	const int iters = 2000, blocks = 256 * 8;
	k<W><<<blocks, 256>>>(buf, out, 50, d_off);
	hipDeviceSynchronize();
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	k<W><<<blocks, 256>>>(buf, out, iters, d_off);
	hipEventRecord(e1);
	hipError_t err = hipDeviceSynchronize();
	if (err != hipSuccess) { printf("HIP error: %s\n", hipGetErrorString(err)); exit(3); }
	float ms; hipEventElapsedTime(&ms, e0, e1);
	return ms * 1e6 / ((double) iters * 8 * 4 * 8);      // 8 blocks of 4 waves per CU, 8 loads per iteration
}

int main() {
	signal(SIGPIPE, SIG_IGN); setvbuf(stdout, NULL, _IONBF, 0);
	uint8_t *buf; uint32_t *out, *d_off; uint32_t h[64];
	hipMalloc(&buf, 4 << 20); hipMemset(buf, 1, 4 << 20); hipMalloc(&out, 256 * 2048 * 4); hipMalloc(&d_off, 256);
	auto quad = [](int l) { return l >> 2; };
	for (int l = 0; l < 64; l++) h[l] = quad(l) * 16 + (l & 3) * 4;
	printf("%-14s %7.2f\n", "dword chunk", run<4>(buf, out, d_off, h));
	for (int l = 0; l < 64; l++) h[l] = quad(l) * 16;
	printf("%-14s %7.2f\n", "dword same", run<4>(buf, out, d_off, h));
	printf("%-14s %7.2f\n", "x4 same", run<16>(buf, out, d_off, h));
	for (int l = 0; l < 64; l++) h[l] = (quad(l) + ((l >> 1) & 1)) * 16;
	printf("%-14s %7.2f\n", "x4 straddle", run<16>(buf, out, d_off, h));
	for (int l = 0; l < 64; l++) h[l] = l * 16;
	printf("%-14s %7.2f\n", "x4 distinct", run<16>(buf, out, d_off, h));
	for (int l = 0; l < 64; l++) h[l] = l * 64;
	printf("%-14s %7.2f\n", "x4 col64", run<16>(buf, out, d_off, h));
	for (int l = 0; l < 64; l++) h[l] = quad(l) * 64;
	printf("%-14s %7.2f\n", "x4 same/64", run<16>(buf, out, d_off, h));
	for (int l = 0; l < 64; l++) h[l] = quad(l) * 16;
	printf("%-14s %7.2f\n", "x2 same", run<8>(buf, out, d_off, h));
	for (int l = 0; l < 64; l++) h[l] = quad(l) * 36 + 12;
	printf("%-14s %7.2f\n", "x2 run", run<8>(buf, out, d_off, h));
	for (int l = 0; l < 64; l++) h[l] = quad(l) * 36 + 12;
	printf("%-14s %7.2f\n", "x4 run36 (4B aligned)", run<16>(buf, out, d_off, h));
	// half the lanes (EXEC-masked gathers are not measured here: all lanes active)
	return 0;
}
