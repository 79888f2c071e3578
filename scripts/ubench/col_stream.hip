// Microbenchmark for the column march (round 4): what HBM rate the march's READ PATTERN reaches, for three orders of the same 5.7 GB
// of 256-byte pieces (one piece = the 16-byte windows of the 4x4 cell columns of one lateral block at one window index w):
//   A  [bv][bu][w]   the product's order: every wave streams its own 87-KB run, 8192 resident waves = 8192 streams 87 KB apart
//   B  [w][bv][bu]   window-major: the waves of the chip, all near the same w, read one contiguous region
//   C  [bv][w][bu]   a row of blocks per w contiguous (64 KiB), rows 22 MB apart
//   D  [bv/2][w][bv%2][bu]  what ONE workgroup (4x2 waves) reads at a step is two 1-KiB pieces 64 KiB apart -> here one 128-KiB row pair
// Each wave = one lateral block (bu, bv) as in the product (workgroup = 4x2 blocks, workgroups numbered in 8x8 blocks of the frame), reads its
// W = 341 pieces in order with DEPTH loads in flight (the product: 3 ahead), 8 waves per SIMD.  Prints ms and TB/s per order and depth.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <csignal>

constexpr uint32_t NB = 256, W = 341;        // 256 x 256 lateral blocks, 341 windows: 1024^3 voxels as column windows

template <int ORDER>
__device__ __forceinline__ uint64_t piece(uint32_t bu, uint32_t bv, uint32_t w) {
	if (ORDER == 0) return ((uint64_t) (bv * NB + bu) * W + w) * 256u;
	if (ORDER == 1) return ((uint64_t) (w * NB + bv) * NB + bu) * 256u;
	if (ORDER == 2) return ((uint64_t) (bv * W + w) * NB + bu) * 256u;
	return ((((uint64_t) (bv >> 1) * W + w) * 2u + (bv & 1u)) * NB + bu) * 256u;
}

template <int ORDER, int DEPTH>
__global__ __launch_bounds__(512) void k(const uint8_t *__restrict__ buf, uint32_t *out) {
	const uint32_t g = blockIdx.x, blk8 = g >> 6, in = g & 63u;
	const uint32_t wgx = (blk8 & 7u) * 8u + (in & 7u), wgy = (blk8 >> 3) * 8u + (in >> 3);
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	const uint32_t bu = wgx * 4u + (wave & 3u), bv = wgy * 2u + (wave >> 2);
	const uint32_t lo = (lane >> 2) * 16u;
	uint4 slot[DEPTH];
	uint32_t acc = 0;
#pragma unroll
	for (int d = 0; d < DEPTH; d++) slot[d] = *(const uint4 *) (buf + piece<ORDER>(bu, bv, d) + lo);
	for (uint32_t w = 0; w + DEPTH <= W; w += DEPTH) {
#pragma unroll
		for (int d = 0; d < DEPTH; d++) {
			const uint4 v = slot[d];
			const uint32_t nw = w + DEPTH + d;
			slot[d] = *(const uint4 *) (buf + piece<ORDER>(bu, bv, nw < W ? nw : W - 1) + lo);
			acc += (v.x ^ v.y) + (v.z ^ v.w);
		}
	}
	out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int ORDER, int DEPTH>
static void run(const uint8_t *buf, uint32_t *out, const char *name) {
	const int blocks = 64 * 128;
	k<ORDER, DEPTH><<<blocks, 512>>>(buf, out);
	hipDeviceSynchronize();
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9f, sum = 0;
	for (int r = 0; r < 5; r++) {
		hipEventRecord(e0);
		k<ORDER, DEPTH><<<blocks, 512>>>(buf, out);
		hipEventRecord(e1);
		hipError_t err = hipDeviceSynchronize();
		if (err != hipSuccess) { printf("HIP error: %s\n", hipGetErrorString(err)); exit(3); }
		float ms; hipEventElapsedTime(&ms, e0, e1);
		best = ms < best ? ms : best; sum += ms;
	}
	const double bytes = (double) NB * NB * W * 256.0;
	printf("%-22s depth %2d  mean %6.3f ms  best %6.3f ms  %5.2f TB/s\n", name, DEPTH, sum / 5, best, bytes / (best * 1e-3) / 1e12);
}

int main() {
	signal(SIGPIPE, SIG_IGN); setvbuf(stdout, NULL, _IONBF, 0);
	const size_t bytes = (size_t) NB * NB * W * 256u;
	uint8_t *buf; uint32_t *out;
	if (hipMalloc(&buf, bytes) != hipSuccess) { printf("no memory\n"); return 2; }
	hipMemset(buf, 1, bytes); hipMalloc(&out, 64 * 128 * 512 * 4);
	run<0, 4>(buf, out, "A [bv][bu][w]");
	run<1, 4>(buf, out, "B [w][bv][bu]");
	run<2, 4>(buf, out, "C [bv][w][bu]");
	run<3, 4>(buf, out, "D [bv/2][w][bv%2][bu]");
	run<0, 8>(buf, out, "A [bv][bu][w]");
	run<1, 8>(buf, out, "B [w][bv][bu]");
	run<2, 8>(buf, out, "C [bv][w][bu]");
	run<3, 8>(buf, out, "D [bv/2][w][bv%2][bu]");
	run<0, 2>(buf, out, "A [bv][bu][w]");
	run<1, 2>(buf, out, "B [w][bv][bu]");
	return 0;
}
