// Microbenchmark: vector-memory (TA/TCP) issue cost per wave64 load on gfx950 for the access shapes of a ray-march
// sample fetch: load width (1/2/4/8/16 B), alignment, and how many distinct 128-byte lines a wave touches.
// Data set is tiny (L1/L2 resident); 8 waves per SIMD; each wave issues UNROLL independent loads per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int BYTES> struct LoadT;
template <> struct LoadT<1>  { typedef uint8_t  T; };
template <> struct LoadT<2>  { typedef uint16_t T; };
template <> struct LoadT<4>  { typedef uint32_t T; };
template <> struct LoadT<8>  { typedef uint2    T; };
template <> struct LoadT<16> { typedef uint4    T; };

__device__ inline uint32_t fold(uint8_t v) { return v; }
__device__ inline uint32_t fold(uint16_t v) { return v; }
__device__ inline uint32_t fold(uint32_t v) { return v; }
__device__ inline uint32_t fold(uint2 v) { return v.x ^ v.y; }
__device__ inline uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// lane l reads at base + (l / lanes_per_line) * line_stride + (l % lanes_per_line) * BYTES + misalign, plus a per-iteration
// offset that walks a small window so that loads are not trivially identical
template <int BYTES>
__global__ __launch_bounds__(256) void k(const uint8_t *buf, uint32_t *out, int iters, int lanes_per_line, int line_stride, int misalign, uint32_t window) {
	typedef typename LoadT<BYTES>::T T;
	const int lane = threadIdx.x & 63;
	uint32_t off = (lane / lanes_per_line) * line_stride + (lane % lanes_per_line) * BYTES + misalign + (threadIdx.x >> 6) * 8192 + (blockIdx.x & 7) * 65536;
	uint32_t acc = 0;
	uint32_t walk = 0;
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int u = 0; u < 8; u++) {
			T v;
			__builtin_memcpy(&v, buf + off + walk + u * 4096, BYTES);   // 8 independent loads in flight
			acc ^= fold(v);
		}
		walk = (walk + 256) & (window - 1);
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int BYTES> void run(const uint8_t *buf, uint32_t *out, int lpl, int stride, int mis, const char *what) {
	const int iters = 4000, blocks = 256 * 8;
	k<BYTES><<<blocks, 256>>>(buf, out, 50, lpl, stride, mis, 2048);
	hipDeviceSynchronize();
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	k<BYTES><<<blocks, 256>>>(buf, out, iters, lpl, stride, mis, 2048);
	hipEventRecord(e1); hipDeviceSynchronize();
	float ms; hipEventElapsedTime(&ms, e0, e1);
	double loads_per_cu = (double) iters * 8 * 4 * 8;          // 8 loads x 4 waves/block x 8 blocks per CU
	printf("%2d B  %-34s %7.2f ns per wave-load per CU  (%.2f ms)\n", BYTES, what, ms * 1e6 / loads_per_cu, ms);
}

int main() {
	uint8_t *buf; uint32_t *out;
	hipMalloc(&buf, 8 << 20); hipMemset(buf, 1, 8 << 20); hipMalloc(&out, 256 * 2048 * 4);
	// all 64 lanes in one line (fully coalesced); 8 lanes per line (8 lines); 2 lanes per line (32 lines); 1 lane per line
	run<1>(buf, out, 64, 128, 0, "64 lanes/line");
	run<1>(buf, out, 8, 128, 0, "8 lanes/line (8 lines)");
	run<2>(buf, out, 64, 128, 0, "64 lanes/line");
	run<2>(buf, out, 8, 128, 0, "8 lanes/line (8 lines)");
	run<2>(buf, out, 8, 128, 1, "8 lanes/line, odd address");
	run<2>(buf, out, 2, 128, 0, "2 lanes/line (32 lines)");
	run<2>(buf, out, 1, 128, 0, "1 lane/line (64 lines)");
	run<4>(buf, out, 32, 128, 0, "32 lanes/line (2 lines)");
	run<4>(buf, out, 8, 128, 0, "8 lanes/line (8 lines)");
	run<4>(buf, out, 8, 128, 1, "8 lanes/line, unaligned +1");
	run<8>(buf, out, 16, 128, 0, "16 lanes/line (4 lines)");
	run<8>(buf, out, 8, 128, 0, "8 lanes/line (8 lines)");
	run<8>(buf, out, 8, 128, 3, "8 lanes/line, unaligned +3");
	run<16>(buf, out, 8, 128, 0, "8 lanes/line (8 lines)");
	run<16>(buf, out, 4, 128, 0, "4 lanes/line (16 lines)");
	run<16>(buf, out, 4, 128, 5, "4 lanes/line, unaligned +5");
	run<16>(buf, out, 8, 256, 5, "8 lanes overlap, unaligned +5");
	return 0;
}
