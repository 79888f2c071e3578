// Microbenchmark: cost of fetching the 8 corner voxels of a trilinear sample from an LDS-resident window on gfx950.
// A wave (8x8-pixel tile) reads a 16x16x8-voxel window; variants:
//   A  raw u8 voxels, 4 x ds_read_u16 at VOXEL alignment (x pair; odd addresses allowed)
//   B  x-pair elements (u16, 2 bytes per voxel position), 4 x aligned ds_read_u16
//   C  quad elements (u32, 4 bytes per position), 2 x ds_read_b32
//   D  raw u8 voxels, 8 x ds_read_u8
// Lane positions: tile of 8x8 lanes at `pitch` voxels on a plane spanned by two axes, plus an optional tilt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

constexpr int WX = 16, WY = 16, WZ = 8;

template <int VARIANT>
__global__ __launch_bounds__(512) void k(uint32_t *out, int iters, const uint32_t *lane_off, int check) {
	constexpr int ELEM = VARIANT == 1 ? 2 : (VARIANT == 2 ? 4 : 1);
	__shared__ __attribute__((aligned(16))) uint8_t win[8][WX * WY * WZ * ELEM + 64];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint8_t *w = win[wave];
	for (int i = lane; i < WX * WY * WZ; i += 64) {
		// voxel value = low byte of a hash of the position
		auto vox = [](int p) { return (uint8_t) ((p * 2654435761u) >> 13); };
		const int x = i % WX, y = (i / WX) % WY, z = i / (WX * WY);
		const int xn = x + 1 < WX ? i + 1 : i, yn = y + 1 < WY ? WX : 0;
		(void) z;
		if (VARIANT == 1) { w[2 * i] = vox(i); w[2 * i + 1] = vox(xn); }
		else if (VARIANT == 2) { w[4 * i] = vox(i); w[4 * i + 1] = vox(xn); w[4 * i + 2] = vox(i + yn); w[4 * i + 3] = vox(xn + yn); }
		else w[i] = vox(i);
	}
	__syncthreads();
	const uint32_t base = lane_off[lane] * ELEM;
	uint32_t acc = 0, walk = 0;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const uint8_t *p = w + base + (walk + u) * ELEM;
			if (VARIANT == 0) {
				uint16_t a, b, c, d;
				__builtin_memcpy(&a, p, 2); __builtin_memcpy(&b, p + WX, 2); __builtin_memcpy(&c, p + WX * WY, 2); __builtin_memcpy(&d, p + WX * WY + WX, 2);
				acc += a + 3u * b + 5u * c + 7u * d;
			} else if (VARIANT == 1) {
				const uint16_t *q = (const uint16_t *) p;
				acc += q[0] + 3u * q[WX] + 5u * q[WX * WY] + 7u * q[WX * WY + WX];
			} else if (VARIANT == 2) {
				const uint32_t *q = (const uint32_t *) p;
				const uint32_t lo = q[0], hi = q[WX * WY];
				acc += (lo & 0xffffu) + 3u * (lo >> 16) + 5u * (hi & 0xffffu) + 7u * (hi >> 16);
			} else {
				acc += p[0] + 256u * p[1] + 3u * (p[WX] + 256u * p[WX + 1]) + 5u * (p[WX * WY] + 256u * p[WX * WY + 1]) + 7u * (p[WX * WY + WX] + 256u * p[WX * WY + WX + 1]);
			}
		}
		walk = (walk + 1) & 1;
	}
	if (check || acc == 0x12345678u) out[blockIdx.x * 512 + threadIdx.x] = acc;
}

static uint32_t h_off[64];
static void pattern(float pitch, int plane, float tilt) {
	// plane 0: lanes span (x,y); 1: (x,z); 2: (y,z).  tilt moves the third coordinate by tilt * (i + j) voxels.
	for (int l = 0; l < 64; l++) {
		const int qd = l >> 4, i = (qd & 1) * 4 + (l & 3), j = (qd >> 1) * 4 + ((l >> 2) & 3);
		const float a = 0.3f + i * pitch, b = 0.6f + j * pitch, c = 0.2f + tilt * (i + j);
		int x, y, z;
		if (plane == 0) { x = (int) a; y = (int) b; z = (int) c; }
		else if (plane == 1) { x = (int) a; z = (int) b; y = (int) c; }
		else { y = (int) a; z = (int) b; x = (int) c; }
		if (x > WX - 4) x = WX - 4;
		if (y > WY - 2) y = WY - 2;
		if (z > WZ - 2) z = WZ - 2;
		h_off[l] = (z * WY + y) * WX + x;
	}
}

template <int VARIANT> static void run(uint32_t *out, uint32_t *d_off, const char *what, uint32_t *checksum) {
	hipMemcpy(d_off, h_off, sizeof h_off, hipMemcpyHostToDevice);
	const int iters = 2000, blocks = 256 * 4;
	k<VARIANT><<<blocks, 512>>>(out, 20, d_off, 1);
	hipDeviceSynchronize();
	uint32_t h[64]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
	uint32_t cs = 0; for (int l = 0; l < 64; l++) cs = cs * 31 + h[l];
	*checksum = cs;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	k<VARIANT><<<blocks, 512>>>(out, iters, d_off, 0);
	hipEventRecord(e1); hipDeviceSynchronize();
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double samples_per_cu = (double) iters * 4 * 8 * 4;        // 4 samples x 8 waves x 4 blocks per CU
	printf("  %-44s %7.2f ns per wave-sample per CU   checksum %08x\n", what, ms * 1e6 / samples_per_cu, cs);
}

int main() {
	uint32_t *out, *d_off; hipMalloc(&out, 512 * 1024 * 4); hipMalloc(&d_off, 256);
	const float pitches[2] = { 0.865f, 0.5f };
	for (float pitch : pitches) for (int plane = 0; plane < 3; plane++) for (float tilt : { 0.0f, 0.35f }) {
		if (pitch * 7 + 1 > 7 && plane != 0 && false) continue;
		pattern(pitch, plane, tilt);
		printf("pitch %.3f plane %d tilt %.2f\n", pitch, plane, tilt);
		uint32_t c0, c1, c2, c3;
		run<0>(out, d_off, "A raw u8, 4 x ds_read_u16 (voxel aligned)", &c0);
		run<1>(out, d_off, "B x-pair u16 elements, 4 x ds_read_u16", &c1);
		run<2>(out, d_off, "C quad u32 elements, 2 x ds_read_b32", &c2);
		run<3>(out, d_off, "D raw u8, 8 x ds_read_u8", &c3);
		if (c0 != c3 || c1 != c3 || c2 != c3) printf("  !! checksums differ\n");
	}
	return 0;
}
