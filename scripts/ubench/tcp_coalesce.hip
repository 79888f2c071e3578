// Microbenchmark: how does the gfx950 vector L1 (TCP) coalesce a wave64 4-byte gather?
// Pattern A(n): n adjacent lanes share one 128-byte line (contiguous dwords); Pattern B(n): the lanes that share a line are
// interleaved (lane l -> line l % (64/n)); Pattern C(n, sector): like B but lanes of a line spread over different 32B sectors.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k(const uint8_t *buf, uint32_t *out, int iters, const uint32_t *lane_off) {
	const int lane = threadIdx.x & 63;
	uint32_t off = lane_off[lane] + (threadIdx.x >> 6) * 16384 + (blockIdx.x & 7) * 65536;
	uint32_t acc = 0, walk = 0;
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int u = 0; u < 8; u++) {
			acc ^= *(const uint32_t *) (buf + off + walk + u * 128 * 1024);
		}
		walk = (walk + 8192) & 16383;   // stays inside this wave's 16 KiB window
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static void run(const uint8_t *buf, uint32_t *out, uint32_t *d_off, const uint32_t *h_off, const char *what) {
	hipMemcpy(d_off, h_off, 64 * 4, hipMemcpyHostToDevice);
	const int iters = 3000, blocks = 256 * 8;
	k<<<blocks, 256>>>(buf, out, 50, d_off);
	hipDeviceSynchronize();
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	k<<<blocks, 256>>>(buf, out, iters, d_off);
	hipEventRecord(e1); hipDeviceSynchronize();
	float ms; hipEventElapsedTime(&ms, e0, e1);
	double loads_per_cu = (double) iters * 8 * 4 * 8;
	printf("%-58s %7.2f ns per wave-load per CU\n", what, ms * 1e6 / loads_per_cu);
}

int main() {
	uint8_t *buf; uint32_t *out, *d_off; uint32_t h[64]; char name[128];
	hipMalloc(&buf, 4 << 20); hipMemset(buf, 1, 4 << 20); hipMalloc(&out, 256 * 2048 * 4); hipMalloc(&d_off, 256);
	for (int n : {1, 2, 4, 8, 16, 32}) {
		for (int l = 0; l < 64; l++) h[l] = (l / n) * 128 + (l % n) * 4;
		snprintf(name, sizeof name, "A: %2d adjacent lanes per line, contiguous dwords", n); run(buf, out, d_off, h, name);
	}
	for (int n : {2, 4, 8, 16}) {
		int nl = 64 / n;
		for (int l = 0; l < 64; l++) h[l] = (l % nl) * 128 + (l / nl) * 4;
		snprintf(name, sizeof name, "B: %2d interleaved lanes per line, contiguous dwords", n); run(buf, out, d_off, h, name);
	}
	for (int n : {2, 4}) {
		int nl = 64 / n;
		for (int l = 0; l < 64; l++) h[l] = (l / n) * 128 + (l % n) * 32;
		snprintf(name, sizeof name, "C: %2d adjacent lanes per line, one dword per 32B sector", n); run(buf, out, d_off, h, name);
		(void) nl;
	}
	// same dword for pairs / quads of adjacent lanes (2 px per voxel), 16 or 32 distinct dwords, contiguous in ONE line
	for (int l = 0; l < 64; l++) h[l] = (l / 2) * 4; run(buf, out, d_off, h, "D: lane pairs share a dword, 32 dwords contiguous (1 line)");
	for (int l = 0; l < 64; l++) h[l] = (l / 4) * 4; run(buf, out, d_off, h, "D: lane quads share a dword, 16 dwords contiguous");
	// 8x8 pixel tile over a 4x4 element footprint: row r of 8 lanes -> 4 dwords at row stride 32 B (quad-brick slice)
	for (int l = 0; l < 64; l++) h[l] = ((l >> 3) / 2) * 32 + ((l & 7) / 2) * 4; run(buf, out, d_off, h, "E: 8x8 tile -> 4x4 elements, rows 32 B apart (view along z)");
	for (int l = 0; l < 64; l++) h[l] = ((l >> 3) / 2) * 256 + ((l & 7) / 2) * 4; run(buf, out, d_off, h, "E: 8x8 tile -> 4x4 elements, rows 256 B apart");
	for (int l = 0; l < 64; l++) h[l] = ((l >> 3) / 2) * 256 + ((l & 7) / 2) * 32; run(buf, out, d_off, h, "E: 8x8 tile -> 4x4 elements, 32 B and 256 B strides");
	// G: the ray-march kernel's own gather — 8x8-pixel tile (4x4-pixel lane groups) at `pitch` voxels per pixel on an axis
	// plane or an oblique plane, quad elements (4 B) in Morton bricks (z lowest, y middle, x top slot), one slice
	{
		auto dil = [](int v) { return (v & 1) | ((v & 2) << 2) | ((v & 4) << 4); };
		const float pitches[3] = { 0.5f, 0.865f, 1.5f };
		for (float pitch : pitches) for (int plane = 0; plane < 4; plane++) {
			for (int l = 0; l < 64; l++) {
				const int qd = l >> 4, i = (qd & 1) * 4 + (l & 3), j = (qd >> 1) * 4 + ((l >> 2) & 3);
				const float a = 0.3f + i * pitch, b = 0.6f + j * pitch;
				float x, y, z;
				if (plane == 0) { x = a; y = b; z = 3.2f; }
				else if (plane == 1) { x = a; z = b; y = 3.2f; }
				else if (plane == 2) { y = a; z = b; x = 3.2f; }
				else { x = 0.707f * a + 0.408f * b + 2.f; y = -0.707f * a + 0.408f * b + 9.f; z = -0.816f * b + 12.f; }
				const int ix = (int) x, iy = (int) y, iz = (int) z;
				const int brick = (iz >> 3) * 4 + (iy >> 3) * 2 + (ix >> 3);     // 2x2x2 bricks are plenty
				h[l] = (brick * 512 + (dil(iz & 7) | (dil(iy & 7) << 1) | (dil(ix & 7) << 2))) * 4;
			}
			snprintf(name, sizeof name, "G: kernel gather, pitch %.3f, plane %d", pitch, plane); run(buf, out, d_off, h, name);
		}
	}
	// H: the exact lane pattern of the headline frame (ortho, 0.5 voxel per pixel: texel = 0.5 * pixel - 0.5) for a tile at
	// pixel (16, 16) of an axis-aligned view, with two in-brick bit placements (x0 x1 x2 y0 y1 y2 z0 z1 z2 -> bit)
	{
		const int orders[3][9] = { { 2, 5, 8, 1, 4, 7, 0, 3, 6 }, { 0, 2, 5, 1, 3, 6, 4, 7, 8 }, { 4, 7, 8, 0, 2, 5, 1, 3, 6 } };
		const char *oname[3] = { "morton zyx", "flat xy   ", "flat yz   " };
		for (int phase = 16; phase <= 17; phase++) for (int o = 0; o < 3; o++) for (int plane = 0; plane < 3; plane++) {
			auto spread = [&](int v, int axis) { return ((v & 1) << orders[o][3 * axis]) | (((v >> 1) & 1) << orders[o][3 * axis + 1]) | (((v >> 2) & 1) << orders[o][3 * axis + 2]); };
			for (int l = 0; l < 64; l++) {
				const int qd = l >> 4, i = (qd & 1) * 4 + (l & 3), j = (qd >> 1) * 4 + ((l >> 2) & 3);
				const int ca = (phase + i - 1) >> 1, cb = (phase + j - 1) >> 1, cc = 3;
				int ix, iy, iz;
				if (plane == 0) { ix = ca; iy = cb; iz = cc; } else if (plane == 1) { ix = ca; iz = cb; iy = cc; } else { iy = ca; iz = cb; ix = cc; }
				const int brick = (iz >> 3) * 4 + (iy >> 3) * 2 + (ix >> 3);
				h[l] = (brick * 512 + (spread(ix & 7, 0) | spread(iy & 7, 1) | spread(iz & 7, 2))) * 4;
			}
			snprintf(name, sizeof name, "H: headline tile at pixel %d, %s, plane %d", phase, oname[o], plane); run(buf, out, d_off, h, name);
		}
	}
	// I: final brick order (x0 y0 | x1 y1 | z0 | x2 y2 | z1 z2), tile aligned to even cells (2 pixels per cell), the three lane
	// orders of the kernel (rows / columns / 2x2 blocks) on the three axis planes (screen x -> first axis, screen y -> second)
	{
		const int order[9] = { 0, 2, 5, 1, 3, 6, 4, 7, 8 };
		auto spread = [&](int v, int axis) { return ((v & 1) << order[3 * axis]) | (((v >> 1) & 1) << order[3 * axis + 1]) | (((v >> 2) & 1) << order[3 * axis + 2]); };
		const char *mname[3] = { "rows   ", "columns", "blocks " };
		const int planes[3][2] = { { 0, 1 }, { 0, 2 }, { 2, 1 } };           // (x,y) (x,z) (z,y)
		const char *pname[3] = { "(x,y)", "(x,z)", "(z,y)" };
		for (int pl = 0; pl < 3; pl++) for (int map = 0; map < 3; map++) for (int second = 0; second < 2; second++) {
			for (int l = 0; l < 64; l++) {
				const int qd = l >> 4;
				int gu = l & 3, gv = (l >> 2) & 3;
				if (map == 2) { gu = ((l >> 1) & 2) | (l & 1); gv = ((l >> 2) & 2) | ((l >> 1) & 1); }
				else if (map == 1) { const int t = gu; gu = gv; gv = t; }
				const int i = (qd & 1) * 4 + gu, j = (qd >> 1) * 4 + gv;
				int c[3] = { 3, 3, 3 };
				c[planes[pl][0]] = (16 + i) >> 1; c[planes[pl][1]] = (16 + j) >> 1;
				c[2] += second;                                              // the z+1 load of the same sample
				const int brick = (c[2] >> 3) * 4 + (c[1] >> 3) * 2 + (c[0] >> 3);
				h[l] = (brick * 512 + (spread(c[0] & 7, 0) | spread(c[1] & 7, 1) | spread(c[2] & 7, 2))) * 4;
			}
			snprintf(name, sizeof name, "I: plane %s, %s, %s load", pname[pl], mname[map], second ? "z+1" : "z  "); run(buf, out, d_off, h, name);
		}
	}
	// J: how the cost grows with the number of lane quads that straddle two 16-byte chunks (k of 16 quads); the other
	// quads read one chunk each.  "near": the second chunk is 16 B away in the same sector; "far": in another 128-byte line
	for (int far = 0; far < 2; far++) for (int k : {0, 1, 2, 4, 8, 12, 16}) {
		for (int l = 0; l < 64; l++) {
			const int q = l >> 2, i = l & 3;
			h[l] = q * 32 + (i & 1) * 4;                                      // quad q: two dwords of chunk 2q
			if (q < k && i >= 2) h[l] += far ? 4096 : 16;                     // lanes 2,3 of a split quad: another chunk
		}
		snprintf(name, sizeof name, "J: %2d of 16 quads straddle two chunks (%s)", k, far ? "far" : "near"); run(buf, out, d_off, h, name);
	}
	// K: quads whose four lanes read 1, 2 or 4 distinct dwords of ONE chunk, and 4 dwords of 4 different chunks of one line
	for (int v = 0; v < 4; v++) {
		for (int l = 0; l < 64; l++) {
			const int q = l >> 2, i = l & 3;
			if (v == 0) h[l] = q * 16; else if (v == 1) h[l] = q * 16 + (i & 1) * 4; else if (v == 2) h[l] = q * 16 + i * 4;
			else h[l] = (q >> 1) * 128 + (q & 1) * 4 + i * 16;
		}
		const char *kn[4] = { "1 dword per quad", "2 dwords of one chunk", "4 dwords of one chunk", "4 chunks of one line per quad" };
		snprintf(name, sizeof name, "K: %s", kn[v]); run(buf, out, d_off, h, name);
	}
	// all lanes same dword
	for (int l = 0; l < 64; l++) h[l] = 0; run(buf, out, d_off, h, "F: all lanes one dword");
	return 0;
}
