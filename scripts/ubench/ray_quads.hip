// Microbenchmark (round 4, for the next round's plan): what one 8-byte wave-gather on run bricks (36-byte runs along the march axis c, 2-D Morton
// order of the runs inside a brick) costs the L1 per CU when the four lanes of a quad are
//   "pixels"     four PIXELS at the same sample depth (the product: 8x8-pixel waves, 2x2-pixel quads), or
//   "along ray"  four CONSECUTIVE SAMPLES of one ray (4x4-pixel waves): the quad stays inside one run, i.e. one cache line.
// Views: orthogonal along c (0.5 cells per pixel), perspective at three depths (0.375 / 0.75 / 1.125 cells per pixel), oblique (-45,-45).
// Data is L1 / L2 resident; figure = ns per wave-instruction per CU with 32 waves per CU (col_gather.hip's harness).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <csignal>

__global__ __launch_bounds__(256) void k(const uint8_t *buf, uint32_t *out, int iters, const uint32_t *lane_off) {
	const int lane = threadIdx.x & 63;
	uint32_t off = lane_off[lane] + (threadIdx.x >> 6) * 16384 + (blockIdx.x & 7) * 65536;
	uint32_t acc = 0, walk = 0;
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int u = 0; u < 8; u++) {
			const uint2 v = *(const uint2 *) (buf + off + walk + u * 128 * 1024);     // global_load_dwordx2, 4-byte aligned
			acc ^= v.x ^ v.y;
		}
		walk = (walk + 8192) & 16383;
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static double run(const uint8_t *buf, uint32_t *out, uint32_t *d_off, const uint32_t *h_off) {
	hipMemcpy(d_off, h_off, 64 * 4, hipMemcpyHostToDevice);
	const int iters = 2000, blocks = 256 * 8;
	k<<<blocks, 256>>>(buf, out, 50, d_off);
	hipDeviceSynchronize();
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	k<<<blocks, 256>>>(buf, out, iters, d_off);
	hipEventRecord(e1);
	if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); exit(3); }
	float ms; hipEventElapsedTime(&ms, e0, e1);
	return ms * 1e6 / ((double) iters * 8 * 4 * 8);
}

int main() {
	signal(SIGPIPE, SIG_IGN); setvbuf(stdout, NULL, _IONBF, 0);
	uint8_t *buf; uint32_t *out, *d_off;
	hipMalloc(&buf, 4 << 20); hipMemset(buf, 1, 4 << 20); hipMalloc(&out, 256 * 2048 * 4); hipMalloc(&d_off, 256);
	auto mort2 = [](int a, int b) { int r = 0; for (int i = 0; i < 3; i++) r |= (((a >> i) & 1) << (2 * i)) | (((b >> i) & 1) << (2 * i + 1)); return r; };
	auto run9 = [&](float a, float b, float c) {          // byte offset of the element pair (c, c+1) of cell column (a, b): bricks of 8^3 cells, 4 x 4 x n bricks
		const int ia = (int) a & 31, ib = (int) b & 31, ic = (int) c & 15;
		const int brick = (ic >> 3) * 16 + (ib >> 3) * 4 + (ia >> 3);
		return (uint32_t) (brick * 2304 + mort2(ia & 7, ib & 7) * 36 + (ic & 7) * 4);
	};
	struct View { const char *name; float pitch; float da, db, dc; float sa, sb, sc; float ja, jb, jc; };      // per pixel i: (sa, sb, sc), per pixel j: (ja, jb, jc), per sample: (da, db, dc)
	const View views[] = {
		{ "orthogonal along c, 0.5 cell/px",      0.5f,   0.0f, 0.0f, 1.0f,   1, 0, 0,   0, 1, 0 },
		{ "perspective, 0.375 cell/px",            0.375f, 0.1f, 0.1f, 1.0f,   1, 0, 0,   0, 1, 0 },
		{ "perspective, 0.75 cell/px",             0.75f,  0.2f, 0.2f, 1.0f,   1, 0, 0,   0, 1, 0 },
		{ "perspective, 1.125 cell/px",            1.125f, 0.3f, 0.3f, 1.0f,   1, 0, 0,   0, 1, 0 },
		{ "oblique (-45,-45), 0.5 cell/px",        0.5f,   0.5f, -0.5f, 0.707f,   0.707f, -0.707f, 0.35f,   0.5f, 0.5f, 0.707f },
	};
	printf("%-40s %10s %10s %10s\n", "view (ns per wave-gather and CU)", "pixels 2x2", "pixels 4x1", "along ray");
	for (const View &v : views) {
		uint32_t h[3][64];
		for (int l = 0; l < 64; l++) {
			for (int m = 0; m < 3; m++) {
				int i, j, s = 0;
				if (m == 0) { const int qd = l >> 4, gu = ((l >> 1) & 2) | (l & 1), gv = ((l >> 2) & 2) | ((l >> 1) & 1); i = (qd & 1) * 4 + gu; j = (qd >> 1) * 4 + gv; }
				else if (m == 1) { const int qd = l >> 4; i = (qd & 1) * 4 + (l & 3); j = (qd >> 1) * 4 + ((l >> 2) & 3); }
				else { const int p = l >> 2; i = p & 3; j = p >> 2; s = l & 3; }
				const float a = 8.3f + (i * v.sa + j * v.ja) * v.pitch + s * v.da, b = 8.6f + (i * v.sb + j * v.jb) * v.pitch + s * v.db, c = 2.3f + (i * v.sc + j * v.jc) * v.pitch + s * v.dc;
				h[m][l] = run9(a, b, c);
			}
		}
		printf("%-40s %10.2f %10.2f %10.2f\n", v.name, run(buf, out, d_off, h[0]), run(buf, out, d_off, h[1]), run(buf, out, d_off, h[2]));
	}
	return 0;
}
