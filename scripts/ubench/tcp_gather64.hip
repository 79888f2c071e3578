// Microbenchmark: cost of ONE 8-byte gather (global_load_dwordx2, 8- or only 4-byte aligned) against TWO 4-byte gathers for the
// lane patterns of the ray-march kernel, on two brick layouts:
//   "quad"  the current one: 4-byte elements, in-brick bit order a0 b0 | a1 b1 | c0 | a2 b2 | c1 c2 (two loads: slices c, c+1)
//   "run"   candidate: the R elements of a cell column along the march axis c are CONTIGUOUS (R = 8: 32-byte runs, a 128-byte
//           line = 2x2 cells x 8 steps; R = 9: 36-byte runs with the first element of the next brick duplicated), so that the
//           slices c and c+1 of a sample are 8 adjacent bytes: ONE dwordx2 load at 4-byte alignment
// Data is L1/L2 resident (16 KiB window per wave); the figure is TA/TCP time per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
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
			else if (W == 16) { const uint4 v = *(const uint4 *) p; acc ^= v.x ^ v.y ^ v.z ^ v.w; }   // global_load_dwordx4, 8-byte aligned (2-byte voxels)
			else { const uint2 v = *(const uint2 *) p; acc ^= v.x ^ v.y; }     // global_load_dwordx2; the address may be only 4-byte aligned
		}
		walk = (walk + 8192) & 16383;   // stays inside this wave's 16 KiB window
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
	return ms * 1e6 / ((double) iters * 8 * 4 * 8);
}

int main() {
	signal(SIGPIPE, SIG_IGN); setvbuf(stdout, NULL, _IONBF, 0);
	uint8_t *buf; uint32_t *out, *d_off; uint32_t hq0[64], hq1[64], hr[64];
	hipMalloc(&buf, 4 << 20); hipMemset(buf, 1, 4 << 20); hipMalloc(&out, 256 * 2048 * 4); hipMalloc(&d_off, 256);
	const int order[9] = { 0, 2, 5, 1, 3, 6, 4, 7, 8 };      // quad layout, chunk plane (a,b): a0 b0 | a1 b1 | c0 | a2 b2 | c1 c2
	auto spread = [&](int v, int axis) { return ((v & 1) << order[3 * axis]) | (((v >> 1) & 1) << order[3 * axis + 1]) | (((v >> 2) & 1) << order[3 * axis + 2]); };
	auto mort2 = [](int a, int b) { int r = 0; for (int i = 0; i < 3; i++) r |= (((a >> i) & 1) << (2 * i)) | (((b >> i) & 1) << (2 * i + 1)); return r; };
	struct Pat { const char *name; int kind; float pitch; int lanemap; float phase; };
	const Pat pats[] = {
		{ "ortho aligned, 0.5 cell/px, 2x2-px quads, phase ok", 0, 0.5f, 2, 0.0f },
		{ "ortho aligned, 0.5 cell/px, 2x2-px quads, phase off by one px", 0, 0.5f, 2, 0.5f },
		{ "perspective front, 0.375 cell/px", 0, 0.375f, 2, 0.2f },
		{ "perspective middle, 0.75 cell/px", 0, 0.75f, 2, 0.2f },
		{ "perspective back, 1.125 cell/px", 0, 1.125f, 2, 0.2f },
		{ "oblique (-45,-45), 0.5 cell/px, 4x1 quads", 1, 0.5f, 0, 0.3f },
		{ "oblique (-45,-45), 0.5 cell/px, 2x2 quads", 1, 0.5f, 2, 0.3f },
	};
	// "pair c / c+1" (round 3, candidate with 2.5 instead of 4.5 bytes per voxel): elements are PAIRS of voxels along a (2 bytes), the 9 pairs of
	// a column (a,c) along b are contiguous (18 bytes + 2 of padding), a sample = two 4-byte loads at 2-BYTE alignment (slices c and c+1)
	printf("%-64s %9s %9s %9s %9s %9s %9s %9s %9s %9s %9s\n", "pattern (ns per wave-instruction per CU)", "quad c", "quad c+1", "run8 x2", "run9 x2", "run8 al8", "u16 quad", "u16 run9", "u16 oct", "pair c", "pair c+1");
	for (const Pat &p : pats) for (int cstep = 0; cstep < 2; cstep++) {
		uint32_t hr9[64], hr8a[64], hq16[64], hr16[64], hoct[64], hp0[64], hp1[64];   // "u16 oct": ONE aligned 16-byte element per cell (2x2x2 two-byte voxels), quad brick order          // 2-byte voxels: 8-byte quad elements (two loads per sample), 72-byte runs (one 16-byte load)
		for (int l = 0; l < 64; l++) {
			const int qd = l >> 4;
			int gu = l & 3, gv = (l >> 2) & 3;
			if (p.lanemap == 2) { gu = ((l >> 1) & 2) | (l & 1); gv = ((l >> 2) & 2) | ((l >> 1) & 1); }
			const int i = (qd & 1) * 4 + gu, j = (qd >> 1) * 4 + gv;
			float a, b, c;
			if (p.kind == 0) { a = 8.0f + p.phase + i * p.pitch; b = 8.0f + p.phase + j * p.pitch; c = cstep ? 3.3f : 6.3f; }
			else {          // oblique: screen axes and the entry-depth skew of a (-45,-45) view, march axis c = dominant axis
				a = 8.3f + 0.707f * i * p.pitch * 0.5f / 0.5f + 0.5f * j * p.pitch; b = 8.6f + 0.5f * j * p.pitch - 0.707f * i * p.pitch;
				c = (cstep ? 3.3f : 6.3f) + 0.707f * j * p.pitch + 0.35f * i * p.pitch;
			}
			const int ia = (int) a, ib = (int) b, ic = (int) c;
			const int brick = (ic >> 3) * 16 + (ib >> 3) * 4 + (ia >> 3);
			hq0[l] = (brick * 512 + (spread(ia & 7, 0) | spread(ib & 7, 1) | spread(ic & 7, 2))) * 4;
			const int ic1 = ic + 1, brick1 = (ic1 >> 3) * 16 + (ib >> 3) * 4 + (ia >> 3);
			hq1[l] = (brick1 * 512 + (spread(ia & 7, 0) | spread(ib & 7, 1) | spread(ic1 & 7, 2))) * 4;
			hr[l]  = brick * 2048 + mort2(ia & 7, ib & 7) * 32 + (ic & 7) * 4;         // run of 8, 4-byte aligned start (wraps at 7: timing only)
			if ((ic & 7) == 7) hr[l] -= 4;
			hr9[l] = brick * 2304 + mort2(ia & 7, ib & 7) * 36 + (ic & 7) * 4;         // run of 9
			hr8a[l] = brick * 2048 + mort2(ia & 7, ib & 7) * 32 + (ic & 6) * 4;        // 8-byte aligned (even c only)
			hq16[l] = (brick * 512 + (spread(ia & 7, 0) | spread(ib & 7, 1) | spread(ic & 7, 2))) * 8;
			hr16[l] = brick * 4608 + mort2(ia & 7, ib & 7) * 72 + (ic & 7) * 8;
			hoct[l] = (brick * 512 + (spread(ia & 7, 0) | spread(ib & 7, 1) | spread(ic & 7, 2))) * 16;
			hp0[l] = brick * 1280 + ((ia & 7) + (ic & 7) * 8) * 20 + (ib & 7) * 2;
			hp1[l] = brick1 * 1280 + ((ia & 7) + (ic1 & 7) * 8) * 20 + (ib & 7) * 2;
		}
		char name[160];
		snprintf(name, sizeof name, "%s, c&7=%d", p.name, cstep ? 3 : 6);
		printf("%-64s %9.2f %9.2f %9.2f %9.2f %9.2f %9.2f %9.2f %9.2f %9.2f %9.2f\n", name, run<4>(buf, out, d_off, hq0), run<4>(buf, out, d_off, hq1),
		       run<8>(buf, out, d_off, hr), run<8>(buf, out, d_off, hr9), run<8>(buf, out, d_off, hr8a), run<8>(buf, out, d_off, hq16), run<16>(buf, out, d_off, hr16), run<16>(buf, out, d_off, hoct),
		       run<4>(buf, out, d_off, hp0), run<4>(buf, out, d_off, hp1));
	}
	return 0;
}
