// vr_multi.cpp — one PROCESS driving several MI355X behind the same C ABI: the multi-device flavour of include/vr_hip.h.
//
// The reference has no multi-GPU code (it picks one device, VolR.cpp:141-172); its per-frame call is
// `renderers[id]->render_volume(buffer, raycaster)` (VolR.cpp:110).  vr_hip_multi_* lets that one call fan out:
//   * one vr_ctx + one stream per device, volume / TF / ESL replicated (rays are independent);
//   * the frame is cut into interleaved bands of rows (band b belongs to device b mod n) — the same partition
//     volume-rendering_amd/distributed.py uses across processes, so long centre rays and short edge rays mix on every device;
//   * the RGBA8 bands travel to device 0 over xGMI: RCCL point-to-point (ncclSend / ncclRecv in one group, communicators from
//     ncclCommInitAll; librccl is dlopen'ed so that single-GPU users never need it), or plain peer copies when RCCL is not
//     available or the device list names one GPU twice (how the path is tested on a one-GPU box);
//   * a copy kernel on device 0 de-interleaves the bands into the caller's frame.
// Built only on the public single-device ABI + the HIP runtime: nothing here touches vr_ctx internals.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and enum values only; the functions are resolved with dlsym

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/vr_hip.h"

namespace {

struct Rccl {
	void *lib = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	bool load() {
		// a process that already holds an RCCL (torch ships one) gets that one: same SONAME, already mapped
		for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
			lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (lib) break;
		}
		if (!lib) return false;
		CommInitAll = (decltype(CommInitAll)) dlsym(lib, "ncclCommInitAll");
		CommDestroy = (decltype(CommDestroy)) dlsym(lib, "ncclCommDestroy");
		GroupStart = (decltype(GroupStart)) dlsym(lib, "ncclGroupStart");
		GroupEnd = (decltype(GroupEnd)) dlsym(lib, "ncclGroupEnd");
		Send = (decltype(Send)) dlsym(lib, "ncclSend");
		Recv = (decltype(Recv)) dlsym(lib, "ncclRecv");
		GetErrorString = (decltype(GetErrorString)) dlsym(lib, "ncclGetErrorString");
		return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv;
	}
};

enum Transport { kSingle = 0, kRccl = 1, kPeerCopy = 2 };

}  // namespace

struct vr_multi {
	int n = 0;
	std::vector<int> dev;
	std::vector<vr_ctx *> ctx;
	std::vector<hipStream_t> stream;
	std::vector<hipEvent_t> rendered;
	std::vector<void *> local;               // rank r's bands on its own device (rank 0: its slice of `staging`)
	void *staging = nullptr;                 // device 0: [n][local_rows][width] RGBA8
	void *frame0 = nullptr;                  // device 0: assembled frame of the host-buffer entry point
	uint32_t width = 0, height = 0, band_rows = 0, per_rank = 0, local_rows = 0;
	Transport transport = kSingle;
	Rccl rccl;
	std::vector<ncclComm_t> comm;
	float last_total_ms = 0;
	std::string err;
};

namespace {

int fail(vr_multi *m, int code, const char *what, hipError_t e = hipSuccess) {
	if (m) {
		char buf[512];
		if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s (%d)", what, hipGetErrorString(e), (int) e);
		else snprintf(buf, sizeof buf, "%s", what);
		m->err = buf;
	}
	return code;
}

#define VRM_TRY(m, expr)                                                                         \
	do {                                                                                         \
		hipError_t e_ = (expr);                                                                  \
		if (e_ != hipSuccess) {                                                                  \
			(void) hipGetLastError();                                                            \
			return fail((m), e_ == hipErrorOutOfMemory ? VR_ERR_ALLOC : VR_ERR_HIP, #expr, e_);  \
		}                                                                                        \
	} while (0)

int forward(vr_multi *m, int rank, int rc) {
	if (rc != VR_OK) {
		char buf[640];
		snprintf(buf, sizeof buf, "device %d (rank %d): %s", m->dev[rank], rank, vr_hip_last_error(m->ctx[rank]));
		m->err = buf;
	}
	return rc;
}

// de-interleave: frame row y  <-  staging[rank][local_row], 16 bytes (4 pixels) or 4 bytes per thread
template <typename T>
__global__ __launch_bounds__(256)
void assemble_kernel(const T *__restrict__ staging, T *__restrict__ frame, uint32_t row_elems, uint32_t height, uint32_t n,
                     uint32_t band_rows, uint32_t local_rows) {
	const uint32_t y = blockIdx.y;
	const uint32_t band = y / band_rows, rank = band % n, local_row = (band / n) * band_rows + (y - band * band_rows);
	const T *src = staging + ((size_t) rank * local_rows + local_row) * row_elems;
	T *dst = frame + (size_t) y * row_elems;
	for (uint32_t x = blockIdx.x * 256 + threadIdx.x; x < row_elems; x += gridDim.x * 256) dst[x] = src[x];
	(void) height;
}

void release_buffers(vr_multi *m) {
	for (int r = 1; r < m->n; r++)
		if (m->local[r]) { (void) hipSetDevice(m->dev[r]); (void) hipFree(m->local[r]); m->local[r] = nullptr; }
	if (m->n > 0) (void) hipSetDevice(m->dev[0]);
	if (m->staging) { (void) hipFree(m->staging); m->staging = nullptr; }
	if (m->frame0) { (void) hipFree(m->frame0); m->frame0 = nullptr; }
	if (m->n > 0) m->local[0] = nullptr;
}

}  // namespace

extern "C" {

// Where frame row y of an n-device frame lives: the band map shared by the split, the gather and the assemble kernel.
void vr_hip_multi_band_map(uint32_t n, uint32_t band_rows, uint32_t y, uint32_t *rank_out, uint32_t *local_row_out) {
	const uint32_t band = y / band_rows;
	if (rank_out) *rank_out = band % n;
	if (local_row_out) *local_row_out = (band / n) * band_rows + (y - band * band_rows);
}

// 128-row bands (8 workgroup tile rows: the kernel walks its tiles in 8x8-tile blocks) when every device still gets at least
// two of them, else 16-row bands (one tile row); one device takes the whole frame.  Same rule as distributed.default_band_rows.
uint32_t vr_hip_multi_default_band_rows(uint32_t height, uint32_t n) {
	if (n <= 1) return height ? height : 1;
	if (height % 128u == 0 && height / 128u >= 2u * n) return 128u;
	return 16u;
}

int vr_hip_multi_create(int n, const int *devices, vr_multi **out) {
	if (out == nullptr) return VR_ERR_INVALID;
	*out = nullptr;
	if (n <= 0 || n > 64 || devices == nullptr) return VR_ERR_INVALID;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void) hipGetLastError(); return VR_ERR_NO_DEVICE; }
	for (int r = 0; r < n; r++) if (devices[r] < 0 || devices[r] >= count) return VR_ERR_NO_DEVICE;
	vr_multi *m = new (std::nothrow) vr_multi();
	if (m == nullptr) return VR_ERR_ALLOC;
	*out = m;
	m->n = n;
	m->dev.assign(devices, devices + n);
	m->ctx.assign(n, nullptr); m->stream.assign(n, nullptr); m->rendered.assign(n, nullptr); m->local.assign(n, nullptr);
	bool distinct = true;
	for (int a = 0; a < n; a++) for (int b = a + 1; b < n; b++) if (devices[a] == devices[b]) distinct = false;
	for (int r = 0; r < n; r++) {
		int rc = vr_hip_create(devices[r], &m->ctx[r]);
		if (rc != VR_OK) return fail(m, rc, m->ctx[r] ? vr_hip_last_error(m->ctx[r]) : "vr_hip_create failed");
		VRM_TRY(m, hipSetDevice(devices[r]));
		VRM_TRY(m, hipStreamCreateWithFlags(&m->stream[r], hipStreamNonBlocking));
		VRM_TRY(m, hipEventCreateWithFlags(&m->rendered[r], hipEventDisableTiming));
	}
	m->transport = kSingle;
	if (n > 1) {
		m->transport = kPeerCopy;
		for (int r = 1; r < n && distinct; r++) {           // direct xGMI access between device 0 and every other one
			int can = 0;
			if (hipDeviceCanAccessPeer(&can, devices[0], devices[r]) == hipSuccess && can) {
				(void) hipSetDevice(devices[0]);
				hipError_t e = hipDeviceEnablePeerAccess(devices[r], 0);
				if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void) hipGetLastError();
			}
		}
		const char *force = getenv("VR_MULTI_TRANSPORT");        // "peer" forces the copy path (A/B, debugging)
		if (distinct && !(force && strcmp(force, "peer") == 0) && m->rccl.load()) {
			m->comm.assign(n, nullptr);
			if (m->rccl.CommInitAll(m->comm.data(), n, devices) == ncclSuccess) m->transport = kRccl;
			else m->comm.clear();
		}
	}
	return VR_OK;
}

void vr_hip_multi_destroy(vr_multi *m) {
	if (m == nullptr) return;
	for (int r = 0; r < m->n; r++) if (m->stream[r]) { (void) hipSetDevice(m->dev[r]); (void) hipStreamSynchronize(m->stream[r]); }
	for (size_t r = 0; r < m->comm.size(); r++) if (m->comm[r]) (void) m->rccl.CommDestroy(m->comm[r]);
	release_buffers(m);
	for (int r = 0; r < m->n; r++) {
		(void) hipSetDevice(m->dev[r]);
		if (m->rendered[r]) (void) hipEventDestroy(m->rendered[r]);
		if (m->stream[r]) (void) hipStreamDestroy(m->stream[r]);
		vr_hip_destroy(m->ctx[r]);
	}
	delete m;
}

const char *vr_hip_multi_last_error(const vr_multi *m) { return m ? m->err.c_str() : "no context"; }
int vr_hip_multi_count(const vr_multi *m) { return m ? m->n : 0; }
vr_ctx *vr_hip_multi_context(vr_multi *m, int rank) { return (m && rank >= 0 && rank < m->n) ? m->ctx[rank] : nullptr; }
const char *vr_hip_multi_transport(const vr_multi *m) {
	if (m == nullptr) return "none";
	return m->transport == kRccl ? "rccl" : (m->transport == kPeerCopy ? "peer-copy" : "single");
}

int vr_hip_multi_set_window(vr_multi *m, uint32_t w, uint32_t h) {
	if (m == nullptr) return VR_ERR_INVALID;
	if (w == 0 || h == 0 || w > 65535u || h > 65535u) return fail(m, VR_ERR_INVALID, "window dims out of range (1..65535)");
	release_buffers(m);
	m->width = w; m->height = h;
	m->band_rows = vr_hip_multi_default_band_rows(h, (uint32_t) m->n);
	const uint32_t nbands = (h + m->band_rows - 1) / m->band_rows;
	m->per_rank = (nbands + (uint32_t) m->n - 1) / (uint32_t) m->n;
	m->local_rows = m->per_rank * m->band_rows;
	const size_t slice = (size_t) m->local_rows * w * 4;
	VRM_TRY(m, hipSetDevice(m->dev[0]));
	VRM_TRY(m, hipMalloc(&m->staging, slice * m->n));
	VRM_TRY(m, hipMalloc(&m->frame0, (size_t) w * h * 4));
	m->local[0] = m->staging;                                // device 0 renders straight into its slice
	for (int r = 1; r < m->n; r++) {
		VRM_TRY(m, hipSetDevice(m->dev[r]));
		VRM_TRY(m, hipMalloc(&m->local[r], slice));
	}
	return VR_OK;
}

int vr_hip_multi_set_transfer_fn(vr_multi *m, const float *tf, const uint32_t *esl) {
	if (m == nullptr) return VR_ERR_INVALID;
	for (int r = 0; r < m->n; r++) { int rc = forward(m, r, vr_hip_set_transfer_fn(m->ctx[r], tf, esl)); if (rc) return rc; }
	return VR_OK;
}

int vr_hip_multi_set_volume(vr_multi *m, const void *host, uint32_t x, uint32_t y, uint32_t z, uint32_t bpv) {
	if (m == nullptr) return VR_ERR_INVALID;
	for (int r = 0; r < m->n; r++) { int rc = forward(m, r, vr_hip_set_volume(m->ctx[r], host, x, y, z, bpv)); if (rc) return rc; }
	return VR_OK;
}

int vr_hip_multi_generate_volume(vr_multi *m, uint32_t kind, uint32_t n, uint32_t seed, uint32_t bpv) {
	if (m == nullptr) return VR_ERR_INVALID;
	for (int r = 0; r < m->n; r++) { int rc = forward(m, r, vr_hip_generate_volume(m->ctx[r], kind, n, seed, bpv)); if (rc) return rc; }
	return VR_OK;
}

// whole frame into `dev_rgba` on device 0 (devices[0] of the create call); synchronous from the caller's view like every
// reference renderer call
int vr_hip_multi_render_device(vr_multi *m, const vr_params *p, void *dev_rgba) {
	if (m == nullptr) return VR_ERR_INVALID;
	if (p == nullptr || dev_rgba == nullptr) return fail(m, VR_ERR_INVALID, "params / buffer is NULL");
	if (p->view.width != m->width || p->view.height != m->height || m->staging == nullptr)
		return fail(m, VR_ERR_NOT_READY, "view dims differ from the window: call vr_hip_multi_set_window first");
	hipEvent_t t0 = nullptr, t1 = nullptr;
	VRM_TRY(m, hipSetDevice(m->dev[0]));
	VRM_TRY(m, hipEventCreate(&t0)); VRM_TRY(m, hipEventCreate(&t1));
	VRM_TRY(m, hipEventRecord(t0, m->stream[0]));
	const size_t slice = (size_t) m->local_rows * m->width * 4;
	for (int r = 0; r < m->n; r++) {
		vr_params pr = *p;
		pr.x0 = 0; pr.out_width = m->width;
		if (m->n == 1) { pr.out_rows = m->height; pr.band_rows = m->height; pr.band_stride = 1; pr.band_first = 0; }
		else { pr.out_rows = m->local_rows; pr.band_rows = m->band_rows; pr.band_stride = (uint32_t) m->n; pr.band_first = (uint32_t) r; }
		VRM_TRY(m, hipSetDevice(m->dev[r]));
		int rc = forward(m, r, vr_hip_render_device(m->ctx[r], &pr, m->n == 1 ? dev_rgba : m->local[r], m->stream[r]));
		if (rc) return rc;
		VRM_TRY(m, hipEventRecord(m->rendered[r], m->stream[r]));
	}
	if (m->n > 1) {
		if (m->transport == kRccl) {
			// one group: every other device sends its bands, device 0 receives them behind its own render
			if (m->rccl.GroupStart() != ncclSuccess) return fail(m, VR_ERR_HIP, "ncclGroupStart failed");
			for (int r = 1; r < m->n; r++) {
				ncclResult_t a = m->rccl.Send(m->local[r], slice, ncclUint8, 0, m->comm[r], m->stream[r]);
				ncclResult_t b = m->rccl.Recv((uint8_t *) m->staging + slice * r, slice, ncclUint8, r, m->comm[0], m->stream[0]);
				if (a != ncclSuccess || b != ncclSuccess) { (void) m->rccl.GroupEnd(); return fail(m, VR_ERR_HIP, "ncclSend / ncclRecv failed"); }
			}
			if (m->rccl.GroupEnd() != ncclSuccess) return fail(m, VR_ERR_HIP, "ncclGroupEnd failed");
		} else {
			VRM_TRY(m, hipSetDevice(m->dev[0]));
			for (int r = 1; r < m->n; r++) {
				VRM_TRY(m, hipStreamWaitEvent(m->stream[0], m->rendered[r], 0));
				if (m->dev[r] == m->dev[0])
					VRM_TRY(m, hipMemcpyAsync((uint8_t *) m->staging + slice * r, m->local[r], slice, hipMemcpyDeviceToDevice, m->stream[0]));
				else
					VRM_TRY(m, hipMemcpyPeerAsync((uint8_t *) m->staging + slice * r, m->dev[0], m->local[r], m->dev[r], slice, m->stream[0]));
			}
		}
		VRM_TRY(m, hipSetDevice(m->dev[0]));
		const size_t row_bytes = (size_t) m->width * 4;
		if (row_bytes % 16 == 0 && ((uintptr_t) dev_rgba % 16) == 0) {
			const uint32_t elems = (uint32_t) (row_bytes / 16);
			hipLaunchKernelGGL(assemble_kernel<uint4>, dim3((elems + 255) / 256, m->height), dim3(256), 0, m->stream[0],
			                   (const uint4 *) m->staging, (uint4 *) dev_rgba, elems, m->height, (uint32_t) m->n, m->band_rows, m->local_rows);
		} else {
			hipLaunchKernelGGL(assemble_kernel<uint32_t>, dim3((m->width + 255) / 256, m->height), dim3(256), 0, m->stream[0],
			                   (const uint32_t *) m->staging, (uint32_t *) dev_rgba, m->width, m->height, (uint32_t) m->n, m->band_rows, m->local_rows);
		}
		VRM_TRY(m, hipGetLastError());
	}
	VRM_TRY(m, hipSetDevice(m->dev[0]));
	VRM_TRY(m, hipEventRecord(t1, m->stream[0]));
	for (int r = m->n - 1; r >= 0; r--) {
		VRM_TRY(m, hipSetDevice(m->dev[r]));
		VRM_TRY(m, hipStreamSynchronize(m->stream[r]));
	}
	(void) hipEventElapsedTime(&m->last_total_ms, t0, t1);
	(void) hipEventDestroy(t0); (void) hipEventDestroy(t1);
	return VR_OK;
}

int vr_hip_multi_render(vr_multi *m, const vr_params *p, uint8_t *host_rgba) {
	if (m == nullptr) return VR_ERR_INVALID;
	if (host_rgba == nullptr) return fail(m, VR_ERR_INVALID, "buffer is NULL");
	if (m->frame0 == nullptr) return fail(m, VR_ERR_NOT_READY, "call vr_hip_multi_set_window first");
	int rc = vr_hip_multi_render_device(m, p, m->frame0);
	if (rc) return rc;
	VRM_TRY(m, hipSetDevice(m->dev[0]));
	VRM_TRY(m, hipMemcpy(host_rgba, m->frame0, (size_t) m->width * m->height * 4, hipMemcpyDeviceToHost));
	return VR_OK;
}

// per-device kernel time of the last frame (n floats) and device-0 stream time from the first launch to the assembled frame
int vr_hip_multi_timing(vr_multi *m, float *per_rank_kernel_ms, float *total_ms) {
	if (m == nullptr) return VR_ERR_INVALID;
	for (int r = 0; r < m->n; r++) {
		vr_timing t;
		int rc = forward(m, r, vr_hip_timing(m->ctx[r], &t));
		if (rc) return rc;
		if (per_rank_kernel_ms) per_rank_kernel_ms[r] = t.kernel_ms;
	}
	if (total_ms) *total_ms = m->last_total_ms;
	return VR_OK;
}

}  // extern "C"
