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
//   * a copy kernel on device 0 de-interleaves the bands into the caller's frame;
//   * THREE frames in flight (vr_hip_multi_render_device_async + vr_hip_multi_sync): band buffers, staging, timing events AND STREAMS
//     exist three times — with two or more devices the frames render concurrently, each slot on streams of its own: a device's share
//     of a frame fills the chip only briefly (at 8 devices: one load of waves, and the launch lasts as long as its longest wave), and
//     the next frames' workgroups fill that tail (measured with the band sets of an N-rank run on one GPU, scripts/overlap_probe.py,
//     ms per frame with 1 / 2 / 3 / 4 streams: N = 2 1.40 / 1.18 / 1.16 / 1.20, N = 4 0.86 / 0.60 / 0.59 / 0.65, N = 8 0.57 / 0.36 /
//     0.29 / 0.35); a single device keeps one stream for every slot (a whole frame fills the chip);
//     frame i+1 also renders while the bands of frame i travel and are assembled; nothing is created or destroyed per frame.
//     vr_hip_multi_render_device / vr_hip_multi_render are the synchronous calls the reference's interface needs (async + sync).
// Built only on the public single-device ABI + the HIP runtime: nothing here touches vr_ctx internals.
//
// State of verification: the DISTINCT-device branches — ncclCommInitAll over n devices, send / recv between devices,
// hipMemcpyPeerAsync and cross-device hipStreamWaitEvent — have not run on hardware yet (the build box has one GPU).  What has:
// the split, the band map, the copy gather, the assemble kernel and the two-frame pipeline with device lists [0,0,…]; and the RCCL
// calls themselves through VR_MULTI_TRANSPORT=rccl-self (one communicator of size 1, every band sent to and received from rank 0
// inside one group — the same call sequence with peer = self).  Until the distinct-device path has run once, the FIRST frame after
// every set_window on distinct devices is self-checked: device 0 renders the other ranks' bands itself and compares them with what
// arrived (VR_MULTI_SELFCHECK=0 switches that off, =1 forces it for duplicate lists too); a wrong gather fails loudly.
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
		// A process that already holds an RCCL (torch ships one) gets THAT one: RTLD_NOLOAD only succeeds for an object that is
		// already mapped.  Otherwise load it privately (RTLD_LOCAL): its symbols must not interpose anybody else's.
		const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
		for (const char *name : names) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD); if (lib) break; }
		for (const char *name : names) { if (lib) break; lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); }
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

enum Transport { kSingle = 0, kRccl = 1, kPeerCopy = 2, kRcclSelf = 3 };
constexpr int kFrames = 3;       // frames in flight

}  // namespace

struct vr_multi {
	int n = 0;
	std::vector<int> dev;
	std::vector<vr_ctx *> ctx;
	std::vector<hipStream_t> streams[kFrames];       // [slot][rank]: the frames in flight run on streams of their own
	// per frame slot
	std::vector<hipEvent_t> rendered[kFrames];      // rank r's bands of the slot's frame are rendered (recorded on stream[r])
	std::vector<void *> local[kFrames];             // rank r's bands on its own device (rank 0: its slice of staging[slot])
	void *staging[kFrames] = {};                    // device 0: [n][local_rows][width] RGBA8
	hipEvent_t gathered[kFrames] = {};              // device 0 has read every local[r] of the slot (peer-copy transport)
	hipEvent_t t0[kFrames] = {}, t1[kFrames] = {};  // device-0 stream time of the slot's frame
	bool in_flight[kFrames] = {};
	uint64_t frames = 0;                            // frames queued since create
	void *frame0 = nullptr;                         // device 0: assembled frame of the host-buffer entry point
	void *check = nullptr;                          // device 0: self-check scratch (one band slice) + mismatch counter
	uint32_t *check_count = nullptr;
	int check_left = 0;                             // self-check this many more frames: the first kFrames + 1 after set_window on distinct devices, i.e. frame 0
	                                                // and the first REUSE of every pipeline slot (frame 0 alone says nothing about the retire ordering)
	uint32_t width = 0, height = 0, band_rows = 0, per_rank = 0, local_rows = 0;
	Transport transport = kSingle;
	bool distinct = true;
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

// Kernels of the ranks launched so far may still be writing band buffers when a later step fails: never return to the caller
// (who may free or reuse buffers) before every stream has drained.
void quiesce(vr_multi *m) {
	for (int s = 0; s < kFrames; s++)
		for (int r = 0; r < m->n && r < (int) m->streams[s].size(); r++)
			if (m->streams[s][r]) { (void) hipSetDevice(m->dev[r]); (void) hipStreamSynchronize(m->streams[s][r]); }
	(void) hipGetLastError();
	for (int s = 0; s < kFrames; s++) m->in_flight[s] = false;
}

#define VRM_TRY(m, expr)                                                                         \
	do {                                                                                         \
		hipError_t e_ = (expr);                                                                  \
		if (e_ != hipSuccess) {                                                                  \
			(void) hipGetLastError();                                                            \
			return fail((m), e_ == hipErrorOutOfMemory ? VR_ERR_ALLOC : VR_ERR_HIP, #expr, e_);  \
		}                                                                                        \
	} while (0)

// inside a frame: drain every stream before reporting
#define VRM_TRY_FRAME(m, expr)                                                                   \
	do {                                                                                         \
		hipError_t e_ = (expr);                                                                  \
		if (e_ != hipSuccess) {                                                                  \
			(void) hipGetLastError();                                                            \
			quiesce(m);                                                                          \
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

// self-check: number of differing 4-byte pixels between what a rank delivered and device 0's own render of the same bands
__global__ __launch_bounds__(256)
void compare_kernel(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, size_t words, uint32_t *__restrict__ mismatches) {
	uint32_t bad = 0;
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t) gridDim.x * 256) bad += a[i] != b[i] ? 1u : 0u;
	if (bad) atomicAdd(mismatches, bad);
}

void release_buffers(vr_multi *m) {
	if (m->n > 0) quiesce(m);
	for (int s = 0; s < kFrames; s++) {
		for (int r = 1; r < m->n; r++)
			if (m->local[s][r]) { (void) hipSetDevice(m->dev[r]); (void) hipFree(m->local[s][r]); m->local[s][r] = nullptr; }
		if (m->n > 0) (void) hipSetDevice(m->dev[0]);
		if (m->staging[s]) { (void) hipFree(m->staging[s]); m->staging[s] = nullptr; }
		if (m->n > 0) m->local[s][0] = nullptr;
	}
	if (m->n > 0) (void) hipSetDevice(m->dev[0]);
	if (m->frame0) { (void) hipFree(m->frame0); m->frame0 = nullptr; }
	if (m->check) { (void) hipFree(m->check); m->check = nullptr; }
	if (m->check_count) { (void) hipFree(m->check_count); m->check_count = nullptr; }
}

// waits for the frame in `slot` (device-0 stream time -> last_total_ms)
int retire(vr_multi *m, int slot) {
	if (!m->in_flight[slot]) return VR_OK;
	VRM_TRY_FRAME(m, hipSetDevice(m->dev[0]));
	VRM_TRY_FRAME(m, hipEventSynchronize(m->t1[slot]));
	(void) hipEventElapsedTime(&m->last_total_ms, m->t0[slot], m->t1[slot]);
	m->in_flight[slot] = false;
	return VR_OK;
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
	m->ctx.assign(n, nullptr);
	for (int s = 0; s < kFrames; s++) m->streams[s].assign(n, nullptr);
	for (int s = 0; s < kFrames; s++) { m->rendered[s].assign(n, nullptr); m->local[s].assign(n, nullptr); }
	m->distinct = true;
	for (int a = 0; a < n; a++) for (int b = a + 1; b < n; b++) if (devices[a] == devices[b]) m->distinct = false;
	for (int r = 0; r < n; r++) {
		int rc = vr_hip_create(devices[r], &m->ctx[r]);
		if (rc != VR_OK) return fail(m, rc, m->ctx[r] ? vr_hip_last_error(m->ctx[r]) : "vr_hip_create failed");
		VRM_TRY(m, hipSetDevice(devices[r]));
		VRM_TRY(m, hipStreamCreateWithFlags(&m->streams[0][r], hipStreamNonBlocking));
		for (int s = 1; s < kFrames; s++) {
			if (n >= 2) VRM_TRY(m, hipStreamCreateWithFlags(&m->streams[s][r], hipStreamNonBlocking));
			else m->streams[s][r] = m->streams[0][r];         // one device: every slot on one stream
		}
		for (int s = 0; s < kFrames; s++) VRM_TRY(m, hipEventCreateWithFlags(&m->rendered[s][r], hipEventDisableTiming));
	}
	VRM_TRY(m, hipSetDevice(devices[0]));
	for (int s = 0; s < kFrames; s++) {
		VRM_TRY(m, hipEventCreate(&m->t0[s])); VRM_TRY(m, hipEventCreate(&m->t1[s]));
		VRM_TRY(m, hipEventCreateWithFlags(&m->gathered[s], hipEventDisableTiming));
	}
	m->transport = kSingle;
	const char *force = getenv("VR_MULTI_TRANSPORT");        // "peer": the copy path (A/B, debugging); "rccl-self": see the header comment
	if (n > 1) {
		m->transport = kPeerCopy;
		for (int r = 1; r < n && m->distinct; r++) {           // direct xGMI access between device 0 and every other one
			int can = 0;
			if (hipDeviceCanAccessPeer(&can, devices[0], devices[r]) == hipSuccess && can) {
				(void) hipSetDevice(devices[0]);
				hipError_t e = hipDeviceEnablePeerAccess(devices[r], 0);
				if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void) hipGetLastError();
			}
		}
		if (m->distinct && !(force && strcmp(force, "peer") == 0) && m->rccl.load()) {
			m->comm.assign(n, nullptr);
			if (m->rccl.CommInitAll(m->comm.data(), n, devices) == ncclSuccess) m->transport = kRccl;
			else m->comm.clear();
		}
		// Test switch for a one-GPU box: every rank is the same device, ONE communicator of size 1, and the bands travel by
		// ncclSend / ncclRecv with peer = self inside one group — dlopen, communicator setup, the grouped calls and their stream
		// ordering all execute on hardware.
		if (!m->distinct && force && strcmp(force, "rccl-self") == 0) {
			bool same = true;
			for (int r = 1; r < n; r++) same = same && devices[r] == devices[0];
			if (!same || !m->rccl.load()) return fail(m, VR_ERR_INVALID, "VR_MULTI_TRANSPORT=rccl-self needs a device list that repeats ONE device, and librccl");
			m->comm.assign(1, nullptr);
			const int one = devices[0];
			if (m->rccl.CommInitAll(m->comm.data(), 1, &one) != ncclSuccess) { m->comm.clear(); return fail(m, VR_ERR_HIP, "ncclCommInitAll (1 device) failed"); }
			m->transport = kRcclSelf;
		}
	}
	return VR_OK;
}

void vr_hip_multi_destroy(vr_multi *m) {
	if (m == nullptr) return;
	quiesce(m);
	for (size_t r = 0; r < m->comm.size(); r++) if (m->comm[r]) (void) m->rccl.CommDestroy(m->comm[r]);
	release_buffers(m);
	if (m->n > 0) (void) hipSetDevice(m->dev[0]);
	for (int s = 0; s < kFrames; s++) {
		if (m->t0[s]) (void) hipEventDestroy(m->t0[s]);
		if (m->t1[s]) (void) hipEventDestroy(m->t1[s]);
		if (m->gathered[s]) (void) hipEventDestroy(m->gathered[s]);
	}
	for (int r = 0; r < m->n; r++) {
		(void) hipSetDevice(m->dev[r]);
		for (int s = 0; s < kFrames; s++) if (m->rendered[s][r]) (void) hipEventDestroy(m->rendered[s][r]);
		for (int s = 1; s < kFrames; s++) if (m->streams[s][r] && m->streams[s][r] != m->streams[0][r]) (void) hipStreamDestroy(m->streams[s][r]);
		if (m->streams[0][r]) (void) hipStreamDestroy(m->streams[0][r]);
		vr_hip_destroy(m->ctx[r]);
	}
	delete m;
}

const char *vr_hip_multi_last_error(const vr_multi *m) { return m ? m->err.c_str() : "no context"; }
int vr_hip_multi_count(const vr_multi *m) { return m ? m->n : 0; }
vr_ctx *vr_hip_multi_context(vr_multi *m, int rank) { return (m && rank >= 0 && rank < m->n) ? m->ctx[rank] : nullptr; }
const char *vr_hip_multi_transport(const vr_multi *m) {
	if (m == nullptr) return "none";
	switch (m->transport) { case kRccl: return "rccl"; case kPeerCopy: return "peer-copy"; case kRcclSelf: return "rccl-self"; default: return "single"; }
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
	VRM_TRY(m, hipMalloc(&m->frame0, (size_t) w * h * 4));
	for (int s = 0; s < kFrames; s++) {
		VRM_TRY(m, hipSetDevice(m->dev[0]));
		VRM_TRY(m, hipMalloc(&m->staging[s], slice * m->n));
		m->local[s][0] = m->staging[s];                      // device 0 renders straight into its slice
		for (int r = 1; r < m->n; r++) {
			VRM_TRY(m, hipSetDevice(m->dev[r]));
			VRM_TRY(m, hipMalloc(&m->local[s][r], slice));
		}
	}
	const char *sc = getenv("VR_MULTI_SELFCHECK");
	const bool checking = m->n > 1 && (sc ? atoi(sc) != 0 : m->distinct);
	m->check_left = checking ? (sc && atoi(sc) > 1 ? atoi(sc) : kFrames + 1) : 0;          // VR_MULTI_SELFCHECK=<n> (n > 1): that many frames
	if (checking && m->check == nullptr) {
		VRM_TRY(m, hipSetDevice(m->dev[0]));
		VRM_TRY(m, hipMalloc(&m->check, slice));
		VRM_TRY(m, hipMalloc((void **) &m->check_count, sizeof(uint32_t)));
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

int vr_hip_multi_prepare(vr_multi *m, uint32_t copies) {
	if (m == nullptr) return VR_ERR_INVALID;
	for (int r = 0; r < m->n; r++) { int rc = forward(m, r, vr_hip_prepare(m->ctx[r], copies)); if (rc) return rc; }
	return VR_OK;
}

// Queues one whole frame into `dev_rgba` on devices[0] and returns without waiting for it.  At most kFrames (three) frames are in
// flight: the call first waits for the frame queued three calls ago (its band buffers are this frame's); frames in flight must not
// share `dev_rgba`.  `consumer_stream` (a hipStream_t
// on devices[0], may be NULL) is made to wait for the assembled frame, so work queued on it afterwards may read `dev_rgba`.
int vr_hip_multi_render_device_async(vr_multi *m, const vr_params *p, void *dev_rgba, void *consumer_stream) {
	if (m == nullptr) return VR_ERR_INVALID;
	if (p == nullptr || dev_rgba == nullptr) return fail(m, VR_ERR_INVALID, "params / buffer is NULL");
	if (p->view.width != m->width || p->view.height != m->height || m->staging[0] == nullptr)
		return fail(m, VR_ERR_NOT_READY, "view dims differ from the window: call vr_hip_multi_set_window first");
	const int slot = (int) (m->frames % kFrames);
	int rc = retire(m, slot);                                // frame i - kFrames used this slot's buffers and events
	if (rc) return rc;
	const size_t slice = (size_t) m->local_rows * m->width * 4;
	void *const *local = m->local[slot].data();
	uint8_t *staging = (uint8_t *) m->staging[slot];
	const hipStream_t *stream = m->streams[slot].data();     // this slot's stream on every device
	VRM_TRY_FRAME(m, hipSetDevice(m->dev[0]));
	VRM_TRY_FRAME(m, hipEventRecord(m->t0[slot], stream[0]));
	for (int r = 0; r < m->n; r++) {
		vr_params pr = *p;
		pr.x0 = 0; pr.out_width = m->width;
		if (m->n == 1) { pr.out_rows = m->height; pr.band_rows = m->height; pr.band_stride = 1; pr.band_first = 0; }
		else { pr.out_rows = m->local_rows; pr.band_rows = m->band_rows; pr.band_stride = (uint32_t) m->n; pr.band_first = (uint32_t) r; }
		VRM_TRY_FRAME(m, hipSetDevice(m->dev[r]));
		// peer-copy transport: device 0's stream read local[r] of frame i - kFrames; this render must not overwrite it earlier.  (RCCL: the
		// send ran on stream[r] itself.)  `retire` above already waited for that whole frame, so this wait is free — it keeps the
		// ordering on the device even if the host-side wait is ever relaxed.
		if (r > 0 && m->transport == kPeerCopy && m->frames >= (uint64_t) kFrames) VRM_TRY_FRAME(m, hipStreamWaitEvent(stream[r], m->gathered[slot], 0));
		rc = forward(m, r, vr_hip_render_device(m->ctx[r], &pr, m->n == 1 ? dev_rgba : local[r], stream[r]));
		if (rc) { quiesce(m); return rc; }
		VRM_TRY_FRAME(m, hipEventRecord(m->rendered[slot][r], stream[r]));
	}
	if (m->n > 1) {
		if (m->transport == kRccl) {
			// one group: every other device sends its bands, device 0 receives them behind its own render
			if (m->rccl.GroupStart() != ncclSuccess) { quiesce(m); return fail(m, VR_ERR_HIP, "ncclGroupStart failed"); }
			for (int r = 1; r < m->n; r++) {
				ncclResult_t a = m->rccl.Send(local[r], slice, ncclUint8, 0, m->comm[r], stream[r]);
				ncclResult_t b = m->rccl.Recv(staging + slice * r, slice, ncclUint8, r, m->comm[0], stream[0]);
				if (a != ncclSuccess || b != ncclSuccess) { (void) m->rccl.GroupEnd(); quiesce(m); return fail(m, VR_ERR_HIP, "ncclSend / ncclRecv failed"); }
			}
			if (m->rccl.GroupEnd() != ncclSuccess) { quiesce(m); return fail(m, VR_ERR_HIP, "ncclGroupEnd failed"); }
		} else if (m->transport == kRcclSelf) {
			// one communicator of size 1 on stream[0]: wait for every rank's render, then send-to-self / recv-from-self per band slice
			VRM_TRY_FRAME(m, hipSetDevice(m->dev[0]));
			for (int r = 1; r < m->n; r++) VRM_TRY_FRAME(m, hipStreamWaitEvent(stream[0], m->rendered[slot][r], 0));
			for (int r = 1; r < m->n; r++) {                   // one send / recv pair per group: pairs to the same peer match in order
				if (m->rccl.GroupStart() != ncclSuccess) { quiesce(m); return fail(m, VR_ERR_HIP, "ncclGroupStart failed"); }
				ncclResult_t a = m->rccl.Send(local[r], slice, ncclUint8, 0, m->comm[0], stream[0]);
				ncclResult_t b = m->rccl.Recv(staging + slice * r, slice, ncclUint8, 0, m->comm[0], stream[0]);
				if (a != ncclSuccess || b != ncclSuccess) { (void) m->rccl.GroupEnd(); quiesce(m); return fail(m, VR_ERR_HIP, "ncclSend / ncclRecv (self) failed"); }
				if (m->rccl.GroupEnd() != ncclSuccess) { quiesce(m); return fail(m, VR_ERR_HIP, "ncclGroupEnd failed"); }
			}
		} else {
			VRM_TRY_FRAME(m, hipSetDevice(m->dev[0]));
			for (int r = 1; r < m->n; r++) {
				VRM_TRY_FRAME(m, hipStreamWaitEvent(stream[0], m->rendered[slot][r], 0));
				if (m->dev[r] == m->dev[0])
					VRM_TRY_FRAME(m, hipMemcpyAsync(staging + slice * r, local[r], slice, hipMemcpyDeviceToDevice, stream[0]));
				else
					VRM_TRY_FRAME(m, hipMemcpyPeerAsync(staging + slice * r, m->dev[0], local[r], m->dev[r], slice, stream[0]));
			}
			VRM_TRY_FRAME(m, hipEventRecord(m->gathered[slot], stream[0]));
		}
		VRM_TRY_FRAME(m, hipSetDevice(m->dev[0]));
		const size_t row_bytes = (size_t) m->width * 4;
		if (row_bytes % 16 == 0 && ((uintptr_t) dev_rgba % 16) == 0) {
			const uint32_t elems = (uint32_t) (row_bytes / 16);
			hipLaunchKernelGGL(assemble_kernel<uint4>, dim3((elems + 255) / 256, m->height), dim3(256), 0, stream[0],
			                   (const uint4 *) staging, (uint4 *) dev_rgba, elems, m->height, (uint32_t) m->n, m->band_rows, m->local_rows);
		} else {
			hipLaunchKernelGGL(assemble_kernel<uint32_t>, dim3((m->width + 255) / 256, m->height), dim3(256), 0, stream[0],
			                   (const uint32_t *) staging, (uint32_t *) dev_rgba, m->width, m->height, (uint32_t) m->n, m->band_rows, m->local_rows);
		}
		VRM_TRY_FRAME(m, hipGetLastError());
	}
	VRM_TRY_FRAME(m, hipSetDevice(m->dev[0]));
	VRM_TRY_FRAME(m, hipEventRecord(m->t1[slot], stream[0]));
	if (consumer_stream) VRM_TRY_FRAME(m, hipStreamWaitEvent((hipStream_t) consumer_stream, m->t1[slot], 0));
	m->in_flight[slot] = true;
	m->frames++;
	if (m->check_left > 0 && m->n > 1) {
		// One of the first frames on this window: device 0 renders every other rank's bands itself and compares them with what arrived.
		m->check_left--;
		VRM_TRY_FRAME(m, hipSetDevice(m->dev[0]));
		VRM_TRY_FRAME(m, hipMemsetAsync(m->check_count, 0, sizeof(uint32_t), stream[0]));
		for (int r = 1; r < m->n; r++) {
			vr_params pr = *p;
			pr.x0 = 0; pr.out_width = m->width; pr.out_rows = m->local_rows; pr.band_rows = m->band_rows; pr.band_stride = (uint32_t) m->n; pr.band_first = (uint32_t) r;
			rc = forward(m, 0, vr_hip_render_device(m->ctx[0], &pr, m->check, stream[0]));
			if (rc) { quiesce(m); return rc; }
			hipLaunchKernelGGL(compare_kernel, dim3(1024), dim3(256), 0, stream[0], (const uint32_t *) m->check,
			                   (const uint32_t *) (staging + slice * r), slice / 4, m->check_count);
			VRM_TRY_FRAME(m, hipGetLastError());
		}
		uint32_t bad = 0;
		VRM_TRY_FRAME(m, hipMemcpyAsync(&bad, m->check_count, sizeof bad, hipMemcpyDeviceToHost, stream[0]));
		VRM_TRY_FRAME(m, hipStreamSynchronize(stream[0]));
		if (bad != 0) {
			quiesce(m);
			char buf[256];
			snprintf(buf, sizeof buf, "gather self-check failed: %u pixels of the bands that arrived on device %d (%s) differ from its own render of the same rows",
			         bad, m->dev[0], vr_hip_multi_transport(m));
			return fail(m, VR_ERR_HIP, buf);
		}
	}
	return VR_OK;
}

// waits for every queued frame
int vr_hip_multi_sync(vr_multi *m) {
	if (m == nullptr) return VR_ERR_INVALID;
	// oldest first, so that last_total_ms ends up as the newest frame's
	for (int k = 0; k < kFrames; k++) { int rc = retire(m, (int) ((m->frames + (uint64_t) k) % kFrames)); if (rc) return rc; }
	for (int s = 0; s < kFrames; s++)
		for (int r = m->n - 1; r >= 0; r--) {                // sends on the other devices' streams have completed as well
			VRM_TRY(m, hipSetDevice(m->dev[r]));
			VRM_TRY(m, hipStreamSynchronize(m->streams[s][r]));
		}
	return VR_OK;
}

// whole frame into `dev_rgba` on device 0 (devices[0] of the create call); synchronous from the caller's view like every
// reference renderer call
int vr_hip_multi_render_device(vr_multi *m, const vr_params *p, void *dev_rgba) {
	int rc = vr_hip_multi_render_device_async(m, p, dev_rgba, nullptr);
	if (rc) return rc;
	return vr_hip_multi_sync(m);
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
