// One frame on all GPUs of a node from one host process: tile split, grouped ncclSend / ncclRecv of the packed tile
// buffers to the root over xGMI, reassembly on the root.  See include/mirt_node.h.  Host code only (HIP runtime + RCCL).
#include "../../include/mirt_node.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "mirt_internal.h"

namespace {

const uint32_t TILE_W = 64, TILE_H = 16; // one tile = 16 wavefront-sized 8x8 pixel blocks (the split bench.py uses)

#define HIPCHK(expr)                                                                                                      \
	do {                                                                                                                  \
		hipError_t e_ = (expr);                                                                                           \
		if (e_ != hipSuccess) return mirt_set_error(e_ == hipErrorOutOfMemory ? MIRT_E_NOMEM : MIRT_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
	} while (0)
#define NCCLCHK(expr)                                                                                 \
	do {                                                                                              \
		ncclResult_t r_ = (expr);                                                                     \
		if (r_ != ncclSuccess) return mirt_set_error(MIRT_E_DEVICE, "%s: %s", #expr, ncclGetErrorString(r_)); \
	} while (0)

struct Rank {
	int device = 0;
	mirt_scene* scene = nullptr;
	ncclComm_t comm = nullptr;
	hipStream_t stream[MIRT_NODE_SLOTS] = {};
	void* tiles[MIRT_NODE_SLOTS] = {}; // this rank's packed tile buffer per slot (the root renders into its part of `gathered`)
	size_t tilesBytes[MIRT_NODE_SLOTS] = {};
};

} // namespace

struct mirt_node_ctx {
	std::vector<Rank> ranks;
	bool rccl = false;
	// root only, per slot
	void* gathered[MIRT_NODE_SLOTS] = {};
	size_t gatheredBytes[MIRT_NODE_SLOTS] = {};
	void* image[MIRT_NODE_SLOTS] = {};
	size_t imageBytes[MIRT_NODE_SLOTS] = {};
	hipEvent_t rendered[MIRT_NODE_SLOTS] = {}; // rehearsal mode: the copies on the root's stream wait for the other ranks' frames
	void* selfTest = nullptr;
};

namespace {

int ensure(void** p, size_t* have, size_t need) {
	if (*have >= need) return MIRT_OK;
	if (*p) (void)hipFree(*p);
	*p = nullptr;
	*have = 0;
	HIPCHK(hipMalloc(p, need));
	*have = need;
	return MIRT_OK;
}

mirt_render_params rank_params(const mirt_render_params* whole, uint32_t rank, uint32_t n) {
	mirt_render_params q = *whole;
	if (n > 1) {
		q.tile_w = TILE_W;
		q.tile_h = TILE_H;
		q.first_tile = rank;
		q.tile_stride = n;
	}
	return q;
}

} // namespace

extern "C" {

int mirt_node_create(const mirt_scene_desc* desc, const int* devices, uint32_t n, mirt_node_ctx** out) {
	if (!out) return mirt_set_error(MIRT_E_INVALID, "out is NULL");
	*out = nullptr;
	if (!desc || n == 0 || n > 64) return mirt_set_error(MIRT_E_INVALID, "scene description is NULL or device count %u outside 1..64", n);
	const char* re = getenv("MIRT_NODE_REHEARSE");
	const bool rehearse = re && *re == '1';
	std::vector<int> dev(n);
	for (uint32_t i = 0; i < n; ++i) dev[i] = devices ? devices[i] : (int)i;
	if (!rehearse)
		for (uint32_t i = 0; i < n; ++i)
			for (uint32_t j = 0; j < i; ++j)
				if (dev[i] == dev[j]) return mirt_set_error(MIRT_E_INVALID, "device %d is listed twice (allowed only with MIRT_NODE_REHEARSE=1)", dev[i]);
	mirt_node_ctx* c = new (std::nothrow) mirt_node_ctx();
	if (!c) return mirt_set_error(MIRT_E_NOMEM, "out of host memory");
	c->ranks.resize(n);
	c->rccl = !rehearse;
	auto bail = [&](int e) { mirt_node_destroy(c); return e; };
	for (uint32_t i = 0; i < n; ++i) {
		Rank& r = c->ranks[i];
		r.device = dev[i];
		int e = mirt_scene_create(desc, dev[i], &r.scene); // the scene is replicated: every GPU holds all of it
		if (e) return bail(e);
		if (hipSetDevice(dev[i]) != hipSuccess) return bail(mirt_set_error(MIRT_E_DEVICE, "hipSetDevice(%d) failed", dev[i]));
		for (int s = 0; s < MIRT_NODE_SLOTS; ++s)
			if (hipStreamCreateWithFlags(&r.stream[s], hipStreamNonBlocking) != hipSuccess) return bail(mirt_set_error(MIRT_E_DEVICE, "hipStreamCreate failed on device %d", dev[i]));
		(void)mirt_set_frames_in_flight(r.scene, 1);
	}
	if (hipSetDevice(dev[0]) != hipSuccess) return bail(mirt_set_error(MIRT_E_DEVICE, "hipSetDevice(%d) failed", dev[0]));
	for (int s = 0; s < MIRT_NODE_SLOTS; ++s)
		if (hipEventCreateWithFlags(&c->rendered[s], hipEventDisableTiming) != hipSuccess) return bail(mirt_set_error(MIRT_E_DEVICE, "hipEventCreate failed"));
	if (c->rccl) {
		std::vector<ncclComm_t> comms(n);
		ncclResult_t r = ncclCommInitAll(comms.data(), (int)n, dev.data());
		if (r != ncclSuccess) return bail(mirt_set_error(MIRT_E_DEVICE, "ncclCommInitAll over %u devices: %s", n, ncclGetErrorString(r)));
		for (uint32_t i = 0; i < n; ++i) c->ranks[i].comm = comms[i];
		// self-test of the communicator: every rank contributes its rank number + 1, all must read n (n + 1) / 2
		std::vector<void*> word(n, nullptr);
		bool ok = true;
		for (uint32_t i = 0; i < n && ok; ++i) {
			const uint32_t v = i + 1;
			ok = hipSetDevice(dev[i]) == hipSuccess && hipMalloc(&word[i], 4) == hipSuccess && hipMemcpy(word[i], &v, 4, hipMemcpyHostToDevice) == hipSuccess;
		}
		if (ok) {
			ok = ncclGroupStart() == ncclSuccess;
			for (uint32_t i = 0; i < n && ok; ++i) ok = ncclAllReduce(word[i], word[i], 1, ncclUint32, ncclSum, c->ranks[i].comm, c->ranks[i].stream[0]) == ncclSuccess;
			ok = ncclGroupEnd() == ncclSuccess && ok;
		}
		for (uint32_t i = 0; i < n && ok; ++i) {
			uint32_t v = 0;
			ok = hipSetDevice(dev[i]) == hipSuccess && hipStreamSynchronize(c->ranks[i].stream[0]) == hipSuccess && hipMemcpy(&v, word[i], 4, hipMemcpyDeviceToHost) == hipSuccess && v == n * (n + 1) / 2;
		}
		for (uint32_t i = 0; i < n; ++i)
			if (word[i]) { (void)hipSetDevice(dev[i]); (void)hipFree(word[i]); }
		if (!ok) return bail(mirt_set_error(MIRT_E_DEVICE, "RCCL self-test (all-reduce of one word over %u devices) failed", n));
	}
	*out = c;
	return MIRT_OK;
}

void mirt_node_destroy(mirt_node_ctx* c) {
	if (!c) return;
	for (Rank& r : c->ranks) {
		(void)hipSetDevice(r.device);
		(void)hipDeviceSynchronize();
	}
	for (size_t i = 0; i < c->ranks.size(); ++i) {
		Rank& r = c->ranks[i];
		(void)hipSetDevice(r.device);
		if (r.comm) (void)ncclCommDestroy(r.comm);
		if (r.scene) mirt_scene_destroy(r.scene);
		for (int s = 0; s < MIRT_NODE_SLOTS; ++s) {
			if (r.tiles[s]) (void)hipFree(r.tiles[s]);
			if (r.stream[s]) (void)hipStreamDestroy(r.stream[s]);
		}
	}
	if (!c->ranks.empty()) (void)hipSetDevice(c->ranks[0].device);
	for (int s = 0; s < MIRT_NODE_SLOTS; ++s) {
		if (c->gathered[s]) (void)hipFree(c->gathered[s]);
		if (c->image[s]) (void)hipFree(c->image[s]);
		if (c->rendered[s]) (void)hipEventDestroy(c->rendered[s]);
	}
	delete c;
}

int mirt_node_uses_rccl(const mirt_node_ctx* c, uint32_t* n) {
	if (!c) return 0;
	if (n) *n = (uint32_t)c->ranks.size();
	return c->rccl ? 1 : 0;
}

int mirt_node_render_device(mirt_node_ctx* c, const mirt_render_params* p, uint32_t slot, void** d_bgr_out) {
	if (!c || !p || !d_bgr_out) return mirt_set_error(MIRT_E_INVALID, "NULL argument");
	if (slot >= MIRT_NODE_SLOTS) return mirt_set_error(MIRT_E_INVALID, "slot %u: at most %d frames in flight", slot, MIRT_NODE_SLOTS);
	if (p->struct_size != sizeof(mirt_render_params)) return mirt_set_error(MIRT_E_INVALID, "mirt_render_params.struct_size %u != %zu", p->struct_size, sizeof(mirt_render_params));
	if (p->tile_stride != 1 || p->first_tile != 0 || p->tile_w != 0 || p->tile_h != 0) return mirt_set_error(MIRT_E_INVALID, "params must describe the whole image (tile_stride 1, default tiles): the split over the GPUs is made by the library");
	const uint32_t n = (uint32_t)c->ranks.size();
	Rank& root = c->ranks[0];
	const size_t imageBytes = (size_t)p->width * p->height * 3;
	int e;
	HIPCHK(hipSetDevice(root.device));
	if (n == 1) {
		if ((e = ensure(&c->image[slot], &c->imageBytes[slot], imageBytes))) return e;
		if ((e = mirt_render_device(root.scene, p, c->image[slot], imageBytes, root.stream[slot]))) return e;
		*d_bgr_out = c->image[slot];
		return MIRT_OK;
	}
	const mirt_render_params p0 = rank_params(p, 0, n);
	const size_t perRank = mirt_tile_buffer_bytes(&p0); // every rank's buffer has the size of rank 0's, the largest
	if (perRank == 0) return mirt_set_error(MIRT_E_INVALID, "image size %ux%u not supported", p->width, p->height);
	if ((e = ensure(&c->gathered[slot], &c->gatheredBytes[slot], perRank * n))) return e;
	if ((e = ensure(&c->image[slot], &c->imageBytes[slot], imageBytes))) return e;
	// every rank's share, each on its own device and stream; the root renders straight into its place in the landing area
	for (uint32_t i = 0; i < n; ++i) {
		Rank& r = c->ranks[i];
		HIPCHK(hipSetDevice(r.device));
		void* dst = c->gathered[slot];
		if (i > 0) {
			if ((e = ensure(&r.tiles[slot], &r.tilesBytes[slot], perRank))) return e;
			dst = r.tiles[slot];
		}
		const mirt_render_params q = rank_params(p, i, n);
		if ((e = mirt_render_device(r.scene, &q, dst, perRank, r.stream[slot]))) return e;
	}
	// the one exchange step of the frame: N-1 buffers into the root, each over its own xGMI link
	if (c->rccl) {
		NCCLCHK(ncclGroupStart());
		for (uint32_t i = 1; i < n; ++i) {
			Rank& r = c->ranks[i];
			NCCLCHK(ncclSend(r.tiles[slot], perRank, ncclUint8, 0, r.comm, r.stream[slot]));
			NCCLCHK(ncclRecv((uint8_t*)c->gathered[slot] + (size_t)i * perRank, perRank, ncclUint8, (int)i, root.comm, root.stream[slot]));
		}
		NCCLCHK(ncclGroupEnd());
	}
	else {
		for (uint32_t i = 1; i < n; ++i) {
			Rank& r = c->ranks[i];
			HIPCHK(hipSetDevice(r.device));
			HIPCHK(hipEventRecord(c->rendered[slot], r.stream[slot]));
			HIPCHK(hipSetDevice(root.device));
			HIPCHK(hipStreamWaitEvent(root.stream[slot], c->rendered[slot], 0));
			HIPCHK(hipMemcpyAsync((uint8_t*)c->gathered[slot] + (size_t)i * perRank, r.tiles[slot], perRank, hipMemcpyDeviceToDevice, root.stream[slot]));
		}
	}
	HIPCHK(hipSetDevice(root.device));
	if ((e = mirt_assemble_tiles(root.device, &p0, n, c->gathered[slot], perRank, c->image[slot], root.stream[slot]))) return e;
	*d_bgr_out = c->image[slot];
	return MIRT_OK;
}

int mirt_node_sync(mirt_node_ctx* c, uint32_t slot) {
	if (!c || slot >= MIRT_NODE_SLOTS) return mirt_set_error(MIRT_E_INVALID, "bad context or slot");
	// the senders' streams first (their sends complete with the root's receives), then the root's
	for (size_t i = c->ranks.size(); i-- > 0;) {
		HIPCHK(hipSetDevice(c->ranks[i].device));
		HIPCHK(hipStreamSynchronize(c->ranks[i].stream[slot]));
	}
	return MIRT_OK;
}

int mirt_node_render(mirt_node_ctx* c, const mirt_render_params* p, uint8_t* bgr_out, mirt_stats* stats) {
	if (!bgr_out) return mirt_set_error(MIRT_E_INVALID, "bgr_out is NULL");
	void* d = nullptr;
	int e = mirt_node_render_device(c, p, 0, &d);
	if (e) return e;
	Rank& root = c->ranks[0];
	HIPCHK(hipSetDevice(root.device));
	HIPCHK(hipMemcpyAsync(bgr_out, d, (size_t)p->width * p->height * 3, hipMemcpyDeviceToHost, root.stream[0]));
	if ((e = mirt_node_sync(c, 0))) return e;
	if (stats) {
		memset(stats, 0, sizeof *stats);
		for (Rank& r : c->ranks) {
			mirt_stats s;
			if ((e = mirt_get_stats(r.scene, r.stream[0], &s))) return e;
			stats->primary_rays += s.primary_rays;
			stats->shadow_rays += s.shadow_rays;
			stats->reflection_rays += s.reflection_rays;
			stats->shadow_rays_traced += s.shadow_rays_traced;
			stats->reflection_rays_traced += s.reflection_rays_traced;
			stats->primary_hits += s.primary_hits;
			if (s.device_ms > stats->device_ms) stats->device_ms = s.device_ms; // the slowest GPU's share
		}
	}
	return MIRT_OK;
}

} // extern "C"
