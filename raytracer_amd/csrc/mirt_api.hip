// C ABI of the trace path (include/mirt.h): scene upload, frame orchestration, parity hooks.
// Host code here only re-packs data and launches kernels; all tracing arithmetic lives in mirt_kernels.hip.
#include "../../include/mirt.h"
#include "mirt_kernels.hip"
#include "mirt_bvh.h"
#include "mirt_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#pragma clang fp contract(off)

// blocks per CU in a trace launch = blocks of k_trace that fit a CU at once (7 waves per SIMD): all of them start
// together and share the work through the launch's cursor
#ifndef GRID_BLOCKS_PER_CU
#define GRID_BLOCKS_PER_CU 7
#endif

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	g_err = buf;
	return code;
}

} // namespace

int mirt_set_error(int code, const char* fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	g_err = buf;
	return code;
}

namespace {

#define HIPCHK(expr)                                                                                          \
	do {                                                                                                      \
		hipError_t e_ = (expr);                                                                               \
		if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? MIRT_E_NOMEM : MIRT_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
	} while (0)

struct DevBuf {
	void* p = nullptr;
	size_t bytes = 0;
	int ensure(size_t need) {
		if (need <= bytes) return MIRT_OK;
		if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
		HIPCHK(hipMalloc(&p, need));
		bytes = need;
		return MIRT_OK;
	}
	void release() {
		if (p) (void)hipFree(p);
		p = nullptr;
		bytes = 0;
	}
};

struct Workspace {
	DevBuf hitO, hitD, hitN, shRes, active, pathDepth, litL, litS, counters, out, ovf;
	void release() {
		hitO.release(); hitD.release(); hitN.release(); shRes.release(); active.release(); pathDepth.release();
		litL.release(); litS.release(); counters.release(); out.release(); ovf.release();
	}
};

// host-side float helpers, same formulas as the device ones (GLM 0.9.8.5 scalar code)
struct h3 { float x, y, z; };
inline h3 H3(float x, float y, float z) { h3 r = {x, y, z}; return r; }
inline h3 hsub(h3 a, h3 b) { return H3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline float hdot(h3 a, h3 b) { float x = a.x * b.x, y = a.y * b.y, z = a.z * b.z; return x + y + z; }
inline h3 hcross(h3 a, h3 b) { return H3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }

// Coefficients of tri_hit's filter: out = { v3, E } { a0, c1 } { a1, K } with, for A_i = cross(N, e_i) and
// C_i = dot(A_i, v3 - v_i'):  a_i = f A_i,  c1 = f C_1,  K = f (C_1 + C_2),  f = sign(area) 2^-g,
//   2^g >= sum_c |N_c| (|e_a| + |e_b|) for every edge, and >= |N|_1 L;   E = 2^-18 L,  L = largest edge extent;
// e_i are the reference's float edge vectors and v_i' the vertex its i-th test subtracts (Triangle.cpp:148-150, 163-165).
// Everything is evaluated in double (products of floats are exact there) and rounded once.  A triangle whose magnitudes
// could make either evaluation overflow or underflow, or make (s/2)/area underflow for an s the filter calls certain,
// gets E = NaN: the filter then decides nothing and the reference's operations run.
inline void tri_filter(float* out, h3 N, h3 v1, h3 v2, h3 v3, float area) {
	memset(out, 0, 12 * sizeof(float));
	out[0] = v3.x; out[1] = v3.y; out[2] = v3.z;
	out[3] = NAN;
	if (!std::isfinite(area) || area == 0.0f) return;
	const float all[12] = {N.x, N.y, N.z, v1.x, v1.y, v1.z, v2.x, v2.y, v2.z, v3.x, v3.y, v3.z};
	for (float f : all) if (!std::isfinite(f) || fabsf(f) > 1e10f) return;
	const h3 ef[3] = {hsub(v3, v2), hsub(v1, v3), hsub(v2, v1)};
	const h3 vr[3] = {v3, v1, v2};
	const double n[3] = {N.x, N.y, N.z};
	double L = 0.0, top = 0.0;
	double A[3][3], C[3];
	for (int i = 0; i < 3; ++i) {
		const double e[3] = {ef[i].x, ef[i].y, ef[i].z};
		const double dv[3] = {(double)v3.x - vr[i].x, (double)v3.y - vr[i].y, (double)v3.z - vr[i].z};
		const double emax = std::max(fabs(e[0]), std::max(fabs(e[1]), fabs(e[2])));
		if (!(emax >= 1e-12)) return;
		L = std::max(L, emax);
		A[i][0] = n[1] * e[2] - e[1] * n[2];
		A[i][1] = n[2] * e[0] - e[2] * n[0];
		A[i][2] = n[0] * e[1] - e[0] * n[1];
		C[i] = A[i][0] * dv[0] + A[i][1] * dv[1] + A[i][2] * dv[2];
		top = std::max(top, fabs(n[0]) * (fabs(e[1]) + fabs(e[2])) + fabs(n[1]) * (fabs(e[0]) + fabs(e[2])) + fabs(n[2]) * (fabs(e[0]) + fabs(e[1])));
	}
	L *= 1.0 + 0x1p-20; // the float edges are rounded differences
	top = std::max(top, (fabs(n[0]) + fabs(n[1]) + fabs(n[2])) * L);
	if (!(top > 0.0)) return;
	int ex = 0;
	(void)frexp(top, &ex); // top = m 2^ex, m in [0.5, 1): 2^ex > top
	const double f = ldexp(area > 0.0f ? 1.0 : -1.0, -ex);
	// a certain sign means |s_i| >= 2^g (1 - 27.2/64) eps >= 2^g 0.5 2^-18 L; (s_i / 2) / area must stay a normal float
	if (!(ldexp(0.5 * 0x1p-18 * L, ex) / (2.0 * fabs((double)area)) >= 1e-37)) return;
	float tmp[8] = {(float)(A[0][0] * f), (float)(A[0][1] * f), (float)(A[0][2] * f), (float)(C[1] * f),
	                (float)(A[1][0] * f), (float)(A[1][1] * f), (float)(A[1][2] * f), (float)((C[1] + C[2]) * f)};
	for (float t : tmp) if (!std::isfinite(t)) return;
	memcpy(out + 4, tmp, sizeof tmp);
	out[3] = (float)(L * 0x1p-18 * (1.0 + 0x1p-20));
}
inline h3 hnorm(h3 a) { float s = 1.0f / sqrtf(hdot(a, a)); return H3(a.x * s, a.y * s, a.z * s); }

}  // namespace

// Working set of the frames issued on one stream.  Frames on different streams own different workspaces, so they may
// overlap on the GPU (the short, latency-bound deep-bounce launches of one frame run beside the bulk of the next).
#if MIRT_TIMELINE
#define DBG_WORDS (40 + 4 * 8192) // (timeline builds: four words per wave of a k_frame launch behind the counters, query 1000 + i)
#else
#define DBG_WORDS 40      // 64-bit words of mirt_scene::dbg: 0..31 event counters of diagnostic builds, 32..33 the clock probe
#endif
#define DBG_CLOCK 32

struct FrameCtx {
	hipStream_t stream = nullptr;
	Workspace ws;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	// last frame, for mirt_get_stats / mirt_get_launch_times
	uint64_t lastPrimary = 0;
	uint32_t lastFeatures = 0;
	bool lastValid = false;
	bool lastSingle = false; // the last frame ran as one launch (k_frame): its counters are laid out differently
	// The counters buffer holds TWO blocks of CTR_COUNT words.  A frame uses block ctrPhase, which is all zeros when the
	// frame starts: the frame before it on this stream cleared it (k_frame does that itself, the per-depth path with a
	// memset at its own start), and both were cleared when the buffer was allocated.
	uint32_t ctrPhase = 0;
	const uint32_t* lastCtr = nullptr; // the block of the last frame
	std::vector<hipEvent_t> launchEv;
	std::vector<int32_t> launchKind;
	uint32_t nLaunches = 0;
};

// ---- scratch of the entry point that allocates nothing (mirt_render_device_noalloc) ------------------------------------------
// One file-scope array per GPU (every device the code object is loaded on holds its own copy until the process ends); it
// takes the place of the per-stream workspace.  `scratch` below is only the host-side handle of the symbol: the device
// address is looked up once per device, with that device current (scratch_init, called from mirt_scene_create), and only
// that address is ever given to kernels, memsets or copies.
static __device__ __align__(256) unsigned char scratch[MIRT_NOALLOC_SCRATCH_BYTES];
#define MIRT_MAX_DEVICES 64
static unsigned char* g_scratchPtr[MIRT_MAX_DEVICES];

// Device buffer -> pageable host buffer through page-locked staging, band by band, with helper threads (see mirt_scene).
struct CopyBack {
	static const int MAX_BANDS = 32;
	int device;
	void* staging = nullptr;
	size_t stagingBytes = 0;
	hipEvent_t ev[MAX_BANDS];
	int nEv = 0;
	// one job at a time; workers sleep on the condition variable between jobs
	std::vector<std::thread> workers;
	std::mutex mu;
	std::condition_variable cv, cvDone;
	uint64_t jobNo = 0;
	bool quit = false;
	int busy = 0;
	// the job
	uint8_t* dst = nullptr;
	size_t bytes = 0, band = 0;
	int nBands = 0;
	std::atomic<int> next{0};
	std::atomic<bool> failed{false};
	bool clean[MAX_BANDS];    // the band holds nothing but the zeros it was cleared to: it is not fetched, the destination is zeroed
	uint32_t* flags = nullptr; // page-locked: the frame's notes of the parts it wrote (CTR_NPARTS words)

	CopyBack(int dev, int nThreads) : device(dev) {
		for (int i = 0; i < nThreads; ++i) workers.emplace_back([this]() { loop(); });
	}
	~CopyBack() {
		{ std::lock_guard<std::mutex> lk(mu); quit = true; }
		cv.notify_all();
		for (auto& t : workers) t.join();
		for (int i = 0; i < nEv; ++i) (void)hipEventDestroy(ev[i]);
		if (staging) (void)hipHostFree(staging);
		if (flags) (void)hipHostFree(flags);
	}
	// A band is moved in PIECES quarters, whoever gets them: wait for the band's DMA, move the quarter.  (One thread per whole
	// band leaves the last band's 2 MB to a single memcpy after the last DMA has ended: 0.2 ms at the end of every frame.)
	static const int PIECES = 4;
	void drain() {
		for (;;) {
			const int j = next.fetch_add(1);
			if (j >= nBands * PIECES) return;
			const int k = j / PIECES, q = j % PIECES;
			const size_t at = (size_t)k * band, n = at + band <= bytes ? band : bytes - at;
			const size_t piece = ((n + PIECES - 1) / PIECES + 63) & ~(size_t)63, a = (size_t)q * piece;
			if (clean[k]) { if (a < n) memset(dst + at + a, 0, a + piece <= n ? piece : n - a); continue; }
			if (hipEventSynchronize(ev[k]) != hipSuccess) { failed = true; continue; }
			if (a < n) memcpy(dst + at + a, (const uint8_t*)staging + at + a, a + piece <= n ? piece : n - a);
		}
	}
	void loop() {
		(void)hipSetDevice(device);
		uint64_t seen = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&]() { return quit || jobNo != seen; });
				if (quit) return;
				seen = jobNo;
			}
			drain();
			{ std::lock_guard<std::mutex> lk(mu); --busy; }
			cvDone.notify_one();
		}
	}
	// false: staging or events could not be had (nothing was copied).
	// dNotes != NULL: CTR_NPARTS words on the device, written by the frame that precedes this call on the stream; word i != 0
	// says that bytes [i << shift, (i + 1) << shift) of the output hold something else than the zeros they were cleared to.
	// Bands without such a part are not fetched at all.  The price is that the copies cannot be queued behind the frame in
	// advance: the host waits for the frame's end and its notes (128 bytes) first.
	bool run(const void* dsrc, uint8_t* hdst, size_t n, hipStream_t st, const uint32_t* dNotes = nullptr, uint32_t shift = 0) {
		if (n > stagingBytes) {
			if (staging) (void)hipHostFree(staging);
			staging = nullptr;
			stagingBytes = 0;
			if (hipHostMalloc(&staging, n, hipHostMallocDefault) != hipSuccess) { staging = nullptr; return false; }
			stagingBytes = n;
		}
		static const size_t bandBytes = []() { const char* e = getenv("MIRT_COPY_BAND_MB"); const int v = e ? atoi(e) : 0; return (size_t)(v >= 1 && v <= 64 ? v : 2) << 20; }();
		int want = (int)((n + bandBytes - 1) / bandBytes); // bands of about 2 MB: 1 MB bands cost more per copy than they gain at the end (+0.1 ms per 4K frame)
		want = want < 1 ? 1 : (want > MAX_BANDS ? MAX_BANDS : want);
		while (nEv < want) { if (hipEventCreateWithFlags(&ev[nEv], hipEventDisableTiming) != hipSuccess) return false; ++nEv; }
		band = ((n + want - 1) / want + 255) & ~(size_t)255;
		nBands = (int)((n + band - 1) / band);
		dst = hdst;
		bytes = n;
		failed = false;
		for (int k = 0; k < nBands; ++k) clean[k] = false;
		if (dNotes) {
			if (!flags && hipHostMalloc((void**)&flags, CTR_NPARTS * 4, hipHostMallocDefault) != hipSuccess) { flags = nullptr; return false; }
			if (hipMemcpyAsync(flags, dNotes, CTR_NPARTS * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return false;
			for (int k = 0; k < nBands; ++k) {
				const size_t at = (size_t)k * band, len = at + band <= n ? band : n - at;
				bool any = false;
				for (size_t i = at >> shift; i <= (at + len - 1) >> shift; ++i) any = any || i >= CTR_NPARTS || flags[i] != 0;
				clean[k] = !any;
			}
		}
		for (int k = 0; k < nBands; ++k) {
			if (clean[k]) continue;
			const size_t at = (size_t)k * band, len = at + band <= n ? band : n - at;
			if (hipMemcpyAsync((uint8_t*)staging + at, (const uint8_t*)dsrc + at, len, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
			if (hipEventRecord(ev[k], st) != hipSuccess) return false;
		}
		next = 0;
		{ std::lock_guard<std::mutex> lk(mu); busy = (int)workers.size(); ++jobNo; }
		cv.notify_all();
		drain(); // the calling thread moves bands too
		{ std::unique_lock<std::mutex> lk(mu); cvDone.wait(lk, [&]() { return busy == 0; }); }
		if (failed) return hipStreamSynchronize(st) == hipSuccess && hipMemcpy(hdst, dsrc, n, hipMemcpyDeviceToHost) == hipSuccess;
		return true;
	}
};

// mirt_render_png: PNG rows made on the device behind the frame, brought over in bands, deflated as they arrive.
struct PngPipe {
	static const int MAX_BANDS = 64;
	void* staging = nullptr;      // page-locked: the PNG rows of the whole image
	size_t stagingBytes = 0;
	uint32_t* flags = nullptr;    // page-locked: MIRT_PNG_MAX_STRIPES words, 1 = the stripe holds something
	DevBuf rows, dflags;
	hipEvent_t evFlags = nullptr, ev[MAX_BANDS];
	int nEv = 0;
	~PngPipe() {
		if (staging) (void)hipHostFree(staging);
		if (flags) (void)hipHostFree(flags);
		if (evFlags) (void)hipEventDestroy(evFlags);
		for (int i = 0; i < nEv; ++i) (void)hipEventDestroy(ev[i]);
		rows.release();
		dflags.release();
	}
};

// The 8x8 pixel blocks of one frame geometry that can see the scene (see visible_blocks), on the device.
struct BlockList {
	uint32_t width, height, tileW, tileH, first, stride, packed;
	DevBuf list;
	uint32_t count = 0;
	bool all = false; // every block is visible (or the scene reaches behind the eye plane): no list is used
	bool deep = false; // a second copy of the list follows the first: the blocks deepest inside the visible region first
};

struct mirt_scene {
	int device = 0;
	int numCUs = 256;
	DScene ds;
	DevBuf nodes, leafs, bvh, memb, tris, triX, triInfo, spheres, sphInfo, materials, lights, dbg;
	DevBuf hookOvf;          // overflow area of the synchronous parity hooks (mirt_trace_rays, mirt_trace_primary_ids)
	uint32_t ovfLevels = 0;  // traversal-stack entries a walk can need beyond the LDS part (0: none)
	uint32_t ovfStride = 0;  // threads of a trace launch
	bool nested = true;      // reference boxes nested: the primitive hierarchy may be used
	uint32_t bvhDepth = 0;
	bool deepBvh = false;    // hierarchy deeper than the LDS part of the traversal stack
	hipStream_t stream = nullptr; // used by the synchronous entry points
	hipStream_t probeStream = nullptr; // mirt_clock_probe_start's own stream (created at its first call)
	uint32_t nLights = 0;
	int32_t maxDepth = 0;
	uint32_t wideDepth = 0;
	uint32_t refillMin = 64; // < 64: finished lanes of a wave are refilled once this many are idle (k_trace<REFILL>)
	uint32_t diag[8] = {0}; // builder facts: binary interior nodes, binary leaves, wide nodes with 1..4 children
	std::vector<FrameCtx*> ctxs; // one per stream that has rendered with this scene
	bool profiling = false;
	// mirt_render's copy back: the frame goes to a page-locked staging buffer in bands (one DMA each, PCIe speed) and a few
	// host threads move each band into the caller's pageable buffer while the next bands are still on their way.  HIP's own
	// copy to pageable memory takes 1.4 ms for the 24.9 MB of a 4K frame; page-locking the caller's buffer in place
	// (hipHostRegister) is refused on the machines this was developed on.  MIRT_COPY_THREADS=0 goes back to the plain copy.
	CopyBack* copyBack = nullptr;
	PngPipe* png = nullptr;
	int copyThreads = 6;
	// Boxes that together hold every primitive (a breadth-first cut through the hierarchy, 6 floats each), and the lists of
	// visible 8x8 pixel blocks made from them, one per frame geometry that was rendered (visible_blocks)
	std::vector<float> cutBoxes;
	std::vector<BlockList*> blockLists;
	uint32_t framesHint = 0;  // frames the caller says it keeps in flight (mirt_set_frames_in_flight); 0, 1: not said
	bool singleLaunch = true; // frames run as one launch (k_frame) where that kernel applies; MIRT_SINGLE_LAUNCH=0: one launch per depth
};

namespace {

// ---- re-pack the flattened Partition tree into wide nodes ----------------------------------------------------------
// Packs the reference tree as it is (MODE_REFWALK walks it): 64-byte nodes holding both children's boxes, chains of
// single-child nodes skipped.  The recursion runs on one thread near the root and leaves subtrees of at most TASK_NODES
// flat nodes to the thread pool; every subtree is packed into an array of its own, and the arrays are put one after the
// other behind the top's with their node references moved by their offsets.
struct Packer {
	const mirt_scene_desc* d;
	std::vector<uint32_t> leafPos;   // flat node index -> position in the leaf stream
	static const uint32_t TASK_NODES = 1u << 16;
	struct Out {
		std::vector<float> wide;     // 16 floats per wide node
		uint32_t maxDepth = 0;
		bool bad = false;
	};
	struct Task { uint32_t node, depth, end, parent; int slot; };

	static bool isLeaf(const mirt_node& n) { return (n.b & MIRT_LEAF_FLAG) != 0; }
	static bool inside(const mirt_node& c, const mirt_node& p) {
		for (int k = 0; k < 3; ++k)
			if (!(c.bmin[k] >= p.bmin[k] && c.bmax[k] <= p.bmax[k])) return false;
		return true;
	}
	// follow chains of single-child interior nodes whose child box is nested in the parent box
	uint32_t skipChain(uint32_t i) const {
		for (;;) {
			const mirt_node& n = d->nodes[i];
			if (isLeaf(n)) return i;
			const bool hasL = n.a != MIRT_NO_CHILD, hasR = n.b != MIRT_NO_CHILD;
			if (hasL == hasR) return i;
			const uint32_t c = hasL ? n.a : n.b;
			if (!inside(d->nodes[c], n)) return i;
			i = c;
		}
	}
	// `end`: one past the last flat index of the subtree (flat nodes are in pre-order); only the size of tasks depends on it
	uint32_t emit(Out& o, uint32_t flatIdx, uint32_t depth, uint32_t end, std::vector<Task>* tasks) const {
		if (depth > o.maxDepth) o.maxDepth = depth;
		if (depth > 4000) { o.bad = true; return REF_NONE; }
		const uint32_t me = (uint32_t)(o.wide.size() / 16);
		o.wide.resize(o.wide.size() + 16, 0.0f);
		const mirt_node n = d->nodes[flatIdx];
		uint32_t refs[2] = {REF_NONE, REF_NONE};
		const uint32_t ch[2] = {n.a, n.b};
		const uint32_t chEnd[2] = {n.b != MIRT_NO_CHILD && n.b > n.a ? n.b : end, end};
		float boxes[2][6];
		for (int s = 0; s < 2; ++s) {
			for (int k = 0; k < 6; ++k) boxes[s][k] = 0.0f;
			if (ch[s] == MIRT_NO_CHILD) continue;
			const uint32_t c = skipChain(ch[s]);
			const mirt_node& cn = d->nodes[c];
			for (int k = 0; k < 3; ++k) { boxes[s][k] = cn.bmin[k]; boxes[s][3 + k] = cn.bmax[k]; }
			if (isLeaf(cn)) refs[s] = REF_LEAF | leafPos[c];
			else if (tasks && chEnd[s] > c && chEnd[s] - c <= TASK_NODES) tasks->push_back(Task{c, depth + 1, chEnd[s], me, s}); // reference patched in later
			else refs[s] = emit(o, c, depth + 1, chEnd[s], tasks);
		}
		float* w = o.wide.data() + (size_t)me * 16;
		memcpy(w + 0, boxes[0], 12); memcpy(w + 3, &refs[0], 4);
		memcpy(w + 4, boxes[0] + 3, 12); memcpy(w + 7, &refs[1], 4);
		memcpy(w + 8, boxes[1], 12);
		memcpy(w + 12, boxes[1] + 3, 12);
		return me;
	}
	// the whole tree below interior node `root` into `wide`; returns false if it is too deep
	bool run(uint32_t root, std::vector<float>& wide, uint32_t& maxDepth) const {
		Out top;
		std::vector<Task> tasks;
		emit(top, root, 0, d->n_nodes, &tasks);
		std::vector<Out> sub(tasks.size());
		MirtPool::get().run((uint32_t)tasks.size(), [&](uint32_t t) {
			sub[t].wide.reserve((size_t)(tasks[t].end - tasks[t].node) * 8);
			emit(sub[t], tasks[t].node, tasks[t].depth, tasks[t].end, nullptr);
		});
		std::vector<size_t> base(tasks.size() + 1);
		base[0] = top.wide.size() / 16;
		bool bad = top.bad;
		maxDepth = top.maxDepth;
		for (size_t t = 0; t < tasks.size(); ++t) {
			base[t + 1] = base[t] + sub[t].wide.size() / 16;
			bad = bad || sub[t].bad;
			if (sub[t].maxDepth > maxDepth) maxDepth = sub[t].maxDepth;
		}
		if (bad || base[tasks.size()] >= REF_LEAF) return false;
		wide.swap(top.wide);
		wide.resize(base[tasks.size()] * 16);
		for (size_t t = 0; t < tasks.size(); ++t) {
			const uint32_t r = (uint32_t)base[t];
			memcpy(wide.data() + (size_t)tasks[t].parent * 16 + (tasks[t].slot ? 7 : 3), &r, 4);
		}
		MirtPool::get().run((uint32_t)tasks.size(), [&](uint32_t t) {
			float* dst = wide.data() + base[t] * 16;
			memcpy(dst, sub[t].wide.data(), sub[t].wide.size() * 4);
			const uint32_t off = (uint32_t)base[t];
			for (size_t i = 0; i < sub[t].wide.size() / 16; ++i)
				for (int k = 3; k <= 7; k += 4) {
					uint32_t r;
					memcpy(&r, dst + i * 16 + k, 4);
					if (r != REF_NONE && !(r & REF_LEAF)) { r += off; memcpy(dst + i * 16 + k, &r, 4); }
				}
			std::vector<float>().swap(sub[t].wide);
		});
		return true;
	}
};

// Collapses the binary primitive hierarchy into 4-wide nodes with quantised child boxes (layout in mirt_device.h).
// Parallel in the same way as Packer: one thread near the root, subtrees over at most TASK_PRIMS primitives on the pool.
#ifndef MIRT_BVH_FAMILY_ORDER
#define MIRT_BVH_FAMILY_ORDER 1
#endif
struct Collapser {
	const BvhBuilder& bb;
	const std::vector<uint32_t>& kindIndex;
	static const uint32_t TASK_PRIMS = 1u << 15;
	struct Out {
		std::vector<uint32_t> w;     // 16 words per node
		uint32_t maxDepth = 0, maxStack = 0;
		uint32_t diag[8] = {0};
	};
	struct Task { int32_t node; uint32_t depth, acc, parent; int slot; };

	uint32_t leafRef(const BvhNode& n) const {
		return BVH_LEAF | (n.kind ? BVH_SPHERES : 0u) | ((n.count - 1) << BVH_COUNT_SHIFT) | kindIndex[n.first];
	}
	// acc: entries the walk can hold on its stack when it arrives at this node (siblings left behind by the ancestors)
	// slot: the place kept for this node by its parent (NO_SLOT: the next free one).  The interior children of a node get
	// places next to each other before any of them is filled in: a walk that goes from a node to one child and later to
	// its siblings finds them in the same or the neighbouring cache line (depth-first order puts only the first child near).
	static const uint32_t NO_SLOT = 0xffffffffu;
	uint32_t emit(Out& o, int32_t i, uint32_t depth, uint32_t acc, std::vector<Task>* tasks, uint32_t slot = NO_SLOT) const {
		if (depth > o.maxDepth) o.maxDepth = depth;
		uint32_t me = slot;
		if (me == NO_SLOT) {
			me = (uint32_t)(o.w.size() / 16);
			o.w.resize(o.w.size() + 16, 0u);
		}
		// up to four children: open the interior child with the largest surface area until none is left or there are four
		int32_t ch[4] = { bb.nodes[i].left, bb.nodes[i].right, -1, -1 };
		int n = 2;
		while (n < 4) {
			int best = -1;
			float bestArea = -1.0f;
			for (int k = 0; k < n; ++k) {
				const BvhNode& c = bb.nodes[ch[k]];
				if (c.left < 0) continue;
				const float a = BvhBuilder::area(c.lo, c.hi);
				if (a > bestArea) { bestArea = a; best = k; }
			}
			if (best < 0) break;
			const BvhNode c = bb.nodes[ch[best]];
			ch[best] = c.left;
			ch[n++] = c.right;
		}
		const BvhNode& nd = bb.nodes[i];
		uint32_t w[16] = {0};
		uint32_t ex[3];
		float scale[3];
		for (int k = 0; k < 3; ++k) {
			const double ext = (double)nd.hi[k] - (double)nd.lo[k];
			int e = 0;
			(void)frexp(ext / 255.0, &e); // ext/255 = m * 2^e, m in [0.5, 1): 2^e >= ext/255
			int biased = e + 127;
			if (!(ext > 0.0)) biased = 1;
			if (biased < 1) biased = 1;
			if (biased > 200) biased = 200;
			ex[k] = (uint32_t)biased;
			scale[k] = ldexpf(1.0f, biased - 127);
			memcpy(&w[k], &nd.lo[k], 4);
		}
		w[3] = ((ex[0] - 127u) & 0xffu) | (((ex[1] - 127u) & 0xffu) << 8) | (((ex[2] - 127u) & 0xffu) << 16); // signed bytes
		uint32_t refs[4] = { BVH_DONE, BVH_DONE, BVH_DONE, BVH_DONE };
		for (int c = 0; c < 4; ++c) {
			uint32_t qlo[3] = {255, 255, 255}, qhi[3] = {0, 0, 0};
			if (c < n) {
				const BvhNode& cn = bb.nodes[ch[c]];
				for (int k = 0; k < 3; ++k) {
					double a = floor(((double)cn.lo[k] - (double)nd.lo[k]) / (double)scale[k]);
					double b = ceil(((double)cn.hi[k] - (double)nd.lo[k]) / (double)scale[k]);
					if (a < 0.0) a = 0.0;
					if (b > 255.0) b = 255.0; // cannot happen: 255 * scale >= the node's extent
					if (b < a) b = a;
					qlo[k] = (uint32_t)a;
					qhi[k] = (uint32_t)b;
				}
			}
			w[4] |= qlo[0] << (8 * c); w[5] |= qlo[1] << (8 * c); w[6] |= qlo[2] << (8 * c);
			w[7] |= qhi[0] << (8 * c); w[8] |= qhi[1] << (8 * c); w[9] |= qhi[2] << (8 * c);
		}
		o.diag[1 + n]++;
		const uint32_t below = acc + (uint32_t)n - 1u; // a step pushes every hit child but the one it goes to
		if (below > o.maxStack) o.maxStack = below;
		memcpy(o.w.data() + (size_t)me * 16, w, sizeof w);
#if MIRT_BVH_FAMILY_ORDER
		uint32_t place[4] = {NO_SLOT, NO_SLOT, NO_SLOT, NO_SLOT};
		for (int c = 0; c < n; ++c) {
			const BvhNode& cn = bb.nodes[ch[c]];
			if (cn.left < 0 || (tasks && cn.count <= TASK_PRIMS)) continue;
			place[c] = (uint32_t)(o.w.size() / 16);
			o.w.resize(o.w.size() + 16, 0u);
		}
#else
		const uint32_t place[4] = {NO_SLOT, NO_SLOT, NO_SLOT, NO_SLOT};
#endif
		for (int c = 0; c < n; ++c) {
			const BvhNode& cn = bb.nodes[ch[c]];
			if (cn.left < 0) refs[c] = leafRef(cn);
			else if (tasks && cn.count <= TASK_PRIMS) tasks->push_back(Task{ch[c], depth + 1, below, me, c}); // patched in later
			else refs[c] = emit(o, ch[c], depth + 1, below, tasks, place[c]);
		}
		memcpy(o.w.data() + (size_t)me * 16 + 10, refs, sizeof refs);
		return me;
	}
	// the hierarchy below interior node 0; returns the root's node index (0)
	uint32_t run(std::vector<uint32_t>& out, uint32_t& maxDepth, uint32_t& maxStack, uint32_t* diag) const {
		Out top;
		std::vector<Task> tasks;
		emit(top, 0, 1, 0, &tasks);
		std::vector<Out> sub(tasks.size());
		MirtPool::get().run((uint32_t)tasks.size(), [&](uint32_t t) {
			sub[t].w.reserve((size_t)bb.nodes[tasks[t].node].count * 12);
			emit(sub[t], tasks[t].node, tasks[t].depth, tasks[t].acc, nullptr);
		});
		std::vector<size_t> base(tasks.size() + 1);
		base[0] = top.w.size() / 16;
		maxDepth = top.maxDepth;
		maxStack = top.maxStack;
		for (int k = 0; k < 8; ++k) diag[k] += top.diag[k];
		for (size_t t = 0; t < tasks.size(); ++t) {
			base[t + 1] = base[t] + sub[t].w.size() / 16;
			if (sub[t].maxDepth > maxDepth) maxDepth = sub[t].maxDepth;
			if (sub[t].maxStack > maxStack) maxStack = sub[t].maxStack;
			for (int k = 0; k < 8; ++k) diag[k] += sub[t].diag[k];
		}
		out.swap(top.w);
		out.resize(base[tasks.size()] * 16);
		for (size_t t = 0; t < tasks.size(); ++t) out[(size_t)tasks[t].parent * 16 + 10 + tasks[t].slot] = (uint32_t)base[t];
		MirtPool::get().run((uint32_t)tasks.size(), [&](uint32_t t) {
			uint32_t* dst = out.data() + base[t] * 16;
			memcpy(dst, sub[t].w.data(), sub[t].w.size() * 4);
			const uint32_t off = (uint32_t)base[t];
			for (size_t i = 0; i < sub[t].w.size() / 16; ++i)
				for (int k = 10; k < 14; ++k)
					if (!(dst[i * 16 + k] & BVH_LEAF)) dst[i * 16 + k] += off;
			std::vector<uint32_t>().swap(sub[t].w);
		});
		return 0;
	}
};

// the smallest i < n with bad(i), or n: the big arrays of a description are checked on all threads, and the message
// names the same element as a loop from the front would
template <class F>
uint64_t first_bad(uint64_t n, F bad) {
	const uint32_t grain = 1u << 16;
	const uint32_t chunks = (uint32_t)((n + grain - 1) / grain);
	std::atomic<uint64_t> first(n);
	MirtPool::get().run(chunks, [&](uint32_t c) {
		const uint64_t a = (uint64_t)c * grain, b = a + grain < n ? a + grain : n;
		if (a >= first.load(std::memory_order_relaxed)) return;
		for (uint64_t i = a; i < b; ++i) {
			if (!bad(i)) continue;
			uint64_t cur = first.load();
			while (i < cur && !first.compare_exchange_weak(cur, i)) {}
			return;
		}
	});
	return first.load();
}

int validate_desc(const mirt_scene_desc* d) {
	if (!d) return fail(MIRT_E_INVALID, "scene description is NULL");
	if (d->struct_size != sizeof(mirt_scene_desc)) return fail(MIRT_E_INVALID, "mirt_scene_desc.struct_size %u != %zu", d->struct_size, sizeof(mirt_scene_desc));
	if (d->n_nodes && !d->nodes) return fail(MIRT_E_INVALID, "nodes is NULL");
	if (d->n_leaf_prims && !d->leaf_prims) return fail(MIRT_E_INVALID, "leaf_prims is NULL");
	if (d->n_triangles && (!d->tri_verts || !d->tri_prim)) return fail(MIRT_E_INVALID, "triangle arrays are NULL");
	if (d->n_spheres && (!d->spheres || !d->sphere_prim)) return fail(MIRT_E_INVALID, "sphere arrays are NULL");
	if (d->n_prims && (!d->prim_material || !d->materials)) return fail(MIRT_E_INVALID, "material arrays are NULL");
	if (d->n_lights && !d->lights) return fail(MIRT_E_INVALID, "lights is NULL");
	if (d->max_depth > MIRT_MAX_DEPTH_LIMIT) return fail(MIRT_E_INVALID, "max_depth %d exceeds the supported %d", d->max_depth, MIRT_MAX_DEPTH_LIMIT);
	if (d->n_nodes >= 0x7fffffffu || d->n_leaf_prims >= 0x70000000u) return fail(MIRT_E_INVALID, "tree too large for 31-bit references");
	const uint64_t badNode = first_bad(d->n_nodes, [&](uint64_t i) {
		const mirt_node& n = d->nodes[i];
		if (n.b & MIRT_LEAF_FLAG) return (uint64_t)n.a + (n.b & ~MIRT_LEAF_FLAG) > d->n_leaf_prims;
		if (n.a != MIRT_NO_CHILD && (n.a <= i || n.a >= d->n_nodes)) return true;
		if (n.b != MIRT_NO_CHILD && (n.b <= i || n.b >= d->n_nodes)) return true;
		return n.a == MIRT_NO_CHILD && n.b == MIRT_NO_CHILD;
	});
	for (uint32_t i = (uint32_t)badNode; i < d->n_nodes; ++i) {
		const mirt_node& n = d->nodes[i];
		if (n.b & MIRT_LEAF_FLAG) {
			const uint64_t cnt = n.b & ~MIRT_LEAF_FLAG;
			if ((uint64_t)n.a + cnt > d->n_leaf_prims) return fail(MIRT_E_INVALID, "leaf %u: range [%u,+%llu) outside leaf_prims", i, n.a, (unsigned long long)cnt);
		}
		else {
			// pre-order: children come after their parent, which also rules out cycles
			if (n.a != MIRT_NO_CHILD && (n.a <= i || n.a >= d->n_nodes)) return fail(MIRT_E_INVALID, "node %u: bad left child %u", i, n.a);
			if (n.b != MIRT_NO_CHILD && (n.b <= i || n.b >= d->n_nodes)) return fail(MIRT_E_INVALID, "node %u: bad right child %u", i, n.b);
			if (n.a == MIRT_NO_CHILD && n.b == MIRT_NO_CHILD) return fail(MIRT_E_INVALID, "node %u: interior node without children", i);
		}
	}
	for (uint32_t i = (uint32_t)first_bad(d->n_leaf_prims, [&](uint64_t k) { return d->leaf_prims[k] >= d->n_prims; }); i < d->n_leaf_prims; ++i)
		if (d->leaf_prims[i] >= d->n_prims) return fail(MIRT_E_INVALID, "leaf_prims[%u] = %u outside the %u scene objects", i, d->leaf_prims[i], d->n_prims);
	for (uint32_t i = (uint32_t)first_bad(d->n_triangles, [&](uint64_t k) { return d->tri_prim[k] >= d->n_prims; }); i < d->n_triangles; ++i)
		if (d->tri_prim[i] >= d->n_prims) return fail(MIRT_E_INVALID, "tri_prim[%u] out of range", i);
	for (uint32_t i = 0; i < d->n_spheres; ++i)
		if (d->sphere_prim[i] >= d->n_prims) return fail(MIRT_E_INVALID, "sphere_prim[%u] out of range", i);
	// Non-finite geometry: with a non-invertible transform glm::inverse is inf/NaN and Sphere::intersect (Sphere.cpp:59-83)
	// returns a NaN distance, which Intersection::isValidIntersection (SceneObjects.hpp:35-37) takes for a hit; what the
	// reference then draws depends on the order of NaN comparisons all the way up its tree.  Rejected here instead.
	for (uint32_t i = 0; i < d->n_spheres; ++i) {
		const float* f = &d->spheres[i].center[0];
		for (int k = 0; k < 36; ++k)
			if (!isfinite(f[k])) return fail(MIRT_E_INVALID, "sphere %u: non-finite centre, radius, transform or inverse transform (singular transform?)", i);
	}
	// Coordinates beyond 1e18: the hierarchy's quantised child boxes (scales up to 2^73 per step of a byte) could no longer
	// enclose a node, and squared lengths overflow a float anyway.
	for (uint64_t i = first_bad((uint64_t)d->n_triangles * 9, [&](uint64_t k) { return !(fabsf(d->tri_verts[k]) <= 1e18f); }); i < (uint64_t)d->n_triangles * 9; ++i)
		if (!(fabsf(d->tri_verts[i]) <= 1e18f)) return fail(MIRT_E_INVALID, "triangle %llu: vertex coordinate not finite or beyond 1e18", (unsigned long long)(i / 9));
	for (uint32_t i = 0; i < d->n_spheres; ++i) {
		const mirt_sphere& sp = d->spheres[i];
		for (int k = 0; k < 3; ++k) {
			const float ext = fabsf(sp.radius) * (fabsf(sp.M[0 * 4 + k]) + fabsf(sp.M[1 * 4 + k]) + fabsf(sp.M[2 * 4 + k]));
			const float c = fabsf(sp.M[0 * 4 + k] * sp.center[0]) + fabsf(sp.M[1 * 4 + k] * sp.center[1]) + fabsf(sp.M[2 * 4 + k] * sp.center[2]) + fabsf(sp.M[3 * 4 + k]);
			if (!(ext + c <= 1e18f)) return fail(MIRT_E_INVALID, "sphere %u: world-space extent beyond 1e18", i);
		}
	}
	for (int k = 0; k < 3; ++k)
		if (!(fabsf(d->camera.look_from[k]) <= 1e18f)) return fail(MIRT_E_INVALID, "camera position not finite or beyond 1e18");
	for (uint32_t i = (uint32_t)first_bad(d->n_prims, [&](uint64_t k) { return d->prim_material[k] >= d->n_materials; }); i < d->n_prims; ++i)
		if (d->prim_material[i] >= d->n_materials) return fail(MIRT_E_INVALID, "prim_material[%u] out of range", i);
	return MIRT_OK;
}

int upload(DevBuf& b, const void* src, size_t bytes) {
	int e = b.ensure(bytes ? bytes : 16);
	if (e) return e;
	if (bytes) HIPCHK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
	return MIRT_OK;
}

int refwalk_ok(const mirt_scene* s, uint32_t mode);

int check_params(const mirt_scene* s, const mirt_render_params* p, uint32_t& tileW, uint32_t& tileH) {
	if (!s) return fail(MIRT_E_INVALID, "scene handle is NULL");
	if (!p) return fail(MIRT_E_INVALID, "render params are NULL");
	if (p->struct_size != sizeof(mirt_render_params)) return fail(MIRT_E_INVALID, "mirt_render_params.struct_size %u != %zu", p->struct_size, sizeof(mirt_render_params));
	if (p->width == 0 || p->height == 0 || p->width > 65536 || p->height > 65536) return fail(MIRT_E_INVALID, "image size %ux%u not supported", p->width, p->height);
	if (p->debug_mode > MIRT_DEBUG_NONE) return fail(MIRT_E_INVALID, "debug_mode %u unknown", p->debug_mode);
	if (p->trace_mode > MIRT_TRACE_REFWALK) return fail(MIRT_E_INVALID, "trace_mode %u unknown", p->trace_mode);
	if (int e = refwalk_ok(s, p->trace_mode)) return e;
	tileW = p->tile_w ? p->tile_w : 32;
	tileH = p->tile_h ? p->tile_h : 32;
	if ((tileW & 7) || (tileH & 7) || tileW > 1024 || tileH > 1024) return fail(MIRT_E_INVALID, "tile size %ux%u must be multiples of 8, at most 1024", tileW, tileH);
	if (p->tile_stride == 0 || p->first_tile >= p->tile_stride) return fail(MIRT_E_INVALID, "first_tile %u / tile_stride %u invalid", p->first_tile, p->tile_stride);
	return MIRT_OK;
}

struct TileGeom {
	uint32_t tileW, tileH, tilesX, tilesY, nLocal;
	uint64_t nSlots, livePixels;
};

TileGeom tile_geom(const mirt_render_params* p, uint32_t tileW, uint32_t tileH) {
	TileGeom g;
	g.tileW = tileW; g.tileH = tileH;
	g.tilesX = (p->width + tileW - 1) / tileW;
	g.tilesY = (p->height + tileH - 1) / tileH;
	const uint64_t total = (uint64_t)g.tilesX * g.tilesY;
	g.nLocal = total > p->first_tile ? (uint32_t)((total - p->first_tile + p->tile_stride - 1) / p->tile_stride) : 0;
	g.nSlots = (uint64_t)g.nLocal * tileW * tileH;
	g.livePixels = 0;
	for (uint32_t lt = 0; lt < g.nLocal; ++lt) {
		const uint64_t k = p->first_tile + (uint64_t)lt * p->tile_stride;
		const uint32_t tx = (uint32_t)(k % g.tilesX), ty = (uint32_t)(k / g.tilesX);
		const uint32_t w = (tx + 1) * tileW <= p->width ? tileW : p->width - tx * tileW;
		const uint32_t h = (ty + 1) * tileH <= p->height ? tileH : p->height - ty * tileH;
		g.livePixels += (uint64_t)w * h;
	}
	return g;
}

bool packed_layout(const mirt_render_params* p) { return !(p->tile_stride == 1 && p->tile_w == 0 && p->tile_h == 0); }

#define MIRT_MAX_STREAMS 64
FrameCtx* ctx_for(mirt_scene* s, hipStream_t st, bool create) {
	for (FrameCtx* c : s->ctxs) if (c->stream == st) return c;
	if (!create) return nullptr;
	if (s->ctxs.size() >= MIRT_MAX_STREAMS) { fail(MIRT_E_INVALID, "more than %d streams used with one scene", MIRT_MAX_STREAMS); return nullptr; }
	FrameCtx* c = new (std::nothrow) FrameCtx();
	if (!c) { fail(MIRT_E_NOMEM, "out of host memory"); return nullptr; }
	c->stream = st;
	if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) { fail(MIRT_E_DEVICE, "hipEventCreate failed"); delete c; return nullptr; }
	s->ctxs.push_back(c);
	return c;
}

// set-up, once per device: the device address of `scratch` on the device that is current
int scratch_init(int device) {
	if (device < 0 || device >= MIRT_MAX_DEVICES) return MIRT_OK; // such a device simply has no allocation-free entry point
	if (g_scratchPtr[device]) return MIRT_OK;
	void* ptr = nullptr;
	HIPCHK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
	g_scratchPtr[device] = (unsigned char*)ptr;
	return MIRT_OK;
}

// fn(i) for every i < n on the thread pool; the iterations must not depend on each other
template <class F>
void parallel_for(uint32_t n, F fn) {
	MirtPool::get().for_each(n, 1u << 13, [&](uint64_t i) { fn((uint32_t)i); });
}

// A host array that is not initialised when it is allocated: the large ones (hundreds of megabytes) are written once, by
// several threads, and zero-filling them first on one thread costs as much as computing them.
template <class T>
struct HostBuf {
	T* p = nullptr;
	size_t n = 0;
	HostBuf() {}
	HostBuf(const HostBuf&) = delete;
	HostBuf& operator=(const HostBuf&) = delete;
	~HostBuf() { free(p); }
	bool alloc(size_t count) {
		free(p);
		n = count;
		p = (T*)malloc(count * sizeof(T) + 16);
		return p != nullptr;
	}
};

// Host-to-device copies on a thread of their own, in the order they were asked for.  A source array must stay as it is
// until finish() has returned.
struct Uploader {
	struct Item { DevBuf* b; const void* src; size_t bytes; };
	int device;
	std::thread t;
	std::mutex m;
	std::condition_variable cv;
	std::vector<Item> q;
	size_t head = 0;
	bool closing = false, joined = false;
	hipError_t err = hipSuccess;
	explicit Uploader(int dev) : device(dev) { t = std::thread([this]() { loop(); }); }
	~Uploader() { (void)finish(); }
	void push(DevBuf& b, const void* src, size_t bytes) {
		{
			std::lock_guard<std::mutex> g(m);
			q.push_back(Item{&b, src, bytes});
		}
		cv.notify_one();
	}
	int finish() {
		if (!joined) {
			{
				std::lock_guard<std::mutex> g(m);
				closing = true;
			}
			cv.notify_one();
			t.join();
			joined = true;
		}
		if (err != hipSuccess) return fail(MIRT_E_DEVICE, "scene upload: %s", hipGetErrorString(err));
		return MIRT_OK;
	}
	void loop() {
		hipError_t e = hipSetDevice(device);
		for (;;) {
			Item it;
			{
				std::unique_lock<std::mutex> g(m);
				cv.wait(g, [&]() { return closing || head < q.size(); });
				if (head >= q.size()) break;
				it = q[head++];
			}
			if (e != hipSuccess) continue;
			const size_t need = it.bytes ? it.bytes : 16;
			if (need > it.b->bytes) {
				if (it.b->p) { (void)hipFree(it.b->p); it.b->p = nullptr; it.b->bytes = 0; }
				e = hipMalloc(&it.b->p, need);
				if (e == hipSuccess) it.b->bytes = need;
			}
			if (e == hipSuccess && it.bytes) e = hipMemcpy(it.b->p, it.src, it.bytes, hipMemcpyHostToDevice);
		}
		err = e;
	}
};

// a tree whose boxes are not nested is walked as it is, whatever mode was asked for
int effective_mode(const mirt_scene* s, uint32_t mode) { return s->nested ? (int)mode : MODE_REFWALK; }
// MODE_REFWALK keeps one stack entry per level of the packed reference tree in private memory (STACK_MAX of them)
int refwalk_ok(const mirt_scene* s, uint32_t mode) {
	if (effective_mode(s, mode) == MODE_REFWALK && s->wideDepth + 2 > STACK_MAX)
		return fail(MIRT_E_INVALID, "MIRT_TRACE_REFWALK: Partition tree depth %u exceeds the traversal stack (%d); use MIRT_TRACE_EXACT", s->wideDepth, STACK_MAX);
	return MIRT_OK;
}

// event after launch i (event 0 precedes the first launch); only when profiling is on
void mark(mirt_scene* s, FrameCtx* c, hipStream_t st, int kind) {
	if (!c || !s->profiling) return;
	const size_t need = (size_t)c->nLaunches + 2;
	while (c->launchEv.size() < need) {
		hipEvent_t e = nullptr;
		if (hipEventCreate(&e) != hipSuccess) { s->profiling = false; return; }
		c->launchEv.push_back(e);
	}
	if (kind < 0) { (void)hipEventRecord(c->launchEv[0], st); return; }
	c->launchKind.resize(c->nLaunches + 1);
	c->launchKind[c->nLaunches] = kind;
	(void)hipEventRecord(c->launchEv[c->nLaunches + 1], st);
	c->nLaunches++;
}

// One frame.  T(d) traces every ray of depth d in one launch: the reflection rays of that depth together with the shadow
// queries of the hits of depth d-1; shade(d) evaluates the lights for the hits of depth d once those queries are answered:
//     T(0) | T(1) shade(0) | T(2) shade(1) | ... | T(last+1: shadow queries only) shade(last) | resolve
template <int MODE, bool OVF>
void launch_frame(mirt_scene* s, FrameCtx* c, const DFrame& F, uint8_t* out, hipStream_t st) {
	const int grid = s->numCUs * GRID_BLOCKS_PER_CU;
	const int sgrid = s->numCUs * 8;
	int launch = 0;
	if (c) c->nLaunches = 0;
	mark(s, c, st, -1);
#ifdef MIRT_ALWAYS_REFILL
	const bool refill = MODE != MODE_REFWALK;
#else
	const bool refill = MODE != MODE_REFWALK && F.refillMin < 64u;
#endif
	if (refill) hipLaunchKernelGGL((k_trace<MODE, OVF, true, true>), dim3(grid), dim3(BLOCK), 0, st, s->ds, F, out, 0, launch++, 0);
	else hipLaunchKernelGGL((k_trace<MODE, OVF, true, false>), dim3(grid), dim3(BLOCK), 0, st, s->ds, F, out, 0, launch++, 0);
	mark(s, c, st, MIRT_K_PRIMARY);
	if (F.debug != D_PRIMARY_MAP) {
		const int last = (F.features & F_REFLECTIONS) ? s->maxDepth : 0;
		const bool shadowRays = (F.features & F_SHADOWS) && s->nLights > 0;
		for (int depth = 0; depth <= last; ++depth) {
			const int spawn = depth < last ? 1 : 0; // reflection rays exist while depth+1 <= maxDepth (main.cpp:177)
			if (spawn || shadowRays) {
				if (refill) hipLaunchKernelGGL((k_trace<MODE, OVF, false, true>), dim3(grid), dim3(BLOCK), 0, st, s->ds, F, out, depth + 1, launch++, spawn);
				else hipLaunchKernelGGL((k_trace<MODE, OVF, false, false>), dim3(grid), dim3(BLOCK), 0, st, s->ds, F, out, depth + 1, launch++, spawn);
				mark(s, c, st, MIRT_K_TRACE);
			}
			hipLaunchKernelGGL(k_shade, dim3(sgrid), dim3(BLOCK), 0, st, s->ds, F, depth);
			mark(s, c, st, MIRT_K_SHADE);
		}
		hipLaunchKernelGGL(k_resolve, dim3(sgrid), dim3(BLOCK), 0, st, F, out);
		mark(s, c, st, MIRT_K_RESOLVE);
	}
}

FastDiv make_div(uint32_t d) {
	FastDiv f = {0u, 0xffffffffu, d ? d : 1u, 0u};
	if (d <= 1) return f;
	uint32_t lg = 31;
	while (!(d >> lg)) --lg; // floor(log2 d)
	if ((d & (d - 1)) == 0) { f.magic = 0; f.shift = lg - 1; return f; }
	const uint64_t num = 1ull << (32 + lg);
	uint64_t m = num / d;
	const uint64_t rem = num % d;
	m += m;
	if (2 * rem >= d) m += 1;
	f.magic = (uint32_t)(m + 1);
	f.shift = lg;
	return f;
}

// Device memory of one frame beyond its output: the arrays of DFrame, in this order.
enum { WS_HITO, WS_HITD, WS_HITN, WS_SHRES, WS_LITL, WS_LITS, WS_PATHDEPTH, WS_ACTIVE, WS_COUNTERS, WS_OVF, WS_N };
struct FrameLayout {
	size_t bytes[WS_N];
};
// Does this frame run as ONE launch (k_frame)?  Not the walk of the reference's own tree (fallback mode, one wave per
// SIMD), not scenes big enough for lane refilling to pay (k_trace<REFILL> has no counterpart there), not frames whose slot
// numbers do not fit beside the depth in one word.
bool single_launch(const mirt_scene* s, const mirt_render_params* p, const TileGeom& g) {
	return s->singleLaunch && effective_mode(s, p->trace_mode) != MODE_REFWALK && s->refillMin >= 64u && s->maxDepth <= 31;
}

// hit records of a k_frame launch: FRAME_RECORDS for every wave of the largest grid the library launches
size_t frame_records(const mirt_scene* s) {
	return (size_t)s->numCUs * GRID_BLOCKS_PER_CU * (BLOCK / FRAME_BLOCK) * (FRAME_BLOCK / WAVE) * FRAME_RECORDS;
}

// Does a single-launch frame keep its per-path data in hit records (k_frame<RECORDS>) instead of in arrays as long as the
// image?  Where memory counts more than the last 3 % of speed: the entry point that allocates nothing (its scratch array then
// takes frames of any size), callers who ask for it (MIRT_FRAME_RECORDS=1), and frames whose slot numbers do not fit beside
// the depth in one word (2^27 slots and more).
bool frame_uses_records(const TileGeom& g, bool noalloc) {
	const char* e = getenv("MIRT_FRAME_RECORDS"); // (read per frame: tests switch it)
	if (e && *e) return *e != '0';
	return noalloc || g.nSlots >= (1ull << FRAME_SLOT_BITS);
}

FrameLayout frame_layout(const mirt_scene* s, size_t n, bool single, bool records) {
	const size_t depths = (size_t)s->maxDepth + 1, nl = s->nLights;
	FrameLayout L;
	memset(&L, 0, sizeof L);
	L.bytes[WS_COUNTERS] = 2 * CTR_COUNT * 4; // two blocks (FrameCtx::ctrPhase)
	L.bytes[WS_OVF] = s->deepBvh ? (size_t)s->ovfLevels * s->ovfStride * 4 : 0;
	if (single) {
		// k_frame: one 48-byte hit record and one light sum (with the material index) per depth, for every pixel slot -- or,
		// with records, for each of the FRAME_RECORDS records of every wave of the largest launch, whatever the image's size
		const size_t nRec = records ? frame_records(s) : n;
		L.bytes[WS_HITO] = L.bytes[WS_HITD] = L.bytes[WS_HITN] = nRec * 16;
		L.bytes[WS_LITL] = depths * nRec * 16;
		return L;
	}
	L.bytes[WS_HITO] = L.bytes[WS_HITD] = L.bytes[WS_HITN] = 2 * n * 16;
	L.bytes[WS_SHRES] = (nl ? n * nl : 1) * 8;
	L.bytes[WS_LITL] = L.bytes[WS_LITS] = depths * n * 16;
	L.bytes[WS_PATHDEPTH] = n;
	L.bytes[WS_ACTIVE] = 2 * n * 4;
	return L;
}

// Checks shared by every way of rendering a frame.  *empty: nothing to launch (no slots).
int frame_checks(mirt_scene* s, const mirt_render_params* p, void* d_out, size_t d_out_bytes, bool packed, uint32_t& tileW, uint32_t& tileH, TileGeom& g, size_t& need) {
	int e = check_params(s, p, tileW, tileH);
	if (e) return e;
	if (!d_out) return fail(MIRT_E_INVALID, "output buffer is NULL");
	g = tile_geom(p, tileW, tileH);
	need = packed ? (size_t)g.nSlots * 3 : (size_t)p->width * p->height * 3;
	if (d_out_bytes < need) return fail(MIRT_E_INVALID, "output buffer holds %zu bytes, %zu needed", d_out_bytes, need);
	if (g.nSlots >= 0xffffffc0ull) return fail(MIRT_E_INVALID, "too many pixel slots for one launch");
	if ((size_t)g.nSlots * ((size_t)s->nLights + 1) >= 0xffffffffull) return fail(MIRT_E_INVALID, "too many rays per depth for one launch (%llu slots x %u lights)", (unsigned long long)g.nSlots, s->nLights);
	return MIRT_OK;
}

// Boxes that together hold every primitive: a breadth-first cut through the 4-wide hierarchy (`wide`: 16 words per node,
// layout in mirt_device.h), opened level by level until there are about `want` of them.  A child's box is decoded from
// its parent's quantised bytes, i.e. it is the box the walk itself tests (it contains the padded primitive bounds).
void cut_boxes(const uint32_t* wide, size_t nNodes, uint32_t root, const float rootLo[3], const float rootHi[3], size_t want, std::vector<float>& out) {
	struct Item { uint32_t ref; float b[6]; };
	std::vector<Item> level, next;
	out.clear();
	Item r;
	r.ref = root;
	for (int k = 0; k < 3; ++k) { r.b[k] = rootLo[k]; r.b[3 + k] = rootHi[k]; }
	if (root == BVH_DONE) return; // an empty scene: no box, nothing is visible
	level.push_back(r);
	for (;;) {
		size_t interior = 0;
		for (const Item& it : level) interior += !(it.ref & BVH_LEAF) && it.ref < nNodes;
		if (interior == 0 || level.size() >= want) break;
		next.clear();
		for (const Item& it : level) {
			if ((it.ref & BVH_LEAF) || it.ref >= nNodes) { next.push_back(it); continue; }
			const uint32_t* w = wide + (size_t)it.ref * 16;
			float org[3], scale[3];
			for (int k = 0; k < 3; ++k) {
				memcpy(&org[k], &w[k], 4);
				scale[k] = ldexpf(1.0f, (int)(int8_t)((w[3] >> (8 * k)) & 0xffu));
			}
			for (int c = 0; c < 4; ++c) {
				const uint32_t ref = w[10 + c];
				if (ref == BVH_DONE) continue;
				Item ch;
				ch.ref = ref;
				for (int k = 0; k < 3; ++k) {
					ch.b[k] = org[k] + (float)((w[4 + k] >> (8 * c)) & 0xffu) * scale[k];
					ch.b[3 + k] = org[k] + (float)((w[7 + k] >> (8 * c)) & 0xffu) * scale[k];
				}
				next.push_back(ch);
			}
		}
		level.swap(next);
	}
	out.reserve(level.size() * 6);
	for (const Item& it : level) out.insert(out.end(), it.b, it.b + 6);
}

// boxes of the cut (MIRT_CUT_BOXES sets another number)
size_t cut_box_count() {
	const char* e = getenv("MIRT_CUT_BOXES");
	const long v = e ? atol(e) : 0;
	return v >= 1 && v <= (1l << 24) ? (size_t)v : 4096;
}

// Which 8x8 pixel blocks of a frame can see the scene?  Every box of the scene's cut is projected the way
// Camera::createRayToPixel (Camera.cpp:15-24) maps pixels to directions -- its eight corners, in double -- and the blocks
// under the corners' bounding rectangle, widened by 2.5 pixels (orders of magnitude more than the float rounding of a
// ray's direction), are marked.  A camera ray through an unmarked block misses every box, so it misses every primitive:
// the block's pixels are black, and the frame's clear has already made them so.  A box with a corner at or behind the
// eye plane has no such rectangle: then every block is visited.  Lists are kept per frame geometry (a scene's camera is
// fixed); a block's number is its slot number / 64.  Returns NULL when every block has to be visited.
// create = false: only a list that exists already is used (the entry point that allocates nothing).
const BlockList* visible_blocks(mirt_scene* s, const mirt_render_params* p, const TileGeom& g, uint32_t tileW, uint32_t tileH, bool packed, bool create) {
	for (BlockList* b : s->blockLists)
		if (b->width == p->width && b->height == p->height && b->tileW == tileW && b->tileH == tileH && b->first == p->first_tile && b->stride == p->tile_stride && b->packed == (packed ? 1u : 0u))
			return b->all ? nullptr : b;
	if (!create) return nullptr;
	if (s->blockLists.size() >= 32) { // (callers that render ever new sizes: start over)
		(void)hipDeviceSynchronize();
		for (BlockList* b : s->blockLists) { b->list.release(); delete b; }
		s->blockLists.clear();
	}
	BlockList* bl = new (std::nothrow) BlockList();
	if (!bl) return nullptr;
	bl->width = p->width; bl->height = p->height; bl->tileW = tileW; bl->tileH = tileH; bl->first = p->first_tile; bl->stride = p->tile_stride; bl->packed = packed ? 1u : 0u;
	s->blockLists.push_back(bl);
	bl->all = true;
	const DScene& ds = s->ds;
	const uint64_t nBlocks = g.nSlots / 64;
	if (nBlocks == 0 || nBlocks > (1ull << 31)) return nullptr;
	std::vector<uint8_t> mark(nBlocks, 0);
	const double W = p->width, H = p->height, T = ds.cam.tanHalf, aspect = (double)((float)(int)p->width / (int)p->height);
	if (!(T > 1e-6)) return nullptr;
	const uint32_t sbx = tileW / 8, perTile = sbx * (tileH / 8);
	for (size_t i = 0; i + 6 <= s->cutBoxes.size(); i += 6) {
		const float* b = &s->cutBoxes[i];
		double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
		for (int corner = 0; corner < 8; ++corner) {
			double v[3];
			for (int k = 0; k < 3; ++k) v[k] = (double)((corner >> k) & 1 ? b[3 + k] : b[k]) - ds.cam.eye[k];
			const double a = v[0] * ds.cam.u[0] + v[1] * ds.cam.u[1] + v[2] * ds.cam.u[2];
			const double bb = v[0] * ds.cam.v[0] + v[1] * ds.cam.v[1] + v[2] * ds.cam.v[2];
			const double c = -(v[0] * ds.cam.w[0] + v[1] * ds.cam.w[1] + v[2] * ds.cam.w[2]); // along the viewing direction
			const double len = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
			if (!(c > 1e-3 * len) || !std::isfinite(c)) return nullptr; // at or behind the eye plane
			// dir = alpha u + beta v - w (Camera.cpp:21): the corner lies on the ray with alpha = a / c, beta = b / c
			const double px = (a / c) / (T * aspect) * (W / 2) + W / 2, py = H / 2 - (bb / c) / T * (H / 2);
			if (!std::isfinite(px) || !std::isfinite(py)) return nullptr;
			x0 = std::min(x0, px); x1 = std::max(x1, px); y0 = std::min(y0, py); y1 = std::max(y1, py);
		}
		x0 -= 2.5; y0 -= 2.5; x1 += 2.5; y1 += 2.5; // pixel centres lie at +0.5
		if (x1 < 0 || y1 < 0 || x0 >= W || y0 >= H) continue;
		const uint32_t bx0 = (uint32_t)(std::max(0.0, x0) / 8), bx1 = (uint32_t)(std::min(W - 1, x1) / 8);
		const uint32_t by0 = (uint32_t)(std::max(0.0, y0) / 8), by1 = (uint32_t)(std::min(H - 1, y1) / 8);
		for (uint32_t by = by0; by <= by1; ++by) {
			const uint32_t ty = by * 8 / tileH, sy = (by * 8 % tileH) / 8;
			for (uint32_t bx = bx0; bx <= bx1; ++bx) {
				const uint32_t tx = bx * 8 / tileW, sx = (bx * 8 % tileW) / 8;
				const uint64_t k = (uint64_t)ty * g.tilesX + tx;
				if (k < p->first_tile || (k - p->first_tile) % p->tile_stride) continue; // another rank's tile
				const uint64_t blk = (k - p->first_tile) / p->tile_stride * perTile + (uint64_t)sy * sbx + sx;
				if (blk < nBlocks) mark[blk] = 1;
			}
		}
	}
	std::vector<uint32_t> list;
	for (uint64_t i = 0; i < nBlocks; ++i) if (mark[i]) list.push_back((uint32_t)i);
	if (list.size() == nBlocks) return nullptr; // everything is visible: no list needed
	if (list.empty()) list.push_back(0);         // nothing is: one block keeps the launch simple
	const size_t count = list.size();
	// A second order for a frame that has the GPU to itself (see enqueue_frame): the blocks deepest inside the visible region
	// first -- where every camera ray hits and the paths are long -- and the region's rim, whose batches are done in a round or
	// two, last.  Depth = rounds of peeling the region's edge blocks away (whole-image layout only).
	if (p->tile_stride == 1 && !packed && count >= 1024) {
		const uint32_t bw = (p->width + 7) / 8, bh = (p->height + 7) / 8;
		auto blockOf = [&](uint32_t bx, uint32_t by) -> uint64_t { // the slot order: tiles row-major, 8x8 blocks row-major inside a tile
			const uint32_t tx = bx * 8 / tileW, sx = (bx * 8 % tileW) / 8, ty = by * 8 / tileH, sy = (by * 8 % tileH) / 8;
			return ((uint64_t)ty * g.tilesX + tx) * perTile + (uint64_t)sy * sbx + sx;
		};
		auto vis = [&](int bx, int by) { if (bx < 0 || by < 0 || bx >= (int)bw || by >= (int)bh) return false; const uint64_t b = blockOf(bx, by); return b < nBlocks && mark[b] != 0; };
		std::vector<uint16_t> depth(nBlocks, 0);
		std::vector<uint32_t> cur, next;
		for (uint32_t by = 0; by < bh; ++by) for (uint32_t bx = 0; bx < bw; ++bx)
			if (vis(bx, by) && !(vis(bx - 1, by) && vis(bx + 1, by) && vis(bx, by - 1) && vis(bx, by + 1))) { depth[blockOf(bx, by)] = 1; cur.push_back(by * bw + bx); }
		for (uint16_t d = 1; !cur.empty() && d < 60000; ++d) {
			next.clear();
			for (uint32_t c : cur) {
				const int bx = c % bw, by = c / bw;
				const int nb[4][2] = {{bx - 1, by}, {bx + 1, by}, {bx, by - 1}, {bx, by + 1}};
				for (auto& q : nb) if (vis(q[0], q[1]) && depth[blockOf(q[0], q[1])] == 0) { depth[blockOf(q[0], q[1])] = d + 1; next.push_back(q[1] * bw + q[0]); }
			}
			cur.swap(next);
		}
		std::vector<uint32_t> deep(list);
		std::stable_sort(deep.begin(), deep.end(), [&](uint32_t a, uint32_t b) { return depth[a] > depth[b]; });
		list.insert(list.end(), deep.begin(), deep.end());
		bl->deep = true;
	}
	if (bl->list.ensure(list.size() * 4) != MIRT_OK || hipMemcpy(bl->list.p, list.data(), list.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
		(void)hipGetLastError();
		bl->deep = false;
		return nullptr;
	}
	bl->count = (uint32_t)count;
	bl->all = false;
	if (getenv("MIRT_VERBOSE")) fprintf(stderr, "mirt: %ux%u: %zu of %llu pixel blocks can see the scene (%zu boxes)\n", p->width, p->height, count, (unsigned long long)nBlocks, s->cutBoxes.size() / 6);
	return bl;
}

// Puts one frame on the stream: clears, every launch of the frame.  mem[] are the arrays of frame_layout(); c may be NULL
// (no per-launch profiling then).  Allocates nothing.
int enqueue_frame(mirt_scene* s, FrameCtx* c, const mirt_render_params* p, const TileGeom& g, uint32_t tileW, uint32_t tileH, bool packed, bool single, bool records,
                  void* const mem[WS_N], void* d_out, size_t need, hipStream_t st, int othersInFlight = 0, bool ctrReady = false) {
	const size_t n = (size_t)g.nSlots;
	DFrame F;
	F.ovf = (uint32_t*)mem[WS_OVF];
	F.ovfStride = s->ovfStride;
	F.hitO = (float4*)mem[WS_HITO]; F.hitD = (float4*)mem[WS_HITD]; F.hitN = (float4*)mem[WS_HITN];
	F.shRes = (uint2*)mem[WS_SHRES];
	F.litL = (float4*)mem[WS_LITL]; F.litS = (float4*)mem[WS_LITS];
	F.pathDepth = (uint8_t*)mem[WS_PATHDEPTH];
	F.active = (uint32_t*)mem[WS_ACTIVE];
	F.nRec = records ? (uint32_t)frame_records(s) : (uint32_t)g.nSlots;
	F.nSlots = (uint32_t)g.nSlots;
	F.width = p->width; F.height = p->height;
	F.tileW = tileW; F.tileH = tileH; F.tilesX = g.tilesX; F.nTilesLocal = g.nLocal;
	F.firstTile = p->first_tile; F.tileStride = p->tile_stride;
	F.features = p->feature_bits; F.debug = p->debug_mode;
	F.packedTiles = packed ? 1u : 0u;
	F.refillMin = s->refillMin;
	F.divArea = make_div(tileW * tileH);
	F.divSbx = make_div(tileW >> 3);
	F.divTilesX = make_div(g.tilesX);
	// bytes of tiles that are not rendered must stay untouched in the full-image layout, so only clear when all are
	F.preCleared = (packed || p->tile_stride == 1) ? 1u : 0u;
	F.blockList = nullptr;
	F.nListed = 0;
	F.pullBelow = 64;
	F.wideClear = (((uintptr_t)d_out & 3u) == 0 && (packed || (p->width & 3u) == 0)) ? 1u : 0u;
	// A camera ray that misses the scene's bounds hits nothing, and with the output cleared beforehand its pixel needs no
	// visit at all: k_frame is handed the list of the 8x8 pixel blocks that can see the scene (visible_blocks).
	uint32_t nBatchesFrame = (uint32_t)(((uint64_t)F.nSlots + 63u) / 64u);
	// A frame that has the GPU to itself is as long as its busiest wave: the batches with the long paths go out first, and a wave
	// takes more pixel slots only when it holds few hits (with frames of other streams beside it, full rounds count for more).
	// Stanford dragon 3840x2160 alone: 1.56-1.59 ms -> 1.47 with the threshold at 16-24 hits (32: 1.51, 8: 1.51, 1: 1.49) -> 1.44-1.45
	// with the order as well (the order without the threshold: 1.57); 7680x4320 3.59 -> 3.40, 1200 spheres 0.665 -> 0.62, Cornell box
	// 0.515 -> 0.49.  The waves that end last were those that kept taking slots while they held hits with long paths ahead
	// (scripts/frame_timeline.py: 28 rounds against a mean of 16).  MIRT_ALONE=0 switches both off, MIRT_PULL_BELOW=n sets the threshold.
	static const int aloneEnv = []() { const char* e = getenv("MIRT_ALONE"); return e && *e ? atoi(e) : 1; }();
	const bool alone = single && othersInFlight == 0 && aloneEnv != 0;
	bool deepOrder = false;
	const char* cikEarly = getenv("MIRT_CLEAR_IN_KERNEL");
	if (single && F.preCleared && !(cikEarly && *cikEarly == '1')) {
		const char* re = getenv("MIRT_BLOCK_LIST"); // (read per frame: tests switch it)
		if (!(re && *re == '0')) {
			const BlockList* bl = visible_blocks(s, p, g, tileW, tileH, packed, c != nullptr);
			if (bl) {
				F.blockList = (const uint32_t*)bl->list.p;
				F.nListed = bl->count;
				nBatchesFrame = bl->count;
				if (alone && bl->deep) { F.blockList += bl->count; deepOrder = true; }
			}
		}
	}
	// The order of k_frame's batches.  In image order the blocks that see the scene come in long runs (the rows the object
	// covers) between runs that see nothing, and all waves are in the same kind of run at the same time.  Stepping through
	// the blocks with a stride of about 1/128 of the frame (made coprime to the number of blocks, so every block comes
	// exactly once) mixes the two at every moment: Stanford dragon 4K +2.5 % with frames in flight, 8K +8 % and 7 % less for a
	// frame alone; frames below 1080p do not care.  (Handing out the blocks that see the scene FIRST does the opposite.)
	// MIRT_BATCH_STRIDE=0: image order.
	F.batchMul = 0;
	if (single) {
		static const int parts = []() { const char* e = getenv("MIRT_BATCH_STRIDE"); const int v = e ? atoi(e) : 128; return v < 0 ? 0 : v; }(); // (a number: that fraction of the frame)
		const uint64_t nb = nBatchesFrame;
		if (alone) {
			static const int pull = []() { const char* e = getenv("MIRT_PULL_BELOW"); const int v = e ? atoi(e) : 20; return v >= 1 && v <= 64 ? v : 20; }();
			F.pullBelow = (uint32_t)pull;
		}
		if (parts && nb >= 16384 && !deepOrder) {
			uint64_t m = (nb / (uint64_t)parts) | 1u;
			auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; };
			while (gcd(m, nb) != 1) m += 2;
			F.batchMul = (uint32_t)m;
		}
	}
	// Clears: of the counters, of the output (the background is black, main.cpp:124,180) and of the path depths.
	// k_frame can do without the two queued in front of it (MIRT_CLEAR_IN_KERNEL=1): every pixel slot then writes its own
	// pixel, black ones included (blocks of 8x8 that see nothing as 48 words), and the counter block it uses was cleared by
	// the stream's previous frame (ctrReady; without a stream context -- the entry point that allocates nothing -- it is
	// cleared here).  With eight frames in flight a queued clear lasts 0.28-0.44 ms on average (5 us alone: it has to find
	// room beside the frames of other streams) and the two together are 13-16 % of the summed launch durations -- but that
	// time is not lost: measured on the Stanford dragon at 3840x2160, frames without queued clears run 14.00 against 14.10
	// Grays/s over 80 frames and 13.0 against 13.7 over 20 (a burst of frames that all start at the same moment also end
	// together; the clears in front of them spread their starts).  So the queued clears stay the default.
	// (Measured again with the visible-block lists, three interleaved runs: 13.0-13.5 against 14.6-14.9 Grays/s over 20 frames,
	// 14.1 against 15.4 over 80, a frame alone 1.56-1.64 against 1.43-1.45 ms -- a frame that clears for itself has to visit
	// every block of the image, the 57 % that see nothing included.)
	const char* cikEnv = getenv("MIRT_CLEAR_IN_KERNEL"); // (read per frame: tests switch it)
	const bool clearInKernel = cikEnv && *cikEnv == '1';
	uint32_t* const ctrBase = (uint32_t*)mem[WS_COUNTERS];
	const uint32_t phase = c ? c->ctrPhase : 0u;
	F.counters = ctrBase + (size_t)phase * CTR_COUNT;
	F.countersNext = nullptr;
	if (single && clearInKernel) {
		F.preCleared = 0u;
		if (ctrReady) F.countersNext = ctrBase + (size_t)(1u - phase) * CTR_COUNT;
		else HIPCHK(hipMemsetAsync(F.counters, 0, CTR_COUNT * 4, st));
	}
	else {
		HIPCHK(hipMemsetAsync(ctrBase, 0, 2 * CTR_COUNT * 4, st)); // this frame's block and the next frame's
		if (F.preCleared) {
			HIPCHK(hipMemsetAsync(d_out, 0, need, st));
			if (!single) HIPCHK(hipMemsetAsync(F.pathDepth, 0, n, st));
		}
	}
	if (c) {
		c->lastCtr = F.counters;
		c->ctrPhase = 1u - phase;
	}
	if (single) {
		// Waves of the launch per SIMD.  A frame alone wants every place on the GPU (7): its time is the chain of rounds
		// behind its last batches, and more waves make every chain shorter.  A frame that shares the GPU with frames on
		// other streams wants waves with enough batches each to gather their hits into full rounds -- about 32 batches of
		// 64 pixel slots per wave, never fewer than two waves per SIMD.  Measured with eight frames in flight on 16 hardware
		// queues, Grays/s over 20 frames by waves per SIMD -- Stanford dragon 3840x2160 (56 000 batches inside the screen
		// rectangle of the scene): 1: 13.4, 2: 14.4, 3: 14.0, 7: 13.4; 7680x4320 (233 000): 2: 17.5, 4-7: 17.8; 1920x1080
		// (15 000): 1: 9.4, 2: 9.85, 3: 9.9; 1200 spheres 1920x1080 (32 000): 1: 15.8, 2: 16.5, 3: 15.6; Cornell box 640x480
		// (4 800): 1: 13.3, 2: 14.5, 3: 14.1.  MIRT_FRAME_GRID=n fixes the number.
		static const int gwEnv = []() { const char* e = getenv("MIRT_FRAME_GRID"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= MIRT_FRAME_WAVES ? v : 0; }();
		int gw = MIRT_FRAME_WAVES;
		if (gwEnv) gw = gwEnv;
		else if (othersInFlight >= 3) {
			const uint64_t nBatches = nBatchesFrame, perWave = (uint64_t)s->numCUs * 4u * 32u;
			const uint64_t want = (nBatches + perWave / 2) / perWave;
			gw = want < 2 ? 2 : (want > (uint64_t)MIRT_FRAME_WAVES ? MIRT_FRAME_WAVES : (int)want);
		}
		const int grid = s->numCUs * (gw < GRID_BLOCKS_PER_CU ? gw : GRID_BLOCKS_PER_CU) * (BLOCK / FRAME_BLOCK); // every block resident at once
		if (c) c->nLaunches = 0;
		mark(s, c, st, -1);
#define MIRT_LAUNCH_FRAME(MODE, OVF, SHARE)                                                                                                                  \
	do {                                                                                                                                                \
		if (records) hipLaunchKernelGGL((k_frame<MODE, OVF, SHARE, true>), dim3(grid), dim3(FRAME_BLOCK), 0, st, s->ds, F, (uint8_t*)d_out);           \
		else hipLaunchKernelGGL((k_frame<MODE, OVF, SHARE, false>), dim3(grid), dim3(FRAME_BLOCK), 0, st, s->ds, F, (uint8_t*)d_out);                   \
	} while (0)
		if (effective_mode(s, p->trace_mode) == MODE_FAST) {
			if (s->deepBvh) MIRT_LAUNCH_FRAME(MODE_FAST, true, false);
			else MIRT_LAUNCH_FRAME(MODE_FAST, false, false);
		}
		else {
			// Lanes of a wave help each other through long walks (walk_wave_share) where walks are long enough for that to pay:
			// Stanford dragon 4K 14.3 against 13.4 Grays/s, 8K 17.1 against 16.5; the 200 ellipsoids and the Cornell box, whose
			// rays are done in a handful of steps, lose 4 % to the larger loop.  MIRT_SHARE=0/1 decides for every scene.
			const char* shareStr = getenv("MIRT_SHARE"); // (read per frame: tests switch it between frames)
			const int shareEnv = shareStr && *shareStr ? atoi(shareStr) : -1;
			const bool share = FRAME_BLOCK == WAVE && (shareEnv >= 0 ? shareEnv != 0 : s->bvh.bytes / 64 >= 4096);
			if (s->deepBvh) {
				if (share) MIRT_LAUNCH_FRAME(MODE_EXACT, true, true);
				else MIRT_LAUNCH_FRAME(MODE_EXACT, true, false);
			}
			else {
				if (share) MIRT_LAUNCH_FRAME(MODE_EXACT, false, true);
				else MIRT_LAUNCH_FRAME(MODE_EXACT, false, false);
			}
		}
#undef MIRT_LAUNCH_FRAME
		mark(s, c, st, MIRT_K_FRAME);
		HIPCHK(hipGetLastError());
		return MIRT_OK;
	}
	switch (effective_mode(s, p->trace_mode)) {
	case MODE_FAST:
		if (s->deepBvh) launch_frame<MODE_FAST, true>(s, c, F, (uint8_t*)d_out, st);
		else launch_frame<MODE_FAST, false>(s, c, F, (uint8_t*)d_out, st);
		break;
	case MODE_REFWALK: launch_frame<MODE_REFWALK, false>(s, c, F, (uint8_t*)d_out, st); break;
	default:
		if (s->deepBvh) launch_frame<MODE_EXACT, true>(s, c, F, (uint8_t*)d_out, st);
		else launch_frame<MODE_EXACT, false>(s, c, F, (uint8_t*)d_out, st);
		break;
	}
	HIPCHK(hipGetLastError());
	return MIRT_OK;
}

int render_async(mirt_scene* s, const mirt_render_params* p, void* d_out, size_t d_out_bytes, hipStream_t st, bool packed) {
	uint32_t tileW, tileH;
	TileGeom g;
	size_t need;
	int e = frame_checks(s, p, d_out, d_out_bytes, packed, tileW, tileH, g, need);
	if (e) return e;
	HIPCHK(hipSetDevice(s->device));
	FrameCtx* c = ctx_for(s, st, true);
	if (!c) return MIRT_E_INVALID;
	c->lastValid = false;
	if (g.nSlots == 0) { c->lastPrimary = 0; c->lastFeatures = p->feature_bits; return MIRT_OK; }
	if (s->maxDepth < 0) {
		// depth 0 already exceeds maxDepth: computePixelColor returns Color() for every pixel (main.cpp:177,199)
		HIPCHK(hipMemsetAsync(d_out, 0, need, st));
		c->lastPrimary = g.livePixels; // the camera ray of every pixel is still made (main.cpp:152)
		c->lastFeatures = p->feature_bits;
		return MIRT_OK;
	}
	Workspace& w = c->ws;
	const bool single = single_launch(s, p, g);
	const bool records = single && frame_uses_records(g, false);
	const FrameLayout L = frame_layout(s, (size_t)g.nSlots, single, records);
	DevBuf* const bufs[WS_N] = {&w.hitO, &w.hitD, &w.hitN, &w.shRes, &w.litL, &w.litS, &w.pathDepth, &w.active, &w.counters, &w.ovf};
	void* mem[WS_N];
	for (int k = 0; k < WS_N; ++k) {
		const void* before = bufs[k]->p;
		if (L.bytes[k] && (e = bufs[k]->ensure(L.bytes[k]))) return e;
		mem[k] = bufs[k]->p;
		if (k == WS_COUNTERS && bufs[k]->p != before) { // a new counters buffer: both blocks start as zeros
			HIPCHK(hipMemsetAsync(bufs[k]->p, 0, 2 * CTR_COUNT * 4, st));
			c->ctrPhase = 0;
		}
	}
	// frames of this scene still on the GPU on other streams (their closing events have not happened yet)
	int others = 0;
	for (FrameCtx* o : s->ctxs)
		if (o != c && o->lastValid && hipEventQuery(o->ev1) == hipErrorNotReady) ++others;
	(void)hipGetLastError(); // (hipErrorNotReady is an answer, not a failure)
	if (s->framesHint > 1 && others < (int)s->framesHint - 1) others = (int)s->framesHint - 1;
	HIPCHK(hipEventRecord(c->ev0, st));
	if ((e = enqueue_frame(s, c, p, g, tileW, tileH, packed, single, records, mem, d_out, need, st, others, true))) return e;
	HIPCHK(hipEventRecord(c->ev1, st));
	c->lastSingle = single;
	c->lastPrimary = g.livePixels;
	c->lastFeatures = p->feature_bits;
	c->lastValid = true;
	return MIRT_OK;
}

int read_stats(mirt_scene* s, hipStream_t st, mirt_stats* out) {
	memset(out, 0, sizeof *out);
	HIPCHK(hipStreamSynchronize(st));
	FrameCtx* c = ctx_for(s, st, false);
	if (!c) return MIRT_OK;
	out->primary_rays = c->lastPrimary;
	if (!c->lastValid) return MIRT_OK;
	uint32_t cnt[CTR_COUNT];
	HIPCHK(hipMemcpy(cnt, c->lastCtr, sizeof cnt, hipMemcpyDeviceToHost));
	uint64_t hits = 0, refl = 0;
	if (c->lastSingle) {
		hits = cnt[CTR_F_HITS_TOTAL];
		// a reflection ray for every hit that is not at the last depth (main.cpp:177,189-190)
		refl = (c->lastFeatures & MIRT_F_REFLECTIONS) ? hits - cnt[CTR_F_HITS_LAST] : 0;
		out->primary_hits = cnt[CTR_F_HITS_PRIMARY];
		out->shadow_rays = hits * s->nLights;
		out->shadow_rays_traced = cnt[CTR_F_SHADOW_TRACED];
		out->reflection_rays = refl;
		out->reflection_rays_traced = refl - cnt[CTR_F_REFL_SKIPPED];
		float ms1 = 0.0f;
		HIPCHK(hipEventElapsedTime(&ms1, c->ev0, c->ev1));
		out->device_ms = ms1;
		return MIRT_OK;
	}
	for (int d = 0; d <= s->maxDepth; ++d) hits += cnt[CTR_NHITS(d)];
	// a reflection ray is traced for every hit at depth d < maxDepth when REFLECTIONS is on (main.cpp:177,189-190)
	if (c->lastFeatures & MIRT_F_REFLECTIONS) for (int d = 0; d < s->maxDepth; ++d) refl += cnt[CTR_NHITS(d)];
	out->primary_hits = cnt[CTR_NHITS(0)];
	out->shadow_rays = hits * s->nLights; // main.cpp:208-223: one query per light per hit, whatever the feature bits
	uint64_t traced = 0;
	for (int d = 0; d <= s->maxDepth; ++d) traced += cnt[CTR_NSHADOW(d)];
	out->shadow_rays_traced = traced;
	out->reflection_rays = out->reflection_rays_traced = refl;
	float ms = 0.0f;
	HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
	out->device_ms = ms;
	return MIRT_OK;
}

}  // namespace

extern "C" {

const char* mirt_last_error(void) { return g_err.c_str(); }

int mirt_scene_create(const mirt_scene_desc* d, int device, mirt_scene** out) {
	if (!out) return fail(MIRT_E_INVALID, "out is NULL");
	*out = nullptr;
	// MIRT_VERBOSE=1: seconds per phase on stderr
	const bool verbose = getenv("MIRT_VERBOSE") != nullptr;
	auto tPhase = std::chrono::steady_clock::now();
	auto phase = [&](const char* name) {
		if (!verbose) return;
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "mirt_scene_create: %-28s %.3f s\n", name, std::chrono::duration<double>(now - tPhase).count());
		tPhase = now;
	};
	int e = validate_desc(d);
	if (e) return e;
	phase("checks of the description");
	// Host arrays that are uploaded live until the uploads have finished: they are declared before the uploader, which
	// copies on a thread of its own while the next arrays are being prepared.
	std::vector<int64_t> primSlot(d->n_prims);
	std::vector<uint32_t> mCnt(d->n_prims, 0), mOfs((size_t)d->n_prims + 1, 0);
	HostBuf<float> memb, tris, triX;
	HostBuf<uint32_t> triInfo, refStream;
	std::vector<float> sph, mats, lights, refWide;
	std::vector<uint32_t> sphInfo, bvhWide;
	mirt_scene* s = new (std::nothrow) mirt_scene();
	if (!s) return fail(MIRT_E_NOMEM, "out of host memory");
	s->device = device;
	memset(&s->ds, 0, sizeof s->ds);
	DScene& ds = s->ds;
	Uploader up(device); // (its thread is also the first to touch the device: the runtime starts up beside the work below)
	auto bail = [&](int err) {
		const std::string keep = g_err;
		(void)up.finish();
		mirt_scene_destroy(s);
		g_err = keep;
		return err;
	};

	// primitive id -> triangle index, or -1 - sphere index; objects that are neither never hit (Shape.cpp:44-46)
	parallel_for(d->n_prims, [&](uint32_t i) { primSlot[i] = INT64_MIN; });
	parallel_for(d->n_triangles, [&](uint32_t i) { primSlot[d->tri_prim[i]] = (int64_t)i; });
	for (uint32_t i = 0; i < d->n_spheres; ++i) primSlot[d->sphere_prim[i]] = -1 - (int64_t)i;

	// ---- membership lists: per primitive, the reference leaves holding it, in visiting order ------------------------
	// (Nodes are handled in parallel: counts and slots by atomic increments, then every primitive's few entries are put
	// into visiting order by their rank, so the lists do not depend on the thread count.)
	parallel_for(d->n_nodes, [&](uint32_t i) {
		const mirt_node& n = d->nodes[i];
		if (!(n.b & MIRT_LEAF_FLAG)) return;
		const uint32_t cnt = n.b & ~MIRT_LEAF_FLAG;
		for (uint32_t k = 0; k < cnt; ++k) __atomic_fetch_add(&mCnt[d->leaf_prims[n.a + k]], 1u, __ATOMIC_RELAXED);
	});
	for (uint32_t p = 0; p < d->n_prims; ++p) mOfs[p + 1] = mOfs[p] + mCnt[p];
	if (!memb.alloc((size_t)mOfs[d->n_prims] * 8)) return bail(fail(MIRT_E_NOMEM, "out of host memory"));
	{
		std::vector<uint32_t> fill(d->n_prims);
		parallel_for(d->n_prims, [&](uint32_t p) { fill[p] = mOfs[p]; });
		parallel_for(d->n_nodes, [&](uint32_t i) {
			const mirt_node& n = d->nodes[i];
			if (!(n.b & MIRT_LEAF_FLAG)) return;
			const uint32_t cnt = n.b & ~MIRT_LEAF_FLAG;
			for (uint32_t k = 0; k < cnt; ++k) {
				const uint32_t pos = n.a + k; // rank in the reference's visiting order
				float* m = memb.p + (size_t)__atomic_fetch_add(&fill[d->leaf_prims[pos]], 1u, __ATOMIC_RELAXED) * 8;
				m[0] = n.bmin[0]; m[1] = n.bmin[1]; m[2] = n.bmin[2]; memcpy(m + 3, &pos, 4);
				m[4] = n.bmax[0]; m[5] = n.bmax[1]; m[6] = n.bmax[2]; m[7] = 0.0f;
			}
		});
		parallel_for(d->n_prims, [&](uint32_t p) {
			if (mCnt[p] < 2) return;
			struct Entry { float f[8]; };
			Entry* e = (Entry*)(memb.p + (size_t)mOfs[p] * 8);
			std::sort(e, e + mCnt[p], [](const Entry& a, const Entry& b) { uint32_t x, y; memcpy(&x, a.f + 3, 4); memcpy(&y, b.f + 3, 4); return x < y; });
		});
	}
	up.push(s->memb, memb.p, memb.n * 4);
	// the hierarchy-plus-membership formulation needs child boxes nested in parent boxes (see mirt_bvh.h)
	std::atomic<bool> notNested(false);
	parallel_for(d->n_nodes, [&](uint32_t i) {
		const mirt_node& n = d->nodes[i];
		if (n.b & MIRT_LEAF_FLAG) return;
		if (n.a != MIRT_NO_CHILD && !Packer::inside(d->nodes[n.a], n)) notNested.store(true, std::memory_order_relaxed);
		if (n.b != MIRT_NO_CHILD && !Packer::inside(d->nodes[n.b], n)) notNested.store(true, std::memory_order_relaxed);
	});
	const bool nested = !notNested.load();
	s->nested = nested;

	phase("membership lists");
	// ---- primitive hierarchy ---------------------------------------------------------------------------------------------
	// Bounds are padded so that every point the reference's tests can accept lies inside them: a triangle hit P = o + t*d
	// sits within ~1e-7 * (|v1| + |o|) of the triangle's plane (rounding of t), a sphere hit within
	// ~1e-7 * |o - c|^2 / r of the surface (rounding of the quadratic).  D bounds |o|: rays start at the eye or on a surface.
	BvhBuilder bb;
	float slo[3] = {d->camera.look_from[0], d->camera.look_from[1], d->camera.look_from[2]}, shi[3] = {slo[0], slo[1], slo[2]};
	const uint32_t GRAIN = 1u << 14;
	const uint32_t triChunks = (d->n_triangles + GRAIN - 1) / GRAIN;
	std::vector<uint32_t> chunkKept((size_t)triChunks + 1, 0);
	{
		// scene extent (min and max: the order of combination does not matter) and, per chunk, the triangles that sit in a leaf
		struct Ext { float lo[3], hi[3]; };
		std::vector<Ext> ext(triChunks);
		MirtPool::get().run(triChunks, [&](uint32_t c) {
			const uint32_t a = c * GRAIN, b = a + GRAIN < d->n_triangles ? a + GRAIN : d->n_triangles;
			Ext e = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
			uint32_t kept = 0;
			for (uint32_t i = a; i < b; ++i) {
				const float* v = d->tri_verts + (size_t)i * 9;
				for (int q = 0; q < 3; ++q) for (int k = 0; k < 3; ++k) { e.lo[k] = fminf(e.lo[k], v[3 * q + k]); e.hi[k] = fmaxf(e.hi[k], v[3 * q + k]); }
				kept += mCnt[d->tri_prim[i]] != 0 ? 1u : 0u; // in no leaf (dropped by Partition::insert, Partition.cpp:55-71): never tested
			}
			ext[c] = e;
			chunkKept[c + 1] = kept;
		});
		for (const Ext& e : ext) for (int k = 0; k < 3; ++k) { slo[k] = fminf(slo[k], e.lo[k]); shi[k] = fmaxf(shi[k], e.hi[k]); }
		for (uint32_t c = 0; c < triChunks; ++c) chunkKept[c + 1] += chunkKept[c];
	}
	std::vector<float> sphBox((size_t)d->n_spheres * 6);
	uint32_t sphKept = 0;
	for (uint32_t i = 0; i < d->n_spheres; ++i) {
		const mirt_sphere& sp = d->spheres[i];
		const float* M = sp.M;
		for (int k = 0; k < 3; ++k) {
			const float c = M[0 * 4 + k] * sp.center[0] + M[1 * 4 + k] * sp.center[1] + M[2 * 4 + k] * sp.center[2] + M[3 * 4 + k];
			const float h = fabsf(sp.radius) * sqrtf(M[0 * 4 + k] * M[0 * 4 + k] + M[1 * 4 + k] * M[1 * 4 + k] + M[2 * 4 + k] * M[2 * 4 + k]);
			sphBox[(size_t)i * 6 + k] = c;
			sphBox[(size_t)i * 6 + 3 + k] = h;
			slo[k] = fminf(slo[k], c - h);
			shi[k] = fmaxf(shi[k], c + h);
		}
		sphKept += mCnt[d->sphere_prim[i]] != 0 ? 1u : 0u;
	}
	// D: how far a ray's origin can be from anything it is tested against -- rays start at the camera (slo/shi began there) or on a primitive
	const float D = sqrtf((shi[0] - slo[0]) * (shi[0] - slo[0]) + (shi[1] - slo[1]) * (shi[1] - slo[1]) + (shi[2] - slo[2]) * (shi[2] - slo[2]));
	const uint32_t triKept = chunkKept[triChunks];
	bb.prims.resize((size_t)triKept + sphKept);
	MirtPool::get().run(triChunks, [&](uint32_t c) {
		const uint32_t a = c * GRAIN, b = a + GRAIN < d->n_triangles ? a + GRAIN : d->n_triangles;
		BvhPrim* out = bb.prims.data() + chunkKept[c];
		for (uint32_t i = a; i < b; ++i) {
			if (mCnt[d->tri_prim[i]] == 0) continue;
			const float* v = d->tri_verts + (size_t)i * 9;
			BvhPrim p;
			float maxabs = 0.0f;
			for (int k = 0; k < 3; ++k) {
				p.lo[k] = fminf(v[k], fminf(v[3 + k], v[6 + k]));
				p.hi[k] = fmaxf(v[k], fmaxf(v[3 + k], v[6 + k]));
				maxabs = fmaxf(maxabs, fmaxf(fabsf(p.lo[k]), fabsf(p.hi[k])));
			}
			const float pad = 1e-5f * (maxabs + D) + 1e-7f;
			for (int k = 0; k < 3; ++k) { p.lo[k] -= pad; p.hi[k] += pad; }
			p.ref = i;
			p.kind = 0;
			*out++ = p;
		}
	});
	{
		BvhPrim* out = bb.prims.data() + triKept;
		for (uint32_t i = 0; i < d->n_spheres; ++i) {
			if (mCnt[d->sphere_prim[i]] == 0) continue;
			BvhPrim p;
			const mirt_sphere& sp = d->spheres[i];
			// the largest stretch of the linear part of M^-1 is at most sqrt(largest column sum x largest row sum)
			float colMax = 0.0f, rowMax = 0.0f;
			for (int a = 0; a < 3; ++a) {
				colMax = fmaxf(colMax, fabsf(sp.Minv[a * 4 + 0]) + fabsf(sp.Minv[a * 4 + 1]) + fabsf(sp.Minv[a * 4 + 2]));
				rowMax = fmaxf(rowMax, fabsf(sp.Minv[0 * 4 + a]) + fabsf(sp.Minv[1 * 4 + a]) + fabsf(sp.Minv[2 * 4 + a]));
			}
			const float far = sqrtf(colMax * rowMax) * D;
			for (int k = 0; k < 3; ++k) {
				const float c = sphBox[(size_t)i * 6 + k], h = sphBox[(size_t)i * 6 + 3 + k];
				// Sphere::intersect decides from b^2 - 4ac in OBJECT space (Sphere.cpp:60-66), where the ray's origin is up to
				// far = |M^-1| D from the centre: the rounding of b and c leaves a noise of up to 8e-7 far^2 in the discriminant
				// (a few ulps of b^2 ~ 4 far^2; 2e-7 far^2 is the most the random-scene tests have shown), so the test answers as if the radius were sqrt(r^2 + 8e-7 far^2) -- for a
				// sphere flattened a thousandfold and seen from a few hundred units, hundreds of radii.  The hierarchy's box has to
				// hold every point the reference can report, so it is the box of THAT sphere.
				// (a radius of zero included: from 500 units away it is hit like any other.)
				const float rowNorm = sqrtf(sp.M[0 * 4 + k] * sp.M[0 * 4 + k] + sp.M[1 * 4 + k] * sp.M[1 * 4 + k] + sp.M[2 * 4 + k] * sp.M[2 * 4 + k]);
				const float hh = fmaxf(h, sqrtf(sp.radius * sp.radius + 8e-7f * (far * far)) * rowNorm);
				const float pad = (std::isfinite(hh) ? hh - h : 0.0f) + h * 0.01f + 1e-5f * (fabsf(c) + D) + 1e-7f;
				p.lo[k] = c - h - pad;
				p.hi[k] = c + h + pad;
			}
			p.ref = i;
			p.kind = 1;
			*out++ = p;
		}
	}
	phase("primitive bounds");
	bb.run();
	phase("hierarchy build");

	// primitives in hierarchy order (then the unreferenced ones), per kind
	std::vector<uint32_t> triNew(d->n_triangles), sphNew(d->n_spheres, 0xffffffffu), triOld(d->n_triangles), sphOld;
	std::vector<uint32_t> kindIndex(bb.prims.size());
	parallel_for(d->n_triangles, [&](uint32_t i) { triNew[i] = 0xffffffffu; });
	{
		const uint32_t nP = (uint32_t)bb.prims.size(), chunks = (nP + GRAIN - 1) / GRAIN;
		std::vector<uint32_t> triBefore((size_t)chunks + 1, 0);
		MirtPool::get().run(chunks, [&](uint32_t c) {
			const uint32_t a = c * GRAIN, b = a + GRAIN < nP ? a + GRAIN : nP;
			uint32_t n = 0;
			for (uint32_t i = a; i < b; ++i) n += bb.prims[i].kind == 0 ? 1u : 0u;
			triBefore[c + 1] = n;
		});
		for (uint32_t c = 0; c < chunks; ++c) triBefore[c + 1] += triBefore[c];
		sphOld.resize(nP - triBefore[chunks]);
		MirtPool::get().run(chunks, [&](uint32_t c) {
			const uint32_t a = c * GRAIN, b = a + GRAIN < nP ? a + GRAIN : nP;
			uint32_t t = triBefore[c], sp = a - triBefore[c];
			for (uint32_t i = a; i < b; ++i) {
				const BvhPrim& p = bb.prims[i];
				if (p.kind == 0) { kindIndex[i] = t; triNew[p.ref] = t; triOld[t] = p.ref; ++t; }
				else { kindIndex[i] = sp; sphNew[p.ref] = sp; sphOld[sp] = p.ref; ++sp; }
			}
		});
		uint32_t nt = triBefore[chunks];
		if (nt < d->n_triangles) for (uint32_t i = 0; i < d->n_triangles; ++i) if (triNew[i] == 0xffffffffu) { triNew[i] = nt; triOld[nt] = i; ++nt; }
		for (uint32_t i = 0; i < d->n_spheres; ++i) if (sphNew[i] == 0xffffffffu) { sphNew[i] = (uint32_t)sphOld.size(); sphOld.push_back(i); }
	}
	if (triOld.size() > BVH_FIRST_MASK || sphOld.size() > BVH_FIRST_MASK) return bail(fail(MIRT_E_INVALID, "more than %u primitives of one kind", BVH_FIRST_MASK));

	// triangles in hierarchy order: { N, dot(v1,N) } { v3, E } { a0, c1 } { a1, K } and { v1, area } { v2, 0 }
	// (E, a_i, c1, K: tri_hit's filter)
	if (!tris.alloc((size_t)d->n_triangles * TRI_F4 * 4) || !triX.alloc((size_t)d->n_triangles * 8) || !triInfo.alloc((size_t)d->n_triangles * 4))
		return bail(fail(MIRT_E_NOMEM, "out of host memory"));
	parallel_for(d->n_triangles, [&](uint32_t ni) {
		const uint32_t i = triOld[ni];
		const float* v = d->tri_verts + (size_t)i * 9;
		const h3 v1 = H3(v[0], v[1], v[2]), v2 = H3(v[3], v[4], v[5]), v3 = H3(v[6], v[7], v[8]);
		const h3 N = hnorm(hcross(hsub(v2, v1), hsub(v3, v1)));      // Triangle.cpp:173-175
		const float dv = hdot(v1, N);                                // Triangle.cpp:153
		const h3 e1 = hsub(v3, v2), e2 = hsub(v1, v3);               // Triangle.cpp:148-149
		const float area = hdot(hcross(e1, e2), N) / 2.0f;           // Triangle.cpp:162
		float* t = tris.p + (size_t)ni * TRI_F4 * 4;
		t[0] = N.x; t[1] = N.y; t[2] = N.z; t[3] = dv;
		tri_filter(t + 4, N, v1, v2, v3, area);
		float* x = triX.p + (size_t)ni * 8;
		x[0] = v1.x; x[1] = v1.y; x[2] = v1.z; x[3] = area;
		x[4] = v2.x; x[5] = v2.y; x[6] = v2.z; x[7] = 0.0f;
		const uint32_t prim = d->tri_prim[i];
		uint32_t* inf = triInfo.p + (size_t)ni * 4;
		inf[0] = prim; inf[1] = d->prim_material[prim]; inf[2] = mOfs[prim]; inf[3] = mCnt[prim];
	});
	up.push(s->tris, tris.p, tris.n * 4);
	up.push(s->triX, triX.p, triX.n * 4);
	up.push(s->triInfo, triInfo.p, triInfo.n * 4);
	// spheres: { c, powf(r,2) } M[4] Minv[4]
	sph.assign((size_t)d->n_spheres * SPHERE_F4 * 4, 0.0f);
	sphInfo.assign((size_t)d->n_spheres * 4, 0);
	for (uint32_t ni = 0; ni < d->n_spheres; ++ni) {
		const uint32_t i = sphOld[ni];
		const mirt_sphere& sp = d->spheres[i];
		float* t = sph.data() + (size_t)ni * SPHERE_F4 * 4;
		t[0] = sp.center[0]; t[1] = sp.center[1]; t[2] = sp.center[2];
		t[3] = sp.radius * sp.radius;                                // Sphere.cpp:64 glm::pow(radius, 2.0f): g++ folds the constant exponent 2 into a multiplication
		memcpy(t + 4, sp.M, 64);
		memcpy(t + 20, sp.Minv, 64);
		const uint32_t prim = d->sphere_prim[i];
		uint32_t* inf = sphInfo.data() + (size_t)ni * 4;
		inf[0] = prim; inf[1] = d->prim_material[prim]; inf[2] = mOfs[prim]; inf[3] = mCnt[prim];
	}
	mats.assign((size_t)d->n_materials * 16, 0.0f);
	for (uint32_t i = 0; i < d->n_materials; ++i) {
		const mirt_material& m = d->materials[i];
		float* t = mats.data() + (size_t)i * 16;
		t[0] = m.diffuse[0]; t[1] = m.diffuse[1]; t[2] = m.diffuse[2]; t[3] = m.shininess;
		t[4] = m.specular[0]; t[5] = m.specular[1]; t[6] = m.specular[2];
		t[8] = m.emission[0]; t[9] = m.emission[1]; t[10] = m.emission[2];
		t[12] = m.ambient[0]; t[13] = m.ambient[1]; t[14] = m.ambient[2];
	}
	lights.assign((size_t)d->n_lights * 8, 0.0f);
	for (uint32_t i = 0; i < d->n_lights; ++i) {
		const mirt_light& l = d->lights[i];
		float* t = lights.data() + (size_t)i * 8;
		memcpy(t, l.location, 16);
		memcpy(t + 4, l.color, 12);
	}
	up.push(s->spheres, sph.data(), sph.size() * 4);
	up.push(s->sphInfo, sphInfo.data(), sphInfo.size() * 4);
	up.push(s->materials, mats.data(), mats.size() * 4);
	up.push(s->lights, lights.data(), lights.size() * 4);
	phase("primitive records");

	// 4-wide nodes with quantised child boxes (layout in mirt_device.h)
	uint32_t bvhRoot = BVH_DONE, bvhWideDepth = 0, bvhMaxStack = 0;
	uint32_t bvhDiag[8] = {0};
	float bvhMin[3] = {0, 0, 0}, bvhMax[3] = {-1, -1, -1};
	if (!bb.nodes.empty()) {
		const Collapser col = {bb, kindIndex};
		for (const BvhNode& bn : bb.nodes) bvhDiag[bn.left < 0 ? 1 : 0]++;
		bvhRoot = bb.nodes[0].left < 0 ? col.leafRef(bb.nodes[0]) : col.run(bvhWide, bvhWideDepth, bvhMaxStack, bvhDiag);
		memcpy(bvhMin, bb.nodes[0].lo, 12);
		memcpy(bvhMax, bb.nodes[0].hi, 12);
	}
	up.push(s->bvh, bvhWide.data(), bvhWide.size() * 4);
	cut_boxes(bvhWide.data(), bvhWide.size() / 16, bvhRoot, bvhMin, bvhMax, cut_box_count(), s->cutBoxes);
	phase("hierarchy collapse");

	// ---- the reference tree as it is (MODE_REFWALK) ------------------------------------------------------------------------
	// leaf stream: per leaf a count and the primitives that can be hit at all, under their hierarchy-order numbers
	Packer pk;
	pk.d = d;
	pk.leafPos.resize(d->n_nodes);
	{
		const uint32_t chunks = (d->n_nodes + GRAIN - 1) / GRAIN;
		std::vector<uint64_t> words((size_t)chunks + 1, 0);
		MirtPool::get().run(chunks, [&](uint32_t c) {
			const uint32_t a = c * GRAIN, b = a + GRAIN < d->n_nodes ? a + GRAIN : d->n_nodes;
			uint64_t w = 0;
			for (uint32_t i = a; i < b; ++i) {
				const mirt_node& n = d->nodes[i];
				if (!(n.b & MIRT_LEAF_FLAG)) continue;
				const uint32_t cnt = n.b & ~MIRT_LEAF_FLAG;
				w += 1;
				for (uint32_t k = 0; k < cnt; ++k) w += primSlot[d->leaf_prims[n.a + k]] != INT64_MIN ? 1u : 0u;
			}
			words[c + 1] = w;
		});
		for (uint32_t c = 0; c < chunks; ++c) words[c + 1] += words[c];
		if (words[chunks] >= 0x7fffffffull) return bail(fail(MIRT_E_INVALID, "leaf stream too long"));
		if (!refStream.alloc((size_t)words[chunks])) return bail(fail(MIRT_E_NOMEM, "out of host memory"));
		MirtPool::get().run(chunks, [&](uint32_t c) {
			const uint32_t a = c * GRAIN, b = a + GRAIN < d->n_nodes ? a + GRAIN : d->n_nodes;
			uint32_t at = (uint32_t)words[c];
			for (uint32_t i = a; i < b; ++i) {
				const mirt_node& n = d->nodes[i];
				pk.leafPos[i] = 0;
				if (!(n.b & MIRT_LEAF_FLAG)) continue;
				pk.leafPos[i] = at;
				const uint32_t cnt = n.b & ~MIRT_LEAF_FLAG;
				uint32_t kept = 0;
				for (uint32_t k = 0; k < cnt; ++k) {
					const int64_t sl = primSlot[d->leaf_prims[n.a + k]];
					if (sl == INT64_MIN) continue;
					refStream.p[at + 1 + kept] = sl >= 0 ? triNew[(uint32_t)sl] : (PRIM_SPHERE | sphNew[(uint32_t)(-1 - sl)]);
					++kept;
				}
				refStream.p[at] = kept;
				at += 1 + kept;
			}
		});
	}
	up.push(s->leafs, refStream.p, refStream.n * 4);
	// DScene::nanHit: the first sphere of the last leaf (flat nodes are in pre-order: left to right) that holds one
	ds.nanHit = REF_NONE;
	if (d->n_spheres) {
		for (uint32_t i = d->n_nodes; i-- > 0 && ds.nanHit == REF_NONE;) {
			const mirt_node& n = d->nodes[i];
			if (!(n.b & MIRT_LEAF_FLAG)) continue;
			const uint32_t cnt = n.b & ~MIRT_LEAF_FLAG;
			for (uint32_t k = 0; k < cnt; ++k) {
				const int64_t sl = primSlot[d->leaf_prims[n.a + k]];
				if (sl < 0 && sl != INT64_MIN) { ds.nanHit = PRIM_SPHERE | sphNew[(uint32_t)(-1 - sl)]; break; }
			}
		}
	}
	ds.rootRef = REF_NONE;
	ds.refCull = d->n_spheres == 0 ? 1u : 0u;
	for (int k = 0; k < 3; ++k) { ds.rootMin[k] = 0.0f; ds.rootMax[k] = -1.0f; }
	uint32_t refDepth = 0;
	if (d->n_nodes) {
		const uint32_t r = pk.skipChain(0);
		const mirt_node& rn = d->nodes[r];
		for (int k = 0; k < 3; ++k) { ds.rootMin[k] = rn.bmin[k]; ds.rootMax[k] = rn.bmax[k]; }
		if (Packer::isLeaf(rn)) ds.rootRef = REF_LEAF | pk.leafPos[r];
		else {
			ds.rootRef = 0;
			// A tree deeper than MIRT_TRACE_REFWALK's private stack is still a scene: the other modes never walk it (renders that
			// ask for REFWALK are refused, see refwalk_ok).  Only a tree whose boxes are not nested has no other mode.
			if (!pk.run(r, refWide, refDepth) || (refDepth + 2 > STACK_MAX && !s->nested))
				return bail(fail(MIRT_E_INVALID, "Partition tree depth %u exceeds the traversal stack (%d)", refDepth, STACK_MAX));
		}
	}
	up.push(s->nodes, refWide.data(), refWide.size() * 4);
	s->wideDepth = refDepth;
	phase("reference tree pack");
	s->bvhDepth = bvhWideDepth;
	// Long walks (big scenes) make refilling idle lanes pay; short ones do not (see k_trace).  MIRT_REFILL_MIN overrides.
	s->refillMin = bvhWide.size() / 16 > 500000u ? 36u : 64u;
	if (const char* env = getenv("MIRT_SINGLE_LAUNCH")) s->singleLaunch = atoi(env) != 0;
	if (const char* env = getenv("MIRT_COPY_THREADS")) { const int v = atoi(env); if (v >= 0 && v <= 32) s->copyThreads = v; }
	if (const char* env = getenv("MIRT_REFILL_MIN")) { const int v = atoi(env); if (v >= 1 && v <= 64) s->refillMin = (uint32_t)v; }
	memcpy(s->diag, bvhDiag, sizeof bvhDiag);
	ds.bvhRoot = bvhRoot;
	memcpy(ds.bvhMin, bvhMin, 12);
	memcpy(ds.bvhMax, bvhMax, 12);
	// camera basis (Camera.cpp:4-7,16-19): identical for every pixel, so evaluated once here
	{
		const mirt_camera& c = d->camera;
		const h3 from = H3(c.look_from[0], c.look_from[1], c.look_from[2]);
		const h3 at = H3(c.look_at[0], c.look_at[1], c.look_at[2]);
		const h3 up3 = H3(c.up[0], c.up[1], c.up[2]);
		const h3 fwd = hsub(at, from);
		const h3 nf = hnorm(fwd);
		const h3 w = H3(-nf.x, -nf.y, -nf.z);
		const h3 u = hnorm(hcross(up3, w));
		const h3 v = hnorm(hcross(w, u));
		ds.cam.eye[0] = from.x; ds.cam.eye[1] = from.y; ds.cam.eye[2] = from.z;
		ds.cam.u[0] = u.x; ds.cam.u[1] = u.y; ds.cam.u[2] = u.z;
		ds.cam.v[0] = v.x; ds.cam.v[1] = v.y; ds.cam.v[2] = v.z;
		ds.cam.w[0] = w.x; ds.cam.w[1] = w.y; ds.cam.w[2] = w.z;
		const float rad = (c.fovy_deg / 2.0f) * 0.01745329251994329576923690768489f; // glm::radians
		ds.cam.tanHalf = tanf(rad);
	}
	ds.nLights = d->n_lights;
	ds.maxDepth = d->max_depth;
	for (int k = 0; k < 3; ++k) ds.atten[k] = d->attenuation[k];
	// Shadow queries whose light term is multiplied by an exact zero are not traced (item_ray).  That needs products
	// that cannot turn into inf or NaN on the way -- every material and light colour finite and modest -- and
	// pow(0, shininess) = 0, i.e. shininess > 0, everywhere.  MIRT_SHADOW_CULL=0 switches it off (measurements).
	{
		bool ok = true;
		for (uint32_t i = 0; i < d->n_materials && ok; ++i) {
			const mirt_material& m = d->materials[i];
			for (int k = 0; k < 3; ++k) ok = ok && fabsf(m.diffuse[k]) <= 1e15f && fabsf(m.specular[k]) <= 1e15f;
			ok = ok && m.shininess > 0.0f && std::isfinite(m.shininess);
		}
		for (uint32_t i = 0; i < d->n_lights && ok; ++i) for (int k = 0; k < 3; ++k) ok = ok && fabsf(d->lights[i].color[k]) <= 1e15f;
		if (const char* env = getenv("MIRT_SHADOW_CULL")) ok = ok && atoi(env) != 0;
		ds.shadowCull = ok ? 1u : 0u;
	}
	{
		// cull_bound(): where a ray of this scene can start -- at the camera, or on a primitive
		float far = (fabsf(ds.cam.eye[0]) + fabsf(ds.cam.eye[1])) + fabsf(ds.cam.eye[2]), box = 0.0f;
		for (int k = 0; k < 3; ++k) box += fmaxf(fmaxf(fabsf(ds.bvhMin[k]), fabsf(ds.bvhMax[k])), fmaxf(fabsf(ds.rootMin[k]), fabsf(ds.rootMax[k])));
		far = fmaxf(far, box);
		ds.cullSlack = std::isfinite(far) ? 1e-5f + far * 0x1p-17f * 1.01f : __builtin_inff(); // (camera rays: |alpha u + beta v - w| >= 1, Camera.cpp:21-23)
	}
	s->nLights = d->n_lights;
	s->maxDepth = d->max_depth;

	e = up.finish();
	if (!e && hipSetDevice(device) != hipSuccess) e = fail(MIRT_E_DEVICE, "hipSetDevice(%d) failed", device);
	hipDeviceProp_t prop;
	if (!e && hipGetDeviceProperties(&prop, device) != hipSuccess) e = fail(MIRT_E_DEVICE, "hipGetDeviceProperties failed");
	if (!e) s->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	if (!e) e = scratch_init(device);
	if (!e) { e = s->dbg.ensure(DBG_WORDS * 8); if (!e) (void)hipMemset(s->dbg.p, 0, DBG_WORDS * 8); }
	ds.dbg = (unsigned long long*)s->dbg.p;
	s->ovfStride = (uint32_t)s->numCUs * GRID_BLOCKS_PER_CU * BLOCK;
	// the most entries a walk can hold: along any path, every node leaves behind all its children but one
	const uint32_t stackNeed = bvhMaxStack + 1u;
	s->diag[6] = stackNeed;
	if (getenv("MIRT_VERBOSE")) fprintf(stderr, "mirt: hierarchy of %zu wide nodes, %u levels; a walk's stack can hold %u entries\n", bvhWide.size() / 16, bvhWideDepth, stackNeed);
	// Every kernel keeps at least this many entries in LDS (each variant its own number), so an area sized for the smaller
	// number serves all of them, and a hierarchy that fits the smaller number needs no overflow variant anywhere.
	const uint32_t ldsEntries = std::min(std::min((uint32_t)STACK_LDS_REFILL, (uint32_t)STACK_LDS), (uint32_t)STACK_LDS_FRAME);
	s->deepBvh = stackNeed > ldsEntries;
	s->ovfLevels = s->deepBvh ? stackNeed - ldsEntries : 0u;
	if (!e && s->deepBvh) e = s->hookOvf.ensure((size_t)s->ovfLevels * s->ovfStride * 4);
	if (!e && hipStreamCreate(&s->stream) != hipSuccess) e = fail(MIRT_E_DEVICE, "hipStreamCreate failed");
	if (e) return bail(e);
	phase("upload (what was left of it)");
	ds.nodes = (const float4*)s->nodes.p;
	ds.leafs = (const uint32_t*)s->leafs.p;
	ds.bvh = (const float4*)s->bvh.p;
	ds.memb = (const float4*)s->memb.p;
	ds.tris = (const float4*)s->tris.p;
	ds.triX = (const float4*)s->triX.p;
	ds.triInfo = (const uint4*)s->triInfo.p;
	ds.spheres = (const float4*)s->spheres.p;
	ds.sphInfo = (const uint4*)s->sphInfo.p;
	ds.materials = (const float4*)s->materials.p;
	ds.lights = (const float4*)s->lights.p;
	*out = s;
	return MIRT_OK;
}

void mirt_scene_destroy(mirt_scene* s) {
	if (!s) return;
	(void)hipSetDevice(s->device);
	// the whole device, not stream by stream: a caller's stream may have been destroyed since its last frame
	(void)hipDeviceSynchronize();
	delete s->copyBack;
	s->copyBack = nullptr;
	delete s->png;
	s->png = nullptr;
	for (BlockList* b : s->blockLists) { b->list.release(); delete b; }
	s->blockLists.clear();
	s->nodes.release(); s->leafs.release(); s->bvh.release(); s->memb.release(); s->tris.release(); s->triX.release(); s->triInfo.release();
	s->spheres.release(); s->sphInfo.release(); s->materials.release(); s->lights.release(); s->hookOvf.release(); s->dbg.release();
	for (FrameCtx* c : s->ctxs) {
		c->ws.release();
		if (c->ev0) (void)hipEventDestroy(c->ev0);
		if (c->ev1) (void)hipEventDestroy(c->ev1);
		for (hipEvent_t e : c->launchEv) (void)hipEventDestroy(e);
		delete c;
	}
	if (s->stream) (void)hipStreamDestroy(s->stream);
	if (s->probeStream) (void)hipStreamDestroy(s->probeStream);
	delete s;
}

// ---- scene cache: the device form of a scene in a file (what mirt_scene_create spends its seconds on) ----------------------------
// Layout: header { magic, version, sizeof(DScene), sizeof(SceneMeta), user[4], payload checksum }, the DScene record
// (pointers meaningless), SceneMeta, then for each of the eleven device arrays a 64-bit byte count and the bytes.
namespace {
const char SCENE_MAGIC[8] = {'M', 'I', 'R', 'T', 'S', 'C', 'N', '1'};
const uint32_t SCENE_VERSION = 4;
struct SceneHeader {
	char magic[8];
	uint32_t version, dsceneBytes, metaBytes, pad;
	uint32_t user[4];
	uint64_t checksum; // sum_of() chained over everything after the header, piece by piece as written
};
struct SceneMeta {
	uint32_t nested, bvhDepth, deepBvh, nLights;
	int32_t maxDepth;
	uint32_t wideDepth, refillMin, ovfLevels;
	uint32_t diag[8];
};
inline uint64_t fnv1a(uint64_t h, const void* p, size_t n) {
	// eight bytes at a time (a byte-wise pass over the 3.5 GB of a 10 M-triangle scene would take seconds)
	const uint8_t* b = (const uint8_t*)p;
	size_t i = 0;
	for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, b + i, 8); h = (h ^ w) * 0x100000001b3ull; }
	for (; i < n; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
	return h;
}
// The checksum of one piece of the file, chained onto h: FNV-1a of the piece if it is small, otherwise FNV-1a of the
// FNV-1a digests of its 1-MiB blocks, which are taken on all threads (the result does not depend on their number).
inline uint64_t sum_of(uint64_t h, const void* p, size_t n) {
	const size_t blk = (size_t)1 << 20;
	if (n <= blk) return fnv1a(h, p, n);
	const uint32_t blocks = (uint32_t)((n + blk - 1) / blk);
	std::vector<uint64_t> dig(blocks);
	MirtPool::get().run(blocks, [&](uint32_t b) {
		const size_t a = (size_t)b * blk, len = a + blk < n ? blk : n - a;
		dig[b] = fnv1a(0xcbf29ce484222325ull, (const uint8_t*)p + a, len);
	});
	return fnv1a(h, dig.data(), (size_t)blocks * 8);
}
DevBuf* scene_buffers(mirt_scene* s, int k) {
	DevBuf* const all[11] = {&s->nodes, &s->leafs, &s->bvh, &s->memb, &s->tris, &s->triX, &s->triInfo, &s->spheres, &s->sphInfo, &s->materials, &s->lights};
	return all[k];
}
void scene_bind(mirt_scene* s) {
	DScene& ds = s->ds;
	ds.nodes = (const float4*)s->nodes.p;
	ds.leafs = (const uint32_t*)s->leafs.p;
	ds.bvh = (const float4*)s->bvh.p;
	ds.memb = (const float4*)s->memb.p;
	ds.tris = (const float4*)s->tris.p;
	ds.triX = (const float4*)s->triX.p;
	ds.triInfo = (const uint4*)s->triInfo.p;
	ds.spheres = (const float4*)s->spheres.p;
	ds.sphInfo = (const uint4*)s->sphInfo.p;
	ds.materials = (const float4*)s->materials.p;
	ds.lights = (const float4*)s->lights.p;
	ds.dbg = (unsigned long long*)s->dbg.p;
}
}  // namespace

int mirt_scene_save(mirt_scene* s, const char* path, const uint32_t user[4]) {
	if (!s || !path) return fail(MIRT_E_INVALID, "NULL argument");
	HIPCHK(hipSetDevice(s->device));
	HIPCHK(hipDeviceSynchronize());
	FILE* f = fopen(path, "wb");
	if (!f) return fail(MIRT_E_INVALID, "cannot create %s", path);
	SceneHeader h;
	memset(&h, 0, sizeof h);
	memcpy(h.magic, SCENE_MAGIC, 8);
	h.version = SCENE_VERSION;
	h.dsceneBytes = sizeof(DScene);
	h.metaBytes = sizeof(SceneMeta);
	if (user) memcpy(h.user, user, sizeof h.user);
	SceneMeta m;
	memset(&m, 0, sizeof m);
	m.nested = s->nested; m.bvhDepth = s->bvhDepth; m.deepBvh = s->deepBvh; m.nLights = s->nLights; m.maxDepth = s->maxDepth;
	m.wideDepth = s->wideDepth; m.refillMin = s->refillMin; m.ovfLevels = s->ovfLevels;
	memcpy(m.diag, s->diag, sizeof m.diag);
	bool ok = fwrite(&h, sizeof h, 1, f) == 1;
	uint64_t sum = 0xcbf29ce484222325ull;
	auto put = [&](const void* p, size_t n) { sum = sum_of(sum, p, n); ok = ok && (n == 0 || fwrite(p, n, 1, f) == 1); };
	put(&s->ds, sizeof s->ds);
	put(&m, sizeof m);
	HostBuf<uint8_t> tmp;
	for (int k = 0; k < 11 && ok; ++k) {
		DevBuf* b = scene_buffers(s, k);
		const uint64_t n = b->bytes;
		put(&n, 8);
		if (!tmp.alloc(n)) { fclose(f); return fail(MIRT_E_NOMEM, "out of host memory"); }
		if (n && hipMemcpy(tmp.p, b->p, n, hipMemcpyDeviceToHost) != hipSuccess) { fclose(f); return fail(MIRT_E_DEVICE, "mirt_scene_save: copy from the device failed"); }
		put(tmp.p, n);
	}
	h.checksum = sum;
	ok = ok && fseek(f, 0, SEEK_SET) == 0 && fwrite(&h, sizeof h, 1, f) == 1;
	ok = (fclose(f) == 0) && ok;
	if (!ok) return fail(MIRT_E_INVALID, "writing %s failed", path);
	return MIRT_OK;
}

int mirt_scene_load(const char* path, int device, mirt_scene** out, uint32_t user_out[4]) {
	if (!path || !out) return fail(MIRT_E_INVALID, "NULL argument");
	*out = nullptr;
	// The file is mapped, not read: the arrays go to the device straight from the mapping on the uploader's thread while
	// the pool's threads take the checksum of the same pages.
	struct Mapping {
		int fd = -1;
		const uint8_t* p = nullptr;
		size_t n = 0;
		~Mapping() {
			if (p) munmap((void*)p, n);
			if (fd >= 0) close(fd);
		}
	} map;
	map.fd = open(path, O_RDONLY);
	if (map.fd < 0) return fail(MIRT_E_INVALID, "cannot open %s", path);
	struct stat sb;
	if (fstat(map.fd, &sb) != 0 || (size_t)sb.st_size < sizeof(SceneHeader)) return fail(MIRT_E_INVALID, "%s is not a scene cache file", path);
	map.n = (size_t)sb.st_size;
	void* mp = mmap(nullptr, map.n, PROT_READ, MAP_PRIVATE, map.fd, 0);
	if (mp == MAP_FAILED) return fail(MIRT_E_INVALID, "cannot map %s", path);
	map.p = (const uint8_t*)mp;
	SceneHeader h;
	memcpy(&h, map.p, sizeof h);
	if (memcmp(h.magic, SCENE_MAGIC, 8) != 0) return fail(MIRT_E_INVALID, "%s is not a scene cache file", path);
	if (h.version != SCENE_VERSION || h.dsceneBytes != sizeof(DScene) || h.metaBytes != sizeof(SceneMeta))
		return fail(MIRT_E_INVALID, "%s was written by another version of the library (format %u, this library reads %u): build the scene again", path, h.version, SCENE_VERSION);
	mirt_scene* s = new (std::nothrow) mirt_scene();
	if (!s) return fail(MIRT_E_NOMEM, "out of host memory");
	s->device = device;
	SceneMeta m;
	memset(&m, 0, sizeof m);
	uint64_t sum = 0xcbf29ce484222325ull;
	size_t at = sizeof h;
	const char* why = "truncated file";
	bool ok = map.n - at >= sizeof s->ds + sizeof m;
	int devErr = MIRT_OK;
	std::string devMsg;
	const uint8_t* wideAt = nullptr; // the hierarchy's nodes in the mapping
	size_t wideBytes = 0;
	{
		Uploader up(device); // (also the first to touch the device: the runtime starts up beside the checksum)
		if (ok) {
			memcpy(&s->ds, map.p + at, sizeof s->ds);
			sum = sum_of(sum, map.p + at, sizeof s->ds);
			at += sizeof s->ds;
			memcpy(&m, map.p + at, sizeof m);
			sum = sum_of(sum, map.p + at, sizeof m);
			at += sizeof m;
		}
		static const uint32_t recordBytes[11] = {64, 4, 64, 32, 64, 32, 16, 144, 16, 64, 32};
		for (int k = 0; k < 11 && ok; ++k) {
			uint64_t n = 0;
			if (map.n - at < 8) { ok = false; break; }
			memcpy(&n, map.p + at, 8);
			sum = sum_of(sum, map.p + at, 8);
			at += 8;
			if (n > (1ull << 40) || (n % recordBytes[k] != 0 && n != 16)) { ok = false; why = "array size not a whole number of records"; break; }
			if (map.n - at < n) { ok = false; break; }
			up.push(*scene_buffers(s, k), map.p + at, n);
			if (k == 2) { wideAt = map.p + at; wideBytes = (size_t)n; }
			sum = sum_of(sum, map.p + at, n);
			at += n;
		}
		devErr = up.finish();
		if (devErr) devMsg = g_err;
	}
	if (ok && sum != h.checksum) { ok = false; why = "checksum mismatch (file damaged)"; }
	if (ok && (m.maxDepth > MIRT_MAX_DEPTH_LIMIT || m.nLights != s->ds.nLights || s->lights.bytes < (size_t)m.nLights * 32)) { ok = false; why = "inconsistent header"; }
	if (ok && devErr) {
		mirt_scene_destroy(s);
		g_err = devMsg;
		return devErr;
	}
	hipDeviceProp_t prop;
	if (ok && (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess)) {
		mirt_scene_destroy(s);
		return fail(MIRT_E_DEVICE, "device %d: no properties", device);
	}
	if (ok) {
		s->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
		const int e = scratch_init(device);
		if (e) {
			const std::string keep = g_err;
			mirt_scene_destroy(s);
			g_err = keep;
			return e;
		}
	}
	if (ok && (s->dbg.ensure(DBG_WORDS * 8) || hipMemset(s->dbg.p, 0, DBG_WORDS * 8) != hipSuccess)) { ok = false; why = "device allocation failed"; }
	if (!ok) {
		mirt_scene_destroy(s);
		return fail(MIRT_E_INVALID, "%s: %s", path, why);
	}
	s->nested = m.nested != 0; s->bvhDepth = m.bvhDepth; s->nLights = m.nLights; s->maxDepth = m.maxDepth;
	s->wideDepth = m.wideDepth; s->refillMin = m.refillMin;
	memcpy(s->diag, m.diag, sizeof m.diag);
	if (wideAt && ((uintptr_t)wideAt & 3u) == 0)
		cut_boxes((const uint32_t*)wideAt, wideBytes / 64, s->ds.bvhRoot, s->ds.bvhMin, s->ds.bvhMax, cut_box_count(), s->cutBoxes);
	else if (s->ds.bvhRoot != BVH_DONE) { // (cannot happen with files this library wrote) the bounds of the whole scene
		s->cutBoxes.assign(s->ds.bvhMin, s->ds.bvhMin + 3);
		s->cutBoxes.insert(s->cutBoxes.end(), s->ds.bvhMax, s->ds.bvhMax + 3);
	}
	// How much of the traversal stack lies beyond LDS depends on THIS build's STACK_LDS, not on the build that wrote the
	// file: derived again from the stack need of the hierarchy (diag[6]), exactly as mirt_scene_create does.
	{
		const uint32_t ldsEntries = std::min(std::min((uint32_t)STACK_LDS_REFILL, (uint32_t)STACK_LDS), (uint32_t)STACK_LDS_FRAME);
		const uint32_t stackNeed = s->diag[6];
		if (stackNeed == 0 || stackNeed > 4096) { mirt_scene_destroy(s); return fail(MIRT_E_INVALID, "%s: inconsistent header (traversal stack need %u)", path, stackNeed); }
		s->deepBvh = stackNeed > ldsEntries;
		s->ovfLevels = s->deepBvh ? stackNeed - ldsEntries : 0u;
	}
	s->ovfStride = (uint32_t)s->numCUs * GRID_BLOCKS_PER_CU * BLOCK;
	if (const char* env = getenv("MIRT_SINGLE_LAUNCH")) s->singleLaunch = atoi(env) != 0;
	if (const char* env = getenv("MIRT_COPY_THREADS")) { const int v = atoi(env); if (v >= 0 && v <= 32) s->copyThreads = v; }
	if (const char* env = getenv("MIRT_REFILL_MIN")) { const int v = atoi(env); if (v >= 1 && v <= 64) s->refillMin = (uint32_t)v; }
	if (s->deepBvh && s->hookOvf.ensure((size_t)s->ovfLevels * s->ovfStride * 4)) { mirt_scene_destroy(s); return MIRT_E_NOMEM; }
	if (hipStreamCreate(&s->stream) != hipSuccess) { mirt_scene_destroy(s); return fail(MIRT_E_DEVICE, "hipStreamCreate failed"); }
	scene_bind(s);
	if (user_out) memcpy(user_out, h.user, sizeof h.user);
	*out = s;
	return MIRT_OK;
}

int mirt_release_stream(mirt_scene* s, void* hip_stream) {
	if (!s) return fail(MIRT_E_INVALID, "scene handle is NULL");
	HIPCHK(hipSetDevice(s->device));
	for (size_t i = 0; i < s->ctxs.size(); ++i) {
		FrameCtx* c = s->ctxs[i];
		if (c->stream != (hipStream_t)hip_stream) continue;
		HIPCHK(hipStreamSynchronize(c->stream));
		c->ws.release();
		if (c->ev0) (void)hipEventDestroy(c->ev0);
		if (c->ev1) (void)hipEventDestroy(c->ev1);
		for (hipEvent_t e : c->launchEv) (void)hipEventDestroy(e);
		delete c;
		s->ctxs.erase(s->ctxs.begin() + (long)i);
		return MIRT_OK;
	}
	return MIRT_OK; // no frame was ever issued on it: nothing is held
}

size_t mirt_tile_buffer_bytes(const mirt_render_params* p) {
	if (!p || p->tile_stride == 0 || p->first_tile >= p->tile_stride) return 0;
	if (!packed_layout(p)) return (size_t)p->width * p->height * 3;
	const uint32_t tw = p->tile_w ? p->tile_w : 32, th = p->tile_h ? p->tile_h : 32;
	// every rank's buffer has the size of rank 0's (the largest), so gathered buffers are equally spaced
	mirt_render_params q = *p;
	q.first_tile = 0;
	return (size_t)tile_geom(&q, tw, th).nSlots * 3;
}

int mirt_render_device(mirt_scene* s, const mirt_render_params* p, void* d_out, size_t d_out_bytes, void* hip_stream) {
	if (!p) return fail(MIRT_E_INVALID, "render params are NULL");
	return render_async(s, p, d_out, d_out_bytes, (hipStream_t)hip_stream, packed_layout(p));
}

int mirt_render_device_noalloc(mirt_scene* s, const mirt_render_params* p, void* d_out, size_t d_out_bytes, void* hip_stream) {
	if (!p) return fail(MIRT_E_INVALID, "render params are NULL");
	const bool packed = packed_layout(p);
	hipStream_t st = (hipStream_t)hip_stream;
	uint32_t tileW, tileH;
	TileGeom g;
	size_t need;
	int e = frame_checks(s, p, d_out, d_out_bytes, packed, tileW, tileH, g, need);
	if (e) return e;
	if (s->device < 0 || s->device >= MIRT_MAX_DEVICES || !g_scratchPtr[s->device]) return fail(MIRT_E_INVALID, "no scratch array on device %d", s->device);
	HIPCHK(hipSetDevice(s->device));
	if (g.nSlots == 0) return MIRT_OK;
	if (s->maxDepth < 0) { // computePixelColor returns Color() for every pixel (main.cpp:177,199)
		HIPCHK(hipMemsetAsync(d_out, 0, need, st));
		return MIRT_OK;
	}
	// the arrays of a frame, carved out of this device's scratch array one after the other, each 256-byte aligned
	const bool single = single_launch(s, p, g);
	const bool records = single && frame_uses_records(g, true);
	const FrameLayout L = frame_layout(s, (size_t)g.nSlots, single, records);
	unsigned char* const base = g_scratchPtr[s->device];
	void* mem[WS_N];
	size_t at = 0;
	for (int k = 0; k < WS_N; ++k) {
		mem[k] = base + at;
		at += (L.bytes[k] + 255) & ~(size_t)255;
	}
	if (at > (size_t)MIRT_NOALLOC_SCRATCH_BYTES)
		return fail(MIRT_E_INVALID, "mirt_render_device_noalloc: this frame needs %zu bytes of working memory (%llu pixel slots, %u lights, maxdepth %d), the fixed scratch array holds %zu; use mirt_render_device",
		            at, (unsigned long long)g.nSlots, s->nLights, s->maxDepth, (size_t)MIRT_NOALLOC_SCRATCH_BYTES);
	// counters, path depths and the output are cleared inside enqueue_frame; every other array is written before it is read
	return enqueue_frame(s, nullptr, p, g, tileW, tileH, packed, single, records, mem, d_out, need, st);
}

int mirt_scene_query(mirt_scene* s, uint32_t what, uint64_t* v) {
	if (!s || !v) return fail(MIRT_E_INVALID, "NULL argument");
	switch (what) {
	case MIRT_Q_HIERARCHY_DEPTH: *v = s->bvhDepth; break;
	case MIRT_Q_HIERARCHY_NODES: *v = s->bvh.bytes / 64; break;
	case MIRT_Q_TREE_NESTED: *v = s->nested ? 1 : 0; break;
	case MIRT_Q_MEMBERSHIP_ENTRIES: *v = s->memb.bytes / 32; break;
	case MIRT_Q_DEVICE_BYTES:
		*v = s->nodes.bytes + s->leafs.bytes + s->bvh.bytes + s->memb.bytes + s->tris.bytes + s->triX.bytes + s->triInfo.bytes + s->spheres.bytes + s->sphInfo.bytes +
		     s->materials.bytes + s->lights.bytes + s->hookOvf.bytes;
		break;
	case MIRT_Q_TREE_WIDE_DEPTH: *v = s->wideDepth; break;
	default:
		if (what >= 120 && what < 128) { *v = s->diag[what - 120]; break; }
#if MIRT_STATS
		if (what == 116) { // triangle tests that fell back from the filter to the reference's operations (read and clear)
			unsigned long long x = 0, z = 0;
			HIPCHK(hipSetDevice(s->device));
			HIPCHK(hipDeviceSynchronize());
			HIPCHK(hipMemcpyFromSymbol(&x, HIP_SYMBOL(g_statFallback), 8));
			HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_statFallback), &z, 8));
			*v = x;
			break;
		}
		if (what == 119) { // leaf boxes tested by the membership check (read and clear)
			unsigned long long x = 0, z = 0;
			HIPCHK(hipSetDevice(s->device));
			HIPCHK(hipDeviceSynchronize());
			HIPCHK(hipMemcpyFromSymbol(&x, HIP_SYMBOL(g_statMembEntries), 8));
			HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_statMembEntries), &z, 8));
			*v = x;
			break;
		}
		if (what == 117 || what == 118) { // traversal-stack pushes: all / to the overflow area (read and clear)
			unsigned long long x[2] = {0, 0}, z[2] = {0, 0};
			HIPCHK(hipSetDevice(s->device));
			HIPCHK(hipDeviceSynchronize());
			HIPCHK(hipMemcpyFromSymbol(x, HIP_SYMBOL(g_statPush), 16));
			if (what == 118) HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_statPush), z, 16));
			*v = x[what - 117];
			break;
		}
#endif
		if (what >= 140 && what < 156) { // k_frame timeline words of MIRT_STATS builds (read and clear), see k_frame
			unsigned long long x = 0;
			HIPCHK(hipSetDevice(s->device));
			HIPCHK(hipDeviceSynchronize());
			HIPCHK(hipMemcpy(&x, (char*)s->dbg.p + (16 + what - 140) * 8, 8, hipMemcpyDeviceToHost));
			HIPCHK(hipMemset((char*)s->dbg.p + (16 + what - 140) * 8, 0, 8));
			*v = x;
			break;
		}
#if MIRT_TIMELINE
		if (what >= 1000 && what < 1000 + 4 * 8192) {
			unsigned long long x = 0;
			HIPCHK(hipSetDevice(s->device));
			HIPCHK(hipDeviceSynchronize());
			HIPCHK(hipMemcpy(&x, (char*)s->dbg.p + (40 + what - 1000) * 8, 8, hipMemcpyDeviceToHost));
			*v = x;
			break;
		}
#endif
		if (what >= 100 && what < 116) { // diagnostic event counters of MIRT_STATS builds (read and clear)
			unsigned long long x = 0;
			HIPCHK(hipSetDevice(s->device));
			HIPCHK(hipDeviceSynchronize());
			HIPCHK(hipMemcpy(&x, (char*)s->dbg.p + (what - 100) * 8, 8, hipMemcpyDeviceToHost));
			HIPCHK(hipMemset((char*)s->dbg.p + (what - 100) * 8, 0, 8));
			*v = x;
			break;
		}
		return fail(MIRT_E_INVALID, "unknown query %u", what);
	}
	return MIRT_OK;
}

int mirt_set_frames_in_flight(mirt_scene* s, uint32_t n) {
	if (!s) return fail(MIRT_E_INVALID, "scene handle is NULL");
	s->framesHint = n;
	return MIRT_OK;
}

int mirt_clock_probe_start(mirt_scene* s, uint32_t microseconds) {
	if (!s) return fail(MIRT_E_INVALID, "scene handle is NULL");
	if (microseconds == 0 || microseconds > 1000000u) return fail(MIRT_E_INVALID, "clock probe of %u us: 1..1000000 allowed", microseconds);
	HIPCHK(hipSetDevice(s->device));
	if (!s->probeStream) HIPCHK(hipStreamCreateWithFlags(&s->probeStream, hipStreamNonBlocking));
	unsigned long long* out = (unsigned long long*)s->dbg.p + DBG_CLOCK;
	HIPCHK(hipMemsetAsync(out, 0, 16, s->probeStream));
	hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s->probeStream, out, (unsigned long long)microseconds * 100ull);
	HIPCHK(hipGetLastError());
	return MIRT_OK;
}

int mirt_clock_probe_read(mirt_scene* s, double* shader_hz) {
	if (!s || !shader_hz) return fail(MIRT_E_INVALID, "NULL argument");
	*shader_hz = 0.0;
	if (!s->probeStream) return fail(MIRT_E_INVALID, "no clock probe was started");
	HIPCHK(hipSetDevice(s->device));
	HIPCHK(hipStreamSynchronize(s->probeStream));
	unsigned long long v[2] = {0, 0};
	HIPCHK(hipMemcpy(v, (unsigned long long*)s->dbg.p + DBG_CLOCK, 16, hipMemcpyDeviceToHost));
	if (v[1] == 0) return fail(MIRT_E_DEVICE, "the clock probe left no reading");
	*shader_hz = (double)v[0] / (double)v[1] * 1e8;
	return MIRT_OK;
}

int mirt_set_profiling(mirt_scene* s, int enabled) {
	if (!s) return fail(MIRT_E_INVALID, "scene handle is NULL");
	s->profiling = enabled != 0;
	for (FrameCtx* c : s->ctxs) c->nLaunches = 0;
	return MIRT_OK;
}

int mirt_get_launch_times(mirt_scene* s, void* hip_stream, float* ms_out, int32_t* kinds_out, uint32_t capacity, uint32_t* n_out) {
	if (!s || !ms_out || !kinds_out || !n_out) return fail(MIRT_E_INVALID, "NULL argument");
	HIPCHK(hipSetDevice(s->device));
	HIPCHK(hipStreamSynchronize((hipStream_t)hip_stream));
	*n_out = 0;
	if (!s->profiling) return fail(MIRT_E_INVALID, "profiling is off (mirt_set_profiling)");
	FrameCtx* c = ctx_for(s, (hipStream_t)hip_stream, false);
	if (!c && !hip_stream) c = ctx_for(s, s->stream, false); // NULL also names the stream of the synchronous mirt_render
	if (!c) return MIRT_OK;
	for (uint32_t i = 0; i < c->nLaunches && i < capacity; ++i) {
		HIPCHK(hipEventElapsedTime(&ms_out[i], c->launchEv[i], c->launchEv[i + 1]));
		kinds_out[i] = c->launchKind[i];
		*n_out = i + 1;
	}
	return MIRT_OK;
}

int mirt_get_stats(mirt_scene* s, void* hip_stream, mirt_stats* stats) {
	if (!s || !stats) return fail(MIRT_E_INVALID, "NULL argument");
	HIPCHK(hipSetDevice(s->device));
	return read_stats(s, (hipStream_t)hip_stream, stats);
}

int mirt_render(mirt_scene* s, const mirt_render_params* p, uint8_t* bgr_out, mirt_stats* stats) {
	uint32_t tileW, tileH;
	int e = check_params(s, p, tileW, tileH);
	if (e) return e;
	if (!bgr_out) return fail(MIRT_E_INVALID, "bgr_out is NULL");
	HIPCHK(hipSetDevice(s->device));
	const bool whole = p->tile_stride == 1;
	const TileGeom g = tile_geom(p, tileW, tileH);
	const size_t bytes = whole ? (size_t)p->width * p->height * 3 : (size_t)g.nSlots * 3;
	if (bytes == 0) { if (stats) memset(stats, 0, sizeof *stats); return MIRT_OK; }
	FrameCtx* c = ctx_for(s, s->stream, true);
	if (!c) return MIRT_E_INVALID;
	if ((e = c->ws.out.ensure(bytes))) return e;
	if ((e = render_async(s, p, c->ws.out.p, bytes, s->stream, !whole))) return e;
	// Behind the frame a small launch notes which parts of the image hold anything but black (k_note_parts, in words of the
	// frame's counter block that the frame itself does not use), and the copy back fetches only the bands with such a part.
	// MIRT_COPY_SPARSE=0: every band is fetched, queued behind the frame in advance.
	const char* sparseEnv = getenv("MIRT_COPY_SPARSE"); // (read per call: tests switch it)
	const bool sparse = !(sparseEnv && *sparseEnv == '0');
	const bool staged = whole && s->copyThreads > 0 && bytes >= (4u << 20);
	uint32_t shift = 20;
	while ((bytes >> shift) >= CTR_NPARTS) ++shift;
	// (only a frame that really ran has a counter block to note in: with maxdepth < 0 the image is a plain clear and no
	// workspace exists)
	const bool noted = staged && sparse && c->lastValid && c->lastCtr != nullptr;
	if (noted) {
		const uint32_t nParts = (uint32_t)((bytes + ((size_t)1 << shift) - 1) >> shift), blocksPerPart = 64;
		hipLaunchKernelGGL(k_note_parts, dim3(nParts * blocksPerPart), dim3(BLOCK), 0, s->stream, (const uint8_t*)c->ws.out.p, bytes, shift, blocksPerPart, const_cast<uint32_t*>(c->lastCtr) + CTR_PARTS);
	}
	if (whole) {
		if (staged) {
			if (!s->copyBack) s->copyBack = new (std::nothrow) CopyBack(s->device, s->copyThreads);
			if (s->copyBack && s->copyBack->run(c->ws.out.p, bgr_out, bytes, s->stream, noted ? c->lastCtr + CTR_PARTS : nullptr, shift)) {
				if (stats) return read_stats(s, s->stream, stats);
				return MIRT_OK;
			}
			(void)hipGetLastError(); // staging not available: the plain copy below
		}
		HIPCHK(hipMemcpyAsync(bgr_out, c->ws.out.p, bytes, hipMemcpyDeviceToHost, s->stream));
		HIPCHK(hipStreamSynchronize(s->stream));
	}
	else {
		std::vector<uint8_t> tmp(bytes);
		HIPCHK(hipMemcpyAsync(tmp.data(), c->ws.out.p, bytes, hipMemcpyDeviceToHost, s->stream));
		HIPCHK(hipStreamSynchronize(s->stream));
		for (uint32_t lt = 0; lt < g.nLocal; ++lt) {
			const uint64_t k = p->first_tile + (uint64_t)lt * p->tile_stride;
			const uint32_t x0 = (uint32_t)(k % g.tilesX) * tileW, y0 = (uint32_t)(k / g.tilesX) * tileH;
			for (uint32_t y = 0; y < tileH && y0 + y < p->height; ++y) {
				const uint32_t wpx = x0 + tileW <= p->width ? tileW : p->width - x0;
				memcpy(bgr_out + ((size_t)(y0 + y) * p->width + x0) * 3, tmp.data() + ((size_t)lt * tileW * tileH + (size_t)y * tileW) * 3, (size_t)wpx * 3);
			}
		}
	}
	if (stats) return read_stats(s, s->stream, stats);
	return MIRT_OK;
}

int mirt_render_png(mirt_scene* s, const mirt_render_params* p, const char* path, mirt_stats* stats) {
	uint32_t tileW, tileH;
	int e = check_params(s, p, tileW, tileH);
	if (e) return e;
	if (!path) return fail(MIRT_E_INVALID, "path is NULL");
	if (packed_layout(p)) return fail(MIRT_E_INVALID, "mirt_render_png writes whole images only (tile_stride 1, default tiles)");
	HIPCHK(hipSetDevice(s->device));
	const uint32_t w = p->width, h = p->height;
	const size_t bytes = (size_t)w * h * 3, line = 1 + (size_t)w * 3, rowBytes = line * h;
	FrameCtx* c = ctx_for(s, s->stream, true);
	if (!c) return MIRT_E_INVALID;
	if ((e = c->ws.out.ensure(bytes))) return e;
	if (!s->png) s->png = new (std::nothrow) PngPipe();
	PngPipe* pp = s->png;
	if (!pp) return fail(MIRT_E_NOMEM, "out of host memory");
	if ((e = pp->rows.ensure(rowBytes)) || (e = pp->dflags.ensure(MIRT_PNG_MAX_STRIPES * 4))) return e;
	if (rowBytes > pp->stagingBytes) {
		if (pp->staging) (void)hipHostFree(pp->staging);
		pp->staging = nullptr;
		pp->stagingBytes = 0;
		HIPCHK(hipHostMalloc(&pp->staging, rowBytes, hipHostMallocDefault));
		pp->stagingBytes = rowBytes;
	}
	if (!pp->flags) HIPCHK(hipHostMalloc((void**)&pp->flags, MIRT_PNG_MAX_STRIPES * 4, hipHostMallocDefault));
	if (!pp->evFlags) HIPCHK(hipEventCreateWithFlags(&pp->evFlags, hipEventDisableTiming));
	// bands of about 2 MB, each a whole number of 256-byte pieces; a stripe's rows have arrived when the band that holds
	// their last byte has
	int want = (int)((rowBytes + (2u << 20) - 1) / (2u << 20));
	want = want < 1 ? 1 : (want > PngPipe::MAX_BANDS ? PngPipe::MAX_BANDS : want);
	while (pp->nEv < want) { HIPCHK(hipEventCreateWithFlags(&pp->ev[pp->nEv], hipEventDisableTiming)); ++pp->nEv; }
	const size_t band = ((rowBytes + want - 1) / want + 255) & ~(size_t)255;
	const int nBands = (int)((rowBytes + band - 1) / band);
	const uint32_t nStripes = mirt_png_stripes(w, h), rowsPerStripe = (h + nStripes - 1) / nStripes;
	// the frame, its PNG rows, the stripes' flags, then the rows band by band: all queued before the host waits for anything
	if ((e = render_async(s, p, c->ws.out.p, bytes, s->stream, false))) return e;
	HIPCHK(hipMemsetAsync(pp->dflags.p, 0, MIRT_PNG_MAX_STRIPES * 4, s->stream));
	{
		const uint64_t n = (uint64_t)w * h;
		const uint64_t blocks = (n + BLOCK - 1) / BLOCK;
		hipLaunchKernelGGL(k_png_rows, dim3((uint32_t)(blocks < 65536u * 16u ? blocks : 65536u * 16u)), dim3(BLOCK), 0, s->stream, (const uint8_t*)c->ws.out.p, w, h, rowsPerStripe,
		                   (uint8_t*)pp->rows.p, (uint32_t*)pp->dflags.p);
		HIPCHK(hipGetLastError());
	}
	HIPCHK(hipMemcpyAsync(pp->flags, pp->dflags.p, MIRT_PNG_MAX_STRIPES * 4, hipMemcpyDeviceToHost, s->stream));
	HIPCHK(hipEventRecord(pp->evFlags, s->stream));
	for (int k = 0; k < nBands; ++k) {
		const size_t at = (size_t)k * band, len = at + band <= rowBytes ? band : rowBytes - at;
		HIPCHK(hipMemcpyAsync((uint8_t*)pp->staging + at, (const uint8_t*)pp->rows.p + at, len, hipMemcpyDeviceToHost, s->stream));
		HIPCHK(hipEventRecord(pp->ev[k], s->stream));
	}
	const int device = s->device;
	e = mirt_png_encode(path, w, h, 0, 2, 3 /* Z_RLE: the rows are filtered */, [&](uint32_t stripe, uint32_t row0, uint32_t row1, std::vector<uint8_t>&, bool* zero) -> const uint8_t* {
		static thread_local int current = -1;
		if (current != device) { if (hipSetDevice(device) != hipSuccess) return nullptr; current = device; }
		if (hipEventSynchronize(pp->evFlags) != hipSuccess) return nullptr;
		if (stripe < MIRT_PNG_MAX_STRIPES && pp->flags[stripe] == 0u) { *zero = true; return nullptr; }
		const size_t lastByte = (size_t)row1 * line - 1;
		if (hipEventSynchronize(pp->ev[lastByte / band]) != hipSuccess) return nullptr;
		return (const uint8_t*)pp->staging + (size_t)row0 * line;
	});
	// (bands of stripes that were not looked at may still be on their way: the staging buffer is this scene's until they are in)
	const hipError_t se = hipStreamSynchronize(s->stream);
	if (e) return e;
	if (se != hipSuccess) return fail(MIRT_E_DEVICE, "hipStreamSynchronize: %s", hipGetErrorString(se));
	if (stats) return read_stats(s, s->stream, stats);
	return MIRT_OK;
}

int mirt_assemble_tiles(int device, const mirt_render_params* p, uint32_t n_ranks, const void* d_tiles, size_t per_rank_bytes, void* d_bgr, void* hip_stream) {
	if (!p || !d_tiles || !d_bgr || n_ranks == 0) return fail(MIRT_E_INVALID, "NULL or zero argument");
	const uint32_t tw = p->tile_w ? p->tile_w : 32, th = p->tile_h ? p->tile_h : 32;
	if ((tw & 7) || (th & 7)) return fail(MIRT_E_INVALID, "tile size must be multiples of 8");
	mirt_render_params q = *p;
	q.first_tile = 0;
	q.tile_stride = n_ranks;
	const TileGeom g = tile_geom(&q, tw, th);
	if (per_rank_bytes < (size_t)g.nSlots * 3) return fail(MIRT_E_INVALID, "per_rank_bytes %zu smaller than a rank's tile buffer %zu", per_rank_bytes, (size_t)g.nSlots * 3);
	HIPCHK(hipSetDevice(device));
	const uint64_t n = (uint64_t)p->width * p->height;
	if ((tw * 3u) % 16u == 0 && (p->width * 3ull) % 16u == 0 && per_rank_bytes % 16u == 0 && ((uintptr_t)d_tiles | (uintptr_t)d_bgr) % 16u == 0) {
		const uint64_t nTiles = (uint64_t)g.tilesX * g.tilesY, pieces = nTiles * th * (tw * 3u / 16u);
		const int grid16 = (int)((pieces + BLOCK - 1) / BLOCK < 16384 ? (pieces + BLOCK - 1) / BLOCK : 16384);
		hipLaunchKernelGGL(k_assemble16, dim3(grid16), dim3(BLOCK), 0, (hipStream_t)hip_stream, p->width, p->height, tw, th, g.tilesX, (uint32_t)nTiles, n_ranks,
		                   (const uint8_t*)d_tiles, per_rank_bytes, (uint8_t*)d_bgr);
		HIPCHK(hipGetLastError());
		return MIRT_OK;
	}
	const int grid = (int)((n + BLOCK - 1) / BLOCK < 8192 ? (n + BLOCK - 1) / BLOCK : 8192);
	hipLaunchKernelGGL(k_assemble, dim3(grid), dim3(BLOCK), 0, (hipStream_t)hip_stream, p->width, p->height, tw, th, g.tilesX, n_ranks,
	                   (const uint8_t*)d_tiles, per_rank_bytes, (uint8_t*)d_bgr);
	HIPCHK(hipGetLastError());
	return MIRT_OK;
}

int mirt_trace_rays(mirt_scene* s, uint32_t trace_mode, uint64_t n, const float* origins, const float* dirs, int32_t* prim_id_out, float* dist_out, float* normal_out) {
	if (!s) return fail(MIRT_E_INVALID, "scene handle is NULL");
	if (trace_mode > MIRT_TRACE_REFWALK) return fail(MIRT_E_INVALID, "trace_mode %u unknown", trace_mode);
	if (int e = refwalk_ok(s, trace_mode)) return e;
	if (n == 0) return MIRT_OK;
	if (!origins || !dirs || !prim_id_out || !dist_out) return fail(MIRT_E_INVALID, "NULL ray or result array");
	HIPCHK(hipSetDevice(s->device));
	DevBuf o, d, id, dist, nrm;
	int e = 0;
	if (!e) e = upload(o, origins, n * 12);
	if (!e) e = upload(d, dirs, n * 12);
	if (!e) e = id.ensure(n * 4);
	if (!e) e = dist.ensure(n * 4);
	if (!e && normal_out) e = nrm.ensure(n * 12);
	if (!e) {
		const int grid = (int)((n + BLOCK - 1) / BLOCK < (uint64_t)s->numCUs * GRID_BLOCKS_PER_CU ? (n + BLOCK - 1) / BLOCK : (uint64_t)s->numCUs * GRID_BLOCKS_PER_CU);
		const int mode = effective_mode(s, trace_mode);
		// (cull_bound(): these rays start wherever the caller says)
		DScene dsl = s->ds;
		{
			float far = 0.0f, shortest = 1.0f;
			for (uint64_t i = 0; i < n; ++i) {
				far = fmaxf(far, (fabsf(origins[i * 3]) + fabsf(origins[i * 3 + 1])) + fabsf(origins[i * 3 + 2]));
				const float len = sqrtf(dirs[i * 3] * dirs[i * 3] + dirs[i * 3 + 1] * dirs[i * 3 + 1] + dirs[i * 3 + 2] * dirs[i * 3 + 2]);
				if (len < shortest) shortest = len; // (a NaN direction never walks)
			}
			const float c = 1e-5f + far * 0x1p-17f * 1.01f / shortest;
			dsl.cullSlack = std::isfinite(c) ? fmaxf(dsl.cullSlack, c) : __builtin_inff();
		}
#define LAUNCH_LIST(M, O) hipLaunchKernelGGL((k_trace_list<M, O>), dim3(grid), dim3(BLOCK), 0, s->stream, dsl, (uint32_t*)s->hookOvf.p, s->ovfStride, (MODE_FAST != M && s->wideDepth + 2 <= STACK_MAX) ? 1u : 0u, n, (const float*)o.p, (const float*)d.p, (int32_t*)id.p, (float*)dist.p, (float*)nrm.p)
		if (mode == MODE_REFWALK) LAUNCH_LIST(MODE_REFWALK, false);
		else if (mode == MODE_FAST) { if (s->deepBvh) LAUNCH_LIST(MODE_FAST, true); else LAUNCH_LIST(MODE_FAST, false); }
		else { if (s->deepBvh) LAUNCH_LIST(MODE_EXACT, true); else LAUNCH_LIST(MODE_EXACT, false); }
#undef LAUNCH_LIST
		hipError_t he = hipGetLastError();
		if (he == hipSuccess) he = hipStreamSynchronize(s->stream);
		if (he == hipSuccess) he = hipMemcpy(prim_id_out, id.p, n * 4, hipMemcpyDeviceToHost);
		if (he == hipSuccess) he = hipMemcpy(dist_out, dist.p, n * 4, hipMemcpyDeviceToHost);
		if (he == hipSuccess && normal_out) he = hipMemcpy(normal_out, nrm.p, n * 12, hipMemcpyDeviceToHost);
		if (he != hipSuccess) e = fail(MIRT_E_DEVICE, "mirt_trace_rays: %s", hipGetErrorString(he));
	}
	o.release(); d.release(); id.release(); dist.release(); nrm.release();
	return e;
}

int mirt_test_powf(int device, uint64_t n, const float* x, const float* y, float* out) {
	if (n == 0) return MIRT_OK;
	if (!x || !y || !out) return fail(MIRT_E_INVALID, "NULL array");
	HIPCHK(hipSetDevice(device));
	DevBuf dx, dy, dout;
	int e = upload(dx, x, n * 4);
	if (!e) e = upload(dy, y, n * 4);
	if (!e) e = dout.ensure(n * 4);
	if (!e) {
		const int grid = (int)((n + BLOCK - 1) / BLOCK < 4096 ? (n + BLOCK - 1) / BLOCK : 4096);
		hipLaunchKernelGGL(k_test_powf, dim3(grid), dim3(BLOCK), 0, nullptr, n, (const float*)dx.p, (const float*)dy.p, (float*)dout.p);
		hipError_t he = hipGetLastError();
		if (he == hipSuccess) he = hipDeviceSynchronize();
		if (he == hipSuccess) he = hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost);
		if (he != hipSuccess) e = fail(MIRT_E_DEVICE, "mirt_test_powf: %s", hipGetErrorString(he));
	}
	dx.release(); dy.release(); dout.release();
	return e;
}

int mirt_trace_primary_ids(mirt_scene* s, uint32_t trace_mode, uint32_t width, uint32_t height, int32_t* prim_id_out, float* dist_out) {
	if (!s) return fail(MIRT_E_INVALID, "scene handle is NULL");
	if (trace_mode > MIRT_TRACE_REFWALK) return fail(MIRT_E_INVALID, "trace_mode %u unknown", trace_mode);
	if (int e = refwalk_ok(s, trace_mode)) return e;
	if (!prim_id_out || !dist_out || width == 0 || height == 0) return fail(MIRT_E_INVALID, "NULL result array or empty image");
	HIPCHK(hipSetDevice(s->device));
	const uint64_t n = (uint64_t)width * height;
	DevBuf id, dist;
	int e = id.ensure(n * 4);
	if (!e) e = dist.ensure(n * 4);
	if (!e) {
		const int grid = (int)((n + BLOCK - 1) / BLOCK < (uint64_t)s->numCUs * GRID_BLOCKS_PER_CU ? (n + BLOCK - 1) / BLOCK : (uint64_t)s->numCUs * GRID_BLOCKS_PER_CU);
		const int mode = effective_mode(s, trace_mode);
#define LAUNCH_IDS(M, O) hipLaunchKernelGGL((k_primary_ids<M, O>), dim3(grid), dim3(BLOCK), 0, s->stream, s->ds, (uint32_t*)s->hookOvf.p, s->ovfStride, width, height, (int32_t*)id.p, (float*)dist.p)
		if (mode == MODE_REFWALK) LAUNCH_IDS(MODE_REFWALK, false);
		else if (mode == MODE_FAST) { if (s->deepBvh) LAUNCH_IDS(MODE_FAST, true); else LAUNCH_IDS(MODE_FAST, false); }
		else { if (s->deepBvh) LAUNCH_IDS(MODE_EXACT, true); else LAUNCH_IDS(MODE_EXACT, false); }
#undef LAUNCH_IDS
		hipError_t he = hipGetLastError();
		if (he == hipSuccess) he = hipStreamSynchronize(s->stream);
		if (he == hipSuccess) he = hipMemcpy(prim_id_out, id.p, n * 4, hipMemcpyDeviceToHost);
		if (he == hipSuccess) he = hipMemcpy(dist_out, dist.p, n * 4, hipMemcpyDeviceToHost);
		if (he != hipSuccess) e = fail(MIRT_E_DEVICE, "mirt_trace_primary_ids: %s", hipGetErrorString(he));
	}
	id.release(); dist.release();
	return e;
}

}  // extern "C"
