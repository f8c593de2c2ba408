// Host-side builder of the device acceleration structure: a binned-SAH bounding volume hierarchy over the scene's
// PRIMITIVES (each primitive exactly once), used as a conservative filter in front of the reference's own tests.
//
// Why this is the same function as the reference's tree walk.  Partition::intersect (Partition.cpp:35-53) tests a
// primitive A exactly when A sits in a leaf whose box, and all ancestor boxes, pass AABB::intersect.  Child boxes are
// halves of their parent's box (AABB::splitLeft/Right, AABB.cpp:121-155), and AABB::intersect's t-values are monotone in
// the box bounds (IEEE subtraction and division are monotone), so "leaf box hit" implies "every ancestor hit": A is
// tested  <=>  some leaf holding A has its own box hit.  The result of the walk is therefore
//     argmin over { A : A->intersect(ray) valid  and  exists leaf L containing A with L.box hit }  of (dist, first visit)
// and any structure that enumerates a superset of those A -- here: primitives whose padded bounds the ray crosses --
// followed by the exact "is A in a hit leaf" check gives the identical answer.  mirt_scene_create verifies the nesting
// premise on the tree it is handed and falls back to walking that tree as it is when the premise does not hold.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

struct BvhPrim {
	float lo[3], hi[3];
	uint32_t ref; // caller's handle
	uint32_t kind; // 0 triangle, 1 sphere: leaves are homogeneous
};

struct BvhNode {
	float lo[3], hi[3];
	int32_t left, right; // children, or -1/-1 for a leaf
	uint32_t first, count; // leaf: range in the ordered primitive list
	uint32_t kind;
};

struct BvhBuilder {
	static const int BINS = 16;
#ifndef MIRT_LEAF_MIN
#define MIRT_LEAF_MIN 1
#endif
#ifndef MIRT_COST_PRIM
#define MIRT_COST_PRIM 4.0f
#endif
	uint32_t LEAF_MIN = MIRT_LEAF_MIN;  // ranges this small always become a leaf
	static const uint32_t LEAF_CAP = 8; // a leaf reference holds a 3-bit count
	float COST_NODE = 1.0f, COST_PRIM = MIRT_COST_PRIM;
	static const int SAH_DEPTH = 28; // below this depth splits are balanced medians, which bounds the total depth

	std::vector<BvhPrim> prims; // reordered in place: leaf ranges are contiguous
	std::vector<BvhNode> nodes;
	uint32_t maxDepth = 0;

	// Parallel build: the recursion runs on one thread down to ranges of at most CUT primitives; those ranges are
	// independent (each works inside its own part of `prims`), so they are built on several threads into node arrays of
	// their own and spliced in afterwards.  CUT does not depend on the thread count: the tree is the same for any.
	static const uint32_t CUT = 1u << 16;
	struct Task { uint32_t start, end, depth; };
	std::vector<Task> tasks;
	bool cutting = false;
	BvhPrim* P = nullptr; // the array build() works on

	static float area(const float* lo, const float* hi) {
		const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
		return 2.0f * (dx * dy + dy * dz + dz * dx);
	}
	static void grow(float* lo, float* hi, const float* plo, const float* phi) {
		for (int k = 0; k < 3; ++k) {
			lo[k] = plo[k] < lo[k] ? plo[k] : lo[k];
			hi[k] = phi[k] > hi[k] ? phi[k] : hi[k];
		}
	}

	int32_t build(uint32_t start, uint32_t end, uint32_t depth) {
		if (cutting && depth > 0 && end - start <= CUT) {
			tasks.push_back(Task{start, end, depth});
			return -2 - (int32_t)(tasks.size() - 1); // placeholder, patched by run()
		}
		const int32_t me = (int32_t)nodes.size();
		nodes.push_back(BvhNode());
		if (depth > maxDepth) maxDepth = depth;
		float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
		float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
		bool mixed = false;
		for (uint32_t i = start; i < end; ++i) {
			grow(lo, hi, P[i].lo, P[i].hi);
			float c[3];
			for (int k = 0; k < 3; ++k) c[k] = 0.5f * (P[i].lo[k] + P[i].hi[k]);
			grow(clo, chi, c, c);
			mixed |= P[i].kind != P[start].kind;
		}
		BvhNode n;
		memcpy(n.lo, lo, 12);
		memcpy(n.hi, hi, 12);
		n.left = n.right = -1;
		n.first = start;
		n.count = end - start;
		n.kind = P[start].kind;
		const uint32_t cnt = end - start;
		if (cnt <= LEAF_MIN && !mixed) { nodes[me] = n; return me; }

		uint32_t mid = start;
		if (mixed && cnt <= LEAF_MIN) {
			// small mixed range: split by kind so that every leaf holds one kind of primitive
			mid = (uint32_t)(std::partition(P + start, P + end, [](const BvhPrim& p) { return p.kind == 0; }) - P);
		}
		else {
			// binned SAH over all three axes; the split must beat the cost of testing the range as one leaf
			int axis = 0;
			float ext = chi[0] - clo[0];
			for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > ext) { ext = chi[k] - clo[k]; axis = k; }
			bool done = false;
			if (ext > 0.0f && depth < (uint32_t)SAH_DEPTH) {
				float bestCost = INFINITY, bestScale = 0.0f;
				int bestBin = -1, bestAxis = -1;
				for (int ax = 0; ax < 3; ++ax) {
					const float e = chi[ax] - clo[ax];
					if (!(e > 0.0f)) continue;
					uint32_t bc[BINS] = {0};
					float blo[BINS][3], bhi[BINS][3];
					for (int b = 0; b < BINS; ++b) for (int k = 0; k < 3; ++k) { blo[b][k] = INFINITY; bhi[b][k] = -INFINITY; }
					const float scale = (float)BINS / e;
					for (uint32_t i = start; i < end; ++i) {
						const float c = 0.5f * (P[i].lo[ax] + P[i].hi[ax]);
						int b = (int)((c - clo[ax]) * scale);
						b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
						bc[b]++;
						grow(blo[b], bhi[b], P[i].lo, P[i].hi);
					}
					float rightArea[BINS];
					uint32_t rightCnt[BINS];
					float rlo[3] = {INFINITY, INFINITY, INFINITY}, rhi[3] = {-INFINITY, -INFINITY, -INFINITY};
					uint32_t rc = 0;
					for (int b = BINS - 1; b > 0; --b) {
						if (bc[b]) grow(rlo, rhi, blo[b], bhi[b]);
						rc += bc[b];
						rightArea[b] = rc ? area(rlo, rhi) : 0.0f;
						rightCnt[b] = rc;
					}
					float llo[3] = {INFINITY, INFINITY, INFINITY}, lhi[3] = {-INFINITY, -INFINITY, -INFINITY};
					uint32_t lc = 0;
					for (int b = 0; b < BINS - 1; ++b) {
						if (bc[b]) grow(llo, lhi, blo[b], bhi[b]);
						lc += bc[b];
						if (lc == 0 || rightCnt[b + 1] == 0) continue;
						const float cost = area(llo, lhi) * (float)lc + rightArea[b + 1] * (float)rightCnt[b + 1];
						if (cost < bestCost) { bestCost = cost; bestBin = b; bestAxis = ax; bestScale = scale; }
					}
				}
				if (bestBin >= 0) {
					// leaf cost n*Ci against split cost Ct + (children cost)/area(node); Ci/Ct from the kernel's instruction counts
					const float nodeArea = area(lo, hi);
					const float splitCost = COST_NODE + COST_PRIM * bestCost / (nodeArea > 0.0f ? nodeArea : 1.0f);
					if (cnt <= LEAF_CAP && !mixed && (float)cnt * COST_PRIM <= splitCost) { nodes[me] = n; return me; }
					const int ax = bestAxis;
					const float ca = clo[ax], scale = bestScale;
					const int bb = bestBin;
					mid = (uint32_t)(std::partition(P + start, P + end, [=](const BvhPrim& p) {
						const float c = 0.5f * (p.lo[ax] + p.hi[ax]);
						int b = (int)((c - ca) * scale);
						b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
						return b <= bb;
					}) - P);
					done = mid > start && mid < end;
				}
			}
			if (!done) {
				// balanced median split (also the fallback for coincident centroids)
				mid = start + cnt / 2;
				std::nth_element(P + start, P + mid, P + end, [=](const BvhPrim& a, const BvhPrim& b) {
					return (a.lo[axis] + a.hi[axis]) < (b.lo[axis] + b.hi[axis]);
				});
			}
		}
		const int32_t l = build(start, mid, depth + 1);
		const int32_t r = build(mid, end, depth + 1);
		n.left = l;
		n.right = r;
		nodes[me] = n;
		return me;
	}

	void run() {
		// tuning aid: MIRT_BVH_LEAF_MIN / MIRT_BVH_COST_PRIM override the built-in leaf parameters
		if (const char* e = getenv("MIRT_BVH_LEAF_MIN")) { const int v = atoi(e); if (v >= 1 && v <= (int)LEAF_CAP) LEAF_MIN = (uint32_t)v; }
		if (const char* e = getenv("MIRT_BVH_COST_PRIM")) { const float v = (float)atof(e); if (v > 0.0f) COST_PRIM = v; }
		nodes.clear();
		tasks.clear();
		maxDepth = 0;
		if (prims.empty()) return;
		nodes.reserve(prims.size() / 2);
		cutting = prims.size() > 4u * CUT;
		P = prims.data();
		build(0, (uint32_t)prims.size(), 0);
		cutting = false;
		if (tasks.empty()) return;
		unsigned nThreads = std::thread::hardware_concurrency();
		if (nThreads > 16) nThreads = 16;
		if (nThreads < 1) nThreads = 1;
		if (nThreads > tasks.size()) nThreads = (unsigned)tasks.size();
		std::vector<std::vector<BvhNode>> sub(tasks.size());
		std::vector<uint32_t> subDepth(tasks.size(), 0);
		std::atomic<size_t> next(0);
		auto work = [&]() {
			BvhBuilder local; // shares nothing but the primitives of its tasks' ranges
			local.LEAF_MIN = LEAF_MIN;
			local.COST_PRIM = COST_PRIM;
			for (;;) {
				const size_t t = next.fetch_add(1);
				if (t >= tasks.size()) break;
				local.nodes.clear();
				local.maxDepth = 0;
				local.P = prims.data(); // ranges of different tasks are disjoint
				local.build(tasks[t].start, tasks[t].end, tasks[t].depth);
				sub[t].swap(local.nodes);
				subDepth[t] = local.maxDepth;
			}
		};
		std::vector<std::thread> th;
		for (unsigned i = 1; i < nThreads; ++i) th.emplace_back(work);
		work();
		for (auto& t : th) t.join();
		// splice: subtree t's nodes go to the end, its root replaces placeholder -2 - t
		std::vector<int32_t> rootOf(tasks.size());
		for (size_t t = 0; t < tasks.size(); ++t) {
			const int32_t off = (int32_t)nodes.size();
			rootOf[t] = off;
			for (BvhNode n : sub[t]) {
				if (n.left >= 0) { n.left += off; n.right += off; }
				nodes.push_back(n);
			}
			std::vector<BvhNode>().swap(sub[t]);
			if (subDepth[t] > maxDepth) maxDepth = subDepth[t];
		}
		for (BvhNode& n : nodes) {
			if (n.left <= -2) n.left = rootOf[(size_t)(-2 - n.left)];
			if (n.right <= -2) n.right = rootOf[(size_t)(-2 - n.right)];
		}
	}
};
