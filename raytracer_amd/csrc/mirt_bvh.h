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

#include "mirt_pool.h"

struct BvhPrim {
	BvhPrim() {} // no initialisation: arrays of millions are resized and then filled by several threads
	float lo[3], hi[3];
	uint32_t ref; // caller's handle
	uint32_t kind; // 0 triangle, 1 sphere: leaves are homogeneous
};

struct BvhNode {
	BvhNode() {} // (as BvhPrim)
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

	// Parallel build.  Ranges of more than CUT primitives (the top of the tree: few nodes, each over a large part of the
	// array) are split by passes that run on all threads -- bounds, binning and a stable partition, each over fixed chunks
	// of GRAIN primitives whose partial results are combined in chunk order.  Ranges of at most CUT primitives are
	// independent (each works inside its own part of `prims`): they are built whole, one per thread at a time, into node
	// arrays of their own and spliced in afterwards.  Neither CUT nor GRAIN depends on the thread count: the tree is the
	// same for any.
	static const uint32_t CUT = 1u << 16;
	static const uint32_t GRAIN = 1u << 14;
	struct Task { uint32_t start, end, depth; };
	std::vector<Task> tasks;
	bool cutting = false;
	BvhPrim* P = nullptr; // the array build() works on
	BvhPrim* tmp = nullptr; // as large as `prims`: the partition of a large range goes through it

	struct Bins {
		uint32_t bc[3][BINS];
		float blo[3][BINS][3], bhi[3][BINS][3];
		void clear() {
			for (int a = 0; a < 3; ++a) for (int b = 0; b < BINS; ++b) { bc[a][b] = 0; for (int k = 0; k < 3; ++k) { blo[a][b][k] = INFINITY; bhi[a][b][k] = -INFINITY; } }
		}
	};
	static int bin_of(const BvhPrim& p, int ax, float clo, float scale) {
		const float c = 0.5f * (p.lo[ax] + p.hi[ax]);
		const int b = (int)((c - clo) * scale);
		return b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
	}
	// stable partition of [start, end) by pred, chunk by chunk through `tmp`; returns the first index of the second part
	template <class Pred>
	uint32_t partition_large(uint32_t start, uint32_t end, Pred pred) {
		const uint32_t nChunks = (end - start + GRAIN - 1) / GRAIN;
		std::vector<uint32_t> nTrue(nChunks + 1, 0);
		MirtPool::get().run(nChunks, [&](uint32_t c) {
			const uint32_t a = start + c * GRAIN, b = a + GRAIN < end ? a + GRAIN : end;
			uint32_t n = 0;
			for (uint32_t i = a; i < b; ++i) n += pred(P[i]) ? 1u : 0u;
			nTrue[c + 1] = n;
		});
		for (uint32_t c = 0; c < nChunks; ++c) nTrue[c + 1] += nTrue[c];
		const uint32_t mid = start + nTrue[nChunks];
		MirtPool::get().run(nChunks, [&](uint32_t c) {
			const uint32_t a = start + c * GRAIN, b = a + GRAIN < end ? a + GRAIN : end;
			uint32_t t = start + nTrue[c], f = mid + (a - start) - nTrue[c];
			for (uint32_t i = a; i < b; ++i) {
				if (pred(P[i])) tmp[t++] = P[i];
				else tmp[f++] = P[i];
			}
		});
		MirtPool::get().run(nChunks, [&](uint32_t c) {
			const uint32_t a = start + c * GRAIN, b = a + GRAIN < end ? a + GRAIN : end;
			memcpy(P + a, tmp + a, (size_t)(b - a) * sizeof(BvhPrim));
		});
		return mid;
	}

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
		const bool large = tmp && end - start > CUT;
		const uint32_t nChunks = large ? (end - start + GRAIN - 1) / GRAIN : 1u;
		if (large) {
			struct Part { float lo[3], hi[3], clo[3], chi[3]; bool mixed; };
			std::vector<Part> part(nChunks);
			const uint32_t kind0 = P[start].kind;
			MirtPool::get().run(nChunks, [&](uint32_t ch) {
				const uint32_t a = start + ch * GRAIN, b = a + GRAIN < end ? a + GRAIN : end;
				Part q;
				for (int k = 0; k < 3; ++k) { q.lo[k] = q.clo[k] = INFINITY; q.hi[k] = q.chi[k] = -INFINITY; }
				q.mixed = false;
				for (uint32_t i = a; i < b; ++i) {
					grow(q.lo, q.hi, P[i].lo, P[i].hi);
					float c[3];
					for (int k = 0; k < 3; ++k) c[k] = 0.5f * (P[i].lo[k] + P[i].hi[k]);
					grow(q.clo, q.chi, c, c);
					q.mixed |= P[i].kind != kind0;
				}
				part[ch] = q;
			});
			for (const Part& q : part) { grow(lo, hi, q.lo, q.hi); grow(clo, chi, q.clo, q.chi); mixed |= q.mixed; }
		}
		else for (uint32_t i = start; i < end; ++i) {
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
				// a large range is binned along all three axes in one pass over its chunks
				Bins* all = nullptr;
				std::vector<Bins> partBins;
				Bins total;
				if (large) {
					partBins.resize(nChunks);
					float sc[3];
					for (int ax = 0; ax < 3; ++ax) sc[ax] = chi[ax] - clo[ax] > 0.0f ? (float)BINS / (chi[ax] - clo[ax]) : 0.0f;
					MirtPool::get().run(nChunks, [&](uint32_t ch) {
						const uint32_t a = start + ch * GRAIN, b = a + GRAIN < end ? a + GRAIN : end;
						Bins& q = partBins[ch];
						q.clear();
						for (uint32_t i = a; i < b; ++i) {
							for (int ax = 0; ax < 3; ++ax) {
								if (!(sc[ax] > 0.0f)) continue;
								const int bi = bin_of(P[i], ax, clo[ax], sc[ax]);
								q.bc[ax][bi]++;
								grow(q.blo[ax][bi], q.bhi[ax][bi], P[i].lo, P[i].hi);
							}
						}
					});
					total.clear();
					for (const Bins& q : partBins)
						for (int ax = 0; ax < 3; ++ax) for (int b = 0; b < BINS; ++b) {
							total.bc[ax][b] += q.bc[ax][b];
							grow(total.blo[ax][b], total.bhi[ax][b], q.blo[ax][b], q.bhi[ax][b]);
						}
					all = &total;
				}
				for (int ax = 0; ax < 3; ++ax) {
					const float e = chi[ax] - clo[ax];
					if (!(e > 0.0f)) continue;
					uint32_t bcLocal[BINS] = {0};
					float bloLocal[BINS][3], bhiLocal[BINS][3];
					const float scale = (float)BINS / e;
					if (!all) {
						for (int b = 0; b < BINS; ++b) for (int k = 0; k < 3; ++k) { bloLocal[b][k] = INFINITY; bhiLocal[b][k] = -INFINITY; }
						for (uint32_t i = start; i < end; ++i) {
							const int b = bin_of(P[i], ax, clo[ax], scale);
							bcLocal[b]++;
							grow(bloLocal[b], bhiLocal[b], P[i].lo, P[i].hi);
						}
					}
					const uint32_t* bc = all ? all->bc[ax] : bcLocal;
					const float (*blo)[3] = all ? all->blo[ax] : bloLocal;
					const float (*bhi)[3] = all ? all->bhi[ax] : bhiLocal;
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
					auto goesLeft = [=](const BvhPrim& p) { return bin_of(p, ax, ca, scale) <= bb; };
					mid = large ? partition_large(start, end, goesLeft) : (uint32_t)(std::partition(P + start, P + end, goesLeft) - P);
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
		if (prims.size() <= 4u * CUT) nodes.reserve(prims.size());
		cutting = prims.size() > 4u * CUT;
		P = prims.data();
		std::vector<BvhPrim> scratch;
		if (cutting) scratch.resize(prims.size());
		tmp = cutting ? scratch.data() : nullptr;
		build(0, (uint32_t)prims.size(), 0);
		cutting = false;
		tmp = nullptr;
		std::vector<BvhPrim>().swap(scratch);
		if (tasks.empty()) return;
		std::vector<std::vector<BvhNode>> sub(tasks.size());
		std::vector<uint32_t> subDepth(tasks.size(), 0);
		MirtPool::get().run((uint32_t)tasks.size(), [&](uint32_t t) {
			BvhBuilder local; // shares nothing but the primitives of its task's range (ranges of different tasks are disjoint)
			local.LEAF_MIN = LEAF_MIN;
			local.COST_PRIM = COST_PRIM;
			local.P = prims.data();
			local.nodes.reserve((size_t)(tasks[t].end - tasks[t].start));
			local.build(tasks[t].start, tasks[t].end, tasks[t].depth);
			sub[t].swap(local.nodes);
			subDepth[t] = local.maxDepth;
		});
		// splice: subtree t's nodes go to the end, its root replaces placeholder -2 - t
		std::vector<int32_t> rootOf(tasks.size());
		size_t total = nodes.size();
		for (size_t t = 0; t < tasks.size(); ++t) {
			rootOf[t] = (int32_t)total;
			total += sub[t].size();
			if (subDepth[t] > maxDepth) maxDepth = subDepth[t];
		}
		const size_t nTop = nodes.size();
		nodes.resize(total);
		for (size_t i = 0; i < nTop; ++i) {
			BvhNode& n = nodes[i];
			if (n.left <= -2) n.left = rootOf[(size_t)(-2 - n.left)];
			if (n.right <= -2) n.right = rootOf[(size_t)(-2 - n.right)];
		}
		MirtPool::get().run((uint32_t)tasks.size(), [&](uint32_t t) {
			const int32_t off = rootOf[t];
			BvhNode* dst = nodes.data() + off;
			for (size_t i = 0; i < sub[t].size(); ++i) {
				BvhNode n = sub[t][i];
				if (n.left >= 0) { n.left += off; n.right += off; }
				dst[i] = n;
			}
			std::vector<BvhNode>().swap(sub[t]);
		});
	}
};
