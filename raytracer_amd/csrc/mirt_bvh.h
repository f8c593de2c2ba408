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

#include <immintrin.h>

#include <algorithm>
#include <chrono>
#include <stdio.h>
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
	static const int BINS = 32; // (Stanford dragon: 1.3 % fewer wave steps per frame than with 16; 64 bins, or trying every split position in
	                            // ranges of up to 256 primitives, take off another 0.1 %)
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

	// Boxes are handled four floats at a time (x, y, z and a lane that is not used).
	typedef __m128 V4;
	static V4 lane_mask() { return _mm_castsi128_ps(_mm_set_epi32(0, -1, -1, -1)); }
	static V4 lo_of(const BvhPrim& p) { return _mm_and_ps(_mm_loadu_ps(p.lo), lane_mask()); }
	static V4 hi_of(const BvhPrim& p) { return _mm_and_ps(_mm_loadu_ps(p.hi), lane_mask()); } // (the fourth float read is p.ref: masked off)
	static float area(V4 lo, V4 hi) {
		alignas(16) float d[4];
		_mm_store_ps(d, _mm_sub_ps(hi, lo));
		return 2.0f * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
	}
	static float area(const float* lo, const float* hi) {
		const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
		return 2.0f * (dx * dy + dy * dz + dz * dx);
	}
	struct Bins {
		uint32_t bc[3][BINS];
		V4 lo[3][BINS], hi[3][BINS];
		void clear() {
			const V4 pinf = _mm_set1_ps(INFINITY), ninf = _mm_set1_ps(-INFINITY);
			for (int a = 0; a < 3; ++a) for (int b = 0; b < BINS; ++b) { bc[a][b] = 0; lo[a][b] = pinf; hi[a][b] = ninf; }
		}
		void add(const Bins& q) {
			for (int a = 0; a < 3; ++a) for (int b = 0; b < BINS; ++b) {
				bc[a][b] += q.bc[a][b];
				lo[a][b] = _mm_min_ps(lo[a][b], q.lo[a][b]);
				hi[a][b] = _mm_max_ps(hi[a][b], q.hi[a][b]);
			}
		}
	};
	struct Extent {
		V4 lo, hi, clo, chi; // bounds of the boxes and of their centres
		bool mixed;
		void clear() { lo = clo = _mm_set1_ps(INFINITY); hi = chi = _mm_set1_ps(-INFINITY); mixed = false; }
	};
	static int bin_of(float c, float clo, float scale) {
		const int b = (int)((c - clo) * scale);
		return b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
	}
	static int bin_of(const BvhPrim& p, int ax, float clo, float scale) { return bin_of(0.5f * (p.lo[ax] + p.hi[ax]), clo, scale); }
	void extent_of(uint32_t a, uint32_t b, uint32_t kind0, Extent& e) const {
		const V4 half = _mm_set1_ps(0.5f);
		for (uint32_t i = a; i < b; ++i) {
			const V4 l = lo_of(P[i]), h = hi_of(P[i]);
			const V4 c = _mm_mul_ps(half, _mm_add_ps(l, h));
			e.lo = _mm_min_ps(e.lo, l);
			e.hi = _mm_max_ps(e.hi, h);
			e.clo = _mm_min_ps(e.clo, c);
			e.chi = _mm_max_ps(e.chi, c);
			e.mixed |= P[i].kind != kind0;
		}
	}
	// all three axes in one pass; sc[ax] = bins per unit length, or 0 for an axis along which the centres do not spread
	void bin_range(uint32_t a, uint32_t b, const float* clo, const float* sc, Bins& q) const {
		const V4 half = _mm_set1_ps(0.5f);
		for (uint32_t i = a; i < b; ++i) {
			const V4 l = lo_of(P[i]), h = hi_of(P[i]);
			alignas(16) float c[4];
			_mm_store_ps(c, _mm_mul_ps(half, _mm_add_ps(l, h)));
			for (int ax = 0; ax < 3; ++ax) {
				if (!(sc[ax] > 0.0f)) continue;
				const int bi = bin_of(c[ax], clo[ax], sc[ax]);
				q.bc[ax][bi]++;
				q.lo[ax][bi] = _mm_min_ps(q.lo[ax][bi], l);
				q.hi[ax][bi] = _mm_max_ps(q.hi[ax][bi], h);
			}
		}
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

	int32_t build(uint32_t start, uint32_t end, uint32_t depth) {
		if (cutting && depth > 0 && end - start <= CUT) {
			tasks.push_back(Task{start, end, depth});
			return -2 - (int32_t)(tasks.size() - 1); // placeholder, patched by run()
		}
		const int32_t me = (int32_t)nodes.size();
		nodes.push_back(BvhNode());
		if (depth > maxDepth) maxDepth = depth;
		const bool large = tmp && end - start > CUT;
		const uint32_t nChunks = large ? (end - start + GRAIN - 1) / GRAIN : 1u;
		const uint32_t kind0 = P[start].kind;
		Extent ex;
		ex.clear();
		if (large) {
			std::vector<Extent> part(nChunks);
			MirtPool::get().run(nChunks, [&](uint32_t ch) {
				const uint32_t a = start + ch * GRAIN, b = a + GRAIN < end ? a + GRAIN : end;
				Extent q;
				q.clear();
				extent_of(a, b, kind0, q);
				part[ch] = q;
			});
			for (const Extent& q : part) {
				ex.lo = _mm_min_ps(ex.lo, q.lo); ex.hi = _mm_max_ps(ex.hi, q.hi);
				ex.clo = _mm_min_ps(ex.clo, q.clo); ex.chi = _mm_max_ps(ex.chi, q.chi);
				ex.mixed |= q.mixed;
			}
		}
		else extent_of(start, end, kind0, ex);
		const bool mixed = ex.mixed;
		alignas(16) float lo[4], hi[4], clo[4], chi[4];
		_mm_store_ps(lo, ex.lo); _mm_store_ps(hi, ex.hi); _mm_store_ps(clo, ex.clo); _mm_store_ps(chi, ex.chi);
		BvhNode n;
		memcpy(n.lo, lo, 12);
		memcpy(n.hi, hi, 12);
		n.left = n.right = -1;
		n.first = start;
		n.count = end - start;
		n.kind = kind0;
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
				float sc[3];
				for (int ax = 0; ax < 3; ++ax) sc[ax] = chi[ax] - clo[ax] > 0.0f ? (float)BINS / (chi[ax] - clo[ax]) : 0.0f;
				Bins bins;
				bins.clear();
				if (large) {
					std::vector<Bins> part(nChunks);
					MirtPool::get().run(nChunks, [&](uint32_t ch) {
						const uint32_t a = start + ch * GRAIN, b = a + GRAIN < end ? a + GRAIN : end;
						part[ch].clear();
						bin_range(a, b, clo, sc, part[ch]);
					});
					for (const Bins& q : part) bins.add(q);
				}
				else bin_range(start, end, clo, sc, bins);
				float bestCost = INFINITY, bestScale = 0.0f;
				int bestBin = -1, bestAxis = -1;
				const V4 pinf = _mm_set1_ps(INFINITY), ninf = _mm_set1_ps(-INFINITY);
				for (int ax = 0; ax < 3; ++ax) {
					if (!(sc[ax] > 0.0f)) continue;
					const uint32_t* bc = bins.bc[ax];
					float rightArea[BINS];
					uint32_t rightCnt[BINS];
					V4 rlo = pinf, rhi = ninf;
					uint32_t rc = 0;
					for (int b = BINS - 1; b > 0; --b) {
						if (bc[b]) { rlo = _mm_min_ps(rlo, bins.lo[ax][b]); rhi = _mm_max_ps(rhi, bins.hi[ax][b]); }
						rc += bc[b];
						rightArea[b] = rc ? area(rlo, rhi) : 0.0f;
						rightCnt[b] = rc;
					}
					V4 llo = pinf, lhi = ninf;
					uint32_t lc = 0;
					for (int b = 0; b < BINS - 1; ++b) {
						if (bc[b]) { llo = _mm_min_ps(llo, bins.lo[ax][b]); lhi = _mm_max_ps(lhi, bins.hi[ax][b]); }
						lc += bc[b];
						if (lc == 0 || rightCnt[b + 1] == 0) continue;
						const float cost = area(llo, lhi) * (float)lc + rightArea[b + 1] * (float)rightCnt[b + 1];
						if (cost < bestCost) { bestCost = cost; bestBin = b; bestAxis = ax; bestScale = sc[ax]; }
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
		const auto t0 = std::chrono::steady_clock::now();
		build(0, (uint32_t)prims.size(), 0);
		const auto t1 = std::chrono::steady_clock::now();
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
		const auto t2 = std::chrono::steady_clock::now();
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
		if (getenv("MIRT_VERBOSE")) fprintf(stderr, "  bvh: top %.3f tasks %.3f (%zu) splice %.3f\n", std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(), tasks.size(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t2).count());
	}
};
