// Host-only check of the hierarchy builder (raytracer_amd/csrc/mirt_bvh.h): builds a hierarchy over seeded random boxes,
// checks its invariants and prints a digest of nodes and primitive order.  The test runs it with one thread and with
// several and compares the digests: the tree must not depend on the thread count.
//   bvh_check <n_prims> <seed>
#include <stdio.h>

#include "../../raytracer_amd/csrc/mirt_bvh.h"

static uint64_t fnv(uint64_t h, const void* p, size_t n) {
	const unsigned char* b = (const unsigned char*)p;
	for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
	return h;
}

int main(int argc, char** argv) {
	const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 100000u;
	uint64_t st = argc > 2 ? (uint64_t)atoll(argv[2]) : 1u;
	auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 40) & 0xffffff) / 16777216.0f; };
	BvhBuilder bb;
	bb.prims.resize(n);
	for (uint32_t i = 0; i < n; ++i) {
		BvhPrim p;
		// clustered: a few blobs, so that SAH splits are uneven; every 97th primitive is a sphere-kind box
		const float cx = floorf(rnd() * 8.0f) * 10.0f, cy = floorf(rnd() * 4.0f) * 7.0f;
		for (int k = 0; k < 3; ++k) {
			const float c = (k == 0 ? cx : (k == 1 ? cy : 0.0f)) + rnd() * 6.0f, h = 0.001f + rnd() * 0.05f;
			p.lo[k] = c - h;
			p.hi[k] = c + h;
		}
		p.ref = i;
		p.kind = i % 97u == 0u ? 1u : 0u;
		bb.prims[i] = p;
	}
	bb.run();
	// invariants: every primitive in exactly one leaf, leaves homogeneous and at most LEAF_CAP long, boxes enclose
	std::vector<uint8_t> seen(n, 0);
	size_t leaves = 0;
	for (size_t i = 0; i < bb.nodes.size(); ++i) {
		const BvhNode& nd = bb.nodes[i];
		if (nd.left < 0) {
			++leaves;
			if (nd.count < 1 || nd.count > BvhBuilder::LEAF_CAP) { printf("FAIL leaf %zu count %u\n", i, nd.count); return 1; }
			for (uint32_t k = 0; k < nd.count; ++k) {
				const BvhPrim& p = bb.prims[nd.first + k];
				if (p.kind != nd.kind) { printf("FAIL leaf %zu mixes kinds\n", i); return 1; }
				if (seen[p.ref]++) { printf("FAIL primitive %u twice\n", p.ref); return 1; }
				for (int c = 0; c < 3; ++c) if (p.lo[c] < nd.lo[c] || p.hi[c] > nd.hi[c]) { printf("FAIL leaf %zu does not enclose\n", i); return 1; }
			}
		}
		else {
			if (nd.left <= (int32_t)i || nd.right <= (int32_t)i || (size_t)nd.left >= bb.nodes.size() || (size_t)nd.right >= bb.nodes.size()) { printf("FAIL node %zu children %d %d\n", i, nd.left, nd.right); return 1; }
			const BvhNode& a = bb.nodes[nd.left];
			const BvhNode& b = bb.nodes[nd.right];
			for (int c = 0; c < 3; ++c) if (a.lo[c] < nd.lo[c] || a.hi[c] > nd.hi[c] || b.lo[c] < nd.lo[c] || b.hi[c] > nd.hi[c]) { printf("FAIL node %zu does not enclose its children\n", i); return 1; }
			if (a.count + b.count != nd.count || a.first != nd.first || b.first != nd.first + a.count) { printf("FAIL node %zu ranges\n", i); return 1; }
		}
	}
	for (uint32_t i = 0; i < n; ++i) if (seen[i] != 1) { printf("FAIL primitive %u in %d leaves\n", i, (int)seen[i]); return 1; }
	uint64_t h = 1469598103934665603ull;
	for (const BvhNode& nd : bb.nodes) { h = fnv(h, nd.lo, 24); h = fnv(h, &nd.left, 8); h = fnv(h, &nd.first, 12); }
	for (const BvhPrim& p : bb.prims) h = fnv(h, &p.ref, 4);
	printf("OK nodes %zu leaves %zu depth %u threads %u digest %016llx\n", bb.nodes.size(), leaves, bb.maxDepth, MirtPool::get().threads(), (unsigned long long)h);
	return 0;
}
