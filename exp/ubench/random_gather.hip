// Microbenchmark: how many bytes per second does the memory system deliver to INDEPENDENT random reads of 64-byte records,
// as a function of the size of the table they are spread over?  This is the ceiling of a hierarchy walk whose working set does
// not fit the caches (the 10-million-triangle scene: nodes and triangle records of 64 bytes all over 3.5 GB), before any
// dependence between a ray's steps is counted: here every lane has DEPTH requests in flight and none depends on another.
//   mode 0: a lane fetches its own 64-byte record with four 16-byte loads (the walk's node fetch)
//   mode 1: four neighbouring lanes fetch one record, 16 bytes each (a quarter of the requests per byte)
//   mode 2: a lane fetches 16 bytes of its own record only (the cost of a request: 64 bytes move, 16 are used)
//   mode 3: as mode 0 with 128-byte records (eight loads): a full cache line per request
//   mode 4: as mode 0, but every second record is the other half of the line of the record before it, fetched one step LATER
//           (does a miss on half a line bring the whole line?)
// usage: random_gather [table MiB ...]      prints GB/s of record bytes asked for, per table size and mode
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ table, uint64_t nRecords, int steps, uint32_t* out) {
	const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
	uint64_t x = ((uint64_t)((MODE == 1) ? (tid >> 2) : tid) + 1) * 0x9e3779b97f4a7c15ull;
	uint32_t acc = 0;
	uint64_t prev[2] = {0, 1};
	for (int s = 0; s < steps; ++s) {
		// four records per step, indices from a generator that does not look at the data
		uint4 v[4];
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
			uint64_t r = __umul64hi(x * 0x2545f4914f6cdd1dull, nRecords); // uniform over the table, no division
			if (MODE == 4) {
				if (q & 1) r = prev[q >> 1] ^ 1; // the line-mate of a record of the step before
				else prev[q >> 1] = r;
			}
			if (MODE == 0 || MODE == 4) {
				const uint4* p = table + r * 4;
				const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
				v[q] = make_uint4(a.x, b.y, c.z, d.w);
			}
			else if (MODE == 1) v[q] = table[r * 4 + (threadIdx.x & 3)];
			else if (MODE == 2) v[q] = table[r * 4];
			else {
				const uint4* p = table + (r >> 1) * 8;
				const uint4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], h = p[7];
				v[q] = make_uint4(a.x ^ e.x, b.y ^ f.y, c.z ^ g.z, d.w ^ h.w);
			}
		}
		acc += v[0].x ^ v[1].y ^ v[2].z ^ v[3].w;
	}
	out[tid] = acc;
}

int main(int argc, char** argv) {
	std::vector<size_t> sizes;
	for (int a = 1; a < argc; ++a) sizes.push_back((size_t)atol(argv[a]));
	if (sizes.empty()) sizes = {16, 64, 192, 512, 1024, 3584};
	const size_t maxBytes = 3584ull << 20;
	uint4* table;
	uint32_t* out;
	const int blocks = 256 * 8, steps = 400;
	CHECK(hipMalloc(&table, maxBytes));
	CHECK(hipMemset(table, 1, maxBytes));
	CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	printf("%10s %12s %12s %12s %12s %12s   (GB/s of record bytes asked for)\n", "table MiB", "64 B own", "64 B quad", "16 of 64 B", "128 B own", "64 B + mate");
	for (size_t mib : sizes) {
		if ((mib << 20) > maxBytes) continue;
		const uint64_t nRec = (mib << 20) / 64;
		double gbs[5];
		for (int mode = 0; mode < 5; ++mode) {
			float best = 1e30f;
			for (int rep = 0; rep < 3; ++rep) {
				CHECK(hipEventRecord(e0));
				if (mode == 0) k<0><<<blocks, 256>>>(table, nRec, steps, out);
				else if (mode == 1) k<1><<<blocks, 256>>>(table, nRec, steps, out);
				else if (mode == 2) k<2><<<blocks, 256>>>(table, nRec, steps, out);
				else if (mode == 3) k<3><<<blocks, 256>>>(table, nRec, steps, out);
				else k<4><<<blocks, 256>>>(table, nRec, steps, out);
				CHECK(hipEventRecord(e1));
				CHECK(hipEventSynchronize(e1));
				float ms;
				CHECK(hipEventElapsedTime(&ms, e0, e1));
				if (ms < best) best = ms;
			}
			// bytes asked for: per lane and step 4 records of 64 B (mode 1: a quarter record per lane; mode 2: 16 B used, 64 moved; mode 3: 128 B)
			const double perLane = mode == 1 ? 64.0 : mode == 3 ? 512.0 : 256.0;
			gbs[mode] = (double)blocks * 256 * steps * perLane / (best * 1e-3) / 1e9;
		}
		printf("%10zu %12.0f %12.0f %12.0f %12.0f %12.0f\n", mib, gbs[0], gbs[1], gbs[2], gbs[3], gbs[4]);
		fflush(stdout);
	}
	return 0;
}
