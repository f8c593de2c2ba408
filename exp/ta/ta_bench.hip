// Microbenchmark: what does the vector memory path charge for a wave's node fetch?
//   mode 0: every lane loads its own 64-byte node with four 16-byte loads (four diverging requests per lane)
//   mode 1: four loads, each fetching 16 nodes with four neighbouring lanes per node (each lane one quarter)
//   mode 2: as mode 0 but one 16-byte load per lane (a quarter of the bytes): the cost of a diverging request itself
// Node indices are pseudo-random over a table that fits in L2; every step's indices depend on the loaded data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ nodes, uint32_t mask, int steps, uint32_t* out) {
	const int lane = threadIdx.x & 63;
	uint32_t idx = ((blockIdx.x * 256 + threadIdx.x) >> ((MODE == 3 || MODE == 4) ? 2 : 0)) * 2654435761u; // modes 3, 4: one index per four lanes
	uint32_t acc = 0;
	for (int s = 0; s < steps; ++s) {
		idx &= mask;
		if (MODE == 0) {
			const uint4* p = nodes + (size_t)idx * 4;
			const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
			acc += a.x ^ b.y ^ c.z ^ d.w;
			idx = idx * 1664525u + 1013904223u;
		}
		else if (MODE == 2) {
			const uint4 a = nodes[(size_t)idx * 4];
			acc += a.x;
			idx = idx * 1664525u + 1013904223u;
		}
		else if (MODE == 3 || MODE == 4) {
			// quads of lanes share a node by construction (no exchange of indices): MODE 3 four loads, MODE 4 one load per step
			uint32_t t = 0;
#pragma unroll
			for (int kk = 0; kk < (MODE == 3 ? 4 : 1); ++kk) {
				const uint32_t other = (idx * 7919u + (uint32_t)kk * 0x9e3779b9u) & mask;
				const uint4 q = nodes[(size_t)other * 4 + (lane & 3)];
				t ^= q.x + q.w;
			}
			acc += t;
			idx = idx * 1664525u + 1013904223u;
		}
		else {
			uint32_t t = 0;
#pragma unroll
			for (int kk = 0; kk < 4; ++kk) {
				const uint32_t other = __shfl(idx, 16 * kk + (lane >> 2)); // the node of lane 16k + l/4
				const uint4 q = nodes[(size_t)other * 4 + (lane & 3)];
				t ^= q.x + q.w;
			}
			acc += t;
			idx = idx * 1664525u + 1013904223u;
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
	const uint32_t nNodes = argc > 1 ? (uint32_t)atoi(argv[1]) : (1u << 15); // x 64 B
	const int steps = 2000, blocks = 256 * 7;
	std::vector<uint4> h(nNodes * 4);
	for (size_t i = 0; i < h.size(); ++i) h[i] = make_uint4((uint32_t)(i * 2654435761u), (uint32_t)i, 7, (uint32_t)(i >> 3));
	uint4* d; uint32_t* o;
	CHECK(hipMalloc(&d, h.size() * 16));
	CHECK(hipMalloc(&o, blocks * 256 * 4));
	CHECK(hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	for (int mode = 0; mode < 5; ++mode) {
		float best = 1e9f;
		for (int rep = 0; rep < 3; ++rep) {
			CHECK(hipEventRecord(e0));
			if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, nNodes - 1, steps, o);
			else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, nNodes - 1, steps, o);
			else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, nNodes - 1, steps, o);
			else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, nNodes - 1, steps, o);
			else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, nNodes - 1, steps, o);
			CHECK(hipEventRecord(e1));
			CHECK(hipEventSynchronize(e1));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
			if (ms < best) best = ms;
		}
		const double waveSteps = (double)blocks * 4 * steps;
		printf("table %u KB, mode %d: %.3f ms, %.1f ns per wave step per CU-slot (28 waves per CU): %.2f wave steps per us per CU\n", nNodes / 16, mode, best,
		       best * 1e6 / (waveSteps / 256.0) , waveSteps / 256.0 / (best * 1e3));
	}
	return 0;
}
