// Does a VALU instruction of a wave64 cost less when only some 16-lane groups have active lanes?
// Each wave runs a long chain of FMAs under `if (lane < K)` (or lanes spread with a stride); 8 waves per SIMD, all CUs busy.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <stdio.h>
__global__ void k(float* out, int K, int stride, int iters, int first = 0) {
	const int lane = threadIdx.x & 63;
	float a = lane * 0.001f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 0.125f;
	const bool on = stride ? ((lane % stride) == 0 && lane / stride < K) : (lane >= first && lane < first + K);
	if (on) {
		for (int i = 0; i < iters; ++i) {
			a = __builtin_fmaf(a, b, c); c = __builtin_fmaf(c, b, d); d = __builtin_fmaf(d, b, e); e = __builtin_fmaf(e, b, a);
			a = __builtin_fmaf(a, b, c); c = __builtin_fmaf(c, b, d); d = __builtin_fmaf(d, b, e); e = __builtin_fmaf(e, b, a);
		}
	}
	if (a + c + d + e == 12345.0f) out[threadIdx.x] = a;
}
int main() {
	float* out; hipMalloc(&out, 4096);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 20000, blocks = 256 * 8;
	struct { int K, stride; const char* what; int first; } cases[] = {{12, 0, "lanes 0-11"}, {9, 0, "lanes 0-8"}, {8, 0, "lanes 8-15", 8}, {8, 0, "lanes 16-23", 16}, {8, 0, "lanes 28-35", 28}, {16, 0, "lanes 8-23", 8},{64, 0, "64 lanes"}, {32, 0, "lanes 0-31"}, {16, 0, "lanes 0-15"}, {8, 0, "lanes 0-7"}, {1, 0, "lane 0"},
	                                                      {16, 4, "16 lanes, every 4th"}, {4, 16, "4 lanes, one per 16"}, {2, 32, "2 lanes, one per 32"}};
	for (auto& c : cases) {
		hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, c.K, c.stride, 100, c.first);
		hipDeviceSynchronize();
		hipEventRecord(e0);
		hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, c.K, c.stride, iters, c.first);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		const double insts = (double)blocks * 4 * iters * 8; // wave-level FMAs
		printf("%-22s %8.3f ms  %.1f G wave-FMA/s\n", c.what, ms, insts / ms / 1e6);
	}
	return 0;
}
