// Follow-up to exec_mask.hip: is the slow-down with <= 8 active lanes tied to FMA, to dependent chains, to occupancy?
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
template <int KIND>
__global__ void k(float* out, int K, int iters) {
	const int lane = threadIdx.x & 63;
	float a = lane * 0.001f + 1.0f, c = 0.5f, d = 0.25f, e = 0.125f;
	unsigned u = lane, v = 3, x = 5, y = 7;
	const float b = 0.9999f;
	if (lane < K) {
		for (int i = 0; i < iters; ++i) {
			if (KIND == 0) { // eight independent FMAs per round (four chains, two steps each)
#pragma unroll
				for (int r = 0; r < 2; ++r) { a = __builtin_fmaf(a, b, 0.1f); c = __builtin_fmaf(c, b, 0.1f); d = __builtin_fmaf(d, b, 0.1f); e = __builtin_fmaf(e, b, 0.1f); }
			}
			else if (KIND == 1) { // integer adds/xors, dependent
#pragma unroll
				for (int r = 0; r < 2; ++r) { u = u * 3u + v; v = v ^ x; x = x + y; y = y + u; }
			}
			else { // v_mul_f32 + v_add_f32 dependent
#pragma unroll
				for (int r = 0; r < 2; ++r) { a = a * b; a = a + c; c = c * b; c = c + a; }
			}
		}
	}
	if (a + c + d + e + (float)(u + v + x + y) == 12345.0f) out[threadIdx.x] = a;
}
template <int KIND>
void run(const char* what, float* out, int wavesPerSimd) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 20000, blocks = 256 * wavesPerSimd;
	for (int K : {64, 16, 9, 8, 4, 1}) {
		hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, K, 100);
		hipDeviceSynchronize();
		hipEventRecord(e0);
		hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, K, iters);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		printf("%-34s %d waves/SIMD  K=%2d %8.3f ms\n", what, wavesPerSimd, K, ms);
	}
}
int main() {
	float* out; hipMalloc(&out, 4096);
	run<0>("independent FMA chains", out, 8);
	run<0>("independent FMA chains", out, 2);
	run<1>("integer mul/add/xor", out, 8);
	run<2>("mul+add dependent", out, 8);
	return 0;
}
