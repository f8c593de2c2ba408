// Which VALU instructions slow down when a wave has <= 8 active lanes?  (compile with -ffp-contract=off)
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float pk2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void k(float* out, int K, int iters) {
	const int lane = threadIdx.x & 63;
	float a = lane * 0.001f + 1.0f, c = 0.5f, d = 0.25f, e = 0.125f;
	pk2 p = {a, c}, q = {d, e};
	const pk2 pb = {0.9999f, 0.9999f}, pc = {0.1f, 0.1f};
	unsigned u = lane + 1;
	const float b = 0.9999f;
	if (lane < K) {
		for (int i = 0; i < iters; ++i) {
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				if (KIND == 0) { a = a * b; c = c * b; }                       // v_mul_f32
				else if (KIND == 1) { a = a + b; c = c + b; }                  // v_add_f32
				else if (KIND == 2) { a = __builtin_fmaf(a, b, 0.1f); c = __builtin_fmaf(c, b, 0.1f); } // v_fma_f32
				else if (KIND == 3) { p = __builtin_elementwise_fma(p, pb, pc); q = __builtin_elementwise_fma(q, pb, pc); } // v_pk_fma_f32
				else if (KIND == 4) { a = fminf(a, c + 1.0f); c = fmaxf(c, d); d = fminf(d, a); }            // min/max (+1 add)
				else if (KIND == 5) { a = (float)((u >> 8) & 0xffu); u = u * 5u + (unsigned)a; }             // v_cvt_f32_ubyte + int
				else if (KIND == 6) { a = __builtin_amdgcn_rcpf(a + 1.0f); }                                  // v_rcp_f32
			}
		}
	}
	if (a + c + d + e + p.x + p.y + q.x + q.y + (float)u == 12345.0f) out[threadIdx.x] = a;
}
template <int KIND>
void run(const char* what, float* out) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int iters = 20000, blocks = 256 * 8;
	float t[2];
	int n = 0;
	for (int K : {16, 8}) {
		hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, K, 100);
		hipDeviceSynchronize();
		hipEventRecord(e0);
		hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, K, iters);
		hipEventRecord(e1); hipEventSynchronize(e1);
		hipEventElapsedTime(&t[n++], e0, e1);
	}
	printf("%-28s 16 lanes %7.3f ms   8 lanes %7.3f ms   x%.2f\n", what, t[0], t[1], t[1] / t[0]);
}
int main() {
	float* out; hipMalloc(&out, 4096);
	run<0>("v_mul_f32", out); run<1>("v_add_f32", out); run<2>("v_fma_f32", out); run<3>("v_pk_fma_f32", out);
	run<4>("v_min/max_f32", out); run<5>("v_cvt_f32_ubyte + int", out); run<6>("v_rcp_f32 + add", out);
	return 0;
}
