// Test program (CPU): the restatement of glibc's powf that the device code uses (raytracer_amd/csrc/mirt_powf.h)
// against the C library's own powf, on the kinds of argument the shading path produces.  Prints the number of inputs
// whose results differ in any bit, NaN results compared as NaN.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../raytracer_amd/csrc/mirt_powf.h"

static uint64_t s = 88172645463325252ull;
static inline uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline float unit() { return (float)((rnd() >> 11) * (1.0 / 9007199254740992.0)); }

int main(int argc, char** argv) {
	const long n = argc > 1 ? atol(argv[1]) : 4000000;
	long bad = 0, declined = 0, sq_vs_mul = 0;
	const float shin[] = {1.0f, 2.0f, 3.5f, 10.0f, 20.0f, 50.0f, 100.0f, 128.0f, 1000.0f, 0.5f, 1e-3f, 12345.0f};
	for (long i = 0; i < n; ++i) {
		float x, y;
		switch (i % 4) {
		case 0: x = unit(); y = shin[(i / 4) % 12]; break;                       // Blinn-Phong: dot(H,N) in [0,1], a shininess
		case 1: x = 1.0f - unit() * 1e-3f; y = shin[(i / 4) % 12]; break;        // highlights: dot close to 1
		case 2: x = unit() * 1000.0f + 1e-3f; y = 2.0f; break;                   // attenuation: distance squared
		default: { // any float at all, and now and then a special exponent
			uint32_t u = (uint32_t)rnd(); memcpy(&x, &u, 4);
			const float sp[] = {0.0f, -0.0f, 1.0f, -1.0f, 2.0f, 3.0f, -3.0f, 0.5f, __builtin_inff(), -__builtin_inff(), __builtin_nanf(""), 1e30f, -7.0f, 16777216.0f, 16777217.0f * 2};
			y = (rnd() & 7) ? (unit() - 0.5f) * 64.0f : sp[rnd() % 15];
			if (!(rnd() & 15)) { const float sx[] = {0.0f, -0.0f, 1.0f, -1.0f, __builtin_inff(), -__builtin_inff(), __builtin_nanf(""), -2.0f}; x = sx[rnd() % 8]; }
		} break;
		}
		volatile float vy = y; // no compile-time folding of pow(x, 2) into x * x: the C library's function is what is compared
		const float want = powf(x, vy);
		const float got = mirt_powf::powf_glibc(x, y);
		if (memcmp(&got, &want, 4) && !(got != got && want != want)) { if (++bad <= 5) printf("x=%a y=%a want %a got %a\n", x, y, want, got); }
		if (i % 4 == 2) { const float m = x * x; if (memcmp(&m, &want, 4)) ++sq_vs_mul; } // informational: a squared distance is NOT a powf call in the reference's build (g++ folds it)
	}
	printf("n=%ld bad=%ld declined=%ld powf(d,2)!=d*d: %ld\n", n, bad, declined, sq_vs_mul);
	return bad ? 1 : 0;
}
