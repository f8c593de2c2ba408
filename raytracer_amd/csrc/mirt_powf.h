// powf as the reference's build evaluates it, bit for bit.
//
// The reference calls glm::pow(float, float) for the Blinn-Phong term and for the squared distance of the attenuation
// (RayTracer/main.cpp:263, 267), which is std::pow(float, float) = powf of the C library it is linked against.  The
// oracle is the reference compiled with g++ against glibc 2.35 (x86-64), whose powf is NOT correctly rounded: it
// evaluates exp2(y * log2(x)) in double with small tables and short polynomials (sysdeps/ieee754/flt-32/e_powf.c,
// from ARM's optimized-routines; worst case about 0.52 ULP), and about one result in a thousand differs in the last
// bit from the correctly rounded power.  On a CPU with FMA glibc dispatches to a build of that file in which every
// a*b+c of the two polynomials is one fused operation; that is the variant restated here (operation by operation from
// the machine code of __powf_fma in libm.so.6 of glibc 2.35-0ubuntu3.11, constants read from the same file).
// Double-precision add, multiply and FMA are IEEE on gfx950 as on x86-64, so the result is the same float.
//
// Every argument is covered: zeros, infinities, NaNs and negative bases are decided as glibc's code decides them (the C
// standard's special cases), overflow and underflow return what its error paths return under round-to-nearest.
#pragma once
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define MIRT_HD __host__ __device__ __forceinline__
#else
#define MIRT_HD static inline
#endif

namespace mirt_powf {

// log2 table: x = 2^k z, z in [0x1.66p-1, 0x1.66p0) split into 16 intervals; invc ~ 1/c, logc = log2(c) for the centre c
struct LogEntry { double invc, logc; };
#ifdef __HIPCC__
__device__
#endif
static const LogEntry LOG_TAB[16] = {
	{0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
	{0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2}, {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
	{0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3},
	{0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
	{0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1.0000000000000p+0, 0x0.0p+0},
	{0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3},
	{0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
	{0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};
// exp2 table: bits of 2^(i/32) minus i << 47
#ifdef __HIPCC__
__device__
#endif
static const uint64_t EXP2_TAB[32] = {
	0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
	0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
	0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
	0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
	0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
	0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
	0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
	0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};

MIRT_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
MIRT_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
MIRT_HD uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
MIRT_HD double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

// y as an integer: 0 = not an integer, 1 = odd, 2 = even
MIRT_HD int checkint(uint32_t iy) {
	const int e = (int)(iy >> 23 & 0xff);
	if (e < 0x7f) return 0;
	if (e > 0x7f + 23) return 2;
	if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
	if (iy & (1u << (0x7f + 23 - e))) return 1;
	return 2;
}
MIRT_HD bool issignaling(uint32_t i) { return 2u * (i ^ 0x00400000u) > 2u * 0x7fc00000u; }
MIRT_HD bool zeroinfnan(uint32_t i) { return 2u * i - 1u >= 2u * 0x7f800000u - 1u; }

// powf(x, y), every case (round-to-nearest, no errno): the special cases are the C standard's (Annex F), decided the way
// glibc's code decides them.
MIRT_HD float powf_glibc(float x, float y) {
	uint32_t ix = f2u(x);
	const uint32_t iy = f2u(y);
	uint64_t sign_bias = 0;
	if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || zeroinfnan(iy)) {
		// either (x < 0x1p-126 or inf or nan) or (y is 0 or inf or nan)
		if (zeroinfnan(iy)) {
			if (2u * iy == 0u) return issignaling(ix) ? x + y : 1.0f;
			if (ix == 0x3f800000u) return issignaling(iy) ? x + y : 1.0f;
			if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
			if (2u * ix == 2u * 0x3f800000u) return 1.0f;
			if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f; // |x| < 1 && y == inf or |x| > 1 && y == -inf
			return y * y;
		}
		if (zeroinfnan(ix)) {
			float x2 = x * x;
			if ((ix & 0x80000000u) && checkint(iy) == 1) x2 = -x2;
			if (2u * ix == 0u && (iy & 0x80000000u)) return x2 == 0.0f && f2u(x2) == 0x80000000u ? -__builtin_inff() : __builtin_inff(); // 1 / +-0
			return (iy & 0x80000000u) ? 1.0f / x2 : x2;
		}
		// x and y are non-zero finite
		if (ix & 0x80000000u) { // finite x < 0
			const int yint = checkint(iy);
			if (yint == 0) return __builtin_nanf("");
			if (yint == 1) sign_bias = 1ull << 16;
			ix &= 0x7fffffffu;
		}
		if (ix < 0x00800000u) { // subnormal x: normalise
			ix = f2u(u2f(ix) * 0x1p23f) & 0x7fffffffu;
			ix -= 23u << 23;
		}
	}
	// log2(x) = log1p(z/c - 1)/ln2 + log2(c) + k
	const uint32_t tmp = ix - 0x3f330000u;
	const uint32_t i = (tmp >> 19) & 15u;
	const uint32_t top = tmp & 0xff800000u;
	const uint32_t iz = ix - top;
	const int k = (int32_t)top >> 23;
	const double invc = LOG_TAB[i].invc, logc = LOG_TAB[i].logc;
	const double z = (double)u2f(iz);
	const double r = __builtin_fma(z, invc, -1.0);
	const double y0 = logc + (double)k;
	const double r2 = r * r;
	const double ya = __builtin_fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
	const double p = __builtin_fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
	const double r4 = r2 * r2;
	double q = __builtin_fma(0x1.71547652ab82bp+0, r, y0);
	q = __builtin_fma(p, r2, q);
	const double logx = __builtin_fma(ya, r4, q);
	const double ylogx = (double)y * logx;
	if (((d2u(ylogx) >> 47) & 0xffffu) >= (0x405f800000000000ull >> 47)) {   // |y log2 x| >= 126
		if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_inff() : __builtin_inff();   // overflow
		if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;                                         // 0x1p-95f * 0x1p-95f
		if (ylogx < -149.0) return sign_bias ? -0x1p-149f : 0x1p-149f;                                // 0x1.4p-75f * 0x1.4p-75f
	}
	// 2^(k/32) 2^r with r in [-1/64, 1/64]
	const double SHIFT = 0x1.8p+47;
	double kd = ylogx + SHIFT;
	const uint64_t ki = d2u(kd);
	kd -= SHIFT;
	const double rr = ylogx - kd;
	const uint64_t t = EXP2_TAB[ki & 31u] + ((ki + sign_bias) << 47);
	const double s = u2d(t);
	const double zz = __builtin_fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
	const double rr2 = rr * rr;
	const double yy = __builtin_fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
	return (float)(__builtin_fma(zz, rr2, yy) * s);
}

}  // namespace mirt_powf
