// HIP kernels of the trace path for gfx950 (MI355X).  Compiled with -ffp-contract=off and correctly rounded
// division / square root: in MIRT_TRACE_EXACT mode every floating-point operation below is the operation the
// reference performs, in the reference's order, so results are bit-identical to the x86-64 build of the
// reference (which has no FMA contraction either).  Where an FMA or a reciprocal is wanted (MIRT_TRACE_FAST)
// it is written explicitly.
//
// Reference lines restated here:
//   primary rays        Camera::createRayToPixel                  RayTracer/Camera.cpp:15-24
//   box test            AABB::intersect (hit/miss part)           RayTracer/AABB.cpp:53-72
//   triangle test       Triangle::intersect                       RayTracer/Triangle.cpp:147-175
//   sphere test         Sphere::intersect                         RayTracer/Sphere.cpp:59-87
//   tree walk           Partition::intersect, leaf scan           RayTracer/Partition.cpp:35-53, Partition.h:28-39
//   shading             computePixelColor, calculateLightingColor RayTracer/main.cpp:176-268
//   quantisation        Color::getR/G/B                           RayTracer/Color.cpp:13-23
#include "mirt_device.h"
#include "mirt_powf.h"

#pragma clang fp contract(off)

#define WAVE 64
#define BLOCK 256
#define STACK_MAX 96

#define F_DIFFUSE 1u
#define F_SPECULAR 2u
#define F_SHADOWS 4u
#define F_REFLECTIONS 8u
#define D_DIFFUSE_INTENSITY 0u
#define D_SPECULAR_INTENSITY 1u
#define D_NORMAL_MAP 2u
#define D_SHADOW_MAP 3u
#define D_PRIMARY_MAP 4u
#define D_LIGHT_DIR_MAP 5u
#define D_NONE 6u

#define MODE_EXACT 0
#define MODE_FAST 1
#define MODE_REFWALK 2

// diagnostic build only (-DMIRT_STATS=1): event counters, per wave-iteration and per lane
#if MIRT_STATS
#define STAT_LANE(S, k) atomicAdd((S).dbg + (k), 1ull)
#define STAT_WAVE(S, k) do { if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) atomicAdd((S).dbg + (k), 1ull); } while (0)
#else
#define STAT_LANE(S, k)
#define STAT_WAVE(S, k)
#endif


struct f3 {
	float x, y, z;
};
__device__ __forceinline__ f3 F3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
__device__ __forceinline__ f3 F3(const float4& q) { f3 r = {q.x, q.y, q.z}; return r; }
__device__ __forceinline__ f3 F3(const float* p) { f3 r = {p[0], p[1], p[2]}; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
// GLM 0.9.8.5 scalar formulas (glm/detail/func_geometric.inl:54-99)
__device__ __forceinline__ float dot3(f3 a, f3 b) {
	float x = a.x * b.x, y = a.y * b.y, z = a.z * b.z;
	return x + y + z;
}
__device__ __forceinline__ f3 cross3(f3 a, f3 b) { return F3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
__device__ __forceinline__ float inv_sqrt(float x) { return 1.0f / sqrtf(x); }
__device__ __forceinline__ f3 normalize3(f3 a) { return a * inv_sqrt(dot3(a, a)); }
__device__ __forceinline__ float length3(f3 a) { return sqrtf(dot3(a, a)); }

// glm::pow(float, float) = powf of the C library the reference is linked against (main.cpp:263): glibc's powf, which is
// not correctly rounded; restated bit for bit in mirt_powf.h.
__device__ __forceinline__ float ref_powf(float x, float y) { return mirt_powf::powf_glibc(x, y); }

struct Hit {
	float dist;   // world distance |P - o| (Intersection::distAlongRay); +inf = miss
	uint32_t ref; // typed primitive reference (PRIM_SPHERE | index) or REF_NONE
	uint32_t pos; // position of the winning entry in the leaf stream (tie-break rank)
};

struct RayCtx {
	f3 o, d;
	f3 ie;     // 1/d for the hierarchy's conservative box filter
	f3 oie;    // o * ie
	float invLen; // 1/|d|, used only for conservative culling
};

__device__ __forceinline__ RayCtx make_ray(f3 o, f3 d) {
	RayCtx r;
	r.o = o;
	r.d = d;
	// reciprocal of the TRUE direction for the hierarchy's conservative filter (tiny components clamped, sign kept)
	const float sx = fabsf(d.x) < 1e-20f ? copysignf(1e-20f, d.x) : d.x;
	const float sy = fabsf(d.y) < 1e-20f ? copysignf(1e-20f, d.y) : d.y;
	const float sz = fabsf(d.z) < 1e-20f ? copysignf(1e-20f, d.z) : d.z;
	// v_rcp_f32 / v_rsq_f32 are within 1 ulp: far inside the filter's padding and the 1e-4 culling slack
	r.ie = F3(__builtin_amdgcn_rcpf(sx), __builtin_amdgcn_rcpf(sy), __builtin_amdgcn_rcpf(sz));
	r.invLen = __builtin_amdgcn_rsqf(dot3(d, d));
	r.oie = r.o * r.ie;
	return r;
}

// Ray parameter beyond which nothing can come closer than `dist`.  A hit's distance is |o - (o + t d)| as the reference
// rounds it (Triangle.cpp:158,167; Sphere.cpp:81-82): off t |d| by a few ulps of the COORDINATES, not of the distance -- far
// from the origin that is more than any relative slack on dist -- hence S.cullSlack: 1e-5 plus 2^-17 of the largest
// |x| + |y| + |z| a ray of this scene can start from (the camera, the corners of the scene's bounds), in units of the ray
// parameter of the shortest direction vector in use (camera rays are at least 1 long, all others normalised; the ray-batch
// entry points look at the rays they are given).
__device__ __forceinline__ float cull_bound(const DScene& S, const RayCtx& r, float dist) {
	return dist * r.invLen * 1.0001f + S.cullSlack;
}
// A ray without a direction or without an origin (see DScene::nanHit).  All three components: that is what normalize(0) and
// a point on such a ray are; with only some of them NaN, what the reference's std::min / std::max keep depends on the operand
// order, and the ordinary walk's comparisons -- false for every box -- give its answer (no hit) in the cases seen.
__device__ __forceinline__ bool all_nan(f3 v) { return v.x != v.x && v.y != v.y && v.z != v.z; }

// The direction with AABB.cpp:53-58's substitution: |d| < 1e-5 -> +1e-5.  Not kept in RayCtx: only the reference's own
// box test needs it, and three registers held through the whole walk cost more than recomputing it there.
__device__ __forceinline__ f3 box_dir(const RayCtx& r) {
	return F3((fabsf(r.d.x - 0.0f) < 0.00001f) ? 0.00001f : r.d.x, (fabsf(r.d.y - 0.0f) < 0.00001f) ? 0.00001f : r.d.y,
	          (fabsf(r.d.z - 0.0f) < 0.00001f) ? 0.00001f : r.d.z);
}

// AABB::intersect hit/miss (AABB.cpp:53-72), e = box_dir(r).  tnear receives tmin (ray-parameter units).
template <int MODE>
__device__ __forceinline__ bool box_hit(const RayCtx& r, const f3& e, f3 lo, f3 hi, float& tnear) {
	float t1, t2, t3, t4, t5, t6;
	if (MODE == MODE_EXACT) {
		t1 = (lo.x - r.o.x) / e.x;
		t2 = (hi.x - r.o.x) / e.x;
		t3 = (lo.y - r.o.y) / e.y;
		t4 = (hi.y - r.o.y) / e.y;
		t5 = (lo.z - r.o.z) / e.z;
		t6 = (hi.z - r.o.z) / e.z;
	}
	else {
		t1 = (lo.x - r.o.x) * r.ie.x;
		t2 = (hi.x - r.o.x) * r.ie.x;
		t3 = (lo.y - r.o.y) * r.ie.y;
		t4 = (hi.y - r.o.y) * r.ie.y;
		t5 = (lo.z - r.o.z) * r.ie.z;
		t6 = (hi.z - r.o.z) * r.ie.z;
	}
	// no NaN can reach here (e is never 0), so fminf/fmaxf equal std::min/std::max (AABB.cpp:60-61)
	float tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
	float tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
	tnear = tmin;
	// AABB.cpp:65 compares with the double literal 0.0001; for a float that is "tmax <= 0.0001f"
	// (0.0001f < 0.0001 < nextafterf(0.0001f)).
	if (tmax <= 0.0001f) return false;
	if (tmin > tmax) return false; // :70
	// :73-78, Partition.cpp:36: the box counts if the Intersection it returns is valid, i.e. its distance -- tmin, or tmax
	// when tmin lies behind the origin -- is not infinite (an empty scene's root box is +inf .. -inf; a NaN counts)
	const float tResult = tmin < 0.0001f ? tmax : tmin;
	return !isinf(tResult);
}

// Triangle::intersect (Triangle.cpp:147-175) on the precomputed records.  Returns the hit distance or -1.
// tcull: ray-parameter bound beyond which a hit cannot matter (conservative, see traverse()).
//
// T = { N, dot(v1,N) } { v3, E } { a0, c1 } { a1, K },  X = { v1, area } { v2, 0 }.
// The reference decides "inside" from the signs of s_i = dot(cross(e_i, P - v_i'), N) against the sign of the area
// (:163-166), about fifty dependent operations.  By the triple-product identity s_i = dot(A_i, P - v_i') with
// A_i = cross(N, e_i), an affine function of q = P - v3; and because the edges add up to (almost) zero,
// s_2 = C_1 + C_2 - s_0 - s_1 + dot(cross(N, e_1 + e_2 + e_3), q) with C_i = dot(A_i, v3 - v_i').  The host evaluates
// A_0, A_1, C_1, C_1 + C_2 in double, multiplies them by sign(area) 2^-g, where 2^g bounds G_i = sum_c |N_c| (|e_a| + |e_b|)
// of every edge and |N|_1 L (L = the triangle's largest edge extent), and rounds to float: a0, a1, c1, K.  Then
//   s~0 = a0.q,  s~1 = a1.q + c1  (FMA chains),  s~2 = (K - s~0) - s~1,
// and, with u = 2^-24, |s~_i - sign(area) 2^-g s_i| <= 27.2 u (|q|_inf + L) for every i: the reference's own evaluation is
// within 6 u G_i (|q|_inf + L) of the real value (six roundings per term); a chain is within 5.02 u |q|_inf resp.
// 6.03 u (|q|_inf + L) (2 u per product, 1 u for c1, three chain roundings); K adds 2.02 u L, the two subtractions
// 6 u L + 4 u |q|_inf, and the neglected edge-sum term is below 6 u |q|_inf (|e_1 + e_2 + e_3|_inf <= 3 u L).
// With eps = 2^-18 (|q|_1 + L) = 64 u (...) (E holds 2^-18 L) the sign of s_i is certain when |s~_i| > eps, which leaves a
// ~1e-5 fraction of the tests undecided; those, and every triangle or point whose magnitudes leave room for overflow or
// underflow in either evaluation (host: E = NaN; here: |q|_1 >= 1e10), run the reference's operations as they stand.
#if MIRT_STATS
__device__ unsigned long long g_statFallback[1];
__device__ unsigned long long g_statPush[2]; // traversal-stack pushes: all, and those that went to the global overflow area
__device__ unsigned long long g_statMembEntries[1]; // leaf boxes tested by membership()
#endif
template <int MODE>
__device__ __forceinline__ float tri_hit(const RayCtx& r, const float4* __restrict__ T, const float4* __restrict__ X, float tcull) {
	// The plane in ONE 16-byte load: left to itself the compiler fetches dot(v1,N) only after the first test, one more
	// trip to memory on the path of nearly every test (+3 % when avoided).  Fetching the whole record up front, for one
	// trip instead of two, loses 2 %: a quarter of the tests end at the plane.
	// (Inline asm rather than a volatile access: that would become a FLAT load with system-scope cache policy.)
	typedef float v4f __attribute__((ext_vector_type(4)));
	v4f w0;
	asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(w0) : "v"(T) : "memory");
	const float4 q0 = make_float4(w0.x, w0.y, w0.z, w0.w);
	const f3 N = F3(q0);
	const float dn = dot3(r.d, N);
	if (fabsf(dn - 0.0f) < 0.0001f) return -1.0f;            // :152
	const float t = (q0.w - dot3(r.o, N)) / dn;               // :153-154
	if (t < 0.0001f) return -1.0f;                            // :155
	if (t > tcull) return -1.0f;                              // cannot beat the current best
	const float4 k0 = T[1], k1 = T[2], k2 = T[3];
	const f3 v3 = F3(k0);
	const f3 P = r.o + r.d * t;                               // :158
	const f3 q = P - v3;                                      // the first operand of :163 as well
	const float s0f = __builtin_fmaf(k1.z, q.z, __builtin_fmaf(k1.y, q.y, k1.x * q.x));
	const float s1f = __builtin_fmaf(k2.z, q.z, __builtin_fmaf(k2.y, q.y, __builtin_fmaf(k2.x, q.x, k1.w)));
	const float s2f = (k2.w - s0f) - s1f;
	const float m = (fabsf(q.x) + fabsf(q.y)) + fabsf(q.z);  // NaN propagates into eps and fails both comparisons
	const float eps = __builtin_fmaf(m, 0x1p-18f, k0.w);
	const float lof = fminf(fminf(s0f, s1f), s2f);
	const bool sane = m < 1e10f;
	bool in;
	if (sane && lof > eps) in = true;
	else if (sane && lof < -eps) in = false;
	else {
#if MIRT_STATS
		atomicAdd(g_statFallback, 1ull);
#endif
		const float4 q1 = X[0], q2 = X[1];
		const f3 v1 = F3(q1), v2 = F3(q2);
		const float area = q1.w;                                  // :162, evaluated on the host
		const f3 e1 = v3 - v2, e2 = v1 - v3, e3 = v2 - v1;        // :148-150
		const float s0 = dot3(cross3(e1, q), N);                  // :163
		const float s1 = dot3(cross3(e2, P - v1), N);             // :164
		const float s2 = dot3(cross3(e3, P - v2), N);             // :165
		// (s/2)/area >= 0 is decided by the signs alone unless the quotient can underflow (to a zero that compares >= 0)
		const float lo = fminf(fminf(fabsf(s0), fabsf(s1)), fabsf(s2));
		if (MODE == MODE_FAST || (fabsf(area) < 1e7f && fabsf(area) > 1e-30f && lo > 1e-30f)) {
			in = area > 0.0f ? (s0 >= 0.0f && s1 >= 0.0f && s2 >= 0.0f) : (s0 <= 0.0f && s1 <= 0.0f && s2 <= 0.0f);
		}
		else {
			in = (s0 / 2.0f / area >= 0.0f) && (s1 / 2.0f / area >= 0.0f) && (s2 / 2.0f / area >= 0.0f);
		}
	}
	if (!in) return -1.0f;
	return length3(r.o - P);                                   // :167 glm::distance(P, origin)
}

// m points at 4 float4 columns
__device__ __forceinline__ float4 mat_mul(const float4* __restrict__ m, float4 v) {
	const float4 c0 = m[0], c1 = m[1], c2 = m[2], c3 = m[3];
	float4 r;
	r.x = (c0.x * v.x + c1.x * v.y) + (c2.x * v.z + c3.x * v.w);
	r.y = (c0.y * v.x + c1.y * v.y) + (c2.y * v.z + c3.y * v.w);
	r.z = (c0.z * v.x + c1.z * v.y) + (c2.z * v.z + c3.z * v.w);
	r.w = (c0.w * v.x + c1.w * v.y) + (c2.w * v.z + c3.w * v.w);
	return r;
}
__device__ __forceinline__ float4 mat_tmul(const float4* __restrict__ m, float4 v) {
	const float4 c0 = m[0], c1 = m[1], c2 = m[2], c3 = m[3];
	float4 r;
	r.x = (c0.x * v.x + c0.y * v.y) + (c0.z * v.z + c0.w * v.w);
	r.y = (c1.x * v.x + c1.y * v.y) + (c1.z * v.z + c1.w * v.w);
	r.z = (c2.x * v.x + c2.y * v.y) + (c2.z * v.z + c2.w * v.w);
	r.w = (c3.x * v.x + c3.y * v.y) + (c3.z * v.z + c3.w * v.w);
	return r;
}

// Sphere::intersect (Sphere.cpp:59-87).  Returns the world distance or -1; n (un-normalised) when wanted.
template <bool WANT_N>
__device__ __forceinline__ float sphere_hit(const RayCtx& r, const float4* __restrict__ S, f3& nOut) {
	const float4 q0 = S[0];
	const float4* M = S + 1;
	const float4* Mi = S + 5;
	const float4 oo = mat_mul(Mi, make_float4(r.o.x, r.o.y, r.o.z, 1.0f)); // :60
	const float4 dd = mat_mul(Mi, make_float4(r.d.x, r.d.y, r.d.z, 0.0f));
	const float l4 = (dd.x * dd.x + dd.y * dd.y) + (dd.z * dd.z + dd.w * dd.w); // dot of the vec4
	const float inv = inv_sqrt(l4);
	const f3 o = F3(oo.x, oo.y, oo.z);
	const f3 d = F3(dd.x * inv, dd.y * inv, dd.z * inv);
	const f3 c = F3(q0);
	const f3 nd = normalize3(d);                      // :61
	const float a = dot3(nd, nd);                     // :62
	const f3 oc = o - c;
	const float b = 2.0f * dot3(nd, oc);              // :63
	const float cc = dot3(oc, oc) - q0.w;             // :64, powf(radius, 2) evaluated on the host
	// :85-87: pow(b, 2) with an int exponent is evaluated in double, as is the subtraction
	const float disc = (float)((double)b * (double)b - (double)(4.0f * a * cc));
	if (disc < 0.0f) return -1.0f;                    // :66
	const float sq = sqrtf(disc);
	const float x1 = (-b + sq) / 2.0f * a;            // :69
	const float x2 = (-b - sq) / 2.0f * a;            // :70
	float t = (x2 < x1) ? x2 : x1;                    // :71 std::min
	if (t < 0.001f) {                                 // :72-77
		t = (x1 < x2) ? x2 : x1;
		if (t < 0.001f) return -1.0f;
	}
	const f3 p = o + d * t;                           // :78
	if (WANT_N) {
		const f3 pc = p - c;
		const float4 nn = mat_tmul(Mi, make_float4(2.0f * pc.x, 2.0f * pc.y, 2.0f * pc.z, 0.0f)); // :79
		nOut = F3(nn.x, nn.y, nn.z);
	}
	const float4 pw = mat_mul(M, make_float4(p.x, p.y, p.z, 1.0f)); // :81
	return length3(r.o - F3(pw.x, pw.y, pw.z));       // :82
}

// ---- MODE_REFWALK: walk of the reference's own tree ---------------------------------------------------------------
// ANY = false: nearest hit under the reference's tie rules.  ANY = true: is there a hit with dist < limit (the shadow
// query of main.cpp:223-224, where "closest.dist >= length(lightDir)" decides).
// Children are visited nearer-first and, in scenes made of triangles only, a subtree is skipped when its entry parameter
// exceeds `tcull`, a bound kept slightly beyond the best distance so far: a point of a triangle lies in a leaf that holds
// the triangle (Triangle::isInside is an overlap test, Triangle.cpp:58-95), so every candidate at a distance <= best is
// still visited and the result equals the reference's exhaustive walk (Partition.cpp:42-52).  Sphere::isInside
// (Sphere.cpp:43-57) gives no such guarantee -- a transformed sphere can be hit outside every leaf that holds it -- so
// with spheres in the scene (S.refCull == 0) nothing is skipped.
// One thing the skipping does change: which of two primitives hit at the SAME distance the reference returns is decided
// by the first place either of them occupies among all the leaves the ray's boxes lead to (Partition.cpp:47, Partition.h:33)
// -- and a triangle also sits in leaves far behind the hit, which a bounded walk never sees.  Both rivals are always met
// (each lies in a leaf around the hit point), so the tie itself is noticed, and such a ray is walked a second time without
// the bound.  Stack in private memory: this mode is the fallback and the cross-check, not the fast path.
template <bool ANY>
__device__ __noinline__ bool traverse_ref(const DScene& S, const RayCtx& r, float limit, Hit& best) {
	float tn;
	const f3 e = box_dir(r);
	if (S.rootRef == REF_NONE || !box_hit<MODE_EXACT>(r, e, F3(S.rootMin), F3(S.rootMax), tn)) { // Partition.cpp:36 on the root
		best.dist = __builtin_inff();
		best.ref = REF_NONE;
		return false;
	}
	uint32_t stkRef[STACK_MAX];
	float stkT[STACK_MAX];
	bool cull = S.refCull != 0, tie = false;
 again:
	best.dist = __builtin_inff();
	best.ref = REF_NONE;
	best.pos = 0xffffffffu;
	float tcull = (ANY && cull) ? cull_bound(S, r, limit) : __builtin_inff();
	int sp = 0;
	uint32_t cur = S.rootRef;
	for (;;) {
		if (!(cur & REF_LEAF)) {
			const float4* nd = S.nodes + (size_t)cur * 4;
			const float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
			const uint32_t lref = __float_as_uint(q0.w), rref = __float_as_uint(q1.w);
			float tl, tr;
			const bool hl = (lref != REF_NONE) && box_hit<MODE_EXACT>(r, e, F3(q0), F3(q1), tl) && !(tl > tcull);
			const bool hr = (rref != REF_NONE) && box_hit<MODE_EXACT>(r, e, F3(q2), F3(q3), tr) && !(tr > tcull);
			if (hl && hr) {
				const bool leftFirst = tl <= tr;
				stkRef[sp] = leftFirst ? rref : lref;
				stkT[sp] = leftFirst ? tr : tl;
				++sp;
				cur = leftFirst ? lref : rref;
				continue;
			}
			if (hl) { cur = lref; continue; }
			if (hr) { cur = rref; continue; }
		}
		else {
			const uint32_t p = cur & ~REF_LEAF;
			const uint32_t cnt = S.leafs[p];
			for (uint32_t k = 0; k < cnt; ++k) {               // Partition.h:31
				const uint32_t pos = p + 1 + k;
				const uint32_t pr = S.leafs[pos];
				float dist;
				if (pr & PRIM_SPHERE) {
					f3 dummy;
					dist = sphere_hit<false>(r, S.spheres + (size_t)(pr & ~PRIM_SPHERE) * SPHERE_F4, dummy);
				}
				else {
					dist = tri_hit<MODE_EXACT>(r, S.tris + (size_t)pr * TRI_F4, S.triX + (size_t)pr * 2, tcull);
				}
				if (dist >= 0.0f && !isinf(dist)) {
					if (ANY) {
						if (dist < limit) { best.dist = dist; best.ref = pr; best.pos = pos; return true; }
					}
					else {
						if (dist == best.dist && pr != best.ref) tie = true;
						if (dist < best.dist || (dist == best.dist && pos < best.pos)) { // Partition.h:33, Partition.cpp:47
							best.dist = dist;
							best.ref = pr;
							best.pos = pos;
							if (cull) tcull = cull_bound(S, r, dist);
						}
					}
				}
			}
		}
		// pop the next subtree that can still matter
		for (;;) {
			if (sp == 0) {
				if (!ANY && cull && tie) { cull = false; goto again; }
				return best.ref != REF_NONE;
			}
			--sp;
			cur = stkRef[sp];
			if (!(stkT[sp] > tcull)) break;
		}
	}
}

// ---- rays whose components are not all finite: the reference's recursion as it stands ------------------------------------------
// What Partition::intersect makes of a NaN or an infinity hangs on details no bounded or ordered walk keeps: std::min and
// std::max return their FIRST argument when the comparison fails (AABB.cpp:60-61), an Intersection with a NaN distance counts as
// valid (SceneObjects.hpp:35-37), a leaf keeps the first of two NaN answers (Partition.h:33) and a node the right one
// (Partition.cpp:47).  So such a ray -- none arises in a frame except the all-NaN ones DScene::nanHit answers; the ray-batch
// entry points can be handed any -- is evaluated exactly as the reference recurses: every node, left subtree then right, no
// bound, the comparisons spelled as the reference spells them.  Slow, private arrays, used by the ray-batch kernels only.
__device__ __forceinline__ float min_std(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float max_std(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ bool box_valid_ref(f3 o, f3 e, f3 lo, f3 hi) {
	const float t1 = (lo.x - o.x) / e.x, t2 = (hi.x - o.x) / e.x, t3 = (lo.y - o.y) / e.y, t4 = (hi.y - o.y) / e.y, t5 = (lo.z - o.z) / e.z, t6 = (hi.z - o.z) / e.z;
	const float tmin = max_std(max_std(min_std(t1, t2), min_std(t3, t4)), min_std(t5, t6)); // :60
	const float tmax = min_std(min_std(max_std(t1, t2), max_std(t3, t4)), max_std(t5, t6)); // :61
	if ((double)tmax < 0.0001) return false;   // :65
	if (tmin > tmax) return false;             // :70
	const float tResult = tmin < 0.0001f ? tmax : tmin; // :73-78
	return !isinf(tResult);                    // Partition.cpp:36
}
__device__ __noinline__ void traverse_faithful(const DScene& S, f3 o, f3 d, Hit& best) {
	const RayCtx r = make_ray(o, d);
	const f3 e = box_dir(r);
	uint32_t stkNode[STACK_MAX], stkRef[STACK_MAX];
	float stkDist[STACK_MAX];
	int sp = 0;
	// the subtree to evaluate next: its reference and box
	uint32_t cur = S.rootRef;
	f3 lo = F3(S.rootMin), hi = F3(S.rootMax);
	float resDist;   // result of the subtree just evaluated: +inf = no valid intersection (Intersection(), SceneObjects.hpp:33)
	uint32_t resRef;
	for (;;) {
		resDist = __builtin_inff();
		resRef = REF_NONE;
		bool descend = false;
		if (cur != REF_NONE && box_valid_ref(o, e, lo, hi)) {                     // Partition.cpp:36
			if (cur & REF_LEAF) {                                                  // :39-41, Partition.h:28-39
				const uint32_t p = cur & ~REF_LEAF, cnt = S.leafs[p];
				for (uint32_t k = 0; k < cnt; ++k) {
					const uint32_t pr = S.leafs[p + 1 + k];
					float dist;
					f3 dummy;
					if (pr & PRIM_SPHERE) dist = sphere_hit<false>(r, S.spheres + (size_t)(pr & ~PRIM_SPHERE) * SPHERE_F4, dummy);
					else dist = tri_hit<MODE_EXACT>(r, S.tris + (size_t)pr * TRI_F4, S.triX + (size_t)pr * 2, __builtin_inff());
					const bool valid = dist != -1.0f && !isinf(dist);              // (-1: the tests' "no intersection"; a NaN is valid)
					if (valid && (dist < resDist || isinf(resDist))) { resDist = dist; resRef = pr; } // Partition.h:33
				}
			}
			else descend = true;
		}
		if (descend) {                                                             // :42: the left subtree first
			const float4* nd = S.nodes + (size_t)cur * 4;
			const float4 q0 = nd[0], q1 = nd[1];
			stkNode[sp] = cur;
			stkRef[sp] = 0xfffffffeu; // "left result pending"
			++sp;
			cur = __float_as_uint(q0.w);
			lo = F3(q0); hi = F3(q1);
			continue;
		}
		// a subtree is finished: hand its result to the node above
		for (;;) {
			if (sp == 0) { best.dist = resDist; best.ref = resRef; best.pos = 0; return; }
			const uint32_t node = stkNode[sp - 1];
			if (stkRef[sp - 1] == 0xfffffffeu) {                                   // that was the left subtree: keep it, do the right one (:43)
				stkRef[sp - 1] = resRef;
				stkDist[sp - 1] = resDist;
				const float4* nd = S.nodes + (size_t)node * 4;
				const float4 q1 = nd[1], q2 = nd[2], q3 = nd[3];
				cur = __float_as_uint(q1.w);
				lo = F3(q2); hi = F3(q3);
				break;
			}
			// both done (:44-52)
			const float lDist = stkDist[sp - 1];
			const uint32_t lRef = stkRef[sp - 1];
			const bool lValid = !isinf(lDist), rValid = !isinf(resDist);
			--sp;
			if (!(lValid || rValid)) { resDist = __builtin_inff(); resRef = REF_NONE; }
			else if ((lValid && !rValid) || lDist <= resDist) { resDist = lDist; resRef = lRef; }
			// else: the right one, which is in res already
		}
	}
}
// An ordinary ray: finite origin, and a direction whose squared length neither underflows nor overflows where a primitive test
// normalises it (Sphere.cpp:60-61 does so in object space: normalize of a vector of squared length 0 is NaN)
__device__ __forceinline__ bool all_finite(f3 o, f3 d) {
	const float len2 = dot3(d, d);
	return fabsf(o.x) < __builtin_inff() && fabsf(o.y) < __builtin_inff() && fabsf(o.z) < __builtin_inff() && len2 >= 1e-30f && len2 <= 1e30f;
}

// ---- primitive hierarchy + membership check (MODE_EXACT / MODE_FAST) -----------------------------------------------------
// Slab test for the hierarchy's own boxes: t = b*(1/d) - o*(1/d) as one FMA per plane.  It must never reject a box that
// contains a point the exact primitive tests can accept.  Its rounding error, mapped back to world space, is below
// 2^-22 * (|o| + |b|) per axis; mirt_scene_create pads every box by 1e-5 * (|b| + D) with D >= |o|, two decimal orders
// more, so the plain comparisons below are conservative without any run-time slack.  tnear = entry parameter.
__device__ __forceinline__ bool box_filter(const RayCtx& r, const float4& lo, const float4& hi, float tcull, float& tnear) {
	const float t1 = __builtin_fmaf(lo.x, r.ie.x, -r.oie.x), t2 = __builtin_fmaf(hi.x, r.ie.x, -r.oie.x);
	const float t3 = __builtin_fmaf(lo.y, r.ie.y, -r.oie.y), t4 = __builtin_fmaf(hi.y, r.ie.y, -r.oie.y);
	const float t5 = __builtin_fmaf(lo.z, r.ie.z, -r.oie.z), t6 = __builtin_fmaf(hi.z, r.ie.z, -r.oie.z);
	const float tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
	const float tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
	tnear = tmin;
	return (tmin <= tmax) && (tmax >= 0.0f) && !(tmin > tcull);
}

// Is the primitive held by a reference leaf whose box the ray hits?  Returns the visiting rank of the first such
// (leaf, primitive) entry, or REF_NONE.  AABB::intersect is evaluated exactly (IEEE division, AABB.cpp:53-72).
__device__ __forceinline__ uint32_t membership(const DScene& S, const RayCtx& r, uint32_t ofs, uint32_t cnt) {
	const float4* m = S.memb + (size_t)ofs * 2;
	const f3 e = box_dir(r);
	for (uint32_t k = 0; k < cnt; ++k) {
#if MIRT_STATS
		atomicAdd(g_statMembEntries, 1ull);
#endif
		const float4 lo = m[2 * k], hi = m[2 * k + 1];
		float tn;
		if (box_hit<MODE_EXACT>(r, e, F3(lo), F3(hi), tn)) return __float_as_uint(lo.w);
	}
	return REF_NONE;
}

// Occupancy: the walk is bound by VALU issue and by the latency of its dependent loads, and it gains from every extra
// resident wave up to 7 per SIMD (Stanford dragon at 3840x2160, Grays/s: 4 waves 7.4, 5: 7.9, 6: 8.5, 7: 8.75, 8: 8.85;
// 10 M triangles: 6 waves 2.62, 7: 2.72, 8: 2.26 -- at 8 the spills of a 64-register budget cost more than the wave
// brings).  7 waves per SIMD means 72 VGPRs and, with 7 blocks per CU, 22.5 KB of LDS per block: 18 stack entries per lane
// next to the wave lists.  The translation unit is compiled without the SLP vectoriser: its packed f32 operations need
// operands in even-aligned register pairs, and the moves and the wider live ranges that brings cost 35 registers here.
#ifndef STACK_LDS
#define STACK_LDS 18
#endif
#ifndef STACK_LDS_REFILL
#define STACK_LDS_REFILL 18
#endif
// k_frame keeps its waves' record tables in LDS as well and has one stack entry less for it
#ifndef STACK_LDS_FRAME
#define STACK_LDS_FRAME (STACK_LDS > 17 ? 17 : STACK_LDS)
#endif
#ifndef MIRT_BATCH_WAVES
#define MIRT_BATCH_WAVES 7
#endif
#ifndef MIRT_REFILL_WAVES
#define MIRT_REFILL_WAVES 7
#endif
#ifndef MIRT_FRAME_WAVES
#define MIRT_FRAME_WAVES 7
#endif
// threads of a k_frame block: its waves share nothing, so a block is ONE wave -- a block's place on the CU is free for the
// next frame's block as soon as that wave is done, not when the slowest of four is
#ifndef FRAME_BLOCK
#define FRAME_BLOCK 64
#endif
// (k_trace keeps four waves to a block: with one, the 10-million-triangle scene and the Stanford dragon on the per-depth
// path both came out 2-2.5 % slower)

// Traversal stack: STACK_LDS entries per lane in LDS ([entry][lane]: conflict-free, no address arithmetic beyond a
// shift).  Hierarchies deeper than that (OVF = true, chosen by the host from the built depth) keep the deeper entries in
// a global array; the two cases are separate instantiations so that the common one contains LDS instructions only
// (a merged pointer would turn every pop into a flat_load).
template <bool OVF, int NLDS = STACK_LDS, int STRIDE = BLOCK>
struct LdsStack {
	static constexpr int stride = STRIDE, nlds = NLDS;
	// (The frame kernel has two speeds, 14.3 and 12.3 Grays/s on the Stanford dragon, and small changes anywhere in it flip it
	// to the slow one: then the compiler has picked THIS pointer for spilling and every push and pop of the walk waits for a
	// scratch load.  Check `make asm` for scratch loads in the walk loop after touching the kernel.  Keeping only the block's
	// base here and adding threadIdx.x at every access, behind an empty asm so that the sum is not hoisted and spilled
	// itself, makes the slow form impossible but costs the dragon 0.5-1 % and gives the other scenes nothing reliable; not adopted.)
	uint32_t* lds;   // this thread's column
	uint32_t* ovf;   // [level][global thread], this thread's column
	uint32_t ovfStride;
	// The overflow accesses are written as inline asm on purpose: given two plain loads, one from LDS and one from global
	// memory, the compiler merges them into a single FLAT load through a selected pointer, which makes every pop slow.
	__device__ __forceinline__ void push(int sp, uint32_t v) {
#if MIRT_STATS
		atomicAdd(g_statPush, 1ull);
		if (OVF && sp >= NLDS) atomicAdd(g_statPush + 1, 1ull);
#endif
		if (!OVF || sp < NLDS) lds[sp * STRIDE] = v;
		else {
			uint32_t* p = ovf + (size_t)(sp - NLDS) * ovfStride;
			asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
		}
	}
	__device__ __forceinline__ uint32_t pop(int sp) {
		if (!OVF || sp < NLDS) return lds[sp * STRIDE];
		uint32_t v;
		const uint32_t* p = ovf + (size_t)(sp - NLDS) * ovfStride;
		asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
		return v;
	}
};

// One ray's walk of the primitive hierarchy, written as a resumable object so that a wave can hand a lane a new ray
// as soon as its old one is finished (k_trace).  limit >= 0: shadow query "is there a hit with dist < limit"
// (main.cpp:223-224); limit < 0: nearest hit under the reference's tie rules.
// Children are visited nearer-first and a subtree is skipped when its entry parameter exceeds `tcull`, a bound kept
// slightly beyond the best distance so far, so every candidate at a distance <= best is still visited.
template <int MODE, bool OVF>
struct Walker {
	RayCtx r;
	float limit;
	float tcull;
	Hit best;
	uint32_t cur;
	int sp;
	// MODE_EXACT: a hit counts only if a reference leaf holding the primitive is box-hit (membership()).  Checking that
	// at every candidate runs at ~6 % lane utilisation, so the walk normally defers it: it keeps the nearest candidate
	// unverified and the caller verifies the final answer once (verify()), many lanes at a time.  Only if that check
	// fails, or two different primitives tie in distance (the rank decides, Partition.h:33 / Partition.cpp:47), is the
	// ray walked again in strict mode, where every candidate is checked on the spot.  The two kinds of walk are separate
	// instantiations (STRICT) of the stepping functions, so the common one carries neither the membership code nor the
	// visiting rank through its loop.
	bool needStrict;

	__device__ __forceinline__ bool start(const DScene& S, f3 o, f3 d, float lim) {
		r = make_ray(o, d);
		limit = lim;
		needStrict = false;
		best.dist = __builtin_inff();
		best.ref = REF_NONE;
		best.pos = 0xffffffffu;
		tcull = lim >= 0.0f ? cull_bound(S, r, lim) : __builtin_inff();
		sp = 0;
		if (all_nan(d) || all_nan(o)) { // see DScene::nanHit
			best.ref = S.nanHit;
			best.dist = __builtin_nanf("");
			cur = BVH_DONE;
			return false;
		}
		cur = S.bvhRoot;
		float tn;
		const float4 lo = make_float4(S.bvhMin[0], S.bvhMin[1], S.bvhMin[2], 0.0f), hi = make_float4(S.bvhMax[0], S.bvhMax[1], S.bvhMax[2], 0.0f);
		if (cur == BVH_DONE || !box_filter(r, lo, hi, tcull, tn)) cur = BVH_DONE;
		return cur != BVH_DONE;
	}

	__device__ __forceinline__ bool at_node() const { return !(cur & BVH_LEAF); }

	// One interior node of the 4-wide hierarchy (layout in mirt_device.h): decode the four children's quantised boxes
	// straight into slab parameters (t = q * (scale/d) + (origin - o)/d, one FMA per plane), go to the nearest hit child
	// and push the other hit children, farthest first; none hit: pop.  Call only while at_node().
	// The quantised boxes contain the padded primitive bounds (lower bounds rounded down, upper bounds up), so the filter
	// stays conservative; see box_filter for the rounding argument.
	template <class STK>
	__device__ __forceinline__ void node_step(const DScene& S, STK& stk) {
		STAT_WAVE(S, 0); // node steps (wave iterations)
		STAT_LANE(S, 1); // node steps (lanes)
		const uint4* nd = (const uint4*)S.bvh + (size_t)cur * 4;
		// (The vector memory path takes about one cycle per lane and 16-byte request -- exp/ta/ta_bench.hip: 258 cycles for a
		// wave's four requests per lane, whether they hit in L1 or in L2 -- and with frames in flight the walk asks for 92 % of
		// that rate; yet it is not what holds the walk up: a FIFTH request per node step, waited for, to the same line or to one
		// 1 KB away, leaves the Stanford dragon's frame rate where it is (14.3 Grays/s, twice measured).  A 48-byte node would
		// buy nothing.)
		const uint4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
		// scale/d per axis: the node stores signed power-of-two exponents
		const float ax = __builtin_ldexpf(r.ie.x, (int)(q0.w << 24) >> 24);
		const float ay = __builtin_ldexpf(r.ie.y, (int)(q0.w << 16) >> 24);
		const float az = __builtin_ldexpf(r.ie.z, (int)(q0.w << 8) >> 24);
		const float bx = __builtin_fmaf(__uint_as_float(q0.x), r.ie.x, -r.oie.x);
		const float by = __builtin_fmaf(__uint_as_float(q0.y), r.ie.y, -r.oie.y);
		const float bz = __builtin_fmaf(__uint_as_float(q0.z), r.ie.z, -r.oie.z);
		uint32_t key[4], ref[4];
		ref[0] = q2.z; ref[1] = q2.w; ref[2] = q3.x; ref[3] = q3.y;
		// Which plane of an axis the ray meets first depends on the sign of its direction only (t = q * a + b is monotone
		// in q), so the four children's near planes are picked once per node, a whole word of bytes at a time, instead of
		// taking min and max of every pair of parameters.
		const uint32_t mx = (uint32_t)((int)__float_as_uint(r.ie.x) >> 31), my = (uint32_t)((int)__float_as_uint(r.ie.y) >> 31),
		               mz = (uint32_t)((int)__float_as_uint(r.ie.z) >> 31); // all ones: the ray goes down the axis
		const uint32_t nearX = (mx & q1.w) | (~mx & q1.x), farX = (mx & q1.x) | (~mx & q1.w);
		const uint32_t nearY = (my & q2.x) | (~my & q1.y), farY = (my & q1.y) | (~my & q2.x);
		const uint32_t nearZ = (mz & q2.y) | (~mz & q1.z), farZ = (mz & q1.z) | (~mz & q2.y);
		typedef float pk2 __attribute__((ext_vector_type(2))); // both planes of an axis in one packed FMA
		const pk2 pax = {ax, ax}, pay = {ay, ay}, paz = {az, az}, pbx = {bx, bx}, pby = {by, by}, pbz = {bz, bz};
		// entry no earlier than the ray's start, exit no later than the culling bound (which is never negative):
		// max(tmin, 0) <= min(tmax, tcull)  <=>  tmin <= tmax and tmax >= 0 and tmin <= tcull
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			const int sh = 8 * c;
			const pk2 qx = {(float)((nearX >> sh) & 0xffu), (float)((farX >> sh) & 0xffu)};
			const pk2 qy = {(float)((nearY >> sh) & 0xffu), (float)((farY >> sh) & 0xffu)};
			const pk2 qz = {(float)((nearZ >> sh) & 0xffu), (float)((farZ >> sh) & 0xffu)};
			const pk2 tx = __builtin_elementwise_fma(qx, pax, pbx), ty = __builtin_elementwise_fma(qy, pay, pby), tz = __builtin_elementwise_fma(qz, paz, pbz);
			const float tmin = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, 0.0f));
			const float tmax = fminf(fminf(tx.y, ty.y), fminf(tz.y, tcull));
			const bool hit = (tmin <= tmax) && ref[c] != BVH_DONE;
			// non-negative floats order like their bit patterns
			key[c] = hit ? __float_as_uint(tmin) : 0xffffffffu;
		}
#define MIRT_CSWAP(i, j) { const bool sw = key[j] < key[i]; const uint32_t ki = key[i], ri = ref[i]; \
		key[i] = sw ? key[j] : ki; ref[i] = sw ? ref[j] : ri; key[j] = sw ? ki : key[j]; ref[j] = sw ? ri : ref[j]; }
		// (all five comparators: with four -- middle pair left as it falls -- or three -- only the nearest child in its place --
		// the step is 6 and 12 instructions shorter and the Stanford dragon 2 % slower: the order of the pushes matters more)
		MIRT_CSWAP(0, 1) MIRT_CSWAP(2, 3) MIRT_CSWAP(0, 2) MIRT_CSWAP(1, 3) MIRT_CSWAP(1, 2)
#undef MIRT_CSWAP
		if (key[0] == 0xffffffffu) {
			cur = sp > 0 ? stk.pop(--sp) : BVH_DONE;
			return;
		}
		cur = ref[0];
		if (key[3] != 0xffffffffu) { stk.push(sp, ref[3]); ++sp; }
		if (key[2] != 0xffffffffu) { stk.push(sp, ref[2]); ++sp; }
		if (key[1] != 0xffffffffu) { stk.push(sp, ref[1]); ++sp; }
		// (Measured and dropped: asking for the child that will be popped next right here, with a load that lands in a corner of
		// LDS nobody reads (global_load_lds), so that it is on its way through the caches while the nearer subtree is walked:
		// the 10-million-triangle scene -3.5 %, the Stanford dragon -8 %.  A pushed child that the walk ends or culls before
		// it is popped was fetched for nothing, and every step issues one more request.)
	}

	// Tests the first record of leaf reference `lf` against the ray and updates the best hit.  Returns true if the walk
	// is over (a shadow query found its occluder, or a distance tie asks for the strict walk).
	template <bool STRICT>
	__device__ __forceinline__ bool test_first(const DScene& S, uint32_t lf) {
		STAT_WAVE(S, 2); // primitive tests (wave iterations)
		STAT_LANE(S, 3); // primitive tests (lanes)
		const uint32_t idx = lf & BVH_FIRST_MASK;
		const bool sph = (lf & BVH_SPHERES) != 0;
		const bool any = limit >= 0.0f;
		float dist;
		if (sph) {
			f3 dummy;
			dist = sphere_hit<false>(r, S.spheres + (size_t)idx * SPHERE_F4, dummy);
		}
		else {
			dist = tri_hit<MODE>(r, S.tris + (size_t)idx * TRI_F4, S.triX + (size_t)idx * 2, tcull);
		}
		if (!(dist >= 0.0f) || isinf(dist)) return false;
		const uint32_t ref = sph ? (PRIM_SPHERE | idx) : idx;
		const bool tie = !any && dist == best.dist && ref != best.ref;
		const bool cand = any ? (dist < limit) : (dist < best.dist || tie);
		if (!cand) return false;
		if (MODE == MODE_FAST && tie) {
			// Two primitives at the same distance (the shared edge of two triangles).  The reference takes the one it visits
			// first (Partition.h:33, Partition.cpp:47), which MIRT_TRACE_EXACT settles with the visiting ranks of the leaves.
			// Without them: the lower primitive number -- the order inside a leaf, so the reference's choice whenever one
			// leaf holds both, which is the usual case.
			const uint32_t idNew = sph ? S.sphInfo[idx].x : S.triInfo[idx].x;
			const uint32_t idOld = (best.ref & PRIM_SPHERE) ? S.sphInfo[best.ref & ~PRIM_SPHERE].x : S.triInfo[best.ref].x;
			if (idNew < idOld) best.ref = ref;
			return false;
		}
		uint32_t pos = 0;
		if (MODE == MODE_EXACT) {
			if (!STRICT) {
				if (tie) { needStrict = true; return true; }
			}
			else {
				STAT_WAVE(S, 4); // membership checks (wave)
				STAT_LANE(S, 5); // membership checks (lanes)
				const uint4 info = sph ? S.sphInfo[idx] : S.triInfo[idx];
				pos = membership(S, r, info.z, info.w);
				if (pos == REF_NONE) return false; // the reference's tree never reaches this primitive along this ray
			}
		}
		if (any) {
			best.dist = dist;
			best.ref = ref;
			if (STRICT) best.pos = pos;
			return true;
		}
		// Partition.h:33 (first in a leaf wins), Partition.cpp:47 (left subtree wins) == lower visiting rank wins
		// (without ranks, i.e. not STRICT, a tie has already ended the walk above, or -- MODE_FAST -- the first found stays)
		if (dist < best.dist || (STRICT && MODE == MODE_EXACT && pos < best.pos)) {
			best.dist = dist;
			best.ref = ref;
			if (STRICT) best.pos = pos;
			tcull = cull_bound(S, r, dist);
		}
		return false;
	}
	// a leaf reference without its first record, or 0 if that was the last one
	static __device__ __forceinline__ uint32_t leaf_rest(uint32_t lf) {
		return ((lf >> BVH_COUNT_SHIFT) & 7u) ? lf - (1u << BVH_COUNT_SHIFT) + 1u : 0u;
	}

	// One primitive of the current leaf (1..8 consecutive records of one kind).  Call only while cur is a leaf.
	// Afterwards cur is the rest of the leaf, or the next stack entry, or BVH_DONE.  Returns true while the same leaf goes on.
	template <bool STRICT, class STK>
	__device__ __forceinline__ bool leaf_step(const DScene& S, STK& stk) {
		if (test_first<STRICT>(S, cur)) { cur = BVH_DONE; sp = 0; return false; }
		const uint32_t rest = leaf_rest(cur);
		if (rest) { cur = rest; return true; }
		cur = sp > 0 ? stk.pop(--sp) : BVH_DONE;
		return false;
	}

	// One round: descend to the next leaf, test its primitives, pop.  Returns false when the walk is over.
	template <bool STRICT, class STK>
	__device__ __forceinline__ bool advance(const DScene& S, STK& stk) {
		STAT_WAVE(S, 6); // advance rounds (wave)
		STAT_LANE(S, 7); // advance rounds (lanes)
		while (at_node()) node_step(S, stk);
		if (cur == BVH_DONE) return false;
		while (leaf_step<STRICT>(S, stk)) {}
		return cur != BVH_DONE;
	}

	// After a deferred (not STRICT) walk has ended: is its answer final?  false = walk the ray again, STRICT.
	__device__ __forceinline__ bool verify(const DScene& S) const {
		if (MODE != MODE_EXACT) return true;
		if (needStrict) return false;
		if (best.ref == REF_NONE || best.dist != best.dist) return true; // (a NaN distance: the answer to a NaN direction, see start())
		STAT_WAVE(S, 4);
		STAT_LANE(S, 5);
		const uint4 info = (best.ref & PRIM_SPHERE) ? S.sphInfo[best.ref & ~PRIM_SPHERE] : S.triInfo[best.ref];
		return membership(S, r, info.z, info.w) != REF_NONE;
	}
};

// whole walk at once (test hooks and MODE_REFWALK dispatch)
template <int MODE, bool OVF, class STK>
__device__ __forceinline__ void trace_whole(const DScene& S, f3 o, f3 d, float limit, Hit& best, STK& stk) {
	if (MODE == MODE_REFWALK) {
		if (all_nan(d) || all_nan(o)) { // see DScene::nanHit
			best.ref = S.nanHit;
			best.dist = __builtin_nanf("");
			best.pos = 0;
			return;
		}
		const RayCtx r = make_ray(o, d);
		if (limit >= 0.0f) traverse_ref<true>(S, r, limit, best);
		else traverse_ref<false>(S, r, limit, best);
		return;
	}
	Walker<MODE == MODE_REFWALK ? MODE_EXACT : MODE, OVF> w;
	bool alive = w.start(S, o, d, limit);
	while (alive) alive = w.template advance<false>(S, stk);
	if (!w.verify(S)) {
		alive = w.start(S, o, d, limit);
		while (alive) alive = w.template advance<true>(S, stk);
	}
	best = w.best;
}

// Intersection::intersectNormal of the winning primitive (normalised by the Intersection ctor, SceneObjects.hpp:39)
__device__ __forceinline__ f3 hit_normal(const DScene& S, const RayCtx& r, uint32_t ref) {
	if (ref & PRIM_SPHERE) {
		f3 n;
		sphere_hit<true>(r, S.spheres + (size_t)(ref & ~PRIM_SPHERE) * SPHERE_F4, n);
		return normalize3(n);
	}
	return normalize3(F3(S.tris[(size_t)ref * TRI_F4]));
}
__device__ __forceinline__ uint4 info_of(const DScene& S, uint32_t ref) {
	return (ref & PRIM_SPHERE) ? S.sphInfo[ref & ~PRIM_SPHERE] : S.triInfo[ref];
}

// Camera::createRayToPixel(j + 0.5, i + 0.5, w, h) (Camera.cpp:15-24; basis and tan precomputed on the host with
// the same operations, they do not depend on the pixel)
__device__ __forceinline__ void primary_ray(const DScene& S, uint32_t px, uint32_t py, uint32_t width, uint32_t height, f3& o, f3& d) {
	const float i = (float)px + 0.5f, j = (float)py + 0.5f;
	const int w = (int)width, h = (int)height;
	const float alpha = S.cam.tanHalf * ((float)w / h) * ((i - (w / 2.0f)) / (w / 2.0f)); // :19
	const float beta = S.cam.tanHalf * (((h / 2.0f) - j) / (h / 2.0f));                      // :20
	o = F3(S.cam.eye);
	d = (F3(S.cam.u) * alpha + F3(S.cam.v) * beta) - F3(S.cam.w);                           // :21
}

// slot -> pixel.  A slot is (local tile, 8x8 sub-block, lane): 64 consecutive slots are one 8x8 pixel block, so a
// wave's primary rays are spatially coherent.  Returns false for padding pixels of edge tiles.
__device__ __forceinline__ uint32_t fast_div(uint32_t x, const FastDiv& f) {
	if (f.shift == 0xffffffffu) return x;
	const uint32_t q = __umulhi(x, f.magic);
	return (((x - q) >> 1) + q) >> f.shift;
}
__device__ __forceinline__ bool slot_to_pixel(const DFrame& F, uint32_t slot, uint32_t& px, uint32_t& py, uint32_t& inTile) {
	const uint32_t lt = fast_div(slot, F.divArea), i = slot - lt * F.divArea.d;
	const uint32_t sb = i >> 6, l = i & 63;
	const uint32_t sby = fast_div(sb, F.divSbx), sbxi = sb - sby * F.divSbx.d;
	const uint32_t tx = sbxi * 8 + (l & 7), ty = sby * 8 + (l >> 3);
	const uint32_t k = F.firstTile + lt * F.tileStride;
	const uint32_t ky = fast_div(k, F.divTilesX), kx = k - ky * F.divTilesX.d;
	px = kx * F.tileW + tx;
	py = ky * F.tileH + ty;
	inTile = ty * F.tileW + tx;
	return px < F.width && py < F.height;
}
__device__ __forceinline__ size_t out_offset(const DFrame& F, uint32_t slot, uint32_t px, uint32_t py, uint32_t inTile) {
	if (F.packedTiles) return ((size_t)fast_div(slot, F.divArea) * F.divArea.d + inTile) * 3;
	return ((size_t)py * F.width + px) * 3; // main.cpp:154-156
}

// append this wave's n collected slots (LDS) to a global list: one atomic, coalesced stores
__device__ __forceinline__ void flush_list(const uint32_t* lds, uint32_t n, uint32_t* list, uint32_t* counter) {
	const int lane = threadIdx.x & 63;
	uint32_t base = 0;
	if (lane == 0) base = atomicAdd(counter, n);
	base = __builtin_amdgcn_readfirstlane(base);
	for (uint32_t i = lane; i < n; i += 64) list[base + i] = lds[i];
}

// ---- wave-local supply of work items -------------------------------------------------------------------------------------
// Items 0..n-1 are handed out in blocks: a wave's first block is static (its index), later blocks come from one atomic
// cursor per launch.  The block size adapts to the launch: 256 items when there is plenty of work (one atomic per 256
// items keeps the cursor word far below the ~88 atomics/us one address sustains), down to 64 when there is little, so
// that a small launch spreads over many waves instead of queueing on a few.
// (Measured alternative: the items cut into 8 contiguous ranges, one per XCD -- workgroup i runs on XCD i % 8 -- each with
// a cursor of its own and stealing from the next range when empty, so that an XCD's L2 serves one part of the image: the
// uneven cost of the parts loses far more than the L2 gains, Stanford dragon -35 %, 10 M triangles -11 %.)
struct Supply {
	uint32_t* cursor;
	uint32_t n, cb, ce, stride0, blk;
	__device__ __forceinline__ Supply(uint32_t* c, uint32_t count) : cursor(c), n(count) {
		const uint32_t nWaves = gridDim.x * (BLOCK / WAVE);
		const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6));
		const uint32_t per = count / (2u * nWaves);
		blk = per >= 256u ? 256u : (per >= 128u ? 128u : 64u);
		stride0 = nWaves * blk;
		cb = wave * blk;
		ce = cb + blk;
	}
	__device__ __forceinline__ bool exhausted() const { return cb >= n; }
	// up to `want` consecutive items: returns how many, first index in base
	__device__ __forceinline__ uint32_t take(uint32_t want, uint32_t& base) {
		if (cb >= ce && cb < n) {
			uint32_t c = 0;
			if ((threadIdx.x & 63) == 0) c = atomicAdd(cursor, blk);
			cb = __builtin_amdgcn_readfirstlane(c) + stride0;
			ce = cb + blk;
		}
		if (cb >= n) { cb = n; ce = n; return 0; }
		const uint32_t end = ce < n ? ce : n;
		const uint32_t got = want < end - cb ? want : end - cb;
		base = cb;
		cb += got;
		return got;
	}
};

// How a wave shares its steps between lanes that stand at an interior node and lanes that stand at a leaf: the group
// with more (weighted) lanes goes, the other waits.  A primitive test costs more than a node step, hence the weights.
#ifndef MIRT_NODE_WEIGHT
#define MIRT_NODE_WEIGHT 1
#define MIRT_LEAF_WEIGHT 2
#endif

// Walks every lane's ray to its end.  One step per iteration for the whole wave: either every lane at an interior node
// takes a node step, or every lane at a leaf tests one primitive.  (Letting each lane run to its next leaf and through
// it keeps a wave waiting for its longest descent: measured lane utilisation 0.33 on bounce rays against 0.52 this way.)
template <bool EARLY, bool STRICT, int MODE, bool OVF, class STK>
__device__ __forceinline__ void walk_wave(const DScene& S, Walker<MODE, OVF>& w, STK& stk, uint32_t stopIdle = 65u) {
	// (Tried and dropped: letting a lane put the leaf it reaches into a small queue of its own and go on with its stack,
	// so that primitive steps would find more lanes ready -- primitive steps became more frequent and emptier instead.
	// Also tried and dropped, twice: setting the last few walking rays of a batch aside and walking them on later, 64 to a
	// wave again.  On the Stanford dragon 30 % of the node steps are taken by waves with at most 16 walking lanes
	// (scripts/walk_stats.py prints the histogram), and with the last <= 16 rays of every batch written to a pool (96 B:
	// ray, bound, best candidate, position, 11 stack entries) and a second launch per depth that walks the pool, the wave
	// iterations fall by 16 % and node-step lane utilisation rises from 0.57 to 0.68 -- but a quarter of all rays then
	// passes through the pool (0.6 GB per 4K frame on a memory path that is already 80 % busy, plus their set-up twice)
	// and the frame rate FALLS by 9 % (<= 8 rays set aside), 13 % (16), 19 % (32).  An earlier form with per-wave slices of
	// global memory and <= 12 rays lost the same way.)
	// A lane without a walk holds cur == BVH_DONE, so both questions are single signed comparisons of cur (node references
	// are non-negative, leaf references lie below -1) and the ballots come straight from the compare unit.
	// Returns when no lane walks any more or, with EARLY, as soon as at least stopIdle lanes do not (65: never).
	for (;;) {
		const int c = (int)w.cur;
		const bool wantNode = c >= 0, wantLeaf = c < -1;
		const uint32_t nNode = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(wantNode)), nLeaf = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(wantLeaf));
		if (nNode + nLeaf == 0u || (EARLY && 64u - (nNode + nLeaf) >= stopIdle)) break;
		// (Measured and dropped: letting the smaller group take its step in the same iteration whenever it has at least k
		// lanes -- nobody waits, but both steps are then issued for part of the wave each: Stanford dragon -5 % for k = 1 and
		// 12, -4 % for 24, -3 % for 40.)
		if (nNode * MIRT_NODE_WEIGHT >= nLeaf * MIRT_LEAF_WEIGHT) {
#if MIRT_STATS
			if (wantNode) STAT_WAVE(S, 8 + ((nNode + nLeaf - 1u) >> 3)); // histogram of node steps by walking lanes (at nodes or leaves), 8 per bucket
#endif
			if (wantNode) w.node_step(S, stk);
		}
		else if (wantLeaf) w.template leaf_step<STRICT>(S, stk);
	}
}

// walk_wave for a wave whose lanes need not keep to their own rays (k_frame, MODE_EXACT, the deferred walk only).
// A round lasts as long as its longest ray: 64 secondary rays of the Stanford dragon take 55 wave iterations, their mean
// ray 16.  Here a lane that has nothing of its own left -- no ray in this round, or a ray that ended without a hit, so that
// no result waits in its registers -- takes the BOTTOM entry of a walking lane's traversal stack (the farthest and largest
// subtree still to be visited) together with a copy of that lane's ray, and walks it.  The lanes of one ray form a group
// named after the ray's own lane ("owner"):
//   * the group's culling bound lives in s_cull[owner] (bit pattern of a non-negative float, lowered with an atomic min by
//     whoever finds a nearer candidate, read by all before every step); 0 = a shadow query has its answer, everybody stops;
//   * a stolen entry is overwritten with BVH_DONE: stolen entries are always the lowest ones of a stack, so a lane that
//     pops one has nothing left below it and ends as if its stack were empty;
//   * when all lanes are done, the helpers' candidates are folded into the owner's with an atomic min over
//     (distance bits, primitive) in LDS; two different primitives at the winning distance, or a tie seen by any lane of the
//     group, send the ray to the strict walk exactly as a tie inside one lane does.  The answer is the one the lane would
//     have found alone: the subtrees are disjoint, every candidate no farther than the best is still visited (the shared
//     bound is never below the bound of a lane walking alone at the same point), and the rule for ties is the same.
// An owner that has given work away keeps its ray (verify() and the hit record need it) and does not help others.
// grp: bits 0-5 owner, bit 6 "owner that gave work away", bits 8+ number of stolen entries at the bottom of the own stack.
// s_cull: 64 words of LDS of this wave; s_key: 64 x 8 bytes of LDS of this wave that nothing else uses once every walk
// of the wave has ended (the traversal stacks' own space).  `base`: the wave's lane-0 column of the traversal stack.
#ifndef MIRT_SHARE_MIN_FREE
#define MIRT_SHARE_MIN_FREE 16 // lanes without a walk before work is handed over
#define MIRT_SHARE_GAP 4       // wave iterations between two hand-overs at least
#endif
#ifndef MIRT_SHARE_LEVELS
#define MIRT_SHARE_LEVELS 4    // stack entries one lane hands over at a time at most (1..4)
#endif
#ifndef MIRT_SHARE_FIRST
#define MIRT_SHARE_FIRST 4     // wave iterations before the first hand-over of a round
#endif
template <int MODE, bool OVF, class STK>
__device__ __forceinline__ void walk_wave_share(const DScene& S, Walker<MODE, OVF>& w, STK& stk, uint32_t* base, uint32_t* s_cull, unsigned long long* s_key) {
	const uint32_t lane = threadIdx.x & 63u;
	const unsigned long long ltMask = (1ull << lane) - 1ull;
	uint32_t grp = lane;
	bool shared = false; // (uniform) some ray of the wave is walked by more than one lane
	int gap = MIRT_SHARE_GAP - MIRT_SHARE_FIRST; // (uniform)
	s_cull[lane] = __float_as_uint(w.tcull);
#if MIRT_TIMELINE >= 2
	uint32_t tlIters = 0, tlBusy = 0; // S.dbg[4..7]: the longest walk of a wave in iterations, all iterations, walks, walks of more than 200 iterations
	struct TlEnd { const DScene& S; uint32_t& it; uint32_t& busy; uint32_t lane; __device__ ~TlEnd() { if (lane == 0) { atomicMax(S.dbg + 4, (unsigned long long)it); atomicAdd(S.dbg + 5, (unsigned long long)it); atomicAdd(S.dbg + 6, 1ull); if (it > 200u) { atomicAdd(S.dbg + 7, 1ull); atomicAdd(S.dbg + 8, (unsigned long long)busy); } } } } tlEnd{S, tlIters, tlBusy, lane};
#endif
	for (;;) {
		// (asking for the group's bound here and using it after the step, so that the step's loads hide the LDS latency, was
		// slower: 13.9 against 14.25 Grays/s on the Stanford dragon)
		if (shared) {
			const uint32_t sc = s_cull[grp & 63u];
			if (sc == 0u) w.cur = BVH_DONE;
			else w.tcull = fminf(w.tcull, __uint_as_float(sc));
		}
		const int c = (int)w.cur;
		const bool wantNode = c >= 0, wantLeaf = c < -1;
		const uint32_t nNode = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(wantNode)), nLeaf = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(wantLeaf));
		if (nNode + nLeaf == 0u) break;
#if MIRT_TIMELINE >= 2
		++tlIters; tlBusy += nNode + nLeaf;
#endif
#if MIRT_TIMELINE >= 2 // (atomics on one word each per wave iteration: this build's times are many times the library's)
		{
			// idle lanes of this iteration by kind: holding a hit of their own / owners that gave work away / free to help
			const uint32_t nHold = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(c == -1 && (grp & 63u) == lane && w.best.ref != REF_NONE));
			const uint32_t nLent = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(c == -1 && (grp & 64u) && w.best.ref == REF_NONE));
			if (lane == 0) { atomicAdd(S.dbg + 8, (unsigned long long)nHold); atomicAdd(S.dbg + 9, (unsigned long long)nLent); atomicAdd(S.dbg + 10, (unsigned long long)(64u - nNode - nLeaf)); atomicAdd(S.dbg + 11, 1ull); }
		}
#endif
		++gap;
		if (64u - (nNode + nLeaf) >= MIRT_SHARE_MIN_FREE && gap >= MIRT_SHARE_GAP) {
			const bool free = c == -1 && w.best.ref == REF_NONE && !w.needStrict && !(grp & 64u);
			const bool gives = c != -1 && w.sp > (int)(grp >> 8) && (grp >> 8) < (uint32_t)STK::nlds;
			const unsigned long long freeM = __builtin_amdgcn_ballot_w64(free), giveM = __builtin_amdgcn_ballot_w64(gives);
			const uint32_t nF = (uint32_t)__popcll(freeM), nG = (uint32_t)__popcll(giveM);
			if (nF >= MIRT_SHARE_MIN_FREE / 2 && nG) {
#if MIRT_TIMELINE >= 2 // (atomics on one word each per wave iteration: this build's times are many times the library's)
				{
					const uint32_t nHold = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(c == -1 && w.best.ref != REF_NONE && (grp & 63u) == lane && !(grp & 64u)));
					if (lane == 0) { atomicAdd(S.dbg + 12, 1ull); atomicAdd(S.dbg + 13, (unsigned long long)(nF < nG ? nF : nG)); atomicAdd(S.dbg + 14, (unsigned long long)nHold); atomicAdd(S.dbg + 15, (unsigned long long)nF); }
				}
#endif
				gap = 0;
				shared = true;
				const uint32_t rankF = (uint32_t)__popcll(freeM & ltMask), rankG = (uint32_t)__popcll(giveM & ltMask);
				// the k-th giving lane's number goes to lane k (every lane sends to a lane of its own), the k-th free lane picks it up there
				const uint32_t dest = gives ? rankG : nG + (uint32_t)__popcll(~giveM & ltMask);
				const uint32_t packed = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest << 2), (int)lane);
				// with more free lanes than giving ones, a giving lane serves up to MIRT_SHARE_LEVELS of them: its lowest entry, the
				// one above it, ... as far as its stack goes
				const uint32_t level = (uint32_t)(rankF >= nG) + (uint32_t)(rankF >= 2u * nG) + (uint32_t)(rankF >= 3u * nG);
				const uint32_t from = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((rankF - level * nG) << 2), (int)packed);
				const bool asks = free && rankF < MIRT_SHARE_LEVELS * nG;
				const uint32_t asked = (uint32_t)(nF > rankG) + (uint32_t)(nF > rankG + nG && MIRT_SHARE_LEVELS > 1) + (uint32_t)(nF > rankG + 2u * nG && MIRT_SHARE_LEVELS > 2) +
				                       (uint32_t)(nF > rankG + 3u * nG && MIRT_SHARE_LEVELS > 3);
				const uint32_t top = (uint32_t)w.sp < (uint32_t)STK::nlds ? (uint32_t)w.sp : (uint32_t)STK::nlds; // (only entries in LDS are handed over)
				const uint32_t have = gives ? top - (grp >> 8) : 0u;
				const uint32_t lost = asked < have ? asked : have; // entries this (giving) lane hands over
				const bool robbed = gives && lost != 0u;
				// the giving lanes bring the group's bound up to date first
				if (robbed) atomicMin(s_cull + (grp & 63u), __float_as_uint(w.tcull));
				const int src = (int)((asks ? from : lane) << 2);
#define MIRT_PULL(x) __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)__float_as_uint(x)))
				const uint32_t gs = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(grp | (top << 16)));
				const uint32_t g = gs & 0xffffu;
				const bool takes = asks && (g >> 8) + level < (gs >> 16);
				const float ox = MIRT_PULL(w.r.o.x), oy = MIRT_PULL(w.r.o.y), oz = MIRT_PULL(w.r.o.z);
				const float dx = MIRT_PULL(w.r.d.x), dy = MIRT_PULL(w.r.d.y), dz = MIRT_PULL(w.r.d.z);
				const float ix = MIRT_PULL(w.r.ie.x), iy = MIRT_PULL(w.r.ie.y), iz = MIRT_PULL(w.r.ie.z);
				const float px = MIRT_PULL(w.r.oie.x), py = MIRT_PULL(w.r.oie.y), pz = MIRT_PULL(w.r.oie.z);
				const float il = MIRT_PULL(w.r.invLen), lm = MIRT_PULL(w.limit);
#undef MIRT_PULL
				if (takes) {
					uint32_t* at = base + (size_t)((g >> 8) + level) * STK::stride + from;
					w.cur = *at;
					*at = BVH_DONE;
					w.sp = 0;
					// (all thirteen words are fetched: setting the reciprocals up again from o and d instead costs registers in the
					// whole loop -- 12.2 against 14.3 Grays/s)
					w.r.o = F3(ox, oy, oz); w.r.d = F3(dx, dy, dz); w.r.ie = F3(ix, iy, iz); w.r.oie = F3(px, py, pz);
					w.r.invLen = il;
					w.limit = lm;
					grp = g & 63u;
					w.tcull = __uint_as_float(s_cull[grp]);
				}
				if (robbed) grp = (grp + 256u * lost) | ((grp & 63u) == lane ? 64u : 0u);
				continue;
			}
		}
		if (nNode * MIRT_NODE_WEIGHT >= nLeaf * MIRT_LEAF_WEIGHT) {
			if (wantNode) w.node_step(S, stk);
		}
		else if (wantLeaf) {
			const float t0 = w.tcull;
			w.template leaf_step<false>(S, stk);
			if (shared) {
				if (w.limit >= 0.0f) { if (w.best.ref != REF_NONE) s_cull[grp & 63u] = 0u; }
				else if (w.tcull < t0) atomicMin(s_cull + (grp & 63u), __float_as_uint(w.tcull));
			}
		}
	}
	if (shared) {
		// fold the helpers' candidates into the owners'
		const bool helper = (grp & 63u) != lane;
		const unsigned long long key = ((unsigned long long)__float_as_uint(w.best.dist) << 32) | w.best.ref;
		s_key[lane] = helper ? ~0ull : key;
		s_cull[lane] = (!helper && w.needStrict) ? 1u : 0u;
		if (helper && w.best.ref != REF_NONE) atomicMin(s_key + (grp & 63u), key);
		const unsigned long long fin = s_key[grp & 63u];
		const bool tie = w.best.ref != REF_NONE && (uint32_t)(key >> 32) == (uint32_t)(fin >> 32) && (uint32_t)key != (uint32_t)fin;
		if (helper && (tie || w.needStrict)) s_cull[grp & 63u] = 1u;
		if (!helper && tie) s_cull[lane] = 1u;
		if (helper) {
			w.best.ref = REF_NONE;
			w.best.dist = __builtin_inff();
			w.needStrict = false;
		}
		else {
			w.best.dist = __uint_as_float((uint32_t)(fin >> 32));
			w.best.ref = (uint32_t)fin;
			w.needStrict = s_cull[lane] != 0u;
		}
	}
}

// ---- the tracing kernel: every ray of one depth in one launch ---------------------------------------------------------------
// Per-path data (hit records, shadow answers, light sums) is indexed by pixel slot: a pixel has one path (main.cpp:176-201
// recurses once per hit), so records are never moved.  What a depth works on is the list F.active[parity] of the slots
// whose path hit something at the previous depth; a wave collects the slots it finds hits for in LDS and appends them to
// the next list 192+ at a time (one atomic per flush).
//
// PRIMARY (depth 0): items are pixel slots; the ray comes from the camera (main.cpp:152).
// depth d >= 1: with n = hits of depth d-1, items [0, n*nLights) are the shadow queries of those hits -- ray
//   (P, normalize(L)), "is anything closer than |L|" (main.cpp:222-224) -- and items from n*nLights on are their
//   reflection rays (P, R) (main.cpp:188-190); both are derived here from the hit record of depth d-1, so no ray is ever
//   stored.  Results: a hit record for depth d (nearest-hit rays), or the occluder in F.shRes (shadow queries; the
//   SHADOW_MAP view asks for the nearest occluder, main.cpp:250-252).
struct TraceItems {
	uint32_t nShadowItems;
	const uint32_t* prevList;
	size_t rd, wr;
	int depth;
};

// The ray of work item `item`.  Returns false for a padding pixel of an edge tile (no ray).
template <bool PRIMARY>
__device__ __forceinline__ bool item_ray(const DScene& S, const DFrame& F, const TraceItems& T, uint32_t item, f3& o, f3& d, float& lim, uint32_t& tag, uint32_t& nShadowTraced) {
	lim = -1.0f;
	if (PRIMARY) {
		uint32_t px, py, inTile;
		const bool live = slot_to_pixel(F, item, px, py, inTile);
		tag = item;
		if (!live && !F.preCleared) F.pathDepth[item] = 0; // no path
		primary_ray(S, px, py, F.width, F.height, o, d);
		return live;
	}
	const bool shadow = item < T.nShadowItems;
	// consecutive shadow items = one hit's queries towards all lights: lanes that share an origin share the first part of
	// their walks.  (Measured alternative: 64 neighbouring hits towards ONE light per wave -- 4 % slower with 16 lights.)
	const uint32_t h = shadow ? item / S.nLights : item - T.nShadowItems;
	const uint32_t slot = T.prevList[h];
	const float4 ho = F.hitO[T.rd + slot], hd = F.hitD[T.rd + slot];
	const f3 po = F3(ho), pd = F3(hd);
	o = po + normalize3(pd) * hd.w;                               // P, Camera.cpp:26-28
	if (shadow) {
		const uint32_t li = item - h * S.nLights;
		const float4 ll = S.lights[li * 2];
		const f3 L = ll.w > 0.0f ? F3(ll) - o : F3(ll);           // main.cpp:213-219
		d = normalize3(L);                                        // :222
		const float dl = length3(L);
		lim = F.debug == D_SHADOW_MAP ? -2.0f : dl;               // :224
		tag = slot * S.nLights + li;
		// The answer of this query decides whether lc * atten * (diffuse * kd + specular * ks) is added (main.cpp:224-247).
		// If kd and ks -- evaluated here with k_shade's own operations on the same values -- are both exact zeros and no
		// factor can be inf or NaN (S.shadowCull, finite attenuation), the sum is the same either way: the query is
		// answered "not occluded" without a walk.  Such rays stay in the reference's count (mirt_stats.shadow_rays) and
		// are left out of shadow_rays_traced.
		if (S.shadowCull && F.debug == D_NONE) {
			const f3 N = F3(F.hitN[T.rd + slot]);
			const bool kd0 = !(F.features & F_DIFFUSE) || dot3(normalize3(N), normalize3(L)) <= 0.0f;    // std::max(x, 0.0f) is a zero; a NaN is not
			bool ks0 = !(F.features & F_SPECULAR);
			if (!ks0) {
				const f3 H = normalize3(d + normalize3(po - o));      // main.cpp:226-227
				ks0 = dot3(H, N) <= 0.0f;                             // pow(0, shininess > 0) == 0; a NaN stays a NaN
			}
			const float atten = ll.w > 0.0f ? 1.0f / (S.atten[0] + S.atten[1] * dl + S.atten[2] * (dl * dl)) : 1.0f; // :266-268
			if (kd0 && ks0 && fabsf(atten) <= 1e15f) {
				F.shRes[tag] = make_uint2(REF_NONE, __float_as_uint(__builtin_inff()));
				return false;
			}
		}
		++nShadowTraced;
	}
	else {
		const f3 N = F3(F.hitN[T.rd + slot]);
		d = normalize3(pd - N * (2.0f * dot3(pd, N)));            // main.cpp:188
		tag = slot;
	}
	return true;
}

// Stores the answer of a finished ray.  Returns true if the path goes on (a nearest-hit ray that hit something).
template <bool PRIMARY>
__device__ __forceinline__ bool item_result(const DScene& S, const DFrame& F, const TraceItems& T, uint8_t* __restrict__ out, const RayCtx& r, const Hit& best, float limit, uint32_t tag) {
	const bool hit = best.ref != REF_NONE;
	if (limit != -1.0f) {
		F.shRes[tag] = make_uint2(best.ref, __float_as_uint(best.dist));
		return false;
	}
	if (PRIMARY) {
		// background is black (main.cpp:124,180); PRIMARY_INTERSECTION_MAP paints hits pure red (main.cpp:183-185)
		const bool red = hit && F.debug == D_PRIMARY_MAP;
		if (red || (!hit && !F.preCleared)) {
			uint32_t px, py, inTile;
			slot_to_pixel(F, tag, px, py, inTile);
			const size_t k = out_offset(F, tag, px, py, inTile);
			out[k] = 0;
			out[k + 1] = 0;
			out[k + 2] = hit ? 255 : 0;
		}
		const bool goesOn = hit && !red;
		if (goesOn || !F.preCleared) F.pathDepth[tag] = goesOn ? 1 : 0;
	}
	if (!hit) return false;
	const f3 n = hit_normal(S, r, best.ref);
	F.hitO[T.wr + tag] = make_float4(r.o.x, r.o.y, r.o.z, 0.0f);
	F.hitD[T.wr + tag] = make_float4(r.d.x, r.d.y, r.d.z, best.dist);
	F.hitN[T.wr + tag] = make_float4(n.x, n.y, n.z, __uint_as_float(best.ref));
	if (!PRIMARY) F.pathDepth[tag] = (uint8_t)(T.depth + 1);
	return true;
}

// REFILL = false: a wave takes 64 items, walks all of them to the end, stores the results and takes the next 64.
// REFILL = true:  a lane whose ray is finished parks its result; once F.refillMin lanes are parked they store their
//   results together and receive new rays, so the wave does not idle behind its longest ray.
// Which one is faster depends on the scene: storing results and setting up rays for a few lanes at a time costs more
// than it saves while walks are short (Stanford dragon: 7 node steps per ray); with long walks (10 M triangles) it pays.
template <int MODE, bool OVF, bool PRIMARY, bool REFILL>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(MODE == MODE_REFWALK ? 1 : (REFILL ? MIRT_REFILL_WAVES : MIRT_BATCH_WAVES)))) void k_trace(DScene S, DFrame F, uint8_t* __restrict__ out, int depth, int launch, int spawn) {
	constexpr int NLDS = REFILL ? STACK_LDS_REFILL : STACK_LDS;
	__shared__ uint32_t s_stack[NLDS * BLOCK];
	__shared__ uint32_t s_list[BLOCK / WAVE][256];
	LdsStack<OVF, NLDS> stk;
	stk.lds = s_stack + threadIdx.x;
	stk.ovf = OVF ? F.ovf + ((size_t)blockIdx.x * BLOCK + threadIdx.x) : nullptr;
	stk.ovfStride = F.ovfStride;
	const int lane = threadIdx.x & 63;
	const unsigned long long ltMask = (1ull << lane) - 1ull;
	// hit records: read those of depth-1, write those of depth
	TraceItems T;
	T.rd = (size_t)((depth + 1) & 1) * F.nSlots;
	T.wr = (size_t)(depth & 1) * F.nSlots;
	T.depth = depth;
	const uint32_t nPrev = PRIMARY ? 0u : F.counters[CTR_NHITS(depth - 1)];
	T.nShadowItems = (F.features & F_SHADOWS) ? nPrev * S.nLights : 0u;
	T.prevList = F.active + T.rd;
	uint32_t* nextList = F.active + T.wr;
	uint32_t* myList = s_list[threadIdx.x >> 6];
	uint32_t nbuf = 0;
	const uint32_t nTotal = PRIMARY ? F.nSlots : T.nShadowItems + (spawn ? nPrev : 0u);
	Supply sup(F.counters + CTR_CURSOR(launch), nTotal);
	Walker<MODE, OVF> w;
	uint32_t nShadowTraced = 0;
	uint32_t tag = 0;
	bool hasRay = false; // the lane holds a ray: walking (w.cur != BVH_DONE) or finished with its result not stored yet
	w.cur = BVH_DONE;    // no walk yet
	for (;;) {
		bool parked;     // finished rays whose results are stored in this round
		if (!REFILL) {
			uint32_t base = 0;
			const uint32_t got = sup.take(64u, base);
			if (got == 0) break;
			w.cur = BVH_DONE;
			parked = false;
			if ((uint32_t)lane < got) {
				f3 o = F3(0, 0, 0), d = F3(0, 0, 1);
				float lim;
				parked = item_ray<PRIMARY>(S, F, T, base + (uint32_t)lane, o, d, lim, tag, nShadowTraced);
				if (parked) {
					if (MODE == MODE_REFWALK) {
						// the reference's own tree, walked in one go (fallback / cross-check mode)
						w.r = make_ray(o, d);
						w.limit = lim;
						trace_whole<MODE_REFWALK, OVF>(S, o, d, lim, w.best, stk);
					}
					else w.start(S, o, d, lim); // false: missed the whole scene, the result is "no hit"
				}
			}
			if (MODE != MODE_REFWALK) {
				walk_wave<false, false>(S, w, stk);
				// deferred check of the final answers; the rare ray that fails it is walked again, strictly
				if (MODE == MODE_EXACT) {
					const bool redo = parked && !w.verify(S);
					if (__builtin_amdgcn_ballot_w64(redo)) {
						if (redo) w.start(S, w.r.o, w.r.d, w.limit);
						walk_wave<false, true>(S, w, stk);
					}
				}
			}
		}
		else {
			// walk until enough lanes are idle to make handing out new rays worth a pass through the code below; when nothing
			// is left to hand out, until the last rays of the wave have finished
			walk_wave<true, false>(S, w, stk, sup.exhausted() ? 65u : F.refillMin);
			parked = hasRay && w.cur == BVH_DONE;
			// parked lanes make their answer final; the rare ray that fails the check is walked again, strictly, to its end
			// while the lanes that are still on their way wait
			if (MODE == MODE_EXACT) {
				const bool redo = parked && !w.verify(S);
				if (__builtin_amdgcn_ballot_w64(redo)) {
					const uint32_t waiting = w.cur;
					if (redo) w.start(S, w.r.o, w.r.d, w.limit);
					else w.cur = BVH_DONE;
					walk_wave<false, true>(S, w, stk);
					if (!redo) w.cur = waiting;
				}
			}
			if (parked) hasRay = false;
		}
		// finished rays store their results; slots with a new hit go to this wave's LDS list, which is appended to the
		// next depth's list in large pieces
		bool newHit = false;
		if (parked) newHit = item_result<PRIMARY>(S, F, T, out, w.r, w.best, w.limit, tag);
		const unsigned long long hm = __builtin_amdgcn_ballot_w64(newHit);
		if (hm) {
			if (newHit) myList[nbuf + (uint32_t)__popcll(hm & ltMask)] = tag;
			nbuf += (uint32_t)__popcll(hm);
			if (nbuf >= 192u) { flush_list(myList, nbuf, nextList, F.counters + CTR_NHITS(depth)); nbuf = 0; }
		}
		if (REFILL) {
			// idle lanes take new items
			const unsigned long long idle = __builtin_amdgcn_ballot_w64(!hasRay);
			uint32_t base = 0;
			const uint32_t got = sup.take((uint32_t)__popcll(idle), base);
			const uint32_t rank = (uint32_t)__popcll(idle & ltMask);
			if (!hasRay && rank < got) {
				f3 o = F3(0, 0, 0), d = F3(0, 0, 1);
				float lim;
				if (item_ray<PRIMARY>(S, F, T, base + rank, o, d, lim, tag, nShadowTraced)) {
					hasRay = true;
					w.start(S, o, d, lim); // false (missed the whole scene): cur == BVH_DONE, the next round stores "no hit"
				}
			}
			if (got == 0 && __builtin_amdgcn_ballot_w64(hasRay) == 0ull) break;
		}
	}
	if (nbuf) flush_list(myList, nbuf, nextList, F.counters + CTR_NHITS(depth));
	// statistics: one atomic per wave
	for (int off = 32; off > 0; off >>= 1) nShadowTraced += __shfl_down(nShadowTraced, off);
	if (lane == 0 && nShadowTraced) atomicAdd(F.counters + CTR_NSHADOW(depth - 1), nShadowTraced);
}

// One light's turn of the loop in calculateLightingColor (main.cpp:208-253): adds the light's term to `sum`.
// o = origin of the ray that hit, P = the hit point, N = the hit's normal, mat = the hit primitive's material record;
// occRef/occDist = answer of the light's shadow query (REF_NONE: nothing in the way).
__device__ __forceinline__ void shade_light(const DScene& S, uint32_t features, uint32_t debug, f3 o, f3 P, f3 N, const float4* __restrict__ mat, uint32_t li,
                                            bool shadows, uint32_t occRef, float occDist, f3& sum) {
	const float4 ll = S.lights[li * 2], lc = S.lights[li * 2 + 1];
	f3 L;
	float atten;
	if (ll.w > 0.0f) {                                              // :213
		L = F3(ll) - P;                                             // :214
		const float dl = length3(L);                                // :215
		// :266-268; glm::pow(distance, 2.0f) is dl * dl in the reference's build: g++ folds a power with the
		// constant exponent 2 into a multiplication (one mulss in the machine code of calculateAttenuation)
		atten = 1.0f / (S.atten[0] + S.atten[1] * dl + S.atten[2] * (dl * dl));
	}
	else {
		L = F3(ll);                                                 // :219
		atten = 1.0f;
	}
	const f3 Ln = normalize3(L);                                    // :222
	// shadow query: any hit closer than |L| was reported; SHADOW_MAP view: nearest hit, compared here (main.cpp:224)
	const bool lit = !shadows || (occRef == REF_NONE) || (debug == D_SHADOW_MAP && occDist >= length3(L));
	if (lit) {
		// std::max(x, 0.0f) is (x < 0.0f) ? 0.0f : x: a NaN (zero-length L or N) stays a NaN, fmaxf would drop it
		const float kdRaw = dot3(normalize3(N), normalize3(L));
		const float kd = (kdRaw < 0.0f) ? 0.0f : kdRaw;                      // :225, :258-260
		const f3 eye = o - P;                                                // :226
		const f3 H = normalize3(Ln + normalize3(eye));                       // :227
		const float ksRaw = dot3(H, N);
		const float4 mDiff = mat[0], mSpec = mat[1];
		const float ks = ref_powf((ksRaw < 0.0f) ? 0.0f : ksRaw, mDiff.w);   // :262-264, shininess
		if (debug == D_DIFFUSE_INTENSITY) sum = sum + F3(kd, kd, kd);
		else if (debug == D_SPECULAR_INTENSITY) sum = sum + F3(ks, ks, ks);
		else if (debug == D_NORMAL_MAP) sum = sum + N;
		else if (debug == D_LIGHT_DIR_MAP) sum = sum + H;
		else {
			if (features & F_DIFFUSE) sum = sum + ((F3(mDiff) * atten) * kd) * F3(lc);  // :243
			if (features & F_SPECULAR) sum = sum + ((F3(mSpec) * atten) * ks) * F3(lc); // :246
		}
	}
	else if (debug == D_SHADOW_MAP) {
		const float4* om = S.materials + (size_t)info_of(S, occRef).y * 4;
		sum = sum + F3(om[0]);                                      // :250-252
	}
}

// ---- once the shadow queries are answered: calculateLightingColor (main.cpp:203-256) for every hit of one depth ----------------
__global__ __launch_bounds__(BLOCK) void k_shade(DScene S, DFrame F, int depth) {
	const size_t rd = (size_t)(depth & 1) * F.nSlots;
	const bool shadows = (F.features & F_SHADOWS) != 0;
	const uint32_t n = F.counters[CTR_NHITS(depth)];
	const uint32_t* list = F.active + rd;
	for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
		const uint32_t slot = list[i];
		const float4 ho = F.hitO[rd + slot], hd = F.hitD[rd + slot], hn = F.hitN[rd + slot];
		const f3 o = F3(ho), d = F3(hd), N = F3(hn);
		const uint32_t ref = __float_as_uint(hn.w);
		const float4* mat = S.materials + (size_t)info_of(S, ref).y * 4;
		const float4 mSpec = mat[1], mEmis = mat[2], mAmb = mat[3];
		const f3 P = o + normalize3(d) * hd.w;                               // Camera.cpp:26-28
		f3 sum = F3(mAmb) + F3(mEmis);                                      // main.cpp:204
		for (uint32_t li = 0; li < S.nLights; ++li) {                       // :208
			uint32_t occRef = REF_NONE;
			float occDist = __builtin_inff();
			if (shadows) {
				const uint2 res = F.shRes[(size_t)slot * S.nLights + li];
				occRef = res.x;
				occDist = __uint_as_float(res.y);
			}
			shade_light(S, F.features, F.debug, o, P, N, mat, li, shadows, occRef, occDist, sum);
		}
		const size_t k = (size_t)depth * F.nSlots + slot;
		F.litL[k] = make_float4(sum.x, sum.y, sum.z, 0.0f);
		F.litS[k] = make_float4(mSpec.x, mSpec.y, mSpec.z, 0.0f);
	}
}

__device__ __forceinline__ uint8_t quantise(float c) { // Color.cpp:13-23
	const float s = c * 255.0f;
	// x86-64 (int) conversion of NaN / out-of-range yields INT_MIN, which the clamp turns into 0
	if (!(s >= -2147483648.0f && s < 2147483648.0f)) return 0;
	int v = (int)s;
	v = v < 0 ? 0 : v;
	v = v > 255 ? 255 : v;
	return (uint8_t)v;
}

// ---- fold the per-depth results back to front: C = L_k + S_k * C (main.cpp:190) and quantise ---------------------------
__global__ __launch_bounds__(BLOCK) void k_resolve(DFrame F, uint8_t* __restrict__ out) {
	for (uint32_t slot = blockIdx.x * BLOCK + threadIdx.x; slot < F.nSlots; slot += gridDim.x * BLOCK) {
		const int cnt = F.pathDepth[slot];
		if (cnt == 0) continue; // the tracing kernel already wrote the background
		uint32_t px, py, inTile;
		if (!slot_to_pixel(F, slot, px, py, inTile)) continue;
		f3 C = F3(0, 0, 0); // Color() past the last bounce (main.cpp:199) or the black background (main.cpp:124)
		for (int k = cnt - 1; k >= 0; --k) {
			const float4 L = F.litL[(size_t)k * F.nSlots + slot];
			const float4 Sp = F.litS[(size_t)k * F.nSlots + slot];
			if ((F.features & F_REFLECTIONS) || k > 0) C = F3(L) + F3(Sp) * C;
			else C = F3(L); // main.cpp:193
		}
		const size_t o = out_offset(F, slot, px, py, inTile);
		out[o] = quantise(C.z);     // B  main.cpp:154
		out[o + 1] = quantise(C.y); // G
		out[o + 2] = quantise(C.x); // R
	}
}

// ---- the whole frame in ONE launch --------------------------------------------------------------------------------------
// k_trace / k_shade / k_resolve above run a frame as one launch per depth: depth d+1 starts when depth d has ended, and
// the deep bounces -- a few thousand rays each -- leave the GPU nearly empty for most of a frame's duration.  k_frame is
// the same arithmetic as a data flow: a path's next ray starts as soon as the hit it leaves from exists.
//
// Work never fans out: a pixel slot yields at most one hit, and a hit at most one further hit (main.cpp:176-201 recurses
// once), so a wave keeps what it produces and no wave ever waits for another.  A wave
//   * takes pixel slots (its own first block, then blocks from the launch's cursor), 64 to a batch, walks their camera
//     rays, and puts the slots that hit into its list in LDS (entry = slot | depth << 27);
//   * whenever the list holds 64 entries it works on those before taking new slots (at the end, on what is left).
//     Working on a hit is calculateLightingColor and the reflection of computePixelColor (main.cpp:187-190) in one go,
//     all lanes in step: for every light in file order one batch of shadow queries, each lane adding the light's term to
//     its own sum right after the walk (shade_light: the operations and the order of k_shade); then one batch of
//     reflection rays, whose hits go back into the list with depth + 1; a path that ends (reflections off, a miss, or
//     the last depth) is folded back to front over its stored light sums, C = L_k + S_k C (k_resolve's loop), and its
//     pixel is written.
// Per-path data: one hit record per slot (the record of depth d is dead once its lane has written the record of depth
// d+1 over it) and one light sum per slot and depth; both are written and read by the same wave only.
//
// The end of a wave's work: when the slots have run out it works on what is left in its list, however few hits that is,
// round after round with most lanes idle.  Pooling such left-overs was measured twice and dropped: among the four waves
// of a block (in LDS; a wave leaves its hits to those still at work) the frame rate FELL by 7 % and a frame alone took
// 11 % longer, because the left-overs of four waves then pass through one wave, one batch after the other; in one pool
// for the whole launch (global memory, compare-and-swap on a cursor) thousands of waves polling two words made a 4K
// frame take 70 ms.  A third form needs no pool at all -- a wave that has run out of slots leaves its last hits (one word
// each: the records lie in the frame's arrays) in a list and ends, and one to four follow-up launches of the same kernel
// take the list 64 entries at a time: every follow-up launch starts when the slowest wave of the launch before it has
// ended, and the frame rate fell by 6 % (one follow-up) to 9 % (four), a frame alone from 1.8 to 2.1-2.3 ms.
// Nor does it help to hand out first the batches that can see the scene at all (the screen rectangle of its bounding box):
// a frame alone ends no earlier, and with frames in flight the rate falls by 9 % -- the batches that cost nothing are what
// lets a wave gather its hits while its neighbours do the same.
// At the end of a frame, waves working side by side on little beat fewer waves working on more.
#define FRAME_SLOT_MASK ((1u << FRAME_SLOT_BITS) - 1u)
#define FRAME_RECORDS 128u // hit records of a wave: fewer than 128 paths are under way in a wave at any time (its list + its batch)
#define FRAME_EMPTY 0xffffffffu

template <int MODE, bool OVF, bool SHARE, bool RECORDS>
__global__ __launch_bounds__(FRAME_BLOCK) __attribute__((amdgpu_waves_per_eu(MIRT_FRAME_WAVES))) void k_frame(DScene S, DFrame F, uint8_t* __restrict__ out) {
	constexpr int NLDS = RECORDS ? STACK_LDS_FRAME : STACK_LDS; // (the record tables take the place of one stack entry)
	__shared__ __align__(8) uint32_t s_stack[NLDS * FRAME_BLOCK];
	__shared__ uint32_t s_list[FRAME_BLOCK / WAVE][128];
	__shared__ uint8_t s_free[FRAME_BLOCK / WAVE][FRAME_RECORDS];
	__shared__ uint32_t s_slotOf[FRAME_BLOCK / WAVE][FRAME_RECORDS]; // [record of the wave] the pixel slot of the path that holds it
	LdsStack<OVF, NLDS, FRAME_BLOCK> stk;
	stk.lds = s_stack + threadIdx.x;
	stk.ovf = OVF ? F.ovf + ((size_t)blockIdx.x * FRAME_BLOCK + threadIdx.x) : nullptr;
	stk.ovfStride = F.ovfStride;
	const int lane = threadIdx.x & 63;
	const unsigned long long ltMask = (1ull << lane) - 1ull;
	uint32_t* myList = s_list[threadIdx.x >> 6];
	uint32_t nbuf = 0;            // entries in the list: fewer than 64 before a batch, fewer than 128 after
	// Where per-path data lives (F.hitO/hitD/hitN[i], F.litL[depth][i]).  RECORDS = false: i = the path's pixel slot -- the
	// arrays are as long as the image has slots (1.2 GB for 3840x2160 at maxdepth 5), nothing has to be managed, and the list
	// entries hold the slot.  RECORDS = true: i = one of FRAME_RECORDS hit records of this wave, taken when a camera ray hits and
	// given back when its path ends (free records: a stack of numbers in LDS; the path's pixel slot waits in LDS too), so a
	// frame's working memory is sized by the waves of the launch (136 MB at maxdepth 5) whatever the image -- for callers who
	// are short of memory and for the entry point that allocates nothing.  It costs 3 % of the frame rate (Stanford dragon
	// 13.1 against 13.5 Grays/s on the same machine: about a hundred more instructions per round of 64 rays, and a frame has
	// 200 000 rounds), which is why it is not the default.
	uint8_t* myFree = s_free[threadIdx.x >> 6];
	uint32_t* slotOf = s_slotOf[threadIdx.x >> 6];
	uint32_t nFree = FRAME_RECORDS; // (uniform)
	if (RECORDS) {
		myFree[lane] = (uint8_t)lane;
		myFree[lane + 64] = (uint8_t)(lane + 64);
	}
	const bool reflections = (F.features & F_REFLECTIONS) != 0;
	const bool shadows = (F.features & F_SHADOWS) != 0;
	const uint32_t last = reflections ? (uint32_t)S.maxDepth : 0u; // deepest hit a path can have (main.cpp:177)
	// Pixel slots come 64 at a time (one batch: the finer the grain, the later a wave can still be given the next batch
	// instead of sitting on a long queue of its own).  A wave's first batch is its own; later ones come from CTR_F_NCUR
	// counters that hand out the batch numbers in turn -- one address sustains only ~88 atomics per microsecond, a 4K frame has
	// 130 000 batches.  A wave starts at one counter and moves on to the next when that one has run out.
	// (Measured and dropped: giving the waves of one XCD -- one L2 -- a band of the image of their own, by the block's
	// place in the dispatcher's round or by the hardware's XCD number, instead of every eighth batch.  The L2 hit rate of a
	// Stanford dragon frame rises from 0.69 to 0.74 and the HBM traffic falls by 18 %, the frame rate does not move, a
	// frame alone ends 3 % later, and a 10-million-triangle scene loses 8 %.)
	const uint32_t nWaves = gridDim.x * (FRAME_BLOCK / WAVE);
	const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (FRAME_BLOCK / WAVE) + (threadIdx.x >> 6));
	const uint32_t nBatches = F.blockList ? F.nListed : (F.nSlots + 63u) >> 6;
	const uint32_t recBase = RECORDS ? wave * FRAME_RECORDS : 0u; // this wave's records
	const uint32_t recStride = RECORDS ? F.nRec : F.nSlots;       // entries of the per-path arrays
	// The counter block of the NEXT frame on this stream is cleared here, by one wave (nothing else in this launch touches
	// it, and the next frame starts when this launch has ended): no clear has to be queued in front of a frame.
	if (wave == 0u && F.countersNext) {
		for (uint32_t i = (uint32_t)lane; i < CTR_COUNT; i += 64u) F.countersNext[i] = 0u;
	}
	// (The number of the next batch is fetched only when the wave is ready for it: fetched ahead, every wave holds back one
	// batch while it works on the hits of another, and the frame ends a whole batch's chain of rounds later.)
	uint32_t pbatch = wave;       // the batch of slots to do next; >= nBatches: to be fetched, or none left
	bool pfirst = true;           // pbatch is the wave's own first batch
	uint32_t pcur = wave % CTR_F_NCUR, pcurTried = 0;
	uint32_t nShadowTraced = 0;   // per wave
	uint32_t nHit0 = 0, nHitAll = 0, nHitLast = 0; // per wave
	uint32_t nReflSkipped = 0;    // per lane (MIRT_TRACE_FAST only)
#if MIRT_STATS || MIRT_TIMELINE
	// timeline of the launch in ticks of the constant 100 MHz clock: S.dbg[16..] = first start (stored as ~t for atomicMax),
	// last end, first and last time a wave found the slots exhausted, summed wave lifetimes, summed time after exhaustion, waves
	const unsigned long long tStart = wall_clock64();
	unsigned long long tExhausted = 0;
	uint32_t nRounds = 0, nLaneRounds = 0;
	unsigned long long tRoundMax[3] = {0, 0, 0}, tRoundSum[3] = {0, 0, 0}; // camera / light / reflection rounds
#endif
	Walker<MODE, OVF> w;
	// The wave works in rounds; every round is: set up to 64 rays, walk them (ONE copy of the walk in the kernel), use the
	// answers.  ROUND_PRIMARY: 64 camera rays.  A batch of hits takes ROUND_LIGHT once per light, then ROUND_REFLECT.
	// (Measured and dropped: at the end of a wave's work, when the hits that are left and all their rays fit into the 64 lanes
	// at once, one merged round per depth -- reflection rays in the first lanes, shadow queries in the others, answers passed
	// through LDS.  Two rounds fewer per wave, but the larger kernel spills more registers in every round: a frame alone 6 %
	// slower, eight frames in flight 12 % slower.)
	enum { ROUND_PRIMARY, ROUND_LIGHT, ROUND_REFLECT };
	int round = ROUND_PRIMARY;
	uint32_t li = 0;              // the light of ROUND_LIGHT
	uint32_t entry = FRAME_EMPTY; // this lane's hit of the current batch (pixel slot, or record of the wave, | depth << 27), or none
	uint32_t tag = 0;             // ROUND_PRIMARY: this lane's slot
	// (the hit's light sum so far, main.cpp:204-253, and its material index wait in F.litL while the rays are walked: four
	// registers less to carry through the walk)
	bool inBatch = false;         // a batch of hits is being worked on
	for (;;) {
		// ---- which round, and its rays --------------------------------------------------------------------------------
		bool walk = false;        // this lane has a ray in this round
		w.cur = BVH_DONE;
		w.best.ref = REF_NONE;
		w.best.dist = __builtin_inff();
		w.needStrict = false;
		w.tcull = 0.0f;
		if (!inBatch) {
			if (nbuf < F.pullBelow && !pfirst) {
				pbatch = nBatches;
				while (pcurTried < CTR_F_NCUR) {
					uint32_t t = 0;
					if (lane == 0) t = atomicAdd(F.counters + CTR_F_PCUR(pcur), 1u);
					const uint32_t b = nWaves + pcur + CTR_F_NCUR * __builtin_amdgcn_readfirstlane(t);
					if (b < nBatches) { pbatch = b; break; }
					pcur = (pcur + 1u) % CTR_F_NCUR;
					++pcurTried;
				}
			}
			if (nbuf < F.pullBelow && pbatch < nBatches) {
				// up to 64 camera rays (main.cpp:152-153, depth 0)
				round = ROUND_PRIMARY;
				pfirst = false;
				// (batches are not handed out in image order: see where the host sets batchMul)
				uint32_t batch = F.batchMul ? (uint32_t)(((unsigned long long)pbatch * F.batchMul) % nBatches) : pbatch;
				if (F.blockList) batch = F.blockList[batch]; // (only the blocks that can see the scene are on the list)
				tag = batch * 64u + (uint32_t)lane;
				const uint32_t got = F.nSlots - batch * 64u < 64u ? F.nSlots - batch * 64u : 64u;
				if ((uint32_t)lane < got) {
					uint32_t px, py, inTile;
					if (slot_to_pixel(F, tag, px, py, inTile)) { // false: padding pixel of an edge tile
						f3 o, d;
						primary_ray(S, px, py, F.width, F.height, o, d);
						w.start(S, o, d, -1.0f);
						walk = true;
					}
				}
			}
			else {
				// up to 64 hits: their lights, then their reflection rays (main.cpp:187-190)
				if (nbuf >= 64u) {
					nbuf -= 64u;
					entry = myList[nbuf + (uint32_t)lane];
				}
				else {
					// no slots left: the rest of the list, however little it is
					if (nbuf == 0u) break; // this wave is done
					entry = (uint32_t)lane < nbuf ? myList[lane] : FRAME_EMPTY;
					nbuf = 0;
				}
				if (entry != FRAME_EMPTY) {
					const uint32_t matIndex = info_of(S, __float_as_uint(F.hitN[recBase + (entry & FRAME_SLOT_MASK)].w)).y;
					const float4* mat = S.materials + (size_t)matIndex * 4;
					const f3 sum = F3(mat[3]) + F3(mat[2]);                  // main.cpp:204: ambient + emission
					F.litL[(size_t)(entry >> FRAME_SLOT_BITS) * recStride + recBase + (entry & FRAME_SLOT_MASK)] = make_float4(sum.x, sum.y, sum.z, __uint_as_float(matIndex));
				}
				inBatch = true;
				li = 0;
				round = S.nLights ? ROUND_LIGHT : ROUND_REFLECT;
			}
		}
		const bool has = inBatch && entry != FRAME_EMPTY;
		const uint32_t rec = recBase + (entry & FRAME_SLOT_MASK), depth = has ? entry >> FRAME_SLOT_BITS : 0u;
		if (inBatch && round == ROUND_LIGHT) {
			if (has && shadows) {                                            // main.cpp:208-224
				const float4 ho = F.hitO[rec], hd = F.hitD[rec];
				const f3 po = F3(ho), pd = F3(hd);
				const f3 P = po + normalize3(pd) * hd.w;                     // Camera.cpp:26-28
				const float4 ll = S.lights[li * 2];
				const f3 L = ll.w > 0.0f ? F3(ll) - P : F3(ll);              // main.cpp:213-219
				const f3 dir = normalize3(L);                                // :222
				const float dl = length3(L);
				walk = true;
				// a query whose light term is multiplied by exact zeros is answered without a walk (see item_ray)
				if (S.shadowCull && F.debug == D_NONE) {
					const f3 N = F3(F.hitN[rec]);
					const bool kd0 = !(F.features & F_DIFFUSE) || dot3(normalize3(N), normalize3(L)) <= 0.0f;
					bool ks0 = !(F.features & F_SPECULAR);
					if (!ks0) {
						const f3 H = normalize3(dir + normalize3(po - P));
						ks0 = dot3(H, N) <= 0.0f;
					}
					const float atten = ll.w > 0.0f ? 1.0f / (S.atten[0] + S.atten[1] * dl + S.atten[2] * (dl * dl)) : 1.0f;
					if (kd0 && ks0 && fabsf(atten) <= 1e15f) walk = false;
				}
				if (walk) w.start(S, P, dir, F.debug == D_SHADOW_MAP ? -2.0f : dl); // :223-224 (SHADOW_MAP wants the nearest occluder)
			}
		}
		else if (inBatch) { // ROUND_REFLECT: a path goes on while depth + 1 <= maxDepth (main.cpp:177,189-190)
			bool spawn = has && depth < last;
			if (MODE == MODE_FAST && spawn) {
				// The reference traces the reflection ray even when its weight mat.specular is zero in all channels
				// (main.cpp:190), and so does MIRT_TRACE_EXACT: 0 x (what comes back) is not 0 when an inf or a NaN comes back.
				// MIRT_TRACE_FAST leaves that case aside and ends the path here.
				const float4 sp = S.materials[(size_t)__float_as_uint(F.litL[(size_t)depth * recStride + rec].w) * 4 + 1];
				if (sp.x == 0.0f && sp.y == 0.0f && sp.z == 0.0f) { spawn = false; ++nReflSkipped; }
			}
			if (spawn) {
				const float4 ho = F.hitO[rec], hd = F.hitD[rec], hn = F.hitN[rec];
				const f3 pd = F3(hd), N = F3(hn);
				const f3 P = F3(ho) + normalize3(pd) * hd.w;
				w.start(S, P, normalize3(pd - N * (2.0f * dot3(pd, N))), -1.0f);  // main.cpp:188
				walk = true;
			}
		}
		// ---- the walk ------------------------------------------------------------------------------------------------
		const unsigned long long walked = __builtin_amdgcn_ballot_w64(walk);
		if (inBatch && round == ROUND_LIGHT) nShadowTraced += (uint32_t)__popcll(walked);
#if MIRT_STATS || MIRT_TIMELINE
		if (pbatch >= nBatches && !tExhausted) tExhausted = wall_clock64();
		if (walked) { ++nRounds; nLaneRounds += (uint32_t)__popcll(walked); }
		const unsigned long long tRound = wall_clock64();
#endif
		if (walked) {
			// (the list holds fewer than 64 entries while rays are walked: its upper half is free)
			if (SHARE) walk_wave_share(S, w, stk, s_stack, myList + 64, (unsigned long long*)s_stack);
			else walk_wave<false, false>(S, w, stk);
			// deferred check of the final answers; the rare ray that fails it is walked again, strictly
			if (MODE == MODE_EXACT) {
				const bool redo = walk && !w.verify(S);
				if (__builtin_amdgcn_ballot_w64(redo)) {
#if MIRT_TIMELINE
					const unsigned long long tRedo = wall_clock64(); // S.dbg[0..3]: strict second walks, their summed and longest time, their rays
#endif
					if (redo) w.start(S, w.r.o, w.r.d, w.limit);
					walk_wave<false, true>(S, w, stk);
#if MIRT_TIMELINE
					if (lane == 0) {
						const unsigned long long dt = wall_clock64() - tRedo;
						atomicAdd(S.dbg + 0, 1ull); atomicAdd(S.dbg + 1, dt); atomicMax(S.dbg + 2, dt);
						atomicAdd(S.dbg + 3, (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(redo)));
					}
#endif
				}
			}
		}
#if MIRT_STATS || MIRT_TIMELINE
		if (walked) {
			const unsigned long long dt = wall_clock64() - tRound;
			const int kind = !inBatch ? 0 : (round == ROUND_LIGHT ? 1 : 2);
			if (dt > tRoundMax[kind]) tRoundMax[kind] = dt;
			tRoundSum[kind] += dt;
		}
#endif
		const bool hit = walk && w.best.ref != REF_NONE;
		// ---- what the answers mean -------------------------------------------------------------------------------------
		if (!inBatch) {
			const bool red = hit && F.debug == D_PRIMARY_MAP;   // PRIMARY_INTERSECTION_MAP paints hits pure red (main.cpp:183-185)
			// The background is black (main.cpp:124,180).  Unless the buffer was cleared beforehand, every slot of the batch that
			// did not hit writes its own black pixel -- the padding pixels of an edge tile too where they have a place (packed
			// tiles), so that no byte of the output depends on what the buffer held before.
			// A batch is an 8x8 block of pixels, eight pieces of 24 bytes.  When none of its 64 camera rays hit anything -- four
			// batches in five of the Stanford dragon's frame -- 48 lanes write one aligned word each instead of 64 lanes three
			// bytes each (F.wideClear: rows of the output start on word boundaries).
			const bool allMiss = F.wideClear && !F.preCleared && walked == ~0ull && __builtin_amdgcn_ballot_w64(hit) == 0ull;
			if (allMiss) {
				if (lane < 48) {
					uint32_t px, py, inTile;
					const uint32_t first = tag - (uint32_t)lane + 8u * ((uint32_t)lane / 6u); // first slot of the lane's row of the block
					slot_to_pixel(F, first, px, py, inTile);
					*(uint32_t*)(out + out_offset(F, first, px, py, inTile) + 4u * ((uint32_t)lane % 6u)) = 0u;
				}
			}
			else if (red || (!hit && !F.preCleared && tag < F.nSlots)) {
				uint32_t px, py, inTile;
				const bool live = slot_to_pixel(F, tag, px, py, inTile);
				if (live || F.packedTiles) {
					const size_t k = out_offset(F, tag, px, py, inTile);
					out[k] = 0;
					out[k + 1] = 0;
					out[k + 2] = hit ? 255 : 0;
				}
			}
			const bool newHit = hit && !red;
			const unsigned long long hm = __builtin_amdgcn_ballot_w64(newHit);
			uint32_t r = tag; // where the new path's data goes: its pixel slot, or (RECORDS) a record taken from the wave's stack
			if (RECORDS && newHit) { // (there is always one: fewer than FRAME_RECORDS paths are under way)
				r = myFree[nFree - 1u - (uint32_t)__popcll(hm & ltMask)];
				slotOf[r] = tag;
			}
			if (newHit) {
				const f3 n = hit_normal(S, w.r, w.best.ref);
				F.hitO[recBase + r] = make_float4(w.r.o.x, w.r.o.y, w.r.o.z, 0.0f);
				F.hitD[recBase + r] = make_float4(w.r.d.x, w.r.d.y, w.r.d.z, w.best.dist);
				F.hitN[recBase + r] = make_float4(n.x, n.y, n.z, __uint_as_float(w.best.ref));
			}
			const uint32_t nh = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(hit));
			nHit0 += nh;
			nHitAll += nh;
			if (last == 0u) nHitLast += nh;
			if (hm) {
				if (newHit) myList[nbuf + (uint32_t)__popcll(hm & ltMask)] = r; // depth 0
				nbuf += (uint32_t)__popcll(hm);
				if (RECORDS) nFree -= (uint32_t)__popcll(hm);
			}
			continue;
		}
		if (round == ROUND_LIGHT) {
			if (has) {
				const float4 ho = F.hitO[rec], hd = F.hitD[rec], hn = F.hitN[rec];
				const f3 po = F3(ho);
				const f3 P = po + normalize3(F3(hd)) * hd.w;
				float4* lit = F.litL + (size_t)depth * recStride + rec;
				const float4 l4 = *lit;
				f3 sum = F3(l4);
				shade_light(S, F.features, F.debug, po, P, F3(hn), S.materials + (size_t)__float_as_uint(l4.w) * 4, li, shadows,
				            hit ? w.best.ref : REF_NONE, hit ? w.best.dist : __builtin_inff(), sum);
				*lit = make_float4(sum.x, sum.y, sum.z, l4.w);
			}
			if (++li >= S.nLights) round = ROUND_REFLECT;
			continue;
		}
		// ROUND_REFLECT: the hit's light sum is complete (in F.litL)
		if (hit) {
			const f3 n = hit_normal(S, w.r, w.best.ref);
			F.hitO[rec] = make_float4(w.r.o.x, w.r.o.y, w.r.o.z, 0.0f);
			F.hitD[rec] = make_float4(w.r.d.x, w.r.d.y, w.r.d.z, w.best.dist);
			F.hitN[rec] = make_float4(n.x, n.y, n.z, __uint_as_float(w.best.ref));
		}
		if (has && !hit) {
			// the path ends here: fold its light sums back to front, C = L_k + S_k * C (main.cpp:190), starting from Color()
			// beyond the last hit (main.cpp:199) or the black background (main.cpp:124); quantise (Color.cpp:13-23)
			f3 C = F3(0, 0, 0);
			for (int k = (int)depth; k >= 0; --k) {
				const float4 L = F.litL[(size_t)k * recStride + rec];
				const float4 Sp = S.materials[(size_t)__float_as_uint(L.w) * 4 + 1];
				if (reflections || k > 0) C = F3(L) + F3(Sp) * C;
				else C = F3(L); // main.cpp:193
			}
			uint32_t px, py, inTile;
			const uint32_t slot = RECORDS ? slotOf[entry & FRAME_SLOT_MASK] : (entry & FRAME_SLOT_MASK);
			if (slot_to_pixel(F, slot, px, py, inTile)) {
				const size_t o = out_offset(F, slot, px, py, inTile);
				out[o] = quantise(C.z);     // B  main.cpp:154
				out[o + 1] = quantise(C.y); // G
				out[o + 2] = quantise(C.x); // R
			}
		}
		const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
		if (hm) {
			nHitAll += (uint32_t)__popcll(hm);
			nHitLast += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(hit && depth + 1u == last));
			if (hit) myList[nbuf + (uint32_t)__popcll(hm & ltMask)] = (entry & FRAME_SLOT_MASK) | ((depth + 1u) << FRAME_SLOT_BITS);
			nbuf += (uint32_t)__popcll(hm);
		}
		// the records of the paths that ended go back on the stack
		const unsigned long long em = RECORDS ? __builtin_amdgcn_ballot_w64(has && !hit) : 0ull;
		if (RECORDS && em) {
			if (has && !hit) myFree[nFree + (uint32_t)__popcll(em & ltMask)] = (uint8_t)(entry & FRAME_SLOT_MASK);
			nFree += (uint32_t)__popcll(em);
		}
		inBatch = false;
	}
#if MIRT_STATS || MIRT_TIMELINE
	if (lane == 0) {
		const unsigned long long tEnd = wall_clock64();
		if (!tExhausted) tExhausted = tEnd;
		atomicMax(S.dbg + 16, ~tStart);
		atomicMax(S.dbg + 17, tEnd);
		atomicMax(S.dbg + 18, ~tExhausted);
		atomicMax(S.dbg + 19, tExhausted);
		atomicAdd(S.dbg + 20, tEnd - tStart);
		atomicAdd(S.dbg + 21, tEnd - tExhausted);
		atomicAdd(S.dbg + 22, 1ull);
#if MIRT_TIMELINE
		if (wave < 8192u) { // per wave: start, end, when it found the slots exhausted, rounds | longest round << 32
			const unsigned long long longest = tRoundMax[0] > tRoundMax[1] ? (tRoundMax[0] > tRoundMax[2] ? tRoundMax[0] : tRoundMax[2]) : (tRoundMax[1] > tRoundMax[2] ? tRoundMax[1] : tRoundMax[2]);
			S.dbg[40 + 4 * wave] = tStart; S.dbg[41 + 4 * wave] = tEnd; S.dbg[42 + 4 * wave] = tExhausted; S.dbg[43 + 4 * wave] = (unsigned long long)nRounds | (longest << 32);
		}
#endif
		atomicAdd(S.dbg + 23, (unsigned long long)nRounds);
		atomicAdd(S.dbg + 24, (unsigned long long)nLaneRounds);
		for (int k = 0; k < 3; ++k) {
			atomicMax(S.dbg + 25 + k, tRoundMax[k]);
			atomicAdd(S.dbg + 28 + k, tRoundSum[k]);
		}
	}
#endif
	// statistics: a few atomics per wave
	if (MODE == MODE_FAST) {
		for (int off = 32; off > 0; off >>= 1) nReflSkipped += __shfl_down(nReflSkipped, off);
		if (lane == 0 && nReflSkipped) atomicAdd(F.counters + CTR_F_REFL_SKIPPED, nReflSkipped);
	}
	if (lane == 0) {
		if (nShadowTraced) atomicAdd(F.counters + CTR_F_SHADOW_TRACED, nShadowTraced);
		if (nHit0) atomicAdd(F.counters + CTR_F_HITS_PRIMARY, nHit0);
		if (nHitAll) atomicAdd(F.counters + CTR_F_HITS_TOTAL, nHitAll);
		if (nHitLast) atomicAdd(F.counters + CTR_F_HITS_LAST, nHitLast);
	}
}

// ---- which parts of a finished image are not black ---------------------------------------------------------------------------
// For the copy back to the host: parts[i] = 1 if bytes [i << shift, (i + 1) << shift) of the image hold anything but zeros.
// A block reads one piece of one part (16 bytes per lane and step); 25 MB take about 10 us.  (Having k_frame note where it
// writes instead costs that kernel 16 % of its speed: two more values live across its rounds.)
__global__ __launch_bounds__(BLOCK) void k_note_parts(const uint8_t* __restrict__ img, size_t bytes, uint32_t shift, uint32_t blocksPerPart, uint32_t* __restrict__ parts) {
	const uint32_t part = blockIdx.x / blocksPerPart, piece = blockIdx.x % blocksPerPart;
	const size_t partBytes = (size_t)1 << shift, pieceBytes = partBytes / blocksPerPart; // (both multiples of 16)
	const size_t from = (size_t)part * partBytes + (size_t)piece * pieceBytes;
	size_t to = from + pieceBytes;
	if (from >= bytes) return;
	if (to > bytes) to = bytes;
	uint32_t any = 0;
	const size_t n16 = (to - from) >> 4;
	const uint4* q = (const uint4*)(img + from);
	for (size_t i = threadIdx.x; i < n16; i += BLOCK) {
		const uint4 v = q[i];
		any |= v.x | v.y | v.z | v.w;
	}
	for (size_t i = from + (n16 << 4) + threadIdx.x; i < to; i += BLOCK) any |= img[i];
	if (__builtin_amdgcn_ballot_w64(any != 0u) && (threadIdx.x & 63u) == 0u) parts[part] = 1u;
}

// ---- PNG rows of a finished image --------------------------------------------------------------------------------------------
// What Renderer::createImage hands to the PNG writer (Renderer.cpp:13-14), made where the image lies: row i of `rows` is
// the filter-type byte 2 ("Up", PNG specification 6.3) followed by R,G,B of every pixel, each byte minus the same byte of
// the row above (row 0: minus nothing), modulo 256.  stripeAny[i / rowsPerStripe] = 1 when a stripe of rows holds a
// filtered byte that is not 0 (the words are cleared before the launch); a stripe without one deflates to a stream that
// depends on its shape alone.  One thread per pixel.
__global__ __launch_bounds__(BLOCK) void k_png_rows(const uint8_t* __restrict__ bgr, uint32_t width, uint32_t height, uint32_t rowsPerStripe,
                                                   uint8_t* __restrict__ rows, uint32_t* __restrict__ stripeAny) {
	const uint64_t n = (uint64_t)width * height;
	const size_t line = 1 + (size_t)width * 3;
	for (uint64_t k = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; k < n; k += (uint64_t)gridDim.x * BLOCK) {
		const uint32_t i = (uint32_t)(k / width), j = (uint32_t)(k - (uint64_t)i * width);
		const uint8_t* src = bgr + 3 * k;
		uint8_t b = src[0], g = src[1], r = src[2]; // main.cpp:154-156 stores B,G,R
		if (i > 0) {
			const uint8_t* up = src - (size_t)width * 3;
			b = (uint8_t)(b - up[0]);
			g = (uint8_t)(g - up[1]);
			r = (uint8_t)(r - up[2]);
		}
		uint8_t* dst = rows + (size_t)i * line;
		if (j == 0) dst[0] = 2;
		dst += 1 + (size_t)j * 3;
		dst[0] = r;
		dst[1] = g;
		dst[2] = b;
		if (r | g | b) stripeAny[i / rowsPerStripe] = 1u;
	}
}

// ---- narrow seam: closest hit for a list of rays ----------------------------------------------------------------------
template <int MODE, bool OVF>
__global__ __launch_bounds__(BLOCK) void k_trace_list(DScene S, uint32_t* ovf, uint32_t ovfStride, uint32_t faithful, uint64_t n, const float* __restrict__ o3, const float* __restrict__ d3,
                                                      int32_t* __restrict__ primId, float* __restrict__ dist, float* __restrict__ normal) {
	__shared__ uint32_t s_stack[MODE == MODE_REFWALK ? 1 : STACK_LDS * BLOCK];
	LdsStack<OVF> stk;
	stk.lds = s_stack + (MODE == MODE_REFWALK ? 0 : threadIdx.x);
	stk.ovf = OVF ? ovf + ((size_t)blockIdx.x * BLOCK + threadIdx.x) : nullptr;
	stk.ovfStride = ovfStride;
	for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
		const f3 o = F3(o3 + 3 * i), d = F3(d3 + 3 * i);
		const RayCtx r = make_ray(o, d);
		Hit h;
		if (faithful && !all_finite(o, d)) traverse_faithful(S, o, d, h); // (faithful: the reference's tree is not deeper than that walk's stack)
		else trace_whole<MODE, OVF>(S, o, d, -1.0f, h, stk);
		if (h.ref != REF_NONE) {
			primId[i] = (int32_t)info_of(S, h.ref).x;
			dist[i] = h.dist;
			if (normal) {
				const f3 nn = hit_normal(S, r, h.ref);
				normal[3 * i] = nn.x; normal[3 * i + 1] = nn.y; normal[3 * i + 2] = nn.z;
			}
		}
		else {
			primId[i] = -1;
			dist[i] = __builtin_inff();
			if (normal) { normal[3 * i] = 0; normal[3 * i + 1] = 0; normal[3 * i + 2] = 0; }
		}
	}
}

template <int MODE, bool OVF>
__global__ __launch_bounds__(BLOCK) void k_primary_ids(DScene S, uint32_t* ovf, uint32_t ovfStride, uint32_t width, uint32_t height, int32_t* __restrict__ primId, float* __restrict__ dist) {
	__shared__ uint32_t s_stack[MODE == MODE_REFWALK ? 1 : STACK_LDS * BLOCK];
	LdsStack<OVF> stk;
	stk.lds = s_stack + (MODE == MODE_REFWALK ? 0 : threadIdx.x);
	stk.ovf = OVF ? ovf + ((size_t)blockIdx.x * BLOCK + threadIdx.x) : nullptr;
	stk.ovfStride = ovfStride;
	const uint64_t n = (uint64_t)width * height;
	for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
		f3 o, d;
		primary_ray(S, (uint32_t)(i % width), (uint32_t)(i / width), width, height, o, d);
		Hit h;
		trace_whole<MODE, OVF>(S, o, d, -1.0f, h, stk);
		primId[i] = h.ref != REF_NONE ? (int32_t)info_of(S, h.ref).x : -1;
		dist[i] = h.ref != REF_NONE ? h.dist : __builtin_inff();
	}
}

// ---- parity hook: the device's powf for a list of arguments ----------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_test_powf(uint64_t n, const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out) {
	for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) out[i] = ref_powf(x[i], y[i]);
}

// ---- measurement hook: the shader clock the GPU holds while this wave is resident ------------------------------------------------
// One wave notes the shader-cycle counter and the constant 100 MHz counter, sleeps until `ticks` of the latter have passed
// (the host bounds them: the loop ends whatever else the GPU does) and notes both again: out = { shader cycles, 100 MHz ticks }.
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long* __restrict__ out, unsigned long long ticks) {
	const unsigned long long r0 = wall_clock64(), c0 = (unsigned long long)clock64();
	unsigned long long r1 = r0;
	while (r1 - r0 < ticks) {
		__builtin_amdgcn_s_sleep(64);
		r1 = wall_clock64();
	}
	const unsigned long long c1 = (unsigned long long)clock64();
	if (threadIdx.x == 0) {
		out[0] = c1 - c0;
		out[1] = r1 - r0;
	}
}

// ---- multi-GPU: put gathered per-rank tile buffers back into one image --------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_assemble(uint32_t width, uint32_t height, uint32_t tileW, uint32_t tileH, uint32_t tilesX,
                                                   uint32_t nRanks, const uint8_t* __restrict__ tiles, size_t perRank, uint8_t* __restrict__ bgr) {
	const uint64_t n = (uint64_t)width * height;
	for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
		const uint32_t px = (uint32_t)(i % width), py = (uint32_t)(i / width);
		const uint32_t k = (py / tileH) * tilesX + (px / tileW);
		const uint32_t rank = k % nRanks, lt = k / nRanks;
		const size_t src = (size_t)rank * perRank + ((size_t)lt * tileW * tileH + (size_t)(py % tileH) * tileW + (px % tileW)) * 3;
		bgr[3 * i] = tiles[src];
		bgr[3 * i + 1] = tiles[src + 1];
		bgr[3 * i + 2] = tiles[src + 2];
	}
}

// The same in 16-byte pieces, for the usual case: a tile row (tileW * 3 bytes) and an image row (width * 3 bytes) are
// multiples of 16 bytes and both buffers are 16-byte aligned, so every piece is aligned on both sides.  One thread moves
// one piece; the pieces of a tile are consecutive in the gathered buffer (a tile is tileW * tileH * 3 contiguous bytes).
// The last bytes of an edge tile's rows (image width not a multiple of the tile width) go one by one.
__global__ __launch_bounds__(BLOCK) void k_assemble16(uint32_t width, uint32_t height, uint32_t tileW, uint32_t tileH, uint32_t tilesX, uint32_t nTiles,
                                                     uint32_t nRanks, const uint8_t* __restrict__ tiles, size_t perRank, uint8_t* __restrict__ bgr) {
	const uint32_t chunks = tileW * 3u / 16u;          // pieces per tile row
	const uint32_t perTile = tileH * chunks;
	const uint64_t n = (uint64_t)nTiles * perTile;
	for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
		const uint32_t k = (uint32_t)(i / perTile), r = (uint32_t)(i - (uint64_t)k * perTile);
		const uint32_t y = r / chunks, c = r - y * chunks;
		const uint32_t px0 = (k % tilesX) * tileW, py = (k / tilesX) * tileH + y;
		if (py >= height) continue;
		const uint32_t rowBytes = (px0 + tileW <= width ? tileW : width - px0) * 3u; // bytes of this tile's row that lie in the image
		const uint32_t rank = k % nRanks, lt = k / nRanks;
		const uint8_t* src = tiles + (size_t)rank * perRank + ((size_t)lt * tileW * tileH + (size_t)y * tileW) * 3 + (size_t)c * 16;
		uint8_t* dst = bgr + ((size_t)py * width + px0) * 3 + (size_t)c * 16;
		if (c * 16u + 16u <= rowBytes) *(uint4*)dst = *(const uint4*)src;
		else for (uint32_t b = c * 16u; b < rowBytes; ++b) dst[b - c * 16u] = src[b - c * 16u];
	}
}
