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

#ifndef MIRT_BOX_FMA
#define MIRT_BOX_FMA 0
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

struct Hit {
	float dist;   // world distance |P - o| (Intersection::distAlongRay); +inf = miss
	uint32_t ref; // typed primitive reference (PRIM_SPHERE | index) or REF_NONE
	uint32_t pos; // position of the winning entry in the leaf stream (tie-break rank)
};

struct RayCtx {
	f3 o, d;
	f3 e;      // direction with AABB.cpp:53-58's substitution: |d| < 1e-5 -> +1e-5
	f3 ie;     // 1/d for the hierarchy's conservative box filter
	float invLen; // 1/|d|, used only for conservative culling
#if MIRT_BOX_FMA
	f3 oie;    // o * ie
	float ferr; // bound on the rounding error of the fused slab parameters
#endif
};

__device__ __forceinline__ RayCtx make_ray(f3 o, f3 d) {
	RayCtx r;
	r.o = o;
	r.d = d;
	r.e.x = (fabsf(d.x - 0.0f) < 0.00001f) ? 0.00001f : d.x;
	r.e.y = (fabsf(d.y - 0.0f) < 0.00001f) ? 0.00001f : d.y;
	r.e.z = (fabsf(d.z - 0.0f) < 0.00001f) ? 0.00001f : d.z;
	// reciprocal of the TRUE direction for the hierarchy's conservative filter (tiny components clamped, sign kept)
	const float sx = fabsf(d.x) < 1e-20f ? copysignf(1e-20f, d.x) : d.x;
	const float sy = fabsf(d.y) < 1e-20f ? copysignf(1e-20f, d.y) : d.y;
	const float sz = fabsf(d.z) < 1e-20f ? copysignf(1e-20f, d.z) : d.z;
	r.ie = F3(1.0f / sx, 1.0f / sy, 1.0f / sz);
	r.invLen = inv_sqrt(dot3(d, d));
#if MIRT_BOX_FMA
	r.oie = r.o * r.ie;
	// |fl(o*ie) - o*ie| <= 2^-24 |o*ie| per axis, the FMA adds half an ulp of the (smaller) result: 2^-22 max|o*ie| covers both boxes' planes
	r.ferr = 2.4e-7f * fmaxf(fmaxf(fabsf(r.oie.x), fabsf(r.oie.y)), fabsf(r.oie.z)) + 1e-30f;
#endif
	return r;
}

// AABB::intersect hit/miss (AABB.cpp:53-72).  tnear receives tmin (ray-parameter units).
template <int MODE>
__device__ __forceinline__ bool box_hit(const RayCtx& r, f3 lo, f3 hi, float& tnear) {
	float t1, t2, t3, t4, t5, t6;
	if (MODE == MODE_EXACT) {
		t1 = (lo.x - r.o.x) / r.e.x;
		t2 = (hi.x - r.o.x) / r.e.x;
		t3 = (lo.y - r.o.y) / r.e.y;
		t4 = (hi.y - r.o.y) / r.e.y;
		t5 = (lo.z - r.o.z) / r.e.z;
		t6 = (hi.z - r.o.z) / r.e.z;
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
	return true;
}

// Triangle::intersect (Triangle.cpp:147-175) on the precomputed record.  Returns the hit distance or -1.
// tcull: ray-parameter bound beyond which a hit cannot matter (conservative, see traverse()).
template <int MODE>
__device__ __forceinline__ float tri_hit(const RayCtx& r, const float4* __restrict__ T, float tcull) {
	const float4 q0 = T[0];
	const f3 N = F3(q0);
	const float dn = dot3(r.d, N);
	if (fabsf(dn - 0.0f) < 0.0001f) return -1.0f;            // :152
	const float t = (q0.w - dot3(r.o, N)) / dn;               // :153-154
	if (t < 0.0001f) return -1.0f;                            // :155
	if (t > tcull) return -1.0f;                              // cannot beat the current best
	const float4 q1 = T[1], q2 = T[2], q3 = T[3];
	const f3 v1 = F3(q1), v2 = F3(q2), v3 = F3(q3);
	const float area = q1.w;                                  // :162, evaluated on the host
	const f3 e1 = v3 - v2, e2 = v1 - v3, e3 = v2 - v1;        // :148-150
	const f3 P = r.o + r.d * t;                               // :158
	const float s0 = dot3(cross3(e1, P - v3), N);             // :163
	const float s1 = dot3(cross3(e2, P - v1), N);             // :164
	const float s2 = dot3(cross3(e3, P - v2), N);             // :165
	bool in;
	// (s/2)/area >= 0 is decided by the signs alone unless the quotient can underflow or overflow
	const float lo = fminf(fminf(fabsf(s0), fabsf(s1)), fabsf(s2));
	if (MODE == MODE_FAST || (fabsf(area) < 1e30f && (lo > 1e-30f))) {
		in = (s0 * area >= 0.0f) && (s1 * area >= 0.0f) && (s2 * area >= 0.0f);
	}
	else {
		in = (s0 / 2.0f / area >= 0.0f) && (s1 / 2.0f / area >= 0.0f) && (s2 / 2.0f / area >= 0.0f);
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
// Children are visited nearer-first and a subtree is skipped when its entry parameter exceeds `tcull`, a bound kept
// slightly beyond the best distance so far; every candidate at a distance <= best is still visited, so the result
// equals the reference's exhaustive walk (Partition.cpp:42-52).  Stack in private memory: this mode is the fallback
// and the cross-check, not the fast path.
template <bool ANY>
__device__ __noinline__ bool traverse_ref(const DScene& S, const RayCtx& r, float limit, Hit& best) {
	best.dist = __builtin_inff();
	best.ref = REF_NONE;
	best.pos = 0xffffffffu;
	float tcull = ANY ? (limit * r.invLen * 1.0001f + 1e-5f) : __builtin_inff();
	float tn;
	if (S.rootRef == REF_NONE || !box_hit<MODE_EXACT>(r, F3(S.rootMin), F3(S.rootMax), tn)) return false; // Partition.cpp:36 on the root
	uint32_t stkRef[STACK_MAX];
	float stkT[STACK_MAX];
	int sp = 0;
	uint32_t cur = S.rootRef;
	for (;;) {
		if (!(cur & REF_LEAF)) {
			const float4* nd = S.nodes + (size_t)cur * 4;
			const float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
			const uint32_t lref = __float_as_uint(q0.w), rref = __float_as_uint(q1.w);
			float tl, tr;
			const bool hl = (lref != REF_NONE) && box_hit<MODE_EXACT>(r, F3(q0), F3(q1), tl) && !(tl > tcull);
			const bool hr = (rref != REF_NONE) && box_hit<MODE_EXACT>(r, F3(q2), F3(q3), tr) && !(tr > tcull);
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
					dist = tri_hit<MODE_EXACT>(r, S.tris + (size_t)pr * 4, tcull);
				}
				if (dist >= 0.0f && !isinf(dist)) {
					if (ANY) {
						if (dist < limit) { best.dist = dist; best.ref = pr; best.pos = pos; return true; }
					}
					else if (dist < best.dist || (dist == best.dist && pos < best.pos)) { // Partition.h:33, Partition.cpp:47
						best.dist = dist;
						best.ref = pr;
						best.pos = pos;
						tcull = dist * r.invLen * 1.0001f + 1e-5f;
					}
				}
			}
		}
		// pop the next subtree that can still matter
		for (;;) {
			if (sp == 0) return best.ref != REF_NONE;
			--sp;
			cur = stkRef[sp];
			if (!(stkT[sp] > tcull)) break;
		}
	}
}

// ---- primitive hierarchy + membership check (MODE_EXACT / MODE_FAST) -----------------------------------------------------
// Conservative slab test for the hierarchy's own (padded) boxes: reciprocal multiply plus slack, never rejects a box the
// exact arithmetic would accept.  tnear = entry parameter.
__device__ __forceinline__ bool box_filter(const RayCtx& r, const float4& lo, const float4& hi, float tcull, float& tnear) {
#if MIRT_BOX_FMA
	// t = lo*ie - o*ie in one FMA; its absolute error is bounded by r.ferr (set in make_ray), which widens the accept test
	const float t1 = __builtin_fmaf(lo.x, r.ie.x, -r.oie.x), t2 = __builtin_fmaf(hi.x, r.ie.x, -r.oie.x);
	const float t3 = __builtin_fmaf(lo.y, r.ie.y, -r.oie.y), t4 = __builtin_fmaf(hi.y, r.ie.y, -r.oie.y);
	const float t5 = __builtin_fmaf(lo.z, r.ie.z, -r.oie.z), t6 = __builtin_fmaf(hi.z, r.ie.z, -r.oie.z);
#else
	const float t1 = (lo.x - r.o.x) * r.ie.x, t2 = (hi.x - r.o.x) * r.ie.x;
	const float t3 = (lo.y - r.o.y) * r.ie.y, t4 = (hi.y - r.o.y) * r.ie.y;
	const float t5 = (lo.z - r.o.z) * r.ie.z, t6 = (hi.z - r.o.z) * r.ie.z;
#endif
	const float tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
	const float tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
	tnear = tmin;
#if MIRT_BOX_FMA
	const float slack = 4e-6f * fmaxf(fabsf(tmin), fabsf(tmax)) + r.ferr;
	return (tmin <= tmax + slack) && (tmax >= -r.ferr) && !(tmin - r.ferr > tcull);
#else
	const float slack = 4e-6f * fmaxf(fabsf(tmin), fabsf(tmax)) + 1e-30f;
	return (tmin <= tmax + slack) && (tmax >= 0.0f) && !(tmin > tcull);
#endif
}

// Is the primitive held by a reference leaf whose box the ray hits?  Returns the visiting rank of the first such
// (leaf, primitive) entry, or REF_NONE.  AABB::intersect is evaluated exactly (IEEE division, AABB.cpp:53-72).
__device__ __forceinline__ uint32_t membership(const DScene& S, const RayCtx& r, uint32_t ofs, uint32_t cnt) {
	const float4* e = S.memb + (size_t)ofs * 2;
	for (uint32_t k = 0; k < cnt; ++k) {
		const float4 lo = e[2 * k], hi = e[2 * k + 1];
		float tn;
		if (box_hit<MODE_EXACT>(r, F3(lo), F3(hi), tn)) return __float_as_uint(lo.w);
	}
	return REF_NONE;
}

#define STACK_LDS 32

// Traversal stack: STACK_LDS entries per lane in LDS ([entry][lane]: conflict-free, no address arithmetic beyond a
// shift).  Hierarchies deeper than that (OVF = true, chosen by the host from the built depth) keep the deeper entries in
// a global array; the two cases are separate instantiations so that the common one contains LDS instructions only
// (a merged pointer would turn every pop into a flat_load).
template <bool OVF>
struct LdsStack {
	uint32_t* lds;   // this thread's column
	uint32_t* ovf;   // [level][global thread], this thread's column
	uint32_t ovfStride;
	__device__ __forceinline__ void push(int sp, uint32_t v) {
		if (!OVF || sp < STACK_LDS) lds[sp * BLOCK] = v;
		else __builtin_nontemporal_store(v, ovf + (size_t)(sp - STACK_LDS) * ovfStride);
	}
	__device__ __forceinline__ uint32_t pop(int sp) {
		if (!OVF || sp < STACK_LDS) return lds[sp * BLOCK];
		return __builtin_nontemporal_load(ovf + (size_t)(sp - STACK_LDS) * ovfStride);
	}
};

template <int MODE, bool ANY, bool OVF>
__device__ __forceinline__ bool traverse_bvh(const DScene& S, const RayCtx& r, float limit, Hit& best, LdsStack<OVF>& stk) {
	best.dist = __builtin_inff();
	best.ref = REF_NONE;
	best.pos = 0xffffffffu;
	float tcull = ANY ? (limit * r.invLen * 1.0001f + 1e-5f) : __builtin_inff();
	uint32_t cur = S.bvhRoot;
	{
		float tn;
		const float4 lo = make_float4(S.bvhMin[0], S.bvhMin[1], S.bvhMin[2], 0.0f), hi = make_float4(S.bvhMax[0], S.bvhMax[1], S.bvhMax[2], 0.0f);
		if (cur == BVH_DONE || !box_filter(r, lo, hi, tcull, tn)) return false;
	}
	int sp = 0;
	for (;;) {
		// descend through interior nodes
		while (!(cur & BVH_LEAF)) {
			const float4* nd = S.bvh + (size_t)cur * 4;
			const float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
			const uint32_t lref = __float_as_uint(q0.w), rref = __float_as_uint(q1.w);
			float tl, tr;
			const bool hl = box_filter(r, q0, q1, tcull, tl);
			const bool hr = box_filter(r, q2, q3, tcull, tr);
			if (hl && hr) {
				const bool leftFirst = tl <= tr;
				stk.push(sp, leftFirst ? rref : lref);
				++sp;
				cur = leftFirst ? lref : rref;
			}
			else if (hl) cur = lref;
			else if (hr) cur = rref;
			else if (sp > 0) cur = stk.pop(--sp);
			else cur = BVH_DONE;
		}
		if (cur == BVH_DONE) break;
		// leaf: 1..8 consecutive records of one kind
		const uint32_t first = cur & BVH_FIRST_MASK;
		const uint32_t cnt = ((cur >> BVH_COUNT_SHIFT) & 7u) + 1u;
		const bool sph = (cur & BVH_SPHERES) != 0;
		for (uint32_t k = 0; k < cnt; ++k) {
			const uint32_t idx = first + k;
			float dist;
			if (sph) {
				f3 dummy;
				dist = sphere_hit<false>(r, S.spheres + (size_t)idx * SPHERE_F4, dummy);
			}
			else {
				dist = tri_hit<MODE>(r, S.tris + (size_t)idx * 4, tcull);
			}
			if (!(dist >= 0.0f) || isinf(dist)) continue;
			const uint32_t ref = sph ? (PRIM_SPHERE | idx) : idx;
			if (ANY) {
				if (dist < limit) {
					uint32_t pos = 0;
					if (MODE == MODE_EXACT) {
						const uint4 info = sph ? S.sphInfo[idx] : S.triInfo[idx];
						pos = membership(S, r, info.z, info.w);
					}
					if (pos != REF_NONE) { best.dist = dist; best.ref = ref; best.pos = pos; return true; }
				}
			}
			else if (dist < best.dist || (dist == best.dist && ref != best.ref)) {
				uint32_t pos = 0;
				if (MODE == MODE_EXACT) {
					const uint4 info = sph ? S.sphInfo[idx] : S.triInfo[idx];
					pos = membership(S, r, info.z, info.w);
				}
				// Partition.h:33 (first in a leaf wins), Partition.cpp:47 (left subtree wins) == lower visiting rank wins
				if (pos != REF_NONE && (dist < best.dist || pos < best.pos)) {
					best.dist = dist;
					best.ref = ref;
					best.pos = pos;
					tcull = dist * r.invLen * 1.0001f + 1e-5f;
				}
			}
		}
		if (sp == 0) break;
		cur = stk.pop(--sp);
	}
	return best.ref != REF_NONE;
}

template <int MODE, bool ANY, bool OVF>
__device__ __forceinline__ bool trace_ray(const DScene& S, const RayCtx& r, float limit, Hit& best, LdsStack<OVF>& stk) {
	if (MODE == MODE_REFWALK) return traverse_ref<ANY>(S, r, limit, best);
	return traverse_bvh<MODE, ANY, OVF>(S, r, limit, best, stk);
}

// Intersection::intersectNormal of the winning primitive (normalised by the Intersection ctor, SceneObjects.hpp:39)
__device__ __forceinline__ f3 hit_normal(const DScene& S, const RayCtx& r, uint32_t ref) {
	if (ref & PRIM_SPHERE) {
		f3 n;
		sphere_hit<true>(r, S.spheres + (size_t)(ref & ~PRIM_SPHERE) * SPHERE_F4, n);
		return normalize3(n);
	}
	return normalize3(F3(S.tris[(size_t)ref * 4]));
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
__device__ __forceinline__ bool slot_to_pixel(const DFrame& F, uint32_t slot, uint32_t& px, uint32_t& py, uint32_t& inTile) {
	const uint32_t tileArea = F.tileW * F.tileH;
	const uint32_t lt = slot / tileArea, i = slot % tileArea;
	const uint32_t sb = i >> 6, l = i & 63;
	const uint32_t sbx = F.tileW >> 3;
	const uint32_t tx = (sb % sbx) * 8 + (l & 7), ty = (sb / sbx) * 8 + (l >> 3);
	const uint32_t k = F.firstTile + lt * F.tileStride;
	px = (k % F.tilesX) * F.tileW + tx;
	py = (k / F.tilesX) * F.tileH + ty;
	inTile = ty * F.tileW + tx;
	return px < F.width && py < F.height;
}
__device__ __forceinline__ size_t out_offset(const DFrame& F, uint32_t slot, uint32_t px, uint32_t py, uint32_t inTile) {
	if (F.packedTiles) return ((size_t)(slot / (F.tileW * F.tileH)) * (F.tileW * F.tileH) + inTile) * 3;
	return ((size_t)py * F.width + px) * 3; // main.cpp:154-156
}

// wave-aggregated queue append: one atomic per wave
__device__ __forceinline__ uint32_t queue_push(uint32_t* counter, bool want) {
	const unsigned long long m = __ballot(want);
	if (m == 0ull) return 0;
	const int lane = threadIdx.x & 63;
	const int leader = __ffsll((long long)m) - 1;
	uint32_t base = 0;
	if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
	base = __shfl(base, leader);
	return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// Work distribution: a wave owns GRAB consecutive 64-item chunks at a time.  Its first batch is static (wave index),
// later batches come from one atomic cursor per launch -- one atomic per GRAB*64 items keeps the cursor word far
// below the ~88 atomics/us a single address sustains.
template <int GRAB>
struct Dealer {
	uint32_t* cursor;
	uint32_t cb, ce, stride0;
	__device__ __forceinline__ Dealer(uint32_t* c) : cursor(c) {
		const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (BLOCK / WAVE) + (threadIdx.x >> 6));
		stride0 = gridDim.x * (BLOCK / WAVE) * (WAVE * GRAB);
		cb = wave * (WAVE * GRAB);
		ce = cb + WAVE * GRAB;
	}
	// base of the next 64-item chunk (>= n means: no more work)
	__device__ __forceinline__ uint32_t next() {
		if (cb >= ce) {
			uint32_t c = 0;
			if ((threadIdx.x & 63) == 0) c = atomicAdd(cursor, (uint32_t)(WAVE * GRAB));
			cb = __builtin_amdgcn_readfirstlane(c) + stride0;
			ce = cb + WAVE * GRAB;
		}
		const uint32_t b = cb;
		cb += WAVE;
		return b;
	}
};

// ---- depth 0: generate + trace primary rays -----------------------------------------------------------------------
template <int MODE, bool OVF>
__global__ __launch_bounds__(BLOCK) void k_primary(DScene S, DFrame F, uint8_t* __restrict__ out, int launch) {
	__shared__ uint32_t s_stack[MODE == MODE_REFWALK ? 1 : STACK_LDS * BLOCK];
	LdsStack<OVF> stk;
	stk.lds = s_stack + (MODE == MODE_REFWALK ? 0 : threadIdx.x);
	stk.ovf = OVF ? S.ovf + ((size_t)blockIdx.x * BLOCK + threadIdx.x) : nullptr;
	stk.ovfStride = S.ovfStride;
	Dealer<4> deal(F.counters + CTR_CURSOR(launch));
	for (;;) {
		const uint32_t base = deal.next();
		if (base >= F.nSlots) return;
		const uint32_t slot = base + (threadIdx.x & 63);
		uint32_t px, py, inTile;
		const bool live = slot_to_pixel(F, slot, px, py, inTile);
		Hit h;
		h.ref = REF_NONE;
		f3 o = F3(0, 0, 0), d = F3(0, 0, 1);
		RayCtx r;
		if (live) {
			primary_ray(S, px, py, F.width, F.height, o, d);
			r = make_ray(o, d);
			trace_ray<MODE, false, OVF>(S, r, 0.0f, h, stk);
		}
		const bool hit = live && h.ref != REF_NONE;
		if (live && (!hit || F.debug == D_PRIMARY_MAP)) {
			// background is black (main.cpp:124,180); PRIMARY_INTERSECTION_MAP paints hits pure red (main.cpp:183-185)
			const size_t k = out_offset(F, slot, px, py, inTile);
			out[k] = 0;
			out[k + 1] = 0;
			out[k + 2] = hit ? 255 : 0;
		}
		const bool enq = hit && F.debug != D_PRIMARY_MAP;
		const uint32_t q = queue_push(F.counters + CTR_NHITS(0), enq);
		if (enq) {
			const f3 n = hit_normal(S, r, h.ref);
			F.hitO[q] = make_float4(o.x, o.y, o.z, __uint_as_float(slot));
			F.hitD[q] = make_float4(d.x, d.y, d.z, h.dist);
			F.hitN[q] = make_float4(n.x, n.y, n.z, __uint_as_float(h.ref));
			F.hit0[q] = slot;
		}
	}
}

// ---- depth >= 1: trace queued reflection rays ---------------------------------------------------------------------
template <int MODE, bool OVF>
__global__ __launch_bounds__(BLOCK) void k_trace(DScene S, DFrame F, int depth, int launch) {
	__shared__ uint32_t s_stack[MODE == MODE_REFWALK ? 1 : STACK_LDS * BLOCK];
	LdsStack<OVF> stk;
	stk.lds = s_stack + (MODE == MODE_REFWALK ? 0 : threadIdx.x);
	stk.ovf = OVF ? S.ovf + ((size_t)blockIdx.x * BLOCK + threadIdx.x) : nullptr;
	stk.ovfStride = S.ovfStride;
	const uint32_t n = F.counters[CTR_NRAYS(depth)];
	const size_t half = (size_t)(depth & 1) * F.nSlots;
	Dealer<1> deal(F.counters + CTR_CURSOR(launch));
	for (;;) {
		const uint32_t base = deal.next();
		if (base >= n) return;
		const uint32_t i = base + (threadIdx.x & 63);
		const bool live = i < n;
		Hit h;
		h.ref = REF_NONE;
		float4 ro = make_float4(0, 0, 0, 0), rd = make_float4(0, 0, 1, 0);
		RayCtx r;
		if (live) {
			ro = F.rayO[half + i];
			rd = F.rayD[half + i];
			r = make_ray(F3(ro), F3(rd));
			trace_ray<MODE, false, OVF>(S, r, 0.0f, h, stk);
		}
		const bool hit = live && h.ref != REF_NONE;
		const uint32_t q = queue_push(F.counters + CTR_NHITS(depth), hit);
		if (hit) {
			const f3 nn = hit_normal(S, r, h.ref);
			F.hitO[q] = ro;
			F.hitD[q] = make_float4(rd.x, rd.y, rd.z, h.dist);
			F.hitN[q] = make_float4(nn.x, nn.y, nn.z, __uint_as_float(h.ref));
		}
	}
}

// ---- shade the hits of one depth: lights, shadow rays, spawn the reflection ray -------------------------------------
template <int MODE, bool OVF>
__global__ __launch_bounds__(BLOCK) void k_shade(DScene S, DFrame F, int depth, int launch) {
	__shared__ uint32_t s_stack[MODE == MODE_REFWALK ? 1 : STACK_LDS * BLOCK];
	LdsStack<OVF> stk;
	stk.lds = s_stack + (MODE == MODE_REFWALK ? 0 : threadIdx.x);
	stk.ovf = OVF ? S.ovf + ((size_t)blockIdx.x * BLOCK + threadIdx.x) : nullptr;
	stk.ovfStride = S.ovfStride;
	const uint32_t n = F.counters[CTR_NHITS(depth)];
	const size_t nextHalf = (size_t)((depth + 1) & 1) * F.nSlots;
	const bool spawn = (F.features & F_REFLECTIONS) && (depth + 1 <= S.maxDepth) && F.debug != D_PRIMARY_MAP; // main.cpp:177,189
	Dealer<1> deal(F.counters + CTR_CURSOR(launch));
	for (;;) {
		const uint32_t base = deal.next();
		if (base >= n) return;
		const uint32_t i = base + (threadIdx.x & 63);
		const bool live = i < n;
		f3 P = F3(0, 0, 0), R = F3(0, 0, 1);
		uint32_t slot = 0;
		if (live) {
			const float4 ho = F.hitO[i], hd = F.hitD[i], hn = F.hitN[i];
			const f3 o = F3(ho), d = F3(hd), N = F3(hn);
			slot = __float_as_uint(ho.w);
			const float dist = hd.w;
			const uint32_t ref = __float_as_uint(hn.w);
			const float4* mat = S.materials + (size_t)info_of(S, ref).y * 4;
			const float4 mDiff = mat[0], mSpec = mat[1], mEmis = mat[2], mAmb = mat[3];
			const float shininess = mDiff.w;
			P = o + normalize3(d) * dist;                                   // Camera.cpp:26-28
			f3 sum = F3(mAmb) + F3(mEmis);                                  // main.cpp:204
			for (uint32_t li = 0; li < S.nLights; ++li) {                   // :208
				const float4 ll = S.lights[li * 2], lc = S.lights[li * 2 + 1];
				f3 L;
				float atten;
				if (ll.w > 0.0f) {                                          // :213
					L = F3(ll) - P;                                         // :214
					const float dl = length3(L);                            // :215
					atten = 1.0f / (S.atten[0] + S.atten[1] * dl + S.atten[2] * (dl * dl)); // :266-268
				}
				else {
					L = F3(ll);                                             // :219
					atten = 1.0f;
				}
				const f3 Ln = normalize3(L);                                // :222
				const float lenL = length3(L);                              // :224
				bool lit = true;
				Hit occ;
				occ.ref = REF_NONE;
				if (F.features & F_SHADOWS) {
					const RayCtx sr = make_ray(P, Ln);
					if (F.debug == D_SHADOW_MAP) {
						trace_ray<MODE, false, OVF>(S, sr, 0.0f, occ, stk);
						lit = (occ.ref == REF_NONE) || (occ.dist >= lenL);
					}
					else {
						lit = !trace_ray<MODE, true, OVF>(S, sr, lenL, occ, stk);
					}
				}
				if (lit) {
					const float kd = fmaxf(dot3(normalize3(N), normalize3(L)), 0.0f); // :225, :258-260
					const f3 eye = o - P;                                              // :226
					const f3 H = normalize3(Ln + normalize3(eye));                     // :227
					// :262-264 powf; evaluated in double and rounded once, which reproduces a correctly rounded powf
					const float ks = (float)pow((double)fmaxf(dot3(H, N), 0.0f), (double)shininess);
					if (F.debug == D_DIFFUSE_INTENSITY) sum = sum + F3(kd, kd, kd);
					else if (F.debug == D_SPECULAR_INTENSITY) sum = sum + F3(ks, ks, ks);
					else if (F.debug == D_NORMAL_MAP) sum = sum + N;
					else if (F.debug == D_LIGHT_DIR_MAP) sum = sum + H;
					else {
						if (F.features & F_DIFFUSE) sum = sum + ((F3(mDiff) * atten) * kd) * F3(lc);  // :243
						if (F.features & F_SPECULAR) sum = sum + ((F3(mSpec) * atten) * ks) * F3(lc); // :246
					}
				}
				else if (F.debug == D_SHADOW_MAP) {
					const float4* om = S.materials + (size_t)info_of(S, occ.ref).y * 4;
					sum = sum + F3(om[0]);                                  // :250-252
				}
			}
			const size_t k = (size_t)depth * F.nSlots + slot;
			F.litL[k] = make_float4(sum.x, sum.y, sum.z, 0.0f);
			F.litS[k] = make_float4(mSpec.x, mSpec.y, mSpec.z, 0.0f);
			F.depthCnt[slot] = (uint8_t)(depth + 1);
			R = normalize3(d - N * (2.0f * dot3(d, N)));                    // :188
		}
		if (spawn) {
			const uint32_t q = queue_push(F.counters + CTR_NRAYS(depth + 1), live);
			if (live) {
				F.rayO[nextHalf + q] = make_float4(P.x, P.y, P.z, __uint_as_float(slot));
				F.rayD[nextHalf + q] = make_float4(R.x, R.y, R.z, 0.0f);
			}
		}
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
	const uint32_t n = F.counters[CTR_NHITS(0)];
	for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
		const uint32_t slot = F.hit0[i];
		const int cnt = F.depthCnt[slot];
		f3 C = F3(0, 0, 0); // Color() past the last bounce (main.cpp:199) or the black background (main.cpp:124)
		for (int k = cnt - 1; k >= 0; --k) {
			const float4 L = F.litL[(size_t)k * F.nSlots + slot];
			const float4 Sp = F.litS[(size_t)k * F.nSlots + slot];
			if ((F.features & F_REFLECTIONS) || k > 0) C = F3(L) + F3(Sp) * C;
			else C = F3(L); // main.cpp:193
		}
		uint32_t px, py, inTile;
		slot_to_pixel(F, slot, px, py, inTile);
		const size_t o = out_offset(F, slot, px, py, inTile);
		out[o] = quantise(C.z);     // B  main.cpp:154
		out[o + 1] = quantise(C.y); // G
		out[o + 2] = quantise(C.x); // R
	}
}

// ---- narrow seam: closest hit for a list of rays ----------------------------------------------------------------------
template <int MODE, bool OVF>
__global__ __launch_bounds__(BLOCK) void k_trace_list(DScene S, uint64_t n, const float* __restrict__ o3, const float* __restrict__ d3,
                                                      int32_t* __restrict__ primId, float* __restrict__ dist, float* __restrict__ normal) {
	__shared__ uint32_t s_stack[MODE == MODE_REFWALK ? 1 : STACK_LDS * BLOCK];
	LdsStack<OVF> stk;
	stk.lds = s_stack + (MODE == MODE_REFWALK ? 0 : threadIdx.x);
	stk.ovf = OVF ? S.ovf + ((size_t)blockIdx.x * BLOCK + threadIdx.x) : nullptr;
	stk.ovfStride = S.ovfStride;
	for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
		const f3 o = F3(o3 + 3 * i), d = F3(d3 + 3 * i);
		const RayCtx r = make_ray(o, d);
		Hit h;
		trace_ray<MODE, false, OVF>(S, r, 0.0f, h, stk);
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
__global__ __launch_bounds__(BLOCK) void k_primary_ids(DScene S, uint32_t width, uint32_t height, int32_t* __restrict__ primId, float* __restrict__ dist) {
	__shared__ uint32_t s_stack[MODE == MODE_REFWALK ? 1 : STACK_LDS * BLOCK];
	LdsStack<OVF> stk;
	stk.lds = s_stack + (MODE == MODE_REFWALK ? 0 : threadIdx.x);
	stk.ovf = OVF ? S.ovf + ((size_t)blockIdx.x * BLOCK + threadIdx.x) : nullptr;
	stk.ovfStride = S.ovfStride;
	const uint64_t n = (uint64_t)width * height;
	for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
		f3 o, d;
		primary_ray(S, (uint32_t)(i % width), (uint32_t)(i / width), width, height, o, d);
		const RayCtx r = make_ray(o, d);
		Hit h;
		trace_ray<MODE, false, OVF>(S, r, 0.0f, h, stk);
		primId[i] = h.ref != REF_NONE ? (int32_t)info_of(S, h.ref).x : -1;
		dist[i] = h.ref != REF_NONE ? h.dist : __builtin_inff();
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
