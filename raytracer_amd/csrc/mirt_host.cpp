// Host side of the trace path: scene text -> primitives -> the reference's Partition tree, flattened (include/mirt_host.h).
//
// Everything here must produce the numbers the reference produces, bit for bit: the device breaks ties by the position
// of a primitive in the reference's leaf order, and a leaf box that differs in its last bit changes which rays reach it.
// So the arithmetic follows the reference's expressions operation by operation (GLM 0.9.8.5 evaluation order spelled out
// below) and this file is compiled without floating-point contraction.  It is plain C++ (no HIP): it runs on CPU threads.
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/mirt_host.h"
#include "mirt_internal.h"
#include "mirt_pool.h"

namespace {

// ---- GLM 0.9.8.5 arithmetic, in its evaluation order -----------------------------------------------------------------------
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
struct M4 { V4 c[4]; }; // column-major: c[col]

inline V3 v3(float x, float y, float z) { V3 r = { x, y, z }; return r; }
inline V4 v4(float x, float y, float z, float w) { V4 r = { x, y, z, w }; return r; }
inline V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V4 mul(V4 a, float s) { return v4(a.x * s, a.y * s, a.z * s, a.w * s); }
inline V4 mul(V4 a, V4 b) { return v4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline V4 add(V4 a, V4 b) { return v4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline V4 sub(V4 a, V4 b) { return v4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
// func_geometric.inl compute_dot<tvec3>: tmp = a * b; tmp.x + tmp.y + tmp.z
inline float dot(V3 a, V3 b) {
	const float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
	return tx + ty + tz;
}
// func_geometric.inl compute_cross
inline V3 cross(V3 x, V3 y) { return v3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
// normalize = v * inversesqrt(dot(v, v)), inversesqrt = 1 / sqrt
inline V3 normalize(V3 v) {
	const float inv = 1.0f / sqrtf(dot(v, v));
	return v3(v.x * inv, v.y * inv, v.z * inv);
}
// type_mat4x4.inl operator*(mat4, vec4): (m[0]*v.x + m[1]*v.y) + (m[2]*v.z + m[3]*v.w)
inline V4 mul(const M4& m, V4 v) {
	const V4 add0 = add(mul(m.c[0], v.x), mul(m.c[1], v.y));
	const V4 add1 = add(mul(m.c[2], v.z), mul(m.c[3], v.w));
	return add(add0, add1);
}
// type_mat4x4.inl:595-613 operator*(mat4, mat4): ((A0*b.x + A1*b.y) + A2*b.z) + A3*b.w per column
inline M4 mul(const M4& a, const M4& b) {
	M4 r;
	for (int j = 0; j < 4; ++j)
		r.c[j] = add(add(add(mul(a.c[0], b.c[j].x), mul(a.c[1], b.c[j].y)), mul(a.c[2], b.c[j].z)), mul(a.c[3], b.c[j].w));
	return r;
}
inline float m(const M4& a, int col, int row) { return (&a.c[col].x)[row]; }
// func_matrix.inl:297-355 compute_inverse<tmat4x4>
M4 inverse(const M4& a) {
	const float c00 = m(a, 2, 2) * m(a, 3, 3) - m(a, 3, 2) * m(a, 2, 3);
	const float c02 = m(a, 1, 2) * m(a, 3, 3) - m(a, 3, 2) * m(a, 1, 3);
	const float c03 = m(a, 1, 2) * m(a, 2, 3) - m(a, 2, 2) * m(a, 1, 3);
	const float c04 = m(a, 2, 1) * m(a, 3, 3) - m(a, 3, 1) * m(a, 2, 3);
	const float c06 = m(a, 1, 1) * m(a, 3, 3) - m(a, 3, 1) * m(a, 1, 3);
	const float c07 = m(a, 1, 1) * m(a, 2, 3) - m(a, 2, 1) * m(a, 1, 3);
	const float c08 = m(a, 2, 1) * m(a, 3, 2) - m(a, 3, 1) * m(a, 2, 2);
	const float c10 = m(a, 1, 1) * m(a, 3, 2) - m(a, 3, 1) * m(a, 1, 2);
	const float c11 = m(a, 1, 1) * m(a, 2, 2) - m(a, 2, 1) * m(a, 1, 2);
	const float c12 = m(a, 2, 0) * m(a, 3, 3) - m(a, 3, 0) * m(a, 2, 3);
	const float c14 = m(a, 1, 0) * m(a, 3, 3) - m(a, 3, 0) * m(a, 1, 3);
	const float c15 = m(a, 1, 0) * m(a, 2, 3) - m(a, 2, 0) * m(a, 1, 3);
	const float c16 = m(a, 2, 0) * m(a, 3, 2) - m(a, 3, 0) * m(a, 2, 2);
	const float c18 = m(a, 1, 0) * m(a, 3, 2) - m(a, 3, 0) * m(a, 1, 2);
	const float c19 = m(a, 1, 0) * m(a, 2, 2) - m(a, 2, 0) * m(a, 1, 2);
	const float c20 = m(a, 2, 0) * m(a, 3, 1) - m(a, 3, 0) * m(a, 2, 1);
	const float c22 = m(a, 1, 0) * m(a, 3, 1) - m(a, 3, 0) * m(a, 1, 1);
	const float c23 = m(a, 1, 0) * m(a, 2, 1) - m(a, 2, 0) * m(a, 1, 1);
	const V4 f0 = v4(c00, c00, c02, c03), f1 = v4(c04, c04, c06, c07), f2 = v4(c08, c08, c10, c11);
	const V4 f3 = v4(c12, c12, c14, c15), f4 = v4(c16, c16, c18, c19), f5 = v4(c20, c20, c22, c23);
	const V4 e0 = v4(m(a, 1, 0), m(a, 0, 0), m(a, 0, 0), m(a, 0, 0));
	const V4 e1 = v4(m(a, 1, 1), m(a, 0, 1), m(a, 0, 1), m(a, 0, 1));
	const V4 e2 = v4(m(a, 1, 2), m(a, 0, 2), m(a, 0, 2), m(a, 0, 2));
	const V4 e3 = v4(m(a, 1, 3), m(a, 0, 3), m(a, 0, 3), m(a, 0, 3));
	const V4 i0 = add(sub(mul(e1, f0), mul(e2, f1)), mul(e3, f2));
	const V4 i1 = add(sub(mul(e0, f0), mul(e2, f3)), mul(e3, f4));
	const V4 i2 = add(sub(mul(e0, f1), mul(e1, f3)), mul(e3, f5));
	const V4 i3 = add(sub(mul(e0, f2), mul(e1, f4)), mul(e2, f5));
	const V4 sa = v4(+1, -1, +1, -1), sb = v4(-1, +1, -1, +1);
	M4 inv;
	inv.c[0] = mul(i0, sa);
	inv.c[1] = mul(i1, sb);
	inv.c[2] = mul(i2, sa);
	inv.c[3] = mul(i3, sb);
	const V4 row0 = v4(inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x);
	const V4 d0 = mul(a.c[0], row0);
	const float det = (d0.x + d0.y) + (d0.z + d0.w);
	const float ood = 1.0f / det;
	for (int j = 0; j < 4; ++j) inv.c[j] = mul(inv.c[j], ood);
	return inv;
}
M4 identity() {
	M4 r;
	r.c[0] = v4(1, 0, 0, 0);
	r.c[1] = v4(0, 1, 0, 0);
	r.c[2] = v4(0, 0, 1, 0);
	r.c[3] = v4(0, 0, 0, 1);
	return r;
}
// Transform.h:5-12
M4 translation(float x, float y, float z) {
	M4 r = identity();
	r.c[3] = v4(x, y, z, 1.0f);
	return r;
}
// Transform.h:14-16
M4 scaling(float x, float y, float z) {
	M4 r = identity();
	r.c[0].x = x;
	r.c[1].y = y;
	r.c[2].z = z;
	return r;
}
// Transform.h:18-30: Rodrigues' formula as three 3x3 matrices, each scaled element by element, added left to right
M4 rotation(float ax, float ay, float az, float degrees) {
	const float rad = degrees * 0.01745329251994329576923690768489f; // glm::radians
	const V3 u = normalize(v3(ax, ay, az));
	const float c = cosf(rad), s = sinf(rad), omc = 1.0f - cosf(rad);
	// the three matrices after glm::transpose, as [col][row]
	const float I[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
	const float xx = powf(u.x, 2.0f), yy = powf(u.y, 2.0f), zz = powf(u.z, 2.0f);
	const float A[3][3] = { { xx, u.x * u.y, u.x * u.z }, { u.x * u.y, yy, u.y * u.z }, { u.x * u.z, u.y * u.z, zz } };
	const float B[3][3] = { { 0.0f, u.z, -u.y }, { -u.z, 0.0f, u.x }, { u.y, -u.x, 0.0f } };
	M4 r = identity();
	for (int col = 0; col < 3; ++col) {
		float* out = &r.c[col].x;
		for (int row = 0; row < 3; ++row) out[row] = (I[col][row] * c + A[col][row] * omc) + B[col][row] * s;
		out[3] = 0.0f;
	}
	return r;
}
inline float fmin2(float a, float b) { return (b < a) ? b : a; } // std::min
inline float fmax2(float a, float b) { return (a < b) ? b : a; } // std::max

// ---- primitives as the tree build sees them ---------------------------------------------------------------------------------
struct Box { V3 lo, hi; };

struct Prims {
	// per primitive id: triangle slot, or ~sphere slot
	std::vector<int64_t> slot;
	const float* triVerts = nullptr; // 9 per triangle, world space
	const mirt_sphere* spheres = nullptr;
	bool allFinite = false; // every triangle coordinate is finite: tri_in_box_finite applies
};

// Triangle::GetInterval(axis), Triangle.cpp:104-118
inline void tri_interval(const V3* v, V3 axis, float& lo, float& hi) {
	lo = dot(axis, v[0]);
	hi = lo;
	float value = dot(axis, v[1]);
	lo = fmin2(lo, value);
	hi = fmax2(hi, value);
	value = dot(axis, v[2]);
	lo = fmin2(lo, value);
	hi = fmax2(hi, value);
}
// Triangle::GetInterval(box, axis), Triangle.cpp:120-145
inline void box_interval(const Box& b, V3 axis, float& lo, float& hi) {
	// the eight corner projections (axis.x*X + axis.y*Y) + axis.z*Z of the reference, with the shared products and
	// partial sums computed once: every value is produced by the same operations in the same order, so it is identical
	const float x0 = axis.x * b.lo.x, x1 = axis.x * b.hi.x, y0 = axis.y * b.lo.y, y1 = axis.y * b.hi.y, z0 = axis.z * b.lo.z, z1 = axis.z * b.hi.z;
	const float s01 = x0 + y1, s00 = x0 + y0, s11 = x1 + y1, s10 = x1 + y0;
	// corner order of Triangle.cpp:124-133
	const float p[8] = { s01 + z1, s01 + z0, s00 + z1, s00 + z0, s11 + z1, s11 + z0, s10 + z1, s10 + z0 };
	lo = hi = p[0];
	for (int k = 1; k < 8; ++k) {
		lo = (p[k] < lo) ? p[k] : lo;
		hi = (p[k] > hi) ? p[k] : hi;
	}
}
// Triangle::OverlapOnAxis, Triangle.cpp:97-101
inline bool overlap_on_axis(const Box& b, const V3* v, V3 axis) {
	float alo, ahi, blo, bhi;
	box_interval(b, axis, alo, ahi);
	tri_interval(v, axis, blo, bhi);
	return (blo <= ahi) && (alo <= bhi);
}
// Triangle::isInside, Triangle.cpp:58-95: separating-axis test, 13 axes in the reference's order
bool tri_in_box(const Box& b, const float* tv) {
	const V3 v[3] = { v3(tv[0], tv[1], tv[2]), v3(tv[3], tv[4], tv[5]), v3(tv[6], tv[7], tv[8]) };
	const V3 f0 = sub(v[1], v[0]), f1 = sub(v[2], v[1]), f2 = sub(v[0], v[2]);
	const V3 u0 = v3(1.0f, 0.0f, 0.0f), u1 = v3(0.0f, 1.0f, 0.0f), u2 = v3(0.0f, 0.0f, 1.0f);
	if (!overlap_on_axis(b, v, u0)) return false;
	if (!overlap_on_axis(b, v, u1)) return false;
	if (!overlap_on_axis(b, v, u2)) return false;
	if (!overlap_on_axis(b, v, cross(f0, f1))) return false;
	const V3 us[3] = { u0, u1, u2 };
	const V3 fs[3] = { f0, f1, f2 };
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j)
			if (!overlap_on_axis(b, v, cross(us[i], fs[j]))) return false;
	return true;
}
// The same test for finite coordinates.  Ten of the thirteen axes have components that are exactly zero (the box
// normals, and the cross products of a box normal with an edge); a zero component contributes +-0 to the reference's
// sums, which changes no value except the sign of a zero, and no comparison.  Dropping those terms leaves every
// remaining operation as the reference performs it, so every decision is the reference's.
inline bool overlap2(float lo1, float hi1, float lo2, float hi2, float a1, float a2, float p0a, float p0b, float p1a, float p1b, float p2a, float p2b) {
	// box: a1*X + a2*Y over the four corners; triangle: a1*v.a + a2*v.b over the three vertices
	const float x0 = a1 * lo1, x1 = a1 * hi1, y0 = a2 * lo2, y1 = a2 * hi2;
	const float c0 = x0 + y0, c1 = x0 + y1, c2 = x1 + y0, c3 = x1 + y1;
	const float alo = fmin2(fmin2(c0, c1), fmin2(c2, c3)), ahi = fmax2(fmax2(c0, c1), fmax2(c2, c3));
	const float t0 = a1 * p0a + a2 * p0b, t1 = a1 * p1a + a2 * p1b, t2 = a1 * p2a + a2 * p2b;
	const float blo = fmin2(fmin2(t0, t1), t2), bhi = fmax2(fmax2(t0, t1), t2);
	return (blo <= ahi) && (alo <= bhi);
}
bool tri_in_box_finite(const Box& b, const float* tv) {
	const V3 v[3] = { v3(tv[0], tv[1], tv[2]), v3(tv[3], tv[4], tv[5]), v3(tv[6], tv[7], tv[8]) };
	// box normals: the intervals are the coordinate ranges themselves
	if (!(fmin2(fmin2(v[0].x, v[1].x), v[2].x) <= b.hi.x && b.lo.x <= fmax2(fmax2(v[0].x, v[1].x), v[2].x))) return false;
	if (!(fmin2(fmin2(v[0].y, v[1].y), v[2].y) <= b.hi.y && b.lo.y <= fmax2(fmax2(v[0].y, v[1].y), v[2].y))) return false;
	if (!(fmin2(fmin2(v[0].z, v[1].z), v[2].z) <= b.hi.z && b.lo.z <= fmax2(fmax2(v[0].z, v[1].z), v[2].z))) return false;
	const V3 f0 = sub(v[1], v[0]), f1 = sub(v[2], v[1]), f2 = sub(v[0], v[2]);
	if (!overlap_on_axis(b, v, cross(f0, f1))) return false;
	const V3 fs[3] = { f0, f1, f2 };
	// cross(u0, f) = (0, -f.z, f.y): a1 = -f.z on y, a2 = f.y on z
	for (int j = 0; j < 3; ++j)
		if (!overlap2(b.lo.y, b.hi.y, b.lo.z, b.hi.z, -fs[j].z, fs[j].y, v[0].y, v[0].z, v[1].y, v[1].z, v[2].y, v[2].z)) return false;
	// cross(u1, f) = (f.z, 0, -f.x): a1 = f.z on x, a2 = -f.x on z
	for (int j = 0; j < 3; ++j)
		if (!overlap2(b.lo.x, b.hi.x, b.lo.z, b.hi.z, fs[j].z, -fs[j].x, v[0].x, v[0].z, v[1].x, v[1].z, v[2].x, v[2].z)) return false;
	// cross(u2, f) = (-f.y, f.x, 0): a1 = -f.y on x, a2 = f.x on y
	for (int j = 0; j < 3; ++j)
		if (!overlap2(b.lo.x, b.hi.x, b.lo.y, b.hi.y, -fs[j].y, fs[j].x, v[0].x, v[0].y, v[1].x, v[1].y, v[2].x, v[2].y)) return false;
	return true;
}
inline M4 load_m4(const float* p) {
	M4 r;
	memcpy(&r, p, sizeof r);
	return r;
}
// AABB::contains(point), AABB.cpp:102-104
inline bool box_has_point(const Box& b, V3 p) { return p.x <= b.hi.x && p.x >= b.lo.x && p.y <= b.hi.y && p.y >= b.lo.y && p.z <= b.hi.z && p.z >= b.lo.z; }
// Sphere::isInside, Sphere.cpp:43-57 (the box corners go to object space, the centre test stays in world space, as written)
bool sphere_in_box(const Box& b, const mirt_sphere& s) {
	const M4 Mi = load_m4(s.Minv);
	const V4 lo4 = mul(Mi, v4(b.lo.x, b.lo.y, b.lo.z, 1.0f)), hi4 = mul(Mi, v4(b.hi.x, b.hi.y, b.hi.z, 1.0f));
	const V3 c = v3(s.center[0], s.center[1], s.center[2]);
	const float x = fmax2(lo4.x, fmin2(c.x, hi4.x));
	const float y = fmax2(lo4.y, fmin2(c.y, hi4.y));
	const float z = fmax2(lo4.z, fmin2(c.z, hi4.z));
	const float distance = sqrtf((x - c.x) * (x - c.x) + (y - c.y) * (y - c.y) + (z - c.z) * (z - c.z));
	return box_has_point(b, c) || distance < s.radius;
}
inline bool box_contains(const Prims& P, const Box& b, uint32_t prim) {
	const int64_t s = P.slot[prim];
	if (s >= 0) return P.allFinite ? tri_in_box_finite(b, P.triVerts + 9 * s) : tri_in_box(b, P.triVerts + 9 * s);
	if (s == INT64_MIN) return false; // neither triangle nor sphere: Shape::isInside is false (Shape.cpp)
	return sphere_in_box(b, P.spheres[~s]);
}
// Triangle::getMin*/getMax* (Triangle.cpp:33-55), Sphere::getMin*/getMax* (Sphere.cpp:13-41)
void prim_bounds(const Prims& P, uint32_t prim, float lo[3], float hi[3]) {
	const int64_t s = P.slot[prim];
	if (s >= 0) {
		const float* t = P.triVerts + 9 * s;
		for (int k = 0; k < 3; ++k) {
			lo[k] = fmin2(t[6 + k], fmin2(t[k], t[3 + k]));
			hi[k] = fmax2(t[6 + k], fmax2(t[k], t[3 + k]));
		}
		return;
	}
	const mirt_sphere& sp = P.spheres[~s];
	const M4 M = load_m4(sp.M);
	const float c[3] = { sp.center[0], sp.center[1], sp.center[2] };
	for (int k = 0; k < 3; ++k) {
		float a[4] = { 0.0f, 0.0f, 0.0f, 1.0f }, b[4] = { 0.0f, 0.0f, 0.0f, 1.0f };
		a[k] = c[k] - sp.radius;
		b[k] = c[k] + sp.radius;
		const V4 ra = mul(M, v4(a[0], a[1], a[2], a[3])), rb = mul(M, v4(b[0], b[1], b[2], b[3]));
		lo[k] = (&ra.x)[k];
		hi[k] = (&rb.x)[k];
	}
}

// ---- Partition build (Partition.cpp:4-14, 55-101) ---------------------------------------------------------------------------
struct PNode {
	Box box;
	PNode* left = nullptr;
	PNode* right = nullptr;
	std::vector<uint32_t> objs; // PartitionNode::objects, insertion order; released once the node has been split
	bool leaf = true;
	int32_t task = -1;          // the subtree below this node is built, flattened and freed as task number `task`
};

// The top of the tree -- nodes holding more than TASK_OBJS objects -- is split node by node, each node's objects
// classified and handed to the children on all threads (fixed chunks, children's lists filled in chunk order, so the
// lists are in insertion order whatever the thread count).  A node at or below TASK_OBJS is not split there: the whole
// subtree under it is one task for the thread pool, built by one thread's recursion.
struct Builder {
	const Prims& P;
	std::atomic<bool> oom;
	// Partition::split (Partition.cpp:77-101) does not always terminate.  When a box is one float step wide on its longest
	// axis, "max - (max - min) / 2" rounds to one of its ends, so one child gets the parent's own box (and with it the
	// parent's own object list), the other a flat one.  If fewer than half of the objects also touch the flat child, that
	// first child is split again in exactly the same way, for ever: the reference overflows its stack and crashes.  It
	// happens in practice once a vertex lies exactly on a split plane with its triangles all on the far side (the near
	// child then holds them by that one point and shrinks around it); the chance grows with |coordinate| / triangle size
	// and with the triangle count -- a 10 x 10 grid of Stanford dragons has such vertices, one dragon does not.
	// Here the repeating child is kept as a leaf and counted; everything above it is the reference's tree.
	std::atomic<uint64_t> truncated;
	static const size_t TASK_OBJS = 1u << 15;
	static const uint32_t GRAIN = 1u << 12;
	struct Task { PNode* node; int prevMatches; uint64_t nodes, refs, nodeBase, refBase; };
	std::vector<Task> tasks;
	bool serial; // one thread asked for: no pool, no tasks
	Builder(const Prims& p, int threads) : P(p), oom(false), truncated(0), serial(threads <= 1) {}

	// AABB::getLongestAxis / getMidPoint / splitLeft / splitRight, AABB.cpp:106-155
	static void split_boxes(const Box& b, Box& l, Box& r) {
		const float ex = b.hi.x - b.lo.x, ey = b.hi.y - b.lo.y, ez = b.hi.z - b.lo.z;
		const float maxAxis = fmax2(ez, fmax2(ex, ey));
		const int axis = (ez == maxAxis) ? 2 : (ey == maxAxis) ? 1 : 0;
		const V3 mid = v3(b.hi.x - (ex / 2.0f), b.hi.y - (ey / 2.0f), b.hi.z - (ez / 2.0f));
		l = b;
		r = b;
		if (axis == 0) { l.hi.x = mid.x; r.lo.x = mid.x; }
		else if (axis == 1) { l.hi.y = mid.y; r.lo.y = mid.y; }
		else { l.hi.z = mid.z; r.lo.z = mid.z; }
	}

	// which of the objects each box contains: bit 0 = lb, bit 1 = rb (Partition.cpp:82-88)
	void classify(const uint32_t* objs, size_t n, const Box& lb, const Box& rb, uint8_t* side) const {
		for (size_t i = 0; i < n; ++i) side[i] = (uint8_t)((box_contains(P, lb, objs[i]) ? 1 : 0) | (box_contains(P, rb, objs[i]) ? 2 : 0));
	}
	// the children's object lists (insertion order kept) and the number of objects in both; returns that number
	int distribute(const std::vector<uint32_t>& objs, const Box& lb, const Box& rb, std::vector<uint32_t>& lo, std::vector<uint32_t>& ro, bool top) const {
		const size_t n = objs.size();
		std::vector<uint8_t> side(n);
		if (!top || n <= TASK_OBJS) {
			classify(objs.data(), n, lb, rb, side.data());
			size_t nl = 0, nr = 0;
			for (size_t i = 0; i < n; ++i) { nl += side[i] & 1; nr += side[i] >> 1; }
			lo.reserve(nl);
			ro.reserve(nr);
			int matches = 0;
			for (size_t i = 0; i < n; ++i) {
				if (side[i] & 1) lo.push_back(objs[i]);
				if (side[i] & 2) ro.push_back(objs[i]);
				if (side[i] == 3) ++matches;
			}
			return matches;
		}
		const uint32_t chunks = (uint32_t)((n + GRAIN - 1) / GRAIN);
		struct Cnt { size_t l, r, m; };
		std::vector<Cnt> cnt((size_t)chunks + 1, Cnt{0, 0, 0});
		MirtPool::get().run(chunks, [&](uint32_t c) {
			const size_t a = (size_t)c * GRAIN, b = a + GRAIN < n ? a + GRAIN : n;
			classify(objs.data() + a, b - a, lb, rb, side.data() + a);
			Cnt k = {0, 0, 0};
			for (size_t i = a; i < b; ++i) { k.l += side[i] & 1; k.r += side[i] >> 1; k.m += side[i] == 3 ? 1u : 0u; }
			cnt[c + 1] = k;
		});
		for (uint32_t c = 0; c < chunks; ++c) { cnt[c + 1].l += cnt[c].l; cnt[c + 1].r += cnt[c].r; cnt[c + 1].m += cnt[c].m; }
		lo.resize(cnt[chunks].l);
		ro.resize(cnt[chunks].r);
		MirtPool::get().run(chunks, [&](uint32_t c) {
			const size_t a = (size_t)c * GRAIN, b = a + GRAIN < n ? a + GRAIN : n;
			uint32_t* pl = lo.data() + cnt[c].l;
			uint32_t* pr = ro.data() + cnt[c].r;
			for (size_t i = a; i < b; ++i) {
				if (side[i] & 1) *pl++ = objs[i];
				if (side[i] & 2) *pr++ = objs[i];
			}
		});
		return (int)cnt[chunks].m;
	}

	// Partition::split, Partition.cpp:77-101.  top: on the calling thread near the root (subtrees become tasks)
	void split(PNode* node, int prevMatches, bool top) {
		if (node == nullptr || oom.load(std::memory_order_relaxed)) return;
		if (top && !serial && node->objs.size() <= TASK_OBJS) {
			node->task = (int32_t)tasks.size();
			tasks.push_back(Task{node, prevMatches, 0, 0, 0, 0});
			return;
		}
		if (!(((float)prevMatches / (float)node->objs.size()) < 0.5f)) return; // 0/0 is NaN: an empty node is never split
		Box lb, rb;
		split_boxes(node->box, lb, rb);
		PNode *l = nullptr, *r = nullptr;
		int matches = 0;
		try {
			l = new PNode();
			r = new PNode();
			l->box = lb;
			r->box = rb;
			matches = distribute(node->objs, lb, rb, l->objs, r->objs, top && !serial);
		} catch (const std::bad_alloc&) {
			delete l;
			delete r;
			oom.store(true);
			return;
		}
		const size_t nObjs = node->objs.size();
		std::vector<uint32_t>().swap(node->objs);
		node->leaf = false;
		// Partition.cpp:90-97: an empty child next to a non-empty one is deleted; two empty children both stay
		if (l->objs.empty() && !r->objs.empty()) { delete l; l = nullptr; }
		else if (r->objs.empty() && !l->objs.empty()) { delete r; r = nullptr; }
		node->left = l;
		node->right = r;
		// a child with the parent's box repeats the parent's split, matches included: see `truncated`
		if (((float)matches / (float)nObjs) < 0.5f) {
			const bool sameL = l && memcmp(&l->box, &node->box, sizeof(Box)) == 0, sameR = r && memcmp(&r->box, &node->box, sizeof(Box)) == 0;
			if (sameL || sameR) {
				truncated.fetch_add(1);
				if (!sameL) split(l, matches, top);
				if (!sameR) split(r, matches, top);
				return;
			}
		}
		split(l, matches, top);
		split(r, matches, top);
	}

	// the subtrees left to the pool, each by one thread; afterwards every task knows the size of its subtree
	void run_tasks() {
		MirtPool::get().run((uint32_t)tasks.size(), [&](uint32_t t) {
			Task& k = tasks[t];
			try {
				split(k.node, k.prevMatches, false);
				std::vector<const PNode*> st(1, k.node);
				while (!st.empty()) {
					const PNode* x = st.back();
					st.pop_back();
					++k.nodes;
					if (x->leaf) k.refs += x->objs.size();
					if (x->left) st.push_back(x->left);
					if (x->right) st.push_back(x->right);
				}
			} catch (const std::bad_alloc&) {
				oom.store(true);
			}
		});
	}
};

// the nodes of a subtree; explicit stack: single-child chains can be long.  Children that are tasks are left alone.
void free_tree(PNode* n, bool skipTasks = false) {
	std::vector<PNode*> st;
	if (n) st.push_back(n);
	while (!st.empty()) {
		PNode* x = st.back();
		st.pop_back();
		if (x->left && !(skipTasks && x->left->task >= 0)) st.push_back(x->left);
		if (x->right && !(skipTasks && x->right->task >= 0)) st.push_back(x->right);
		delete x;
	}
}

} // namespace

struct mirt_host_scene {
	std::vector<mirt_node> nodes;
	std::vector<uint32_t> leafPrims;
	std::vector<float> triVerts;
	std::vector<uint32_t> triPrim;
	std::vector<mirt_sphere> spheres;
	std::vector<uint32_t> sphPrim;
	std::vector<mirt_material> materials;
	std::vector<uint32_t> primMaterial;
	std::vector<mirt_light> lights;
	mirt_camera camera;
	float attenuation[3];
	int32_t maxDepth;
	uint32_t nPrims;
	uint32_t width, height;
	bool haveSize, haveCamera;
	std::string outputName;
	double parseSeconds, buildSeconds;
	uint64_t truncatedNodes;
};

namespace {

// One subtree in depth-first pre-order, left first, into S.nodes from index nextNode and S.leafPrims from nextRef; explicit
// stack (chains of single children make the tree deep).  parent/side: where the subtree root's index is written (-1: nowhere).
// With `tasks`, a node that is a task's root is not descended into: its place is reserved (the task's sizes are known),
// its index written to its parent, and the task flattens its own subtree later.
void flatten_subtree(PNode* root, mirt_host_scene& S, uint64_t nextNode, uint64_t nextRef, uint32_t parent, int side, std::vector<Builder::Task>* tasks) {
	struct Item { PNode* n; uint32_t parent; int side; };
	std::vector<Item> st;
	st.push_back(Item{ root, parent, side });
	while (!st.empty()) {
		const Item it = st.back();
		st.pop_back();
		const uint32_t me = (uint32_t)nextNode;
		if (it.side == 0) S.nodes[it.parent].a = me;
		else if (it.side == 1) S.nodes[it.parent].b = me;
		if (tasks && it.n->task >= 0) {
			Builder::Task& k = (*tasks)[(size_t)it.n->task];
			k.nodeBase = nextNode;
			k.refBase = nextRef;
			nextNode += k.nodes;
			nextRef += k.refs;
			continue;
		}
		++nextNode;
		mirt_node& rec = S.nodes[me];
		rec.bmin[0] = it.n->box.lo.x; rec.bmin[1] = it.n->box.lo.y; rec.bmin[2] = it.n->box.lo.z;
		rec.bmax[0] = it.n->box.hi.x; rec.bmax[1] = it.n->box.hi.y; rec.bmax[2] = it.n->box.hi.z;
		if (it.n->leaf) {
			rec.a = (uint32_t)nextRef;
			rec.b = MIRT_LEAF_FLAG | (uint32_t)it.n->objs.size();
			for (uint32_t o : it.n->objs) S.leafPrims[nextRef++] = o;
		}
		else {
			rec.a = MIRT_NO_CHILD;
			rec.b = MIRT_NO_CHILD;
			if (it.n->right) st.push_back(Item{ it.n->right, me, 1 }); // popped after the whole left subtree
			if (it.n->left) st.push_back(Item{ it.n->left, me, 0 });
		}
	}
}

int flatten(PNode* root, mirt_host_scene& S, Builder& B) {
	// sizes: the tasks know theirs, the top is counted here
	uint64_t nNodes = 0, nRefs = 0;
	{
		std::vector<PNode*> st(1, root);
		while (!st.empty()) {
			PNode* x = st.back();
			st.pop_back();
			if (x->task >= 0) { nNodes += B.tasks[(size_t)x->task].nodes; nRefs += B.tasks[(size_t)x->task].refs; continue; }
			++nNodes;
			if (x->leaf) nRefs += x->objs.size();
			if (x->left) st.push_back(x->left);
			if (x->right) st.push_back(x->right);
		}
	}
	if (nNodes >= MIRT_NO_CHILD || nRefs >= 0x7fffffffull) return mirt_set_error(MIRT_E_INVALID, "Partition tree too large for 31-bit indices: %llu nodes, %llu leaf entries", (unsigned long long)nNodes, (unsigned long long)nRefs);
	S.nodes.resize(nNodes);
	S.leafPrims.resize(nRefs);
	flatten_subtree(root, S, 0, 0, 0, -1, &B.tasks);
	MirtPool::get().run((uint32_t)B.tasks.size(), [&](uint32_t t) {
		const Builder::Task& k = B.tasks[t];
		flatten_subtree(k.node, S, k.nodeBase, k.refBase, 0, -1, nullptr); // (its index is in its parent's record already)
	});
	return MIRT_OK;
}

// Partition::Partition, Partition.cpp:4-14, then flatten.  Needs S.triVerts/triPrim/spheres/sphPrim/nPrims.
// threads: 1 = everything on the calling thread; anything else = the library's thread pool.
int build_tree(mirt_host_scene& S, uint32_t threads) {
	const auto t0 = std::chrono::steady_clock::now();
	Prims P;
	P.slot.assign(S.nPrims, INT64_MIN);
	P.triVerts = S.triVerts.data();
	P.spheres = S.spheres.data();
	for (size_t i = 0; i < S.triPrim.size(); ++i) P.slot[S.triPrim[i]] = (int64_t)i;
	for (size_t i = 0; i < S.sphPrim.size(); ++i) P.slot[S.sphPrim[i]] = ~(int64_t)i;
	P.allFinite = true;
	for (float c : S.triVerts) if (!isfinite(c)) { P.allFinite = false; break; }
	PNode* root = nullptr;
	int rc = MIRT_OK;
	Builder B(P, threads == 1 ? 1 : 2);
	try {
		// AABB box; starts as (+inf, -inf) (AABB.h:34-35), then expand over every object, AABB.cpp:157-164
		Box box;
		box.lo = v3(INFINITY, INFINITY, INFINITY);
		box.hi = v3(-INFINITY, -INFINITY, -INFINITY);
		for (uint32_t p = 0; p < S.nPrims; ++p) {
			if (P.slot[p] == INT64_MIN) continue;
			float lo[3], hi[3];
			prim_bounds(P, p, lo, hi);
			if (lo[0] < box.lo.x) box.lo.x = lo[0];
			if (lo[1] < box.lo.y) box.lo.y = lo[1];
			if (lo[2] < box.lo.z) box.lo.z = lo[2];
			if (hi[0] > box.hi.x) box.hi.x = hi[0];
			if (hi[1] > box.hi.y) box.hi.y = hi[1];
			if (hi[2] > box.hi.z) box.hi.z = hi[2];
		}
		root = new PNode();
		root->box = box;
		// insert(obj, root): the root is a leaf, the object is kept iff root.box.contains(obj), Partition.cpp:55-59
		{
			std::vector<uint32_t> all(S.nPrims), none;
			for (uint32_t p = 0; p < S.nPrims; ++p) all[p] = p;
			(void)B.distribute(all, box, box, root->objs, none, !B.serial);
		}
		B.split(root, 0, true);
		B.run_tasks();
		if (B.oom.load()) rc = mirt_set_error(MIRT_E_NOMEM, "out of host memory while building the Partition tree");
		else { S.truncatedNodes = B.truncated.load(); rc = flatten(root, S, B); }
	} catch (const std::bad_alloc&) {
		rc = mirt_set_error(MIRT_E_NOMEM, "out of host memory while building the Partition tree");
	} catch (const std::exception& e) {
		rc = mirt_set_error(MIRT_E_INVALID, "Partition build failed: %s", e.what());
	}
	// the top's nodes, then every task's subtree on the pool
	if (root && root->task < 0) free_tree(root, true); // (a small scene's root is itself a task)
	MirtPool::get().run((uint32_t)B.tasks.size(), [&](uint32_t t) { free_tree(B.tasks[t].node); });
	S.buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	return rc;
}

// ---- scene text (Scene.cpp:15-224) -------------------------------------------------------------------------------------------
inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

// What `stringstream >> float` accepts: optional sign, digits with at most one '.', optional exponent; the collected
// characters must convert completely.  Returns false (the stream's failbit) otherwise.
bool read_float(const char*& p, const char* end, float& out) {
	while (p < end && is_space(*p)) ++p;
	char buf[64];
	int n = 0;
	const char* q = p;
	if (q < end && (*q == '+' || *q == '-')) buf[n++] = *q++;
	bool dot = false, expo = false, digits = false;
	while (q < end && n < 62) {
		const char ch = *q;
		if (ch >= '0' && ch <= '9') { buf[n++] = ch; digits = true; }
		else if (ch == '.' && !dot && !expo) { buf[n++] = ch; dot = true; }
		else if ((ch == 'e' || ch == 'E') && !expo && digits) {
			buf[n++] = ch;
			expo = true;
			if (q + 1 < end && (q[1] == '+' || q[1] == '-')) buf[n++] = *++q;
		}
		else break;
		++q;
	}
	p = q;
	if (!digits) return false;
	buf[n] = 0;
	char* stop = nullptr;
	const float v = strtof(buf, &stop);
	if (stop == buf || *stop != 0) return false;
	if (isinf(v)) return false; // out of range sets failbit
	out = v;
	return true;
}
bool read_vals(const char*& p, const char* end, int n, float* values) { // Scene::readvals, Scene.cpp:235-244
	for (int i = 0; i < n; ++i)
		if (!read_float(p, end, values[i])) return false;
	return true;
}

struct MatKey {
	uint32_t w[13];
	bool operator==(const MatKey& o) const { return memcmp(w, o.w, sizeof w) == 0; }
};
struct MatHash {
	size_t operator()(const MatKey& k) const {
		uint64_t h = 1469598103934665603ull;
		for (int i = 0; i < 13; ++i) { h ^= k.w[i]; h *= 1099511628211ull; }
		return (size_t)h;
	}
};
struct MaterialTable {
	std::unordered_map<MatKey, uint32_t, MatHash> index;
	uint32_t intern(mirt_host_scene& S, const mirt_material& mm) {
		MatKey k;
		memcpy(k.w, &mm, sizeof k.w);
		auto it = index.find(k);
		if (it != index.end()) return it->second;
		const uint32_t id = (uint32_t)S.materials.size();
		S.materials.push_back(mm);
		index.emplace(k, id);
		return id;
	}
};

void set_defaults(mirt_host_scene& S) {
	S.attenuation[0] = 1.0f; S.attenuation[1] = 0.0f; S.attenuation[2] = 0.0f; // Scene.cpp:10
	S.maxDepth = 5;                                                               // :11
	S.outputName = "test.png";                                                    // :12
	S.nPrims = 0;
	S.width = S.height = 0;
	S.haveSize = S.haveCamera = false;
	memset(&S.camera, 0, sizeof S.camera);
	S.parseSeconds = S.buildSeconds = 0.0;
	S.truncatedNodes = 0;
}

// A run of well-formed tri lines, taken in one go on all threads.  text..end starts at the beginning of a tri line.
// First the extent of the run is found -- the first line that does not begin with "tri" and a blank, looked for in
// rounds of 64 pieces of 64 KB (a glance per line, no parsing) -- then the run is cut into pieces at line ends and the
// pieces are parsed side by side.  The run ends before the first line that is anything but "tri a b c" with three numbers
// that index the vertex array (that line is left to the caller's loop, which knows what to do with it).  The triangles
// are appended exactly as the caller's loop would append them.
struct TriRun { uint64_t lines; size_t bytes; };
TriRun tri_run(const char* text, const char* end, const std::vector<V3>& verts, const M4& T, mirt_host_scene& S, uint32_t curMat) {
	const char* runEnd = text;
	{
		const size_t PIECE = (size_t)64 << 10;
		const uint32_t ROUND = 64;
		bool open = true;
		while (open && runEnd < end && (size_t)(runEnd - text) < ((size_t)1 << 30)) {
			// pieces of this round start at line starts: piece i begins at the first line start at or after runEnd + i * PIECE
			const char* starts[ROUND + 1];
			uint32_t n = 0;
			starts[0] = runEnd;
			while (n < ROUND && starts[n] < end) {
				const char* c = starts[n] + PIECE;
				const char* nl = c < end ? (const char*)memchr(c, '\n', (size_t)(end - c)) : nullptr;
				starts[++n] = nl ? nl + 1 : end;
			}
			const char* stop[ROUND];
			MirtPool::get().run(n, [&](uint32_t i) {
				const char* p = starts[i];
				const char* const pe = starts[i + 1];
				while (p < pe) {
					if (!(pe - p > 4 && p[0] == 't' && p[1] == 'r' && p[2] == 'i' && is_space(p[3]))) break;
					const char* eol = (const char*)memchr(p, '\n', (size_t)(pe - p));
					p = eol ? eol + 1 : pe;
				}
				stop[i] = p;
			});
			for (uint32_t i = 0; i < n; ++i) {
				runEnd = stop[i];
				if (stop[i] < starts[i + 1]) { open = false; break; }
			}
		}
	}
	if (runEnd == text) return TriRun{0, 0};
	const size_t want = (size_t)(runEnd - text) / (4 * (size_t)MirtPool::get().threads()) + 1;
	const size_t PIECE = want < ((size_t)64 << 10) ? ((size_t)64 << 10) : want;
	std::vector<const char*> cut; // piece i = [cut[i], cut[i+1])
	cut.push_back(text);
	while (cut.back() < runEnd) {
		const char* c = cut.back() + PIECE;
		if (c >= runEnd) { cut.push_back(runEnd); break; }
		const char* nl = (const char*)memchr(c, '\n', (size_t)(runEnd - c));
		cut.push_back(nl ? nl + 1 : runEnd);
	}
	const uint32_t pieces = (uint32_t)cut.size() - 1;
	struct Piece { std::vector<int> idx; size_t goodBytes; bool whole; };
	std::vector<Piece> pc(pieces);
	MirtPool::get().run(pieces, [&](uint32_t i) {
		Piece& k = pc[i];
		k.idx.reserve(3 * (size_t)(cut[i + 1] - cut[i]) / 12);
		const char* p = cut[i];
		const char* const pe = cut[i + 1];
		k.whole = true;
		while (p < pe) {
			const char* eol = (const char*)memchr(p, '\n', (size_t)(pe - p));
			const char* le = eol ? eol : pe;
			const char* q = p;
			float v[3];
			// exactly the plain case: "tri" in column 0, three values, indices inside the array
			bool good = le - p > 4 && p[0] == 't' && p[1] == 'r' && p[2] == 'i' && is_space(p[3]);
			if (good) { q = p + 3; good = read_vals(q, le, 3, v); }
			if (good) {
				const int i0 = (int)v[0], i1 = (int)v[1], i2 = (int)v[2];
				good = i0 >= 0 && i1 >= 0 && i2 >= 0 && (size_t)i0 < verts.size() && (size_t)i1 < verts.size() && (size_t)i2 < verts.size();
				if (good) { k.idx.push_back(i0); k.idx.push_back(i1); k.idx.push_back(i2); }
			}
			if (!good) { k.whole = false; break; }
			p = eol ? eol + 1 : pe;
		}
		k.goodBytes = (size_t)(p - cut[i]);
	});
	// the run: whole pieces, then the good start of the first piece that is not whole
	uint32_t used = 0;
	uint64_t lines = 0;
	size_t bytes = 0;
	std::vector<uint64_t> first(pieces + 1, 0);
	for (uint32_t i = 0; i < pieces; ++i) {
		first[i] = lines;
		lines += pc[i].idx.size() / 3;
		bytes += pc[i].goodBytes;
		++used;
		if (!pc[i].whole) break;
	}
	if (lines == 0) return TriRun{0, 0};
	const size_t tri0 = S.triPrim.size();
	const uint32_t prim0 = S.nPrims;
	S.triVerts.resize((tri0 + lines) * 9);
	S.triPrim.resize(tri0 + lines);
	S.primMaterial.resize(S.primMaterial.size() + lines, curMat);
	S.nPrims += (uint32_t)lines;
	MirtPool::get().run(used, [&](uint32_t i) {
		const Piece& k = pc[i];
		float* out = S.triVerts.data() + (tri0 + first[i]) * 9;
		for (size_t t = 0; t < k.idx.size() / 3; ++t) {
			for (int c = 0; c < 3; ++c) { // Triangle.cpp:23-25
				const V3 in = verts[(size_t)k.idx[3 * t + c]];
				const V4 w = mul(T, v4(in.x, in.y, in.z, 1.0f));
				*out++ = w.x; *out++ = w.y; *out++ = w.z;
			}
			S.triPrim[tri0 + first[i] + t] = prim0 + (uint32_t)(first[i] + t);
		}
	});
	return TriRun{lines, bytes};
}

int parse_text(const char* text, size_t len, mirt_host_scene& S) {
	const auto t0 = std::chrono::steady_clock::now();
	float diffuse[3] = { 0, 0, 0 }, specular[3] = { 0, 0, 0 }, emission[3] = { 0, 0, 0 }, ambient[3] = { 0.2f, 0.2f, 0.2f }; // Scene.cpp:16
	float shininess = 0;
	std::vector<V3> verts, vertNorms;
	bool haveVerts = false, haveNorms = false;
	size_t vertIndex = 0, normIndex = 0;
	std::vector<M4> stack(1, identity()); // :30-31
	MaterialTable mats;
	uint32_t curMat = 0;
	bool matDirty = true;
	const char* p = text;
	const char* const end = text + len;
	uint64_t lineNo = 0;
	uint32_t triStreak = 0; // tri lines in a row so far
	while (p < end) {
		const char* eol = (const char*)memchr(p, '\n', (size_t)(end - p));
		const char* le = eol ? eol : end;
		const char* q = p;
		const char* const ls = p;
		p = eol ? eol + 1 : end;
		++lineNo;
		while (q < le && is_space(*q)) ++q;
		if (q == le || *ls == '#') continue; // blank, or '#' in column 0 (Scene.cpp:34)
		const char* cb = q;
		while (q < le && !is_space(*q)) ++q;
		const size_t cl = (size_t)(q - cb);
		auto is = [&](const char* name) { return strlen(name) == cl && memcmp(cb, name, cl) == 0; };
		float v[10];
		if (is("tri")) { // :158-165
			// A long run of tri lines (a mesh) is taken by all threads: see tri_run.  Only after 64 of them in a row, and with
			// the material of the run interned, so that the loop below is what handles everything unusual.
			if (triStreak >= 64 && !matDirty && haveVerts && (size_t)(end - ls) > (1u << 20)) {
				const TriRun run = tri_run(ls, end, verts, stack.back(), S, curMat);
				if (run.lines) {
					p = ls + run.bytes;
					lineNo += run.lines - 1;
					continue;
				}
				triStreak = 0; // the next line is not a plain tri line: back to one line at a time
			}
			++triStreak;
			if (!read_vals(q, le, 3, v)) continue;
			if (!haveVerts) return mirt_set_error(MIRT_E_INVALID, "line %llu: tri before maxverts", (unsigned long long)lineNo);
			const int i0 = (int)v[0], i1 = (int)v[1], i2 = (int)v[2];
			if (i0 < 0 || i1 < 0 || i2 < 0 || (size_t)i0 >= verts.size() || (size_t)i1 >= verts.size() || (size_t)i2 >= verts.size())
				return mirt_set_error(MIRT_E_INVALID, "line %llu: vertex index outside maxverts", (unsigned long long)lineNo);
			if (matDirty) {
				mirt_material mm;
				memcpy(mm.diffuse, diffuse, 12); memcpy(mm.specular, specular, 12); memcpy(mm.emission, emission, 12); memcpy(mm.ambient, ambient, 12);
				mm.shininess = shininess;
				curMat = mats.intern(S, mm); // Material snapshot per object, :147
				matDirty = false;
			}
			const M4& T = stack.back();
			const V3 in[3] = { verts[i0], verts[i1], verts[i2] };
			for (int k = 0; k < 3; ++k) { // Triangle.cpp:23-25
				const V4 w = mul(T, v4(in[k].x, in[k].y, in[k].z, 1.0f));
				S.triVerts.push_back(w.x); S.triVerts.push_back(w.y); S.triVerts.push_back(w.z);
			}
			S.triPrim.push_back(S.nPrims++);
			S.primMaterial.push_back(curMat);
		}
		else if ((triStreak = 0, is("vertex"))) { // :75-81
			if (!read_vals(q, le, 3, v)) continue;
			if (!haveVerts || vertIndex >= verts.size()) return mirt_set_error(MIRT_E_INVALID, "line %llu: more vertices than maxverts", (unsigned long long)lineNo);
			verts[vertIndex++] = v3(v[0], v[1], v[2]);
		}
		else if (is("sphere") || is("trinormal")) { // :146-173
			const bool sph = is("sphere");
			if (!read_vals(q, le, sph ? 4 : 6, v)) continue;
			if (matDirty) {
				mirt_material mm;
				memcpy(mm.diffuse, diffuse, 12); memcpy(mm.specular, specular, 12); memcpy(mm.emission, emission, 12); memcpy(mm.ambient, ambient, 12);
				mm.shininess = shininess;
				curMat = mats.intern(S, mm);
				matDirty = false;
			}
			const M4& T = stack.back();
			if (sph) { // :150-157, Sphere.cpp:5
				mirt_sphere s;
				s.center[0] = v[0]; s.center[1] = v[1]; s.center[2] = v[2];
				s.radius = v[3];
				const M4 Ti = inverse(T); // Sphere.cpp:44,60
				memcpy(s.M, &T, sizeof s.M);
				memcpy(s.Minv, &Ti, sizeof s.Minv);
				S.spheres.push_back(s);
				S.sphPrim.push_back(S.nPrims++);
			}
			else { // :166-173: positions are entries 0, 2, 4 of the vertex-normal array (the normals do not reach the trace path)
				if (!haveNorms) return mirt_set_error(MIRT_E_INVALID, "line %llu: trinormal before maxvertnorms", (unsigned long long)lineNo);
				const int idx[3] = { (int)v[0], (int)v[2], (int)v[4] };
				for (int k = 0; k < 3; ++k)
					if (idx[k] < 0 || (size_t)idx[k] >= vertNorms.size()) return mirt_set_error(MIRT_E_INVALID, "line %llu: vertex index outside maxvertnorms", (unsigned long long)lineNo);
				for (int k = 0; k < 3; ++k) {
					const V3 in = vertNorms[idx[k]];
					const V4 w = mul(T, v4(in.x, in.y, in.z, 1.0f));
					S.triVerts.push_back(w.x); S.triVerts.push_back(w.y); S.triVerts.push_back(w.z);
				}
				S.triPrim.push_back(S.nPrims++);
			}
			S.primMaterial.push_back(curMat);
		}
		else if (is("directional") || is("point")) { // :39-52
			if (!read_vals(q, le, 6, v)) continue;
			mirt_light l;
			l.location[0] = v[0]; l.location[1] = v[1]; l.location[2] = v[2];
			l.location[3] = is("point") ? 1.0f : 0.0f;
			l.color[0] = v[3]; l.color[1] = v[4]; l.color[2] = v[5];
			l.pad = 0.0f;
			S.lights.push_back(l);
		}
		else if (is("attenuation")) { if (read_vals(q, le, 3, v)) memcpy(S.attenuation, v, 12); }                   // :53-58
		else if (is("maxverts")) { // :60-66: a fresh zeroed array; the running vertex index is not reset
			if (!read_vals(q, le, 1, v)) continue;
			const int n = (int)v[0];
			if (n < 0) return mirt_set_error(MIRT_E_INVALID, "line %llu: negative maxverts", (unsigned long long)lineNo);
			verts.assign((size_t)n, v3(0, 0, 0));
			haveVerts = true;
		}
		else if (is("maxvertnorms")) { // :68-74
			if (!read_vals(q, le, 1, v)) continue;
			const int n = (int)v[0] * 2;
			if (n < 0) return mirt_set_error(MIRT_E_INVALID, "line %llu: negative maxvertnorms", (unsigned long long)lineNo);
			vertNorms.assign((size_t)n, v3(0, 0, 0));
			haveNorms = true;
		}
		else if (is("vertexnormal")) { // :82-90
			if (!read_vals(q, le, 6, v)) continue;
			if (!haveNorms || normIndex + 2 > vertNorms.size()) return mirt_set_error(MIRT_E_INVALID, "line %llu: more vertexnormals than maxvertnorms", (unsigned long long)lineNo);
			vertNorms[normIndex++] = v3(v[0], v[1], v[2]);
			vertNorms[normIndex++] = v3(v[3], v[4], v[5]);
		}
		else if (is("ambient")) { if (read_vals(q, le, 3, v)) { memcpy(ambient, v, 12); matDirty = true; } }        // :91-96
		else if (is("diffuse")) { if (read_vals(q, le, 3, v)) { memcpy(diffuse, v, 12); matDirty = true; } }        // :97-102
		else if (is("specular")) { if (read_vals(q, le, 3, v)) { memcpy(specular, v, 12); matDirty = true; } }      // :103-108
		else if (is("emission")) { if (read_vals(q, le, 3, v)) { memcpy(emission, v, 12); matDirty = true; } }      // :109-114
		else if (is("shininess")) { if (read_vals(q, le, 1, v)) { shininess = v[0]; matDirty = true; } }            // :115-120
		else if (is("size")) { // :121-127
			if (read_vals(q, le, 2, v)) {
				const int w = (int)v[0], h = (int)v[1];
				if (w <= 0 || h <= 0) return mirt_set_error(MIRT_E_INVALID, "line %llu: size must be positive", (unsigned long long)lineNo);
				S.width = (uint32_t)w;
				S.height = (uint32_t)h;
				S.haveSize = true;
			}
		}
		else if (is("maxdepth")) { if (read_vals(q, le, 1, v)) S.maxDepth = (int)v[0]; }                              // :128-133
		else if (is("output")) { // :134-139
			while (q < le && is_space(*q)) ++q;
			const char* b = q;
			while (q < le && !is_space(*q)) ++q;
			if (q > b) S.outputName.assign(b, (size_t)(q - b));
		}
		else if (is("camera")) { // :140-145, Camera.cpp:4: `up` is stored normalised
			if (!read_vals(q, le, 10, v)) continue;
			memcpy(S.camera.look_from, v, 12);
			memcpy(S.camera.look_at, v + 3, 12);
			const V3 up = normalize(v3(v[6], v[7], v[8]));
			S.camera.up[0] = up.x; S.camera.up[1] = up.y; S.camera.up[2] = up.z;
			S.camera.fovy_deg = v[9];
			S.haveCamera = true;
		}
		else if (is("translate")) { if (read_vals(q, le, 3, v)) stack.back() = mul(stack.back(), translation(v[0], v[1], v[2])); }      // :175-180
		else if (is("scale")) { if (read_vals(q, le, 3, v)) stack.back() = mul(stack.back(), scaling(v[0], v[1], v[2])); }              // :181-186
		else if (is("rotate")) { if (read_vals(q, le, 4, v)) stack.back() = mul(stack.back(), rotation(v[0], v[1], v[2], v[3])); }      // :187-192
		else if (is("pushTransform")) { const M4 top = stack.back(); stack.push_back(top); }                                            // :193-195
		else if (is("popTransform")) { if (stack.size() > 1) stack.pop_back(); }                                                        // :196-203
		// anything else: "Unknown Command ... Skipping", :204-206
	}
	if (!S.haveCamera) return mirt_set_error(MIRT_E_INVALID, "scene has no camera line (the reference would render with an uninitialised camera)");
	S.parseSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	return MIRT_OK;
}

int finish(mirt_host_scene* S, int rc, mirt_host_scene** out) {
	if (rc != MIRT_OK) { delete S; return rc; }
	*out = S;
	return MIRT_OK;
}

} // namespace

extern "C" {

int mirt_host_scene_parse(const char* text, size_t len, uint32_t threads, mirt_host_scene** out) {
	if (!out) return mirt_set_error(MIRT_E_INVALID, "out is NULL");
	*out = nullptr;
	if (!text && len) return mirt_set_error(MIRT_E_INVALID, "text is NULL");
	mirt_host_scene* S = nullptr;
	try {
		S = new mirt_host_scene();
		set_defaults(*S);
		int rc = parse_text(text, len, *S);
		if (rc == MIRT_OK) rc = build_tree(*S, threads);
		return finish(S, rc, out);
	} catch (const std::bad_alloc&) {
		delete S;
		return mirt_set_error(MIRT_E_NOMEM, "out of host memory while loading the scene");
	}
}

int mirt_host_scene_load(const char* path, uint32_t threads, mirt_host_scene** out) {
	if (!out) return mirt_set_error(MIRT_E_INVALID, "out is NULL");
	*out = nullptr;
	if (!path) return mirt_set_error(MIRT_E_INVALID, "path is NULL");
	FILE* f = fopen(path, "rb");
	if (!f) return mirt_set_error(MIRT_E_INVALID, "Unable to open file %s", path); // Scene.cpp:212-215
	std::vector<char> text;
	try {
		char chunk[1 << 16];
		size_t got;
		while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) text.insert(text.end(), chunk, chunk + got);
	} catch (const std::bad_alloc&) {
		fclose(f);
		return mirt_set_error(MIRT_E_NOMEM, "out of host memory while reading %s", path);
	}
	fclose(f);
	return mirt_host_scene_parse(text.data(), text.size(), threads, out);
}

int mirt_host_scene_from_prims(const mirt_scene_desc* d, uint32_t width, uint32_t height, uint32_t threads, mirt_host_scene** out) {
	if (!out) return mirt_set_error(MIRT_E_INVALID, "out is NULL");
	*out = nullptr;
	if (!d || d->struct_size != sizeof(mirt_scene_desc)) return mirt_set_error(MIRT_E_INVALID, "desc is NULL or struct_size does not match this library");
	if ((d->n_triangles && (!d->tri_verts || !d->tri_prim)) || (d->n_spheres && (!d->spheres || !d->sphere_prim)) ||
	    (d->n_prims && !d->prim_material) || (d->n_materials && !d->materials) || (d->n_lights && !d->lights))
		return mirt_set_error(MIRT_E_INVALID, "desc has a non-zero count with a NULL array");
	for (uint32_t i = 0; i < d->n_triangles; ++i)
		if (d->tri_prim[i] >= d->n_prims) return mirt_set_error(MIRT_E_INVALID, "tri_prim[%u] outside n_prims", i);
	for (uint32_t i = 0; i < d->n_spheres; ++i)
		if (d->sphere_prim[i] >= d->n_prims) return mirt_set_error(MIRT_E_INVALID, "sphere_prim[%u] outside n_prims", i);
	for (uint32_t i = 0; i < d->n_prims; ++i)
		if (d->prim_material[i] >= d->n_materials) return mirt_set_error(MIRT_E_INVALID, "prim_material[%u] outside n_materials", i);
	mirt_host_scene* S = nullptr;
	try {
		S = new mirt_host_scene();
		set_defaults(*S);
		S->nPrims = d->n_prims;
		S->triVerts.assign(d->tri_verts, d->tri_verts + (size_t)d->n_triangles * 9);
		S->triPrim.assign(d->tri_prim, d->tri_prim + d->n_triangles);
		S->spheres.assign(d->spheres, d->spheres + d->n_spheres);
		S->sphPrim.assign(d->sphere_prim, d->sphere_prim + d->n_spheres);
		S->materials.assign(d->materials, d->materials + d->n_materials);
		S->primMaterial.assign(d->prim_material, d->prim_material + d->n_prims);
		S->lights.assign(d->lights, d->lights + d->n_lights);
		S->camera = d->camera;
		memcpy(S->attenuation, d->attenuation, 12);
		S->maxDepth = d->max_depth;
		S->width = width;
		S->height = height;
		S->haveSize = width && height;
		S->haveCamera = true;
		return finish(S, build_tree(*S, threads), out);
	} catch (const std::bad_alloc&) {
		delete S;
		return mirt_set_error(MIRT_E_NOMEM, "out of host memory while copying the primitives");
	}
}

int mirt_host_scene_desc(const mirt_host_scene* S, mirt_scene_desc* d, uint32_t* width, uint32_t* height) {
	if (!S || !d) return mirt_set_error(MIRT_E_INVALID, "scene or desc_out is NULL");
	memset(d, 0, sizeof *d);
	d->struct_size = sizeof *d;
	d->n_nodes = (uint32_t)S->nodes.size();
	d->nodes = S->nodes.data();
	d->n_leaf_prims = (uint32_t)S->leafPrims.size();
	d->leaf_prims = S->leafPrims.data();
	d->n_prims = S->nPrims;
	d->n_triangles = (uint32_t)S->triPrim.size();
	d->tri_verts = S->triVerts.data();
	d->tri_prim = S->triPrim.data();
	d->n_spheres = (uint32_t)S->sphPrim.size();
	d->spheres = S->spheres.data();
	d->sphere_prim = S->sphPrim.data();
	d->n_materials = (uint32_t)S->materials.size();
	d->materials = S->materials.data();
	d->prim_material = S->primMaterial.data();
	d->n_lights = (uint32_t)S->lights.size();
	d->lights = S->lights.data();
	d->max_depth = S->maxDepth;
	d->camera = S->camera;
	memcpy(d->attenuation, S->attenuation, 12);
	if (width) *width = S->width;
	if (height) *height = S->height;
	return MIRT_OK;
}

const char* mirt_host_scene_output_name(const mirt_host_scene* S) { return S ? S->outputName.c_str() : ""; }

int mirt_host_scene_truncated_nodes(const mirt_host_scene* S, uint64_t* count) {
	if (!S || !count) return mirt_set_error(MIRT_E_INVALID, "scene or count is NULL");
	*count = S->truncatedNodes;
	return MIRT_OK;
}

int mirt_host_scene_timing(const mirt_host_scene* S, double* parse_s, double* build_s) {
	if (!S) return mirt_set_error(MIRT_E_INVALID, "scene is NULL");
	if (parse_s) *parse_s = S->parseSeconds;
	if (build_s) *build_s = S->buildSeconds;
	return MIRT_OK;
}

void mirt_host_scene_destroy(mirt_host_scene* S) { delete S; }

} // extern "C"
