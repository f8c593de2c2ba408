/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement, in plain C, of the reference's per-pixel trace path, evaluated over the flattened
 * scene arrays of include/mirt.h.  It is the checker the HIP path is compared with; the product never
 * loads it.  Every function cites the reference lines it follows (paths relative to the reference root).
 *
 * Pinning: tests/test_oracle_vs_reference.py compares every function here bit for bit with the
 * reference's own compiled classes (oracle/_ref, built by oracle/Makefile from the reference sources)
 * on all fixture scenes, and tests/golden/ holds outputs generated from that build.  The shading
 * driver (main.cpp:176-268) cannot be compiled here (main.cpp needs windows.h); for that part the
 * comparison is against a restatement that calls the reference's compiled Color / Camera / Scene code.
 *
 * Arithmetic rules that make the results bit-identical to the reference built with g++ on x86-64:
 * binary32 everywhere, no FMA contraction (-ffp-contract=off), GLM 0.9.8.5 scalar formulas:
 *   dot3  = (x*x' + y*y') + z*z'                       glm/detail/func_geometric.inl:54-61
 *   dot4  = (x*x' + y*y') + (z*z' + w*w')              glm/detail/func_geometric.inl:64-72
 *   cross = (ay*bz - by*az, az*bx - bz*ax, ax*by - bx*ay)   func_geometric.inl:75-87
 *   normalize(v) = v * (1/sqrt(dot(v,v)))              func_geometric.inl:90-99, func_exponential.inl:128-133
 *   mat4*vec4 row r = (m0r*v0 + m1r*v1) + (m2r*v2 + m3r*v3)   glm/detail/type_mat4x4.inl:526-537
 */
#include "mirt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3s(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { float x = a.x * b.x, y = a.y * b.y, z = a.z * b.z; return x + y + z; }
static inline float dot4(v4 a, v4 b) { float x = a.x * b.x, y = a.y * b.y, z = a.z * b.z, w = a.w * b.w; return (x + y) + (z + w); }
static inline v3 cross3(v3 a, v3 b) { return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
static inline float invsqrt(float x) { return 1.0f / sqrtf(x); }
static inline v3 normalize3(v3 a) { return mul3s(a, invsqrt(dot3(a, a))); }
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
/* m is column-major: m[c*4+r] */
static inline v4 mat4_mul(const float* m, v4 v) {
	v4 r;
	r.x = (m[0] * v.x + m[4] * v.y) + (m[8] * v.z + m[12] * v.w);
	r.y = (m[1] * v.x + m[5] * v.y) + (m[9] * v.z + m[13] * v.w);
	r.z = (m[2] * v.x + m[6] * v.y) + (m[10] * v.z + m[14] * v.w);
	r.w = (m[3] * v.x + m[7] * v.y) + (m[11] * v.z + m[15] * v.w);
	return r;
}
/* transpose(m) * v without forming the transpose: row r uses column r of m */
static inline v4 mat4_tmul(const float* m, v4 v) {
	v4 r;
	r.x = (m[0] * v.x + m[1] * v.y) + (m[2] * v.z + m[3] * v.w);
	r.y = (m[4] * v.x + m[5] * v.y) + (m[6] * v.z + m[7] * v.w);
	r.z = (m[8] * v.x + m[9] * v.y) + (m[10] * v.z + m[11] * v.w);
	r.w = (m[12] * v.x + m[13] * v.y) + (m[14] * v.z + m[15] * v.w);
	return r;
}
static inline float minf_std(float a, float b) { return (b < a) ? b : a; } /* std::min */
static inline float maxf_std(float a, float b) { return (a < b) ? b : a; } /* std::max */

typedef struct { v3 o, d; } ray_t;
typedef struct { float dist; v3 n; int32_t prim; } hit_t;

static inline hit_t no_hit(void) { hit_t h; h.dist = INFINITY; h.n = V3(0, 0, 0); h.prim = -1; return h; }
static inline int hit_valid(const hit_t* h) { return !isinf(h->dist); } /* SceneObjects.hpp:35-37 */

/* ---- AABB::intersect, hit/miss part (AABB.cpp:53-72) -------------------------------------------------- */
static int box_hit(const float* bmin, const float* bmax, const ray_t* r) {
	float dx = (fabsf(r->d.x - 0.0f) < 0.00001f) ? 0.00001f : r->d.x; /* glm::epsilonEqual, gtc/epsilon.inl:15-23 */
	float dy = (fabsf(r->d.y - 0.0f) < 0.00001f) ? 0.00001f : r->d.y;
	float dz = (fabsf(r->d.z - 0.0f) < 0.00001f) ? 0.00001f : r->d.z;
	float t1 = (bmin[0] - r->o.x) / dx;
	float t2 = (bmax[0] - r->o.x) / dx;
	float t3 = (bmin[1] - r->o.y) / dy;
	float t4 = (bmax[1] - r->o.y) / dy;
	float t5 = (bmin[2] - r->o.z) / dz;
	float t6 = (bmax[2] - r->o.z) / dz;
	float tmin = maxf_std(maxf_std(minf_std(t1, t2), minf_std(t3, t4)), minf_std(t5, t6)); /* :60 */
	float tmax = minf_std(minf_std(maxf_std(t1, t2), maxf_std(t3, t4)), maxf_std(t5, t6)); /* :61 */
	if ((double)tmax < 0.0001) return 0; /* :65 -- the literal there is a double */
	if (tmin > tmax) return 0;            /* :70 */
	float t_result = tmin;                /* :73 */
	if (tmin < 0.0001f) t_result = tmax;  /* :76-78 */
	return !isinf(t_result);              /* what Partition.cpp:36 asks of the returned Intersection(t_result, ...): SceneObjects.hpp:35-37
	                                         (an empty scene's root box is +inf .. -inf: every t is infinite) */
}

/* ---- Triangle::intersect (Triangle.cpp:147-175) --------------------------------------------------------- */
static hit_t tri_hit(const float* v, const ray_t* r) {
	v3 v1 = V3(v[0], v[1], v[2]), v2 = V3(v[3], v[4], v[5]), vv3 = V3(v[6], v[7], v[8]);
	v3 e1 = sub3(vv3, v2), e2 = sub3(v1, vv3), e3 = sub3(v2, v1);           /* :148-150 */
	v3 N = normalize3(cross3(sub3(v2, v1), sub3(vv3, v1)));                 /* :151, :173-175 */
	float dn = dot3(r->d, N);
	if (!(fabsf(dn - 0.0f) < 0.0001f)) {                                    /* :152 */
		float d = dot3(v1, N);                                              /* :153 */
		float t = (d - dot3(r->o, N)) / dot3(r->d, N);                      /* :154 */
		if (t < 0.0001f) return no_hit();                                   /* :155-157 */
		v3 P = add3(r->o, mul3s(r->d, t));                                  /* :158 */
		v3 d1 = sub3(P, v1), d2 = sub3(P, v2), d3 = sub3(P, vv3);           /* :159-161 */
		float area = dot3(cross3(e1, e2), N) / 2.0f;                        /* :162 */
		int b0 = (dot3(cross3(e1, d3), N) / 2.0f / area) >= 0.0f;           /* :163 */
		int b1 = (dot3(cross3(e2, d1), N) / 2.0f / area) >= 0.0f;           /* :164 */
		int b2 = (dot3(cross3(e3, d2), N) / 2.0f / area) >= 0.0f;           /* :165 */
		if (b0 && b1 && b2) {
			hit_t h;
			h.dist = length3(sub3(r->o, P));  /* glm::distance(P, origin) = length(origin - P), :167 */
			h.n = normalize3(N);              /* Intersection ctor normalises again, SceneObjects.hpp:39 */
			h.prim = -1;
			return h;
		}
	}
	return no_hit();
}

/* ---- Sphere::intersect (Sphere.cpp:59-87) --------------------------------------------------------------- */
static hit_t sphere_hit(const mirt_sphere* s, const ray_t* r) {
	v4 o4 = {r->o.x, r->o.y, r->o.z, 1.0f}, d4 = {r->d.x, r->d.y, r->d.z, 0.0f};
	v4 oo = mat4_mul(s->Minv, o4);                                  /* :60 */
	v4 dd = mat4_mul(s->Minv, d4);
	float inv = invsqrt(dot4(dd, dd));                              /* normalize() of the vec4 */
	v3 o = V3(oo.x, oo.y, oo.z);
	v3 d = V3(dd.x * inv, dd.y * inv, dd.z * inv);                  /* truncated to vec3 by the Ray ctor */
	v3 c = V3(s->center[0], s->center[1], s->center[2]);
	v3 nd = normalize3(d);                                          /* :61 */
	float a = dot3(nd, nd);                                         /* :62 */
	v3 oc = sub3(o, c);
	float b = 2.0f * dot3(nd, oc);                                  /* :63 */
	float cc = dot3(oc, oc) - powf(s->radius, 2.0f);                /* :64 glm::pow(float,float) = powf */
	/* :85-87  glm::pow(b, 2) has an int exponent -> std::pow promotes to double; the subtraction is done in
	 * double and the result rounded to float by the return type. */
	float disc = (float)(pow((double)b, 2.0) - (double)(4 * a * cc));
	if (disc < 0.0f) return no_hit();                               /* :66-68 */
	float x1 = (-b + sqrtf(disc)) / 2.0f * a;                       /* :69 */
	float x2 = (-b - sqrtf(disc)) / 2.0f * a;                       /* :70 */
	float t = minf_std(x1, x2);                                     /* :71 */
	if (t < 0.001f) {                                               /* :72-77 */
		t = maxf_std(x1, x2);
		if (t < 0.001f) return no_hit();
	}
	v3 p = add3(o, mul3s(d, t));                                    /* :78 */
	v3 pc = sub3(p, c);
	v4 n4 = {2.0f * pc.x, 2.0f * pc.y, 2.0f * pc.z, 0.0f};
	v4 nn = mat4_tmul(s->Minv, n4);                                 /* :79 */
	v4 p4 = {p.x, p.y, p.z, 1.0f};
	v4 pw = mat4_mul(s->M, p4);                                     /* :81 */
	hit_t h;
	h.dist = length3(sub3(r->o, V3(pw.x, pw.y, pw.z)));             /* :82 */
	h.n = normalize3(V3(nn.x, nn.y, nn.z));                         /* SceneObjects.hpp:39 */
	h.prim = -1;
	return h;
}

/* ---- flattened-scene lookup tables ------------------------------------------------------------------------ */
typedef struct {
	const mirt_scene_desc* d;
	int32_t* prim_slot; /* prim id -> index in tri (>=0) or ~index in spheres (<0) */
	uint64_t* visits;   /* box tests, leaf visits, prim tests (may be NULL) */
} octx;

static int octx_init(octx* c, const mirt_scene_desc* d) {
	c->d = d;
	c->visits = NULL;
	c->prim_slot = (int32_t*)malloc(sizeof(int32_t) * (d->n_prims ? d->n_prims : 1));
	if (!c->prim_slot) return MIRT_E_NOMEM;
	for (uint32_t i = 0; i < d->n_prims; ++i) c->prim_slot[i] = INT32_MIN;
	for (uint32_t i = 0; i < d->n_triangles; ++i) {
		if (d->tri_prim[i] >= d->n_prims) return MIRT_E_INVALID;
		c->prim_slot[d->tri_prim[i]] = (int32_t)i;
	}
	for (uint32_t i = 0; i < d->n_spheres; ++i) {
		if (d->sphere_prim[i] >= d->n_prims) return MIRT_E_INVALID;
		c->prim_slot[d->sphere_prim[i]] = ~(int32_t)i;
	}
	return MIRT_OK;
}
static void octx_free(octx* c) { free(c->prim_slot); }

static hit_t prim_hit(const octx* c, uint32_t prim, const ray_t* r) {
	int32_t s = c->prim_slot[prim];
	if (s == INT32_MIN) return no_hit(); /* Shape::intersect base class, Shape.cpp:44-46 */
	if (s >= 0) return tri_hit(c->d->tri_verts + 9 * (size_t)s, r);
	return sphere_hit(&c->d->spheres[~s], r);
}

/* ---- Partition::intersect (Partition.cpp:35-53) + PartitionNode::findClosestIntersection (Partition.h:28-39) -- */
static hit_t traverse(const octx* c, uint32_t node, const ray_t* r) {
	if (node == MIRT_NO_CHILD) return no_hit();                              /* :36 nullptr */
	const mirt_node* n = &c->d->nodes[node];
	if (c->visits) c->visits[0]++;
	if (!box_hit(n->bmin, n->bmax, r)) return no_hit();                      /* :36 */
	if (n->b & MIRT_LEAF_FLAG) {                                             /* :39-41 */
		if (c->visits) c->visits[1]++;
		hit_t best = no_hit();
		uint32_t cnt = n->b & ~MIRT_LEAF_FLAG;
		for (uint32_t k = 0; k < cnt; ++k) {                                 /* Partition.h:31 */
			uint32_t prim = c->d->leaf_prims[n->a + k];
			if (c->visits) c->visits[2]++;
			hit_t cur = prim_hit(c, prim, r);
			if (hit_valid(&cur) && (cur.dist < best.dist || !hit_valid(&best))) { /* Partition.h:33 */
				best = cur;
				best.prim = (int32_t)prim;
			}
		}
		return best;
	}
	hit_t l = traverse(c, n->a, r);                                          /* :42 */
	hit_t rr = traverse(c, n->b, r);                                         /* :43 */
	int lv = hit_valid(&l), rv = hit_valid(&rr);
	if (!(lv || rv)) return no_hit();                                        /* :44-46 */
	if ((lv && !rv) || l.dist <= rr.dist) return l;                          /* :47-49 */
	return rr;                                                               /* :50-52 */
}

static hit_t closest(const octx* c, const ray_t* r) { /* Scene.cpp:299-301, Partition.cpp:31-33 */
	if (c->d->n_nodes == 0) return no_hit();
	return traverse(c, 0, r);
}

/* ---- Camera (Camera.cpp:4-28) --------------------------------------------------------------------------------- */
typedef struct { v3 eye, u, v, w; float tanHalf; } cam_t;

static cam_t cam_init(const mirt_camera* m) {
	cam_t c;
	v3 from = V3(m->look_from[0], m->look_from[1], m->look_from[2]);
	v3 at = V3(m->look_at[0], m->look_at[1], m->look_at[2]);
	v3 up = V3(m->up[0], m->up[1], m->up[2]);
	v3 fwd = sub3(at, from);                          /* :6 */
	c.eye = from;
	c.w = neg3(normalize3(fwd));                      /* :16 */
	c.u = normalize3(cross3(up, c.w));                /* :17 */
	c.v = normalize3(cross3(c.w, c.u));               /* :18 */
	float rad = (m->fovy_deg / 2.0f) * 0.01745329251994329576923690768489f; /* glm::radians, func_trigonometric.inl:12-17 */
	c.tanHalf = tanf(rad);
	return c;
}

static ray_t cam_ray(const cam_t* c, float i, float j, int width, int height) { /* Camera.cpp:15-24 */
	float alpha = c->tanHalf * ((float)width / height) * ((i - (width / 2.0f)) / (width / 2.0f));   /* :19 */
	float beta = c->tanHalf * (((height / 2.0f) - j) / (height / 2.0f));                              /* :20 */
	ray_t r;
	r.o = c->eye;
	r.d = sub3(add3(mul3s(c->u, alpha), mul3s(c->v, beta)), c->w);                                    /* :21 */
	return r;
}

static v3 point_on_ray(const ray_t* r, float t) { return add3(r->o, mul3s(normalize3(r->d), t)); } /* :26-28 */

/* ---- shading (main.cpp:176-268) ------------------------------------------------------------------------------- */
typedef struct { float r, g, b; } col;
static inline col C3(float r, float g, float b) { col c = {r, g, b}; return c; }
static inline col cadd(col a, col b) { return C3(a.r + b.r, a.g + b.g, a.b + b.b); }   /* Color.cpp:25-35 */
static inline col cmuls(col a, float x) { return C3(a.r * x, a.g * x, a.b * x); }      /* Color.cpp:41-43 */
static inline col cmul(col a, col b) { return C3(a.r * b.r, a.g * b.g, a.b * b.b); }   /* Color.cpp:45-47 */
static inline col cfrom(const float* p) { return C3(p[0], p[1], p[2]); }

typedef struct {
	octx cx;
	cam_t cam;
	uint32_t features, debug;
	uint64_t primary, shadow, reflection, primary_hits;
	uint64_t vis[3][3]; /* [ray kind][box, leaf, prim] */
	int count_visits;
} rctx;

static hit_t closest_kind(rctx* rc, const ray_t* r, int kind) {
	if (rc->count_visits) rc->cx.visits = rc->vis[kind];
	return closest(&rc->cx, r);
}

static col shade_lights(rctx* rc, v3 P, v3 N, const mirt_material* mat, v3 view) { /* main.cpp:203-256 */
	const mirt_scene_desc* d = rc->cx.d;
	col sum = cadd(cfrom(mat->ambient), cfrom(mat->emission));                    /* :204 */
	for (uint32_t li = 0; li < d->n_lights; ++li) {                               /* :208 */
		const mirt_light* L = &d->lights[li];
		v3 ld;
		float atten;
		if (L->location[3] > 0.0f) {                                              /* :213, SceneObjects.hpp:27 */
			ld = sub3(V3(L->location[0], L->location[1], L->location[2]), P);     /* :214 */
			float dist = length3(ld);                                             /* :215 */
			atten = 1.0f / (d->attenuation[0] + d->attenuation[1] * dist + d->attenuation[2] * powf(dist, 2.0f)); /* :216, :266-268 */
		}
		else {
			ld = V3(L->location[0], L->location[1], L->location[2]);              /* :219 */
			atten = 1.0f;
		}
		ray_t sr;
		sr.o = P;
		sr.d = normalize3(ld);                                                    /* :222 */
		rc->shadow++;
		hit_t occ = closest_kind(rc, &sr, 1);                                     /* :223 */
		if (!hit_valid(&occ) || occ.dist >= length3(ld) || !(rc->features & MIRT_F_SHADOWS)) { /* :224 */
			float kd = maxf_std(dot3(normalize3(N), normalize3(ld)), 0.0f);       /* :225, :258-260 */
			v3 eye = sub3(view, P);                                               /* :226 */
			v3 H = normalize3(add3(normalize3(ld), normalize3(eye)));             /* :227 */
			float ks = powf(maxf_std(dot3(H, N), 0.0f), mat->shininess);          /* :228, :262-264 */
			if (rc->debug == MIRT_DEBUG_DIFFUSE_INTENSITY) sum = cadd(sum, C3(kd, kd, kd));        /* :229-231 */
			else if (rc->debug == MIRT_DEBUG_SPECULAR_INTENSITY) sum = cadd(sum, C3(ks, ks, ks));  /* :232-234 */
			else if (rc->debug == MIRT_DEBUG_NORMAL_MAP) sum = cadd(sum, C3(N.x, N.y, N.z));       /* :235-237 */
			else if (rc->debug == MIRT_DEBUG_LIGHT_DIRECTION_MAP) sum = cadd(sum, C3(H.x, H.y, H.z)); /* :238-240 */
			else {
				if (rc->features & MIRT_F_DIFFUSE)                                /* :242-244 */
					sum = cadd(sum, cmul(cmuls(cmuls(cfrom(mat->diffuse), atten), kd), cfrom(L->color)));
				if (rc->features & MIRT_F_SPECULAR)                               /* :245-247 */
					sum = cadd(sum, cmul(cmuls(cmuls(cfrom(mat->specular), atten), ks), cfrom(L->color)));
			}
		}
		else if (rc->debug == MIRT_DEBUG_SHADOW_MAP) {                            /* :250-252 */
			const mirt_material* om = &d->materials[d->prim_material[occ.prim]];
			sum = cadd(sum, cfrom(om->diffuse));
		}
	}
	return sum;
}

static col shade_ray(rctx* rc, const ray_t* r, int depth) { /* main.cpp:176-201 */
	const mirt_scene_desc* d = rc->cx.d;
	if (depth <= d->max_depth) {                                                  /* :177 */
		hit_t h = closest_kind(rc, r, depth == 0 ? 0 : 2);                        /* :178 */
		if (!hit_valid(&h)) return C3(0, 0, 0);                                   /* :179-181, background forced black :124 */
		if (depth == 0) rc->primary_hits++;
		if (rc->debug == MIRT_DEBUG_PRIMARY_MAP) return C3(1.0f, 0.0f, 0.0f);     /* :183-185 */
		const mirt_material* mat = &d->materials[d->prim_material[h.prim]];
		v3 P = point_on_ray(r, h.dist);
		col lit = shade_lights(rc, P, h.n, mat, r->o);                            /* :187 */
		ray_t rf;
		rf.o = P;
		rf.d = normalize3(sub3(r->d, mul3s(h.n, 2.0f * dot3(r->d, h.n))));        /* :188 */
		if (rc->features & MIRT_F_REFLECTIONS) {                                  /* :189-191 */
			++depth;
			if (depth <= d->max_depth) rc->reflection++;
			return cadd(lit, cmul(cfrom(mat->specular), shade_ray(rc, &rf, depth)));
		}
		return lit;                                                               /* :193 */
	}
	return C3(0, 0, 0);                                                           /* :199 */
}

static inline uint8_t quantise(float c) { /* Color.cpp:13-23 */
	float s = c * 255;
	int v;
	/* (int) of an out-of-range or NaN float is what cvttss2si returns on x86-64: INT_MIN */
	if (!(s >= -2147483648.0f && s < 2147483648.0f)) v = INT32_MIN;
	else v = (int)s;
	if (v < 0) v = 0;
	if (v > 255) v = 255;
	return (uint8_t)v;
}

static int tile_selected(const mirt_render_params* p, uint32_t i, uint32_t j) {
	if (p->tile_stride <= 1) return 1;
	uint32_t tw = p->tile_w ? p->tile_w : 32, th = p->tile_h ? p->tile_h : 32;
	uint32_t tiles_x = (p->width + tw - 1) / tw;
	uint32_t k = (i / th) * tiles_x + (j / tw);
	return (k % p->tile_stride) == p->first_tile;
}

static int render_impl(const mirt_scene_desc* d, const mirt_render_params* p, uint32_t row0, uint32_t row1, uint32_t row_step,
                       uint32_t col_step, uint8_t* bgr, mirt_stats* stats, uint64_t* visits9) {
	if (!d || !p || !bgr) return MIRT_E_INVALID;
	rctx rc;
	memset(&rc, 0, sizeof rc);
	int e = octx_init(&rc.cx, d);
	if (e) { octx_free(&rc.cx); return e; }
	rc.cam = cam_init(&d->camera);
	rc.features = p->feature_bits;
	rc.debug = p->debug_mode;
	rc.count_visits = visits9 != NULL;
	if (col_step < 1) col_step = 1;
	if (row_step < 1) row_step = 1;
	for (uint32_t i = row0; i < row1; i += row_step) {                            /* main.cpp:137 */
		for (uint32_t j = 0; j < p->width; j += col_step) {                       /* :138 */
			if (!tile_selected(p, i, j)) continue;
			ray_t r = cam_ray(&rc.cam, j + 0.5f, i + 0.5f, (int)p->width, (int)p->height); /* :150-152 */
			rc.primary++;
			col c = shade_ray(&rc, &r, 0);                                        /* :153 */
			size_t k = ((size_t)i * p->width + j) * 3;
			bgr[k] = quantise(c.b);                                               /* :154 */
			bgr[k + 1] = quantise(c.g);                                           /* :155 */
			bgr[k + 2] = quantise(c.r);                                           /* :156 */
		}
	}
	if (stats) {
		memset(stats, 0, sizeof *stats);
		stats->primary_rays = rc.primary;
		stats->shadow_rays = stats->shadow_rays_traced = rc.shadow;
		stats->reflection_rays = stats->reflection_rays_traced = rc.reflection;
		stats->primary_hits = rc.primary_hits;
	}
	if (visits9) memcpy(visits9, rc.vis, sizeof rc.vis);
	octx_free(&rc.cx);
	return MIRT_OK;
}

int mirt_oracle_render(const mirt_scene_desc* d, const mirt_render_params* p, uint8_t* bgr, mirt_stats* stats, uint64_t* visits9) {
	if (!p) return MIRT_E_INVALID;
	return render_impl(d, p, 0, p->height, 1, 1, bgr, stats, visits9);
}

int mirt_oracle_render_rows(const mirt_scene_desc* d, const mirt_render_params* p, uint32_t row0, uint32_t row1, uint32_t row_step,
                            uint32_t col_step, uint8_t* bgr, mirt_stats* stats, uint64_t* visits9) {
	if (!p || row1 > p->height || row0 > row1) return MIRT_E_INVALID;
	return render_impl(d, p, row0, row1, row_step, col_step, bgr, stats, visits9);
}

int mirt_oracle_trace_rays(const mirt_scene_desc* d, uint64_t n, const float* origins, const float* dirs,
                           int32_t* prim_id_out, float* dist_out, float* normal_out, uint64_t* visits) {
	octx c;
	int e = octx_init(&c, d);
	if (e) { octx_free(&c); return e; }
	c.visits = visits;
	for (uint64_t k = 0; k < n; ++k) {
		ray_t r;
		r.o = V3(origins[3 * k], origins[3 * k + 1], origins[3 * k + 2]);
		r.d = V3(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]);
		hit_t h = closest(&c, &r);
		if (prim_id_out) prim_id_out[k] = hit_valid(&h) ? h.prim : -1;
		if (dist_out) dist_out[k] = h.dist;
		if (normal_out) { normal_out[3 * k] = h.n.x; normal_out[3 * k + 1] = h.n.y; normal_out[3 * k + 2] = h.n.z; }
	}
	octx_free(&c);
	return MIRT_OK;
}

int mirt_oracle_primary_rays(const mirt_scene_desc* d, uint32_t w, uint32_t h, float* origins, float* dirs) {
	cam_t c = cam_init(&d->camera);
	for (uint32_t i = 0; i < h; ++i) for (uint32_t j = 0; j < w; ++j) {
		ray_t r = cam_ray(&c, j + 0.5f, i + 0.5f, (int)w, (int)h);
		size_t k = ((size_t)i * w + j) * 3;
		origins[k] = r.o.x; origins[k + 1] = r.o.y; origins[k + 2] = r.o.z;
		dirs[k] = r.d.x; dirs[k + 1] = r.d.y; dirs[k + 2] = r.d.z;
	}
	return MIRT_OK;
}

static void store_hit(const hit_t* h, uint64_t k, float* dist, float* normal) {
	dist[k] = h->dist;
	if (normal) { normal[3 * k] = h->n.x; normal[3 * k + 1] = h->n.y; normal[3 * k + 2] = h->n.z; }
}

int mirt_oracle_triangle(const float* v9, uint64_t n, const float* origins, const float* dirs, float* dist, float* normal) {
	for (uint64_t k = 0; k < n; ++k) {
		ray_t r;
		r.o = V3(origins[3 * k], origins[3 * k + 1], origins[3 * k + 2]);
		r.d = V3(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]);
		hit_t h = tri_hit(v9, &r);
		store_hit(&h, k, dist, normal);
	}
	return MIRT_OK;
}

int mirt_oracle_sphere(const mirt_sphere* s, uint64_t n, const float* origins, const float* dirs, float* dist, float* normal) {
	for (uint64_t k = 0; k < n; ++k) {
		ray_t r;
		r.o = V3(origins[3 * k], origins[3 * k + 1], origins[3 * k + 2]);
		r.d = V3(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]);
		hit_t h = sphere_hit(s, &r);
		store_hit(&h, k, dist, normal);
	}
	return MIRT_OK;
}

int mirt_oracle_box(const float* bmin, const float* bmax, uint64_t n, const float* origins, const float* dirs, uint8_t* hit) {
	for (uint64_t k = 0; k < n; ++k) {
		ray_t r;
		r.o = V3(origins[3 * k], origins[3 * k + 1], origins[3 * k + 2]);
		r.d = V3(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]);
		hit[k] = (uint8_t)box_hit(bmin, bmax, &r);
	}
	return MIRT_OK;
}
