// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
//
// C-ABI harness around the *unmodified* reference classes (Scene, Partition, AABB, Triangle,
// Sphere, Camera, Color), compiled from the sources where they lie under /root/reference by
// oracle/Makefile into oracle/_ref/libref_oracle.so.  The harness adds no arithmetic of its own to
// the intersection / traversal / camera / quantisation code: it calls the reference's functions.
//
// The shading driver (computePixelColor / calculateLightingColor, RayTracer/main.cpp:176-268) lives in
// main.cpp, which as a file is unbuildable here (it includes <windows.h> and "Shlwapi.h").  The driver itself
// needs neither, so the build recipe (oracle/Makefile) cuts exactly those lines out of main.cpp, unchanged, into
// two include files that are compiled below inside namespace reftext: every pixel ref_render_rows returns is
// computed by the reference's own text.  The reference does not count rays; the counts come from a restated copy
// of the same ~90 lines (shadeRay / shadeLights, each step citing the main.cpp line it follows), which is run
// beside the reference's text on request and must then produce the same bytes.
//
// Private members (Partition::root, Scene::objectTree, Sphere::center, ...) are reached by
// redefining the access keywords in this translation unit only; class layout is unchanged.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>
#include <stack>
#include <string>
#include <unordered_map>
#include <vector>
#include <inttypes.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <wchar.h>

#define private public
#define protected public
#include "Scene.h"
#undef private
#undef protected

// ---- the reference's own text: main.cpp:24-42 | 57-59, 65-66, 69-73, 176-268, 287-293 -----------------------------
namespace reftext {
#include "main_enums.inc"
int featureFlags = 15;            // main.cpp:91, set per call by ref_render_rows
Debug debugFlag = Debug::NONE;    // main.cpp:92
#include "main_shading.inc"
}  // namespace reftext

namespace {

typedef Partition::PartitionNode PNode;

struct RefScene {
	Scene* scene;
	std::unordered_map<const Shape*, int> shapeId;
	RefScene() : scene(nullptr) {}
};

enum FeatureBits { F_DIFFUSE = 1, F_SPECULAR = 2, F_SHADOWS = 4, F_REFLECTIONS = 8 };  // main.cpp:35-42
enum DebugMode { D_DIFFUSE_INTENSITY = 0, D_SPECULAR_INTENSITY, D_NORMAL_MAP, D_SHADOW_MAP, D_PRIMARY_MAP, D_LIGHT_DIR_MAP, D_NONE };  // main.cpp:24-33

struct ShadeCtx {
	const Scene* scene;
	int features;
	int debug;
	uint64_t primary, shadow, reflection;
};

struct Counted {
	uint64_t boxTests, leafVisits, primTests;
};

// Mirror of Partition::intersect (Partition.cpp:35-53) + PartitionNode::findClosestIntersection
// (Partition.h:28-39) that also reports WHICH shape won and counts visits.  The distances it returns
// are produced by the reference's own AABB::intersect / Shape::intersect calls.
struct IdHit {
	float dist;
	const Shape* shape;
	glm::vec3 normal;
	IdHit() : dist(std::numeric_limits<float>::infinity()), shape(nullptr) {}
	bool valid() const { return !glm::isinf(dist); }
};

IdHit mirrorTraverse(const Ray& ray, const PNode* node, Counted* cnt) {
	if (node == nullptr) return IdHit();
	if (cnt) cnt->boxTests++;
	if (!node->box.intersect(ray).isValidIntersection()) return IdHit();
	if (node->isLeaf()) {
		if (cnt) cnt->leafVisits++;
		IdHit best;
		for (Shape* obj : node->objects) {
			if (cnt) cnt->primTests++;
			Intersection cur = obj->intersect(ray);
			if (cur.isValidIntersection() && (cur.distAlongRay < best.dist || !best.valid())) {
				best.dist = cur.distAlongRay;
				best.shape = obj;
				best.normal = cur.intersectNormal;
			}
		}
		return best;
	}
	IdHit l = mirrorTraverse(ray, node->left, cnt);
	IdHit r = mirrorTraverse(ray, node->right, cnt);
	if (!(l.valid() || r.valid())) return IdHit();
	if ((l.valid() && !r.valid()) || l.dist <= r.dist) return l;
	return r;
}

// ---- restated from RayTracer/main.cpp:176-268, with ray counters (the pixels come from reftext above) ----------

Color scaleColor(float x, const Color& c) { return c * x; }  // main.cpp:57-59

float lambert(const glm::vec3& n, const glm::vec3& toLight) {  // main.cpp:258-260
	return std::max(glm::dot(glm::normalize(n), glm::normalize(toLight)), 0.0f);
}

float blinnPhong(const Material& m, const glm::vec3& n, const glm::vec3& h) {  // main.cpp:262-264
	return glm::pow(std::max(glm::dot(h, n), 0.0f), m.shininess);
}

float attenuationAt(const glm::vec3& a, float d) {  // main.cpp:266-268
	return 1.0f / (a.x + a.y * d + a.z * glm::pow(d, 2.0f));
}

Color shadeLights(ShadeCtx& cx, const glm::vec3& P, const glm::vec3& N, const Material& mat, const glm::vec3& viewPoint) {  // main.cpp:203-256
	Color sum = mat.ambient + mat.emission;  // :204
	for (Light light : cx.scene->getLights()) {  // :208
		glm::vec3 L;
		float atten;
		if (light.isPointLight()) {  // :213-217
			L = glm::vec3(light.location) - P;
			float d = glm::length(L);
			atten = attenuationAt(cx.scene->attenuation, d);
		}
		else {  // :218-221
			L = glm::vec3(light.location);
			atten = 1.0f;
		}
		Ray sray(P, glm::normalize(L));  // :222
		cx.shadow++;
		Intersection occ = cx.scene->findClosestIntersection(sray);  // :223
		if (!occ.isValidIntersection() || occ.distAlongRay >= glm::length(L) || !(cx.features & F_SHADOWS)) {  // :224
			float kd = lambert(N, L);  // :225
			glm::vec3 eye = viewPoint - P;  // :226
			glm::vec3 H = glm::normalize(glm::normalize(L) + glm::normalize(eye));  // :227
			float ks = blinnPhong(mat, N, H);  // :228
			if (cx.debug == D_DIFFUSE_INTENSITY) sum += Color(kd, kd, kd);  // :229-231
			else if (cx.debug == D_SPECULAR_INTENSITY) sum += Color(ks, ks, ks);  // :232-234
			else if (cx.debug == D_NORMAL_MAP) sum += Color(N.x, N.y, N.z);  // :235-237
			else if (cx.debug == D_LIGHT_DIR_MAP) sum += Color(H.x, H.y, H.z);  // :238-240
			else {
				if (cx.features & F_DIFFUSE) sum += scaleColor(atten, mat.diffuse) * kd * light.color;  // :243
				if (cx.features & F_SPECULAR) sum += scaleColor(atten, mat.specular) * ks * light.color;  // :246
			}
		}
		else if (cx.debug == D_SHADOW_MAP) {  // :250-252
			sum += occ.mat.diffuse;
		}
	}
	return sum;
}

Color shadeRay(ShadeCtx& cx, const Ray& ray, int depth) {  // main.cpp:176-201
	if (depth <= cx.scene->maxDepth) {  // :177
		Intersection hit = cx.scene->findClosestIntersection(ray);  // :178
		if (!hit.isValidIntersection()) return cx.scene->backgroundColor;  // :179-181
		if (cx.debug == D_PRIMARY_MAP) return Color(1.0f, 0.0f, 0.0f);  // :183-185
		glm::vec3 P = Camera::createPointFromRay(ray, hit.distAlongRay);
		Color lit = shadeLights(cx, P, hit.intersectNormal, hit.mat, ray.origin);  // :187
		Ray refl(P, glm::normalize(ray.dir - 2.0f * glm::dot(ray.dir, hit.intersectNormal) * hit.intersectNormal));  // :188
		if (cx.features & F_REFLECTIONS) {  // :189-191
			++depth;
			if (depth <= cx.scene->maxDepth) cx.reflection++;
			return lit + hit.mat.specular * shadeRay(cx, refl, depth);
		}
		return lit;  // :193
	}
	return Color();  // :199
}

}  // namespace

extern "C" {

// ---- scene lifetime -----------------------------------------------------------------------------

void* ref_scene_load(const char* path) {
	RefScene* rs = new RefScene();
	rs->scene = new Scene(std::string(path));  // Scene.cpp:15
	if (!rs->scene->loaded()) {
		delete rs;  // ~Scene would touch an unset objectTree; leak the tiny failed Scene on purpose
		return nullptr;
	}
	const std::vector<Shape*>& objs = rs->scene->getSceneObjects();
	for (size_t i = 0; i < objs.size(); ++i) rs->shapeId[objs[i]] = (int)i;
	rs->scene->backgroundColor = Color(0, 0, 0);  // main.cpp:124
	return rs;
}

void ref_scene_free(void* h) {
	RefScene* rs = (RefScene*)h;
	if (!rs) return;
	delete rs->scene;
	delete rs;
}

// ---- flattening -----------------------------------------------------------------------------------
// counts[0..9] = width, height, maxDepth, nNodes, nLeafRefs, nPrims, nTris, nSpheres, nLights, maxTreeDepth

static void countTree(const PNode* n, int depth, int64_t* nodes, int64_t* refs, int* maxDepth) {
	if (!n) return;
	(*nodes)++;
	if (depth > *maxDepth) *maxDepth = depth;
	if (n->isLeaf()) { *refs += (int64_t)n->objects.size(); return; }
	countTree(n->left, depth + 1, nodes, refs, maxDepth);
	countTree(n->right, depth + 1, nodes, refs, maxDepth);
}

void ref_scene_counts(void* h, int64_t* counts) {
	RefScene* rs = (RefScene*)h;
	const Scene* s = rs->scene;
	int64_t nodes = 0, refs = 0; int md = 0;
	countTree(s->objectTree->root, 0, &nodes, &refs, &md);
	int64_t nt = 0, ns = 0;
	for (Shape* o : s->getSceneObjects()) {
		if (dynamic_cast<Triangle*>(o)) nt++;
		else if (dynamic_cast<Sphere*>(o)) ns++;
	}
	counts[0] = s->getWidth(); counts[1] = s->getHeight(); counts[2] = s->maxDepth;
	counts[3] = nodes; counts[4] = refs; counts[5] = (int64_t)s->getSceneObjects().size();
	counts[6] = nt; counts[7] = ns; counts[8] = (int64_t)s->getLights().size(); counts[9] = md;
}

struct FlatOut {
	float* nodes;       // 8 x 4 bytes per node: bmin.xyz, a, bmax.xyz, b   (a, b are uint32 bit patterns)
	uint32_t* leafPrims;
	int64_t nextNode, nextRef;
	const std::unordered_map<const Shape*, int>* ids;
};

static const uint32_t NO_CHILD = 0x7fffffffu;
static const uint32_t LEAF_FLAG = 0x80000000u;

static uint32_t flattenNode(const PNode* n, FlatOut& out) {
	uint32_t me = (uint32_t)out.nextNode++;
	float* rec = out.nodes + (size_t)me * 8;
	glm::vec3 lo = n->box.getMin(), hi = n->box.getMax();
	rec[0] = lo.x; rec[1] = lo.y; rec[2] = lo.z;
	rec[4] = hi.x; rec[5] = hi.y; rec[6] = hi.z;
	uint32_t a, b;
	if (n->isLeaf()) {
		a = (uint32_t)out.nextRef;
		b = LEAF_FLAG | (uint32_t)n->objects.size();
		for (Shape* o : n->objects) out.leafPrims[out.nextRef++] = (uint32_t)out.ids->at(o);
	}
	else {
		a = n->left ? flattenNode(n->left, out) : NO_CHILD;
		b = n->right ? flattenNode(n->right, out) : NO_CHILD;
	}
	memcpy(rec + 3, &a, 4);
	memcpy(rec + 7, &b, 4);
	return me;
}

// Arrays must be sized from ref_scene_counts.  Layouts:
//  nodes      float[8*nNodes] (see FlatOut)          leafPrims  uint32[nLeafRefs] (scene object index)
//  triVerts   float[9*nTris] v1,v2,v3 world space    triPrim    uint32[nTris]
//  spheres    float[36*nSpheres] c.xyz, r, M[16] column-major, inverse(M)[16] column-major
//  sphPrim    uint32[nSpheres]
//  materials  float[13*nPrims] diffuse, specular, emission, ambient, shininess (one per object, not deduplicated)
//  lights     float[8*nLights] location.xyzw, color.rgb, 0
//  camera     float[10] lookFrom, lookAt, up (as stored by Camera: normalised), fovy
//  atten      float[3]
void ref_scene_flatten(void* h, float* nodes, uint32_t* leafPrims, float* triVerts, uint32_t* triPrim,
                       float* spheres, uint32_t* sphPrim, float* materials, float* lights, float* camera, float* atten) {
	RefScene* rs = (RefScene*)h;
	const Scene* s = rs->scene;
	FlatOut out; out.nodes = nodes; out.leafPrims = leafPrims; out.nextNode = 0; out.nextRef = 0; out.ids = &rs->shapeId;
	flattenNode(s->objectTree->root, out);
	int64_t it = 0, is = 0, ip = 0;
	for (Shape* o : s->getSceneObjects()) {
		if (Triangle* t = dynamic_cast<Triangle*>(o)) {
			float* r = triVerts + 9 * it;
			r[0] = t->v1.x; r[1] = t->v1.y; r[2] = t->v1.z;
			r[3] = t->v2.x; r[4] = t->v2.y; r[5] = t->v2.z;
			r[6] = t->v3.x; r[7] = t->v3.y; r[8] = t->v3.z;
			triPrim[it++] = (uint32_t)ip;
		}
		else if (Sphere* sp = dynamic_cast<Sphere*>(o)) {
			float* r = spheres + 36 * is;
			r[0] = sp->center.x; r[1] = sp->center.y; r[2] = sp->center.z; r[3] = sp->radius;
			glm::mat4 M = sp->transform;
			glm::mat4 Mi = glm::inverse(sp->transform);  // Sphere.cpp:60
			for (int c = 0; c < 4; ++c) for (int rr = 0; rr < 4; ++rr) { r[4 + c * 4 + rr] = M[c][rr]; r[20 + c * 4 + rr] = Mi[c][rr]; }
			sphPrim[is++] = (uint32_t)ip;
		}
		const Material& m = o->getMaterial();
		float* mr = materials + 13 * ip;
		mr[0] = m.diffuse.r; mr[1] = m.diffuse.g; mr[2] = m.diffuse.b;
		mr[3] = m.specular.r; mr[4] = m.specular.g; mr[5] = m.specular.b;
		mr[6] = m.emission.r; mr[7] = m.emission.g; mr[8] = m.emission.b;
		mr[9] = m.ambient.r; mr[10] = m.ambient.g; mr[11] = m.ambient.b;
		mr[12] = m.shininess;
		ip++;
	}
	int64_t il = 0;
	for (const Light& l : s->getLights()) {
		float* r = lights + 8 * il++;
		r[0] = l.location.x; r[1] = l.location.y; r[2] = l.location.z; r[3] = l.location.w;
		r[4] = l.color.r; r[5] = l.color.g; r[6] = l.color.b; r[7] = 0.0f;
	}
	const Camera& c = s->getCamera();
	camera[0] = c.lookFrom.x; camera[1] = c.lookFrom.y; camera[2] = c.lookFrom.z;
	camera[3] = c.lookAt.x; camera[4] = c.lookAt.y; camera[5] = c.lookAt.z;
	camera[6] = c.up.x; camera[7] = c.up.y; camera[8] = c.up.z;
	camera[9] = c.fovy;
	atten[0] = s->attenuation.x; atten[1] = s->attenuation.y; atten[2] = s->attenuation.z;
}

// ---- the reference's own functions, batched ---------------------------------------------------------

// Camera::createRayToPixel (Camera.cpp:15-24) for every pixel centre, row-major like main.cpp:137-152.
void ref_primary_rays(void* h, int w, int hgt, float* origins, float* dirs) {
	RefScene* rs = (RefScene*)h;
	const Camera& cam = rs->scene->getCamera();
	for (int i = 0; i < hgt; ++i) for (int j = 0; j < w; ++j) {
		Ray r = cam.createRayToPixel(j + 0.5f, i + 0.5f, w, hgt);
		size_t k = ((size_t)i * w + j) * 3;
		origins[k] = r.origin.x; origins[k + 1] = r.origin.y; origins[k + 2] = r.origin.z;
		dirs[k] = r.dir.x; dirs[k + 1] = r.dir.y; dirs[k + 2] = r.dir.z;
	}
}

// Scene::findClosestIntersection (Scene.cpp:299) for n rays; ids from the mirror traversal.  Returns the
// number of rays whose mirror distance differs bitwise from the reference's own (must be 0).
int64_t ref_trace_closest(void* h, int64_t n, const float* origins, const float* dirs, float* dist, int32_t* id, float* normal, uint64_t* visitCounts) {
	RefScene* rs = (RefScene*)h;
	Counted cnt = {0, 0, 0};
	int64_t mismatches = 0;
	for (int64_t k = 0; k < n; ++k) {
		Ray ray(glm::vec3(origins[3 * k], origins[3 * k + 1], origins[3 * k + 2]), glm::vec3(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]));
		Intersection ref = rs->scene->findClosestIntersection(ray);
		IdHit m = mirrorTraverse(ray, rs->scene->objectTree->root, &cnt);
		if (memcmp(&ref.distAlongRay, &m.dist, 4) != 0) mismatches++;
		dist[k] = ref.distAlongRay;
		id[k] = m.shape ? rs->shapeId[m.shape] : -1;
		if (normal) {
			normal[3 * k] = ref.intersectNormal.x; normal[3 * k + 1] = ref.intersectNormal.y; normal[3 * k + 2] = ref.intersectNormal.z;
		}
	}
	if (visitCounts) { visitCounts[0] = cnt.boxTests; visitCounts[1] = cnt.leafVisits; visitCounts[2] = cnt.primTests; }
	return mismatches;
}

// One primitive against n rays through the reference's virtual Shape::intersect (Triangle.cpp:147, Sphere.cpp:59).
void ref_prim_intersect(void* h, int prim, int64_t n, const float* origins, const float* dirs, float* dist, float* normal) {
	RefScene* rs = (RefScene*)h;
	const Shape* s = rs->scene->getSceneObjects()[prim];
	for (int64_t k = 0; k < n; ++k) {
		Ray ray(glm::vec3(origins[3 * k], origins[3 * k + 1], origins[3 * k + 2]), glm::vec3(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]));
		Intersection r = s->intersect(ray);
		dist[k] = r.distAlongRay;
		normal[3 * k] = r.intersectNormal.x; normal[3 * k + 1] = r.intersectNormal.y; normal[3 * k + 2] = r.intersectNormal.z;
	}
}

// AABB::intersect (AABB.cpp:49-96) hit/miss for n rays against one box.
void ref_box_intersect(const float* bmin, const float* bmax, int64_t n, const float* origins, const float* dirs, uint8_t* hit) {
	AABB box(glm::vec3(bmin[0], bmin[1], bmin[2]), glm::vec3(bmax[0], bmax[1], bmax[2]));
	for (int64_t k = 0; k < n; ++k) {
		Ray ray(glm::vec3(origins[3 * k], origins[3 * k + 1], origins[3 * k + 2]), glm::vec3(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]));
		hit[k] = box.intersect(ray).isValidIntersection() ? 1 : 0;
	}
}

// ---- pixel loop (main.cpp:137-157) over rows [row0,row1) ----
// Every pixel is computed by the reference's own computePixelColor (reftext, see the top); the loop header and the
// three stores restate main.cpp:137-138, 152-156.  bgr: full-image buffer (w*h*3), only the requested rows are
// written.  Returns seconds spent in that loop (steady_clock), excluding scene load.
// rayCounts (may be NULL) = primary, shadow, reflection: the reference keeps no such counts, so when they are asked for
// the rows are rendered a second time, outside the timed region, by the restated driver with counters; its bytes must
// equal the reference text's, otherwise the call returns -1.
double ref_render_rows(void* h, int features, int debug, int w, int hgt, int row0, int row1, int colStep, uint8_t* bgr, uint64_t* rayCounts) {
	RefScene* rs = (RefScene*)h;
	const Scene& scene = *rs->scene;
	Camera cam = rs->scene->getCamera();  // main.cpp:129
	if (colStep < 1) colStep = 1;
	reftext::featureFlags = features;
	reftext::debugFlag = (reftext::Debug)debug;
	auto t0 = std::chrono::steady_clock::now();
	for (int i = row0; i < row1; ++i) {
		for (int j = 0; j < w; j += colStep) {
			Ray ray = cam.createRayToPixel(j + 0.5f, i + 0.5f, w, hgt);  // main.cpp:152
			Color c = reftext::computePixelColor(ray, scene, 0);  // main.cpp:153
			size_t k = ((size_t)i * w + j) * 3;
			bgr[k] = c.getB();  // main.cpp:154-156
			bgr[k + 1] = c.getG();
			bgr[k + 2] = c.getR();
		}
	}
	auto t1 = std::chrono::steady_clock::now();
	if (rayCounts) {
		ShadeCtx cx; cx.scene = rs->scene; cx.features = features; cx.debug = debug; cx.primary = cx.shadow = cx.reflection = 0;
		for (int i = row0; i < row1; ++i) {
			for (int j = 0; j < w; j += colStep) {
				Ray ray = cam.createRayToPixel(j + 0.5f, i + 0.5f, w, hgt);
				cx.primary++;
				Color c = shadeRay(cx, ray, 0);
				size_t k = ((size_t)i * w + j) * 3;
				if (bgr[k] != (uint8_t)c.getB() || bgr[k + 1] != (uint8_t)c.getG() || bgr[k + 2] != (uint8_t)c.getR()) return -1.0;
			}
		}
		rayCounts[0] = cx.primary; rayCounts[1] = cx.shadow; rayCounts[2] = cx.reflection;
	}
	return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
