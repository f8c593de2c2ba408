// Test harness (CPU): compiles the MirtFlattener of the integration guide -- extracted from the guide into
// mirt_flattener_snippet.inc by tests/test_integration_snippet.py -- against the reference's own headers and compiled
// classes, fills its arrays from a scene file loaded by the reference's Scene, and compares them byte for byte with arrays
// handed in by the caller (the oracle harness's flattening of the same file).  Private members are reached the way the
// guide says: the access keywords are redefined in this translation unit only.
#include <algorithm>
#include <cmath>
#include <cstdint>
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
#include "mirt_flattener_snippet.inc"
#undef private
#undef protected

// Returns 0 when every array equals the caller's, otherwise the (1-based) number of the first array that differs, or -1
// when the scene does not load or a count differs.
extern "C" int flattener_check(const char* path, uint32_t nNodes, const void* nodes, uint32_t nLeaf, const void* leaf, uint32_t nTris, const void* triVerts,
                               const void* triPrim, uint32_t nSph, const void* spheres, const void* sphPrim, uint32_t nPrims, const void* mats,
                               uint32_t nLights, const void* lights, const float* camera10, const float* atten3, int maxDepth) {
	Scene scene{std::string(path)};
	if (!scene.loaded()) return -1;
	MirtFlattener f;
	const mirt_scene_desc d = f.fill(scene);
	if (d.struct_size != sizeof d || d.n_nodes != nNodes || d.n_leaf_prims != nLeaf || d.n_triangles != nTris || d.n_spheres != nSph || d.n_prims != nPrims ||
	    d.n_materials != nPrims || d.n_lights != nLights || d.max_depth != maxDepth) return -1;
	if (memcmp(d.nodes, nodes, (size_t)nNodes * sizeof(mirt_node))) return 1;
	if (memcmp(d.leaf_prims, leaf, (size_t)nLeaf * 4)) return 2;
	if (memcmp(d.tri_verts, triVerts, (size_t)nTris * 36)) return 3;
	if (memcmp(d.tri_prim, triPrim, (size_t)nTris * 4)) return 4;
	if (memcmp(d.spheres, spheres, (size_t)nSph * sizeof(mirt_sphere))) return 5;
	if (memcmp(d.sphere_prim, sphPrim, (size_t)nSph * 4)) return 6;
	if (memcmp(d.materials, mats, (size_t)nPrims * sizeof(mirt_material))) return 7;
	for (uint32_t i = 0; i < nPrims; ++i) if (d.prim_material[i] != i) return 8;
	if (memcmp(d.lights, lights, (size_t)nLights * sizeof(mirt_light))) return 9;
	if (memcmp(d.camera.look_from, camera10, 12) || memcmp(d.camera.look_at, camera10 + 3, 12) || memcmp(d.camera.up, camera10 + 6, 12) || memcmp(&d.camera.fovy_deg, camera10 + 9, 4)) return 10;
	if (memcmp(d.attenuation, atten3, 12)) return 11;
	return 0;
}
