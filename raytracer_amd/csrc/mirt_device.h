// Device-side data layout of a mirt_scene (internal; the public ABI is include/mirt.h).
//
// mirt_scene_create re-packs the flattened scene into two structures:
//
// (1) The primitive hierarchy ("bvh"): a binned-SAH BVH over the primitives, each primitive once, built on the host
//     (mirt_bvh.h) and collapsed to four children per node.  It is a conservative FILTER: it only decides which
//     primitives are tested at all.  The tracing kernels are limited by the number of 16-byte loads a lane issues
//     (the L1 data path serves about one diverging lane per cycle), so a node is 64 B for FOUR child boxes:
//       q0 = { origin.xyz (float), ex | ey << 8 | ez << 16 }     signed bytes: scale_k = 2^e_k
//       q1 = { lo.x[4], lo.y[4], lo.z[4], hi.x[4] }              one byte per child: bound = origin_k + byte * scale_k,
//       q2 = { hi.y[4], hi.z[4], ref0, ref1 }                    lower bounds rounded down, upper bounds rounded up
//       q3 = { ref2, ref3, 0, 0 }
//       ref: BVH_DONE (empty slot) | interior node index | BVH_LEAF [| BVH_SPHERES] | (count-1) << 27 | first
//       A leaf names `count` (1..8) consecutive records of the triangle or sphere array: no indirection.
//     Triangles and spheres are stored in hierarchy order.
//
// (2) The membership lists ("memb"): for every primitive, the leaves of the REFERENCE's Partition tree that hold it,
//     in the reference's depth-first order, each as { leaf.bmin.xyz, position } { leaf.bmax.xyz, 0 } (32 B).
//     `position` is the index of that (leaf, primitive) entry in mirt_scene_desc::leaf_prims, i.e. its rank in the
//     reference's visiting order.  A candidate hit counts only if one of these leaf boxes passes AABB::intersect
//     (evaluated with the reference's exact arithmetic); the first one that does gives the tie-break rank.
//     This reproduces Partition::intersect's result exactly (argument in mirt_bvh.h).
//
// (3) The reference tree itself, as wide nodes with the reference's own boxes plus a count-prefixed leaf stream,
//     walked exhaustively-equivalent by MODE_REFWALK.  Used as the fallback when the handed-over tree's boxes are
//     not nested, and by tests to cross-check (1)+(2) at full image sizes.
//
//   triangle, 64 B = 4 x float4:  { N.xyz, dot(v1,N) } { v3.xyz, E } { a0.xyz, c1 } { a1.xyz, K }   (tri_hit's filter)
//             + 32 B = 2 x float4:  { v1.xyz, area } { v2.xyz, 0 }   read only when the filter is undecided
//     N = normalize(cross(v2-v1, v3-v1)), dot(v1,N) and area = dot(cross(e1,e2),N)/2 are evaluated on the host with
//     exactly the operations Triangle::intersect performs per call (Triangle.cpp:151-153,162), so loading them
//     is bit-identical to recomputing them.
//   sphere, 144 B = 9 x float4: { c.xyz, powf(r,2) } M columns 0..3, inverse(M) columns 0..3
//   info (per triangle / per sphere), 16 B: { primitive id, material index, memb offset, memb count }
//   material, 64 B = 4 x float4: { diffuse, shininess } { specular, 0 } { emission, 0 } { ambient, 0 }
//   light, 32 B = 2 x float4:    { location.xyzw } { color.rgb, 0 }
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define REF_NONE 0xffffffffu
#define REF_LEAF 0x80000000u
#define PRIM_SPHERE 0x80000000u
#define BVH_DONE 0xffffffffu
#define BVH_LEAF 0x80000000u
#define BVH_SPHERES 0x40000000u
#define BVH_COUNT_SHIFT 27
#define BVH_FIRST_MASK 0x07ffffffu
#define MIRT_MAX_DEPTH_LIMIT 30

#define SPHERE_F4 9
#define TRI_F4 4

struct DCamera {
	float eye[3], u[3], v[3], w[3];
	float tanHalf;
};

struct DScene {
	// (1) primitive hierarchy
	const float4* bvh;
	const float4* tris;   // TRI_F4 per triangle: plane, v3 and the filter's edge functions (tri_hit)
	const float4* triX;   // 2 per triangle: v1, area, v2 -- the rest of what the reference's own inside test reads
	const uint4* triInfo;
	const float4* spheres;
	const uint4* sphInfo;
	// (2) membership lists
	const float4* memb;
	unsigned long long* dbg; // diagnostic event counters (MIRT_STATS builds)
	// What a ray whose direction (or origin) is NaN "hits" in the reference (a light exactly at the hit point: normalize(0), main.cpp:222):
	// every box passes (AABB.cpp:60-72: no comparison with a NaN is true), no triangle does (Triangle.cpp:163-166), every
	// sphere returns a NaN distance, which counts as valid (SceneObjects.hpp:35-37), and between two NaN answers a node takes
	// the right one (Partition.cpp:47), a leaf the first (Partition.h:33): the first sphere of the last leaf that holds one.
	// REF_NONE in a scene without spheres.
	uint32_t nanHit;
	uint32_t bvhRoot;
	float bvhMin[3], bvhMax[3];
	// (3) reference tree
	const float4* nodes;
	const uint32_t* leafs;
	float rootMin[3], rootMax[3];
	uint32_t rootRef;
	uint32_t refCull; // 1: triangles only, the walk of the reference tree may skip subtrees beyond the best hit
	// shading inputs
	const float4* materials;
	const float4* lights;
	uint32_t nLights;
	int32_t maxDepth;
	float atten[3];
	float cullSlack;     // see cull_bound()
	uint32_t shadowCull; // 1: shadow queries whose light term is provably multiplied by zero may be skipped (host-checked premises)
	DCamera cam;
};

// Division by a number fixed for the launch: q = ((x - hi(x * magic)) >> 1) + hi(x * magic)) >> shift, exact for every
// 32-bit x and every divisor >= 2 (round-up method with a 33-bit multiplier, the add folded in); shift = ~0u marks d == 1.
struct FastDiv {
	uint32_t magic, shift, d, pad;
};

// One frame's working set.  Every array is indexed by pixel slot (a pixel has exactly one path); k_frame<RECORDS> indexes
// hitO/hitD/hitN/litL by hit record instead (FRAME_RECORDS per wave of the launch, see k_frame).
struct DFrame {
	float4* hitO;      // hit records, two sets of nSlots (depth parity): { ray o.xyz, 0 }
	float4* hitD;      //            { ray d.xyz, dist }
	float4* hitN;      //            { n.xyz, prim ref }
	uint2* shRes;      // [slot * nLights + light] answer of a shadow query: { occluder ref or REF_NONE, dist bits }
	uint32_t* active;  // two lists of up to nSlots slots (depth parity): the slots whose path hit something at that depth
	uint8_t* pathDepth; // [slot] number of hits the pixel's path has had so far (0 = primary ray missed)
	float4* litL;      // [depth][slot] light sum of the hit at that depth
	float4* litS;      // [depth][slot] mat.specular of that hit (weight of the next bounce)
	uint32_t nRec;     // k_frame: stride of litL's depth index = entries of hitO/hitD/hitN (pixel slots, or records of the launch)
	uint32_t pullBelow; // k_frame: a wave takes the next batch of pixel slots while it holds fewer hits than this (64, or less: see enqueue_frame)
	uint32_t* counters; // see CTR_* below
	uint32_t* countersNext; // k_frame: the counter block of the stream's next frame, cleared by this one (or NULL)
	// Traversal-stack overflow area (entries beyond the LDS part), [level][thread of the launch].  It belongs to the
	// frame's workspace, one per stream: launches of frames on different streams run at the same time and every one of
	// them indexes the area by its own block and thread numbers.
	uint32_t* ovf;
	uint32_t ovfStride; // threads of a trace launch
	uint32_t nSlots;    // pixel slots in this launch (tiles x tile area)
	uint32_t width, height;
	uint32_t tileW, tileH, tilesX, nTilesLocal, firstTile, tileStride;
	uint32_t features, debug;
	uint32_t packedTiles; // output layout: 0 = full image, 1 = packed tile buffer
	uint32_t refillMin;   // k_trace<REFILL>: parked lanes that trigger a refill
	uint32_t preCleared;  // the output buffer and pathDepth were zeroed before the frame: missed primary rays store nothing (per-depth launches)
	uint32_t wideClear;   // k_frame: every row of an 8x8 pixel block starts on a 4-byte boundary of the output (black blocks are written as words)
	const uint32_t* blockList; // k_frame: the 8x8 pixel blocks (slot >> 6) the launch has to visit, ascending -- those that can see the
	uint32_t nListed;          // scene at all; the others are black and were cleared.  NULL: every block of the frame
	uint32_t batchMul;    // k_frame hands out 8x8 pixel block (b * batchMul) mod nBatches as its b-th batch (0: block b)
	FastDiv divArea, divSbx, divTilesX; // tile area, 8x8 blocks per tile row, tiles per image row
};

// counters[]: work cursors and statistics, zeroed at the start of every frame
#define CTR_CURSOR(launch) (launch)              // 0..63  one dynamic-chunk cursor per kernel launch of the frame
#define CTR_NHITS(depth) (64 + (depth))          // hits found at this depth
#define CTR_NSHADOW(depth) (64 + 64 + (depth))   // shadow queries traced for the hits of this depth
// k_frame (one launch per frame) uses the same block differently; words that many waves hammer lie 64 bytes apart:
#define CTR_F_NCUR 8
#define CTR_F_PCUR(k) (16 * (k)) // k < CTR_F_NCUR: block counters of the pixel slots (cursor k hands out blocks k, k + 8, ...)
#define CTR_F_HITS_PRIMARY 160  // primary rays that hit
#define CTR_F_HITS_TOTAL 161    // hits at all depths
#define CTR_F_HITS_LAST 162     // hits at the last depth (they spawn no reflection ray)
#define CTR_F_SHADOW_TRACED 163 // shadow queries that were walked
#define CTR_F_REFL_SKIPPED 164  // reflection rays with weight zero that MIRT_TRACE_FAST did not trace
// both paths: what k_note_parts finds in the finished image (words no launch of a frame uses)
#define CTR_PARTS 192           // 32 words: 1 = that part of the output holds something else than zeros
#define CTR_NPARTS 32
#define CTR_COUNT 256
#define FRAME_SLOT_BITS 27    // k_frame packs slot (or a wave's record number) and depth into one word: fewer than 2^27 slots, depth <= 31
