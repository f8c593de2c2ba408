// Device-side data layout of a mirt_scene (internal; the public ABI is include/mirt.h).
//
// The public flattened Partition tree (one box per node) is re-packed at mirt_scene_create into a layout
// built for one-fetch-per-level traversal on CDNA4:
//
//   wide node, 64 B = 4 x float4 (one 64-B L2 sector, fetched by a single lane as 4 dwordx4 loads):
//       q0 = { left.bmin.xyz,  left_ref  }     q1 = { left.bmax.xyz,  right_ref }
//       q2 = { right.bmin.xyz, 0 }             q3 = { right.bmax.xyz, 0 }
//     *_ref : REF_NONE | REF_LEAF|position-in-leaf-stream | index of another wide node
//     The boxes are the reference's own child boxes (PartitionNode::box), so each child's hit/miss decision is
//     AABB::intersect's decision for that node.  Interior nodes with a single child (Partition.cpp:90-97 deletes
//     an empty sibling) are skipped when the child box is contained in the parent box: child hit => parent hit
//     by monotonicity of IEEE subtraction and division, so the set of leaves a ray reaches is unchanged.
//
//   leaf stream (uint32): for every leaf, in the reference's depth-first order: count, then `count` primitive
//     references (bit 31 = sphere, low bits = index into tris / spheres).  The position of an entry in this
//     stream is its rank in the reference's visiting order; ties in distance are resolved towards the lower
//     position, which reproduces "first object in a leaf wins" (Partition.h:33) and "left subtree wins"
//     (Partition.cpp:47) under any traversal order.
//
//   triangle, 64 B = 4 x float4:  { N.xyz, dot(v1,N) } { v1.xyz, area } { v2.xyz, prim id } { v3.xyz, material }
//     N = normalize(cross(v2-v1, v3-v1)), dot(v1,N) and area = dot(cross(e1,e2),N)/2 are evaluated on the host with
//     exactly the operations Triangle::intersect performs per call (Triangle.cpp:151-153,162), so loading them
//     is bit-identical to recomputing them.
//
//   sphere, 160 B = 10 x float4: { c.xyz, powf(r,2) } M columns 0..3, inverse(M) columns 0..3, { prim id, material, r, 0 }
//
//   material, 64 B = 4 x float4: { diffuse, shininess } { specular, 0 } { emission, 0 } { ambient, 0 }
//   light, 32 B = 2 x float4:    { location.xyzw } { color.rgb, 0 }
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define REF_NONE 0xffffffffu
#define REF_LEAF 0x80000000u
#define PRIM_SPHERE 0x80000000u
#define MIRT_MAX_DEPTH_LIMIT 30

struct DCamera {
	float eye[3], u[3], v[3], w[3];
	float tanHalf;
};

struct DScene {
	const float4* nodes;
	const uint32_t* leafs;
	const float4* tris;
	const float4* spheres;
	const float4* materials;
	const float4* lights;
	float rootMin[3], rootMax[3];
	uint32_t rootRef;
	uint32_t nLights;
	int32_t maxDepth;
	float atten[3];
	DCamera cam;
};

// One frame's working set.  All queues are sized for every pixel slot of the launch.
struct DFrame {
	float4* rayO;      // ray queue: { o.xyz, slot }   (two halves, ping-pong by depth parity)
	float4* rayD;      //            { d.xyz, 0 }
	float4* hitO;      // hit queue: { o.xyz, slot }
	float4* hitD;      //            { d.xyz, dist }
	float4* hitN;      //            { n.xyz, prim ref }
	uint32_t* hit0;    // slots whose primary ray hit something (input of the resolve pass)
	float4* litL;      // [depth][slot] light sum of the hit at that depth
	float4* litS;      // [depth][slot] mat.specular of that hit (weight of the next bounce)
	uint8_t* depthCnt; // [slot] number of depths with a hit
	uint32_t* counters; // see CTR_* below
	uint32_t nSlots;    // pixel slots in this launch (tiles x tile area)
	uint32_t width, height;
	uint32_t tileW, tileH, tilesX, nTilesLocal, firstTile, tileStride;
	uint32_t features, debug;
	uint32_t packedTiles; // output layout: 0 = full image, 1 = packed tile buffer
};

// counters[]: work cursors and queue lengths, zeroed at the start of every frame
#define CTR_CURSOR(launch) (launch)              // 0..63  one dynamic-chunk cursor per kernel launch of the frame
#define CTR_NHITS(depth) (64 + (depth))          // hits found at this depth
#define CTR_NRAYS(depth) (64 + 32 + (depth))     // rays queued for this depth (depth >= 1)
#define CTR_COUNT 128
