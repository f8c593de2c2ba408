/*
 * mirt.h -- C ABI of the MI355X trace path ("mirt" = MI ray trace).
 *
 * The library replaces the body of the reference's createRender() between scene load and PNG save:
 * the per-pixel loop RayTracer/main.cpp:137-163 and everything it calls (computePixelColor
 * main.cpp:176-201, calculateLightingColor main.cpp:203-268, Scene::findClosestIntersection
 * Scene.cpp:299-301 -> Partition.cpp:31-53 -> Partition.h:28-39, AABB::intersect AABB.cpp:49-78,
 * Triangle::intersect Triangle.cpp:147-175, Sphere::intersect Sphere.cpp:59-87,
 * Camera::createRayToPixel / createPointFromRay Camera.cpp:15-28, Color quantisation Color.cpp:13-23).
 *
 * The host keeps the reference's scene parser, transform stack and Partition tree builder; it hands this
 * library a FLATTENED copy of the constructed Scene (plain arrays, no pointers into the Scene survive
 * mirt_scene_create) and receives the B,G,R byte buffer that Renderer::createImage (Renderer.h:9) takes.
 *
 * Conventions: every function returns 0 on success and a negative MIRT_E_* code on failure, never throws
 * and never calls exit(); the message for the calling thread's last failure is mirt_last_error().
 * A mirt_scene handle belongs to one device and must not be used from two host threads at once.
 * All floating point is IEEE binary32; matrices are column-major (GLM order: m[col*4+row]).
 */
#ifndef MIRT_H
#define MIRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIRT_OK 0
#define MIRT_E_INVALID (-1)  /* bad argument / inconsistent scene description */
#define MIRT_E_DEVICE (-2)   /* HIP runtime error (message holds hipGetErrorString) */
#define MIRT_E_NOMEM (-3)    /* host or device allocation failed */

/* ---- flattened Partition tree (Partition.h:16-43) ------------------------------------------------
 * Nodes are emitted in depth-first pre-order, left subtree first, so the order of the leaf ranges in
 * leaf_prims[] is the order in which Partition::intersect (Partition.cpp:42-43) visits leaves.
 *   interior: a = index of the left child or MIRT_NO_CHILD, b = index of the right child or MIRT_NO_CHILD
 *             (Partition::split deletes an empty child, Partition.cpp:90-97)
 *   leaf:     a = first entry in leaf_prims[], b = MIRT_LEAF_FLAG | entry count (0 is legal)
 * bmin/bmax are AABB::min / AABB::max of PartitionNode::box. */
#define MIRT_NO_CHILD 0x7fffffffu
#define MIRT_LEAF_FLAG 0x80000000u
typedef struct mirt_node {
	float bmin[3];
	uint32_t a;
	float bmax[3];
	uint32_t b;
} mirt_node;

/* Sphere.h:19-21 + Shape::transform (Shape.h:25).  Minv = glm::inverse(M), computed by the host with the
 * same GLM the reference uses so that it is the matrix Sphere::intersect recomputes on every call
 * (Sphere.cpp:60,79). */
typedef struct mirt_sphere {
	float center[3];
	float radius;
	float M[16];
	float Minv[16];
} mirt_sphere;

/* Material (SceneObjects.hpp:16-22), Color channels as raw floats (Color.h:19). */
typedef struct mirt_material {
	float diffuse[3];
	float specular[3];
	float emission[3];
	float ambient[3];
	float shininess;
} mirt_material;

/* Light (SceneObjects.hpp:24-29): location.w > 0 means point light, else directional. */
typedef struct mirt_light {
	float location[4];
	float color[3];
	float pad;
} mirt_light;

/* Camera fields (Camera.h:15-20) as the Camera object holds them: up is already normalised by the
 * Camera constructor (Camera.cpp:4); forward = look_at - look_from is recomputed by the library. */
typedef struct mirt_camera {
	float look_from[3];
	float look_at[3];
	float up[3];
	float fovy_deg;
} mirt_camera;

/* Everything createRender reads from a loaded Scene (Scene.h:13-48).  Primitive ids are indices into
 * Scene::getSceneObjects() (parse order); triangles and spheres live in separate arrays and carry
 * their primitive id.  Triangle vertices are world space (Triangle.cpp:22-25), 9 floats per triangle:
 * v1.xyz v2.xyz v3.xyz. */
typedef struct mirt_scene_desc {
	uint32_t struct_size; /* = sizeof(mirt_scene_desc) */
	uint32_t n_nodes;
	const mirt_node* nodes;
	uint32_t n_leaf_prims;
	const uint32_t* leaf_prims; /* primitive ids, leaf by leaf, each leaf in PartitionNode::objects order */
	uint32_t n_prims;
	uint32_t n_triangles;
	const float* tri_verts;
	const uint32_t* tri_prim;
	uint32_t n_spheres;
	uint32_t n_materials;
	const mirt_sphere* spheres;
	const uint32_t* sphere_prim;
	const uint32_t* prim_material; /* n_prims indices into materials[] */
	const mirt_material* materials;
	uint32_t n_lights;
	int32_t max_depth; /* Scene::maxDepth (Scene.cpp:11,131) */
	const mirt_light* lights; /* file order; lights are not transformed (Scene.cpp:39-52) */
	mirt_camera camera;
	float attenuation[3]; /* Scene::attenuation (Scene.cpp:10,56) */
} mirt_scene_desc;

/* Feature bits = enum class Feature (main.cpp:35-42); debug views = enum class Debug (main.cpp:24-33). */
#define MIRT_F_DIFFUSE 1u
#define MIRT_F_SPECULAR 2u
#define MIRT_F_SHADOWS 4u
#define MIRT_F_REFLECTIONS 8u
#define MIRT_F_ALL 15u
#define MIRT_DEBUG_DIFFUSE_INTENSITY 0u
#define MIRT_DEBUG_SPECULAR_INTENSITY 1u
#define MIRT_DEBUG_NORMAL_MAP 2u
#define MIRT_DEBUG_SHADOW_MAP 3u
#define MIRT_DEBUG_PRIMARY_MAP 4u
#define MIRT_DEBUG_LIGHT_DIRECTION_MAP 5u
#define MIRT_DEBUG_NONE 6u

/* How the nearest hit is found.  All three evaluate Triangle.cpp:148-167 and Sphere.cpp:60-82 in the reference's
 * operation order with IEEE division and square root and no fused multiply-add.
 *   EXACT:   primitives are enumerated through the library's own hierarchy over the primitives; a candidate counts only
 *            if one of the Partition leaves that hold it passes AABB::intersect (AABB.cpp:53-72, evaluated exactly).
 *            Same result as Partition::intersect (Partition.cpp:35-53) including its tie-breaks.
 *   FAST:    as EXACT without the leaf-box check: differs only for a hit the reference's tree would not have reached (the
 *            bounds the reference computes for a rotated or off-centre scaled sphere do not contain it, Sphere.cpp:13-41: the
 *            reference and EXACT draw such a sphere clipped by those bounds, FAST draws all of it), and
 *            where two primitives are hit at the same distance (it takes the lower primitive number; the reference the one
 *            its tree visits first); also ends a path at a material whose specular weight is zero instead of tracing the
 *            weightless reflection.  The triangle test is still the reference's (Triangle.cpp:147-171: plane hit and three
 *            signed-area edge tests), NOT Moeller-Trumbore: the mode relaxes which candidates count, never the arithmetic.
 *   REFWALK: walks the Partition tree that was handed over, box by box (cross-check and fallback; slow).  Its stack holds 94
 *            levels: a render that asks for REFWALK on a deeper tree returns MIRT_E_INVALID (the other modes never walk that tree).
 * In every mode a ray whose direction or origin is NaN in all three components -- the shadow ray towards a light that lies
 * exactly at the hit point (normalize(0), main.cpp:222), every ray of a camera whose up vector lies along its view -- gets the
 * reference's answer: it passes every box, no triangle, and every sphere, with a NaN distance that counts as a hit
 * (SceneObjects.hpp:35-37).  A ray with SOME components NaN, with infinite ones, or with a direction whose squared length
 * underflows or overflows -- no frame produces one; mirt_trace_rays can be handed one -- gets the reference's answer there too
 * (EXACT and REFWALK): that entry point evaluates such a ray exactly as the reference recurses, operand order of its std::min /
 * std::max included (AABB.cpp:60-61).  Inside a frame only the all-NaN case above is looked for. */
#define MIRT_TRACE_EXACT 0u
#define MIRT_TRACE_FAST 1u
#define MIRT_TRACE_REFWALK 2u

/* One image, or one rank's share of it.  The image is cut into tile_w x tile_h tiles numbered row-major;
 * tile k is rendered iff k % tile_stride == first_tile.  tile_stride == 1 renders the whole image. */
typedef struct mirt_render_params {
	uint32_t struct_size; /* = sizeof(mirt_render_params) */
	uint32_t width, height; /* Scene::getWidth/getHeight (main.cpp:120-121) */
	uint32_t feature_bits;  /* replaces the global featureFlags (main.cpp:91) */
	uint32_t debug_mode;    /* replaces the global debugFlag (main.cpp:92) */
	uint32_t trace_mode;
	uint32_t tile_w, tile_h; /* 0,0 = library default */
	uint32_t first_tile, tile_stride;
} mirt_render_params;

/* Ray counts follow the reference's control flow: primary = pixels; shadow = one per light per hit at every
 * depth (main.cpp:208-223); reflection = one per hit at depth d < maxDepth when REFLECTIONS is on (main.cpp:177,190).
 * *_traced are the rays the device really walked: a shadow query whose light term is multiplied by exact zeros is
 * answered without a walk in every mode (provably the same sum); a reflection ray whose weight mat.specular is zero in
 * all channels is left out by MIRT_TRACE_FAST only -- the reference traces it (main.cpp:190) and MIRT_TRACE_EXACT does
 * too, because 0 x (a colour that came back as inf or NaN) is not 0.  (What lies behind a path MIRT_TRACE_FAST ended is
 * then unknown: in that mode shadow_rays and reflection_rays count the paths as far as they were followed.) */
typedef struct mirt_stats {
	uint64_t primary_rays, shadow_rays, reflection_rays;
	uint64_t shadow_rays_traced, reflection_rays_traced;
	uint64_t primary_hits;
	float device_ms; /* HIP-event time of the last mirt_render call, all kernels, excluding the copy back */
	float pad;
} mirt_stats;

typedef struct mirt_scene mirt_scene;

/* Upload a deep copy of the scene to `device` (hipSetDevice ordinal).  Replaces nothing in the reference;
 * it is the hand-over point after `Scene scene(path)` (main.cpp:107). */
int mirt_scene_create(const mirt_scene_desc* desc, int device, mirt_scene** out);
void mirt_scene_destroy(mirt_scene* scene);

/* Scene cache.  mirt_scene_create spends seconds on big scenes (membership lists, hierarchy build, record packing:
 * 5 s for 10 M triangles) after the host has spent its own on Scene::Scene (Scene.cpp:15-224: parse, then `new
 * Partition`).  mirt_scene_save writes the device form of an uploaded scene to a file; mirt_scene_load brings it back
 * onto a device without the reference's loader, the tree or mirt_scene_create, ready to render the same images.
 * user[4] are four words of the caller's own that travel with the file (e.g. image width and height; may be NULL).
 * The file is tied to the library version that wrote it (a newer library refuses it) and carries a checksum; it is
 * trusted otherwise, like the library itself.  No counterpart in the reference. */
int mirt_scene_save(mirt_scene* scene, const char* path, const uint32_t user[4]);
int mirt_scene_load(const char* path, int device, mirt_scene** out, uint32_t user_out[4]);

/* Replaces the pixel loop main.cpp:137-157: fills bgr_out[3*(i*width+j)+{0,1,2}] = B,G,R for every pixel of
 * the tiles selected by params; bytes of other tiles are left untouched.  bgr_out is host memory of
 * width*height*3 bytes, top row first -- exactly what Renderer::createImage(BYTE*, std::string) takes
 * (Renderer.h:9, Renderer.cpp:13).  Synchronous.  stats may be NULL. */
int mirt_render(mirt_scene* scene, const mirt_render_params* params, uint8_t* bgr_out, mirt_stats* stats);

/* Replaces the pixel loop AND the PNG save behind it (main.cpp:137-169: the loop, then Renderer::createImage,
 * Renderer.cpp:12-15) in one call: the frame is rendered, the GPU turns it into PNG rows where it lies (filter byte, R,G,B
 * per pixel, "Up"-filtered), the rows come over in bands and CPU threads deflate each stripe of rows as soon as its band
 * has arrived, while later bands are still on the bus.  The file decodes to exactly the pixels mirt_render delivers (and
 * FreeImage would have written); its bytes differ from FreeImage's (deflate settings are free in PNG).  Whole images only
 * (tile_stride 1, tile_w = tile_h = 0).  Synchronous.  stats may be NULL. */
int mirt_render_png(mirt_scene* scene, const mirt_render_params* params, const char* path, mirt_stats* stats);

/* Same work, asynchronous, output left in device memory.  d_out receives the selected tiles packed one
 * after another in increasing tile number, each tile tile_w*tile_h pixels row-major inside the tile, 3 bytes
 * B,G,R per pixel, edge tiles padded to full size (mirt_tile_buffer_bytes gives the size).  With
 * tile_stride == 1 and tile_w == 0 the layout is the plain width*height*3 image instead.  hip_stream is a
 * hipStream_t (NULL = default stream).  No host synchronisation happens inside the call.
 * The library keeps one workspace per stream (allocated at the stream's first frame, at most 64 streams per scene), so
 * frames issued on different streams overlap on the GPU; that is how many frames are rendered fastest, because the
 * end of a frame -- the chain of bounces behind its last pixels -- leaves most of the GPU idle.  A frame that finds three
 * or more frames of the scene still in flight on other streams is launched with fewer, longer-lived waves (best
 * throughput); a frame that finds the GPU to itself with as many as fit (shortest time).  HIP maps a process's streams onto
 * GPU_MAX_HW_QUEUES hardware queues (environment variable, read when HIP initialises; default 4) and streams that share
 * a queue do not overlap: give the process at least as many queues as frames in flight. */
int mirt_render_device(mirt_scene* scene, const mirt_render_params* params, void* d_out, size_t d_out_bytes, void* hip_stream);
size_t mirt_tile_buffer_bytes(const mirt_render_params* params);

/* mirt_render_device for callers that cannot allocate device memory: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.  Same arguments, same output bytes
 * as mirt_render_device, but the frame's working memory (hit records, lists, counters, the overflow part of the
 * traversal stack) is a fixed array of MIRT_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds from the
 * moment the library is loaded; the call allocates and frees nothing (no hipMalloc/hipFree, no events, no host
 * containers) and keeps no per-stream state, so mirt_get_stats and mirt_get_launch_times do not see its frames.
 * A frame that needs more than the array holds (the message says how much: about 100 + 8*lights + 32*(maxdepth+1)
 * bytes per pixel slot) is refused with MIRT_E_INVALID.  All calls on one GPU share that GPU's array: calls whose
 * frames overlap in time on one GPU -- different streams, different scenes, different host threads -- race. */
#define MIRT_NOALLOC_SCRATCH_BYTES (256u << 20)
int mirt_render_device_noalloc(mirt_scene* scene, const mirt_render_params* params, void* d_out, size_t d_out_bytes, void* hip_stream);
/* Gives back the workspace the scene holds for hip_stream (waits for the stream's frames first).  Call it before
 * destroying a stream that has rendered with this scene; mirt_scene_destroy releases whatever is left.  No counterpart
 * in the reference (it has no streams). */
int mirt_release_stream(mirt_scene* scene, void* hip_stream);
/* Tells the library how many frames of this scene the caller keeps in flight (one per stream).  A frame's launch is sized
 * by the frames the library FINDS in flight when it is issued, and the first frames of a burst find none: they are
 * launched as if alone, fill the GPU, and the frames behind them wait (a burst of 20 4K frames on 8 streams: 13.4
 * instead of 13.9 Grays/s).  With n >= 2 every frame is launched as one of n; n <= 1 (the default) leaves the sizing to
 * what is found.  A hint only: any value renders the same bytes.  No counterpart in the reference (it has no streams). */
int mirt_set_frames_in_flight(mirt_scene* scene, uint32_t n);
/* Counters of the most recent mirt_render_device call on this scene (synchronises hip_stream). */
int mirt_get_stats(mirt_scene* scene, void* hip_stream, mirt_stats* stats);

/* Facts about the uploaded scene (sizes of the device structures, for reports and tests).  No reference counterpart;
 * the reference prints uninitialised object counters (Scene.h:47, main.cpp:337-342). */
#define MIRT_Q_HIERARCHY_DEPTH 0u  /* depth of the primitive hierarchy */
#define MIRT_Q_HIERARCHY_NODES 1u  /* its 64-byte nodes */
#define MIRT_Q_TREE_NESTED 2u      /* 1 if every Partition child box lies inside its parent's (EXACT/FAST usable) */
#define MIRT_Q_MEMBERSHIP_ENTRIES 3u /* (leaf, primitive) pairs of the Partition tree */
#define MIRT_Q_DEVICE_BYTES 4u     /* bytes of HBM held by the scene (without frame workspaces) */
#define MIRT_Q_TREE_WIDE_DEPTH 5u  /* depth of the Partition tree after single-child chains are skipped */
int mirt_scene_query(mirt_scene* scene, uint32_t what, uint64_t* value_out);

/* Per-kernel timing of the frames rendered on this scene.  After mirt_set_profiling(scene, 1) every kernel launch of a
 * frame is bracketed by HIP events recorded on the frame's own stream; mirt_get_launch_times synchronises that stream and
 * returns the duration of each launch of the most recent frame in issue order: primary, shade(0), trace(1), shade(1), ...,
 * resolve.  kinds_out[i] is MIRT_K_PRIMARY / _TRACE / _SHADE / _RESOLVE.  Replaces the reference's wall-clock report
 * (main.cpp:131,171, 1-second resolution). */
#define MIRT_K_PRIMARY 0
#define MIRT_K_TRACE 1
#define MIRT_K_SHADE 2
#define MIRT_K_RESOLVE 3
#define MIRT_K_FRAME 4 /* the whole frame in one launch (the usual case; then it is the only entry) */
int mirt_set_profiling(mirt_scene* scene, int enabled);
int mirt_get_launch_times(mirt_scene* scene, void* hip_stream, float* ms_out, int32_t* kinds_out, uint32_t capacity, uint32_t* n_out);

/* Measurement hook: the shader clock the GPU really holds under load (it lowers its clock when busy, so rates priced in
 * cycles need the clock of the same moment).  mirt_clock_probe_start puts one wave on a stream of the library's own that
 * reads the shader-cycle counter and the constant 100 MHz counter `microseconds` apart (1..1000000) and returns at once;
 * frames issued meanwhile run beside it.  mirt_clock_probe_read waits for that wave and gives cycles / second.  Replaces
 * nothing in the reference (its report knows whole seconds only, main.cpp:131,171). */
int mirt_clock_probe_start(mirt_scene* scene, uint32_t microseconds);
int mirt_clock_probe_read(mirt_scene* scene, double* shader_hz);

/* Reassemble an image from `n_ranks` packed tile buffers laid back to back in d_tiles (rank r's buffer at
 * r * per_rank_bytes, rendered with first_tile = r, tile_stride = n_ranks) into d_bgr (width*height*3).
 * The step after the RCCL gather; has no counterpart in the single-threaded reference. */
int mirt_assemble_tiles(int device, const mirt_render_params* params, uint32_t n_ranks, const void* d_tiles, size_t per_rank_bytes, void* d_bgr, void* hip_stream);

/* Narrow seam: Intersection Scene::findClosestIntersection(const Ray&) const (Scene.h:37) for n rays given as
 * host arrays of 3 floats.  prim_id_out[k] = primitive id or -1, dist_out[k] = Intersection::distAlongRay
 * (+inf on a miss), normal_out (may be NULL) = Intersection::intersectNormal. */
int mirt_trace_rays(mirt_scene* scene, uint32_t trace_mode, uint64_t n, const float* origins, const float* dirs, int32_t* prim_id_out, float* dist_out, float* normal_out);

/* Parity hook: nearest hit of every primary ray (Camera::createRayToPixel(j+0.5, i+0.5), main.cpp:152). */
int mirt_trace_primary_ids(mirt_scene* scene, uint32_t trace_mode, uint32_t width, uint32_t height, int32_t* prim_id_out, float* dist_out);

/* Parity hook: out[k] = the device's evaluation of glm::pow(x[k], y[k]) as calculateSpecularLighting uses it
 * (main.cpp:262-264), i.e. the powf of the C library the reference is linked against, restated on the device.
 * Host arrays. */
int mirt_test_powf(int device, uint64_t n, const float* x, const float* y, float* out);

const char* mirt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_H */
