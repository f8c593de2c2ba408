/*
 * mirt_host.h -- host side of the trace path: scene file -> flattened scene (C ABI, same library as mirt.h).
 *
 * Replaces `Scene scene(testFilePath)` (RayTracer/main.cpp:107), i.e. the scene-file parser and transform stack
 * Scene::Scene (Scene.cpp:15-224, Transform.h:5-30), the primitive constructors (Triangle.cpp:22-31, Sphere.cpp:5),
 * and the Partition tree build (Partition.cpp:4-14, 55-101 with AABB.cpp:98-164, Triangle.cpp:33-145,
 * Sphere.cpp:13-57), for callers that do not link the reference at all.  The result is the mirt_scene_desc that
 * mirt_scene_create takes.  The tree is the reference's tree, node for node and bit for bit: same boxes, same
 * children, same leaf contents in the same order -- that is what makes the device's tie-breaks the reference's.
 * Runs on CPU threads; no GPU is needed or touched by these calls.
 */
#ifndef MIRT_HOST_H
#define MIRT_HOST_H

#include "mirt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mirt_host_scene mirt_host_scene;

/* Parse a scene file in the reference's text format and build its Partition tree.
 * Scene.cpp:15-224.  Where the reference has undefined behaviour (vertex index outside `maxverts`, `vertex` before
 * `maxverts`, no `size` line) this returns MIRT_E_INVALID instead; lines the reference skips with a message
 * (unknown command, unreadable value: Scene.cpp:204-206, 235-244) are skipped silently here too.
 * threads: 1 = the tree is built on the calling thread alone; any other value = on the library's thread pool (one
 * thread per CPU, at most 16; MIRT_THREADS in the environment sets another number).  The tree is the same either way. */
int mirt_host_scene_load(const char* path, uint32_t threads, mirt_host_scene** out);

/* The same from memory (text need not be NUL-terminated). */
int mirt_host_scene_parse(const char* text, size_t len, uint32_t threads, mirt_host_scene** out);

/* Build the Partition tree for primitives that are already constructed: `prims` is a mirt_scene_desc whose
 * nodes / leaf_prims are ignored (may be NULL); everything else is deep-copied.  Partition::Partition
 * (Partition.cpp:4-14) over Scene::objects in primitive-id order. */
int mirt_host_scene_from_prims(const mirt_scene_desc* prims, uint32_t width, uint32_t height, uint32_t threads, mirt_host_scene** out);

/* The flattened scene.  desc_out's pointers point into the host scene and stay valid until it is destroyed.
 * width/height = the `size` command (Scene.cpp:121-126); either may be NULL. */
int mirt_host_scene_desc(const mirt_host_scene* scene, mirt_scene_desc* desc_out, uint32_t* width, uint32_t* height);

/* The `output` command's file name (Scene.cpp:134-139), default "test.png" (Scene.cpp:12). */
const char* mirt_host_scene_output_name(const mirt_host_scene* scene);

/* Partition::split (Partition.cpp:77-101) recurses for ever on some inputs (a box one float step wide whose only
 * objects never reach its other half); the reference then overflows its stack.  The library keeps such a node as a
 * leaf instead and counts it here.  0 means the tree is the reference's tree, node for node; otherwise the reference
 * cannot load this scene at all and the tree is the reference's down to those leaves. */
int mirt_host_scene_truncated_nodes(const mirt_host_scene* scene, uint64_t* count);

/* Seconds spent parsing and building the tree (the reference prints neither). */
int mirt_host_scene_timing(const mirt_host_scene* scene, double* parse_s, double* build_s);

void mirt_host_scene_destroy(mirt_host_scene* scene);

/* Replaces Renderer::createImage(BYTE* pixels, std::string outputFileName) (Renderer.h:9, Renderer.cpp:12-15): writes
 * the B,G,R buffer that mirt_render filled (top row first) as an 8-bit RGB PNG.  The file decodes to the pixels
 * FreeImage's would; its bytes differ (deflate settings).  Rows are compressed in stripes on `threads` CPU threads
 * (0 = number of CPUs, at most 16). */
int mirt_write_png(const char* path, const uint8_t* bgr, uint32_t width, uint32_t height, uint32_t threads);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_HOST_H */
