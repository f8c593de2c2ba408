/*
 * mirt_node.h -- one frame on all GPUs of a node, from ONE host process (C ABI, libmirt_node.so = libmirt.so + RCCL).
 *
 * The reference's pixel loop (RayTracer/main.cpp:137-157) has no cross-pixel state, so the frame shards by pixels: the
 * image is cut into 64x16 tiles, tile k (row-major) belongs to GPU k mod N, every GPU holds the whole scene and renders
 * its tiles into a packed buffer (mirt_render_device with first_tile / tile_stride), and the only exchange is one step
 * per frame: ncclSend of every GPU's buffer / ncclRecv of all of them on GPU 0, grouped, over xGMI, followed by
 * mirt_assemble_tiles on GPU 0.  No counterpart in the single-threaded reference; this is the C++ twin of what bench.py
 * does with one process per GPU and torch.distributed.
 */
#ifndef MIRT_NODE_H
#define MIRT_NODE_H

#include "mirt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mirt_node_ctx mirt_node_ctx;

/* One scene handle, stream and tile buffer per listed device, one RCCL communicator over them (ncclCommInitAll), and a
 * one-word all-reduce across it as a self-test.  devices[0] is the root: it receives and assembles.
 * devices == NULL: devices 0 .. n_devices-1.  A device may be listed only once.
 * Rehearsal without several GPUs: with MIRT_NODE_REHEARSE=1 in the environment a device MAY be listed several times (N
 * ranks sharing one GPU); no communicator is made and the exchange is a device-to-device copy per rank instead of
 * ncclSend/ncclRecv -- the tile split, the buffers and the reassembly are the ones of the real thing. */
int mirt_node_create(const mirt_scene_desc* desc, const int* devices, uint32_t n_devices, mirt_node_ctx** out);
void mirt_node_destroy(mirt_node_ctx* ctx);

/* The frame: every GPU renders its tiles, the buffers travel to the root, the root assembles.  params describes the
 * WHOLE image (tile_stride must be 1, tile_w = tile_h = 0); the split is made here.
 * mirt_node_render_device: asynchronous; *d_bgr_out receives the address of the assembled image (width*height*3 bytes,
 * B,G,R, top row first) in the root's memory, owned by the context and valid until the next frame of the same `slot`;
 * the frame is complete when the root's stream of that slot is (mirt_node_sync).  `slot` < MIRT_NODE_SLOTS selects one of
 * the context's sets of streams and buffers, so that several frames can be in flight (as with mirt_render_device's streams).
 * mirt_node_render: synchronous, image copied into host memory; stats (may be NULL) are summed over the GPUs. */
#define MIRT_NODE_SLOTS 16
int mirt_node_render_device(mirt_node_ctx* ctx, const mirt_render_params* params, uint32_t slot, void** d_bgr_out);
int mirt_node_sync(mirt_node_ctx* ctx, uint32_t slot);
int mirt_node_render(mirt_node_ctx* ctx, const mirt_render_params* params, uint8_t* bgr_out, mirt_stats* stats);

/* 1 if the context exchanges through RCCL (ncclSend/ncclRecv), 0 in rehearsal mode; n_devices_out may be NULL. */
int mirt_node_uses_rccl(const mirt_node_ctx* ctx, uint32_t* n_devices_out);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_NODE_H */
