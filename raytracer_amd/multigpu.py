"""Image-tile sharding of one frame over the GPUs of a node.

The reference's pixel loop (RayTracer/main.cpp:137-157) has no cross-pixel state, so the image is cut into fixed tiles,
tile k (row-major) belongs to rank k % world, every rank holds the whole scene, and the only exchange step is the gather
of the per-rank packed tile buffers to rank 0, followed by the reassembly into the B,G,R image.

`torch.distributed` is plumbing here: gather over RCCL ("nccl" backend) on GPUs, or gloo for CPU rehearsals.
"""
import numpy as np

from . import _capi

TILE_W, TILE_H = 64, 16  # one tile = 16 wavefront-sized 8x8 pixel blocks


def rank_params(width, height, rank, world, features=_capi.F_ALL, debug=_capi.DEBUG_NONE, trace_mode=_capi.TRACE_EXACT,
                tile_w=TILE_W, tile_h=TILE_H):
    """Render parameters of one rank's share.  world == 1 gives the plain whole-image layout."""
    if world == 1:
        return _capi.make_params(width, height, features, debug, trace_mode)
    return _capi.make_params(width, height, features, debug, trace_mode, tile_w=tile_w, tile_h=tile_h,
                             first_tile=rank, tile_stride=world)


def tile_counts(width, height, world, tile_w=TILE_W, tile_h=TILE_H):
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    total = tiles_x * tiles_y
    per_rank = [(total - r + world - 1) // world if total > r else 0 for r in range(world)]
    return tiles_x, tiles_y, per_rank


def per_rank_bytes(width, height, world, tile_w=TILE_W, tile_h=TILE_H):
    """Size of every rank's packed tile buffer (rank 0's, the largest, so gathered buffers are equally spaced)."""
    _, _, per_rank = tile_counts(width, height, world, tile_w, tile_h)
    return per_rank[0] * tile_w * tile_h * 3


def assemble_host(gathered, width, height, world, tile_w=TILE_W, tile_h=TILE_H):
    """numpy statement of mirt_assemble_tiles: gathered[r] = rank r's packed tile buffer (uint8, per_rank_bytes long)."""
    tiles_x, tiles_y, _ = tile_counts(width, height, world, tile_w, tile_h)
    img = np.zeros((height, width, 3), np.uint8)
    for k in range(tiles_x * tiles_y):
        r, lt = k % world, k // world
        tx, ty = k % tiles_x, k // tiles_x
        tile = np.asarray(gathered[r][lt * tile_w * tile_h * 3:(lt + 1) * tile_w * tile_h * 3]).reshape(tile_h, tile_w, 3)
        x0, y0 = tx * tile_w, ty * tile_h
        w = min(tile_w, width - x0)
        h = min(tile_h, height - y0)
        img[y0:y0 + h, x0:x0 + w] = tile[:h, :w]
    return img


def pack_tiles_host(img, rank, world, tile_w=TILE_W, tile_h=TILE_H):
    """Inverse of assemble_host for one rank: the packed tile buffer a rank would produce from a full image."""
    height, width = img.shape[:2]
    tiles_x, tiles_y, per_rank = tile_counts(width, height, world, tile_w, tile_h)
    buf = np.zeros(per_rank[0] * tile_w * tile_h * 3, np.uint8)
    for lt in range(per_rank[rank]):
        k = rank + lt * world
        tx, ty = k % tiles_x, k // tiles_x
        x0, y0 = tx * tile_w, ty * tile_h
        w = min(tile_w, width - x0)
        h = min(tile_h, height - y0)
        tile = np.zeros((tile_h, tile_w, 3), np.uint8)
        tile[:h, :w] = img[y0:y0 + h, x0:x0 + w]
        buf[lt * tile_w * tile_h * 3:(lt + 1) * tile_w * tile_h * 3] = tile.reshape(-1)
    return buf


def gather_to_root(local, world, rank, dist, root=0, out=None):
    """One collective per frame: every rank's equally sized buffer to `root`.  `local` is a torch tensor (GPU for RCCL,
    CPU for gloo).  Returns the [world, n] tensor on root (`out` if given: a buffer allocated once per stream, not per
    frame), None elsewhere."""
    import torch
    if world == 1:
        return local.unsqueeze(0)
    if rank == root:
        if out is None:
            out = torch.empty((world, local.numel()), dtype=local.dtype, device=local.device)
        dist.gather(local, gather_list=list(out.unbind(0)), dst=root)
        return out
    dist.gather(local, gather_list=None, dst=root)
    return None
