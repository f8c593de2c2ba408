"""CPU: the N > 1 path (interleaved tiles -> packed buffers -> gather to rank 0 -> reassembly) with world_size 2 over
gloo.  Each rank's tiles are produced by the C oracle (there is no GPU here); what is under test is the partition
arithmetic, the collective and the reassembly that bench.py --gpus N uses."""
import os
import socket

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from raytracer_amd import multigpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, q):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle.bind import CPort
    from raytracer_amd.scene import FlatScene
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    flat = FlatScene.load(os.path.join(GOLDEN, "scene6.flat.npz"))
    p = multigpu.rank_params(w, h, rank, world)
    assert p.first_tile == rank and p.tile_stride == world
    img, st = CPort(flat).render(w, h, tile_w=p.tile_w, tile_h=p.tile_h, first_tile=rank, tile_stride=world)
    packed = torch.from_numpy(multigpu.pack_tiles_host(img, rank, world))
    assert packed.numel() == multigpu.per_rank_bytes(w, h, world)
    got = multigpu.gather_to_root(packed, world, rank, dist)
    # the same into a landing area allocated once (what bench.py does per stream): same bytes, the same tensor comes back
    landing = torch.full((world, packed.numel()), 0x5a, dtype=torch.uint8) if rank == 0 else None
    again = multigpu.gather_to_root(packed, world, rank, dist, out=landing)
    if rank == 0:
        assert again is landing and torch.equal(again, got)
    else:
        assert again is None and got is None
    rays = torch.tensor([st["primary_rays"], st["shadow_rays"], st["reflection_rays"]], dtype=torch.int64)
    dist.all_reduce(rays)
    if rank == 0:
        full = multigpu.assemble_host([got[r].numpy() for r in range(world)], w, h, world)
        q.put((full, rays.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("size", [(160, 120), (150, 70)])
def test_two_ranks_reassemble_the_single_rank_image(size, goldens):
    import torch.multiprocessing as mp
    w, h = size
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, w, h, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, rays = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if size == (160, 120):
        g = goldens("scene6")
        assert np.array_equal(full, g["bgr_f15"])
        assert rays.tolist() == g["rays_f15"].tolist()
    else:
        from oracle.bind import CPort
        from raytracer_amd.scene import FlatScene
        want, st = CPort(FlatScene.load(os.path.join(GOLDEN, "scene6.flat.npz"))).render(w, h)
        assert np.array_equal(full, want)
        assert rays.tolist() == [st["primary_rays"], st["shadow_rays"], st["reflection_rays"]]


def test_tile_partition_covers_every_tile_once():
    for (w, h, world) in [(7680, 4320, 8), (3840, 2160, 4), (100, 50, 3), (64, 16, 2)]:
        tx, ty, per_rank = multigpu.tile_counts(w, h, world)
        assert sum(per_rank) == tx * ty
        assert max(per_rank) == per_rank[0] and max(per_rank) - min(per_rank) <= 1
        assert multigpu.per_rank_bytes(w, h, world) == per_rank[0] * multigpu.TILE_W * multigpu.TILE_H * 3


def test_pack_and_assemble_are_inverse():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(70, 150, 3), dtype=np.uint8)
    for world in (1, 2, 3, 8):
        bufs = [multigpu.pack_tiles_host(img, r, world) for r in range(world)]
        assert np.array_equal(multigpu.assemble_host(bufs, 150, 70, world), img)
