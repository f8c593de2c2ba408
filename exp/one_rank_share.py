#!/usr/bin/env python3
"""What one rank of an N-GPU run does, measured on one GPU: rank r's share of the 4K Dragon frame (interleaved 64x16 tiles) rendered
with several frames in flight.  ms per frame here bounds the N-GPU frame rate from above (the gather and assembly come on top).
Usage: one_rank_share.py [world] [frames in flight]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("MIRT_HWQ", "16")  # (the box exports 4: setdefault would keep that)
import torch
from raytracer_amd import multigpu, workloads
from raytracer_amd.device import DeviceScene
from raytracer_amd.scene import FlatScene
w, h = 3840, 2160
worlds = [int(sys.argv[1])] if len(sys.argv) > 1 else [1, 2, 4, 8]
inflight = int(sys.argv[2]) if len(sys.argv) > 2 else 8
flat = FlatScene.from_text(workloads.dragon(w, h))
with DeviceScene(flat, 0) as s:
    _, st = s.render(w, h)
    rays = st["primary_rays"] + st["shadow_rays"] + st["reflection_rays"]
    for world in worlds:
        worst = 0.0
        per = []
        for rank in range(world):
            p = multigpu.rank_params(w, h, rank, world)
            n = multigpu.per_rank_bytes(w, h, world) if world > 1 else w * h * 3
            streams = [torch.cuda.Stream() for _ in range(inflight)]
            bufs = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(inflight)]
            def frames(k):
                for i in range(k):
                    s.render_device(p, bufs[i % inflight].data_ptr(), n, streams[i % inflight].cuda_stream)
                torch.cuda.synchronize()
            frames(2 * inflight)
            t = time.perf_counter(); frames(160); dt = (time.perf_counter() - t) / 160 * 1e3
            worst = max(worst, dt)
            per.append(round(dt, 3))
            for x in streams: s.release_stream(x.cuda_stream)
        print("world %d, %d frames in flight: slowest rank %.3f ms per frame -> at most %.1f Grays/s for the whole frame; per rank %s" % (world, inflight, worst, rays / worst / 1e6, per), flush=True)
