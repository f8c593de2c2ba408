#!/usr/bin/env python3
"""Host cost of putting one frame on a stream (mirt_render_device through ctypes), measured with a frame so small that the GPU
is never the limit; and the same for a rank's eighth of the 4K frame with the GPU's time beside it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("MIRT_HWQ", "16")  # (the box exports 4: setdefault would keep that)
import torch
from raytracer_amd import _capi, multigpu, workloads
from raytracer_amd.device import DeviceScene
from raytracer_amd.scene import FlatScene
flat = FlatScene.from_text(workloads.dragon(3840, 2160))
with DeviceScene(flat, 0) as s:
    for label, p, n in (("64x64 whole", _capi.make_params(64, 64), 64 * 64 * 3),
                        ("4K, rank 0 of 8", multigpu.rank_params(3840, 2160, 0, 8), multigpu.per_rank_bytes(3840, 2160, 8))):
        streams = [torch.cuda.Stream() for _ in range(8)]
        bufs = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(8)]
        for i in range(16): s.render_device(p, bufs[i % 8].data_ptr(), n, streams[i % 8].cuda_stream)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(200): s.render_device(p, bufs[i % 8].data_ptr(), n, streams[i % 8].cuda_stream)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s: enqueue %.1f us per frame, until the GPU is done %.1f us per frame" % (label, (t1 - t) / 200 * 1e6, (t2 - t) / 200 * 1e6), flush=True)
        s.set_profiling(True)
        s.render_device(p, bufs[0].data_ptr(), n, streams[0].cuda_stream); torch.cuda.synchronize()
        print("   one frame alone on the device: %.3f ms" % s.stats(streams[0].cuda_stream)["device_ms"])
        s.set_profiling(False)
        for x in streams: s.release_stream(x.cuda_stream)
