"""How fast can the host issue frames?  (time of the render_device calls alone vs the time until the GPU is done)"""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = "16"
sys.path.insert(0, '.')
import torch
from raytracer_amd import workloads, _capi
from raytracer_amd.scene import FlatScene
from raytracer_amd.device import DeviceScene
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
w, h = {"cornell": (640, 480), "spheres": (1920, 1080), "dragon": (3840, 2160)}[name]
fs = FlatScene.from_text(getattr(workloads, name)(w, h))
s = DeviceScene(fs, 0)
p = _capi.make_params(w, h)
streams = [torch.cuda.Stream() for _ in range(8)]
bufs = [torch.empty(w * h * 3, dtype=torch.uint8, device="cuda") for _ in range(8)]
def frames(n):
    t0 = time.perf_counter()
    for i in range(n):
        s.render_device(p, bufs[i % 8].data_ptr(), w * h * 3, streams[i % 8].cuda_stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return t1 - t0, time.perf_counter() - t0
frames(16)
a, b = frames(400)
print("%s: host issue %.3f ms/frame, until done %.3f ms/frame" % (name, a / 400 * 1e3, b / 400 * 1e3))
