import sys, time, numpy as np
sys.path.insert(0, '.')
from raytracer_amd import workloads
from raytracer_amd.scene import FlatScene
from raytracer_amd.device import DeviceScene
fs = FlatScene.from_text(workloads.dragon(3840, 2160))
with DeviceScene(fs, 0) as s:
    out = np.zeros((2160, 3840, 3), np.uint8)
    for _ in range(3): s.render(3840, 2160, out=out)
    t0 = time.perf_counter(); n = 20
    for _ in range(n): _, st = s.render(3840, 2160, out=out)
    dt = (time.perf_counter() - t0) / n
    rays = st["primary_rays"] + st["shadow_rays"] + st["reflection_rays"]
    print("mirt_render synchronous: %.3f ms per frame, %.0f Mrays/s, device %.3f ms" % (dt * 1e3, rays / dt / 1e6, st["device_ms"]))
