#!/usr/bin/env python3
"""Wall time of mirt_render_png (frame + PNG file) for the 4K and 8K Stanford dragon, against mirt_render + mirt_write_png.  GPU only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from raytracer_amd import workloads  # noqa: E402
from raytracer_amd.device import DeviceScene  # noqa: E402
from raytracer_amd.png import write_png  # noqa: E402
from raytracer_amd.scene import FlatScene  # noqa: E402

out_dir = sys.argv[1] if len(sys.argv) > 1 else "/dev/shm"
for name, gen, w, h in (("dragon 4K", workloads.dragon, 3840, 2160), ("dragon 8K", workloads.dragon, 7680, 4320), ("spheres 1080p", workloads.spheres, 1920, 1080)):
    flat = FlatScene.from_text(gen(w, h))
    path = os.path.join(out_dir, "mirt_png_time.png")
    with DeviceScene(flat, 0) as s:
        s.render_png(path)
        ts = []
        for _ in range(7):
            t = time.perf_counter()
            s.render_png(path)
            ts.append((time.perf_counter() - t) * 1e3)
        size = os.path.getsize(path)
        img = np.zeros((h, w, 3), np.uint8)
        s.render(out=img)
        write_png(path, img)
        t2 = []
        for _ in range(5):
            t = time.perf_counter()
            s.render(out=img)
            t1 = time.perf_counter()
            write_png(path, img)
            t2.append(((t1 - t) * 1e3, (time.perf_counter() - t1) * 1e3))
        print("%s: mirt_render_png %.1f ms (min %.1f; file %d bytes) | mirt_render %.1f ms + mirt_write_png %.1f ms (file %d bytes) -> %s" % (
            name, float(np.median(ts)), min(ts), size, float(np.median([a for a, _ in t2])), float(np.median([b for _, b in t2])), os.path.getsize(path), out_dir), flush=True)
    os.unlink(path)
