#!/usr/bin/env python3
"""Per-launch device times of one frame (HIP events inside the library), for tuning.  GPU only."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from raytracer_amd import _capi  # noqa: E402
from raytracer_amd.device import DeviceScene  # noqa: E402
from raytracer_amd.scene import FlatScene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="scene7")
ap.add_argument("--workload", default=None, help="dragon | cornell | spheres | grid: generated scene text instead of --scene")
ap.add_argument("--width", type=int, default=0)
ap.add_argument("--height", type=int, default=0)
ap.add_argument("--mode", type=int, default=0)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
if a.workload:
    from raytracer_amd import workloads  # noqa: E402
    gen = {"dragon": (workloads.dragon, 3840, 2160), "dragon8k": (workloads.dragon, 7680, 4320), "cornell": (workloads.cornell, 640, 480),
           "spheres": (workloads.spheres, 1920, 1080), "grid": (workloads.dragon_grid, 3840, 2160)}  # bench.py's sizes
    fn, w0, h0 = gen[a.workload]
    a.width, a.height = a.width or w0, a.height or h0
    flat = FlatScene.from_text(fn(a.width, a.height))
    a.scene = a.workload
else:
    a.width, a.height = a.width or 3840, a.height or 2160
    flat = FlatScene.load(os.path.join(ROOT, "tests", "golden", a.scene + ".flat.npz"))
with DeviceScene(flat, 0) as s:
    out = np.zeros((a.height, a.width, 3), np.uint8)
    print(s.query())
    s.render(a.width, a.height, trace_mode=a.mode, out=out)
    s.set_profiling(True)
    acc = None
    for _ in range(a.reps):
        _, st = s.render(a.width, a.height, trace_mode=a.mode, out=out)
        lt = s.launch_times()
        acc = [(n, t) for n, t in lt] if acc is None else [(n, t0 + t) for (n, t0), (_, t) in zip(acc, lt)]
    print(a.scene, a.width, a.height, "mode", a.mode, {k: st[k] for k in ("primary_rays", "shadow_rays", "reflection_rays", "primary_hits")})
    tot = 0.0
    for n, t in acc:
        print("  %-8s %8.3f ms" % (n, t / a.reps))
        tot += t / a.reps
    print("  total    %8.3f ms  -> %.0f Mrays/s" % (tot, (st["primary_rays"] + st["shadow_rays"] + st["reflection_rays"]) / tot / 1e3))
