#!/usr/bin/env python3
"""Wave steps and lane steps of the 10 M-triangle grid by maxdepth (statistics build): steps per ray and lane utilisation of each depth.
Usage: MIRT_LIB=raytracer_amd/libmirt_stats.so python3 exp/grid_depth_steps.py [w h]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import workloads
from raytracer_amd.device import DeviceScene
from raytracer_amd.scene import FlatScene
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
def q(s, k):
    v = C.c_uint64(); assert s.lib.mirt_scene_query(s.h, k, C.byref(v)) == 0; return v.value
prev = None
for md in (0, 1, 2, 3, 4):
    flat = FlatScene.from_text(workloads.dragon_grid(w, h, max_depth=md))
    with DeviceScene(flat, 0) as s:
        s.render(w, h)
        for k in range(100, 108): q(s, k)
        _, st = s.render(w, h)
        c = [q(s, k) for k in range(100, 108)]
        walked = st["primary_rays"] + st["shadow_rays_traced"] + st["reflection_rays_traced"]
        cur = (walked, c[0], c[1], c[2], c[3])
        d = [a - b for a, b in zip(cur, prev)] if prev else list(cur)
        print("maxdepth %d: +%.2f M rays: node wave steps %.2f M, lanes/step %.1f, node steps/ray %.1f; prim wave steps %.2f M, lanes/step %.1f, prim steps/ray %.1f; wave steps per 64 rays %.1f" % (
            md, d[0] / 1e6, d[1] / 1e6, d[2] / max(d[1], 1), d[2] / max(d[0], 1), d[3] / 1e6, d[4] / max(d[3], 1), d[4] / max(d[0], 1), (d[1] + d[3]) * 64 / max(d[0], 1)), flush=True)
        prev = cur
