#!/usr/bin/env python3
"""Diagnostic: per-launch traversal event counts from a -DMIRT_STATS=1 build (MIRT_LIB=exp/libmirt_stats.so)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracer_amd.device import DeviceScene
from raytracer_amd.scene import FlatScene
name = sys.argv[1] if len(sys.argv) > 1 else "scene7"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
if name in ("grid", "spheres", "cornell", "dragon"):  # generated workloads
    from raytracer_amd import workloads
    gen = {"dragon": workloads.dragon, "cornell": workloads.cornell, "spheres": workloads.spheres, "grid": workloads.dragon_grid}
    flat = FlatScene.from_text(gen[name](w, h))
else:
    flat = FlatScene.load(os.path.join(ROOT, "tests", "golden", name + ".flat.npz"))
names = ["node_wave", "node_lane", "prim_wave", "prim_lane", "memb_wave", "memb_lane", "round_wave", "round_lane"]  # 100..107; 108..115: histogram
def read(s):
    out = {}
    for i, n in enumerate(names):
        v = C.c_uint64()
        s.lib.mirt_scene_query(s.h, 100 + i, C.byref(v))
        out[n] = v.value
    return out
with DeviceScene(flat, 0) as s:
    read(s)
    for feats, label in [(1 | 2, "primary only (no shadows, no reflections)"), (15, "all features")]:
        _, st = s.render(w, h, features=feats)
        c = read(s)
        rays = st["primary_rays"] + (st["shadow_rays_traced"] + st["reflection_rays"] if feats == 15 else 0)
        print(label, st)
        print("  per ray: node steps %.1f, prim tests %.1f, membership %.2f, rounds %.1f" % (c["node_lane"] / rays, c["prim_lane"] / rays, c["memb_lane"] / rays, c["round_lane"] / rays))
        fb = C.c_uint64()
        if s.lib.mirt_scene_query(s.h, 116, C.byref(fb)) == 0:
            print("  triangle filter fallbacks: %d (%.2e of primitive tests)" % (fb.value, fb.value / max(1, c["prim_lane"])))
        hist = []
        for b in range(8):
            v = C.c_uint64()
            s.lib.mirt_scene_query(s.h, 108 + b, C.byref(v))
            hist.append(v.value)
        if sum(hist):
            print("  node steps by walking lanes of the wave (1-8, 9-16, ..., 57-64): " + " ".join("%.3f" % (h / sum(hist)) for h in hist))
        for k in ("node", "prim", "memb", "round"):
            print("  %-6s lane utilisation %.2f  (wave iterations %d)" % (k, c[k + "_lane"] / max(1, 64 * c[k + "_wave"]), c[k + "_wave"]))
