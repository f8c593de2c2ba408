#!/usr/bin/env python3
"""Wave steps of one 4K Dragon frame (statistics build): a measure of the hierarchy's and the scheduler's quality that does not
depend on the machine's mood.  Usage: MIRT_LIB=raytracer_amd/libmirt_stats.so [MIRT_BVH_* knobs] python3 exp/wave_steps.py [workload w h]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import workloads
from raytracer_amd.device import DeviceScene
from raytracer_amd.scene import FlatScene
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
gen = {"dragon": workloads.dragon, "cornell": workloads.cornell, "spheres": workloads.spheres, "grid": workloads.dragon_grid}[name]
flat = FlatScene.from_text(gen(w, h))
def q(s, k):
    v = C.c_uint64(); assert s.lib.mirt_scene_query(s.h, k, C.byref(v)) == 0; return v.value
with DeviceScene(flat, 0) as s:
    s.render(w, h)
    for k in range(100, 108): q(s, k)
    s.render(w, h)
    c = [q(s, k) for k in range(100, 108)]
    print("node wave steps %d (lanes %d), primitive wave steps %d (lanes %d), sum %d" % (c[0], c[1], c[2], c[3], c[0] + c[2]))
