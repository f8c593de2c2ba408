"""Diagnostic: one ray through every trace mode and the oracle.  usage: one_ray.py <plain|hard|nasty> seed ox oy oz dx dy dz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from random_scenes import hard_scene_text, nasty_scene_text, random_scene_text
from raytracer_amd import _capi
from raytracer_amd.scene import FlatScene
from raytracer_amd.device import DeviceScene
from oracle.bind import CPort
gen = {"hard": hard_scene_text, "nasty": nasty_scene_text}.get(sys.argv[1], random_scene_text)
fs = FlatScene.from_text(gen(int(sys.argv[2])))
o = np.array([[float(x) for x in sys.argv[3:6]]], np.float32)
d = np.array([[float(x) for x in sys.argv[6:9]]], np.float32)
a = CPort(fs).trace(o, d)
print("oracle", int(a[0][0]), float(a[1][0]))
for p in range(fs.n_prims):
    tri = np.flatnonzero(fs.tri_prim == p)
    if tri.size:
        dist, _ = CPort(fs).triangle(fs.tri_verts[tri[0]], o, d)
    else:
        k = np.flatnonzero(fs.sphere_prim == p)[0]
        dist, _ = CPort(fs).sphere(fs.spheres[k], o, d)
        print("  sphere prim", p, "record", fs.spheres[k].tolist())
    print("  prim", p, "tri" if tri.size else "sphere", "alone:", float(dist[0]))
with DeviceScene(fs, 0) as s:
    for name, mode in (("exact", _capi.TRACE_EXACT), ("fast", _capi.TRACE_FAST), ("refwalk", _capi.TRACE_REFWALK)):
        b = s.trace_rays(o, d, trace_mode=mode)
        print(name, int(b[0][0]), float(b[1][0]))
