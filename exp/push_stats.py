#!/usr/bin/env python3
"""Traversal-stack pushes of one 4K Dragon frame, all and beyond the LDS part, for a statistics build given by MIRT_LIB."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import workloads
from raytracer_amd.device import DeviceScene
from raytracer_amd.scene import FlatScene
w, h = 3840, 2160
flat = FlatScene.from_text(workloads.dragon(w, h))
def q(s, k):
    v = C.c_uint64(); assert s.lib.mirt_scene_query(s.h, k, C.byref(v)) == 0; return v.value
with DeviceScene(flat, 0) as s:
    s.render(w, h); q(s, 117); q(s, 118)
    s.render(w, h)
    a = q(s, 117); b = q(s, 118)
    print("pushes %d, beyond the LDS part %d (%.4f)" % (a, b, b / max(1, a)))
