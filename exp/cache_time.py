#!/usr/bin/env python3
"""Seconds from nothing to a scene on the device: scene text (parse, Partition build, mirt_scene_create) against the cache file."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracer_amd import workloads
from raytracer_amd.device import DeviceScene
from raytracer_amd.scene import FlatScene
for name, gen in (("dragon", workloads.dragon), ("grid", workloads.dragon_grid)):
    t0 = time.perf_counter(); text = gen(); flat = FlatScene.from_text(text); t1 = time.perf_counter()
    s = DeviceScene(flat, 0); t2 = time.perf_counter()
    a, _ = s.render(640, 360)
    path = "/tmp/%s.mirt" % name
    t3 = time.perf_counter(); s.save(path); t4 = time.perf_counter(); s.close()
    t5 = time.perf_counter(); c, _ = DeviceScene.load(path, 0); t6 = time.perf_counter()
    b, _ = c.render(640, 360); c.close()
    print("%s: text -> host scene %.2f s, mirt_scene_create %.2f s | save %.2f s (%.0f MB), load %.2f s, same image: %s" % (name, t1 - t0, t2 - t1, t4 - t3, os.path.getsize(path) / 1e6, t6 - t5, np.array_equal(a, b)), flush=True)
    os.unlink(path)
