#!/usr/bin/env python3
"""mirt_render (host buffer) time against copy-band size and helper threads: MIRT_COPY_BAND_MB / MIRT_COPY_THREADS are read per process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
from raytracer_amd import workloads
from raytracer_amd.device import DeviceScene
from raytracer_amd.scene import FlatScene
w, h = 3840, 2160
flat = FlatScene.from_text(workloads.dragon(w, h))
with DeviceScene(flat, 0) as s:
    a = np.zeros((h, w, 3), np.uint8)
    for _ in range(3): s.render(w, h, out=a)
    ts = []
    for _ in range(15):
        t = time.perf_counter(); _, st = s.render(w, h, out=a); ts.append((time.perf_counter() - t) * 1e3)
    print("%%.3f ms (device %%.3f)" %% (float(np.median(ts)), st["device_ms"]))
''' % ROOT
for band in (1, 2, 3, 4, 8):
    for th in (3, 6, 10, 14):
        env = dict(os.environ, MIRT_COPY_BAND_MB=str(band), MIRT_COPY_THREADS=str(th))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print("band %d MB, %2d threads: %s" % (band, th, out.stdout.strip() or out.stderr.strip()[-200:]), flush=True)
