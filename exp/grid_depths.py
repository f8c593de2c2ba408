#!/usr/bin/env python3
"""Per-depth rays and launch times of the 10 M-triangle grid at 4K: which depths are slow per ray.
Usage: python3 exp/grid_depths.py [w h]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import workloads
from raytracer_amd.device import DeviceScene
from raytracer_amd.scene import FlatScene
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
prev = None
only = int(os.environ.get("GD_ONLY", "0"))
for md in ([only] if only else range(1, 9)):
    flat = FlatScene.from_text(workloads.dragon_grid(w, h, max_depth=md))
    with DeviceScene(flat, 0) as s:
        s.render(w, h)
        s.set_profiling(True)
        img, st = s.render(w, h)
        walked = st["primary_rays"] + st["shadow_rays_traced"] + st["reflection_rays_traced"]
        lt = s.launch_times()
        tot = sum(ms for _, ms in lt)
        print("maxdepth %d: walked %.2f M (+%.2f M) shadow %.2f M refl %.2f M  device %.2f ms (+%.2f)" % (
            md, walked / 1e6, (walked - (prev[0] if prev else 0)) / 1e6, st["shadow_rays_traced"] / 1e6, st["reflection_rays_traced"] / 1e6,
            tot, tot - (prev[1] if prev else 0)), flush=True)
        print("   image md5", hashlib.md5(img.tobytes()).hexdigest())
        if md == 8:
            for k, ms in lt: print("   %-12s %.3f ms" % (k, ms))
        prev = (walked, tot)
    del flat
