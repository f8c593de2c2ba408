"""A frame of more than 2^27 (and more than 2^32 / 3) pixel slots -- k_frame with hit records, byte offsets beyond 32 bits --
against the same frame rendered as 16 ranks' shares of fewer than 2^27 slots each (plain k_frame).  usage: huge_frame.py [w h]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from raytracer_amd.device import DeviceScene  # noqa: E402
from raytracer_amd.scene import FlatScene  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (40000, 40000)
fs = FlatScene.load("tests/golden/scene7.flat.npz")
with DeviceScene(fs, 0) as s:
    whole = np.zeros((h, w, 3), np.uint8)
    t = time.time()
    _, st = s.render(w, h, out=whole)
    print("whole %dx%d: %.2f s, device %.1f ms, rays" % (w, h, time.time() - t, st["device_ms"]), st["primary_rays"], st["shadow_rays"], st["reflection_rays"], flush=True)
    acc = np.zeros((h, w, 3), np.uint8)
    tot = [0, 0, 0]
    for r in range(16):
        part = np.zeros((h, w, 3), np.uint8)
        t = time.time()
        _, ps = s.render(w, h, out=part, tile_w=1024, tile_h=8, first_tile=r, tile_stride=16)
        acc |= part
        del part
        for i, k in enumerate(("primary_rays", "shadow_rays", "reflection_rays")):
            tot[i] += ps[k]
        print("share %d: %.2f s" % (r, time.time() - t), flush=True)
    same = np.array_equal(acc, whole)
    print("bytes equal:", same, " ray counts equal:", tot == [st["primary_rays"], st["shadow_rays"], st["reflection_rays"]], " non-black bytes:", int(np.count_nonzero(whole[::97, ::89])))
    sys.exit(0 if same else 1)
