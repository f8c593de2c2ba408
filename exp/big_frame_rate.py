"""Device time of one frame alone against its size, across the 2^27-slot line where k_frame switches to hit records."""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from raytracer_amd.device import DeviceScene  # noqa: E402
from raytracer_amd.scene import FlatScene  # noqa: E402

fs = FlatScene.load("tests/golden/scene7.flat.npz")
with DeviceScene(fs, 0) as s:
    sizes = [((3840, 2160), ""), ((3840, 2160), "1"), ((7680, 4320), ""), ((7680, 4320), "1"), ((15360, 8640), ""), ((15360, 8640), "1"),
             ((16400, 8200), ""), ((20000, 11250), ""), ((30720, 17280), "")]
    if len(sys.argv) > 1:
        sizes = [((int(a.split("x")[0]), int(a.split("x")[1])), "") for a in sys.argv[1:]]
    for (w, h), rec in sizes:
        os.environ["MIRT_FRAME_RECORDS"] = rec
        out = np.zeros((h, w, 3), np.uint8)
        best = 1e9
        for _ in range(3):
            _, st = s.render(w, h, out=out)
            best = min(best, st["device_ms"])
        rays = st["primary_rays"] + st["shadow_rays"] + st["reflection_rays"]
        print("%6dx%-6d records=%-2s %8.2f ms  %6.2f Grays/s  (%.1f M slots)" % (w, h, rec or "-", best, rays / best / 1e6, w * h / 1e6), flush=True)
        del out
