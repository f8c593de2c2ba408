#!/usr/bin/env python3
"""The reference's remaining scene files (run in the development container only): every file under test_scenes/ and final_scenes/
that tests/golden does not hold already -- the variants of scene2, scene3 and scene4 its authors kept, and the *_small copies
(which differ in the `size` line only).  Copies the files unchanged to raytracer_amd/data/scenes/more/ (input data) and writes
tests/golden/more_scenes.golden.npz: per file the B,G,R image the reference renders at the file's own size (capped at 320 pixels
of width, as the *_small files do it: the `size` line replaced), and its ray counts, all features on."""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from make_golden import with_size  # noqa: E402
from oracle.bind import RefOracle, build  # noqa: E402

REF = "/root/reference/RayTracer"
HAVE = {"test_scenes/scene1.test", "test_scenes/scene2.test", "test_scenes/scene3.test", "final_scenes/scene4-specular.test",
        "final_scenes/scene5.test", "final_scenes/scene6.test", "final_scenes/scene7.test"}
SKIP_BIG = {"scene7_small.test", "scene7_very_small.test", "scene5_small.test"}  # copies of scene7 / scene5 with another size line

build()
dst = os.path.join(ROOT, "raytracer_amd", "data", "scenes", "more")
os.makedirs(dst, exist_ok=True)
out = {}
names = []
for sub in ("test_scenes", "final_scenes"):
    for f in sorted(os.listdir(os.path.join(REF, sub))):
        rel = sub + "/" + f
        if rel in HAVE or f in SKIP_BIG or os.path.isdir(os.path.join(REF, rel)):
            continue
        src = os.path.join(REF, rel)
        ref = RefOracle(src)
        w, h = ref.width, ref.height
        ref.close()
        if w > 320:
            h = max(1, h * 320 // w)
            w = 320
        tmp = with_size(src, w, h)
        ref = RefOracle(tmp)
        img, st = ref.render(w, h)
        ref.close()
        os.unlink(tmp)
        name = f  # (scene3_light and scene3_light.test are two files)
        shutil.copyfile(src, os.path.join(dst, f))
        names.append(f)
        out[name + ":bgr"] = img
        out[name + ":rays"] = np.array([st["primary"], st["shadow"], st["reflection"]], np.int64)
        print(rel, w, h, "rays", st["primary"], st["shadow"], st["reflection"], "nonblack %.2f" % (img.any(axis=2).mean()))
out["files"] = np.array(names)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "more_scenes.golden.npz"), **out)
print("wrote", len(names), "scenes")
