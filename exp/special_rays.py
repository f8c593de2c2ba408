"""Exploration (GPU): rays with NaN, infinite, denormal-range and huge components through random scenes, device against the C restatement
(which tests/ hold to the reference on the same recipe)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from random_scenes import hard_scene_text, nasty_scene_text, random_scene_text
from raytracer_amd import _capi
from raytracer_amd.scene import FlatScene
from raytracer_amd.device import DeviceScene
from oracle.bind import CPort
rng = np.random.default_rng(5)
specials = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-30, -1e-30, 1e30, 1e38, 3e-6, 1.0, -1.0], np.float32)
tot = {"exact": [0, 0], "refwalk": [0, 0]}
shown = 0
for gen, seeds in ((random_scene_text, range(0, 40)), (hard_scene_text, range(0, 24)), (nasty_scene_text, range(0, 32))):
    for seed in seeds:
        fs = FlatScene.from_text(gen(seed))
        n = 400
        ro = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
        rd = rng.normal(size=(n, 3)).astype(np.float32)
        for k in range(n):
            for _ in range(int(rng.integers(0, 4))):
                (rd if rng.random() < 0.5 else ro)[k, rng.integers(0, 3)] = specials[rng.integers(0, len(specials))]
        ids, dist, _, _ = CPort(fs).trace(ro, rd)
        with DeviceScene(fs, 0) as s:
            for name, mode in (("exact", _capi.TRACE_EXACT), ("refwalk", _capi.TRACE_REFWALK)):
                if mode == _capi.TRACE_REFWALK and s.query()["tree_wide_depth"] + 2 > 96:
                    continue
                gi, gd, _ = s.trace_rays(ro, rd, trace_mode=mode)
                nan = np.isnan(dist)
                bad = np.flatnonzero((gi != ids) | (np.isnan(gd) != nan) | (~nan & (gd.view(np.uint32) != dist.view(np.uint32))))
                tot[name][0] += n; tot[name][1] += len(bad)
                for k in bad[:2]:
                    if shown < 40:
                        shown += 1
                        print(gen.__name__, seed, name, "o", ro[k].tolist(), "d", rd[k].tolist(), "oracle", int(ids[k]), float(dist[k]), "device", int(gi[k]), float(gd[k]))
print(tot)
