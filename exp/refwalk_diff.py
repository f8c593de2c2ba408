"""Diagnostic: where MIRT_TRACE_REFWALK and the C restatement disagree on a random scene (tests/random_scenes.py)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from random_scenes import random_scene_text
from raytracer_amd import _capi
from raytracer_amd.scene import FlatScene
from raytracer_amd.device import DeviceScene
from oracle.bind import CPort
for seed in [int(a) for a in sys.argv[1:]]:
    fs = FlatScene.from_text(random_scene_text(seed))
    port = CPort(fs)
    want, wst = port.render()
    print("seed", seed, fs.width, fs.height, "md", fs.max_depth, "prims", fs.n_prims, "tris", len(fs.tri_prim), "nodes", len(fs.nodes), "lights", len(fs.lights))
    with DeviceScene(fs, 0) as s:
        print(" query", s.query())
        img, st = s.render(trace_mode=_capi.TRACE_REFWALK)
        d = np.argwhere((img != want).any(axis=2))
        print(" refwalk image: differing pixels", len(d), d[:6].tolist(), "stats equal", [st[k] == wst[k] for k in ("primary_rays", "shadow_rays", "reflection_rays")])
        for f in (0, 1, 3, 5, 7, 8, 15):
            a, _ = s.render(trace_mode=_capi.TRACE_REFWALK, features=f)
            b, _ = port.render(features=f)
            print("  features", f, "diff", int((a != b).any(axis=2).sum()))
        o, dd = port.primary_rays(fs.width, fs.height)
        ids, dist, _, _ = port.trace(o, dd)
        gi, gd = s.trace_primary_ids(trace_mode=_capi.TRACE_REFWALK)
        bad = np.flatnonzero((gi != ids) | (gd.view(np.uint32) != dist.view(np.uint32)))
        print(" primary ids differing", len(bad), [(int(i), int(ids[i]), int(gi[i]), float(dist[i]), float(gd[i])) for i in bad[:6]])
        for (y, x) in d[:3]:
            print("  pixel", y, x, "want", want[y, x].tolist(), "got", img[y, x].tolist(), "id", int(ids[y * fs.width + x]))
