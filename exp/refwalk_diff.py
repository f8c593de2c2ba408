"""Diagnostic: where a trace mode and the C restatement disagree on a random scene (tests/random_scenes.py).
usage: refwalk_diff.py <plain|hard|nasty> <exact|refwalk> <scale> seed..."""
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
mode = _capi.TRACE_REFWALK if sys.argv[2] == "refwalk" else _capi.TRACE_EXACT
scale = int(sys.argv[3])
for seed in [int(a) for a in sys.argv[4:]]:
    text = gen(seed)
    print("\n".join(l for l in text.split("\n") if l.split(" ")[0] in ("camera", "point", "directional", "attenuation", "maxdepth")))
    fs = FlatScene.from_text(text)
    port = CPort(fs)
    W, H = fs.width * scale, fs.height * scale
    want, wst = port.render(W, H)
    print("seed", seed, W, H, "md", fs.max_depth, "prims", fs.n_prims, "tris", len(fs.tri_prim), "nodes", len(fs.nodes), "lights", len(fs.lights))
    with DeviceScene(fs, 0) as s:
        print(" query", s.query())
        img, st = s.render(W, H, trace_mode=mode)
        d = np.argwhere((img != want).any(axis=2))
        print(" image: differing pixels", len(d), d[:6].tolist(), "stats", [(st[k], wst[k]) for k in ("primary_rays", "shadow_rays", "reflection_rays")])
        for f in (0, 1, 3, 5, 7, 8, 15):
            a, _ = s.render(W, H, trace_mode=mode, features=f)
            b, _ = port.render(W, H, features=f)
            print("  features", f, "diff", int((a != b).any(axis=2).sum()))
        for dbg in range(6):
            a, _ = s.render(W, H, trace_mode=mode, debug=dbg)
            b, _ = port.render(W, H, debug=dbg)
            print("  debug", dbg, "diff", int((a != b).any(axis=2).sum()))
        o, dd = port.primary_rays(W, H)
        ids, dist, _, _ = port.trace(o, dd)
        gi, gd = s.trace_primary_ids(W, H, trace_mode=mode)
        bad = np.flatnonzero((gi != ids) | (gd.view(np.uint32) != dist.view(np.uint32)))
        print(" primary ids differing", len(bad), [(int(i), int(ids[i]), int(gi[i]), float(dist[i]), float(gd[i])) for i in bad[:6]])
        for (y, x) in d[:4]:
            i = y * W + x
            print("  pixel", y, x, "want", want[y, x].tolist(), "got", img[y, x].tolist(), "id", int(ids[i]), "dir", dd[i].tolist(), "o", o[i].tolist())
        # the shadow queries of the first differing pixels, through the ray-batch entry points
        f32 = np.float32
        for (y, x) in d[:2]:
            i = y * W + x
            dn = dd[i] / f32(np.sqrt(f32(dd[i][0] * dd[i][0] + dd[i][1] * dd[i][1] + dd[i][2] * dd[i][2])))
            P = (o[i] + dn.astype(f32) * dist[i]).astype(f32)
            for li, L in enumerate(fs.lights):
                Lv = (L[:3] - P).astype(f32) if L[3] > 0 else L[:3].astype(f32)
                Ld = (Lv / f32(np.sqrt(f32((Lv * Lv).sum())))).astype(f32)
                a = port.trace(P[None], Ld[None])
                b = s.trace_rays(P[None], Ld[None], trace_mode=mode)
                print("   light", li, L.tolist(), "P", P.tolist(), "dir", Ld.tolist(), "oracle", int(a[0][0]), float(a[1][0]), "device", int(b[0][0]), float(b[1][0]))
        # the reflection chain of the first differing pixel, ray by ray
        def nrm3(v):
            return (v * (f32(1) / f32(np.sqrt(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))))).astype(f32)
        for (y, x) in d[:1]:
            i = y * W + x
            ro, rd = o[i].copy(), dd[i].copy()
            for depth in range(fs.max_depth + 1):
                a = port.trace(ro[None], rd[None])
                b = s.trace_rays(ro[None], rd[None], trace_mode=mode)
                print("   depth", depth, "o", ro.tolist(), "d", rd.tolist(), "oracle", int(a[0][0]), float(a[1][0]), a[2][0].tolist(), "device", int(b[0][0]), float(b[1][0]), b[2][0].tolist())
                if a[0][0] < 0:
                    break
                P = (ro + nrm3(rd) * a[1][0]).astype(f32)
                N = a[2][0]
                dn = f32(f32(f32(rd[0] * N[0]) + f32(rd[1] * N[1])) + f32(rd[2] * N[2]))
                rd = nrm3((rd - N * f32(f32(2.0) * dn)).astype(f32))
                ro = P
