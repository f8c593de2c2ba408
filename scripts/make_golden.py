#!/usr/bin/env python3
"""Generate tests/golden/ from the reference itself (run in the development container only).

Everything written here is DATA produced by running the reference's own compiled classes
(oracle/_ref/libref_oracle.so, built by `make -C oracle ref` from the sources under /root/reference):

  raytracer_amd/data/scenes/<name>.test   the reference's scene files (input data, copied unchanged)
  tests/golden/<name>.flat.npz      the flattened Scene (Partition tree, primitives, materials, lights, camera)
  tests/golden/<name>.golden.npz    expected outputs at 160x120: nearest-hit id / distance / normal of every
                                    primary ray, the B,G,R image for every feature combination the reference's
                                    batch helper renders (main.cpp:353-367) and every debug view (main.cpp:24-33),
                                    ray counts, visit counts, and secondary rays sampled from the render
  tests/golden/kat_prims.npz        known-answer vectors: single rays against single primitives / boxes
"""
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.bind import RefOracle, build  # noqa: E402
from raytracer_amd import _capi  # noqa: E402

REF = "/root/reference/RayTracer"
OUT = os.path.join(ROOT, "tests", "golden")
SCENES = [  # name, path in the reference
    ("scene1", "test_scenes/scene1.test"),
    ("scene2", "test_scenes/scene2.test"),
    ("scene3", "test_scenes/scene3.test"),
    ("scene4-specular", "final_scenes/scene4-specular.test"),
    ("scene5", "final_scenes/scene5.test"),
    ("scene6", "final_scenes/scene6.test"),
    ("scene7", "final_scenes/scene7.test"),
]
W, H = 160, 120
FEATURE_COMBOS = [1, 3, 7, 11, 15]  # diffuse; +specular; +shadows; reflections w/o shadows; all (main.cpp:353-367)


def with_size(path, w, h):
    """Copy of a scene file whose `size` line is replaced (what the reference's *_small variants are)."""
    lines = open(path).read().split("\n")
    out, done = [], False
    for ln in lines:
        if ln.strip().startswith("size ") and not done:
            out.append("size %d %d" % (w, h))
            done = True
        else:
            out.append(ln)
    assert done, path
    tmp = tempfile.NamedTemporaryFile("w", suffix=".test", delete=False)
    tmp.write("\n".join(out))
    tmp.close()
    return tmp.name


def secondary_rays(fs_ids, o, d, dist, nrm, lights, rng, n=4000):
    """Shadow and reflection rays as main.cpp:188,222 would spawn them from a sample of primary hits."""
    hit = np.flatnonzero(fs_ids >= 0)
    if hit.size == 0:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32)
    pick = rng.choice(hit, size=min(n, hit.size), replace=False)
    oo, dd = [], []
    for k in pick:
        dn = d[k] / np.float32(np.sqrt(np.float32(np.dot(d[k], d[k]))))
        P = (o[k] + dn * dist[k]).astype(np.float32)
        N = nrm[k]
        r = (d[k] - np.float32(2.0) * np.float32(np.dot(d[k], N)) * N).astype(np.float32)
        oo.append(P)
        dd.append(r / np.float32(np.linalg.norm(r)))
        for L in lights:
            ld = (L[:3] - P) if L[3] > 0 else L[:3]
            oo.append(P)
            dd.append((ld / np.float32(np.linalg.norm(ld))).astype(np.float32))
    return np.array(oo, np.float32), np.array(dd, np.float32)


def main():
    build(ref=True)
    scene_dir = os.path.join(ROOT, "raytracer_amd", "data", "scenes")  # input data of the package (workloads.py) and of the tests
    os.makedirs(scene_dir, exist_ok=True)
    rng = np.random.default_rng(20251111)
    kat = {}
    for name, rel in SCENES:
        src = os.path.join(REF, rel)
        shutil.copyfile(src, os.path.join(scene_dir, name + ".test"))
        small = with_size(src, W, H)
        ref = RefOracle(small)
        fs = ref.flatten()
        fs.save(os.path.join(OUT, name + ".flat.npz"))
        o, d = ref.primary_rays(W, H)
        ids, dist, nrm, vis = ref.trace(o, d)
        so, sd = secondary_rays(ids, o, d, dist, nrm, fs.lights, rng)
        sids, sdist, snrm, _ = ref.trace(so, sd)
        g = dict(width=W, height=H, prim_id=ids, dist=dist, normal=nrm, visits_primary=vis,
                 sec_origins=so, sec_dirs=sd, sec_prim_id=sids, sec_dist=sdist, sec_normal=snrm)
        for f in FEATURE_COMBOS:
            img, st = ref.render(W, H, features=f)
            g["bgr_f%d" % f] = img
            g["rays_f%d" % f] = np.array([st["primary"], st["shadow"], st["reflection"]], np.uint64)
        for dbg in range(_capi.DEBUG_NONE):
            img, _ = ref.render(W, H, features=15, debug=dbg)
            g["bgr_debug%d" % dbg] = img
        np.savez_compressed(os.path.join(OUT, name + ".golden.npz"), **g)
        print(name, fs.summary(), "hits", int((ids >= 0).sum()), "rays(all)", g["rays_f15"].tolist())

        # known-answer vectors: a few primitives of this scene against rays aimed near them
        prims = rng.choice(fs.n_prims, size=min(6, fs.n_prims), replace=False)
        for p in prims:
            tri = np.flatnonzero(fs.tri_prim == p)
            if tri.size:
                v = fs.tri_verts[tri[0]].reshape(3, 3)
                centre, ext = v.mean(0), np.abs(v - v.mean(0)).max() + 1e-3
            else:
                s = fs.spheres[np.flatnonzero(fs.sphere_prim == p)[0]]
                M = s[4:20].reshape(4, 4).T
                centre = (M @ np.array([s[0], s[1], s[2], 1.0]))[:3]
                ext = abs(s[3]) * max(1e-3, np.abs(M[:3, :3]).max())
            n = 256
            ro = (centre + rng.normal(size=(n, 3)) * ext * 4).astype(np.float32)
            tgt = centre + rng.normal(size=(n, 3)) * ext * 0.7
            rd = (tgt - ro).astype(np.float32)
            rd[: n // 2] /= np.linalg.norm(rd[: n // 2], axis=1, keepdims=True).astype(np.float32)  # half normalised, half not
            kd, kn = ref.prim_intersect(int(p), ro, rd)
            kat["%s_p%d_o" % (name, p)] = ro
            kat["%s_p%d_d" % (name, p)] = rd
            kat["%s_p%d_dist" % (name, p)] = kd
            kat["%s_p%d_n" % (name, p)] = kn
        # boxes: a few tree nodes against primary + secondary rays
        nodes = rng.choice(fs.nodes.shape[0], size=min(4, fs.nodes.shape[0]), replace=False)
        ro = np.concatenate([o[::37], so[::11]])
        rd = np.concatenate([d[::37], sd[::11]])
        # axis-parallel and near-zero direction components exercise the 1e-5 substitution of AABB.cpp:53-58
        extra_d = np.array([[1, 0, 0], [0, -1, 0], [0, 0, 1], [1e-6, 1, 0], [-1e-6, 0, 1], [9.9e-6, -9.9e-6, 1], [1.1e-5, 1, 1]], np.float32)
        extra_o = np.tile(np.array([[0.1, 0.2, 5.0]], np.float32), (extra_d.shape[0], 1))
        ro, rd = np.concatenate([ro, extra_o]), np.concatenate([rd, extra_d])
        for ni in nodes:
            hit = ref.box_intersect(fs.nodes["bmin"][ni], fs.nodes["bmax"][ni], ro, rd)
            kat["%s_b%d_min" % (name, ni)] = fs.nodes["bmin"][ni]
            kat["%s_b%d_max" % (name, ni)] = fs.nodes["bmax"][ni]
            kat["%s_b%d_o" % (name, ni)] = ro
            kat["%s_b%d_d" % (name, ni)] = rd
            kat["%s_b%d_hit" % (name, ni)] = hit
        ref.close()
        os.unlink(small)
    np.savez_compressed(os.path.join(OUT, "kat_prims.npz"), **kat)


if __name__ == "__main__":
    main()
