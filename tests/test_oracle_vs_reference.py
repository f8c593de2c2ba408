"""CPU: the C restatement against the reference's own compiled classes (oracle/_ref/libref_oracle.so), live, on inputs
the committed fixtures do not hold: other resolutions, random rays, every primitive of the small scenes.
Skipped where the reference build is absent (it is produced by `make -C oracle ref` where /root/reference exists and
travels with the tree to the GPU box)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits
from oracle.bind import CPort, REF_LIB, RefOracle

pytestmark = pytest.mark.skipif(not os.path.exists(REF_LIB), reason="reference build oracle/_ref not present")


def _scene_file(name, w, h, tmp_path):
    lines = open(os.path.join(os.path.dirname(os.path.dirname(GOLDEN)), "raytracer_amd", "data", "scenes", name + ".test")).read().split("\n")
    lines = ["size %d %d" % (w, h) if ln.strip().startswith("size ") else ln for ln in lines]
    p = tmp_path / (name + ".test")
    p.write_text("\n".join(lines))
    return str(p)


@pytest.mark.parametrize("name,size", [("scene1", (97, 61)), ("scene2", (120, 160)), ("scene4-specular", (200, 150)),
                                       ("scene5", (128, 72)), ("scene6", (80, 60))])
def test_flatten_trace_render_match_reference(name, size, tmp_path, flat_scenes):
    w, h = size
    ref = RefOracle(_scene_file(name, w, h, tmp_path))
    fs = ref.flatten()
    committed = flat_scenes(name)
    # the committed flattened scene is the reference's tree, whatever the image size
    assert np.array_equal(fs.nodes, committed.nodes) and np.array_equal(fs.leaf_prims, committed.leaf_prims)
    port = CPort(fs)
    o, d = ref.primary_rays(w, h)
    o2, d2 = port.primary_rays(w, h)
    assert np.array_equal(bits(o), bits(o2)) and np.array_equal(bits(d), bits(d2))
    ids, dist, nrm, vis = ref.trace(o, d)
    ids2, dist2, nrm2, vis2 = port.trace(o, d)
    assert np.array_equal(ids, ids2) and np.array_equal(bits(dist), bits(dist2)) and np.array_equal(vis, vis2)
    assert np.array_equal(bits(nrm[ids >= 0]), bits(nrm2[ids >= 0]))
    img, st = ref.render(w, h)
    img2, st2 = port.render(w, h)
    assert np.array_equal(img, img2)
    assert [st["primary"], st["shadow"], st["reflection"]] == [st2["primary_rays"], st2["shadow_rays"], st2["reflection_rays"]]
    ref.close()


def test_random_rays_against_dragon(tmp_path, flat_scenes):
    ref = RefOracle(_scene_file("scene7", 64, 48, tmp_path))
    port = CPort(flat_scenes("scene7"))
    rng = np.random.default_rng(7)
    n = 3000
    o = rng.uniform(-9, 9, size=(n, 3)).astype(np.float32)
    tgt = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32)
    d = (tgt - o).astype(np.float32)
    d[::2] /= np.linalg.norm(d[::2], axis=1, keepdims=True).astype(np.float32)
    d[5::50, 0] = 0.0          # exactly axis-parallel components
    d[7::50, 1] = 3e-6         # below the 1e-5 substitution of AABB.cpp:53-58
    ids, dist, nrm, vis = ref.trace(o, d)
    ids2, dist2, nrm2, vis2 = port.trace(o, d)
    assert (ids >= 0).sum() > 100
    assert np.array_equal(ids, ids2) and np.array_equal(bits(dist), bits(dist2)) and np.array_equal(vis, vis2)
    ref.close()


def test_every_primitive_of_small_scenes(tmp_path):
    rng = np.random.default_rng(3)
    for name in ("scene1", "scene2", "scene6"):
        ref = RefOracle(_scene_file(name, 32, 24, tmp_path))
        fs = ref.flatten()
        port = CPort(fs)
        o = rng.uniform(-3, 3, size=(400, 3)).astype(np.float32)
        d = rng.normal(size=(400, 3)).astype(np.float32)
        for p in range(fs.n_prims):
            want, wn = ref.prim_intersect(p, o, d)
            tri = np.flatnonzero(fs.tri_prim == p)
            if tri.size:
                got, gn = port.triangle(fs.tri_verts[tri[0]], o, d)
            else:
                got, gn = port.sphere(fs.spheres[np.flatnonzero(fs.sphere_prim == p)[0]], o, d)
            assert np.array_equal(bits(want), bits(got)), (name, p)
            hit = ~np.isinf(want)
            assert np.array_equal(bits(wn[hit]), bits(gn[hit])), (name, p)
        ref.close()


def test_cpu_rows_workers_cover_the_sample_once(tmp_path):
    """bench.py's all-cores CPU figure: P worker processes, each with rows r0, r0 + step, ...; together they trace
    exactly the rays of the sampled rows."""
    import json
    import subprocess
    import sys
    path = _scene_file("scene6", 64, 48, tmp_path)
    ref = RefOracle(path)
    _, whole = ref.render(64, 48, row_step=4)
    ref.close()
    P = 3
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = [json.loads(subprocess.run([sys.executable, "-m", "oracle.cpu_rows", path, "64", "48", str(4 * p), str(4 * P)], cwd=root,
                                      check=True, capture_output=True, timeout=120).stdout) for p in range(P)]
    assert sum(o["rays"] for o in outs) == whole["primary"] + whole["shadow"] + whole["reflection"]
    assert all(o["seconds"] > 0 for o in outs)


def test_rays_with_nan_infinite_and_tiny_components(tmp_path):
    """Rays no frame produces: components that are NaN, infinite, signed zeros, 1e-30, 1e30, 1e38.  What the reference makes of
    them hangs on the operand order of its std::min / std::max and on NaN distances counting as hits (SceneObjects.hpp:35-37);
    the restatement must follow it there too: ids, distance bits and visit counts."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from random_scenes import hard_scene_text, nasty_scene_text, random_scene_text
    from raytracer_amd.scene import FlatScene
    rng = np.random.default_rng(5)
    specials = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-30, -1e-30, 1e30, 1e38, 3e-6, 1.0, -1.0], np.float32)
    for gen, seeds in ((random_scene_text, range(0, 12)), (hard_scene_text, range(0, 6)), (nasty_scene_text, range(0, 8))):
        for seed in seeds:
            text = gen(seed)
            p = tmp_path / "s.test"
            p.write_text(text)
            ref = RefOracle(str(p))
            n = 300
            ro = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
            rd = rng.normal(size=(n, 3)).astype(np.float32)
            for k in range(n):
                for _ in range(int(rng.integers(0, 4))):
                    (rd if rng.random() < 0.5 else ro)[k, rng.integers(0, 3)] = specials[rng.integers(0, len(specials))]
            ids, dist, _, vis = ref.trace(ro, rd)
            ids2, dist2, _, vis2 = CPort(FlatScene.from_text(text)).trace(ro, rd)
            nan = np.isnan(dist)
            assert np.array_equal(ids, ids2), (gen.__name__, seed)
            assert np.array_equal(np.isnan(dist2), nan) and np.array_equal(bits(dist[~nan]), bits(dist2[~nan])), (gen.__name__, seed)
            assert np.array_equal(vis, vis2), (gen.__name__, seed)
            ref.close()
