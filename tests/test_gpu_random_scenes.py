"""GPU: seeded random scenes (tests/random_scenes.py: nested transforms, both kinds of light, every material command, and
constructions full of equal-distance ties) -- scene text -> the library's loader -> device, byte for byte against the C
restatement on the same flattened scene, which tests/test_random_scenes.py holds to the reference's own build on the same
seeds.  Every trace mode that promises the reference's bytes, the small-stack build, and a frame five times the size (more
waves, shared walks, several batches per wave)."""
import os

import numpy as np
import pytest

from random_scenes import MANY_LIGHTS, big_scene_text, hard_scene_text, many_lights_scene_text, nasty_scene_text, random_scene_text
from raytracer_amd import _capi
from raytracer_amd.scene import FlatScene

pytestmark = pytest.mark.gpu

KEYS = ("primary_rays", "shadow_rays", "reflection_rays")


# (MIRT_RANDOM_SEEDS=first:end runs another range of seeds, for a longer hunt than the suite's)
# 179, 401, 455, 701: two triangles at the same distance, one of which also sits in a leaf far behind the hit (the walk of
# the reference's tree has to notice the tie and drop its distance bound); 244: a Partition tree 304 levels deep
SEEDS = (range(*(int(x) for x in os.environ["MIRT_RANDOM_SEEDS"].split(":"))) if os.environ.get("MIRT_RANDOM_SEEDS")
         else list(range(48)) + [179, 244, 401, 455, 701])


# 138, 192: a point light exactly at a hit point -- the shadow ray's direction is normalize(0) = NaN, which in the reference
# passes every box, no triangle, and every sphere (DScene::nanHit); 433, 559, 601: 10^4 from the origin, where a hit's distance
# is off its ray parameter by more than a relative bound on the distance allows (cull_bound)
HARD_SEEDS = (range(*(int(x) for x in os.environ["MIRT_HARD_SEEDS"].split(":"))) if os.environ.get("MIRT_HARD_SEEDS")
              else list(range(36)) + [138, 192, 433, 559, 601])


@pytest.mark.parametrize("seed", SEEDS)
def test_random_scene_matches_the_oracle(seed, tmp_path):
    _check(seed, random_scene_text(seed), str(tmp_path))


@pytest.mark.parametrize("seed", HARD_SEEDS)
def test_hard_scene_matches_the_oracle(seed, tmp_path):
    _check(seed, hard_scene_text(seed), str(tmp_path), scale=3 if seed % 6 != 4 else 1)


# 458: a sphere flattened a thousandfold, "hit" by the reference 90 units from where it is (the discriminant's rounding noise in
# object space); 642, 722: the same for a sphere of radius zero; 784: a light direction whose squared length underflows
NASTY_SEEDS = (range(*(int(x) for x in os.environ["MIRT_NASTY_SEEDS"].split(":"))) if os.environ.get("MIRT_NASTY_SEEDS")
               else list(range(48)) + [458, 642, 722, 784])


@pytest.mark.parametrize("seed", NASTY_SEEDS)
def test_nasty_scene_matches_the_oracle(seed, tmp_path):
    _check(seed, nasty_scene_text(seed), str(tmp_path), scale=2, fast=False)


@pytest.mark.parametrize("n_lights", MANY_LIGHTS)
def test_many_lights(n_lights, tmp_path):
    """31 to 700 lights: the light loop at lengths that a mask, a counter or a byte per light could trip over."""
    _check(n_lights, many_lights_scene_text(n_lights, n_lights % 3), str(tmp_path), scale=2, fast=True)


BIG_SEEDS = range(*(int(x) for x in os.environ["MIRT_BIG_SEEDS"].split(":"))) if os.environ.get("MIRT_BIG_SEEDS") else range(10)


@pytest.mark.parametrize("seed", BIG_SEEDS)
def test_big_scene_matches_the_oracle(seed, tmp_path):
    _check(seed, big_scene_text(seed), str(tmp_path), scale=2)


def _check(seed, text, tmp_dir, scale=5, fast=True):
    from oracle.bind import CPort
    from raytracer_amd.device import DeviceScene, MirtError
    fs = FlatScene.from_text(text)
    port = CPort(fs)
    want, wst = port.render()
    W, H = fs.width * scale, fs.height * scale
    big, bst = port.render(W, H)
    libs = [None]
    if seed % 3 == 0:
        libs.append(_capi.load_library(os.path.join(os.path.dirname(_capi.__file__), "libmirt_smallstack.so")))
    for lib in libs:
        with DeviceScene(fs, 0, lib=lib) as s:
            for mode in (_capi.TRACE_EXACT, _capi.TRACE_REFWALK):
                if mode == _capi.TRACE_REFWALK and s.query()["tree_wide_depth"] + 2 > 96:
                    # the walk of the reference's tree keeps a private stack of 96 levels; such a tree is left to the other modes
                    with pytest.raises(MirtError, match="REFWALK"):
                        s.render(trace_mode=mode)
                    continue
                img, st = s.render(trace_mode=mode)
                assert np.array_equal(img, want), (seed, mode, lib is not None)
                assert [st[k] for k in KEYS] == [wst[k] for k in KEYS]
            img, st = s.render(W, H)
            assert np.array_equal(img, big), (seed, "x5", lib is not None)
            assert [st[k] for k in KEYS] == [bst[k] for k in KEYS]
            if lib is None:
                # one of the fifteen other feature sets, one of the six debug views (main.cpp:24-42)
                f, dbg = seed % 15, seed % 6
                a, ast = s.render(features=f)
                b, bst2 = port.render(features=f)
                assert np.array_equal(a, b), (seed, "features", f)
                assert [ast[k] for k in KEYS] == [bst2[k] for k in KEYS]
                a, _ = s.render(debug=dbg)
                b, _ = port.render(debug=dbg)
                assert np.array_equal(a, b), (seed, "debug", dbg)
                # the other ways a frame can be run: one launch per depth, with and without lane refilling; hit records; shared walks (by
                # default only hierarchies of 4096 nodes and more get them);
                # instead of per-pixel slots; and the scene from its cache file
                way = seed % 5
                env = {0: {"MIRT_SINGLE_LAUNCH": "0"}, 1: {"MIRT_SINGLE_LAUNCH": "0", "MIRT_REFILL_MIN": "24"}, 2: {"MIRT_FRAME_RECORDS": "1"},
                       3: {"MIRT_SHARE": "1"}, 4: {}}[way]
                old = {k: os.environ.get(k) for k in env}
                os.environ.update(env)
                try:
                    if way == 4:
                        path = os.path.join(tmp_dir, "scene.mirt")
                        s.save(path)
                        s2, _ = DeviceScene.load(path, 0)
                    else:
                        s2 = DeviceScene(fs, 0)
                    with s2:
                        img, st = s2.render(W, H)
                        assert np.array_equal(img, big), (seed, "way", way)
                        assert [st[k] for k in KEYS] == [bst[k] for k in KEYS]
                finally:
                    for k, v in old.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
            if lib is None and seed % 7 == 0:
                # frame and PNG file in one call: the file decodes to the same bytes (Renderer.cpp:12-15)
                from PIL import Image
                png = os.path.join(tmp_dir, "frame.png")
                s.render_png(png, W, H)
                assert np.array_equal(np.asarray(Image.open(png))[..., ::-1], big), (seed, "png")
            if lib is None and seed % 4 == 1:
                # one rank's share of the frame (every third 16x8 tile): the same share of the oracle's, black elsewhere
                for first in range(3):
                    part = np.full((H, W, 3), 9, np.uint8)
                    s.render(W, H, out=part, tile_w=16, tile_h=8, first_tile=first, tile_stride=3)
                    tiles = (np.arange(H)[:, None] // 8) * ((W + 15) // 16) + np.arange(W)[None, :] // 16
                    mine = tiles % 3 == first
                    assert np.array_equal(part[mine], big[mine]), (seed, "share", first)
                    assert np.all(part[~mine] == 9), (seed, "outside the share", first)
            # the ids and distances of the camera rays, bit for bit
            o, d = port.primary_rays(fs.width, fs.height)
            ids, dist, _, _ = port.trace(o, d)
            gi, gd = s.trace_primary_ids()
            nan = np.isnan(dist)  # (a NaN is a NaN: its sign and payload decide nothing)
            assert np.array_equal(gi, ids) and np.array_equal(np.isnan(gd), nan) and np.array_equal(gd[~nan].view(np.uint32), dist[~nan].view(np.uint32))
            # Scene::findClosestIntersection (Scene.h:37) for rays from anywhere: origins in and around the scene, directions of
            # every length, axis-parallel ones, components below the 1e-5 that AABB.cpp:53-58 replaces -- ids, distances and
            # normals bit for bit, in both modes that promise them
            rng = np.random.default_rng(seed)
            lo = fs.tri_verts.reshape(-1, 3).min(axis=0) if len(fs.tri_verts) else np.full(3, -2.0, np.float32)
            hi = fs.tri_verts.reshape(-1, 3).max(axis=0) if len(fs.tri_verts) else np.full(3, 2.0, np.float32)
            mid, half = (lo + hi) / 2, np.maximum((hi - lo) / 2, 1.0)
            n = 1500
            ro = (mid + rng.uniform(-1.5, 1.5, (n, 3)) * half).astype(np.float32)
            rd = rng.normal(size=(n, 3)).astype(np.float32)
            rd[0::5] /= np.linalg.norm(rd[0::5], axis=1, keepdims=True)
            rd[1::5] *= (10.0 ** rng.uniform(-3, 3, (len(rd[1::5]), 1))).astype(np.float32)
            rd[2::25, 0] = 0.0
            rd[3::25, 1] = 0.0
            rd[3::25, 2] = 0.0
            rd[4::25, 2] = np.float32(3e-6)
            rd[9::25, 0] = np.float32(-9.9e-6)
            want_ids, want_dist, want_n, _ = port.trace(ro, rd)
            for mode in (_capi.TRACE_EXACT, _capi.TRACE_REFWALK):
                if mode == _capi.TRACE_REFWALK and s.query()["tree_wide_depth"] + 2 > 96:
                    continue
                gi, gd, gn = s.trace_rays(ro, rd, trace_mode=mode)
                nan = np.isnan(want_dist)
                assert np.array_equal(gi, want_ids), (seed, "rays", mode)
                assert np.array_equal(np.isnan(gd), nan) and np.array_equal(gd[~nan].view(np.uint32), want_dist[~nan].view(np.uint32)), (seed, "ray distances", mode)
                hit = (want_ids >= 0) & ~nan
                assert np.array_equal(gn[hit].view(np.uint32), want_n[hit].view(np.uint32)), (seed, "ray normals", mode)
            # MIRT_TRACE_FAST drops the leaf-box check: it sees every hit the reference sees, nothing nearer is lost, and more
            # only where the reference's tree cannot reach a primitive (the bounds the reference gives a rotated or scaled
            # sphere do not contain it, Sphere.cpp:13-41: the reference draws such a sphere clipped, FAST draws all of it)
            if not fast:
                continue
            fi, fd = s.trace_primary_ids(trace_mode=_capi.TRACE_FAST)
            assert not np.any((ids >= 0) & (fi < 0))
            both = (ids >= 0) & (fi >= 0)
            assert np.all(fd[both] <= dist[both])
            if fs.n_prims == len(fs.tri_prim):
                assert np.array_equal(fi >= 0, ids >= 0)


@pytest.mark.parametrize("gen,seed", [(random_scene_text, k) for k in range(12)] + [(hard_scene_text, k) for k in range(6)] + [(nasty_scene_text, k) for k in range(16, 24)])
def test_rays_with_nan_infinite_and_tiny_components(gen, seed):
    """mirt_trace_rays with rays no frame produces (the recipe of tests/test_oracle_vs_reference.py, which holds the restatement to
    the reference on it): such a ray is evaluated as the reference recurses (traverse_faithful), in both modes that promise its answer."""
    from oracle.bind import CPort
    from raytracer_amd.device import DeviceScene
    rng = np.random.default_rng(1000 + seed)
    specials = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-30, -1e-30, 1e30, 1e38, 3e-6, 1.0, -1.0], np.float32)
    fs = FlatScene.from_text(gen(seed))
    n = 600
    ro = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    for k in range(n):
        for _ in range(int(rng.integers(0, 4))):
            (rd if rng.random() < 0.5 else ro)[k, rng.integers(0, 3)] = specials[rng.integers(0, len(specials))]
    rd[::50] = rng.choice(np.array([0.0, -0.0, 1e-30, -1e-30], np.float32), (len(rd[::50]), 3))  # directions without a length
    ids, dist, _, _ = CPort(fs).trace(ro, rd)
    nan = np.isnan(dist)
    with DeviceScene(fs, 0) as s:
        for mode in (_capi.TRACE_EXACT, _capi.TRACE_REFWALK):
            if s.query()["tree_wide_depth"] + 2 > 96:
                continue
            gi, gd, _ = s.trace_rays(ro, rd, trace_mode=mode)
            assert np.array_equal(gi, ids), (seed, mode, np.flatnonzero(gi != ids)[:5])
            assert np.array_equal(np.isnan(gd), nan) and np.array_equal(gd[~nan].view(np.uint32), dist[~nan].view(np.uint32)), (seed, mode)
