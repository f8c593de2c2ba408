"""CPU: the C restatement (oracle/mirt_oracle.c) against the golden vectors generated from the reference's own
compiled classes (scripts/make_golden.py).  Bar: bit-exact (float bit patterns, ids, bytes, ray counts)."""
import numpy as np
import pytest

from conftest import SCENE_NAMES, GOLDEN, bits
from oracle.bind import CPort
from raytracer_amd import _capi


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_primary_hits_bit_exact(name, flat_scenes, goldens):
    fs, g = flat_scenes(name), goldens(name)
    w, h = int(g["width"]), int(g["height"])
    port = CPort(fs)
    o, d = port.primary_rays(w, h)
    ids, dist, nrm, vis = port.trace(o, d)
    assert np.array_equal(ids, g["prim_id"])
    assert np.array_equal(bits(dist), bits(g["dist"]))
    hit = ids >= 0
    assert np.array_equal(bits(nrm[hit]), bits(g["normal"][hit]))
    assert np.array_equal(vis, g["visits_primary"])  # box tests, leaf visits, primitive tests


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_secondary_rays_bit_exact(name, flat_scenes, goldens):
    fs, g = flat_scenes(name), goldens(name)
    if g["sec_origins"].shape[0] == 0:
        pytest.skip("no hits")
    ids, dist, nrm, _ = CPort(fs).trace(g["sec_origins"], g["sec_dirs"])
    assert np.array_equal(ids, g["sec_prim_id"])
    assert np.array_equal(bits(dist), bits(g["sec_dist"]))
    hit = ids >= 0
    assert np.array_equal(bits(nrm[hit]), bits(g["sec_normal"][hit]))


@pytest.mark.parametrize("name", SCENE_NAMES)
@pytest.mark.parametrize("features", [1, 3, 7, 11, 15])
def test_render_feature_combos_byte_exact(name, features, flat_scenes, goldens):
    fs, g = flat_scenes(name), goldens(name)
    w, h = int(g["width"]), int(g["height"])
    img, st = CPort(fs).render(w, h, features=features)
    assert np.array_equal(img, g["bgr_f%d" % features])
    assert [st["primary_rays"], st["shadow_rays"], st["reflection_rays"]] == g["rays_f%d" % features].tolist()


@pytest.mark.parametrize("name", ["scene1", "scene5", "scene6", "scene7"])
@pytest.mark.parametrize("debug", range(_capi.DEBUG_NONE))
def test_render_debug_views_byte_exact(name, debug, flat_scenes, goldens):
    fs, g = flat_scenes(name), goldens(name)
    img, _ = CPort(fs).render(int(g["width"]), int(g["height"]), debug=debug)
    assert np.array_equal(img, g["bgr_debug%d" % debug])


def test_known_answer_primitives_and_boxes(flat_scenes):
    """Single primitives / boxes against single rays (incl. un-normalised directions, grazing and
    near-zero direction components), expected values from Shape::intersect / AABB::intersect."""
    with np.load(GOLDEN + "/kat_prims.npz", allow_pickle=False) as z:
        kat = {k: z[k] for k in z.files}
    n_prim = n_box = 0
    for key in kat:
        if key.endswith("_dist"):
            stem = key[:-5]
            name, p = stem.rsplit("_p", 1)
            fs = flat_scenes(name)
            port = CPort(fs)
            p = int(p)
            tri = np.flatnonzero(fs.tri_prim == p)
            if tri.size:
                dist, nrm = port.triangle(fs.tri_verts[tri[0]], kat[stem + "_o"], kat[stem + "_d"])
            else:
                dist, nrm = port.sphere(fs.spheres[np.flatnonzero(fs.sphere_prim == p)[0]], kat[stem + "_o"], kat[stem + "_d"])
            assert np.array_equal(bits(dist), bits(kat[key])), stem
            hit = ~np.isinf(dist)
            assert np.array_equal(bits(nrm[hit]), bits(kat[stem + "_n"][hit])), stem
            n_prim += 1
        elif key.endswith("_hit"):
            stem = key[:-4]
            name = stem.rsplit("_b", 1)[0]
            port = CPort(flat_scenes(name))
            hit = port.box(kat[stem + "_min"], kat[stem + "_max"], kat[stem + "_o"], kat[stem + "_d"])
            assert np.array_equal(hit, kat[key]), stem
            n_box += 1
    assert n_prim >= 20 and n_box >= 10


def test_tile_selection_matches_full_image(flat_scenes, goldens):
    fs, g = flat_scenes("scene6"), goldens("scene6")
    w, h = int(g["width"]), int(g["height"])
    port = CPort(fs)
    acc = np.zeros((h, w, 3), np.uint8)
    total = 0
    for r in range(3):
        img, st = port.render(w, h, tile_w=32, tile_h=32, first_tile=r, tile_stride=3)
        acc |= img
        total += st["primary_rays"]
    assert total == w * h
    assert np.array_equal(acc, g["bgr_f15"])
