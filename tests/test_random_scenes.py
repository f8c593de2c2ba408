"""Random scenes (tests/random_scenes.py), CPU side: the reference's own build (oracle/_ref) loads each text, and the library's
loader and the C restatement must agree with it -- same tree, same image bytes, same ray counts.  The GPU side of the same
seeds is tests/test_gpu_random_scenes.py.  Skipped where the reference build is absent."""
import os

import numpy as np
import pytest

from oracle.bind import CPort, REF_LIB, RefOracle
from random_scenes import MANY_LIGHTS, big_scene_text, hard_scene_text, many_lights_scene_text, mutated_scene_text, nasty_scene_text, random_scene_text
from raytracer_amd.scene import FlatScene

pytestmark = pytest.mark.skipif(not os.path.exists(REF_LIB), reason="reference build oracle/_ref not present")



def _seeds(var, default):
    """(MIRT_RANDOM_SEEDS / MIRT_HARD_SEEDS / MIRT_NASTY_SEEDS / MIRT_BIG_SEEDS = first:end run another range, as on the GPU side)"""
    return range(*(int(x) for x in os.environ[var].split(":"))) if os.environ.get(var) else default


SEEDS = _seeds("MIRT_RANDOM_SEEDS", list(range(36)) + [179, 244, 401, 455, 701])  # (the last five: see tests/test_gpu_random_scenes.py)


@pytest.mark.parametrize("seed", SEEDS)
def test_loader_and_restatement_match_the_reference_on_a_random_scene(seed, tmp_path):
    _check(random_scene_text(seed), tmp_path)


@pytest.mark.parametrize("seed", _seeds("MIRT_HARD_SEEDS", list(range(36)) + [138, 192, 433, 559, 601]))
def test_loader_and_restatement_match_the_reference_on_a_hard_scene(seed, tmp_path):
    _check(hard_scene_text(seed), tmp_path)


@pytest.mark.parametrize("seed", _seeds("MIRT_NASTY_SEEDS", list(range(48)) + [458, 642, 722, 784]))
def test_loader_and_restatement_match_the_reference_on_a_nasty_scene(seed, tmp_path):
    _check(nasty_scene_text(seed), tmp_path)


@pytest.mark.parametrize("seed", _seeds("MIRT_BIG_SEEDS", range(10)))
def test_loader_and_restatement_match_the_reference_on_a_big_scene(seed, tmp_path):
    _check(big_scene_text(seed), tmp_path)


@pytest.mark.parametrize("n_lights", MANY_LIGHTS)
def test_loader_and_restatement_match_the_reference_under_many_lights(n_lights, tmp_path):
    _check(many_lights_scene_text(n_lights, n_lights % 3), tmp_path)


def _check(text, tmp_path):
    fs = FlatScene.from_text(text)
    if getattr(fs, "truncated_nodes", 0):
        pytest.skip("the reference's tree build does not terminate on this scene")
    p = tmp_path / "s.test"
    p.write_text(text)
    ref = RefOracle(str(p))
    rf = ref.flatten()
    assert np.array_equal(rf.nodes, fs.nodes) and np.array_equal(rf.leaf_prims, fs.leaf_prims)
    assert np.array_equal(rf.tri_verts.view(np.uint32), fs.tri_verts.view(np.uint32))
    assert np.array_equal(rf.spheres.view(np.uint32), fs.spheres.view(np.uint32))
    want, wst = ref.render()
    got, gst = CPort(fs).render()
    assert np.array_equal(want, got)
    assert [wst["primary"], wst["shadow"], wst["reflection"]] == [gst[k] for k in ("primary_rays", "shadow_rays", "reflection_rays")]
    # rays from anywhere (the same recipe as the GPU side): the reference's own findClosestIntersection against the restatement
    rng = np.random.default_rng(len(text))
    n = 600
    ro = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd[0::5] /= np.linalg.norm(rd[0::5], axis=1, keepdims=True)
    rd[1::5] *= (10.0 ** rng.uniform(-3, 3, (len(rd[1::5]), 1))).astype(np.float32)
    rd[2::25, 0] = 0.0
    rd[3::25, 1] = 0.0
    rd[3::25, 2] = 0.0
    rd[4::25, 2] = np.float32(3e-6)
    ids, dist, nrm, vis = ref.trace(ro, rd)
    ids2, dist2, nrm2, vis2 = CPort(fs).trace(ro, rd)
    nan = np.isnan(dist)
    assert np.array_equal(ids, ids2) and np.array_equal(np.isnan(dist2), nan) and np.array_equal(dist[~nan].view(np.uint32), dist2[~nan].view(np.uint32))
    hit = (ids >= 0) & ~nan
    assert np.array_equal(nrm[hit].view(np.uint32), nrm2[hit].view(np.uint32)) and np.array_equal(vis, vis2)
    ref.close()


@pytest.mark.parametrize("seed", range(120))
def test_loader_reads_damaged_text_like_the_reference(seed, tmp_path):
    """(1500 seeds were run once, each in a process of its own: the reference never crashed on them, and the only disagreement is
    the one the loader documents -- a scene without a camera line is an error here, an uninitialised camera there.)"""
    from raytracer_amd.scene import _ARRAYS
    text = mutated_scene_text(seed)
    p = tmp_path / "s.test"
    p.write_text(text)
    try:
        mine = FlatScene.from_text(text)
    except ValueError as e:
        assert "no camera line" in str(e)
        return
    ref = RefOracle(str(p))
    rf = ref.flatten()
    for name, _, _ in _ARRAYS:
        x, y = getattr(rf, name), getattr(mine, name)
        assert x.shape == y.shape and x.tobytes() == y.tobytes(), name
    assert (ref.width, ref.height, ref.max_depth) == (mine.width, mine.height, mine.max_depth)
    ref.close()
