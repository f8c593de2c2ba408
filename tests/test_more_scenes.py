"""Every scene file the reference ships, not only the seven of tests/golden/*.flat.npz: the variants of scene2, scene3 and scene4 its
authors kept under test_scenes/ and final_scenes/, and the *_small copies.  The expected images and ray counts were rendered by the
reference's own build (scripts/make_golden_all_scenes.py -> tests/golden/more_scenes.golden.npz; widths above 320 are reduced by
replacing the `size` line, which is what the reference's own *_small files are); the files are input data, copied unchanged to
raytracer_amd/data/scenes/more/.  CPU: the library's loader + the C restatement.  GPU: the device, in both modes that promise the bytes."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from raytracer_amd import _capi, workloads
from raytracer_amd.scene import FlatScene

with np.load(os.path.join(GOLDEN, "more_scenes.golden.npz"), allow_pickle=False) as z:
    MORE = {k: z[k] for k in z.files}
FILES = [str(f) for f in MORE["files"]]
KEYS = ("primary_rays", "shadow_rays", "reflection_rays")


def _scene(f):
    name = f
    want = MORE[name + ":bgr"]
    h, w = want.shape[:2]
    text = open(os.path.join(ROOT, "raytracer_amd", "data", "scenes", "more", f), "rb").read()
    return FlatScene.from_text(workloads._resize(text, w, h)), want, [int(x) for x in MORE[name + ":rays"]]


@pytest.mark.parametrize("f", FILES)
def test_restatement_renders_the_references_image(f):
    from oracle.bind import CPort
    fs, want, rays = _scene(f)
    got, st = CPort(fs).render()
    assert np.array_equal(got, want)
    assert [st[k] for k in KEYS] == rays


@pytest.mark.gpu
@pytest.mark.parametrize("f", FILES)
def test_device_renders_the_references_image(f):
    from raytracer_amd.device import DeviceScene
    fs, want, rays = _scene(f)
    with DeviceScene(fs, 0) as s:
        for mode in (_capi.TRACE_EXACT, _capi.TRACE_REFWALK):
            got, st = s.render(trace_mode=mode)
            assert np.array_equal(got, want), (f, mode)
            assert [st[k] for k in KEYS] == rays
