"""GPU: k_frame with and without the lanes of a wave helping each other through long walks (walk_wave_share).

The library picks the helping kernel by the size of the hierarchy; MIRT_SHARE=1/0 forces one or the other, so the small
reference scenes go through the helping kernel too, and the Stanford dragon through the plain one.  Both must render the
reference's images byte for byte, also on the build whose traversal stack lives almost entirely in the overflow area."""
import os

import numpy as np
import pytest

from raytracer_amd import _capi

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(_capi.__file__))


def _lib(which):
    if which == "default":
        return _capi.load_library()
    path = os.path.join(HERE, "libmirt_%s.so" % which)
    if not os.path.exists(path):
        pytest.fail("%s not built (make -C raytracer_amd/csrc all)" % path)
    return _capi.load_library(path)


@pytest.mark.parametrize("which", ["default", "smallstack"])
@pytest.mark.parametrize("name", ["scene5", "scene6", "scene7"])
def test_goldens_with_and_without_lane_sharing(which, name, flat_scenes, goldens, monkeypatch):
    from raytracer_amd.device import DeviceScene
    g = goldens(name)
    w, h = int(g["width"]), int(g["height"])
    with DeviceScene(flat_scenes(name), 0, lib=_lib(which)) as s:
        for share in ("1", "0"):
            monkeypatch.setenv("MIRT_SHARE", share)
            for features in (15, 7, 11, 3):
                img, st = s.render(w, h, features=features)
                assert np.array_equal(img, g["bgr_f%d" % features]), "MIRT_SHARE=%s features %d" % (share, features)
                assert [st["primary_rays"], st["shadow_rays"], st["reflection_rays"]] == g["rays_f%d" % features].tolist()
            img, _ = s.render(w, h, debug=_capi.DEBUG_SHADOW_MAP)   # shadow queries that ask for the NEAREST occluder
            assert np.array_equal(img, g["bgr_debug%d" % _capi.DEBUG_SHADOW_MAP]), "MIRT_SHARE=%s shadow map" % share


@pytest.mark.parametrize("size", [(1920, 1080), (3840, 2160), (1000, 37)])
def test_big_frames_with_lane_sharing_equal_the_walk_of_the_reference_tree(size, flat_scenes, monkeypatch):
    """Frames far bigger than the goldens: the helping kernel against the plain one and against the exhaustive walk of the tree
    that was handed over (1080p only: that walk is slow)."""
    from raytracer_amd.device import DeviceScene
    w, h = size
    with DeviceScene(flat_scenes("scene7"), 0) as s:
        monkeypatch.setenv("MIRT_SHARE", "1")
        shared, st1 = s.render(w, h)
        monkeypatch.setenv("MIRT_SHARE", "0")
        plain, st0 = s.render(w, h)
        assert np.array_equal(shared, plain), "%d bytes differ" % int((shared != plain).sum())
        assert st1 == {**st0, "device_ms": st1["device_ms"]}
        if w * h <= 1920 * 1080:
            ref, _ = s.render(w, h, trace_mode=_capi.TRACE_REFWALK)
            assert np.array_equal(shared, ref)


def test_cornell_and_ellipsoids_at_full_size_with_lane_sharing_forced(goldens, monkeypatch):
    """Scenes the library would not pick the helping kernel for (few primitives, spheres and ellipsoids among them),
    against the reference's own images at the benchmark sizes."""
    from raytracer_amd import workloads
    from raytracer_amd.device import DeviceScene
    from raytracer_amd.scene import FlatScene
    monkeypatch.setenv("MIRT_SHARE", "1")
    g = goldens("c1_cornell")
    with DeviceScene(FlatScene.from_text(workloads.cornell()), 0) as s:
        img, st = s.render(640, 480)
        assert np.array_equal(img, g["bgr"])
        assert [st["primary_rays"], st["shadow_rays"], st["reflection_rays"]] == g["rays"].tolist()
    g = goldens("c2_spheres")
    with DeviceScene(FlatScene.from_text(workloads.spheres()), 0) as s:
        img, st = s.render(1920, 1080)
        assert np.array_equal(img, g["bgr"])
        assert [st["primary_rays"], st["shadow_rays"], st["reflection_rays"]] == g["rays"].tolist()
