"""GPU: k_frame with its per-path data in hit records (MIRT_FRAME_RECORDS=1: 128 records per wave of the launch, taken when a
camera ray hits, given back when the path ends) instead of in arrays as long as the image.  Same bytes and ray counts as the
goldens and as the slot-indexed frames, with and without lane sharing, on the small-stack build, frame after frame."""
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
@pytest.mark.parametrize("name", ["scene4-specular", "scene5", "scene6", "scene7"])
def test_goldens_with_hit_records(which, name, flat_scenes, goldens, monkeypatch):
    from raytracer_amd.device import DeviceScene
    g = goldens(name)
    w, h = int(g["width"]), int(g["height"])
    monkeypatch.setenv("MIRT_FRAME_RECORDS", "1")
    with DeviceScene(flat_scenes(name), 0, lib=_lib(which)) as s:
        for share in ("1", "0"):
            monkeypatch.setenv("MIRT_SHARE", share)
            for features in (15, 7, 11, 3):
                img, st = s.render(w, h, features=features)
                assert np.array_equal(img, g["bgr_f%d" % features]), "MIRT_SHARE=%s features %d" % (share, features)
                assert [st["primary_rays"], st["shadow_rays"], st["reflection_rays"]] == g["rays_f%d" % features].tolist()
        for view in range(6):
            img, _ = s.render(w, h, debug=view)
            assert np.array_equal(img, g["bgr_debug%d" % view]), "view %d" % view
        img, _ = s.render(w, h, trace_mode=_capi.TRACE_FAST)    # (the FAST kernels have the option too)
        assert (img == g["bgr_f15"]).mean() > 0.99


@pytest.mark.parametrize("size,tiles", [((3840, 2160), {}), ((1000, 37), {}), ((1920, 1080), dict(tile_w=64, tile_h=16, first_tile=3, tile_stride=8))])
def test_big_frames_equal_the_slot_indexed_ones(size, tiles, flat_scenes, monkeypatch):
    from raytracer_amd.device import DeviceScene
    w, h = size
    with DeviceScene(flat_scenes("scene7"), 0) as s:
        monkeypatch.setenv("MIRT_FRAME_RECORDS", "0")
        want, wst = s.render(w, h, **tiles)
        monkeypatch.setenv("MIRT_FRAME_RECORDS", "1")
        for _ in range(2):
            got, st = s.render(w, h, **tiles)
            assert want.any() and np.array_equal(got, want), "%d bytes differ" % int((got != want).sum())
            assert {**st, "device_ms": 0} == {**wst, "device_ms": 0}


def test_a_frame_of_more_than_2_27_pixel_slots(flat_scenes):
    """From 2^27 slots on a slot number no longer fits beside the path's depth in one word and k_frame keeps its paths in hit
    records by itself.  16400x8200 (134.5 M slots, 403 MB of image) against the same frame rendered as two ranks' shares of half
    as many slots each, which take the slot-indexed kernel.  (exp/huge_frame.py does the same at 44000x44000 -- 1.9 G slots,
    byte offsets beyond 32 bits -- against sixteen shares: equal, 17.6 Grays/s.)"""
    from raytracer_amd.device import DeviceScene
    w, h = 16400, 8200
    assert w * h >= 1 << 27
    with DeviceScene(flat_scenes("scene7"), 0) as s:
        whole = np.zeros((h, w, 3), np.uint8)
        _, st = s.render(w, h, out=whole)
        acc = np.zeros((h, w, 3), np.uint8)
        tot = dict.fromkeys(("primary_rays", "shadow_rays", "reflection_rays"), 0)
        for r in range(2):
            part = np.zeros((h, w, 3), np.uint8)
            _, ps = s.render(w, h, out=part, tile_w=1024, tile_h=8, first_tile=r, tile_stride=2)
            acc |= part
            for k in tot:
                tot[k] += ps[k]
        assert whole.any() and np.array_equal(acc, whole)
        assert tot == {k: st[k] for k in tot}
