"""GPU: a single-launch frame visits only the 8x8 pixel blocks that can see the scene: those under the screen rectangles of the
boxes of a cut through the hierarchy (camera rays through the others miss every box, and the image was cleared).  With the
list on and off the bytes and the ray counts must be the same -- for cameras that see all of the scene, a corner of it, none
of it, for cameras inside it or with parts of it behind them (then every block is visited), for a rank's packed tiles and
for a scene that came back from its cache file."""
import numpy as np
import pytest

from raytracer_amd.scene import FlatScene

pytestmark = pytest.mark.gpu

OBJ = ("point 3 4 5 1 1 1\ndiffuse 0.6 0.3 0.2\nspecular 0.4 0.4 0.4\nshininess 20\nmaxverts 4\nvertex -1 -1 0\nvertex 1 -1 0\nvertex 1 1 0\nvertex -1 1 0\n"
       "tri 0 1 2\ntri 0 2 3\nsphere 0.3 0.2 0.5 0.4\npushTransform\ntranslate 1.5 0.5 -0.5\nscale 0.3 0.6 0.3\nsphere 0 0 0 1\npopTransform\n")
CAMERAS = {
    "all of it, small in the frame": "camera 0 0 12 0 0 0 0 1 0 40",
    "a corner of the frame": "camera 0 0 12 4.5 3.2 0 0 1 0 40",
    "half of it cut by the frame's edge": "camera 0 0 6 2.4 0 0 0 1 0 40",
    "nothing on the screen": "camera 0 0 12 40 0 0 0 1 0 20",
    "looking away": "camera 0 0 12 0 0 24 0 1 0 40",
    "inside the bounds": "camera 0.2 0.1 0.2 1 0 -1 0 1 0 70",
    "close, part of the scene behind the eye plane": "camera 1.2 0.4 0.8 0 0 0 0 1 0 90",
    "tilted": "camera 5 7 9 0 0 0 1 1 0 35",
}


@pytest.mark.parametrize("size", [(640, 400), (333, 211)])
@pytest.mark.parametrize("name", sorted(CAMERAS))
def test_frame_with_and_without_the_block_list(name, size, monkeypatch):
    from raytracer_amd.device import DeviceScene
    w, h = size
    fs = FlatScene.from_text("size %d %d\nmaxdepth 3\n%s\n%s" % (w, h, CAMERAS[name], OBJ))
    with DeviceScene(fs, 0) as s:
        monkeypatch.setenv("MIRT_BLOCK_LIST", "0")
        want, wst = s.render()
        monkeypatch.setenv("MIRT_BLOCK_LIST", "1")
        for _ in range(2):
            img = np.full((h, w, 3), 0x77, np.uint8)
            _, st = s.render(out=img)
            assert np.array_equal(img, want), "%s: %d bytes differ" % (name, int((img != want).sum()))
            assert {**st, "device_ms": 0} == {**wst, "device_ms": 0}
    if name in ("nothing on the screen", "looking away"):
        assert not want.any()
    else:
        assert want.any()


def test_against_the_oracle_with_the_scene_in_a_corner():
    from oracle.bind import CPort
    from raytracer_amd.device import DeviceScene
    fs = FlatScene.from_text("size 320 200\nmaxdepth 2\n%s\n%s" % (CAMERAS["a corner of the frame"], OBJ))
    want, wst = CPort(fs).render()
    with DeviceScene(fs, 0) as s:
        img, st = s.render()
    assert np.array_equal(img, want)
    assert [st[k] for k in ("primary_rays", "shadow_rays", "reflection_rays")] == [wst[k] for k in ("primary_rays", "shadow_rays", "reflection_rays")]


@pytest.mark.parametrize("tiles", [dict(tile_w=64, tile_h=16, first_tile=1, tile_stride=3), dict(tile_w=32, tile_h=32, first_tile=0, tile_stride=2)])
def test_a_ranks_packed_tiles(tiles, flat_scenes, monkeypatch):
    from raytracer_amd.device import DeviceScene
    with DeviceScene(flat_scenes("scene7"), 0) as s:
        for w, h in ((1000, 700), (333, 211)):
            monkeypatch.setenv("MIRT_BLOCK_LIST", "0")
            want, wst = s.render(w, h, **tiles)
            monkeypatch.setenv("MIRT_BLOCK_LIST", "1")
            got, st = s.render(w, h, **tiles)
            assert want.any() and np.array_equal(got, want)
            assert {**st, "device_ms": 0} == {**wst, "device_ms": 0}


def test_scene_from_its_cache_file(flat_scenes, goldens, tmp_path, monkeypatch):
    from raytracer_amd.device import DeviceScene
    path = str(tmp_path / "scene7.mirt")
    with DeviceScene(flat_scenes("scene7"), 0) as s:
        s.save(path)
        want, wst = s.render(1280, 720)
    cached, _ = DeviceScene.load(path, 0)
    with cached as s:
        got, st = s.render(1280, 720)
        assert np.array_equal(got, want) and {**st, "device_ms": 0} == {**wst, "device_ms": 0}
        monkeypatch.setenv("MIRT_BLOCK_LIST", "0")
        assert np.array_equal(s.render(1280, 720)[0], want)
