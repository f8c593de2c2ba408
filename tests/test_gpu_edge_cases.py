"""GPU: degenerate scenes and sizes, scene text -> library loader -> device, against the C oracle on the same flattened scene."""
import numpy as np
import pytest

from raytracer_amd import _capi
from raytracer_amd.scene import FlatScene

pytestmark = pytest.mark.gpu

HEAD = "camera 0 0 4 0 0 0 0 1 0 45\n"
TRI = "maxverts 4\nvertex -1 -1 0\nvertex 1 -1 0\nvertex 1 1 0\nvertex -1 1 0\n"
CASES = {
    "empty scene": "size 33 17\n" + HEAD,
    "lights but nothing to hit": "size 40 24\n" + HEAD + "point 0 2 2 1 1 1\ndirectional 0 0 1 0.5 0.5 0.5\n",
    "one triangle, no light": "size 64 64\n" + HEAD + "ambient 0.3 0.1 0.2\n" + TRI + "tri 0 1 2\n",
    "maxdepth 0": "size 50 30\nmaxdepth 0\n" + HEAD + "point 0 2 2 1 1 1\nspecular 1 1 1\ndiffuse 0.5 0.5 0.5\n" + TRI + "tri 0 1 2\ntri 0 2 3\n",
    "maxdepth -1": "size 16 16\nmaxdepth -1\n" + HEAD + "point 0 2 2 1 1 1\n" + TRI + "tri 0 1 2\n",
    "two facing mirrors, maxdepth 30": "size 48 48\nmaxdepth 30\ncamera 0 0 0.9 0 0 -1 0 1 0 60\npoint 0.3 0.4 0 1 1 1\nspecular 0.9 0.9 0.9\ndiffuse 0.2 0.1 0.1\n"
                                       "maxverts 8\nvertex -2 -2 -1\nvertex 2 -2 -1\nvertex 2 2 -1\nvertex -2 2 -1\nvertex -2 -2 1\nvertex 2 -2 1\nvertex 2 2 1\nvertex -2 2 1\n"
                                       "tri 0 1 2\ntri 0 2 3\ntri 4 6 5\ntri 4 7 6\n",
    "1x1 image": "size 1 1\n" + HEAD + "point 0 2 2 1 1 1\ndiffuse 0.7 0.7 0.7\n" + TRI + "tri 0 1 2\ntri 0 2 3\n",
    "sphere under a strongly anisotropic scale": "size 32 32\n" + HEAD + "point 0 2 2 1 1 1\ndiffuse 0.7 0.2 0.2\npushTransform\nscale 1 0.001 1\nsphere 0 0 0 1\npopTransform\nsphere 0 0 -1 0.5\n",
    "degenerate triangles": "size 32 32\n" + HEAD + "point 0 2 2 1 1 1\ndiffuse 0.7 0.2 0.2\n" + TRI + "tri 0 0 0\ntri 0 1 1\ntri 0 1 2\n",
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_degenerate_scene_matches_the_oracle(name):
    from oracle.bind import CPort
    from raytracer_amd.device import DeviceScene
    fs = FlatScene.from_text(CASES[name])
    want, wst = CPort(fs).render()
    with DeviceScene(fs, 0) as s:
        for mode in (_capi.TRACE_EXACT, _capi.TRACE_REFWALK):
            img, st = s.render(trace_mode=mode)
            assert np.array_equal(img, want), (name, mode)
            assert [st[k] for k in ("primary_rays", "shadow_rays", "reflection_rays")] == [wst[k] for k in ("primary_rays", "shadow_rays", "reflection_rays")]


def test_singular_sphere_transform_is_rejected():
    """glm::inverse of a singular matrix is inf/NaN; the reference then takes NaN distances for hits (SceneObjects.hpp:35-37)
    and draws whatever the order of NaN comparisons yields.  The library refuses the scene."""
    from raytracer_amd.device import DeviceScene, MirtError
    fs = FlatScene.from_text("size 32 32\n" + HEAD + "point 0 2 2 1 1 1\npushTransform\nscale 1 0 1\nsphere 0 0 0 1\npopTransform\n")
    with pytest.raises(MirtError, match="singular"):
        DeviceScene(fs, 0)


def test_device_powf_is_the_reference_builds_powf(goldens):
    """calculateSpecularLighting (main.cpp:262-264): the device's powf against the committed outputs of the C library the
    reference build links against (NaN results compared as NaN)."""
    import ctypes as C
    from raytracer_amd import _capi
    lib = _capi.load_library()
    g = goldens("kat_powf")
    x, y, want = (np.ascontiguousarray(g[k], np.float32) for k in ("x", "y", "want"))
    got = np.zeros_like(want)
    assert lib.mirt_test_powf(0, x.size, x.ctypes.data, y.ctypes.data, got.ctypes.data) == 0
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got[~nan].view(np.uint32), want[~nan].view(np.uint32))


def test_maxdepth_below_zero_at_a_size_that_takes_the_staged_copy():
    """With maxdepth < 0 a frame is a plain clear and no workspace exists (main.cpp:177,199: Color() for every pixel).  At
    1920x1080 the image is large enough for mirt_render's banded copy back, which must not look for a frame's notes then."""
    from raytracer_amd.device import DeviceScene
    fs = FlatScene.from_text("size 1920 1080\nmaxdepth -1\n" + HEAD + "point 0 2 2 1 1 1\n" + TRI + "tri 0 1 2\n")
    with DeviceScene(fs, 0) as s:
        img = np.full((1080, 1920, 3), 7, np.uint8)
        _, st = s.render(out=img)
        assert not img.any()
        assert st["primary_rays"] == 1920 * 1080 and st["shadow_rays"] == 0 and st["reflection_rays"] == 0


@pytest.mark.parametrize("size", [(65536, 1), (1, 65536), (65535, 7), (9, 16385)])
def test_images_one_pixel_wide_or_high_at_the_largest_size(size, flat_scenes):
    """main.cpp:137-163 loops over whatever width and height the scene names.  The library takes up to 65536 either way: a row or a
    column of that length, whole and as one rank's share of 64x8 tiles, against the C oracle; one pixel more is refused."""
    from oracle.bind import CPort
    from raytracer_amd.device import DeviceScene, MirtError
    w, h = size
    fs = flat_scenes("scene6")
    want, wst = CPort(fs).render(w, h)
    with DeviceScene(fs, 0) as s:
        for mode in (_capi.TRACE_EXACT, _capi.TRACE_REFWALK):
            img, st = s.render(w, h, trace_mode=mode)
            assert np.array_equal(img, want), (size, mode)
            assert [st[k] for k in ("primary_rays", "shadow_rays", "reflection_rays")] == [wst[k] for k in ("primary_rays", "shadow_rays", "reflection_rays")]
        part, _ = s.render(w, h, tile_w=64, tile_h=8, first_tile=1, tile_stride=3)
        ty, tx = np.meshgrid(np.arange(h) // 8, np.arange(w) // 64, indexing="ij")
        mine = (ty * ((w + 63) // 64) + tx) % 3 == 1
        assert np.array_equal(part[mine], want[mine]) and not part[~mine].any()
        with pytest.raises(MirtError, match="not supported"):
            s.render(65537, 1)
