"""GPU: mirt_render_png -- frame, PNG rows made on the device (Up-filtered), bands deflated as they arrive -- must write a
file that decodes to exactly the bytes mirt_render delivers (Renderer::createImage's input, Renderer.cpp:12-15)."""
import numpy as np
import pytest

from raytracer_amd import _capi
from raytracer_amd.scene import FlatScene

pytestmark = pytest.mark.gpu


def _decode(path):
    from PIL import Image
    im = Image.open(path)
    assert im.mode == "RGB"
    return np.asarray(im)[..., ::-1]   # file order is R,G,B; the library's buffers are B,G,R (main.cpp:154-156)


@pytest.mark.parametrize("name,size", [("scene6", None), ("scene7", None), ("scene5", None), ("scene6", (161, 97)), ("scene7", (1000, 700)), ("scene4-specular", (37, 1))])
def test_png_file_decodes_to_the_rendered_bytes(name, size, flat_scenes, goldens, tmp_path):
    from raytracer_amd.device import DeviceScene
    g = goldens(name)
    w, h = size or (int(g["width"]), int(g["height"]))
    path = tmp_path / "out.png"
    with DeviceScene(flat_scenes(name), 0) as s:
        st = s.render_png(path, w, h)
        want, wst = s.render(w, h)
        got = _decode(path)
        assert got.shape == (h, w, 3) and np.array_equal(got, want)
        assert [st[k] for k in ("primary_rays", "shadow_rays", "reflection_rays")] == [wst[k] for k in ("primary_rays", "shadow_rays", "reflection_rays")]
        if size is None:
            assert np.array_equal(got, g["bgr_f15"])       # and both are the reference's image
        # a second file from the same scene (buffers and cached streams of the first call are reused), another view
        st = s.render_png(path, w, h, debug=_capi.DEBUG_NORMAL_MAP)
        assert np.array_equal(_decode(path), s.render(w, h, debug=_capi.DEBUG_NORMAL_MAP)[0])


def test_png_of_black_and_of_large_frames(tmp_path):
    """Frames whose stripes are all black take their deflate streams from the cache; a 4K frame has 68 stripes in 12 bands."""
    from raytracer_amd import workloads
    from raytracer_amd.device import DeviceScene
    path = tmp_path / "out.png"
    empty = FlatScene.from_text("size 640 480\ncamera 0 0 4 0 0 0 0 1 0 45\n")
    with DeviceScene(empty, 0) as s:
        s.render_png(path)
        got = _decode(path)
        assert got.shape == (480, 640, 3) and not got.any()
    with DeviceScene(FlatScene.from_text(workloads.dragon(3840, 2160)), 0) as s:
        s.render_png(path)
        assert np.array_equal(_decode(path), s.render()[0])
        from raytracer_amd.device import _check
        with pytest.raises(Exception, match="whole images"):
            p = _capi.make_params(640, 480, tile_w=32, tile_h=8, first_tile=1, tile_stride=3)
            _check(s.lib, s.lib.mirt_render_png(s.h, p, str(path).encode(), None))
        with pytest.raises(Exception, match="cannot open"):
            s.render_png("/nonexistent_dir/x.png", 64, 64)
