"""GPU: k_frame with MIRT_CLEAR_IN_KERNEL=1 -- no clear of the output or of the counters is queued in front of a frame: every
pixel slot writes its own pixel (black blocks of 8x8 as words where rows are word-aligned, byte by byte otherwise) and a
frame clears the counter block of the stream's next frame.  Same bytes, same ray counts, frame after frame on one stream,
into buffers that held something else before."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import _capi

pytestmark = pytest.mark.gpu


class _Hip:
    def __init__(self):
        self.rt = C.CDLL("libamdhip64.so")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]

    def alloc(self, n, fill):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), n) == 0
        assert self.rt.hipMemset(p, fill, n) == 0
        return p.value

    def to_host(self, ptr, n):
        out = np.zeros(n, np.uint8)
        assert self.rt.hipDeviceSynchronize() == 0
        assert self.rt.hipMemcpy(out.ctypes.data, ptr, n, 2) == 0
        return out


@pytest.mark.parametrize("name", ["scene7", "scene6", "scene5"])
def test_frames_without_queued_clears_are_the_same_frames(name, flat_scenes, goldens, monkeypatch):
    from raytracer_amd.device import DeviceScene
    g = goldens(name)
    w, h = int(g["width"]), int(g["height"])
    with DeviceScene(flat_scenes(name), 0) as s:
        monkeypatch.setenv("MIRT_CLEAR_IN_KERNEL", "1")
        for features in (15, 7, 15, 3):                  # several frames in a row on the scene's own stream: counters hand over
            img = np.full((h, w, 3), 0x5a, np.uint8)
            _, st = s.render(w, h, features=features, out=img)
            assert np.array_equal(img, g["bgr_f%d" % features])
            assert [st["primary_rays"], st["shadow_rays"], st["reflection_rays"]] == g["rays_f%d" % features].tolist()
        img = np.full((h, w, 3), 0x5a, np.uint8)
        s.render(w, h, debug=_capi.DEBUG_PRIMARY_MAP, out=img)
        assert np.array_equal(img, g["bgr_debug%d" % _capi.DEBUG_PRIMARY_MAP])
        # sizes whose rows are not word-aligned, and a frame after a switch of the clearing mode
        odd = np.full((97, 161, 3), 0x5a, np.uint8)
        s.render(161, 97, out=odd)
        monkeypatch.setenv("MIRT_CLEAR_IN_KERNEL", "0")
        want, wst = s.render(161, 97)
        assert np.array_equal(odd, want)
        monkeypatch.setenv("MIRT_CLEAR_IN_KERNEL", "1")
        again, ast = s.render(161, 97)
        assert np.array_equal(again, want) and {**ast, "device_ms": 0} == {**wst, "device_ms": 0}


def test_packed_tiles_with_padding_and_big_frames(flat_scenes, monkeypatch):
    """Packed tile buffers (a rank's share) hold padding pixels at the image's edges: they are written too.  A 4K frame goes
    through the strided batch order; its black blocks are written as words."""
    from raytracer_amd.device import DeviceScene
    hip = _Hip()
    with DeviceScene(flat_scenes("scene7"), 0) as s:
        for (w, h), tiles in (((200, 136), dict(tile_w=32, tile_h=8, first_tile=1, tile_stride=3)), ((1000, 37), dict(tile_w=64, tile_h=16, first_tile=0, tile_stride=2)),
                              ((3840, 2160), {}), ((3838, 2160), {})):
            p = _capi.make_params(w, h, **tiles)
            n = s.tile_buffer_bytes(p)
            a, b = hip.alloc(n, 0x11), hip.alloc(n, 0xee)
            monkeypatch.setenv("MIRT_CLEAR_IN_KERNEL", "0")
            s.render_device(p, a, n, 0)
            want = hip.to_host(a, n)
            monkeypatch.setenv("MIRT_CLEAR_IN_KERNEL", "1")
            s.render_device(p, b, n, 0)
            s.render_device(p, b, n, 0)
            got = hip.to_host(b, n)
            assert want.any() and np.array_equal(got, want), "%dx%d %s: %d bytes differ" % (w, h, tiles, int((got != want).sum()))
            hip.rt.hipFree(a)
            hip.rt.hipFree(b)
