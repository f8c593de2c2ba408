"""GPU: mirt_render's copy back to the caller's (pageable) buffer.

Frames of 4 MB and more come back through page-locked staging in bands; bands in which the finished image holds nothing but
black are not fetched at all (a small launch behind the frame notes them), the caller's bytes there are zeroed instead.
Whatever the buffer held before, afterwards it must hold the image, byte for byte, with and without that shortcut and
with the plain copy."""
import numpy as np
import pytest

from raytracer_amd import _capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size", [(3840, 2160), (1920, 1080), (2048, 683)])
def test_copy_back_variants_fill_the_whole_buffer(size, flat_scenes, monkeypatch):
    from raytracer_amd.device import DeviceScene
    w, h = size
    with DeviceScene(flat_scenes("scene7"), 0) as s:
        got = {}
        for name, env in (("sparse", {"MIRT_COPY_SPARSE": "1"}), ("all bands", {"MIRT_COPY_SPARSE": "0"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            out = np.full((h, w, 3), 0xA5, np.uint8)          # what a stale buffer might hold
            img, _ = s.render(w, h, out=out)
            assert img is out
            got[name] = out.copy()
        assert np.array_equal(got["sparse"], got["all bands"])
        assert (got["sparse"].reshape(-1, w * 3).max(axis=1) == 0).sum() > h // 8, "no black rows: the shortcut was not exercised"
        # the device image itself, through the asynchronous entry point
        import ctypes as C
        rt = C.CDLL("libamdhip64.so")
        rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rt.hipFree.argtypes = [C.c_void_p]
        d = C.c_void_p()
        assert rt.hipMalloc(C.byref(d), w * h * 3) == 0
        s.render_device(_capi.make_params(w, h), d.value, w * h * 3, 0)
        assert rt.hipDeviceSynchronize() == 0
        dev = np.zeros((h, w, 3), np.uint8)
        assert rt.hipMemcpy(dev.ctypes.data, d, w * h * 3, 2) == 0
        rt.hipFree(d)
        assert np.array_equal(got["sparse"], dev)


def test_debug_view_that_paints_every_hit_comes_back_whole(flat_scenes, goldens, monkeypatch):
    """A frame of another kind (one launch per depth is forced): the notes are taken from the finished image, whoever wrote it."""
    from raytracer_amd.device import DeviceScene
    w, h = 2560, 1440
    with DeviceScene(flat_scenes("scene6"), 0) as s:
        monkeypatch.setenv("MIRT_COPY_SPARSE", "0")
        want, _ = s.render(w, h, trace_mode=_capi.TRACE_REFWALK, out=np.full((h, w, 3), 0x11, np.uint8))
        monkeypatch.setenv("MIRT_COPY_SPARSE", "1")
        got, _ = s.render(w, h, trace_mode=_capi.TRACE_REFWALK, out=np.full((h, w, 3), 0xEE, np.uint8))
        assert np.array_equal(got, want)
        fast, _ = s.render(w, h, out=np.full((h, w, 3), 0x77, np.uint8))
        assert np.array_equal(fast, want)
