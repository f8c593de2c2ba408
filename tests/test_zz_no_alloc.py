"""GPU: mirt_render_device_noalloc (the entry point that allocates nothing: its working memory is a fixed per-GPU array)
against mirt_render_device, through the C ABI, byte for byte -- two calls in a row, the second with a smaller frame, so
that whatever the first call left in the shared array is there when the second runs."""
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
        self.rt.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def free_bytes(self):
        f, t = C.c_size_t(), C.c_size_t()
        assert self.rt.hipDeviceSynchronize() == 0
        assert self.rt.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    def alloc(self, n):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), n) == 0
        assert self.rt.hipMemset(p, 0xa5, n) == 0
        return p.value

    def to_host(self, ptr, n):
        out = np.zeros(n, np.uint8)
        assert self.rt.hipDeviceSynchronize() == 0
        assert self.rt.hipMemcpy(out.ctypes.data, ptr, n, 2) == 0  # hipMemcpyDeviceToHost
        return out


# which kernels a frame runs through: the single launch (k_frame), the same with nearly every traversal-stack push going to
# the overflow part of the scratch array (library built with two LDS entries), one launch per depth (its own layout of the
# array: lists, shadow answers, path depths), and the walk of the reference's tree
VARIANTS = {
    "frame": dict(lib=None, single="1", mode=_capi.TRACE_EXACT),
    "frame-overflow-stack": dict(lib="libmirt_smallstack.so", single="1", mode=_capi.TRACE_EXACT),
    "per-depth": dict(lib=None, single="0", mode=_capi.TRACE_EXACT),
    "per-depth-overflow-stack": dict(lib="libmirt_smallstack.so", single="0", mode=_capi.TRACE_EXACT),
    "refwalk": dict(lib=None, single="1", mode=_capi.TRACE_REFWALK),
}


@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("name", ["scene7", "scene6"])
def test_noalloc_render_equals_render_device_twice_in_a_row(name, variant, flat_scenes, goldens, monkeypatch):
    import os
    from raytracer_amd.device import DeviceScene, MirtError
    v = VARIANTS[variant]
    lib = None
    if v["lib"]:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", v["lib"])
        if not os.path.exists(path):
            pytest.skip(v["lib"] + " not built")
        lib = _capi.load_library(path)
    monkeypatch.setenv("MIRT_SINGLE_LAUNCH", v["single"])  # read when the scene is created
    hip = _Hip()
    g = goldens(name)
    with DeviceScene(flat_scenes(name), 0, lib=lib) as s:
        for (w, h), tiles in (((640, 400), {}), ((160, 120), {}), ((200, 136), dict(tile_w=32, tile_h=8, first_tile=1, tile_stride=3))):
            p = _capi.make_params(w, h, trace_mode=v["mode"], **tiles)
            nbytes = s.tile_buffer_bytes(p)
            a, b = hip.alloc(nbytes), hip.alloc(nbytes)
            s.render_device(p, a, nbytes, 0)            # the existing entry point (library-allocated workspace)
            # (the single launch of the entry point that allocates nothing is another kernel than mirt_render_device's -- hit
            # records instead of per-pixel arrays -- and the HIP runtime sets up a kernel's private memory at its first launch:
            # that is the runtime's doing, once per kernel, so the free memory is compared around the SECOND call)
            s.render_device_noalloc(p, b, nbytes, 0)
            before = hip.free_bytes()
            s.render_device_noalloc(p, b, nbytes, 0)    # the fixed array
            assert hip.free_bytes() == before, "mirt_render_device_noalloc changed the free device memory"
            want, got = hip.to_host(a, nbytes), hip.to_host(b, nbytes)
            assert want.any()
            assert np.array_equal(got, want), "%dx%d: %d bytes differ" % (w, h, int((got != want).sum()))
            if (w, h) == (160, 120):
                assert np.array_equal(got.reshape(h, w, 3), g["bgr_f15"])  # and both are the reference's image
            hip.rt.hipFree(a)
            hip.rt.hipFree(b)
        big = _capi.make_params(3840, 2160, trace_mode=v["mode"])
        n = 3840 * 2160 * 3
        if variant.startswith("frame"):
            # the single launch keeps its per-path data in hit records sized by the launch, not by the image: a frame of any
            # size fits the fixed array
            a, b = hip.alloc(n), hip.alloc(n)
            s.render_device(big, a, n, 0)
            s.render_device_noalloc(big, b, n, 0)
            before = hip.free_bytes()
            s.render_device_noalloc(big, b, n, 0)
            assert hip.free_bytes() == before
            want, got = hip.to_host(a, n), hip.to_host(b, n)
            assert want.any() and np.array_equal(got, want)
            hip.rt.hipFree(a)
            hip.rt.hipFree(b)
        else:
            # one launch per depth indexes everything by pixel slot: a frame beyond the fixed array is refused with a message,
            # nothing is launched
            buf = hip.alloc(64)
            with pytest.raises(MirtError, match="scratch array holds"):
                s.render_device_noalloc(big, buf, n, 0)
            hip.rt.hipFree(buf)
