"""PNG output of a rendered frame (Renderer::createImage, Renderer.cpp:12-15) through the library's encoder."""
import numpy as np

from . import _capi


def write_png(path, bgr, threads=0):
    """bgr: uint8 [h, w, 3] as mirt_render fills it (B,G,R per pixel, top row first)."""
    a = np.ascontiguousarray(bgr, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected an [h, w, 3] uint8 image")
    lib = _capi.load_library()
    if lib.mirt_write_png(str(path).encode(), a.ctypes.data, a.shape[1], a.shape[0], threads) != 0:
        raise OSError(lib.mirt_last_error().decode())
