"""Host-side mirror of the seam the library replaces: a loaded scene goes in, the B,G,R byte buffer that
Renderer::createImage takes (reference RayTracer/Renderer.h:9) comes out.  Thin ctypes calls into libmirt.so;
there is no CPU fallback -- a missing library or device raises."""
import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import (DEBUG_NONE, F_ALL, TRACE_EXACT, TRACE_FAST, TRACE_REFWALK, RenderParams, Stats, make_params)  # noqa: F401


class MirtError(RuntimeError):
    pass


def _check(lib, rc):
    if rc != 0:
        msg = lib.mirt_last_error()
        raise MirtError("mirt error %d: %s" % (rc, msg.decode() if msg else "?"))


class DeviceScene:
    """A scene resident in one GPU's HBM (mirt_scene handle).  The equivalent of the reference's
    `Scene scene(path)` object once loading has finished (RayTracer/main.cpp:107)."""

    def __init__(self, flat_scene, device=0, lib=None):
        self.lib = lib or _capi.load_library()  # lib: another build of the same ABI (tests: small-stack and statistics builds)
        self.flat = flat_scene
        self.device = device
        d = flat_scene.desc()
        h = C.c_void_p()
        _check(self.lib, self.lib.mirt_scene_create(C.byref(d), device, C.byref(h)))
        self.h = h

    # ---- scene cache (mirt_scene_save / mirt_scene_load) -----------------------------------------------------------------------
    def save(self, path, user=(0, 0, 0, 0)):
        u = (C.c_uint32 * 4)(*user)
        _check(self.lib, self.lib.mirt_scene_save(self.h, os.fsencode(path), u))

    @classmethod
    def load(cls, path, device=0, lib=None):
        """A scene from its cache file: no scene text, no tree, no mirt_scene_create.  Returns (scene, user words)."""
        self = cls.__new__(cls)
        self.lib = lib or _capi.load_library()
        self.flat = None
        self.device = device
        h = C.c_void_p()
        u = (C.c_uint32 * 4)()
        _check(self.lib, self.lib.mirt_scene_load(os.fsencode(path), device, C.byref(h), u))
        self.h = h
        return self, tuple(int(x) for x in u)

    def close(self):
        if getattr(self, "h", None):
            self.lib.mirt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- createRender's pixel loop (main.cpp:137-157), synchronous, host buffer out ---------------------------------
    def render(self, width=None, height=None, features=F_ALL, debug=DEBUG_NONE, trace_mode=TRACE_EXACT, out=None, **tiles):
        w = width or self.flat.width
        h = height or self.flat.height
        p = make_params(w, h, features, debug, trace_mode, **tiles)
        if out is None:
            out = np.zeros((h, w, 3), np.uint8)
        assert out.dtype == np.uint8 and out.size == w * h * 3 and out.flags.c_contiguous
        st = Stats()
        _check(self.lib, self.lib.mirt_render(self.h, C.byref(p), out.ctypes.data, C.byref(st)))
        return out, st.as_dict()

    def render_png(self, path, width=None, height=None, features=F_ALL, debug=DEBUG_NONE, trace_mode=TRACE_EXACT):
        """Frame and PNG file in one call (mirt_render_png): the PNG rows are made on the device and deflated as they arrive."""
        p = make_params(width or self.flat.width, height or self.flat.height, features, debug, trace_mode)
        st = Stats()
        _check(self.lib, self.lib.mirt_render_png(self.h, C.byref(p), str(path).encode(), C.byref(st)))
        return st.as_dict()

    # ---- asynchronous, device buffer out (torch tensor or raw pointer) ------------------------------------------------
    def tile_buffer_bytes(self, params):
        return int(self.lib.mirt_tile_buffer_bytes(C.byref(params)))

    def render_device(self, params, d_out_ptr, d_out_bytes, stream_ptr=0):
        _check(self.lib, self.lib.mirt_render_device(self.h, C.byref(params), C.c_void_p(d_out_ptr), d_out_bytes, C.c_void_p(stream_ptr)))

    def render_device_noalloc(self, params, d_out_ptr, d_out_bytes, stream_ptr=0):
        _check(self.lib, self.lib.mirt_render_device_noalloc(self.h, C.byref(params), C.c_void_p(d_out_ptr), d_out_bytes, C.c_void_p(stream_ptr)))

    def release_stream(self, stream_ptr):
        _check(self.lib, self.lib.mirt_release_stream(self.h, C.c_void_p(stream_ptr)))

    def set_frames_in_flight(self, n):
        """Hint: the caller keeps n frames of this scene in flight (one per stream); 0 or 1 = sized by what is found in flight."""
        _check(self.lib, self.lib.mirt_set_frames_in_flight(self.h, int(n)))

    def stats(self, stream_ptr=0):
        st = Stats()
        _check(self.lib, self.lib.mirt_get_stats(self.h, C.c_void_p(stream_ptr), C.byref(st)))
        return st.as_dict()

    def query(self):
        names = ["hierarchy_depth", "hierarchy_nodes", "tree_nested", "membership_entries", "device_bytes", "tree_wide_depth"]
        out = {}
        for i, n in enumerate(names):
            v = C.c_uint64()
            _check(self.lib, self.lib.mirt_scene_query(self.h, i, C.byref(v)))
            out[n] = int(v.value)
        return out

    def set_profiling(self, on=True):
        _check(self.lib, self.lib.mirt_set_profiling(self.h, 1 if on else 0))

    def clock_probe_start(self, microseconds):
        """One wave that reads the shader-cycle counter `microseconds` apart, on a stream of its own; returns at once."""
        _check(self.lib, self.lib.mirt_clock_probe_start(self.h, int(microseconds)))

    def clock_probe_read(self):
        """Shader clock in Hz over the probe's interval (waits for the probe)."""
        hz = C.c_double()
        _check(self.lib, self.lib.mirt_clock_probe_read(self.h, C.byref(hz)))
        return float(hz.value)

    def launch_times(self, stream_ptr=0):
        """[(kernel name, ms)] of the most recent frame, in issue order (needs set_profiling(True))."""
        ms = (C.c_float * 96)()
        kinds = (C.c_int32 * 96)()
        n = C.c_uint32()
        _check(self.lib, self.lib.mirt_get_launch_times(self.h, C.c_void_p(stream_ptr), ms, kinds, 96, C.byref(n)))
        return [(_capi.KERNEL_NAMES[kinds[i]], float(ms[i])) for i in range(n.value)]

    def assemble_tiles(self, params, n_ranks, d_tiles_ptr, per_rank_bytes, d_bgr_ptr, stream_ptr=0):
        _check(self.lib, self.lib.mirt_assemble_tiles(self.device, C.byref(params), n_ranks, C.c_void_p(d_tiles_ptr), per_rank_bytes,
                                                     C.c_void_p(d_bgr_ptr), C.c_void_p(stream_ptr)))

    # ---- Scene::findClosestIntersection for a batch of rays (Scene.h:37) ----------------------------------------------
    def trace_rays(self, origins, dirs, trace_mode=TRACE_EXACT, want_normals=True):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = o.shape[0]
        ids = np.zeros(n, np.int32)
        dist = np.zeros(n, np.float32)
        nrm = np.zeros((n, 3), np.float32) if want_normals else None
        _check(self.lib, self.lib.mirt_trace_rays(self.h, trace_mode, n, o.ctypes.data, d.ctypes.data, ids.ctypes.data, dist.ctypes.data,
                                                 nrm.ctypes.data if want_normals else None))
        return ids, dist, nrm

    def trace_primary_ids(self, width=None, height=None, trace_mode=TRACE_EXACT):
        w = width or self.flat.width
        h = height or self.flat.height
        ids = np.zeros(w * h, np.int32)
        dist = np.zeros(w * h, np.float32)
        _check(self.lib, self.lib.mirt_trace_primary_ids(self.h, trace_mode, w, h, ids.ctypes.data, dist.ctypes.data))
        return ids, dist
