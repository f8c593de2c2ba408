"""GPU: frames in flight on several HIP streams -- the mode bench.py times -- must be the frames a single stream renders.

Every stream owns a workspace inside the library (hit records, lists, counters and the overflow part of the traversal
stack); launches of different streams run at the same time with identical grids, so anything shared between them would
show up here as a wrong byte.  The same is run on a build that keeps only two traversal-stack entries per lane in LDS
(libmirt_smallstack.so), where nearly every push of a walk goes through the global overflow area."""
import ctypes as C
import os

import numpy as np
import pytest

from raytracer_amd import _capi

pytestmark = pytest.mark.gpu

N_STREAMS, FRAMES_PER_STREAM = 8, 3
HERE = os.path.dirname(os.path.abspath(_capi.__file__))


class _Hip:
    def __init__(self):
        self.rt = C.CDLL("libamdhip64.so")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.rt.hipStreamDestroy.argtypes = [C.c_void_p]

    def alloc(self, n):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), n) == 0
        assert self.rt.hipMemset(p, 0x5a, n) == 0
        return p.value

    def stream(self):
        s = C.c_void_p()
        assert self.rt.hipStreamCreate(C.byref(s)) == 0
        return s.value

    def to_host(self, ptr, n):
        out = np.zeros(n, np.uint8)
        assert self.rt.hipMemcpy(out.ctypes.data, ptr, n, 2) == 0  # hipMemcpyDeviceToHost
        return out


def _lib(which):
    if which == "default":
        return _capi.load_library()
    path = os.path.join(HERE, "libmirt_%s.so" % which)
    if not os.path.exists(path):
        pytest.fail("%s not built (make -C raytracer_amd/csrc all)" % path)
    return _capi.load_library(path)


@pytest.mark.parametrize("which", ["default", "smallstack"])
@pytest.mark.parametrize("size", ["golden", "1080p"])
def test_frames_in_flight_on_eight_streams_are_byte_exact(which, size, flat_scenes, goldens):
    from raytracer_amd.device import DeviceScene
    hip = _Hip()
    lib = _lib(which)
    g = goldens("scene7")
    w, h = (int(g["width"]), int(g["height"])) if size == "golden" else (1920, 1080)
    nbytes = w * h * 3
    with DeviceScene(flat_scenes("scene7"), 0, lib=lib) as s:
        need = C.c_uint64()
        assert lib.mirt_scene_query(s.h, 126, C.byref(need)) == 0  # most entries a walk of this hierarchy can hold
        if which == "smallstack":
            assert need.value > 2, "the overflow area would not be used"
        want, st0 = s.render(w, h)  # one stream, synchronous
        if size == "golden":
            assert np.array_equal(want, g["bgr_f15"])
        else:  # the tree handed over, walked as it is
            ref, _ = s.render(w, h, trace_mode=_capi.TRACE_REFWALK)
            assert np.array_equal(want, ref)
        streams = [hip.stream() for _ in range(N_STREAMS)]
        bufs = [[hip.alloc(nbytes) for _ in range(FRAMES_PER_STREAM)] for _ in range(N_STREAMS)]
        p = _capi.make_params(w, h)
        if which == "default":
            s.set_frames_in_flight(N_STREAMS)    # the caller's hint: every frame is launched as one of eight, the first ones too
        for f in range(FRAMES_PER_STREAM):       # round-robin: every stream has a frame queued before any gets its next
            for k, st in enumerate(streams):
                s.render_device(p, bufs[k][f], nbytes, st)
        assert hip.rt.hipDeviceSynchronize() == 0
        for k, st in enumerate(streams):
            stats = s.stats(st)
            assert [stats[x] for x in ("primary_rays", "shadow_rays", "reflection_rays")] == [st0[x] for x in ("primary_rays", "shadow_rays", "reflection_rays")]
            for f in range(FRAMES_PER_STREAM):
                got = hip.to_host(bufs[k][f], nbytes).reshape(h, w, 3)
                assert np.array_equal(got, want), "stream %d frame %d: %d bytes differ" % (k, f, int((got != want).sum()))
        s.set_frames_in_flight(0)
        # a stream's workspace can be handed back, and the stream used again afterwards
        s.release_stream(streams[0])
        s.render_device(p, bufs[0][0], nbytes, streams[0])
        assert hip.rt.hipDeviceSynchronize() == 0
        assert np.array_equal(hip.to_host(bufs[0][0], nbytes).reshape(h, w, 3), want)
        for st in streams:
            s.release_stream(st)
            hip.rt.hipStreamDestroy(st)
        for row in bufs:
            for b in row:
                hip.rt.hipFree(b)


@pytest.mark.parametrize("name", ["scene5", "scene6", "scene7"])
def test_small_stack_build_is_exact_on_the_goldens(name, flat_scenes, goldens):
    """The overflow instantiations of the walk (trace kernels and both parity hooks) against the reference's outputs."""
    from conftest import bits
    from raytracer_amd.device import DeviceScene
    lib = _lib("smallstack")
    g = goldens(name)
    w, h = int(g["width"]), int(g["height"])
    with DeviceScene(flat_scenes(name), 0, lib=lib) as s:
        for features in (15, 7):
            img, st = s.render(w, h, features=features)
            assert np.array_equal(img, g["bgr_f%d" % features])
            assert [st["primary_rays"], st["shadow_rays"], st["reflection_rays"]] == g["rays_f%d" % features].tolist()
        ids, dist = s.trace_primary_ids(w, h)
        assert np.array_equal(ids, g["prim_id"]) and np.array_equal(bits(dist), bits(g["dist"]))
        if g["sec_origins"].shape[0]:
            ids, dist, _ = s.trace_rays(g["sec_origins"], g["sec_dirs"])
            assert np.array_equal(ids, g["sec_prim_id"]) and np.array_equal(bits(dist), bits(g["sec_dist"]))
