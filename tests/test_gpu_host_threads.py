"""GPU: several host threads inside the library at the same time, each with scene handles of its own.

include/mirt.h allows that (one handle must not be used from two threads at once; different handles may).  What the
threads share inside the library is the worker pool of the scene hand-over (one region at a time, a second caller works
alone), the per-thread error text, the current-device bookkeeping and the GPU itself; anything else shared would show up
here as a wrong byte, a wrong ray count or another thread's error message.  ctypes releases the interpreter lock for the
duration of every call, so the calls really overlap."""
import threading

import numpy as np
import pytest

from random_scenes import random_scene_text
from raytracer_amd import _capi
from raytracer_amd.device import DeviceScene, MirtError
from raytracer_amd.scene import FlatScene

pytestmark = pytest.mark.gpu

SCENES = ["scene1", "scene2", "scene3", "scene4-specular", "scene5", "scene6", "scene7"]
ROUNDS = 3


def test_host_threads_with_their_own_scenes(flat_scenes, goldens):
    flats = {n: flat_scenes(n) for n in SCENES}  # (session caches are filled before the threads start)
    golds = {n: goldens(n) for n in SCENES}
    errors = []
    start = threading.Barrier(len(SCENES))

    def work(k):
        try:
            start.wait()
            for r in range(ROUNDS):
                name = SCENES[(k + r) % len(SCENES)]
                g = golds[name]
                w, h = int(g["width"]), int(g["height"])
                with DeviceScene(flats[name], 0) as s:  # scene hand-over: the threads compete for the worker pool
                    img, st = s.render(w, h)
                    assert np.array_equal(img, g["bgr_f15"]), (name, "EXACT")
                    assert st["primary_rays"] == w * h
                    img2, st2 = s.render(w, h, trace_mode=_capi.TRACE_REFWALK)
                    assert np.array_equal(img2, g["bgr_f15"]), (name, "REFWALK")
                    assert [st2[x] for x in ("primary_rays", "shadow_rays", "reflection_rays")] == [st[x] for x in ("primary_rays", "shadow_rays", "reflection_rays")]
                    # a failing call in this thread: its message must be this thread's, whatever the others are doing
                    bad = 10_000 + k
                    with pytest.raises(MirtError) as e:
                        s.render(w, h, trace_mode=bad)
                    assert "trace_mode %d unknown" % bad in str(e.value), str(e.value)
                # the library's loader (text -> Partition tree, threaded) and a frame of what it made, all inside this thread
                fs = FlatScene.from_text(random_scene_text(100 * k + r))
                with DeviceScene(fs, 0) as s:
                    a, sa = s.render(fs.width * 3, fs.height * 3)
                    b, sb = s.render(fs.width * 3, fs.height * 3, trace_mode=_capi.TRACE_REFWALK)
                    assert np.array_equal(a, b) and sa["shadow_rays"] == sb["shadow_rays"], ("random scene", 100 * k + r)
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(e)))
            try:
                start.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(SCENES))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
        assert not t.is_alive(), "a thread is stuck inside the library"
    assert not errors, errors
