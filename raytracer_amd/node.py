"""One frame on all GPUs of a node from ONE process (include/mirt_node.h, libmirt_node.so): the C++ counterpart of
bench.py's one-process-per-GPU path.  Thin ctypes calls; the tile split, the grouped ncclSend/ncclRecv step and the
reassembly are inside the library."""
import ctypes as C

import numpy as np

from . import _capi
from .device import MirtError


class NodeScene:
    def __init__(self, flat_scene, devices):
        self.lib = _capi.load_node_library()
        self.base = _capi.load_library()
        self.flat = flat_scene
        d = flat_scene.desc()
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        self._check(self.lib.mirt_node_create(C.byref(d), devs, len(devices), C.byref(h)))
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise MirtError("mirt error %d: %s" % (rc, (self.base.mirt_last_error() or b"?").decode()))

    def uses_rccl(self):
        n = C.c_uint32()
        return bool(self.lib.mirt_node_uses_rccl(self.h, C.byref(n))), int(n.value)

    def render(self, width=None, height=None, **kw):
        w, h = width or self.flat.width, height or self.flat.height
        p = _capi.make_params(w, h, **kw)
        out = np.zeros((h, w, 3), np.uint8)
        st = _capi.Stats()
        self._check(self.lib.mirt_node_render(self.h, C.byref(p), out.ctypes.data, C.byref(st)))
        return out, st.as_dict()

    def close(self):
        if getattr(self, "h", None):
            self.lib.mirt_node_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
