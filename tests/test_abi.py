"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute calls)."""
import ctypes as C
import os
import re

import numpy as np

from conftest import ROOT
from raytracer_amd import _capi


def test_library_exports_every_declared_symbol():
    lib = _capi.load_library()
    header = "".join(open(os.path.join(ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(ROOT, "include"))) if h.endswith(".h"))
    declared = set(re.findall(r"\b(mirt_[a-z_]+)\s*\(", header))
    bound = {name for name, _, _ in _capi.SYMBOLS}
    node = {name for name, _, _ in _capi.NODE_SYMBOLS}   # include/mirt_node.h lives in libmirt_node.so (libmirt.so + RCCL)
    assert declared == bound | node, (declared ^ (bound | node))
    for name in bound:
        assert getattr(lib, name) is not None
    # libmirt_node.so brings the system's librccl with it; a process that also imports torch (which ships an RCCL of its own)
    # must not hold both, so the node library is looked at in a process of its own
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\nfrom raytracer_amd import _capi\nlib = _capi.load_node_library()\n"
            "missing = [n for n, _, _ in _capi.NODE_SYMBOLS if getattr(lib, n, None) is None]\nprint('missing', missing)\nsys.exit(1 if missing else 0)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr


def test_struct_sizes_match_header_layout():
    assert C.sizeof(_capi.Node) == 32
    assert C.sizeof(_capi.Sphere) == 144
    assert C.sizeof(_capi.Material) == 52
    assert C.sizeof(_capi.Light) == 32
    assert C.sizeof(_capi.Camera) == 40
    assert C.sizeof(_capi.RenderParams) == 40
    assert C.sizeof(_capi.Stats) == 56


def test_invalid_arguments_are_rejected_without_a_device():
    lib = _capi.load_library()
    h = C.c_void_p()
    assert lib.mirt_scene_create(None, 0, C.byref(h)) == -1
    assert b"NULL" in lib.mirt_last_error()
    d = _capi.SceneDesc()
    d.struct_size = 3
    assert lib.mirt_scene_create(C.byref(d), 0, C.byref(h)) == -1
    assert b"struct_size" in lib.mirt_last_error()
    p = _capi.make_params(64, 64, tile_w=32, tile_h=32, first_tile=1, tile_stride=4)
    # 2x2 tiles of 32x32 over 4 ranks: every rank's buffer is sized like rank 0's
    assert lib.mirt_tile_buffer_bytes(C.byref(p)) == 32 * 32 * 3
    p0 = _capi.make_params(100, 50)
    assert lib.mirt_tile_buffer_bytes(C.byref(p0)) == 100 * 50 * 3


def test_non_finite_geometry_is_rejected_before_any_device_call():
    import ctypes as C
    from raytracer_amd.scene import FlatScene
    lib = _capi.load_library()
    fs = FlatScene.from_text("size 8 8\ncamera 0 0 4 0 0 0 0 1 0 45\npushTransform\nscale 1 0 1\nsphere 0 0 0 1\npopTransform\n")
    assert not np.isfinite(fs.spheres).all()   # glm::inverse of the singular matrix, as the reference computes it
    d = fs.desc()
    h = C.c_void_p()
    assert lib.mirt_scene_create(C.byref(d), 0, C.byref(h)) == -1
    assert b"singular" in lib.mirt_last_error()
