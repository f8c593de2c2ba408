"""CPU, development container only (needs /root/reference and the reference build under oracle/_ref): the MirtFlattener
that the integration guide at the repository root tells a maintainer to add is compiled AS IT STANDS IN THE GUIDE against the reference's headers and
classes, and the arrays it fills are held equal, byte for byte, to the oracle harness's flattening of the same scene."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/RayTracer"
REF_OBJS = [os.path.join(ROOT, "oracle", "_ref", n + ".o") for n in ("AABB", "Camera", "Color", "Partition", "Scene", "Shape", "Sphere", "Triangle")]

pytestmark = pytest.mark.skipif(not (os.path.isdir(REF) and all(os.path.exists(o) for o in REF_OBJS)),
                                reason="the reference's sources and its build (make -C oracle ref) exist in the development container only")

SCENES = ["scene1", "scene2", "scene3", "scene4-specular", "scene5", "scene6", "scene7"]


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("flattener")
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```cpp\n(.*?)```", doc, re.S)
    snippet = next(b for b in blocks if "struct MirtFlattener" in b)
    (tmp / "mirt_flattener_snippet.inc").write_text(snippet)
    so = str(tmp / "libflatcheck.so")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-fPIC", "-shared", "-w", "-include", "math.h", "-ffp-contract=off",
                           "-I" + str(tmp), "-I" + REF, "-I" + os.path.join(REF, "glm"), "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "flattener_check.cpp")] + REF_OBJS +
                          ["-L" + os.path.join(ROOT, "raytracer_amd"), "-lmirt", "-Wl,-rpath," + os.path.join(ROOT, "raytracer_amd"), "-o", so])
    lib = C.CDLL(so)
    lib.flattener_check.restype = C.c_int
    return lib


@pytest.mark.parametrize("name", SCENES)
def test_document_flattener_fills_the_same_arrays_as_the_oracle_harness(name, checker):
    from oracle.bind import RefOracle
    path = os.path.join(ROOT, "raytracer_amd", "data", "scenes", name + ".test")
    ref = RefOracle(path)
    nodes = np.zeros(ref.n_nodes * 8, np.float32)
    leaf = np.zeros(ref.n_leaf_refs, np.uint32)
    tv = np.zeros((ref.n_tris, 9), np.float32)
    tp = np.zeros(ref.n_tris, np.uint32)
    sp = np.zeros((ref.n_spheres, 36), np.float32)
    spp = np.zeros(ref.n_spheres, np.uint32)
    mats = np.zeros((ref.n_prims, 13), np.float32)
    lights = np.zeros((ref.n_lights, 8), np.float32)
    cam = np.zeros(10, np.float32)
    att = np.zeros(3, np.float32)
    ref.L.ref_scene_flatten(ref.h, *[a.ctypes.data for a in (nodes, leaf, tv, tp, sp, spp, mats, lights, cam, att)])
    p = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    rc = checker.flattener_check(path.encode(), C.c_uint32(ref.n_nodes), p(nodes), C.c_uint32(ref.n_leaf_refs), p(leaf), C.c_uint32(ref.n_tris), p(tv), p(tp),
                                 C.c_uint32(ref.n_spheres), p(sp), p(spp), C.c_uint32(ref.n_prims), p(mats), C.c_uint32(ref.n_lights), p(lights),
                                 p(cam), p(att), C.c_int(ref.max_depth))
    assert rc == 0, "array %d differs" % rc
    ref.close()
