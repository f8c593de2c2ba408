"""TEST INFRASTRUCTURE ONLY: ctypes bindings of the two checkers.

  RefOracle  -> oracle/_ref/libref_oracle.so : the reference's own compiled hot-path classes + harness
  CPort      -> oracle/liboracle.so          : the C restatement (mirt_oracle.c)

May be imported from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import _capi
from raytracer_amd.scene import FlatScene, NODE_DTYPE, dedup_materials

HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB = os.path.join(HERE, "_ref", "libref_oracle.so")
PORT_LIB = os.path.join(HERE, "liboracle.so")
REFERENCE_ROOT = "/root/reference/RayTracer"


def build(ref=True):
    """Compile the C restatement, and the reference build when the reference sources are present."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    if ref and os.path.isdir(REFERENCE_ROOT):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class RefOracle:
    """The reference's compiled Scene/Partition/AABB/Triangle/Sphere/Camera/Color behind a C ABI."""

    def __init__(self, scene_path):
        if not os.path.exists(REF_LIB):
            raise RuntimeError("oracle/_ref/libref_oracle.so missing (make -C oracle ref, needs /root/reference)")
        L = C.CDLL(REF_LIB)
        L.ref_scene_load.restype = C.c_void_p
        L.ref_scene_load.argtypes = [C.c_char_p]
        L.ref_scene_free.argtypes = [C.c_void_p]
        L.ref_scene_counts.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_scene_flatten.argtypes = [C.c_void_p] + [C.c_void_p] * 10
        L.ref_primary_rays.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_trace_closest.restype = C.c_int64
        L.ref_trace_closest.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 6
        L.ref_prim_intersect.argtypes = [C.c_void_p, C.c_int, C.c_int64] + [C.c_void_p] * 4
        L.ref_box_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_render_rows.restype = C.c_double
        L.ref_render_rows.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_void_p]
        self.L = L
        self.h = L.ref_scene_load(scene_path.encode())
        if not self.h:
            raise RuntimeError("reference Scene failed to load %s" % scene_path)
        c = np.zeros(10, np.int64)
        L.ref_scene_counts(self.h, c.ctypes.data)
        (self.width, self.height, self.max_depth, self.n_nodes, self.n_leaf_refs, self.n_prims,
         self.n_tris, self.n_spheres, self.n_lights, self.tree_depth) = (int(x) for x in c)

    def close(self):
        if self.h:
            self.L.ref_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def flatten(self):
        nodes = np.zeros(self.n_nodes, NODE_DTYPE)
        leaf = np.zeros(self.n_leaf_refs, np.uint32)
        tv = np.zeros((self.n_tris, 9), np.float32)
        tp = np.zeros(self.n_tris, np.uint32)
        sp = np.zeros((self.n_spheres, 36), np.float32)
        spp = np.zeros(self.n_spheres, np.uint32)
        mats = np.zeros((self.n_prims, 13), np.float32)
        lights = np.zeros((self.n_lights, 8), np.float32)
        cam = np.zeros(10, np.float32)
        att = np.zeros(3, np.float32)
        self.L.ref_scene_flatten(self.h, *[a.ctypes.data for a in (nodes, leaf, tv, tp, sp, spp, mats, lights, cam, att)])
        table, index = dedup_materials(mats)
        return FlatScene(self.width, self.height, self.max_depth, nodes=nodes, leaf_prims=leaf, tri_verts=tv, tri_prim=tp,
                         spheres=sp, sphere_prim=spp, prim_material=index, materials=table, lights=lights,
                         camera=cam, attenuation=att)

    def primary_rays(self, w, h):
        o = np.zeros((h * w, 3), np.float32)
        d = np.zeros((h * w, 3), np.float32)
        self.L.ref_primary_rays(self.h, w, h, o.ctypes.data, d.ctypes.data)
        return o, d

    def trace(self, origins, dirs, want_normals=True):
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = o.shape[0]
        dist = np.zeros(n, np.float32)
        ids = np.zeros(n, np.int32)
        nrm = np.zeros((n, 3), np.float32)
        vis = np.zeros(3, np.uint64)
        bad = self.L.ref_trace_closest(self.h, n, o.ctypes.data, d.ctypes.data, dist.ctypes.data, ids.ctypes.data,
                                       nrm.ctypes.data if want_normals else None, vis.ctypes.data)
        if bad:
            raise AssertionError("mirror traversal disagrees with Scene::findClosestIntersection on %d rays" % bad)
        return ids, dist, nrm, vis

    def prim_intersect(self, prim, origins, dirs):
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        dist = np.zeros(o.shape[0], np.float32)
        nrm = np.zeros((o.shape[0], 3), np.float32)
        self.L.ref_prim_intersect(self.h, prim, o.shape[0], o.ctypes.data, d.ctypes.data, dist.ctypes.data, nrm.ctypes.data)
        return dist, nrm

    def box_intersect(self, bmin, bmax, origins, dirs):
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        lo, hi = _f32(bmin), _f32(bmax)
        hit = np.zeros(o.shape[0], np.uint8)
        self.L.ref_box_intersect(lo.ctypes.data, hi.ctypes.data, o.shape[0], o.ctypes.data, d.ctypes.data, hit.ctypes.data)
        return hit

    def render(self, w=None, h=None, features=_capi.F_ALL, debug=_capi.DEBUG_NONE, row0=0, row1=None, row_step=1, col_step=1):
        w = w or self.width
        h = h or self.height
        row1 = h if row1 is None else row1
        bgr = np.zeros((h, w, 3), np.uint8)
        tot = np.zeros(3, np.uint64)
        secs = 0.0
        spans = [(row0, row1)] if row_step == 1 else [(r, r + 1) for r in range(row0, row1, row_step)]
        for a, b in spans:
            rays = np.zeros(3, np.uint64)
            t = self.L.ref_render_rows(self.h, features, debug, w, h, a, b, col_step, bgr.ctypes.data, rays.ctypes.data)
            if t < 0:
                raise AssertionError("rows [%d,%d): the restated shading driver (ray counts) and the reference's own text disagree on a byte" % (a, b))
            secs += t
            tot += rays
        return bgr, dict(primary=int(tot[0]), shadow=int(tot[1]), reflection=int(tot[2]), seconds=secs)


class CPort:
    """The C restatement over a FlatScene."""

    def __init__(self, scene: FlatScene):
        if not os.path.exists(PORT_LIB):
            raise RuntimeError("oracle/liboracle.so missing (make -C oracle)")
        L = C.CDLL(PORT_LIB)
        D, P, S = C.POINTER(_capi.SceneDesc), C.POINTER(_capi.RenderParams), C.POINTER(_capi.Stats)
        L.mirt_oracle_trace_rays.argtypes = [D, C.c_uint64] + [C.c_void_p] * 6
        L.mirt_oracle_primary_rays.argtypes = [D, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.mirt_oracle_render.argtypes = [D, P, C.c_void_p, S, C.c_void_p]
        L.mirt_oracle_render_rows.argtypes = [D, P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, S, C.c_void_p]
        L.mirt_oracle_triangle.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 4
        L.mirt_oracle_sphere.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 4
        L.mirt_oracle_box.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64] + [C.c_void_p] * 3
        self.L = L
        self.scene = scene
        self.d = scene.desc()

    def primary_rays(self, w, h):
        o = np.zeros((h * w, 3), np.float32)
        d = np.zeros((h * w, 3), np.float32)
        assert self.L.mirt_oracle_primary_rays(C.byref(self.d), w, h, o.ctypes.data, d.ctypes.data) == 0
        return o, d

    def trace(self, origins, dirs):
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = o.shape[0]
        dist = np.zeros(n, np.float32)
        ids = np.zeros(n, np.int32)
        nrm = np.zeros((n, 3), np.float32)
        vis = np.zeros(3, np.uint64)
        rc = self.L.mirt_oracle_trace_rays(C.byref(self.d), n, o.ctypes.data, d.ctypes.data, ids.ctypes.data, dist.ctypes.data,
                                           nrm.ctypes.data, vis.ctypes.data)
        assert rc == 0, rc
        return ids, dist, nrm, vis

    def render(self, w=None, h=None, features=_capi.F_ALL, debug=_capi.DEBUG_NONE, count_visits=False, **tiles):
        w = w or self.scene.width
        h = h or self.scene.height
        p = _capi.make_params(w, h, features, debug, **tiles)
        bgr = np.zeros((h, w, 3), np.uint8)
        st = _capi.Stats()
        vis = np.zeros((3, 3), np.uint64)
        rc = self.L.mirt_oracle_render(C.byref(self.d), C.byref(p), bgr.ctypes.data, C.byref(st), vis.ctypes.data if count_visits else None)
        assert rc == 0, rc
        out = st.as_dict()
        if count_visits:
            out["visits"] = vis
        return bgr, out

    def render_rows(self, w, h, row0, row1, row_step=1, col_step=1, features=_capi.F_ALL):
        """Bounded sample of the image: every row_step-th row, every col_step-th column.  Returns the image (only the
        sampled pixels written), ray counts and the [ray kind][box tests, leaf visits, primitive tests] matrix."""
        p = _capi.make_params(w, h, features)
        bgr = np.zeros((h, w, 3), np.uint8)
        st = _capi.Stats()
        vis = np.zeros((3, 3), np.uint64)
        rc = self.L.mirt_oracle_render_rows(C.byref(self.d), C.byref(p), row0, row1, row_step, col_step, bgr.ctypes.data, C.byref(st), vis.ctypes.data)
        assert rc == 0, rc
        out = st.as_dict()
        out["visits"] = vis
        return bgr, out

    def triangle(self, v9, origins, dirs):
        v = _f32(v9).reshape(9)
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        dist = np.zeros(o.shape[0], np.float32)
        nrm = np.zeros((o.shape[0], 3), np.float32)
        self.L.mirt_oracle_triangle(v.ctypes.data, o.shape[0], o.ctypes.data, d.ctypes.data, dist.ctypes.data, nrm.ctypes.data)
        return dist, nrm

    def sphere(self, s36, origins, dirs):
        s = _f32(s36).reshape(36)
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        dist = np.zeros(o.shape[0], np.float32)
        nrm = np.zeros((o.shape[0], 3), np.float32)
        self.L.mirt_oracle_sphere(s.ctypes.data, o.shape[0], o.ctypes.data, d.ctypes.data, dist.ctypes.data, nrm.ctypes.data)
        return dist, nrm

    def box(self, bmin, bmax, origins, dirs):
        lo, hi = _f32(bmin), _f32(bmax)
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        hit = np.zeros(o.shape[0], np.uint8)
        self.L.mirt_oracle_box(lo.ctypes.data, hi.ctypes.data, o.shape[0], o.ctypes.data, d.ctypes.data, hit.ctypes.data)
        return hit
