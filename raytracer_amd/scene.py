"""FlatScene: the flattened form of a loaded reference Scene (Scene.h:13-48) that crosses the C ABI.

Pure data container (numpy arrays in the layouts of include/mirt.h) plus (de)serialisation; no tracing code.
"""
import ctypes as C

import numpy as np

from . import _capi

NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("a", "<u4"), ("bmax", "<f4", 3), ("b", "<u4")])
assert NODE_DTYPE.itemsize == 32

_ARRAYS = [
    ("nodes", NODE_DTYPE, None), ("leaf_prims", np.uint32, None),
    ("tri_verts", np.float32, 9), ("tri_prim", np.uint32, None),
    ("spheres", np.float32, 36), ("sphere_prim", np.uint32, None),
    ("prim_material", np.uint32, None), ("materials", np.float32, 13),
    ("lights", np.float32, 8), ("camera", np.float32, None), ("attenuation", np.float32, None),
]


class FlatScene:
    """nodes[NODE_DTYPE], leaf_prims[u32], tri_verts[n,9], tri_prim[u32], spheres[n,36] (c, r, M, Minv),
    sphere_prim[u32], prim_material[u32], materials[m,13], lights[l,8], camera[10], attenuation[3]."""

    def __init__(self, width, height, max_depth, **arrays):
        self.width, self.height, self.max_depth = int(width), int(height), int(max_depth)
        for name, dtype, cols in _ARRAYS:
            a = np.ascontiguousarray(arrays[name], dtype=dtype)
            if cols is not None:
                a = a.reshape(-1, cols)
            setattr(self, name, a)
        self.n_prims = int(self.prim_material.shape[0])
        self._validate()

    def _validate(self):
        n = self.nodes
        leaf = (n["b"] & _capi.MIRT_LEAF_FLAG) != 0
        first = n["a"][leaf].astype(np.int64)
        cnt = (n["b"][leaf] & 0x7FFFFFFF).astype(np.int64)
        if leaf.any() and (first + cnt).max() > self.leaf_prims.shape[0]:
            raise ValueError("leaf range outside leaf_prims")
        for side in ("a", "b"):
            ch = n[side][~leaf]
            ch = ch[ch != _capi.MIRT_NO_CHILD]
            if ch.size and ch.max() >= n.shape[0]:
                raise ValueError("child index outside nodes")
        if self.leaf_prims.size and self.leaf_prims.max() >= self.n_prims:
            raise ValueError("leaf primitive id outside scene objects")
        if self.prim_material.size and self.prim_material.max() >= self.materials.shape[0]:
            raise ValueError("material index outside table")
        if self.tri_prim.shape[0] != self.tri_verts.shape[0] or self.sphere_prim.shape[0] != self.spheres.shape[0]:
            raise ValueError("primitive id arrays do not match primitive arrays")
        if self.camera.shape != (10,) or self.attenuation.shape != (3,):
            raise ValueError("camera must have 10 floats and attenuation 3")

    # ---- C ABI view -------------------------------------------------------------------------------
    def desc(self):
        """ctypes SceneDesc pointing into this object's arrays (keep `self` alive while it is used)."""
        d = _capi.SceneDesc()
        d.struct_size = C.sizeof(_capi.SceneDesc)

        def ptr(a):
            return a.ctypes.data if a.size else None

        d.n_nodes, d.nodes = self.nodes.shape[0], ptr(self.nodes)
        d.n_leaf_prims, d.leaf_prims = self.leaf_prims.shape[0], ptr(self.leaf_prims)
        d.n_prims = self.n_prims
        d.n_triangles, d.tri_verts, d.tri_prim = self.tri_verts.shape[0], ptr(self.tri_verts), ptr(self.tri_prim)
        d.n_spheres, d.spheres, d.sphere_prim = self.spheres.shape[0], ptr(self.spheres), ptr(self.sphere_prim)
        d.n_materials, d.materials, d.prim_material = self.materials.shape[0], ptr(self.materials), ptr(self.prim_material)
        d.n_lights, d.lights = self.lights.shape[0], ptr(self.lights)
        d.max_depth = self.max_depth
        cam = self.camera
        d.camera.look_from[:] = cam[0:3].tolist()
        d.camera.look_at[:] = cam[3:6].tolist()
        d.camera.up[:] = cam[6:9].tolist()
        d.camera.fovy_deg = float(cam[9])
        d.attenuation[:] = self.attenuation.tolist()
        return d

    # ---- the library's own scene loader (include/mirt_host.h) ------------------------------------------
    @classmethod
    def _from_host_handle(cls, lib, h):
        try:
            d = _capi.SceneDesc()
            w, hgt = C.c_uint32(0), C.c_uint32(0)
            if lib.mirt_host_scene_desc(h, C.byref(d), C.byref(w), C.byref(hgt)) != 0:
                raise RuntimeError(lib.mirt_last_error().decode())

            def arr(ptr, dtype, count):
                dt = np.dtype(dtype)
                if not count:
                    return np.zeros(0, dt)
                buf = (C.c_char * (count * dt.itemsize)).from_address(ptr)
                return np.frombuffer(buf, dtype=dt, count=count).copy()

            cam = np.array(list(d.camera.look_from) + list(d.camera.look_at) + list(d.camera.up) + [d.camera.fovy_deg], np.float32)
            fs = cls(w.value, hgt.value, d.max_depth,
                     nodes=arr(d.nodes, NODE_DTYPE, d.n_nodes), leaf_prims=arr(d.leaf_prims, np.uint32, d.n_leaf_prims),
                     tri_verts=arr(d.tri_verts, np.float32, d.n_triangles * 9), tri_prim=arr(d.tri_prim, np.uint32, d.n_triangles),
                     spheres=arr(d.spheres, np.float32, d.n_spheres * 36), sphere_prim=arr(d.sphere_prim, np.uint32, d.n_spheres),
                     prim_material=arr(d.prim_material, np.uint32, d.n_prims), materials=arr(d.materials, np.float32, d.n_materials * 13),
                     lights=arr(d.lights, np.float32, d.n_lights * 8), camera=cam,
                     attenuation=np.array(list(d.attenuation), np.float32))
            ps, bs = C.c_double(0), C.c_double(0)
            lib.mirt_host_scene_timing(h, C.byref(ps), C.byref(bs))
            fs.parse_seconds, fs.build_seconds = ps.value, bs.value
            fs.output_name = lib.mirt_host_scene_output_name(h).decode()
            tr = C.c_uint64(0)
            lib.mirt_host_scene_truncated_nodes(h, C.byref(tr))
            fs.truncated_nodes = tr.value  # > 0: the reference itself cannot load this scene (its tree build never ends)
            return fs
        finally:
            lib.mirt_host_scene_destroy(h)

    @classmethod
    def from_file(cls, path, threads=0):
        """Parse a scene file in the reference's text format and build its Partition tree (Scene.cpp:15-224,
        Partition.cpp:4-14) with the library's host-side loader."""
        lib = _capi.load_library()
        h = C.c_void_p()
        if lib.mirt_host_scene_load(str(path).encode(), threads, C.byref(h)) != 0:
            raise ValueError(lib.mirt_last_error().decode())
        return cls._from_host_handle(lib, h)

    @classmethod
    def from_text(cls, text, threads=0):
        lib = _capi.load_library()
        raw = text.encode() if isinstance(text, str) else bytes(text)
        h = C.c_void_p()
        if lib.mirt_host_scene_parse(raw, len(raw), threads, C.byref(h)) != 0:
            raise ValueError(lib.mirt_last_error().decode())
        return cls._from_host_handle(lib, h)

    def rebuilt(self, threads=0):
        """Same primitives, Partition tree built again by the library (Partition.cpp:4-14)."""
        lib = _capi.load_library()
        d = self.desc()
        h = C.c_void_p()
        if lib.mirt_host_scene_from_prims(C.byref(d), self.width, self.height, threads, C.byref(h)) != 0:
            raise ValueError(lib.mirt_last_error().decode())
        return self._from_host_handle(lib, h)

    # ---- serialisation ------------------------------------------------------------------------------
    def save(self, path):
        np.savez_compressed(path, meta=np.array([self.width, self.height, self.max_depth], dtype=np.int64),
                            **{name: getattr(self, name) for name, _, _ in _ARRAYS})

    @classmethod
    def load(cls, path):
        with np.load(path, allow_pickle=False) as z:
            w, h, md = (int(x) for x in z["meta"])
            return cls(w, h, md, **{name: z[name] for name, _, _ in _ARRAYS})

    def with_size(self, width, height):
        """Same scene at another resolution (the reference's *_small scene files differ only in `size`)."""
        return FlatScene(width, height, self.max_depth, **{name: getattr(self, name) for name, _, _ in _ARRAYS})

    def summary(self):
        leaf = (self.nodes["b"] & _capi.MIRT_LEAF_FLAG) != 0
        return dict(width=self.width, height=self.height, max_depth=self.max_depth, prims=self.n_prims,
                    triangles=int(self.tri_verts.shape[0]), spheres=int(self.spheres.shape[0]),
                    nodes=int(self.nodes.shape[0]), leaves=int(leaf.sum()), leaf_refs=int(self.leaf_prims.shape[0]),
                    materials=int(self.materials.shape[0]), lights=int(self.lights.shape[0]))


def dedup_materials(per_prim_materials):
    """per-object material rows [n,13] -> (table [m,13], index [n]) keeping first-seen order, compared bitwise."""
    m = np.ascontiguousarray(per_prim_materials, dtype=np.float32).reshape(-1, 13)
    if m.shape[0] == 0:
        return m, np.zeros(0, np.uint32)
    keys = m.view(np.uint32).reshape(-1, 13)
    _, first, inv = np.unique(keys, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first)
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    return m[first[order]], rank[inv.reshape(-1)].astype(np.uint32)
