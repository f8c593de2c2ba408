"""ctypes mirror of include/mirt.h + include/mirt_host.h and loader of the HIP library (libmirt.so).

The product path has no CPU fallback: if libmirt.so is missing or fails to load, importing
``raytracer_amd.device`` raises.  This module only declares the ABI; it performs no arithmetic.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIRT_LIB") or os.path.join(HERE, "libmirt.so")  # MIRT_LIB: A/B builds while tuning

MIRT_NO_CHILD = 0x7FFFFFFF
MIRT_LEAF_FLAG = 0x80000000

F_DIFFUSE, F_SPECULAR, F_SHADOWS, F_REFLECTIONS, F_ALL = 1, 2, 4, 8, 15
(DEBUG_DIFFUSE_INTENSITY, DEBUG_SPECULAR_INTENSITY, DEBUG_NORMAL_MAP, DEBUG_SHADOW_MAP,
 DEBUG_PRIMARY_MAP, DEBUG_LIGHT_DIRECTION_MAP, DEBUG_NONE) = range(7)
TRACE_EXACT, TRACE_FAST, TRACE_REFWALK = 0, 1, 2
K_PRIMARY, K_TRACE, K_SHADE, K_RESOLVE, K_FRAME = range(5)
KERNEL_NAMES = ["primary", "trace", "shade", "resolve", "frame"]


class Node(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("a", C.c_uint32), ("bmax", C.c_float * 3), ("b", C.c_uint32)]


class Sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("M", C.c_float * 16), ("Minv", C.c_float * 16)]


class Material(C.Structure):
    _fields_ = [("diffuse", C.c_float * 3), ("specular", C.c_float * 3), ("emission", C.c_float * 3),
                ("ambient", C.c_float * 3), ("shininess", C.c_float)]


class Light(C.Structure):
    _fields_ = [("location", C.c_float * 4), ("color", C.c_float * 3), ("pad", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("look_from", C.c_float * 3), ("look_at", C.c_float * 3), ("up", C.c_float * 3), ("fovy_deg", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_nodes", C.c_uint32), ("nodes", C.c_void_p),
        ("n_leaf_prims", C.c_uint32), ("leaf_prims", C.c_void_p),
        ("n_prims", C.c_uint32), ("n_triangles", C.c_uint32), ("tri_verts", C.c_void_p), ("tri_prim", C.c_void_p),
        ("n_spheres", C.c_uint32), ("n_materials", C.c_uint32), ("spheres", C.c_void_p), ("sphere_prim", C.c_void_p),
        ("prim_material", C.c_void_p), ("materials", C.c_void_p),
        ("n_lights", C.c_uint32), ("max_depth", C.c_int32), ("lights", C.c_void_p),
        ("camera", Camera), ("attenuation", C.c_float * 3),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
        ("feature_bits", C.c_uint32), ("debug_mode", C.c_uint32), ("trace_mode", C.c_uint32),
        ("tile_w", C.c_uint32), ("tile_h", C.c_uint32), ("first_tile", C.c_uint32), ("tile_stride", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("reflection_rays", C.c_uint64),
        ("shadow_rays_traced", C.c_uint64), ("reflection_rays_traced", C.c_uint64), ("primary_hits", C.c_uint64),
        ("device_ms", C.c_float), ("pad", C.c_float),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "pad"}


def make_params(width, height, features=F_ALL, debug=DEBUG_NONE, trace_mode=TRACE_EXACT,
                tile_w=0, tile_h=0, first_tile=0, tile_stride=1):
    p = RenderParams()
    p.struct_size = C.sizeof(RenderParams)
    p.width, p.height = width, height
    p.feature_bits, p.debug_mode, p.trace_mode = features, debug, trace_mode
    p.tile_w, p.tile_h, p.first_tile, p.tile_stride = tile_w, tile_h, first_tile, tile_stride
    return p


# Every symbol include/mirt.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("mirt_scene_create", C.c_int, [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]),
    ("mirt_scene_destroy", None, [C.c_void_p]),
    ("mirt_scene_save", C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    ("mirt_scene_load", C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]),
    ("mirt_render", C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]),
    ("mirt_render_device", C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_size_t, C.c_void_p]),
    ("mirt_tile_buffer_bytes", C.c_size_t, [C.POINTER(RenderParams)]),
    ("mirt_render_device_noalloc", C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_size_t, C.c_void_p]),
    ("mirt_release_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("mirt_set_frames_in_flight", C.c_int, [C.c_void_p, C.c_uint32]),
    ("mirt_get_stats", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
    ("mirt_scene_query", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]),
    ("mirt_render_png", C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_char_p, C.POINTER(Stats)]),
    ("mirt_set_profiling", C.c_int, [C.c_void_p, C.c_int]),
    ("mirt_clock_probe_start", C.c_int, [C.c_void_p, C.c_uint32]),
    ("mirt_clock_probe_read", C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    ("mirt_get_launch_times", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    ("mirt_assemble_tiles", C.c_int, [C.c_int, C.POINTER(RenderParams), C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    ("mirt_trace_rays", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mirt_trace_primary_ids", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("mirt_test_powf", C.c_int, [C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mirt_last_error", C.c_char_p, []),
    # include/mirt_host.h
    ("mirt_host_scene_load", C.c_int, [C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("mirt_host_scene_parse", C.c_int, [C.c_char_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("mirt_host_scene_from_prims", C.c_int, [C.POINTER(SceneDesc), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("mirt_host_scene_desc", C.c_int, [C.c_void_p, C.POINTER(SceneDesc), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("mirt_host_scene_output_name", C.c_char_p, [C.c_void_p]),
    ("mirt_host_scene_truncated_nodes", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("mirt_host_scene_timing", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("mirt_host_scene_destroy", None, [C.c_void_p]),
    ("mirt_write_png", C.c_int, [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]),
]

# include/mirt_node.h (libmirt_node.so: libmirt.so + RCCL)
NODE_SYMBOLS = [
    ("mirt_node_create", C.c_int, [C.POINTER(SceneDesc), C.POINTER(C.c_int), C.c_uint32, C.POINTER(C.c_void_p)]),
    ("mirt_node_destroy", None, [C.c_void_p]),
    ("mirt_node_render_device", C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_uint32, C.POINTER(C.c_void_p)]),
    ("mirt_node_sync", C.c_int, [C.c_void_p, C.c_uint32]),
    ("mirt_node_render", C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]),
    ("mirt_node_uses_rccl", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
]
NODE_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmirt_node.so")
_node_lib = None


def load_node_library():
    """dlopen libmirt_node.so (it brings libmirt.so and librccl.so with it) and bind include/mirt_node.h."""
    global _node_lib
    if _node_lib is None:
        if not os.path.exists(NODE_LIB_PATH):
            raise OSError("%s not built (make -C raytracer_amd/csrc)" % NODE_LIB_PATH)
        load_library()  # the same libmirt.so the node library resolves its symbols against
        lib = C.CDLL(NODE_LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, res, args in NODE_SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _node_lib = lib
    return _node_lib


_lib = None


def load_library(path=None):
    """dlopen libmirt.so and bind every declared symbol.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            "raytracer_amd: %s not found -- build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback." % p)
    lib = C.CDLL(p)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = restype
        fn.argtypes = argtypes
    if path is None:
        _lib = lib
    return lib
