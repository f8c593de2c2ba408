import sys, ctypes as C, importlib.util
sys.path.insert(0, '.')
spec = importlib.util.spec_from_file_location('t', 'tests/test_gpu_tri_filter.py'); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
from raytracer_amd.scene import FlatScene
from raytracer_amd.device import DeviceScene
tris = t._triangles(); fs = FlatScene.from_text(t._scene_text(tris)); o, d = t._rays(tris)
with DeviceScene(fs, 0) as s:
    v = C.c_uint64(); s.lib.mirt_scene_query(s.h, 116, C.byref(v))
    s.trace_rays(o, d)
    s.lib.mirt_scene_query(s.h, 116, C.byref(v)); print("rays", len(o), "fallbacks", v.value)
