"""One-off: whole images of the hierarchy walk (EXACT) against the walk of the reference's own tree (REFWALK) at full size."""
import sys
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import workloads, _capi
from raytracer_amd.scene import FlatScene
from raytracer_amd.device import DeviceScene
for name, (w, h) in (("dragon", (3840, 2160)), ("spheres", (1920, 1080)), ("cornell", (640, 480))):
    fs = FlatScene.from_text(getattr(workloads, name)(w, h))
    with DeviceScene(fs, 0) as s:
        a, sa = s.render(w, h, trace_mode=_capi.TRACE_EXACT)
        b, sb = s.render(w, h, trace_mode=_capi.TRACE_REFWALK)
        same = np.array_equal(a, b)
        print(name, w, h, "images equal:", same, "| rays", [sa[k] for k in ("primary_rays", "shadow_rays", "reflection_rays")] == [sb[k] for k in ("primary_rays", "shadow_rays", "reflection_rays")],
              "| device ms exact %.2f refwalk %.2f" % (sa["device_ms"], sb["device_ms"]))
        assert same
