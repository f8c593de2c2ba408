import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MIRT_VERBOSE"] = "1"
from raytracer_amd import workloads
from raytracer_amd.device import DeviceScene
from raytracer_amd.scene import FlatScene
flat = FlatScene.from_text(workloads.dragon_grid())
print("parse %.2f build %.2f" % (flat.parse_seconds, flat.build_seconds), flush=True)
t = time.perf_counter(); s = DeviceScene(flat, 0); print("create %.2f" % (time.perf_counter() - t)); s.close()
