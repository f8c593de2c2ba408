"""Scene file in, PNG and report out: what the reference's createRender does for one scene (main.cpp:104-173), on the GPU.

    python -m raytracer_amd scene.test [--out image.png] [--size W H] [--features 15] [--debug 6] [--device 0] [--cache scene.mirt]

With --cache the uploaded scene is written to that file after the first load, and later runs start from it: no scene
text, no Partition build, no hierarchy build (mirt_scene_save / mirt_scene_load; the reference parses and builds every
time, Scene.cpp:15-224).

The output name defaults to the scene's `output` command, as in the reference (main.cpp:113-119).  The report lists the
same facts as createPerformanceReport (main.cpp:314-344), with the object and light counters filled in (the reference
prints uninitialised ones, Scene.h:47) and device times instead of whole seconds.
"""
import argparse
import os
import time

from . import _capi
from .device import DeviceScene
from .scene import FlatScene

FEATURE_NAMES = [(1, "DIFFUSE"), (2, "SPECULAR"), (4, "SHADOWS"), (8, "REFLECTIONS")]            # main.cpp:35-49
DEBUG_NAMES = ["DIFFUSE_INTENSITY", "SPECULAR_INTENSITY", "NORMAL_MAP", "SHADOW_MAP", "PRIMARY_INTERSECTION_MAP",
               "LIGHT_DIRECTION_MAP", "NONE"]                                                     # main.cpp:24-33


def render_cached(cache_path, out_path, size=None, features=_capi.F_ALL, debug=_capi.DEBUG_NONE, device=0, want_image=True):
    """Render from a scene cache file.  Returns (image, stats, report lines)."""
    t0 = time.perf_counter()
    scene, user = DeviceScene.load(cache_path, device)
    t_load = time.perf_counter() - t0
    w, h = size or (user[0], user[1])
    with scene as s:
        t0 = time.perf_counter()
        st = s.render_png(out_path, w, h, features=features, debug=debug)   # frame + PNG file in one call
        t_png = time.perf_counter() - t0
        img = s.render(w, h, features=features, debug=debug)[0] if want_image else None
        facts = s.query()
    rays = st["primary_rays"] + st["shadow_rays"] + st["reflection_rays"]
    lines = ["PERFORMANCE REPORT FOR " + os.path.basename(out_path), "Scene cache: %s (%d bytes on the device)" % (cache_path, facts["device_bytes"]),
             "Resolution: %dx%d" % (w, h), "Render Time: %.3f ms on the device" % st["device_ms"],
             "Scene from cache: %.3f s, frame + PNG file: %.3f s" % (t_load, t_png),
             "Rays: %d primary, %d shadow, %d reflection (%.1f Mrays/s)" % (st["primary_rays"], st["shadow_rays"], st["reflection_rays"], rays / max(st["device_ms"], 1e-9) / 1e3)]
    return img, st, lines


def text_digest(scene_path):
    """Two words that identify the scene text a cache was made from (they travel in the cache file's user words)."""
    import hashlib
    with open(scene_path, "rb") as f:
        d = hashlib.sha256(f.read()).digest()
    return int.from_bytes(d[:4], "little"), int.from_bytes(d[4:8], "little")


def cache_matches(cache_path, scene_path):
    """Was this cache file written from this scene text?  A scene file that cannot be read leaves the cache in charge (that
    is what the cache is for: rendering without the text); a readable one whose digest differs means the text was edited or
    the cache belongs to another scene, and the scene is built again."""
    import struct
    if not os.path.exists(scene_path):
        return True
    with open(cache_path, "rb") as f:
        head = f.read(8 + 4 * 4 + 4 * 4)
    if len(head) < 40:
        return False
    user = struct.unpack_from("<4I", head, 24)
    return (user[2], user[3]) == text_digest(scene_path)


def render_file(scene_path, out_path=None, size=None, features=_capi.F_ALL, debug=_capi.DEBUG_NONE, device=0, report=True, cache=None, want_image=True):
    """The whole of createRender (main.cpp:104-173).  The PNG file comes from mirt_render_png (frame, PNG rows on the device,
    deflate as the rows arrive); want_image also fetches the frame as an array (one more frame) for the caller."""
    if cache and os.path.exists(cache):
        if cache_matches(cache, scene_path):
            return render_cached(cache, out_path or os.path.splitext(os.path.basename(scene_path))[0] + ".png", size, features, debug, device, want_image)
        print("%s was not written from %s as it is now: building the scene again" % (cache, scene_path))
    t0 = time.perf_counter()
    flat = FlatScene.from_file(scene_path)
    t_load = time.perf_counter() - t0
    if size:
        flat = flat.with_size(*size)
    out_path = out_path or flat.output_name
    t0 = time.perf_counter()
    with DeviceScene(flat, device) as s:
        t_upload = time.perf_counter() - t0
        if cache:
            s.save(cache, user=(flat.width, flat.height) + text_digest(scene_path))
        t0 = time.perf_counter()
        st = s.render_png(out_path, flat.width, flat.height, features=features, debug=debug)
        t_png = time.perf_counter() - t0
        img = s.render(flat.width, flat.height, features=features, debug=debug)[0] if want_image else None
    lines = []
    if report:
        n_px = flat.width * flat.height
        rays = st["primary_rays"] + st["shadow_rays"] + st["reflection_rays"]
        point = int((flat.lights[:, 3] > 0).sum())
        lines = [
            "PERFORMANCE REPORT FOR " + os.path.basename(out_path),
            "--------------------------------------------------------------------", "",
            "100% Completed", "",
            "Input Scene File: %s" % scene_path, "Output Image: %s" % out_path,
            "Resolution: %dx%d" % (flat.width, flat.height), "Pixels Processed: %d" % n_px, "",
            "Features Enabled: " + " ".join(n for b, n in FEATURE_NAMES if features & b),
            "Debug Options: " + DEBUG_NAMES[debug], "",
            "Render Time: %.3f ms on the device" % st["device_ms"],
            "Milliseconds Per Pixel: %.3g" % (st["device_ms"] / n_px), "",
            "Time Breakdown",
            "Scene load + tree build: %.3f s, upload: %.3f s, frame + PNG file: %.3f s" % (t_load, t_upload, t_png),
            "Rays: %d primary, %d shadow, %d reflection (%.1f Mrays/s)" % (st["primary_rays"], st["shadow_rays"], st["reflection_rays"], rays / max(st["device_ms"], 1e-9) / 1e3),
            "Total objects: %d" % flat.n_prims, "----- Spheres: %d" % flat.spheres.shape[0], "----- Triangles: %d" % flat.tri_verts.shape[0],
            "Total lights: %d" % flat.lights.shape[0], "----- Directional: %d" % (flat.lights.shape[0] - point), "----- Point: %d" % point,
        ]
        os.makedirs("reports", exist_ok=True)                                                    # main.cpp:22
        with open(os.path.join("reports", os.path.basename(out_path) + "_report.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    return img, st, lines


def main():
    ap = argparse.ArgumentParser(prog="python -m raytracer_amd")
    ap.add_argument("scene")
    ap.add_argument("--out")
    ap.add_argument("--size", type=int, nargs=2, metavar=("W", "H"))
    ap.add_argument("--features", type=int, default=_capi.F_ALL)
    ap.add_argument("--debug", type=int, default=_capi.DEBUG_NONE)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--cache", help="scene cache file: written after the first load, read instead of the scene text afterwards")
    a = ap.parse_args()
    _, _, lines = render_file(a.scene, a.out, a.size, a.features, a.debug, a.device, cache=a.cache, want_image=False)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
