"""GPU: mirt_scene_save / mirt_scene_load -- the device form of a scene in a file.  A scene that comes back from its cache
renders the reference's images byte for byte without the scene text, the tree or mirt_scene_create; a damaged or foreign
file is refused with a message."""
import numpy as np
import pytest

from conftest import bits
from raytracer_amd import _capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["scene5", "scene6", "scene7"])
def test_scene_from_cache_renders_the_same_images(name, flat_scenes, goldens, tmp_path):
    from raytracer_amd.device import DeviceScene
    g = goldens(name)
    w, h = int(g["width"]), int(g["height"])
    path = str(tmp_path / (name + ".mirt"))
    with DeviceScene(flat_scenes(name), 0) as s:
        facts = s.query()
        s.save(path, user=(w, h, 7, 9))
    cached, user = DeviceScene.load(path, 0)
    with cached as s:
        assert user == (w, h, 7, 9)
        assert s.query() == facts
        for features in (15, 7, 11, 3):
            img, st = s.render(w, h, features=features)
            assert np.array_equal(img, g["bgr_f%d" % features])
            assert [st["primary_rays"], st["shadow_rays"], st["reflection_rays"]] == g["rays_f%d" % features].tolist()
        img, _ = s.render(w, h, debug=_capi.DEBUG_SHADOW_MAP)
        assert np.array_equal(img, g["bgr_debug%d" % _capi.DEBUG_SHADOW_MAP])
        ids, dist = s.trace_primary_ids(w, h)
        assert np.array_equal(ids, g["prim_id"]) and np.array_equal(bits(dist), bits(g["dist"]))
        ids, dist = s.trace_primary_ids(w, h, trace_mode=_capi.TRACE_REFWALK)   # the reference's own tree travels too
        assert np.array_equal(ids, g["prim_id"]) and np.array_equal(bits(dist), bits(g["dist"]))


def test_damaged_and_foreign_files_are_refused(flat_scenes, tmp_path):
    from raytracer_amd.device import DeviceScene, MirtError
    path = tmp_path / "scene6.mirt"
    with DeviceScene(flat_scenes("scene6"), 0) as s:
        s.save(str(path))
    raw = bytearray(path.read_bytes())
    bad = tmp_path / "flipped.mirt"
    flipped = bytearray(raw)
    flipped[len(raw) // 2] ^= 0x10
    bad.write_bytes(flipped)
    with pytest.raises(MirtError, match="checksum"):
        DeviceScene.load(str(bad), 0)
    bad.write_bytes(raw[:len(raw) // 3])
    with pytest.raises(MirtError, match="truncated"):
        DeviceScene.load(str(bad), 0)
    bad.write_bytes(b"not a scene at all, just text that is longer than the header of a cache file would be" * 2)
    with pytest.raises(MirtError, match="not a scene cache"):
        DeviceScene.load(str(bad), 0)
    with pytest.raises(MirtError, match="cannot open"):
        DeviceScene.load(str(tmp_path / "missing.mirt"), 0)


def test_command_line_starts_from_the_cache_the_second_time(tmp_path, goldens, monkeypatch):
    import os
    from raytracer_amd.__main__ import render_file
    monkeypatch.chdir(tmp_path)
    scene = os.path.join(os.path.dirname(os.path.dirname(__file__)), "raytracer_amd", "data", "scenes", "scene6.test")
    cache = str(tmp_path / "cornell.mirt")
    a, _, _ = render_file(scene, out_path=str(tmp_path / "a.png"), cache=cache)
    assert os.path.exists(cache)
    b, _, lines = render_file("/nonexistent/scene6.test", out_path=str(tmp_path / "b.png"), cache=cache)  # the text is not read again
    assert np.array_equal(a, b) and np.array_equal(a, goldens("c1_cornell")["bgr"])
    assert any("Scene cache" in l for l in lines)


_WRITE_CACHE = r'''
import sys
sys.path.insert(0, %r)
from raytracer_amd import workloads
from raytracer_amd.device import DeviceScene
from raytracer_amd.scene import FlatScene
flat = FlatScene.from_text(workloads.dragon_grid(640, 360, nx=2, nz=2))   # 400 000 triangles: large ranges, subtree tasks
with DeviceScene(flat, 0) as s:
    s.save(sys.argv[1])
'''


def test_scene_preparation_does_not_depend_on_the_thread_count(tmp_path):
    """The whole hand-over -- text parsed in pieces, Partition tree from tasks, membership lists, hierarchy build, collapse and
    packing on the worker pool -- gives the same device arrays on one thread as on five: the cache files of two processes are
    equal byte for byte behind the record that holds device addresses."""
    import os
    import struct
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    blobs = []
    for threads in (1, 5):
        path = str(tmp_path / ("t%d.mirt" % threads))
        env = dict(os.environ, MIRT_THREADS=str(threads))
        subprocess.run([sys.executable, "-c", _WRITE_CACHE % root, path], env=env, check=True, timeout=600)
        raw = open(path, "rb").read()
        magic, version, dscene_bytes, meta_bytes = struct.unpack_from("<8sIII", raw, 0)
        assert magic == b"MIRTSCN1"
        header = 8 + 4 * 4 + 4 * 4 + 8
        blobs.append(raw[header + dscene_bytes:])   # SceneMeta and the eleven arrays
    assert len(blobs[0]) == len(blobs[1]) and blobs[0] == blobs[1]


def test_cache_written_by_one_build_loads_into_a_build_with_a_smaller_lds_stack(flat_scenes, goldens, tmp_path):
    """How many traversal-stack entries lie beyond LDS is a property of the build that RENDERS, not of the build that wrote the
    file: a cache from the default library (18 entries in LDS) must come back with the overflow area in place in the
    library built with 2."""
    import os
    from raytracer_amd.device import DeviceScene
    small = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "libmirt_smallstack.so")
    if not os.path.exists(small):
        pytest.skip("libmirt_smallstack.so not built")
    lib = _capi.load_library(small)
    g = goldens("scene7")
    w, h = int(g["width"]), int(g["height"])
    path = str(tmp_path / "scene7.mirt")
    with DeviceScene(flat_scenes("scene7"), 0) as s:
        s.save(path)
    cached, _ = DeviceScene.load(path, 0, lib=lib)
    with cached as s:
        img, _ = s.render(w, h)
        assert np.array_equal(img, g["bgr_f15"])
        ids, dist = s.trace_primary_ids(w, h)
        assert np.array_equal(ids, g["prim_id"]) and np.array_equal(bits(dist), bits(g["dist"]))
    # and the other way round
    with DeviceScene(flat_scenes("scene7"), 0, lib=lib) as s:
        s.save(path)
    cached, _ = DeviceScene.load(path, 0)
    with cached as s:
        img, _ = s.render(w, h)
        assert np.array_equal(img, g["bgr_f15"])


def test_command_line_rebuilds_when_the_scene_text_has_changed(tmp_path, monkeypatch):
    import os
    import shutil
    from raytracer_amd.__main__ import render_file
    monkeypatch.chdir(tmp_path)
    src = os.path.join(os.path.dirname(os.path.dirname(__file__)), "raytracer_amd", "data", "scenes", "scene6.test")
    scene = str(tmp_path / "scene.test")
    shutil.copy(src, scene)
    cache = str(tmp_path / "scene.mirt")
    a, _, _ = render_file(scene, out_path=str(tmp_path / "a.png"), cache=cache, size=(160, 120))
    b, _, lines = render_file(scene, out_path=str(tmp_path / "b.png"), cache=cache, size=(160, 120))
    assert np.array_equal(a, b) and any("Scene cache" in l for l in lines)
    text = open(scene).read().replace("maxdepth 5", "maxdepth 0")
    assert text != open(src).read()
    open(scene, "w").write(text)
    c, _, lines = render_file(scene, out_path=str(tmp_path / "c.png"), cache=cache, size=(160, 120))
    assert not any("Scene cache" in l for l in lines)          # built from the text again
    assert not np.array_equal(a, c)                              # and it is the edited scene that was rendered
    d, _, lines = render_file(scene, out_path=str(tmp_path / "d.png"), cache=cache, size=(160, 120))
    assert np.array_equal(c, d) and any("Scene cache" in l for l in lines)  # the cache was written anew
