"""GPU: the benchmark configurations at their FULL sizes, scene text -> the library's own loader -> device -> image,
against outputs of the reference itself at those sizes (tests/golden/c*.golden.npz, scripts/make_golden_configs.py).

Nothing of the reference's pipeline is involved on this side: the scene file is parsed and its Partition tree built by
mirt_host_scene_parse, uploaded by mirt_scene_create and rendered by mirt_render.  Bar: every byte, id, distance bit
pattern and ray count equal."""
import numpy as np
import pytest

from conftest import bits
from raytracer_amd import _capi, workloads
from raytracer_amd.scene import FlatScene

pytestmark = pytest.mark.gpu


def _device(text):
    from raytracer_amd.device import DeviceScene
    return DeviceScene(FlatScene.from_text(text), 0)


def _rays(st):
    return [st["primary_rays"], st["shadow_rays"], st["reflection_rays"]]


def test_c1_cornell_640x480(goldens):
    g = goldens("c1_cornell")
    with _device(workloads.cornell()) as s:
        img, st = s.render(640, 480)
        assert np.array_equal(img, g["bgr"])
        assert _rays(st) == g["rays"].tolist() == [307200, 1712796, 842329]
        ids, dist = s.trace_primary_ids(640, 480)
        assert np.array_equal(ids.reshape(480, 640), g["prim_id"])
        assert np.array_equal(bits(dist).reshape(480, 640), bits(g["dist"]))


def test_c2_spheres_and_ellipsoids_1920x1080(goldens):
    g = goldens("c2_spheres")
    with _device(workloads.spheres()) as s:
        img, st = s.render(1920, 1080)
        assert np.array_equal(img, g["bgr"])
        assert _rays(st) == g["rays"].tolist()
        ids, _ = s.trace_primary_ids(1920, 1080)
        assert np.array_equal(ids.reshape(1080, 1920), g["prim_id"])
        assert (ids >= 1000).sum() > 20000  # the ellipsoids (primitive ids 1000..1199) are really in view
        img, st = s.render(480, 270)
        assert np.array_equal(img, g["bgr_small"])
        assert _rays(st) == g["rays_small"].tolist()
        ids, dist = s.trace_primary_ids(480, 270)
        assert np.array_equal(ids.reshape(270, 480), g["prim_id_small"])
        assert np.array_equal(bits(dist).reshape(270, 480), bits(g["dist_small"]))
        # the library's hierarchy + membership check against a plain walk of the reference's tree, every pixel
        ref_img, _ = s.render(1920, 1080, trace_mode=_capi.TRACE_REFWALK)
        assert np.array_equal(ref_img, g["bgr"])


@pytest.mark.parametrize("tag, w, h", [("4k", 3840, 2160), ("8k", 7680, 4320)])
def test_c3_c4_dragon_rows_of_the_full_size_image(tag, w, h, goldens):
    g = goldens("c3_dragon")
    with _device(workloads.dragon(w, h)) as s:
        img, st = s.render(w, h)
        rows = g["rows_" + tag]
        assert np.array_equal(img[rows], g["bgr_rows_" + tag])
        assert st["primary_rays"] == w * h
        # nearest-hit primitive and distance bits of every primary ray of those rows
        ids, dist = s.trace_primary_ids(w, h)
        assert np.array_equal(ids.reshape(h, w)[rows], g["prim_id_rows_" + tag])
        assert np.array_equal(bits(dist).reshape(h, w)[rows], bits(g["dist_rows_" + tag]))
        # ray counts: the device renders tiles (multiples of 8 rows), so the 8-row band around each golden row is rendered
        # tile by tile (1024x8 tiles, one tile per call) and its counts compared with the reference's for the same band
        tiles_x = (w + 1023) // 1024
        n_tiles = tiles_x * ((h + 7) // 8)
        for r, want in zip(rows, g["rays_bands_" + tag]):
            got = np.zeros(3, np.uint64)
            for tx in range(tiles_x):
                _, bst = s.render(w, h, out=img, tile_w=1024, tile_h=8, first_tile=(int(r) // 8) * tiles_x + tx, tile_stride=n_tiles)
                got += np.array(_rays(bst), np.uint64)
            assert got.tolist() == want.tolist(), (tag, int(r))
        assert np.array_equal(img[rows], g["bgr_rows_" + tag])  # the tile renders wrote the same bytes into the image
        if tag == "4k":
            ref_img, _ = s.render(w, h, trace_mode=_capi.TRACE_REFWALK)
            assert np.array_equal(ref_img, img)


def test_c5_ten_million_triangles_sixteen_lights(goldens):
    """workloads.dragon_grid(): 100 dragons, 16 point lights, maxdepth 8.  The reference needs 227 s to parse the text
    and build its tree (one thread); the digests below are of ITS flattened tree, so equality means the library's
    loader produced the same 19.6 M nodes and 48.7 M leaf entries bit for bit."""
    import hashlib
    g = goldens("c5_grid")
    fs = FlatScene.from_text(workloads.dragon_grid())
    assert fs.truncated_nodes == 0
    assert (fs.nodes.shape[0], fs.leaf_prims.shape[0], fs.n_prims) == (int(g["n_nodes"]), int(g["n_leaf_prims"]), int(g["n_prims"]))
    for name, arr in (("sha_nodes", fs.nodes), ("sha_leaf_prims", fs.leaf_prims), ("sha_tri_verts", fs.tri_verts)):
        assert hashlib.sha256(arr.tobytes()).digest() == g[name].tobytes(), name
    from raytracer_amd.device import DeviceScene
    with DeviceScene(fs, 0) as s:
        img, st = s.render(480, 270)
        assert np.array_equal(img, g["bgr_small"])
        assert _rays(st) == g["rays_small"].tolist()
        ids, dist = s.trace_primary_ids(480, 270)
        assert np.array_equal(ids.reshape(270, 480), g["prim_id_small"])
        assert np.array_equal(bits(dist).reshape(270, 480), bits(g["dist_small"]))
        img, st = s.render(3840, 2160)
        assert np.array_equal(img[g["rows_4k"]], g["bgr_rows_4k"])
        assert st["primary_rays"] == 3840 * 2160 and st["shadow_rays"] == 16 * (st["shadow_rays"] // 16)


def test_scene_file_to_png_end_to_end(tmp_path, goldens, monkeypatch):
    """python -m raytracer_amd: the whole of createRender (main.cpp:104-173) -- load, render, save, report."""
    import os
    from PIL import Image
    from raytracer_amd.__main__ import render_file
    monkeypatch.chdir(tmp_path)
    scene = os.path.join(os.path.dirname(os.path.dirname(__file__)), "raytracer_amd", "data", "scenes", "scene6.test")
    img, st, lines = render_file(scene, out_path=str(tmp_path / "cornell.png"))
    g = goldens("c1_cornell")
    assert np.array_equal(img, g["bgr"])
    assert np.array_equal(np.array(Image.open(tmp_path / "cornell.png")), g["bgr"][..., ::-1])
    assert "Resolution: 640x480" in lines and "----- Spheres: 4" in lines and "----- Triangles: 34" in lines
    assert (tmp_path / "reports" / "cornell.png_report.txt").exists()
