"""GPU: the HIP path, called through the C ABI, against the C oracle and the golden vectors from the reference.

Bars: MIRT_TRACE_EXACT -- bit-exact (ids, float bit patterns of distances and normals, every output byte, ray counts).
      MIRT_TRACE_FAST  -- the tolerance asked of a non-exact mode: primary nearest-hit ids equal on >= 99.9 % of rays,
                          8-bit channels within +-1 LSB on >= 99.9 % of pixels.
"""
import numpy as np
import pytest

from conftest import SCENE_NAMES, bits
from raytracer_amd import _capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev_scenes(flat_scenes):
    from raytracer_amd.device import DeviceScene
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = DeviceScene(flat_scenes(name), 0)
        return cache[name]
    yield get
    for s in cache.values():
        s.close()


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_primary_ids_exact(name, dev_scenes, goldens):
    g = goldens(name)
    w, h = int(g["width"]), int(g["height"])
    ids, dist = dev_scenes(name).trace_primary_ids(w, h)
    assert np.array_equal(ids, g["prim_id"])
    assert np.array_equal(bits(dist), bits(g["dist"]))


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_secondary_rays_exact(name, dev_scenes, goldens):
    g = goldens(name)
    if g["sec_origins"].shape[0] == 0:
        pytest.skip("no hits")
    ids, dist, nrm = dev_scenes(name).trace_rays(g["sec_origins"], g["sec_dirs"])
    assert np.array_equal(ids, g["sec_prim_id"])
    assert np.array_equal(bits(dist), bits(g["sec_dist"]))
    hit = ids >= 0
    assert np.array_equal(bits(nrm[hit]), bits(g["sec_normal"][hit]))


@pytest.mark.parametrize("name", SCENE_NAMES)
@pytest.mark.parametrize("features", [1, 3, 7, 11, 15])
def test_render_exact_all_feature_combos(name, features, dev_scenes, goldens):
    g = goldens(name)
    w, h = int(g["width"]), int(g["height"])
    img, st = dev_scenes(name).render(w, h, features=features)
    want = g["bgr_f%d" % features]
    diff = np.abs(img.astype(int) - want.astype(int))
    assert diff.max() == 0, "%d bytes differ (max %d)" % ((diff > 0).sum(), diff.max())
    assert [st["primary_rays"], st["shadow_rays"], st["reflection_rays"]] == g["rays_f%d" % features].tolist()


@pytest.mark.parametrize("refill_min", [8, 40])
@pytest.mark.parametrize("name", ["scene5", "scene6", "scene7"])
def test_lane_refilling_variant_exact(name, refill_min, flat_scenes, goldens, monkeypatch):
    """The library picks the lane-refilling kernel variant for big scenes only; MIRT_REFILL_MIN forces it (lanes are
    refilled once that many are idle), so the small scenes cover it too: all features and two debug views."""
    from raytracer_amd.device import DeviceScene
    monkeypatch.setenv("MIRT_REFILL_MIN", str(refill_min))
    g = goldens(name)
    w, h = int(g["width"]), int(g["height"])
    with DeviceScene(flat_scenes(name), 0) as s:
        img, st = s.render(w, h, features=15)
        assert np.array_equal(img, g["bgr_f15"])
        assert [st["primary_rays"], st["shadow_rays"], st["reflection_rays"]] == g["rays_f15"].tolist()
        for debug in (_capi.DEBUG_SHADOW_MAP, _capi.DEBUG_PRIMARY_MAP):
            img, _ = s.render(w, h, debug=debug)
            assert np.array_equal(img, g["bgr_debug%d" % debug])


@pytest.mark.parametrize("name", ["scene5", "scene6", "scene7"])
def test_shadow_queries_with_zero_light_terms_are_not_walked(name, flat_scenes, goldens, monkeypatch):
    """A shadow query whose light term is multiplied by exact zeros (kd == 0 and ks == 0) is answered without a walk: the
    image and the reference's ray counts stay the same, only shadow_rays_traced drops; MIRT_SHADOW_CULL=0 walks them all."""
    from raytracer_amd.device import DeviceScene
    g = goldens(name)
    w, h = int(g["width"]), int(g["height"])
    traced = {}
    for cull in ("1", "0"):
        monkeypatch.setenv("MIRT_SHADOW_CULL", cull)
        with DeviceScene(flat_scenes(name), 0) as s:
            for features in (15, 7, 11):
                img, st = s.render(w, h, features=features)
                assert np.array_equal(img, g["bgr_f%d" % features]), (cull, features)
                assert [st["primary_rays"], st["shadow_rays"], st["reflection_rays"]] == g["rays_f%d" % features].tolist()
                traced[cull, features] = st["shadow_rays_traced"]
                if cull == "0" and features & 4:  # with SHADOWS off no query is walked at all
                    assert st["shadow_rays_traced"] == st["shadow_rays"]
    assert all(traced["1", f] <= traced["0", f] for f in (15, 7, 11))
    assert traced["1", 15] < traced["0", 15]


@pytest.mark.parametrize("name", ["scene1", "scene5", "scene6", "scene7"])
@pytest.mark.parametrize("debug", range(_capi.DEBUG_NONE))
def test_render_exact_debug_views(name, debug, dev_scenes, goldens):
    g = goldens(name)
    img, _ = dev_scenes(name).render(int(g["width"]), int(g["height"]), debug=debug)
    assert np.array_equal(img, g["bgr_debug%d" % debug])


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_fast_mode_within_north_star_tolerance(name, dev_scenes, goldens):
    g = goldens(name)
    w, h = int(g["width"]), int(g["height"])
    s = dev_scenes(name)
    ids, dist = s.trace_primary_ids(w, h, trace_mode=_capi.TRACE_FAST)
    # FAST skips the leaf-membership check, which is also what supplies the reference's tie-break rank: where two
    # primitives are hit at bit-identical distance (the shared diagonal of a quad) it takes the lower primitive number,
    # the reference's choice whenever one leaf holds both.
    same = ids == g["prim_id"]
    print(name, "FAST ids equal on %.5f of the primary rays" % same.mean())
    assert same.mean() >= 0.999
    assert np.array_equal(bits(dist), bits(g["dist"]))
    img, _ = s.render(w, h, trace_mode=_capi.TRACE_FAST)
    ok = (np.abs(img.astype(int) - g["bgr_f15"].astype(int)).max(axis=2) <= 1).mean()
    assert ok >= 0.999


@pytest.mark.parametrize("name", ["scene6", "scene7"])
def test_against_oracle_at_other_sizes(name, dev_scenes, flat_scenes):
    """Sizes the fixtures do not cover (non-multiple-of-tile, portrait), checked against the C oracle live."""
    from oracle.bind import CPort
    port = CPort(flat_scenes(name))
    for w, h in [(200, 150), (97, 131)]:
        want, st0 = port.render(w, h)
        img, st = dev_scenes(name).render(w, h)
        assert np.array_equal(img, want), (w, h)
        assert st["shadow_rays"] == st0["shadow_rays"] and st["reflection_rays"] == st0["reflection_rays"]


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_refwalk_mode_exact(name, dev_scenes, goldens):
    """The tree handed over, walked as it is (fallback for non-nested trees), is bit-exact too."""
    g = goldens(name)
    w, h = int(g["width"]), int(g["height"])
    s = dev_scenes(name)
    ids, dist = s.trace_primary_ids(w, h, trace_mode=_capi.TRACE_REFWALK)
    assert np.array_equal(ids, g["prim_id"]) and np.array_equal(bits(dist), bits(g["dist"]))
    img, _ = s.render(w, h, trace_mode=_capi.TRACE_REFWALK)
    assert np.array_equal(img, g["bgr_f15"])


@pytest.mark.parametrize("name,size", [("scene7", (1920, 1080)), ("scene6", (640, 480)), ("scene5", (960, 540))])
def test_hierarchy_equals_tree_walk_at_full_size(name, size, dev_scenes):
    """Size-independent property at sizes the CPU oracle cannot finish quickly: the primitive hierarchy with the
    leaf-membership check (EXACT) and the walk of the reference tree itself (REFWALK) must agree on every primary
    ray's id and distance bits and on every output byte."""
    s = dev_scenes(name)
    w, h = size
    a_ids, a_dist = s.trace_primary_ids(w, h, trace_mode=_capi.TRACE_EXACT)
    b_ids, b_dist = s.trace_primary_ids(w, h, trace_mode=_capi.TRACE_REFWALK)
    assert np.array_equal(a_ids, b_ids)
    assert np.array_equal(bits(a_dist), bits(b_dist))
    a_img, a_st = s.render(w, h, trace_mode=_capi.TRACE_EXACT)
    b_img, b_st = s.render(w, h, trace_mode=_capi.TRACE_REFWALK)
    assert np.array_equal(a_img, b_img)
    for k in ("primary_rays", "shadow_rays", "reflection_rays", "primary_hits"):
        assert a_st[k] == b_st[k]


def test_tiles_cover_image_once(dev_scenes, goldens):
    """Interleaved tiles (the multi-GPU partition) reassemble to the single-launch image."""
    g = goldens("scene6")
    w, h = int(g["width"]), int(g["height"])
    s = dev_scenes("scene6")
    acc = np.zeros((h, w, 3), np.uint8)
    total = 0
    for r in range(3):
        part = np.zeros((h, w, 3), np.uint8)
        _, st = s.render(w, h, out=part, tile_w=32, tile_h=16, first_tile=r, tile_stride=3)
        assert not (acc.astype(bool) & part.astype(bool)).any()
        acc |= part
        total += st["primary_rays"]
    assert total == w * h
    assert np.array_equal(acc, g["bgr_f15"])


def test_render_twice_is_deterministic(dev_scenes):
    s = dev_scenes("scene7")
    a, _ = s.render(160, 120)
    b, _ = s.render(160, 120)
    assert np.array_equal(a, b)


def test_errors_are_reported_not_fatal(dev_scenes):
    from raytracer_amd.device import MirtError
    s = dev_scenes("scene1")
    with pytest.raises(MirtError):
        s.render(64, 64, tile_w=12, tile_h=8, tile_stride=2)
    with pytest.raises(MirtError):
        s.render(64, 64, debug=9)


class _Hip:
    """Just enough of the HIP runtime (the copy libmirt.so already loaded) to own a device buffer in a test."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.rt = C.CDLL("libamdhip64.so")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]

    def alloc(self, n):
        p = self.C.c_void_p()
        assert self.rt.hipMalloc(self.C.byref(p), n) == 0
        assert self.rt.hipMemset(p, 0, n) == 0
        return p.value

    def to_host(self, ptr, n):
        out = np.zeros(n, np.uint8)
        assert self.rt.hipDeviceSynchronize() == 0
        assert self.rt.hipMemcpy(out.ctypes.data, ptr, n, 2) == 0  # hipMemcpyDeviceToHost
        return out

    def free(self, ptr):
        self.rt.hipFree(ptr)


@pytest.mark.parametrize("world,size", [(2, (160, 120)), (3, (150, 70)), (8, (640, 360))])
def test_packed_tiles_and_device_assembly(world, size, dev_scenes):
    """The multi-GPU data path on one device: every rank's mirt_render_device output (packed tiles), laid back to back
    as the gather delivers them, reassembled by mirt_assemble_tiles, equals the single-launch image."""
    from raytracer_amd import multigpu
    s = dev_scenes("scene6")
    hip = _Hip()
    w, h = size
    want, _ = s.render(w, h)
    nbytes = multigpu.per_rank_bytes(w, h, world)
    gathered = hip.alloc(world * nbytes)
    image = hip.alloc(w * h * 3)
    for r in range(world):
        p = multigpu.rank_params(w, h, r, world)
        assert s.tile_buffer_bytes(p) == nbytes
        s.render_device(p, gathered + r * nbytes, nbytes, 0)
    s.assemble_tiles(multigpu.rank_params(w, h, 0, world), world, gathered, nbytes, image, 0)
    got = hip.to_host(image, w * h * 3).reshape(h, w, 3)
    assert np.array_equal(got, want)
    # host-side statement of the same reassembly agrees
    bufs = hip.to_host(gathered, world * nbytes).reshape(world, nbytes)
    assert np.array_equal(multigpu.assemble_host(list(bufs), w, h, world), want)
    hip.free(gathered)
    hip.free(image)


def test_fast_mode_does_not_trace_weightless_reflections(dev_scenes, goldens):
    """scene6 (Cornell) has materials without a specular part: MIRT_TRACE_FAST ends their paths instead of tracing the
    reflection ray the reference traces (main.cpp:190); the reference's counts stay, the image stays, fewer rays are walked."""
    g = goldens("scene6")
    w, h = int(g["width"]), int(g["height"])
    s = dev_scenes("scene6")
    _, exact = s.render(w, h)
    img, fast = s.render(w, h, trace_mode=_capi.TRACE_FAST)
    assert exact["reflection_rays_traced"] == exact["reflection_rays"] == int(g["rays_f15"][2])
    assert fast["reflection_rays_traced"] < fast["reflection_rays"]
    assert (np.abs(img.astype(int) - g["bgr_f15"].astype(int)).max(axis=2) <= 1).mean() >= 0.999
