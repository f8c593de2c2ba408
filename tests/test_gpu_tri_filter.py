"""GPU: the triangle test's sign filter against the C oracle, on rays aimed at the places where the filter is undecided.

The device decides "inside" from three precomputed edge functions when their values are clear of a proven error bound, and
otherwise runs the reference's operations (Triangle.cpp:147-175).  The rays here go through vertices, along edges and a
few ulps to either side of them, on triangles of very different size, shape, position and winding, so both branches and
the hand-over between them are exercised; ids and distances must equal the oracle's bit for bit.
"""
import numpy as np
import pytest

from raytracer_amd import _capi
from raytracer_amd.scene import FlatScene

pytestmark = pytest.mark.gpu


def _triangles():
    """(v1, v2, v3) float32 triples, spread along x so that their boxes do not overlap much."""
    rng = np.random.default_rng(11)
    tris = []
    x = 0.0
    def add(a, b, c, step):
        nonlocal x
        off = np.array([x, 0, 0], np.float64)
        tris.append(np.array([a + off, b + off, c + off], np.float32))
        x += step
    add(np.array([0, 0, 0.]), np.array([1, 0, 0.]), np.array([0, 1, 0.]), 3)                      # plain, axis-aligned edges
    add(np.array([0, 0, 0.]), np.array([0, 1, 0.]), np.array([1, 0, 0.]), 3)                      # the same, other winding
    add(np.array([0, 0, 0.]), np.array([1, 0, 0.3]), np.array([0.2, 1, -0.4]), 3)                 # tilted
    add(np.array([0, 0, 0.]), np.array([2, 0, 0.]), np.array([1, 2e-4, 0.]), 4)                   # sliver, 1 : 1e4
    add(np.array([0, 0, 0.]), np.array([1, 1e-5, 0.5]), np.array([2, 0, 1.0]), 4)                 # sliver, tilted
    add(np.array([0, 0, 0.]), np.array([1e-4, 0, 0.]), np.array([0, 1e-4, 1e-5]), 1)              # tiny
    add(np.array([0, 0, 0.]), np.array([40, 0, 3.]), np.array([0, 50, -2.]), 60)                  # large
    add(np.array([0, 0, 0.]), np.array([1, 0, 0.]), np.array([1, 1e-13, 0.]), 3)                  # an edge below the filter's floor
    for _ in range(8):                                                                            # random shapes and sizes
        s = 10.0 ** rng.uniform(-3, 1)
        p = rng.uniform(-1, 1, (3, 3)) * s
        add(p[0], p[1], p[2], 3 * s + 1)
    far = np.array([3000.0, -2000.0, 1500.0])                                                     # far from the origin
    add(far, far + np.array([0.5, 0, 0.1]), far + np.array([0, 0.5, 0.2]), 3)
    return tris


def _scene_text(tris):
    lines = ["size 8 8", "camera 0 0 50 0 0 0 0 1 0 45", "maxverts %d" % (3 * len(tris))]
    for t in tris:
        for v in t:
            lines.append("vertex %s %s %s" % tuple(repr(float(c)) for c in v))
    for i in range(len(tris)):
        lines.append("tri %d %d %d" % (3 * i, 3 * i + 1, 3 * i + 2))
    return "\n".join(lines) + "\n"


def _rays(tris):
    rng = np.random.default_rng(5)
    O, D = [], []
    for t in tris:
        a, b, c = t.astype(np.float64)
        n = np.cross(b - a, c - a)
        size = np.sqrt(np.linalg.norm(n))
        n = n / (np.linalg.norm(n) + 1e-300)
        targets = [a, b, c, (a + b + c) / 3]
        for p, q in ((a, b), (b, c), (c, a)):
            for u in (0.0, 0.125, 0.3, 0.5, 0.77, 1.0):
                targets.append(p + (q - p) * u)
        targets = np.array(targets)
        # a few ulps to either side of every target, in the triangle's plane
        side = np.cross(n, rng.normal(size=3))
        side /= np.linalg.norm(side) + 1e-300
        near = []
        for k in (-64, -8, -2, -1, 1, 2, 8, 64):
            near.append(targets + side * (np.abs(targets).max(axis=1, keepdims=True) + size) * k * 2.0 ** -24)
        targets = np.concatenate([targets] + near + [a + rng.uniform(-0.2, 1.2, (40, 1)) * (b - a) + rng.uniform(-0.2, 1.2, (40, 1)) * (c - a)])
        for dist in (0.7, 5.0, 200.0):
            for tilt in (0.0, 0.9, -2.5):
                o = (a + b + c) / 3 + n * size * dist + side * size * tilt
                d = targets - o
                O.append(np.broadcast_to(o, d.shape))
                D.append(d)                                               # un-normalised, like primary rays
                O.append(np.broadcast_to(o, d.shape))
                D.append(d / np.linalg.norm(d, axis=1, keepdims=True))    # normalised, like shadow and reflection rays
    return np.concatenate(O).astype(np.float32), np.concatenate(D).astype(np.float32)


def test_triangle_filter_matches_the_oracle_at_its_edges():
    from oracle.bind import CPort
    from raytracer_amd.device import DeviceScene
    tris = _triangles()
    fs = FlatScene.from_text(_scene_text(tris))
    o, d = _rays(tris)
    assert len(o) > 50000
    want_ids, want_dist, want_nrm, _ = CPort(fs).trace(o, d)
    assert (want_ids >= 0).sum() > len(o) // 10 and (want_ids < 0).sum() > len(o) // 10  # both outcomes are well represented
    with DeviceScene(fs, 0) as s:
        for mode in (_capi.TRACE_EXACT, _capi.TRACE_REFWALK):
            ids, dist, nrm = s.trace_rays(o, d, trace_mode=mode)
            assert np.array_equal(ids, want_ids), (mode, int((ids != want_ids).sum()))
            hit = want_ids >= 0
            assert np.array_equal(dist[hit].view(np.uint32), want_dist[hit].view(np.uint32)), mode
            assert np.array_equal(nrm[hit].view(np.uint32), want_nrm[hit].view(np.uint32)), mode
