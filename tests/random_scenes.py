"""Seeded random scene texts in the reference's scene language (Scene.cpp:15-224), for the cross-checks that no fixture
covers: every geometry command, nested transforms, both kinds of light, every material command, attenuation, odd image
sizes, and -- on purpose -- the constructions in which two primitives answer a ray at the same distance (the same triangle
twice, coplanar quads that overlap, quads that share an edge a pixel centre falls on, concentric spheres), because the
reference's answer then depends on the order in which its tree hands the primitives out (Partition.cpp:47, Partition.h:31)."""
import numpy as np


def _f(x):
    return "%.6g" % float(x)


def _vec(v):
    return " ".join(_f(x) for x in v)


def random_scene_text(seed):
    rng = np.random.default_rng(seed)
    kind = seed % 6
    w, h = int(rng.integers(17, 97)), int(rng.integers(9, 73))
    out = ["size %d %d" % (w, h), "maxdepth %d" % int(rng.integers(0, 7))]
    eye = rng.uniform(-1.0, 1.0, 3) + np.array([0.0, 0.0, rng.uniform(4.0, 9.0)])
    out.append("camera %s %s 0 1 0 %s" % (_vec(eye), _vec(rng.uniform(-0.5, 0.5, 3)), _f(rng.uniform(25, 70))))
    if rng.random() < 0.5:
        out.append("attenuation %s %s %s" % (_f(rng.uniform(0.5, 1.5)), _f(rng.uniform(0, 0.2)), _f(rng.uniform(0, 0.05))))
    for _ in range(int(rng.integers(1, 5))):
        if rng.random() < 0.6:
            out.append("point %s %s" % (_vec(rng.uniform(-6, 6, 3) + np.array([0, 3, 3])), _vec(rng.uniform(0.2, 1.0, 3))))
        else:
            out.append("directional %s %s" % (_vec(rng.uniform(-1, 1, 3) + np.array([0, 0.5, 1.5])), _vec(rng.uniform(0.1, 0.7, 3))))

    def material():
        out.append("ambient %s" % _vec(rng.uniform(0, 0.25, 3)))
        out.append("diffuse %s" % _vec(rng.uniform(0, 1, 3)))
        # (black speculars end reflection paths in MIRT_TRACE_FAST only; the reference follows them)
        out.append("specular %s" % (_vec(rng.uniform(0, 0.9, 3)) if rng.random() < 0.8 else "0 0 0"))
        out.append("shininess %s" % _f(rng.choice([1, 5, 20, 64, 200])))
        out.append("emission %s" % (_vec(rng.uniform(0, 0.2, 3)) if rng.random() < 0.3 else "0 0 0"))

    def transform():
        for _ in range(int(rng.integers(0, 4))):
            c = rng.integers(0, 3)
            if c == 0:
                out.append("translate %s" % _vec(rng.uniform(-1.5, 1.5, 3)))
            elif c == 1:
                out.append("rotate %s %s" % (_vec(rng.uniform(-1, 1, 3) + np.array([0, 0, 1.5])), _f(rng.uniform(-180, 180))))
            else:
                out.append("scale %s" % _vec(rng.uniform(0.4, 1.8, 3)))

    verts = []
    tris = []  # (text line index is fixed later: vertices come first, as in the reference's scenes)

    def vertex(p):
        verts.append("vertex %s" % _vec(p))
        return len(verts) - 1

    body = []
    saved, out = out, body  # geometry and materials go to `body`; the vertex pool is put in front of it at the end

    def quad(p0, e1, e2):
        a, b, c, d = vertex(p0), vertex(p0 + e1), vertex(p0 + e1 + e2), vertex(p0 + e2)
        out.append("tri %d %d %d" % (a, b, c))
        out.append("tri %d %d %d" % (a, c, d))
        return a, b, c, d

    if kind in (0, 3, 5):  # a soup of triangles under nested transforms
        material()
        for g in range(int(rng.integers(1, 5))):
            out.append("pushTransform")
            transform()
            if rng.random() < 0.5:
                material()
            for _ in range(int(rng.integers(4, 90))):
                c = rng.uniform(-2.5, 2.5, 3)
                idx = [vertex(c + rng.uniform(-0.8, 0.8, 3)) for _ in range(3)]
                out.append("tri %d %d %d" % tuple(idx))
            if rng.random() < 0.5:  # nested
                out.append("pushTransform")
                transform()
                out.append("sphere %s %s" % (_vec(rng.uniform(-1, 1, 3)), _f(rng.uniform(0.2, 0.9))))
                out.append("popTransform")
            out.append("popTransform")
    if kind in (1, 3):  # spheres: plain, under rotations and non-uniform scales, concentric pairs
        for _ in range(int(rng.integers(2, 25))):
            material()
            out.append("pushTransform")
            transform()
            c, r = rng.uniform(-2, 2, 3), rng.uniform(0.15, 0.9)
            out.append("sphere %s %s" % (_vec(c), _f(r)))
            if rng.random() < 0.2:
                out.append("sphere %s %s" % (_vec(c), _f(r)))  # the same sphere again: every hit is a tie
            out.append("popTransform")
    if kind in (2, 4, 5):  # ties between triangles
        material()
        z = float(rng.integers(-2, 2)) * 0.5
        # a floor of quads whose shared edges lie on exactly representable coordinates
        n = int(rng.integers(2, 6))
        for i in range(n):
            for j in range(n):
                if rng.random() < 0.3:
                    material()
                quad(np.array([-2.0 + i * 4.0 / n, -2.0 + j * 4.0 / n, z]), np.array([4.0 / n, 0, 0]), np.array([0, 4.0 / n, 0]))
        # coplanar quads over it, partly overlapping, and one of them twice
        for _ in range(int(rng.integers(1, 4))):
            material()
            p0 = np.array([float(rng.integers(-4, 3)) * 0.5, float(rng.integers(-4, 3)) * 0.5, z])
            a, b, c, d = quad(p0, np.array([1.0, 0, 0]), np.array([0, 1.0, 0]))
            if rng.random() < 0.5:
                material()
                out.append("tri %d %d %d" % (a, b, c))
        # a box in front of it: reflections and shadows between parallel faces
        out.append("pushTransform")
        out.append("translate %s" % _vec([rng.uniform(-1, 1), rng.uniform(-1, 1), z + 1.0]))
        if rng.random() < 0.5:
            out.append("rotate 0 0 1 %s" % _f(float(rng.integers(0, 8)) * 45.0))
        material()
        s = 0.5
        quad(np.array([-s, -s, s]), np.array([2 * s, 0, 0]), np.array([0, 2 * s, 0]))
        quad(np.array([-s, -s, -s]), np.array([0, 2 * s, 0]), np.array([2 * s, 0, 0]))
        quad(np.array([-s, -s, -s]), np.array([2 * s, 0, 0]), np.array([0, 0, 2 * s]))
        quad(np.array([-s, s, -s]), np.array([0, 0, 2 * s]), np.array([2 * s, 0, 0]))
        quad(np.array([-s, -s, -s]), np.array([0, 0, 2 * s]), np.array([0, 2 * s, 0]))
        quad(np.array([s, -s, -s]), np.array([0, 2 * s, 0]), np.array([0, 0, 2 * s]))
        out.append("popTransform")
    if kind == 4:  # a mirror behind the camera's target and a sphere touching the floor
        material()
        out.append("specular 0.9 0.9 0.9")
        quad(np.array([-3.0, -3.0, -3.0]), np.array([6.0, 0, 0]), np.array([0, 6.0, 0]))
        out.append("sphere 0 0 %s 0.5" % _f(-1.5))
    out = saved
    out.append("maxverts %d" % max(1, len(verts)))
    out += verts
    out += body
    return "\n".join(out) + "\n"


def hard_scene_text(seed):
    """The unkind cases: axis-parallel camera rays over axis-aligned geometry (direction components that are exactly zero or
    below the 1e-5 that AABB.cpp:53-58 replaces), coordinates far from the origin, slivers and specks beside huge triangles, the
    camera inside a sphere or between two mirrors, lights on surfaces and inside objects, one-pixel-wide images, meshes of a few
    thousand triangles with shared vertices (deeper trees, long walks)."""
    rng = np.random.default_rng(100000 + seed)
    kind = seed % 6
    out = []
    verts = []
    body = []

    def vertex(p):
        verts.append("vertex %s" % _vec(p))
        return len(verts) - 1

    def material(spec=None):
        body.append("ambient %s" % _vec(rng.uniform(0, 0.2, 3)))
        body.append("diffuse %s" % _vec(rng.uniform(0.1, 1, 3)))
        body.append("specular %s" % _vec(rng.uniform(0.2, 0.95, 3) if spec is None else spec))
        body.append("shininess %s" % _f(rng.choice([0.5, 2, 30, 128])))

    def quad(p0, e1, e2):
        a, b, c, d = vertex(p0), vertex(p0 + e1), vertex(p0 + e1 + e2), vertex(p0 + e2)
        body.append("tri %d %d %d" % (a, b, c))
        body.append("tri %d %d %d" % (a, c, d))

    def mesh(n, m, f, flip=False):
        """an n x m grid of vertices f(u, v), two triangles per cell, vertices shared"""
        idx = [[vertex(f(i / (n - 1.0), j / (m - 1.0))) for j in range(m)] for i in range(n)]
        for i in range(n - 1):
            for j in range(m - 1):
                a, b, c, d = idx[i][j], idx[i + 1][j], idx[i + 1][j + 1], idx[i][j + 1]
                body.append("tri %d %d %d" % ((a, c, b) if flip else (a, b, c)))
                body.append("tri %d %d %d" % ((a, d, c) if flip else (a, c, d)))

    if kind == 0:
        # camera on an axis, looking along it, at axis-aligned quads: whole rows and columns of rays have a zero component
        w, h = int(rng.integers(8, 40)) * 2 + 1, int(rng.integers(8, 30)) * 2 + 1
        out += ["size %d %d" % (w, h), "maxdepth %d" % int(rng.integers(0, 5)), "camera 0 0 %s 0 0 0 0 1 0 %s" % (_f(rng.integers(3, 8)), _f(rng.choice([30, 45, 60, 90])))]
        out.append("point 0 0 %s 1 1 1" % _f(rng.integers(1, 4)))
        out.append("directional 0 0 1 0.3 0.3 0.3")
        out.append("directional 0 1 0 0.2 0.2 0.2")
        material()
        n = int(rng.integers(1, 5))
        for i in range(-n, n):
            for j in range(-n, n):
                if rng.random() < 0.2:
                    material()
                quad(np.array([float(i), float(j), -float(rng.integers(0, 3))]), np.array([1.0, 0, 0]), np.array([0, 1.0, 0]))
        for _ in range(int(rng.integers(0, 4))):  # walls parallel to the viewing direction
            x = float(rng.integers(-n, n + 1))
            quad(np.array([x, -float(n), -2.0]), np.array([0, 2.0 * n, 0]), np.array([0, 0, 3.0]))
        body.append("sphere 0 0 0 %s" % _f(rng.choice([0.25, 0.5, 1.0])))
    elif kind == 1:
        # far from the origin, and very different sizes side by side
        off = rng.choice([1e2, 1e3, 1e4]) * rng.choice([-1.0, 1.0], 3)
        w, h = int(rng.integers(20, 90)), int(rng.integers(12, 60))
        out += ["size %d %d" % (w, h), "maxdepth %d" % int(rng.integers(0, 6))]
        out.append("camera %s %s 0 1 0 %s" % (_vec(off + np.array([0.3, 0.2, 6.0])), _vec(off), _f(rng.uniform(30, 60))))
        out.append("point %s 1 1 1" % _vec(off + np.array([2.0, 4.0, 5.0])))
        out.append("directional 0.3 0.5 1 0.4 0.4 0.4")
        material()
        quad(off + np.array([-50.0, -50.0, -1.0]), np.array([100.0, 0, 0]), np.array([0, 100.0, 0]))  # huge
        for _ in range(int(rng.integers(10, 120))):
            c = off + rng.uniform(-2, 2, 3)
            s = rng.choice([1e-3, 1e-2, 0.1, 0.6])
            if rng.random() < 0.3:
                material()
            idx = [vertex(c + rng.uniform(-s, s, 3)) for _ in range(3)]
            body.append("tri %d %d %d" % tuple(idx))
        for _ in range(int(rng.integers(0, 6))):  # slivers
            c = off + rng.uniform(-2, 2, 3)
            e = rng.uniform(-1, 1, 3)
            a, b, cc = vertex(c), vertex(c + e), vertex(c + e * 0.5 + rng.uniform(-1e-4, 1e-4, 3))
            body.append("tri %d %d %d" % (a, b, cc))
        body.append("pushTransform")
        body.append("translate %s" % _vec(off))
        body.append("sphere 0 0 0 %s" % _f(rng.uniform(0.3, 1.0)))
        body.append("scale 1 %s 1" % _f(rng.choice([0.05, 3.0])))
        body.append("sphere 1.5 0 0 0.5")
        body.append("popTransform")
    elif kind == 2:
        # camera inside a big sphere, mirrors and lights inside with it; a light exactly on a surface
        w, h = int(rng.integers(20, 70)), int(rng.integers(12, 50))
        out += ["size %d %d" % (w, h), "maxdepth %d" % int(rng.integers(2, 13))]
        out.append("camera %s %s 0 1 0 %s" % (_vec(rng.uniform(-0.5, 0.5, 3)), _vec(rng.uniform(-1, 1, 3) + np.array([0, 0, -3.0])), _f(rng.uniform(40, 100))))
        out.append("point %s 1 0.9 0.8" % _vec(rng.uniform(-1, 1, 3)))
        out.append("point 0 -1 0 0.5 0.5 0.5")  # on the floor quad below
        if rng.random() < 0.5:
            out.append("attenuation 1 0.05 0.01")
        material()
        body.append("sphere 0 0 0 %s" % _f(rng.uniform(4.0, 9.0)))
        material(spec=[0.9, 0.9, 0.9])
        quad(np.array([-2.0, -1.0, -2.0]), np.array([4.0, 0, 0]), np.array([0, 0, 4.0]))
        quad(np.array([-2.0, 1.0, -2.0]), np.array([0, 0, 4.0]), np.array([4.0, 0, 0]))
        for _ in range(int(rng.integers(1, 8))):
            material()
            body.append("sphere %s %s" % (_vec(rng.uniform(-1.5, 1.5, 3)), _f(rng.uniform(0.1, 0.6))))
    elif kind == 3:
        # a few thousand triangles with shared vertices: a bumpy sheet and a tube around the viewing axis
        w, h = int(rng.integers(40, 100)), int(rng.integers(30, 70))
        out += ["size %d %d" % (w, h), "maxdepth %d" % int(rng.integers(1, 5))]
        out.append("camera %s 0 0 0 0 1 0 %s" % (_vec(rng.uniform(-1, 1, 3) + np.array([0, 1.0, 7.0])), _f(rng.uniform(35, 60))))
        for _ in range(int(rng.integers(1, 4))):
            out.append("point %s %s" % (_vec(rng.uniform(-5, 5, 3) + np.array([0, 4, 2])), _vec(rng.uniform(0.3, 1, 3))))
        material()
        n = int(rng.integers(12, 40))
        fx, fy, amp = rng.uniform(1, 6), rng.uniform(1, 6), rng.uniform(0.05, 0.5)
        mesh(n, n, lambda u, v: np.array([-3 + 6 * u, amp * np.sin(fx * u * 6.28) * np.cos(fy * v * 6.28) - 1.0, -3 + 6 * v]))
        material()
        body.append("pushTransform")
        body.append("rotate 0 1 0 %s" % _f(rng.uniform(0, 360)))
        r0 = rng.uniform(0.3, 1.0)
        mesh(int(rng.integers(8, 30)), int(rng.integers(8, 30)), lambda u, v: np.array([r0 * np.cos(6.283185 * u), -1.0 + 2.5 * v, r0 * np.sin(6.283185 * u)]), flip=rng.random() < 0.5)
        body.append("popTransform")
    elif kind == 4:
        # images one pixel wide or high, odd fields of view, nothing or everything in view
        w, h = (1, int(rng.integers(1, 200))) if rng.random() < 0.5 else (int(rng.integers(1, 300)), 1)
        out += ["size %d %d" % (w, h), "maxdepth %d" % int(rng.integers(0, 4))]
        out.append("camera 0 0 3 0 0 0 0 1 0 %s" % _f(rng.choice([1, 10, 120, 170])))
        out.append("point 1 1 3 1 1 1")
        material()
        if rng.random() < 0.8:
            quad(np.array([-1.0, -1.0, 0.0]), np.array([2.0, 0, 0]), np.array([0, 2.0, 0]))
        if rng.random() < 0.5:
            body.append("sphere 0 0 0.5 0.4")
    else:
        # many lights, some inside objects; primitives that pass through one another
        w, h = int(rng.integers(20, 80)), int(rng.integers(12, 60))
        out += ["size %d %d" % (w, h), "maxdepth %d" % int(rng.integers(0, 6))]
        out.append("camera %s 0 0 0 0 1 0 %s" % (_vec(rng.uniform(-2, 2, 3) + np.array([0, 0, 6.0])), _f(rng.uniform(30, 60))))
        for _ in range(int(rng.integers(5, 20))):
            if rng.random() < 0.7:
                out.append("point %s %s" % (_vec(rng.uniform(-2, 2, 3)), _vec(rng.uniform(0.05, 0.3, 3))))
            else:
                out.append("directional %s %s" % (_vec(rng.choice([-1.0, 0.0, 1.0], 3) + np.array([0, 0, 1e-6])), _vec(rng.uniform(0.05, 0.2, 3))))
        for _ in range(int(rng.integers(3, 12))):
            material()
            c = rng.uniform(-1.5, 1.5, 3)
            body.append("sphere %s %s" % (_vec(c), _f(rng.uniform(0.3, 1.2))))
            quad(c + np.array([-1.0, -1.0, 0.0]), np.array([2.0, 0, 0]), np.array([0, 2.0, 0]))  # through the sphere's centre
    out.append("maxverts %d" % max(1, len(verts)))
    return "\n".join(out + verts + body) + "\n"


def nasty_scene_text(seed):
    """Inputs on the edge of the scene language's meaning: a light direction of zero, an up vector along the view, fields of view of
    0.001 and 179.999 degrees, zero and negative radii, mirrored and nearly flat transforms, shininess 0 and negative, light colours
    that are negative or overflow, an attenuation that is zero everywhere, speculars above 1 at maxdepth 30, hundreds of coplanar
    triangles answering at one distance.  What the reference makes of them is arbitrary but it is a function of the input, and
    the library claims the same function."""
    rng = np.random.default_rng(200000 + seed)
    kind = seed % 8
    out, verts, body = [], [], []

    def vertex(p):
        verts.append("vertex %s" % _vec(p))
        return len(verts) - 1

    def quad(p0, e1, e2):
        a, b, c, d = vertex(p0), vertex(p0 + e1), vertex(p0 + e1 + e2), vertex(p0 + e2)
        body.append("tri %d %d %d" % (a, b, c))
        body.append("tri %d %d %d" % (a, c, d))

    def material():
        body.append("ambient %s" % _vec(rng.uniform(0, 0.2, 3)))
        body.append("diffuse %s" % _vec(rng.uniform(0.1, 1, 3)))
        body.append("specular %s" % _vec(rng.uniform(0.0, 0.9, 3)))
        body.append("shininess %s" % _f(rng.choice([1, 10, 100])))

    def some_geometry():
        material()
        quad(np.array([-3.0, -1.0, -3.0]), np.array([6.0, 0, 0]), np.array([0, 0, 6.0]))
        for _ in range(int(rng.integers(1, 6))):
            material()
            body.append("sphere %s %s" % (_vec(rng.uniform(-1.5, 1.5, 3)), _f(rng.uniform(0.2, 0.8))))
        for _ in range(int(rng.integers(0, 20))):
            c = rng.uniform(-2, 2, 3)
            idx = [vertex(c + rng.uniform(-0.7, 0.7, 3)) for _ in range(3)]
            body.append("tri %d %d %d" % tuple(idx))

    w, h = int(rng.integers(16, 64)), int(rng.integers(12, 48))
    out += ["size %d %d" % (w, h), "maxdepth %d" % int(rng.integers(0, 6))]
    cam = "camera %s 0 0 0 0 1 0 %s" % (_vec(rng.uniform(-1, 1, 3) + np.array([0, 0.5, 5.0])), _f(rng.uniform(30, 60)))
    lights = ["point %s 1 1 1" % _vec(rng.uniform(-3, 3, 3) + np.array([0, 3, 2])), "directional 0.2 1 0.5 0.4 0.4 0.4"]
    if kind == 0:  # light directions and positions without a direction
        lights = ["directional 0 0 0 0.8 0.8 0.8", "point %s 1 1 1" % _vec(rng.uniform(-3, 3, 3) + np.array([0, 3, 2]))]
        if rng.random() < 0.5:
            lights.append("directional 1e-30 0 0 0.5 0.5 0.5")  # its squared length underflows to zero
        some_geometry()
        if rng.random() < 0.3:
            body = [ln for ln in body if not ln.startswith("sphere")]  # the same without a sphere to "hit"
    elif kind == 1:  # cameras without a frame, or with hardly any field of view, or nearly all of it
        c = rng.integers(0, 4)
        if c == 0:
            cam = "camera 0 5 0 0 0 0 0 1 0 45"      # up along the view: cross(up, w) = 0
        elif c == 1:
            cam = "camera 0 0 5 0 0 5 0 1 0 45"      # looking at itself
        elif c == 2:
            cam = "camera 0 0.5 5 0 0 0 0 1 0 0.001"
        else:
            cam = "camera 0 0.5 5 0 0 0 0 1 0 179.999"
        some_geometry()
    elif kind == 2:  # radii and transforms that are hardly any
        material()
        body.append("sphere 0 0 0 0")
        body.append("sphere 1 0 0 -0.5")
        body.append("sphere -1 0 0 1e-4")
        body.append("sphere 0 0 -2000 1990")
        for sc in ("-1 1 1", "1 -1 -1", "1 1e-3 1", "1e3 1e3 1e3", "1e-3 1e-3 1e-3"):
            material()
            body.append("pushTransform")
            body.append("translate %s" % _vec(rng.uniform(-1.5, 1.5, 3)))
            body.append("scale %s" % sc)
            body.append("sphere 0 0 0 %s" % _f(rng.choice([0.5, 5e-4 if sc.startswith("1e3") else 0.5, 500.0 if sc.startswith("1e-3") else 0.5])))
            quad(np.array([-0.5, -0.5, 0.3]), np.array([1.0, 0, 0]), np.array([0, 1.0, 0]))
            body.append("popTransform")
    elif kind == 3:  # materials outside the usual ranges
        for sh in (0, -2, 0.5, 1e4, 1e-3):
            body.append("ambient %s" % _vec(rng.uniform(-0.2, 0.3, 3)))
            body.append("diffuse %s" % _vec(rng.uniform(-0.5, 2.0, 3)))
            body.append("specular %s" % _vec(rng.uniform(-0.5, 1.5, 3)))
            body.append("emission %s" % _vec(rng.uniform(-0.2, 1.5, 3)))
            body.append("shininess %s" % _f(sh))
            body.append("sphere %s %s" % (_vec(rng.uniform(-2, 2, 3)), _f(rng.uniform(0.3, 0.9))))
        quad(np.array([-3.0, -1.0, -3.0]), np.array([6.0, 0, 0]), np.array([0, 0, 6.0]))
    elif kind == 4:  # lights outside the usual ranges, attenuations that divide by zero
        lights = ["point 1 2 3 -1 0.5 2", "directional 0 1 0 1e30 1e30 1e30", "point -2 3 1 1e38 1e38 1e38", "point 0 4 0 3e38 0 0"]
        out.append("attenuation %s" % rng.choice(["0 0 0", "0 0 1", "-1 0.5 0", "1e-30 0 0", "0 1e30 1e30"]))
        some_geometry()
    elif kind == 5:  # mirrors that gain energy, followed a long way
        out[1] = "maxdepth %d" % int(rng.integers(12, 31))
        body.append("ambient 0.05 0.05 0.05")
        body.append("diffuse 0.2 0.2 0.2")
        body.append("specular %s" % _vec(rng.uniform(0.9, 1.6, 3)))
        body.append("shininess 50")
        quad(np.array([-2.0, -2.0, -2.0]), np.array([4.0, 0, 0]), np.array([0, 4.0, 0]))
        quad(np.array([-2.0, -2.0, 2.0]), np.array([0, 4.0, 0]), np.array([4.0, 0, 0]))
        quad(np.array([-2.0, -2.0, -2.0]), np.array([0, 0, 4.0]), np.array([4.0, 0, 0]))
        quad(np.array([-2.0, 2.0, -2.0]), np.array([4.0, 0, 0]), np.array([0, 0, 4.0]))
        for _ in range(int(rng.integers(1, 5))):
            body.append("sphere %s %s" % (_vec(rng.uniform(-1.2, 1.2, 3)), _f(rng.uniform(0.2, 0.6))))
        cam = "camera %s 0 0 0 0 1 0 60" % _vec(rng.uniform(-0.5, 0.5, 3) + np.array([0, 0, 1.5]))
        lights = ["point %s 1 1 1" % _vec(rng.uniform(-1, 1, 3))]
    elif kind == 6:  # hundreds of coplanar triangles over one another: one distance, many answers
        material()
        n = int(rng.integers(50, 400))
        z = float(rng.integers(-2, 2))
        for i in range(n):
            if rng.random() < 0.1:
                material()
            c = np.array([float(rng.integers(-8, 8)) * 0.25, float(rng.integers(-8, 8)) * 0.25, z])
            idx = [vertex(c + np.array([float(rng.integers(-4, 5)) * 0.25, float(rng.integers(-4, 5)) * 0.25, 0.0])) for _ in range(3)]
            body.append("tri %d %d %d" % tuple(idx))
    else:  # everything at once, under a deep stack of transforms
        material()
        depth = int(rng.integers(3, 9))
        for _ in range(depth):
            body.append("pushTransform")
            body.append("translate %s" % _vec(rng.uniform(-0.3, 0.3, 3)))
            body.append("rotate %s %s" % (_vec(rng.uniform(-1, 1, 3) + np.array([0, 0, 1.2])), _f(rng.uniform(-60, 60))))
            body.append("scale %s" % _vec(rng.choice([-1.0, 1.0], 3) * rng.uniform(0.7, 1.3, 3)))
            if rng.random() < 0.5:
                material()
            body.append("sphere %s %s" % (_vec(rng.uniform(-1, 1, 3)), _f(rng.uniform(0.2, 0.6))))
            quad(rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3))
        body += ["popTransform"] * depth
    out.append(cam)
    out += lights
    out.append("maxverts %d" % max(1, len(verts)))
    return "\n".join(out + verts + body) + "\n"


def big_scene_text(seed):
    """Tessellated surfaces of 5 000 to 60 000 triangles with shared vertices (a bumpy floor, tori, lumpy balls), a few real
    spheres, several lights: Partition trees 20 to 40 levels deep, hierarchies beyond the 4096 nodes from which walks are shared
    between lanes, walks long enough to overflow a small stack."""
    rng = np.random.default_rng(300000 + seed)
    out, verts, body = [], [], []

    def vertex(p):
        verts.append("vertex %s" % _vec(p))
        return len(verts) - 1

    def material():
        body.append("ambient %s" % _vec(rng.uniform(0, 0.15, 3)))
        body.append("diffuse %s" % _vec(rng.uniform(0.1, 0.9, 3)))
        body.append("specular %s" % _vec(rng.uniform(0.0, 0.8, 3)))
        body.append("shininess %s" % _f(rng.choice([5, 30, 120])))

    def mesh(n, m, f, wrap_u=False, wrap_v=False):
        nu, nv = (n if wrap_u else n + 1), (m if wrap_v else m + 1)
        idx = [[vertex(f(i / float(n), j / float(m))) for j in range(nv)] for i in range(nu)]
        for i in range(n):
            for j in range(m):
                a, b = idx[i][j], idx[(i + 1) % nu][j]
                c, d = idx[(i + 1) % nu][(j + 1) % nv], idx[i][(j + 1) % nv]
                body.append("tri %d %d %d" % (a, b, c))
                body.append("tri %d %d %d" % (a, c, d))

    w, h = int(rng.integers(60, 140)), int(rng.integers(40, 100))
    out += ["size %d %d" % (w, h), "maxdepth %d" % int(rng.integers(1, 6))]
    out.append("camera %s %s 0 1 0 %s" % (_vec(rng.uniform(-2, 2, 3) + np.array([0, 2.5, 9.0])), _vec(rng.uniform(-0.5, 0.5, 3)), _f(rng.uniform(30, 55))))
    for _ in range(int(rng.integers(1, 5))):
        out.append("point %s %s" % (_vec(rng.uniform(-6, 6, 3) + np.array([0, 6, 2])), _vec(rng.uniform(0.3, 0.9, 3))))
    if rng.random() < 0.5:
        out.append("directional %s 0.3 0.3 0.3" % _vec(rng.uniform(-1, 1, 3) + np.array([0, 1.5, 0.5])))
    budget = int(rng.choice([5000, 12000, 30000, 60000]))
    material()
    n = int(np.sqrt(budget * 0.3))
    fx, fz, amp = rng.uniform(0.5, 5), rng.uniform(0.5, 5), rng.uniform(0.05, 0.6)
    mesh(n, n, lambda u, v: np.array([-6 + 12 * u, amp * np.sin(fx * u * 6.28) * np.sin(fz * v * 6.28) - 1.5, -6 + 12 * v]))
    parts = int(rng.integers(2, 6))
    for _ in range(parts):
        material()
        body.append("pushTransform")
        body.append("translate %s" % _vec(rng.uniform(-3, 3, 3) + np.array([0, 0.5, 0])))
        body.append("rotate %s %s" % (_vec(rng.uniform(-1, 1, 3) + np.array([0, 1.2, 0])), _f(rng.uniform(0, 360))))
        k = int(np.sqrt(budget * 0.7 / parts / 2))
        if rng.random() < 0.5:
            R, r = rng.uniform(0.8, 1.6), rng.uniform(0.15, 0.5)
            mesh(k, k, lambda u, v: np.array([(R + r * np.cos(6.283185 * v)) * np.cos(6.283185 * u), r * np.sin(6.283185 * v), (R + r * np.cos(6.283185 * v)) * np.sin(6.283185 * u)]), True, True)
        else:
            R, lump, fl = rng.uniform(0.5, 1.3), rng.uniform(0, 0.25), int(rng.integers(2, 9))
            def ball(u, v):
                th, ph = 6.283185 * u, 3.14159 * (0.02 + 0.96 * v)
                rr = R * (1 + lump * np.sin(fl * th) * np.sin(fl * ph))
                return np.array([rr * np.sin(ph) * np.cos(th), rr * np.cos(ph), rr * np.sin(ph) * np.sin(th)])
            mesh(k, k, ball, True, False)
        body.append("popTransform")
    for _ in range(int(rng.integers(0, 6))):
        material()
        body.append("sphere %s %s" % (_vec(rng.uniform(-4, 4, 3)), _f(rng.uniform(0.2, 0.7))))
    out.append("maxverts %d" % max(1, len(verts)))
    return "\n".join(out + verts + body) + "\n"


# ---- malformed text: what the reference's parser makes of it (Scene.cpp:15-224, readvals :226-246) ------------------------------
NUMS = ["1.", ".5", "+2", "-.5", "1e0", "1E+0", "0x10", "1f", "1,5", "nan", "inf", "-inf", "1e40", "1e-50", "1e-40", "--1", "1-", "1e", "1e+", "007", "1.5.5",
        "1d0", "0.1e1x", "", " ", "\t", "NaN", "INF", "infinity", ".", "-", "+", "1_0", "٣", "1\r"]


def mutate(text, rng):
    lines = text.split("\n")
    for _ in range(int(rng.integers(1, 6))):
        k = int(rng.integers(0, len(lines)))
        ln = lines[k]
        cmd = ln.split(" ")[0]
        if cmd in ("tri", "maxverts", "maxvertnorms", "size"):   # integer arguments: out-of-range indices are undefined behaviour
            continue
        what = int(rng.integers(0, 9))
        tok = ln.split(" ")
        if what == 0 and len(tok) > 1:
            tok[int(rng.integers(1, len(tok)))] = NUMS[int(rng.integers(0, len(NUMS)))]
        elif what == 1 and len(tok) > 1:
            del tok[int(rng.integers(1, len(tok)))]
        elif what == 2:
            tok.insert(int(rng.integers(1, len(tok) + 1)), NUMS[int(rng.integers(0, len(NUMS)))])
        elif what == 3:
            tok[0] = tok[0].upper() if rng.random() < 0.5 else tok[0] + "x"
        elif what == 4:
            tok[0] = (" " if rng.random() < 0.5 else "\t") + tok[0]
        elif what == 5:
            tok = [" ".join(tok).replace(" ", "\t" if rng.random() < 0.5 else "  ")]
        elif what == 6:
            tok[-1] = tok[-1] + ("\r" if rng.random() < 0.5 else " \t ")
        elif what == 7:
            tok = ["#" + " ".join(tok)] if rng.random() < 0.5 else [" #" + " ".join(tok)]
        else:
            lines.insert(k, ["pushTransform", "popTransform", "popTransform", "translate 1 2", "scale 2 2 2", "rotate 0 0 0 45", "rotate 0 0 1 nan",
                             "ambient .1 .2 .3", "emission 1 1", "attenuation 1 0", "maxdepth", "maxdepth 2.7", "maxdepth -3", "output x y", "sphere 0 0 0",
                             "sphere 0 0 0 1 5", "point 1 2 3 1 1", "directional 1 2 3 1 1 1 9", "camera 0 0 1 0 0 0 0 1 0", "vertex 1 2", "shininess"][int(rng.integers(0, 21))])
            continue
        lines[k] = " ".join(tok)
    return "\n".join(lines)


def mutated_scene_text(seed):
    """One of the first 48 plain scenes with one to five lines damaged: numbers in every spelling a stream may or may not read,
    tokens missing or too many, commands in the wrong case or indented, tabs, carriage returns, comment signs in and out of
    column 0, stray stack commands and short argument lists.  (Lines with integer arguments are left alone: an index out of
    range is undefined behaviour in the reference.)"""
    return mutate(random_scene_text(seed % 48), np.random.default_rng(900000 + seed))


def many_lights_scene_text(n_lights, seed=0):
    """A few spheres, a floor and a mirror under n_lights lights of both kinds, some of them inside objects or below the floor:
    the light loop of main.cpp:208-223 at lengths a counter per hit, a bit mask per hit or a byte per light could trip over
    (31, 32, 33, 63, 64, 65, 255, 256, 257 ...)."""
    rng = np.random.default_rng(7000 + 31 * n_lights + seed)
    out = ["size %d %d" % (int(rng.integers(40, 90)), int(rng.integers(30, 60))), "maxdepth %d" % int(rng.integers(1, 4)),
           "camera 0 1.5 6 0 0.3 0 0 1 0 45"]
    if seed % 2:
        out.append("attenuation 1 0.1 0.02")
    for i in range(n_lights):
        col = _vec(rng.uniform(0.2, 1.0, 3) * (3.0 / max(n_lights, 3)))
        if rng.random() < 0.7:
            out.append("point %s %s" % (_vec(rng.uniform(-4, 4, 3) + np.array([0, 2.5, 1])), col))
        else:
            out.append("directional %s %s" % (_vec(rng.uniform(-1, 1, 3) + np.array([0, 1.2, 0.4])), col))
    out += ["maxverts 8", "vertex -5 -0.5 -5", "vertex 5 -0.5 -5", "vertex 5 -0.5 5", "vertex -5 -0.5 5",
            "vertex -3 -0.5 -3", "vertex 3 -0.5 -3", "vertex 3 3 -3", "vertex -3 3 -3"]
    out += ["ambient 0.02 0.02 0.03", "diffuse 0.6 0.6 0.5", "specular 0.2 0.2 0.2", "shininess 30", "tri 0 2 1", "tri 0 3 2",
            "diffuse 0.1 0.1 0.1", "specular 0.8 0.8 0.8", "shininess 100", "tri 4 5 6", "tri 4 6 7"]
    for _ in range(int(rng.integers(2, 6))):
        out += ["diffuse %s" % _vec(rng.uniform(0.1, 0.9, 3)), "specular %s" % _vec(rng.uniform(0, 0.6, 3)), "shininess %d" % int(rng.choice([5, 20, 80])),
                "sphere %s %s" % (_vec(rng.uniform(-2, 2, 3) + np.array([0, 0.6, 0])), _f(rng.uniform(0.3, 0.9)))]
    return "\n".join(out) + "\n"


MANY_LIGHTS = [31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 700]
