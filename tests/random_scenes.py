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
