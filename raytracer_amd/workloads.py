"""Scene texts of the benchmark configurations that are not a file of the reference as it stands.

All of them are written in the reference's own scene language (Scene.cpp:15-224) so that the reference can load them
unchanged; they are deterministic (integer LCG, fixed seeds), so the text generated on any machine is the same bytes.

  dragon(w, h)      final_scenes/scene7.test with only the `size` line changed (what the reference's *_small files are)
  cornell(w, h)     final_scenes/scene6.test, likewise
  spheres(w, h)     final_scenes/scene5.test (1000 translate-only spheres) plus `n_ellipsoids` spheres under
                    translate + rotate + non-uniform scale, so that Sphere::intersect runs with a non-trivial matrix
  dragon_grid(...)  scene7's vertex pool once, `nx` x `nz` instances of its triangles, each under translate + rotate
                    about y, 16 point lights, maxdepth 8: 10 M triangles at 10 x 10
"""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
SCENE_DIR = os.path.join(HERE, "data", "scenes")  # the reference's scene files (input data, copied unchanged)


# Seed of the per-instance rotation angles of dragon_grid().  With 20251111 and 20251112 the 10 x 10 grid contains two
# vertices that make the reference's Partition::split (Partition.cpp:77-101) recurse for ever (the reference overflows
# its stack while loading the scene; the library's loader reports 2 truncated nodes); 20251113 is the first seed after
# them that the reference can load, so it is the one whose outputs could be recorded from the reference.
GRID_SEED = 20251113


class LCG:
    """x <- (1103515245 x + 12345) mod 2^31; uniform() uses the top bits."""

    def __init__(self, seed):
        self.x = seed & 0x7FFFFFFF

    def next(self):
        self.x = (1103515245 * self.x + 12345) & 0x7FFFFFFF
        return self.x

    def uniform(self, lo, hi):
        return lo + (hi - lo) * ((self.next() >> 7) / float(1 << 24))


def _read(name):
    with open(os.path.join(SCENE_DIR, name + ".test"), "rb") as f:
        return f.read()


def _resize(text, w, h):
    out, n = re.subn(rb"(?m)^size[ \t]+\S+[ \t]+\S+[ \t]*\r?$", b"size %d %d" % (w, h), text, count=1)
    assert n == 1, "scene has no size line"
    return out


def dragon(width=3840, height=2160):
    return _resize(_read("scene7"), width, height)


def cornell(width=640, height=480):
    return _resize(_read("scene6"), width, height)


def spheres(width=1920, height=1080, n_ellipsoids=200, seed=1234):
    rng = LCG(seed)
    lines = ["", "# %d ellipsoids: translate, rotate about a seeded axis, non-uniform scale" % n_ellipsoids,
             "diffuse 0.1 0.3 0.5", "specular 0.5 0.5 0.5", "shininess 60"]
    for _ in range(n_ellipsoids):
        x, y, z = rng.uniform(-9.5, 8.5), rng.uniform(-1.2, 2.5), rng.uniform(-50.0, -8.0)
        ax, ay, az = rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.1, 1)
        ang = rng.uniform(0, 360)
        sx, sy, sz = rng.uniform(0.5, 2.0), rng.uniform(0.5, 2.0), rng.uniform(0.5, 2.0)
        r = rng.uniform(0.2, 0.6)
        lines += ["pushTransform", "translate %.4f %.4f %.4f" % (x, y, z), "rotate %.4f %.4f %.4f %.3f" % (ax, ay, az, ang),
                  "scale %.4f %.4f %.4f" % (sx, sy, sz), "sphere 0 0 0 %.4f" % r, "popTransform"]
    return _resize(_read("scene5"), width, height) + "\n".join(lines).encode() + b"\n"


def dragon_grid(width=3840, height=2160, nx=10, nz=10, seed=GRID_SEED, max_depth=8):
    src = _read("scene7").split(b"\n")
    verts = [ln for ln in src if ln.startswith(b"vertex ")]
    tris = b"\n".join(ln for ln in src if ln.startswith(b"tri ")) + b"\n"
    rng = LCG(seed)
    pitch_x, pitch_z = 16.0, 12.0
    head = ["# %d x %d Stanford dragons (scene7's mesh), 16 point lights" % (nx, nz), "size %d %d" % (width, height),
            "camera 0 %g %g 0 0 0 0 1 0 45" % (6.0 * max(nx, nz), 14.5 * max(nx, nz)),
            "output dragon_grid.png", "maxdepth %d" % max_depth, "attenuation 1 0.02 0.002",
            "ambient 0.02 0.02 0.02", "diffuse 0.4 0.1 0.1", "specular 0.3 0.3 0.3", "shininess 20"]
    half_x, half_z = pitch_x * (nx - 1) / 2.0, pitch_z * (nz - 1) / 2.0
    for i in range(4):
        for k in range(4):
            lx = -half_x * 0.85 + i * (2 * half_x * 0.85) / 3.0 if nx > 1 else (i - 1.5) * 4.0
            lz = -half_z * 0.85 + k * (2 * half_z * 0.85) / 3.0 if nz > 1 else (k - 1.5) * 4.0
            head.append("point %.3f 25 %.3f 0.45 0.45 0.45" % (lx, lz))
    head.append("maxverts %d" % len(verts))
    parts = ["\n".join(head).encode() + b"\n", b"\n".join(verts) + b"\n"]
    for ix in range(nx):
        for iz in range(nz):
            ang = rng.next() % 360
            parts.append(("pushTransform\ntranslate %.1f -5 %.1f\nrotate 0 1 0 %d\n" % (ix * pitch_x - half_x, iz * pitch_z - half_z, ang)).encode())
            parts.append(tris)
            parts.append(b"popTransform\n")
    return b"".join(parts)
