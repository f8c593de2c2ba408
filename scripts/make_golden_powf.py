#!/usr/bin/env python3
"""Known-answer vectors for the power function of the shading path: tests/golden/kat_powf.golden.npz.

glm::pow(float, float) in RayTracer/main.cpp:263 is powf of the C library the reference's build links against -- here
glibc (the reference compiled by oracle/Makefile).  The vectors are that function's outputs for arguments of the kind
calculateSpecularLighting produces (a clamped cosine and a shininess), for bases a few ulps around 1, for subnormal bases
and for the special cases (zeros, NaN, exponent zero).  Run in the development container."""
import ctypes as C
import ctypes.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libm = C.CDLL(ctypes.util.find_library("m"))
libm.powf.restype = C.c_float
libm.powf.argtypes = [C.c_float, C.c_float]

rng = np.random.default_rng(20251111)
shin = np.array([0.0, 0.5, 1.0, 2.0, 3.5, 10.0, 20.0, 50.0, 100.0, 128.0, 1000.0, 12345.0, 1e-3], np.float32)
xs, ys = [], []
n = 60000
xs.append(rng.random(n, dtype=np.float32)); ys.append(rng.choice(shin, n))
xs.append((1.0 - rng.random(n, dtype=np.float32) * np.float32(1e-3)).astype(np.float32)); ys.append(rng.choice(shin, n))
one = np.float32(1.0).view(np.uint32)
xs.append((one + rng.integers(-64, 64, 2000)).astype(np.uint32).view(np.float32)); ys.append(rng.choice(shin, 2000))
xs.append(rng.integers(1, 0x00800000, 2000).astype(np.uint32).view(np.float32)); ys.append(rng.choice(shin, 2000))   # subnormal bases
xs.append(rng.random(20000, dtype=np.float32)); ys.append((rng.random(20000, dtype=np.float32) * 200).astype(np.float32))
xs.append(np.array([0.0, 0.0, 0.0, np.nan, np.nan, 1.0, 0.5, -0.0], np.float32)); ys.append(np.array([0.0, 1.0, 20.0, 0.0, 5.0, 1e30, 0.0, 3.0], np.float32))
x = np.concatenate(xs).astype(np.float32)
y = np.concatenate(ys).astype(np.float32)
want = np.array([libm.powf(float(a), float(b)) for a, b in zip(x, y)], np.float32)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "kat_powf.golden.npz"), x=x, y=y, want=want)
exact = (np.float64(x) ** np.float64(y)).astype(np.float32)
print("%d vectors; %d of them differ from the correctly rounded power" % (x.size, int((exact.view(np.uint32) != want.view(np.uint32)).sum())))
