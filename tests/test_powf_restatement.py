"""CPU: the restatement of glibc's powf that the device code uses (raytracer_amd/csrc/mirt_powf.h, compiled here for the
host) against the C library's powf and against the committed vectors.  glm::pow(float, float) of the reference's shading
(RayTracer/main.cpp:263) is that library function; it is not correctly rounded, so the device restates it."""
import ctypes as C
import ctypes.util
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_restatement_equals_libm_on_millions_of_arguments(tmp_path):
    exe = str(tmp_path / "powf_check")
    fma = "-mfma" if "fma" in open("/proc/cpuinfo").read() else "-O2"  # without the instruction __builtin_fma calls the exact fma()
    subprocess.check_call(["g++", "-O2", fma, "-o", exe, os.path.join(ROOT, "tests", "native", "powf_check.cpp"), "-lm"])
    out = subprocess.run([exe, "6000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "bad=0" in out.stdout


def test_committed_vectors_are_what_this_libm_returns(goldens):
    """The fixture was made in the development container; the box that runs the GPU tests must have the same powf,
    otherwise the reference build (oracle/_ref) would shade differently there."""
    g = goldens("kat_powf")
    libm = C.CDLL(ctypes.util.find_library("m"))
    libm.powf.restype = C.c_float
    libm.powf.argtypes = [C.c_float, C.c_float]
    got = np.array([libm.powf(float(a), float(b)) for a, b in zip(g["x"][:20000], g["y"][:20000])], np.float32)
    assert np.array_equal(got.view(np.uint32), g["want"][:20000].view(np.uint32))
    # and it really is not the correctly rounded power: the restatement is needed
    exact = (np.float64(g["x"]) ** np.float64(g["y"])).astype(np.float32)
    assert (exact.view(np.uint32) != g["want"].view(np.uint32)).sum() > 10
