import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (read once, when HIP initialises) and streams that
# share a queue run one after the other: the multi-stream tests want their eight streams to really overlap.
if int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < 16:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"

GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENE_NAMES = ["scene1", "scene2", "scene3", "scene4-specular", "scene5", "scene6", "scene7"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the C restatement is test infrastructure; build it on demand (gcc, < 1 s)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])


@pytest.fixture(scope="session")
def flat_scenes():
    from raytracer_amd.scene import FlatScene
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = FlatScene.load(os.path.join(GOLDEN, name + ".flat.npz"))
        return cache[name]
    return get


@pytest.fixture(scope="session")
def goldens():
    cache = {}

    def get(name):
        if name not in cache:
            with np.load(os.path.join(GOLDEN, name + ".golden.npz"), allow_pickle=False) as z:
                cache[name] = {k: z[k] for k in z.files}
        return cache[name]
    return get


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
