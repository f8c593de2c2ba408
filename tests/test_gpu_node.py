"""GPU: include/mirt_node.h -- one frame on the GPUs of a node from one process (tile split, exchange into the root,
reassembly).  One GPU is all this test can count on: the RCCL communicator is made over one device (its all-reduce
self-test runs), and the N > 1 logic runs in rehearsal mode (MIRT_NODE_REHEARSE=1: N ranks share the GPU, device-to-device
copies stand in for ncclSend/ncclRecv).  Every image must equal the one GPU's, byte for byte, with the same ray counts."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAYS = ("primary_rays", "shadow_rays", "reflection_rays", "primary_hits")


def test_one_device_through_the_rccl_communicator(flat_scenes, goldens):
    from raytracer_amd.node import NodeScene
    g = goldens("scene7")
    with NodeScene(flat_scenes("scene7"), [0]) as n:
        assert n.uses_rccl() == (True, 1)
        img, st = n.render()
        assert np.array_equal(img, g["bgr_f15"])
        assert [st[k] for k in RAYS[:3]] == g["rays_f15"].tolist()


@pytest.mark.parametrize("ranks", [2, 3, 8])
@pytest.mark.parametrize("name,size", [("scene6", None), ("scene7", (1000, 300)), ("scene5", (161, 97))])
def test_ranks_sharing_the_gpu_assemble_the_one_gpu_image(name, size, ranks, flat_scenes, monkeypatch):
    from raytracer_amd.device import DeviceScene
    from raytracer_amd.node import NodeScene
    monkeypatch.setenv("MIRT_NODE_REHEARSE", "1")
    flat = flat_scenes(name)
    w, h = size or (flat.width, flat.height)
    with DeviceScene(flat, 0) as s:
        want, wst = s.render(w, h)
    with NodeScene(flat, [0] * ranks) as n:
        assert n.uses_rccl() == (False, ranks)
        for _ in range(2):
            img, st = n.render(w, h)
            assert np.array_equal(img, want), "%d bytes differ" % int((img != want).sum())
            assert [st[k] for k in RAYS] == [wst[k] for k in RAYS]


def test_a_device_listed_twice_is_refused_outside_rehearsal(flat_scenes, monkeypatch):
    from raytracer_amd.device import MirtError
    from raytracer_amd.node import NodeScene
    monkeypatch.delenv("MIRT_NODE_REHEARSE", raising=False)
    with pytest.raises(MirtError, match="listed twice"):
        NodeScene(flat_scenes("scene6"), [0, 0])


def test_the_cpp_example_program(tmp_path):
    exe = os.path.join(ROOT, "examples", "multi_gpu")
    if not os.path.exists(exe):
        pytest.fail("examples/multi_gpu not built (make -C raytracer_amd/csrc)")
    scene = os.path.join(ROOT, "raytracer_amd", "data", "scenes", "scene6.test")
    out = subprocess.run([exe, scene, "--gpus", "1", "--out", str(tmp_path / "x.png"), "--check", "--frames", "4"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "image and ray counts equal one GPU's" in out.stdout and "RCCL" in out.stdout
    from PIL import Image
    assert Image.open(tmp_path / "x.png").size == (640, 480)
