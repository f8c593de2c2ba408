#!/usr/bin/env python3
"""Does mirt_render_device_noalloc go into a HIP graph?  Captures one frame on a side stream (torch.cuda.CUDAGraph), replays it
three times and compares each replay with mirt_render_device, byte for byte.  (An experiment, not part of the test suite.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import _capi
from raytracer_amd.device import DeviceScene
from raytracer_amd.scene import FlatScene
flat = FlatScene.load(os.path.join(ROOT, "tests", "golden", "scene7.flat.npz"))
w, h = 640, 400
with DeviceScene(flat, 0) as s:
    p = _capi.make_params(w, h)
    n = w * h * 3
    want = torch.zeros(n, dtype=torch.uint8, device="cuda")
    got = torch.zeros(n, dtype=torch.uint8, device="cuda")
    s.render_device(p, want.data_ptr(), n, 0)
    torch.cuda.synchronize()
    s.render_device_noalloc(p, got.data_ptr(), n, 0)  # warm: nothing lazy is left for the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=st):
        s.render_device_noalloc(p, got.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    for i in range(3):
        got.zero_()
        g.replay()
        torch.cuda.synchronize()
        print("replay", i, "equal:", bool(torch.equal(got, want)), "nonzero:", bool(got.any()))
