#!/usr/bin/env python3
"""Where mirt_render's time goes: device frame, copy back (pageable / page-locked by the library / page-locked by the caller)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from raytracer_amd import workloads
from raytracer_amd.device import DeviceScene
from raytracer_amd.scene import FlatScene
w, h = 3840, 2160
flat = FlatScene.from_text(workloads.dragon(w, h))
def med(f, n=9):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts))
with DeviceScene(flat, 0) as s:
    a = np.zeros((h, w, 3), np.uint8)
    s.render(w, h, out=a); s.render(w, h, out=a)
    print("mirt_render, numpy buffer (library pins it): %.3f ms, device %.3f" % (med(lambda: s.render(w, h, out=a)), s.render(w, h, out=a)[1]["device_ms"]))
    t = torch.empty((h, w, 3), dtype=torch.uint8, pin_memory=True)
    b = t.numpy()
    s.render(w, h, out=b); s.render(w, h, out=b)
    print("mirt_render, caller-pinned buffer: %.3f ms" % med(lambda: s.render(w, h, out=b)))
    d = torch.empty(w * h * 3, dtype=torch.uint8, device="cuda")
    def copy_pinned():
        t.view(-1).copy_(d, non_blocking=True); torch.cuda.synchronize()
    print("D2H 24.9 MB to pinned: %.3f ms" % med(copy_pinned))
    c = torch.empty(w * h * 3, dtype=torch.uint8)
    def copy_pageable():
        c.copy_(d); torch.cuda.synchronize()
    print("D2H 24.9 MB to pageable: %.3f ms" % med(copy_pageable))
    from raytracer_amd import _capi
    p = _capi.make_params(w, h)
    st = torch.cuda.Stream()
    def dev_only():
        s.render_device(p, d.data_ptr(), w * h * 3, st.cuda_stream); st.synchronize()
    dev_only()
    print("mirt_render_device + stream sync: %.3f ms" % med(dev_only))
