#!/usr/bin/env python3
"""Time of mirt_assemble_tiles on ONE GPU: the step rank 0 runs after the gather.  Eight synthetic rank buffers of a
7680x4320 frame (64x16 tiles, as bench.py --gpus 8 --workload dragon8k would deliver them), HIP events around 20 launches;
the result is checked against the numpy statement of the same reassembly.  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402
from raytracer_amd import _capi, multigpu  # noqa: E402

w, h, world = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (7680, 4320, 8)
lib = _capi.load_library()
nbytes = multigpu.per_rank_bytes(w, h, world)
tiles = torch.randint(0, 256, (world, nbytes), dtype=torch.uint8, device="cuda")
img = torch.empty(w * h * 3, dtype=torch.uint8, device="cuda")
p = multigpu.rank_params(w, h, 0, world)
st = torch.cuda.current_stream()


def launch():
    rc = lib.mirt_assemble_tiles(0, C.byref(p), world, C.c_void_p(tiles.data_ptr()), nbytes, C.c_void_p(img.data_ptr()), C.c_void_p(st.cuda_stream))
    assert rc == 0, lib.mirt_last_error()


for _ in range(3):
    launch()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
ev[0].record(st)
for i in range(20):
    launch()
    ev[i + 1].record(st)
torch.cuda.synchronize()
ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))
want = multigpu.assemble_host(list(tiles.cpu().numpy()), w, h, world)
ok = bool(np.array_equal(img.cpu().numpy().reshape(h, w, 3), want))
print(json.dumps({"what": "mirt_assemble_tiles, %dx%d, %d rank buffers of %d bytes, one GPU" % (w, h, world, nbytes), "median_us": ms[10] * 1e3, "min_us": ms[0] * 1e3,
                  "bytes_moved": 2 * w * h * 3, "GBs": 2 * w * h * 3 / (ms[10] * 1e-3) / 1e9, "equals_host_reassembly": ok}))
sys.exit(0 if ok else 1)
