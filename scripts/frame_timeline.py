#!/usr/bin/env python3
"""Diagnostic (a -DMIRT_TIMELINE=1 build of mirt_api.hip as exp/libmirt_timeline.so): where the time of ONE k_frame launch goes.
Prints, in microseconds from the first wave's start: when the pixel slots ran out (first / last wave to notice), when the last
wave ended, the mean wave lifetime and the mean time a wave spent after the slots had run out, rounds and lane use, the idle
lanes by kind and the hand-overs of stack entries between lanes (the counters of the last line exist in a -DMIRT_TIMELINE=2 build
only: atomics on one word each, so many that such a build's times are a hundred times the library's; =1 keeps the library's times)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import _capi, workloads
from raytracer_amd.device import DeviceScene
from raytracer_amd.scene import FlatScene
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
lib = _capi.load_library(os.environ.get("MIRT_TIMELINE_LIB") or os.path.join(ROOT, "exp", "libmirt_timeline.so"))  # -DMIRT_TIMELINE=1 build
flat = FlatScene.from_text(workloads.dragon(w, h))
def q(s, k):
    v = C.c_uint64()
    assert lib.mirt_scene_query(s.h, k, C.byref(v)) == 0, lib.mirt_last_error()
    return v.value
with DeviceScene(flat, 0, lib=lib) as s:
    s.render(w, h)
    for k in range(140, 156): q(s, k)
    for k in range(100, 116): q(s, k)
    _, st = s.render(w, h)
    t0 = (~q(s, 140)) & (2**64 - 1); tend = q(s, 141); ex0 = (~q(s, 142)) & (2**64 - 1); ex1 = q(s, 143)
    life, tail, waves, rounds, lanes = (q(s, k) for k in range(144, 149))
    us = lambda t: t / 100.0
    print("device_ms %.3f; slots exhausted at %.0f .. %.0f us, last wave ended at %.0f us" % (st["device_ms"], us(ex0 - t0), us(ex1 - t0), us(tend - t0)))
    print("waves %d, mean lifetime %.0f us, mean time after exhaustion %.0f us, rounds per wave %.1f, lanes per round %.1f" % (waves, us(life / waves), us(tail / waves), rounds / waves, lanes / max(1, rounds)))
    mx = [q(s, k) for k in range(149, 152)]; sm = [q(s, k) for k in range(152, 155)]
    print("longest round (camera, light, reflection) %s us; share of the summed round time %s" % ([round(us(x)) for x in mx], [round(x / max(1, sum(sm)), 3) for x in sm]))
    it = [q(s, k) for k in range(104, 109)]
    print("(-DMIRT_TIMELINE=2 only) walks of a wave (shared): %d, %.1f iterations on average, the longest %d; %d walks of more than 200 iterations, in which %.1f lanes were busy on average" % (
        it[2], it[1] / max(1, it[2]), it[0], it[3], it[4] / max(1.0, float(it[3])) / 200.0))
    redo = [q(s, k) for k in range(100, 104)]
    print("strict second walks: %d rounds held up by %d rays, %.0f us in all (%.2f us per wave of the launch), longest %.0f us" % (redo[0], redo[3], us(redo[1]), us(redo[1]) / waves, us(redo[2])))
    r = [q(s, k) for k in range(108, 116)]
    print("shared walks: %d wave iterations, idle lane slots %d of which %d hold a hit of their own and %d are owners that gave work away; hand-overs %d, entries handed over %d, free lanes then %d, idle lanes holding a hit then %d" % (
        r[3], r[2], r[0], r[1], r[4], r[5], r[7], r[6]))
    # the waves that ended last (per-wave records of a -DMIRT_TIMELINE=1 build)
    import numpy as np
    rec = np.array([[q(s, 1000 + 4 * i + k) for k in range(4)] for i in range(min(waves, 8192))], dtype=np.uint64)
    start, end, exh = (rec[:, k].astype(np.int64) - int(t0) for k in range(3))
    rounds, longest = (rec[:, 3] & np.uint64(0xffffffff)).astype(np.int64), (rec[:, 3] >> np.uint64(32)).astype(np.int64)
    order = np.argsort(end)[::-1]
    print("wave: start, end, slots found exhausted at, rounds, longest round (us); the 12 that ended last, then every 1000th")
    for i in list(order[:12]) + list(order[::1000]):
        print("  %5d: %5.0f %5.0f %5.0f %3d %4.0f" % (i, us(start[i]), us(end[i]), us(exh[i]), rounds[i], us(longest[i])))
    print("ended by (us): 50 %% %.0f, 90 %% %.0f, 99 %% %.0f, all %.0f; rounds per wave: mean %.1f, of the last 1 %% %.1f; longest round: mean %.0f us, of the last 1 %% %.0f us" % (
        us(np.percentile(end, 50)), us(np.percentile(end, 90)), us(np.percentile(end, 99)), us(end.max()), rounds.mean(), rounds[order[:len(order) // 100]].mean(),
        us(longest.mean()), us(longest[order[:len(order) // 100]].mean())))
