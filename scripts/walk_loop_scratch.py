#!/usr/bin/env python3
"""After `make -C raytracer_amd/csrc asm`: which loops of the frame kernels touch scratch memory.

k_frame has two speeds (about 16 % apart): in the slow one the register allocator has spilled something the walk needs at every
push and pop, and the loops that hold the node fetch (four 16-byte loads) wait for scratch loads.  This prints, per kernel, every
loop (a backward branch to a label) that contains a 16-byte global load, with its scratch loads and stores, smallest first.
The walk loops are the ones with 14 such loads: 0 scratch loads in them (5 in the build with an overflow stack) is the fast form,
10 the slow one.  usage: walk_loop_scratch.py [mirt_gfx950.s] [kernel name substring ...]"""
import os
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracer_amd", "csrc", "mirt_gfx950.s")
want = sys.argv[2:] or ["_Z7k_frameILi0ELb0ELb1ELb0E", "_Z7k_frameILi0ELb1ELb1ELb0E", "_Z7k_frameILi0ELb0ELb0ELb0E"]
s = open(path).read()
for m in re.finditer(r"^(_Z\w+):", s, re.M):
    name = m.group(1)
    if not any(w in name for w in want):
        continue
    end = s.find(".Lfunc_end", m.end())
    lines = s[m.end():end].split("\n")
    label = {}
    for n, ln in enumerate(lines):
        k = re.match(r"^(\.LBB\d+_\d+):", ln)
        if k:
            label[k.group(1)] = n
    loops = []
    for n, ln in enumerate(lines):
        k = re.search(r"\s(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", ln)
        if k and k.group(2) in label and label[k.group(2)] <= n:
            body = lines[label[k.group(2)]:n + 1]
            loops.append((n - label[k.group(2)], label[k.group(2)], sum("scratch_load" in x for x in body), sum("scratch_store" in x for x in body),
                          sum("global_load_dwordx4" in x for x in body)))
    print(name, "scratch loads in all:", sum("scratch_load" in x for x in lines))
    for length, at, sl, ss, gl in sorted(loops):
        if gl and length < 2500:
            print("  loop at line %5d, %4d lines: %2d 16-byte loads, %2d scratch loads, %2d scratch stores" % (at, length, gl, sl, ss))
