"""Exploration (CPU): mutated scene texts through the reference's parser (oracle/_ref) and the library's loader; every array that
crosses the C ABI must come out the same.  Each seed runs in a forked child, so a crash of the reference (its behaviour on some
malformed input is undefined) ends that child only.  usage: fuzz_loader.py first end"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from random_scenes import mutated_scene_text
from raytracer_amd.scene import FlatScene, _ARRAYS
from oracle.bind import RefOracle

def one(seed):
    text = mutated_scene_text(seed)
    with tempfile.NamedTemporaryFile("w", suffix=".test", delete=False) as f:
        f.write(text)
        path = f.name
    try:
        try:
            ref = RefOracle(path)
        except RuntimeError:
            ref = None
        try:
            mine = FlatScene.from_text(text)
            err = None
        except ValueError as e:
            mine, err = None, str(e)
        if ref is None or mine is None:
            if (ref is None) != (mine is None):
                print("seed", seed, "reference", "loads" if ref else "fails", "library", "loads" if mine else "fails: " + err[:120], flush=True)
            return
        rf = ref.flatten()
        bad = [n for n, _, _ in _ARRAYS if getattr(rf, n).shape != getattr(mine, n).shape or getattr(rf, n).tobytes() != getattr(mine, n).tobytes()]
        if (ref.width, ref.height, ref.max_depth) != (mine.width, mine.height, mine.max_depth):
            bad.append("size/maxdepth %s %s" % ((ref.width, ref.height, ref.max_depth), (mine.width, mine.height, mine.max_depth)))
        if bad:
            print("seed", seed, "DIFFERENT:", bad, flush=True)
            open("/tmp/fuzz_loader_%d.test" % seed, "w").write(text)
    finally:
        os.unlink(path)


first, end = int(sys.argv[1]), int(sys.argv[2])
crashed = 0
for seed in range(first, end):
    pid = os.fork()
    if pid == 0:
        try:
            one(seed)
        finally:
            os._exit(0)
    _, status = os.waitpid(pid, 0)
    if status != 0:
        crashed += 1
        print("seed", seed, "child ended with status", status, flush=True)
print("done", first, end, "crashed children", crashed)
