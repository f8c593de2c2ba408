"""Writes scene texts for the sanitizer run into a directory: every scene file the package holds, random scenes of the four
generators and damaged texts (tests/random_scenes.py).  usage: make_texts.py <dir> [random-per-kind] [damaged]"""
import glob
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from random_scenes import big_scene_text, hard_scene_text, mutated_scene_text, nasty_scene_text, random_scene_text  # noqa: E402

out = sys.argv[1]
per_kind = int(sys.argv[2]) if len(sys.argv) > 2 else 40
damaged = int(sys.argv[3]) if len(sys.argv) > 3 else 200
os.makedirs(out, exist_ok=True)
n = 0
for p in sorted(glob.glob(os.path.join(ROOT, "raytracer_amd", "data", "scenes", "**", "*"), recursive=True)):
    if os.path.isfile(p) and os.path.getsize(p) < 30_000_000:
        shutil.copy(p, os.path.join(out, "file%03d_%s" % (n, os.path.basename(p))))
        n += 1
for kind, gen, count in (("plain", random_scene_text, per_kind), ("hard", hard_scene_text, per_kind), ("nasty", nasty_scene_text, per_kind),
                         ("big", big_scene_text, max(1, per_kind // 10)), ("damaged", mutated_scene_text, damaged)):
    for seed in range(count):
        with open(os.path.join(out, "%s%04d.test" % (kind, seed)), "w") as f:
            f.write(gen(seed))
        n += 1
print(n, "texts in", out)
