"""Test infrastructure (CPU baseline only): one worker of the all-cores timing of the reference's pixel loop.

  python3 -m oracle.cpu_rows <scene.test> <width> <height> <row0> <row_step>

Loads the scene with the reference's own compiled classes (oracle/_ref), renders rows row0, row0+row_step, ... and
prints {"rays": n, "seconds": s} with s the time inside the pixel loop only.  bench.py starts one such process per host
core, each on its own rows, and adds the rays up over the slowest worker's time.
"""
import json
import sys

from oracle.bind import RefOracle


def main():
    path, w, h, row0, step = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    ref = RefOracle(path)
    _, rs = ref.render(w, h, row0=row0, row_step=step)
    ref.close()
    print(json.dumps({"rays": rs["primary"] + rs["shadow"] + rs["reflection"], "seconds": rs["seconds"]}))


if __name__ == "__main__":
    main()
