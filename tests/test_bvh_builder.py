"""CPU: the host-side hierarchy builder on its own (raytracer_amd/csrc/mirt_bvh.h, compiled into tests/native/bvh_check.cpp):
its invariants hold -- every primitive in exactly one leaf, leaves of one kind and at most eight long, parents enclose
children, child ranges tile the parent's -- and the tree built on several threads is the tree built on one."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("bvh") / "bvh_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", os.path.join(ROOT, "tests", "native", "bvh_check.cpp"), "-o", out])
    return out


def run(exe, n, seed, threads):
    env = dict(os.environ, MIRT_THREADS=str(threads))
    out = subprocess.run([exe, str(n), str(seed)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.match(r"OK nodes (\d+) leaves (\d+) depth (\d+) threads (\d+) digest ([0-9a-f]+)", out.stdout)
    assert m, out.stdout
    assert int(m.group(4)) == threads
    return m.group(1), m.group(2), m.group(3), m.group(5)


# 1: a single leaf; 9: the first split; 5000: one thread's recursion only; 400000: large ranges split by all threads, subtrees as tasks
@pytest.mark.parametrize("n,seed", [(1, 1), (9, 2), (5000, 3), (400000, 4), (700001, 5)])
def test_invariants_and_thread_count_independence(exe, n, seed):
    one = run(exe, n, seed, 1)
    for threads in (3, 8):
        assert run(exe, n, seed, threads) == one
