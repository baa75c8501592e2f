"""CPU: the host worker pool of the topology build and the symbolic analysis (csrc/pgo_pool.h) under concurrent callers, with
ThreadSanitizer when the compiler provides it (a race between the job list's lock and the lock-free slot counter crashed
batched solves on a 256-core box in r02 and never showed on eight cores without it)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "pool_selftest.cpp")


@pytest.mark.parametrize("tsan", [True, False])
def test_pool_selftest(tmp_path, tsan):
    exe = str(tmp_path / "pool_selftest")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", SRC, "-o", exe] + (["-fsanitize=thread"] if tsan else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if tsan:
            pytest.skip("no ThreadSanitizer runtime here: " + r.stderr[-200:])
        raise AssertionError(r.stderr)
    env = dict(os.environ, PGO_HOST_THREADS="12")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "ThreadSanitizer" not in out.stderr
    # (1): 4 callers x 3000 phases x slots x 100 odd terms; (2): 3000 x (every split conserves the sum: 37 + 11 at the root, leaves of 1 -> 48 leaves + inner nodes)
    assert out.stdout.startswith("ok ")
    total = int(out.stdout.split()[1])
    expect1 = 3000 * 100 * (16 + 5 + 2 + 1)
    def tree(r):
        return r if r <= 1 else r + tree(r // 2) + tree(r - r // 2)
    assert total == expect1 + 3000 * (tree(37) + tree(11))
