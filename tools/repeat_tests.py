"""Repeat-log of the kink-robust fast-mode comparisons: runs the given pytest node ids N times in ONE process on the GPU box
and prints one line per run.  python tools/repeat_tests.py N nodeid [nodeid ...] > profiles/r3_fast_mode_repeat.txt"""
import io
import contextlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
import pytest  # noqa: E402

n, ids = int(sys.argv[1]), sys.argv[2:]
bad = 0
for node in ids:
    for i in range(n):
        buf = io.StringIO()
        t0 = time.time()
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(buf):
            rc = pytest.main(["-q", "-x", "-p", "no:cacheprovider", node])
        tail = [l for l in buf.getvalue().splitlines() if " passed" in l or " failed" in l or "error" in l.lower()]
        print(f"run {i + 1:2d}/{n}  {node.split('::')[-1]:70s} {'PASS' if rc == 0 else 'FAIL'}  {time.time() - t0:5.1f} s  {tail[-1] if tail else ''}", flush=True)
        bad += rc != 0
print(f"{'ALL GREEN' if bad == 0 else str(bad) + ' FAILED'}")
sys.exit(1 if bad else 0)
