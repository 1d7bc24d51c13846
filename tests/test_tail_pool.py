"""csrc/tail_pool.hpp, host-only: the claimed shares that keep a call from waiting for a helper thread which has lost its
CPU.  tests/native/tail_pool_stress.cpp posts shares to the idle helpers, stalls some of them for 2 ms BEFORE they look at
their job, and checks that every share runs exactly once, that no round waits for such a helper, and (with
-fsanitize=thread, when the toolchain links it) that the hand-over has no data race.  CPU only."""
import os
import subprocess

import pytest

import util

ROOT = util.ROOT
CSRC = os.path.join(ROOT, "webgpu-msm-bls12-377_amd", "csrc")
SRC = os.path.join(ROOT, "tests", "native", "tail_pool_stress.cpp")
OUT = os.path.join(ROOT, "tests", "native", "_build")


def build(name, flags):
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, name)
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-I", CSRC] + flags + ["-o", exe, SRC], capture_output=True, text=True)
    return exe if r.returncode == 0 else None


def test_shares_run_exactly_once_and_nobody_waits_for_a_stalled_helper():
    exe = build("tail_pool_stress", [])
    assert exe, "g++ could not build the harness"
    r = subprocess.run([exe, "6000", "41"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok:"), r.stdout


def test_shares_under_thread_sanitizer():
    exe = build("tail_pool_stress_tsan", ["-fsanitize=thread"])
    if not exe:
        pytest.skip("this toolchain does not link -fsanitize=thread")
    r = subprocess.run([exe, "1500", "29"], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout + r.stderr[-3000:]
