"""bench.py --gpus N from a plain `python` command (the driver's command form): the parent must start the N ranks
itself, before anything touches the GPU, and relay their output and exit code.  CPU only: on a
box without a GPU every rank must fail loudly ("no HIP device"), which is exactly what proves that the ranks were
started with WORLD_SIZE = N."""
import os
import subprocess
import sys

import pytest

import util

sys.path.insert(0, util.ROOT)
import bench  # noqa: E402


def test_launch_command_shape():
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"], 29512)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29512"
    assert cmd[-7] == os.path.join(util.ROOT, "bench.py") and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert 1024 < bench.free_port() < 65536


def test_plain_command_starts_the_ranks():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU suite runs the real rehearsal (tests/test_bench_gpu.py)")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    proc = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                          capture_output=True, text=True, timeout=600, env=env)
    text = proc.stdout + proc.stderr
    assert proc.returncode != 0  # no GPU here: the ranks refuse to run (there is no CPU fallback) and the parent relays that
    assert "no HIP device visible" in text, text[-2000:]
    assert "WORLD_SIZE=1" not in text  # the old behaviour: the parent itself died on the WORLD_SIZE check


def test_roofline_traffic_comes_from_the_newest_profile():
    """bench.py's roofline.traffic is read from a committed rocprofv3 PMC summary: the directory named in
    profiles/LATEST (a plain name sort put r02_mid before r02_final), and the bytes follow the guide's correction
    (FETCH_SIZE counts half of every 128-byte request on gfx950)."""
    tag, summary = bench.latest_pmc_summary()
    with open(os.path.join(util.ROOT, "profiles", "LATEST")) as f:
        assert tag == f.read().strip()
    acc = [v for k, v in summary.items() if "k_accumulate" in k]
    assert acc and "FETCH_SIZE" in acc[0] and "WRITE_SIZE" in acc[0]
    nbytes, source = bench.pmc_traffic_bytes(20)
    fetch_kb, write_kb = acc[0]["FETCH_SIZE"]["mean_per_launch"], acc[0]["WRITE_SIZE"]["mean_per_launch"]
    assert abs(nbytes - (2 * fetch_kb + write_kb) * 1024) <= 0.01 * nbytes
    assert tag in source
    assert bench.pmc_traffic_bytes(16) == (None, None)  # taken at 2^20 only
