"""bench.py and the sweep harness on the GPU box: the self-launching multi-rank path (two ranks rehearsed on ONE
device with the gloo exchange -- RCCL refuses two ranks per device; the 8-GPU run is the driver's), the single-GPU
line's contract fields, and the reference's benchmark sweep (src/submission/miscellaneous/full_benchmarks.ts:6-162)
run once at 2^16.  `pytest -m gpu`."""
import json
import os
import re
import subprocess
import sys

import pytest

import util

pytestmark = pytest.mark.gpu


def run_bench(extra, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    proc = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=900, env=env)
    assert proc.returncode == 0, (proc.stdout + proc.stderr)[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_and_two_rank_rehearsal():
    one = run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--log-n", "16"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in one, key
    assert one["n_gpus"] == 1 and one["higher_is_better"] is False and one["vs_baseline"] is None
    roof = one["roofline"]
    assert roof["bound"] == "hbm" and 0 < roof["frac"] < 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    sec = roof["secondary"]
    assert sec["bound"] == "int32-mad" and 0 < sec["frac"] < 1 and abs(sec["frac"] - sec["achieved"] / sec["peak"]) < 1e-3
    assert one["cpu_baseline"]["kind"] == "port" and one["cpu_baseline"]["cores"] >= 1
    # plain `python bench.py --gpus 2`: the parent starts both ranks itself
    two = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--log-n", "16"],
                    {"MSM377_BENCH_SINGLE_DEVICE": "1", "MSM377_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["result_x"] == one["result_x"]
    assert "sharded over 2 GPUs" in two["config"]["parallelism"]


def test_rccl_path_with_one_rank():
    """The code path the driver's 8-GPU run takes -- msm377_g1_window_partials_resident, all_gather_into_tensor over RCCL
    straight from the device records, one read-back, msm377_g1_combine_partials_ctx -- rehearsed with ONE rank that owns
    all 16 windows (this box has one GPU): same result as the single-GPU engine call."""
    one = run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--log-n", "17", "--no-cpu-baseline"])
    forced = run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--log-n", "17", "--no-cpu-baseline"], {"MSM377_BENCH_FORCE_SHARDED": "1"})
    assert forced["result_x"] == one["result_x"]
    assert "sharded over 1 GPUs" in forced["config"]["parallelism"]


def test_full_benchmarks_sweep_at_2_16(monkeypatch):
    from webgpu_msm_bls12_377_amd.host import full_benchmarks as fb

    monkeypatch.setattr(fb, "DELAY", 0)
    lines = []
    table = fb.full_benchmarks(start_power=16, end_power=16, out=lines.append)
    rows = table.splitlines()
    assert rows[0].startswith("| MSM size | 1st run | Run 1 | Run 2 | Run 3 | Run 4 | Run 5 | Average (incl 1st) | Average (excl 1st) |")
    assert rows[1].count("-|") == 9 and len(rows) == 3
    m = re.fullmatch(r"\| 2\^16 \|" + r" `([0-9.]+)` \|" * 6 + r" \*\*`([0-9.]+)`\*\* \| \*\*`([0-9.]+)`\*\* \|", rows[2])
    assert m, rows[2]
    vals = [float(v) for v in m.groups()]
    assert all(0 < v < 60000 for v in vals)
    assert abs(vals[6] - sum(vals[:6]) / 6) < 0.02 and abs(vals[7] - sum(vals[1:6]) / 5) < 0.02
    assert any("Running 6 invocations of compute_msm() for 2^16 inputs" in ln for ln in lines)
