"""The reference's benchmark sweep, mirrored (row f1 of SURVEY.md section 8).

Reference: src/submission/miscellaneous/full_benchmarks.ts:6-162 -- powers 16..20 inclusive, for
each: one first run (shader recompile forced only for the first power) + NUM_RUNS warm runs,
100 ms apart, wall clock around the whole compute_msm call, a markdown table, and a WARNING (not a
failure) when the result differs from the known answer.

Inputs: the harness's test-case files when `test_data_dir` holds them (known answers then apply,
host/test_data.py); otherwise the seeded synthetic workload of BASELINE.md section 3, generated on
the GPU, with no expected result to compare (the table is still produced).
    python -m webgpu_msm_bls12_377_amd.host.full_benchmarks [test-data-dir]
"""
import sys
import time
from typing import Dict, Optional

from .codecs import bigIntsToBufferLE
from .engine import MsmEngine
from .submission import compute_msm
from .test_data import load_test_case

DELAY = 100  # ms
NUM_RUNS = 5
START_POWER = 16
END_POWER = 20
R_ORDER = 8444461749428370424248824938781546531375899335154063827935233455917409239041


def delay(duration_ms: float) -> None:
    time.sleep(duration_ms / 1000.0)


def _synthetic_case(power: int) -> Dict[str, object]:
    import numpy as np
    import torch

    n = 1 << power
    with MsmEngine(n) as eng:
        d = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
        eng.generate_bases_device(0x377, n, d.data_ptr())
        points = d.cpu().numpy().tobytes()
    with np.errstate(over="ignore"):
        idx = np.arange(1, 4 * n + 1, dtype=np.uint64)
        z = np.uint64(0x5CA1A5) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    raw = z.astype("<u8").tobytes()
    scalars = b"".join((int.from_bytes(raw[32 * i : 32 * i + 32], "little") % R_ORDER).to_bytes(32, "little") for i in range(n))
    return {"bufferPoints": points, "bufferScalars": scalars, "expectedResult": None}


def full_benchmarks(test_data_dir: Optional[str] = None, start_power: int = START_POWER, end_power: int = END_POWER, out=print) -> str:
    all_results = {}
    out("Running BLS12-377 MSM benchmarks for powers %d to %d (inclusive)" % (start_power, end_power))
    do_recompile = True
    testcases = {}
    for power in range(start_power, end_power + 1):
        if test_data_dir:
            tc = load_test_case(power, test_data_dir)
            xy = []
            for p in tc["baseAffinePoints"]:
                xy += [p["x"], p["y"]]
            testcases[power] = {
                "bufferPoints": bigIntsToBufferLE(xy, 384),
                "bufferScalars": bigIntsToBufferLE(tc["scalars"], 256),
                "expectedResult": tc["expectedResult"],
            }
        else:
            testcases[power] = _synthetic_case(power)
    for power in range(start_power, end_power + 1):
        out("Running %d invocations of compute_msm() for 2^%d inputs, please wait..." % (NUM_RUNS + 1, power))
        tc = testcases[power]
        t0 = time.perf_counter()
        msm = compute_msm(tc["bufferPoints"], tc["bufferScalars"], False, do_recompile)
        do_recompile = False
        first = (time.perf_counter() - t0) * 1e3
        exp = tc["expectedResult"]
        if exp is not None and (msm["x"] != exp["x"] or msm["y"] != exp["y"]):
            out("WARNING: the result of compute_msm is incorrect for 2^%d" % power)
        delay(DELAY)
        runs = []
        for _ in range(NUM_RUNS):
            t0 = time.perf_counter()
            compute_msm(tc["bufferPoints"], tc["bufferScalars"], False, False)
            runs.append((time.perf_counter() - t0) * 1e3)
            delay(DELAY)
        all_results[power] = {
            "first_run_elapsed": first,
            "subsequent_runs": runs,
            "full_average": (first + sum(runs)) / (1 + len(runs)),
            "subsequent_average": sum(runs) / len(runs),
        }
    header = "| MSM size | 1st run |" + "".join(" Run %d |" % (i + 1) for i in range(NUM_RUNS))
    header += " Average (incl 1st) | Average (excl 1st) |\n|-|-|-|-|" + "-|" * NUM_RUNS + "\n"
    body = ""
    for power in range(start_power, end_power + 1):
        r = all_results[power]
        md = "| 2^%d | `%.2f` |" % (power, r["first_run_elapsed"])
        md += "".join(" `%.2f` |" % v for v in r["subsequent_runs"])
        body += md + " **`%.2f`** | **`%.2f`** |\n" % (r["full_average"], r["subsequent_average"])
    table = header + body.strip()
    out(table)
    out("(times in ms, host buffers in and affine result out: H2D upload included, as the reference measures)")
    return table


if __name__ == "__main__":
    full_benchmarks(sys.argv[1] if len(sys.argv) > 1 else None)
