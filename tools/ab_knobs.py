#!/usr/bin/env python3
"""Interleaved A/B timing of engine knobs in ONE process (same box, same thermal state).

    python tools/ab_knobs.py --log-n 20 --reps 6 --iters 8 "MSM377_GLV=0" "MSM377_GLV=1 MSM377_SEG_GLV=96" ...

Every configuration is a set of MSM377_* environment knobs read at context creation; one engine per configuration
is created up front, then the configurations are timed round-robin `reps` times, `iters` MSMs each.  Prints the
median / min ms per MSM and the median accumulate-kernel ms of every configuration.
"""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("configs", nargs="+")
    args = ap.parse_args()
    import torch
    import webgpu_msm_bls12_377_amd as msm
    import bench

    n = 1 << args.log_n
    dev = torch.device("cuda:0")
    d_points = torch.empty(96 * n, dtype=torch.uint8, device=dev)
    d_scalars = torch.frombuffer(bytearray(bench.seeded_scalars(0x5CA1A5, n)), dtype=torch.uint8).to(dev)
    engines = []
    for cfg in args.configs:
        kv = dict(x.split("=", 1) for x in cfg.split())
        for k, v in kv.items():
            os.environ[k] = v
        eng = msm.MsmEngine(n, device=0)
        for k in kv:
            del os.environ[k]
        engines.append(eng)
    engines[0].generate_bases_device(0x377, n, d_points.data_ptr())
    torch.cuda.synchronize()
    ref = None
    for eng in engines:  # warm-up + agreement
        out = eng.msm_device(d_points.data_ptr(), d_scalars.data_ptr(), n)
        ref = ref or out
        assert out == ref, "configurations disagree"
    ms = [[] for _ in engines]
    acc = [[] for _ in engines]
    red = [[] for _ in engines]
    tail = [[] for _ in engines]
    front = [[] for _ in engines]  # convert / decompose / sort / accumulate stage (work list + kernel + merge)
    for _ in range(args.reps):
        for i, eng in enumerate(engines):
            t0 = time.perf_counter()
            for _ in range(args.iters):
                eng.msm_device(d_points.data_ptr(), d_scalars.data_ptr(), n)
            ms[i].append((time.perf_counter() - t0) * 1e3 / args.iters)
            eng.set_timing(True)  # one more call for the stage times: their events cost ~10 us each, kept out of the figure above
            eng.msm_device(d_points.data_ptr(), d_scalars.data_ptr(), n)
            eng.set_timing(False)
            acc[i].append(eng.stage_ms()["accumulate_kernel"])
            red[i].append(eng.stage_ms()["reduce"])
            tail[i].append(eng.stage_ms()["tail"])
            st = eng.stage_ms()
            front[i].append((st["convert"], st["decompose"], st["sort"], st["accumulate"]))
    for cfg, m, a, r, t, f in zip(args.configs, ms, acc, red, tail, front):
        fm = [statistics.median(x[k] for x in f) for k in range(4)]
        print("%-48s median %.4f  min %.4f  acc_kernel %.4f  reduce %.4f  tail %.4f  | convert %.3f decompose %.3f sort %.3f acc_stage %.3f" % (cfg, statistics.median(m), min(m), statistics.median(a), statistics.median(r), statistics.median(t), fm[0], fm[1], fm[2], fm[3]), flush=True)


if __name__ == "__main__":
    main()
