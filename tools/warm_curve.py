#!/usr/bin/env python3
"""ms per MSM as a function of how many MSMs the process has run (first process on a fresh box: how long until the
figure settles?): python tools/warm_curve.py [LOG_N] [CALLS]; prints the mean of every block of 10 calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t_start = time.perf_counter()
import torch
import webgpu_msm_bls12_377_amd as msm
import bench

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 400
n = 1 << log_n
eng = msm.MsmEngine(n, device=0)
d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
eng.generate_bases_device(0x377, n, d_points.data_ptr())
d_scalars = torch.frombuffer(bytearray(bench.seeded_scalars(0x5CA1A5, n)), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
print("setup done %.1f s after start" % (time.perf_counter() - t_start))
ts = []
for i in range(calls):
    t0 = time.perf_counter()
    eng.msm_device(d_points.data_ptr(), d_scalars.data_ptr(), n)
    ts.append((time.perf_counter() - t0) * 1e3)
for b in range(0, calls, 10):
    blk = ts[b : b + 10]
    print("calls %3d..%3d  mean %.3f ms  max %.3f" % (b, b + len(blk) - 1, sum(blk) / len(blk), max(blk)))
