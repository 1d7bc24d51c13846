#!/usr/bin/env python3
"""msm377_g1_msm from host buffers (upload included), the tree in the current directory: median / min of CALLS calls, and the
device-resident figure beside it.  python tools/h2d_one.py [LOG_N] [CALLS]   (run with the tree's root as cwd: A/B of two trees)"""
import os, statistics, sys, time
sys.path.insert(0, os.getcwd())
import torch
import webgpu_msm_bls12_377_amd as msm
import bench

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 15
n = 1 << log_n
eng = msm.MsmEngine(n, device=0)
d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
eng.generate_bases_device(0x377, n, d_points.data_ptr())
pts = d_points.cpu().numpy().tobytes()
scal = bench.seeded_scalars(0x5CA1A5, n)
d_s = torch.frombuffer(bytearray(scal), dtype=torch.uint8).cuda()
ref = eng.msm(pts, scal)
for _ in range(30):
    assert eng.msm_device(d_points.data_ptr(), d_s.data_ptr(), n) == ref
th, td = [], []
for _ in range(calls):
    t0 = time.perf_counter(); eng.msm(pts, scal); th.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); eng.msm_device(d_points.data_ptr(), d_s.data_ptr(), n); td.append((time.perf_counter() - t0) * 1e3)
print("%s: host buffers median %.3f min %.3f ms;  device-resident median %.3f min %.3f ms" % (os.path.basename(os.getcwd()) or "repo", statistics.median(th), min(th), statistics.median(td), min(td)), flush=True)
