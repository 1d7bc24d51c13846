#!/usr/bin/env python3
"""Does the engine's stage timing (HIP events between the stages) cost time?  30 MSMs with timing off / on, interleaved."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import webgpu_msm_bls12_377_amd as msm
import bench

n = 1 << 20
eng = msm.MsmEngine(n, device=0)
d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
eng.generate_bases_device(0x377, n, d_points.data_ptr())
d_scalars = torch.frombuffer(bytearray(bench.seeded_scalars(0x5CA1A5, n)), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
pp, sp = d_points.data_ptr(), d_scalars.data_ptr()
for _ in range(5):
    eng.msm_device(pp, sp, n)
res = {False: [], True: []}
for rep in range(6):
    for on in (False, True):
        eng.set_timing(on)
        t0 = time.perf_counter()
        for _ in range(20):
            eng.msm_device(pp, sp, n)
        res[on].append((time.perf_counter() - t0) * 1e3 / 20)
for on in (False, True):
    print("timing %-5s median %.4f ms  min %.4f" % (on, statistics.median(res[on]), min(res[on])))
