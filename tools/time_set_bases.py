#!/usr/bin/env python3
"""The batched affine conversion alone (msm377_g1_set_bases_device: k_affine_up, host inversion, k_affine_down with the
GPU to themselves): wall time per call.  Under rocprofv3 --kernel-trace --stats it gives the two kernels' durations
without the sort beside them.  python tools/time_set_bases.py [LOG_N] [CALLS]"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import webgpu_msm_bls12_377_amd as msm

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << log_n
eng = msm.MsmEngine(n, device=0)
d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
eng.generate_bases_device(0x377, n, d_points.data_ptr())
torch.cuda.synchronize()
ts = []
for _ in range(calls):
    t0 = time.perf_counter()
    eng.set_bases_device(d_points.data_ptr(), n)
    ts.append((time.perf_counter() - t0) * 1e3)
print("set_bases_device 2^%d: median %.3f ms  min %.3f" % (log_n, statistics.median(ts), min(ts)))
