#!/usr/bin/env python3
"""Runs window_partials_device for windows [b, b + c) of the 2^20 workload a few times (for rocprofv3 --kernel-trace)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import webgpu_msm_bls12_377_amd as msm
import bench
b, c = int(sys.argv[1]), int(sys.argv[2])
n = 1 << 20
eng = msm.MsmEngine(n, device=0)
d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
eng.generate_bases_device(0x377, n, d_points.data_ptr())
d_scalars = torch.frombuffer(bytearray(bench.seeded_scalars(0x5CA1A5, n)), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
for _ in range(4):
    eng.window_partials_device(d_points.data_ptr(), d_scalars.data_ptr(), n, b, c)
