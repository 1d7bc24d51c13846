#!/usr/bin/env python3
"""A few fixed-base MSMs over the 20-bit-window precomputed table, for a kernel trace: python tools/run_wide.py [LOG_N] [CALLS]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import webgpu_msm_bls12_377_amd as msm
import bench

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 1 << log_n
eng = msm.MsmEngine(n, device=0)
d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
eng.generate_bases_device(0x377, n, d_points.data_ptr())
d_scalars = torch.frombuffer(bytearray(bench.seeded_scalars(0x5CA1A5, n)), dtype=torch.uint8).cuda()
eng.set_precompute_window(20)
eng.set_bases_precomputed_device(d_points.data_ptr(), n)
torch.cuda.synchronize()
for _ in range(calls):
    out = eng.msm_fixed_base_device(d_scalars.data_ptr(), n)
print(out[:8].hex())
