#!/usr/bin/env python3
"""Per-piece timeline of the host tail (MSM377_TAIL_TRACE=1) for several contexts created one after the other:
python tools/tail_trace.py [LOG_N] [CONTEXTS]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MSM377_TAIL_TRACE"] = "1"
import torch
import webgpu_msm_bls12_377_amd as msm
import bench

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
count = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 1 << log_n
engines = [msm.MsmEngine(n, device=0) for _ in range(count)]
d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
engines[0].generate_bases_device(0x377, n, d_points.data_ptr())
d_scalars = torch.frombuffer(bytearray(bench.seeded_scalars(0x5CA1A5, n)), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
for rep in range(4):
    for i, eng in enumerate(engines):
        sys.stderr.write("context %d: " % i)
        sys.stderr.flush()
        eng.msm_device(d_points.data_ptr(), d_scalars.data_ptr(), n)
