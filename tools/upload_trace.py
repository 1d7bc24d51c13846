#!/usr/bin/env python3
"""msm377_g1_msm from host buffers with MSM377_UPLOAD_TRACE=1: where the upload-inclusive time goes.  python tools/upload_trace.py [LOG_N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MSM377_UPLOAD_TRACE"] = "1"
import torch
import webgpu_msm_bls12_377_amd as msm
import bench

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
eng = msm.MsmEngine(n, device=0)
d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
eng.generate_bases_device(0x377, n, d_points.data_ptr())
pts = d_points.cpu().numpy().tobytes()
scal = bench.seeded_scalars(0x5CA1A5, n)
for _ in range(6):
    t0 = time.perf_counter()
    eng.msm(pts, scal)
    print("call %.3f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr, flush=True)
