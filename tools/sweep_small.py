#!/usr/bin/env python3
"""ms per MSM (stage timing off, median of 40 calls) and stage times (a second pass with timing on) for small inputs
(2^8 .. 2^17), inputs resident."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import webgpu_msm_bls12_377_amd as msm
import bench

cap = 1 << 17
eng = msm.MsmEngine(cap, device=0)
d_points = torch.empty(96 * cap, dtype=torch.uint8, device="cuda")
eng.generate_bases_device(0x377, cap, d_points.data_ptr())
d_scalars = torch.frombuffer(bytearray(bench.seeded_scalars(0x5CA1A5, cap)), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
pp, sp = d_points.data_ptr(), d_scalars.data_ptr()
for log_n in (8, 10, 12, 14, 15, 16, 17):
    n = 1 << log_n
    for _ in range(3):
        eng.msm_device(pp, sp, n)
    ts = []
    for _ in range(40):  # the figure: stage timing off (its events between the stages cost ~10 us each at these sizes)
        t0 = time.perf_counter()
        eng.msm_device(pp, sp, n)
        ts.append((time.perf_counter() - t0) * 1e3)
    eng.set_timing(True)
    tt, st = [], []
    for _ in range(20):
        t0 = time.perf_counter()
        eng.msm_device(pp, sp, n)
        tt.append((time.perf_counter() - t0) * 1e3)
        st.append(eng.stage_ms())
    eng.set_timing(False)
    med = {k: round(statistics.median(s[k] for s in st), 3) for k in st[0]}
    print("2^%-2d  %.3f ms  (min %.3f; with stage timing on %.3f)  %s" % (log_n, statistics.median(ts), min(ts), statistics.median(tt), med), flush=True)
