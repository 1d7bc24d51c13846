#!/usr/bin/env python3
"""Stage times of one Edwards-BLS12 MSM (BASELINE.json config 3): python tools/stage_ed.py [LOG_N] [REPS]."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import webgpu_msm_bls12_377_amd as msm
import bench

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 1 << log_n
eng = msm.MsmEngine(n, device=0)
d_points = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
eng.ed_generate_bases_device(0x377, n, d_points.data_ptr())
d_scalars = torch.frombuffer(bytearray(bench.seeded_scalars(0x5CA1A5, n)), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
for _ in range(10):
    eng.ed_msm_device(d_points.data_ptr(), d_scalars.data_ptr(), n)
t0 = time.perf_counter()
for _ in range(reps):
    eng.ed_msm_device(d_points.data_ptr(), d_scalars.data_ptr(), n)
ms = (time.perf_counter() - t0) * 1e3 / reps
eng.set_timing(True)
st = []
for _ in range(reps):
    eng.ed_msm_device(d_points.data_ptr(), d_scalars.data_ptr(), n)
    st.append(eng.stage_ms())
med = {k: round(statistics.median(s[k] for s in st), 3) for k in st[0]}
print("Edwards-BLS12 2^%d: %.3f ms; stages %s" % (log_n, ms, med))
