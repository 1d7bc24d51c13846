#!/usr/bin/env python3
"""Stage times of the 2^20 Edwards-BLS12 MSM (BASELINE.json config 3), inputs resident."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import webgpu_msm_bls12_377_amd as msm
import bench
n = 1 << 20
eng = msm.MsmEngine(n, device=0)
sub = 2111115437357092606062206234695386632838870926408408195193685246394721360383
sh = bench.seeded_scalars(0x5CA1A5, n)
ks = b"".join((int.from_bytes(sh[32 * i : 32 * i + 32], "little") % sub).to_bytes(32, "little") for i in range(n))
d_points = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
eng.ed_generate_bases_device(0xED, n, d_points.data_ptr())
d_scalars = torch.frombuffer(bytearray(ks), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
for _ in range(3):
    eng.ed_msm_device(d_points.data_ptr(), d_scalars.data_ptr(), n)
eng.set_timing(True)
ts, st = [], []
for _ in range(20):
    t0 = time.perf_counter()
    eng.ed_msm_device(d_points.data_ptr(), d_scalars.data_ptr(), n)
    ts.append((time.perf_counter() - t0) * 1e3)
    st.append(eng.stage_ms())
print("ed 2^20: %.3f ms  %s" % (statistics.median(ts), {k: round(statistics.median(s[k] for s in st), 3) for k in st[0]}))
