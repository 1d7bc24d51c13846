#!/usr/bin/env python3
"""Stage times of ONE fixed-base MSM over each kind of resident table (plain affine, precomputed 16-bit windows, precomputed
20-bit windows): python tools/stage_fixed.py [LOG_N] [REPS].  HIP-event stage timing on (costs ~10 us per stage boundary)."""
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
d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
eng.generate_bases_device(0x377, n, d_points.data_ptr())
d_scalars = torch.frombuffer(bytearray(bench.seeded_scalars(0x5CA1A5, n)), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
ref = None
for name, setup in (("plain affine table", lambda: eng.set_bases_device(d_points.data_ptr(), n)),
                    ("precomputed, 16-bit windows", lambda: (eng.set_precompute_window(16), eng.set_bases_precomputed_device(d_points.data_ptr(), n))),
                    ("precomputed, 20-bit windows", lambda: (eng.set_precompute_window(20), eng.set_bases_precomputed_device(d_points.data_ptr(), n)))):
    t0 = time.perf_counter()
    setup()
    t_set = (time.perf_counter() - t0) * 1e3
    for _ in range(5):
        out = eng.msm_fixed_base_device(d_scalars.data_ptr(), n)
    ref = ref or out
    assert out == ref, name
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.msm_fixed_base_device(d_scalars.data_ptr(), n)
    ms = (time.perf_counter() - t0) * 1e3 / reps
    eng.set_timing(True)
    st = []
    for _ in range(reps):
        eng.msm_fixed_base_device(d_scalars.data_ptr(), n)
        st.append(eng.stage_ms())
    eng.set_timing(False)
    med = {k: round(statistics.median(s[k] for s in st), 3) for k in st[0]}
    print("%-30s set_bases %.1f ms; single MSM %.3f ms; stages %s" % (name, t_set, ms, med), flush=True)
