"""Interleaved timing of msm377_g1_msm (host buffers, upload included) with 2 / 4 / 6 / 8 copy workers, one process."""
import os, sys, time, statistics
sys.path.insert(0, os.getcwd())
import torch
import webgpu_msm_bls12_377_amd as msm
import bench
n = 1 << 20
d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
scal = bench.seeded_scalars(0x5CA1A5, n)
engs = {}
for nt in ("2", "4", "6", "8"):
    os.environ["MSM377_H2D_THREADS"] = nt
    engs[nt] = msm.MsmEngine(n, device=0)
engs["4"].generate_bases_device(0x377, n, d_points.data_ptr())
pts = d_points.cpu().numpy().tobytes()
ref = None
for name, e in engs.items():
    r = e.msm(pts, scal); r = e.msm(pts, scal)
    ref = ref or r
    assert r == ref
for rep in range(3):
    for name, e in engs.items():
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); e.msm(pts, scal); ts.append((time.perf_counter() - t0) * 1e3)
        print("threads", name, "median %.3f min %.3f" % (statistics.median(ts), min(ts)), flush=True)
