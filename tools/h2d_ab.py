"""Interleaved timing of msm377_g1_msm (host buffers, upload included) with different chunk splits, one process."""
import os, sys, time, statistics
sys.path.insert(0, os.getcwd())
import torch
import webgpu_msm_bls12_377_amd as msm
import bench
n = 1 << 20
d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
scal = bench.seeded_scalars(0x5CA1A5, n)
engs = {}
for name, v, split, k in (("5 chunks 16", "262144", "16", "5"), ("4 chunks 20", "262144", "20", "4"), ("6 chunks 12", "262144", "12", "6"), ("5 chunks 10", "262144", "10", "5"), ("5 chunks 22", "262144", "22", "5"), ("8 chunks 10", "262144", "10", "8"), ("3 chunks 25", "262144", "25", "3"), ("whole", "99999999999", "50", "2")):
    os.environ["MSM377_UPLOAD_CHUNK_MIN"] = v
    os.environ["MSM377_UPLOAD_SPLIT"] = split
    os.environ["MSM377_UPLOAD_CHUNKS"] = k
    engs[name] = msm.MsmEngine(n, device=0)
engs["whole"].generate_bases_device(0x377, n, d_points.data_ptr())
pts = d_points.cpu().numpy().tobytes()
ref = None
for name, e in engs.items():
    r = e.msm(pts, scal); r = e.msm(pts, scal)
    ref = ref or r
    assert r == ref
for rep in range(3):
    for name, e in engs.items():
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); e.msm(pts, scal); ts.append((time.perf_counter() - t0) * 1e3)
        print(name, "median %.3f min %.3f" % (statistics.median(ts), min(ts)), flush=True)
d_s = torch.frombuffer(bytearray(scal), dtype=torch.uint8).cuda()
e = engs["whole"]
ts = []
for _ in range(5):
    t0 = time.perf_counter(); e.msm_device(d_points.data_ptr(), d_s.data_ptr(), n); ts.append((time.perf_counter() - t0) * 1e3)
print("device-resident median %.3f" % statistics.median(ts))
