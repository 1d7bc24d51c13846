#!/usr/bin/env python3
"""Upper bound of what affine base records buy the accumulation kernel: the same 2^20 workload through
msm_device (projective 256-byte records, 8 products per addition) and through a resident affine table
(msm_fixed_base_device, 160-byte records, 7 products), interleaved in one process; stage times from HIP events."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import webgpu_msm_bls12_377_amd as msm
    import bench

    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << log_n
    a = msm.MsmEngine(n, device=0)
    b = msm.MsmEngine(n, device=0)
    d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
    a.generate_bases_device(0x377, n, d_points.data_ptr())
    d_scalars = torch.frombuffer(bytearray(bench.seeded_scalars(0x5CA1A5, n)), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    pp, sp = d_points.data_ptr(), d_scalars.data_ptr()
    t0 = time.perf_counter()
    b.set_bases_device(pp, n)
    print("set_bases (affine table): %.2f ms" % ((time.perf_counter() - t0) * 1e3))
    ref = a.msm_device(pp, sp, n)
    assert b.msm_fixed_base_device(sp, n) == ref
    a.set_timing(True)
    b.set_timing(True)
    res = {"projective": ([], []), "affine": ([], [])}
    for _ in range(8):
        for name, fn, eng in (("projective", lambda: a.msm_device(pp, sp, n), a), ("affine", lambda: b.msm_fixed_base_device(sp, n), b)):
            t0 = time.perf_counter()
            for _ in range(6):
                fn()
            res[name][0].append((time.perf_counter() - t0) * 1e3 / 6)
            res[name][1].append(eng.stage_ms())
    for name, (ms, st) in res.items():
        med = {k: round(statistics.median(s[k] for s in st), 4) for k in st[0]}
        print("%-10s median %.4f ms  stages %s" % (name, statistics.median(ms), med), flush=True)


if __name__ == "__main__":
    main()
