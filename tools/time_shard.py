#!/usr/bin/env python3
"""Per-rank cost of a window-sharded MSM on ONE GPU: times window_partials_device for the block of windows a rank
of an N-GPU run owns (N = 1, 2, 4, 8), plus the host combine of 16 gathered records.  No collective involved."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    import torch
    import webgpu_msm_bls12_377_amd as msm
    import bench

    n = 1 << args.log_n
    eng = msm.MsmEngine(n, device=0)
    d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
    eng.generate_bases_device(0x377, n, d_points.data_ptr())
    d_scalars = torch.frombuffer(bytearray(bench.seeded_scalars(0x5CA1A5, n)), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    pp, sp = d_points.data_ptr(), d_scalars.data_ptr()
    full = eng.window_partials_device(pp, sp, n, 0, 16)
    for world in (1, 2, 4, 8, 16):
        for rank in sorted({0, world - 1}):  # the first rank, and the one that owns the top window (13 significant bits: long rows)
            b, c = msm.windows_for_rank(rank, world)
            eng.window_partials_device(pp, sp, n, b, c)
            eng.set_timing(True)
            t0 = time.perf_counter()
            for _ in range(args.iters):
                eng.window_partials_device(pp, sp, n, b, c)
            ms = (time.perf_counter() - t0) * 1e3 / args.iters
            st = eng.stage_ms()
            eng.set_timing(False)
            print("world %2d rank %2d: %d windows  %.3f ms  %s" % (world, rank, c, ms, {k: round(v, 3) for k, v in st.items()}), flush=True)
    t0 = time.perf_counter()
    for _ in range(args.iters):
        msm.combine_partials(full)
    print("host combine of 16 records: %.3f ms" % ((time.perf_counter() - t0) * 1e3 / args.iters))


if __name__ == "__main__":
    main()
