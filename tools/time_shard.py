#!/usr/bin/env python3
"""Per-rank cost of a sharded MSM on ONE GPU, both partitionings (host/sharding.py): window sharding -- window_partials_device
for the block of windows a rank of an N-GPU run owns (N = 1, 2, 4, 8, 16) plus the host combine of 16 gathered records -- and
point sharding -- a complete MSM over a rank's slice of the points plus the sum of the rank results.  No collective involved.
    python tools/time_shard.py --log-n 20 | 22"""
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
            t0 = time.perf_counter()
            for _ in range(args.iters):
                eng.window_partials_device(pp, sp, n, b, c)
            ms = (time.perf_counter() - t0) * 1e3 / args.iters  # stage events off (they cost ~10 us per stage boundary)
            eng.set_timing(True)
            eng.window_partials_device(pp, sp, n, b, c)
            st = eng.stage_ms()
            eng.set_timing(False)
            print("world %2d rank %2d: %d windows  %.3f ms  %s" % (world, rank, c, ms, {k: round(v, 3) for k, v in st.items()}), flush=True)
    t0 = time.perf_counter()
    for _ in range(args.iters):
        msm.combine_partials(full)
    print("host combine of 16 records: %.3f ms" % ((time.perf_counter() - t0) * 1e3 / args.iters))
    # POINT sharding (host/sharding.py run_points): a rank runs a complete MSM -- all windows, its own host tail -- over its
    # slice of the points; the exchange is 96 bytes per rank and the final step adds `world` affine points.
    from webgpu_msm_bls12_377_amd.host.sharding import points_for_rank
    from webgpu_msm_bls12_377_amd.host.engine import add_points_bytes

    whole = eng.msm_device(pp, sp, n)
    for world in (1, 2, 4, 8):
        results = []
        for rank in range(world):
            first, count = points_for_rank(rank, world, n)
            results.append(eng.msm_device(pp + 96 * first, sp + 32 * first, count))
        assert add_points_bytes(b"".join(results)) == whole, "point shards do not add up to the whole MSM"
        for rank in sorted({0, world - 1}):
            first, count = points_for_rank(rank, world, n)
            t0 = time.perf_counter()
            for _ in range(args.iters):
                eng.msm_device(pp + 96 * first, sp + 32 * first, count)
            ms = (time.perf_counter() - t0) * 1e3 / args.iters  # stage timing off: the figure a rank would see
            print("points: world %2d rank %2d: %d points  %.3f ms" % (world, rank, count, ms), flush=True)
    t0 = time.perf_counter()
    for _ in range(args.iters):
        add_points_bytes(b"".join(results))
    print("host sum of 8 rank results: %.4f ms" % ((time.perf_counter() - t0) * 1e3 / args.iters))


if __name__ == "__main__":
    main()
