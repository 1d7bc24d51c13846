#!/usr/bin/env python3
"""Do two fixed-base batches on two contexts of ONE GPU overlap?  Each context has its own streams and buffers; two host
threads each run a batch of B MSMs (ctypes releases the GIL) -- against one context running 2 B.
python tools/twin_probe.py [LOG_N] [B] [PRECOMP_BITS: 0 | 16 | 20] [CONTEXTS]
(Run with MSM377_TWIN_BATCH=0: the engine's own batches split over a twin context since this probe.)"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import webgpu_msm_bls12_377_amd as msm
import bench

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 20
K = int(sys.argv[4]) if len(sys.argv) > 4 else 2
n = 1 << log_n
engs = [msm.MsmEngine(n, device=0) for _ in range(K)]
d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
engs[0].generate_bases_device(0x377, n, d_points.data_ptr())
sets = b"".join(bench.seeded_scalars(0x5CA1A5 + b, n) for b in range(K * B))
d_scalars = torch.frombuffer(bytearray(sets), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
for e in engs:
    if bits:
        e.set_precompute_window(bits)
        e.set_bases_precomputed_device(d_points.data_ptr(), n)
    else:
        e.set_bases_device(d_points.data_ptr(), n)
base = d_scalars.data_ptr()
ref = engs[0].msm_fixed_base_batch_device(base, n, K * B)
for rep in range(3):
    t0 = time.perf_counter()
    one = engs[0].msm_fixed_base_batch_device(base, n, K * B)
    t_one = (time.perf_counter() - t0) * 1e3
    out = [None] * K
    def run(k):
        out[k] = engs[k].msm_fixed_base_batch_device(base + k * B * n * 32, n, B)
    th = [threading.Thread(target=run, args=(k,)) for k in range(K)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    t_two = (time.perf_counter() - t0) * 1e3
    assert one == ref and sum(out, []) == ref
    print("table bits %d, 2^%d: one context, batch %d: %.3f ms/MSM; %d contexts x %d concurrently: %.3f ms/MSM" % (bits, log_n, K * B, t_one / (K * B), K, B, t_two / (K * B)), flush=True)
