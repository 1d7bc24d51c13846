"""Randomized parity soak for the Edwards-BLS12 entry points (not part of the test suite): sizes 1..70001, uniform /
short / skewed scalars, host and device entry points, against the CPU oracle.  ~90 s; exit code 1 on any mismatch."""
import os, sys, random, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import webgpu_msm_bls12_377_amd as msm
import util, pyref as R
oracle = util.load_oracle()
eng = msm.MsmEngine(1 << 17)
rnd = random.Random(777)
t0 = time.time(); bad = 0; cases = 0
while time.time() - t0 < 90:
    n = rnd.choice([1, 2, 3, 64, 65, 257, 1000, 4096, 10000, 33333, 70001])
    seed = rnd.randrange(1 << 30)
    r2 = random.Random(seed)
    pts = util.oracle_ed_gen_points(oracle, n, r2.randrange(1, 1 << 200), r2.randrange(1, 1 << 200))
    mode = rnd.choice(["uniform", "small", "skew"])
    if mode == "uniform":
        ks = [r2.randrange(R.R_ORDER) for _ in range(n)]
    elif mode == "small":
        ks = [r2.randrange(1 << r2.choice([1, 16, 17, 100])) for _ in range(n)]
    else:
        hot = [r2.randrange(R.R_ORDER) for _ in range(3)]
        ks = [r2.choice(hot) if r2.random() < 0.9 else r2.randrange(R.R_ORDER) for _ in range(n)]
    sb = R.encode_scalars(ks)
    exp = util.oracle_ed_msm(oracle, pts, sb)
    if eng.ed_msm(pts, sb) != exp:
        bad += 1; print("MISMATCH host", n, seed, mode, flush=True)
    d_p = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda(); d_s = torch.frombuffer(bytearray(sb), dtype=torch.uint8).cuda()
    if eng.ed_msm_device(d_p.data_ptr(), d_s.data_ptr(), n) != exp:
        bad += 1; print("MISMATCH device", n, seed, mode, flush=True)
    cases += 1
print("soak ed: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
