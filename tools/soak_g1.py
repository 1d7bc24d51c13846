"""Randomized parity soak (not part of the test suite): sizes 1..100003, uniform / short / skewed / near-r scalars and
scalars of 2^253 and more (the even and wide window geometries rerun those), batches of 5 (the twin context),
all three internal paths, host (chunked upload forced from 3000 points, both schedules), device, fixed-base and
precomputed-table (16- and 20-bit windows) entry points, against
the CPU oracle.  Runs for ~150 s; exit code 1 on any mismatch."""
import os, sys, random, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ["MSM377_UPLOAD_CHUNK_MIN"] = "3000"
import torch
import webgpu_msm_bls12_377_amd as msm
import util, pyref as R
oracle = util.load_oracle()
eng = msm.MsmEngine(1 << 17)
os.environ.update({"MSM377_UPLOAD_SORT_ONCE": "1", "MSM377_UPLOAD_CHUNKS": "6", "MSM377_UPLOAD_SPLIT": "12"})
eng_once = msm.MsmEngine(1 << 17)
for k in ("MSM377_UPLOAD_SORT_ONCE", "MSM377_UPLOAD_CHUNKS", "MSM377_UPLOAD_SPLIT"):
    del os.environ[k]
rnd = random.Random(int(os.environ.get("SOAK_SEED", "20261004")))
t0 = time.time(); bad = 0; cases = 0
while time.time() - t0 < float(os.environ.get("SOAK_SECONDS", "150")):
    n = rnd.choice([1, 2, 3, 5, 17, 64, 65, 255, 256, 257, 1000, 2999, 3000, 3001, 4096, 10000, 33333, 65536, 100003])
    seed = rnd.randrange(1 << 30)
    r2 = random.Random(seed)
    pts = util.oracle_gen_points(oracle, n, r2.randrange(1, 1 << 200), r2.randrange(1, 1 << 200))
    ks = R.encode_scalars(R.rand_scalars(seed, n))
    mode = rnd.choice(["uniform", "small", "skew", "top", "big"])
    if mode == "small":
        ks = R.encode_scalars([rnd.randrange(1 << rnd.choice([1, 16, 17, 64, 128])) for _ in range(n)])
    elif mode == "skew":
        hot = R.rand_scalars(seed, 3)
        ks = R.encode_scalars([rnd.choice(hot) if rnd.random() < 0.9 else rnd.randrange(R.R_ORDER) for _ in range(n)])
    elif mode == "top":
        ks = R.encode_scalars([R.R_ORDER - 1 - rnd.randrange(1 << 20) for _ in range(n)])
    elif mode == "big":  # a few scalars the short top windows cannot hold, below the error threshold 2^255 - 2^239
        kl = R.decode_scalars(ks)
        for _ in range(rnd.randrange(1, 4)):
            kl[rnd.randrange(n)] = rnd.choice([(1 << 253) + rnd.randrange(1 << 200), (0x7FFF << 238) + (0x7FFF << 223) + (0x7FFF << 208) + (1 << 207), (1 << 254) + rnd.randrange(1 << 250)])
        ks = R.encode_scalars(kl)
    exp = util.oracle_msm(oracle, pts, ks)
    for form, glv in (("edwards", "auto"), ("weierstrass", False), ("weierstrass", True)):
        eng.set_g1_form(form); eng.set_glv(glv)
        got = eng.msm(pts, ks)
        if got != exp:
            bad += 1
            print("MISMATCH", n, seed, mode, form, glv, flush=True)
    eng.set_g1_form("edwards"); eng.set_glv("auto")
    d_p = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda(); d_s = torch.frombuffer(bytearray(ks), dtype=torch.uint8).cuda()
    if eng.msm_device(d_p.data_ptr(), d_s.data_ptr(), n) != exp:
        bad += 1; print("MISMATCH device", n, seed, mode, flush=True)
    eng.set_bases(pts)
    if eng.msm_fixed_base(ks) != exp:
        bad += 1; print("MISMATCH fixed", n, seed, mode, flush=True)
    for bits in (16, 20):  # precomputed window multiples: 16 windows folded on the GPU / 13 wide windows over one bucket set
        eng.set_precompute_window(bits)
        eng.set_bases_precomputed(pts)
        if eng.msm_fixed_base(ks) != exp:
            bad += 1; print("MISMATCH precomputed", bits, n, seed, mode, flush=True)
        if n >= 64:  # a batch of 5 over the table: two halves on the twin context
            ks2 = R.encode_scalars(R.rand_scalars(seed + 1, n))
            d_b = torch.frombuffer(bytearray(ks + ks2 + ks + ks2 + ks), dtype=torch.uint8).cuda()
            exp2 = util.oracle_msm(oracle, pts, ks2)
            if eng.msm_fixed_base_batch_device(d_b.data_ptr(), n, 5) != [exp, exp2, exp, exp2, exp]:
                bad += 1; print("MISMATCH precomputed batch", bits, n, seed, mode, flush=True)
    eng.set_precompute_window(16)
    if eng_once.msm(pts, ks) != exp:  # host buffers, sorted once (MSM377_UPLOAD_SORT_ONCE=1), 7 chunks
        bad += 1; print("MISMATCH sort-once upload", n, seed, mode, flush=True)
    cases += 1
print("soak: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
