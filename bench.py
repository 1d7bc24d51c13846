#!/usr/bin/env python3
"""bench.py -- ms per 2^20-point BLS12-377 G1 MSM on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one whole MSM over the synthetic workload of BASELINE.md section 3 (configs[1]):
n = 2^20 points P_i = [a_i]G (a_i = SplitMix64(0x377), generated on the GPU by the engine),
scalars uniform below r from SplitMix64(0x5ca1a5); wire format of the reference harness
(src/ui/AllBenchmarks.tsx:57-68).  Inputs are resident in HBM when the timed region starts;
every step runs convert -> decompose -> sort -> accumulate -> reduce -> D2H of the partial
records -> host Horner + inversion and ends with the affine result on the host, like the
reference's timing bracket (src/ui/Benchmark.tsx:31-34).

N > 1: the 16 window subtasks are sharded over the ranks (rank g owns a contiguous block of
windows, no data-path collective), followed by ONE RCCL all-gather of the partial records and
the Horner combine -- the same 2^20-point problem on more GPUs, so "scaling" is "strong".

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself: the parent
spawns `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process before anything has
touched the GPU (no re-exec of a GPU-initialised process) and relays its output and exit code.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (bucket accumulation) with
HIP-event timings taken by the engine on its own stream; `cpu_baseline` times the CPU oracle
(a port of the reference pipeline, not the reference's WASM) on the same inputs on this host
and doubles as a full-size bit-exactness check.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R_ORDER = 8444461749428370424248824938781546531375899335154063827935233455917409239041
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
LOG_N = 20
NUM_WINDOWS = 16
NUM_BUCKETS = 32768


# Secondary, truly binding roof of the accumulation kernel: 32x32+64-bit multiply-adds.
# A field product on 13 x 29-bit limbs with 14 reduction steps is 13*13 + 14*12 = 337 v_mad_u64_u32 (csrc/field29.hpp
# mul_lz); a bucket addition is 8 products with projective base records, 7 with affine ones (csrc/te377.hpp; XYZZ: 10);
# the chip-wide peak is the microbenchmark figure at 8 waves/SIMD (profiles/microbench_r01.txt: mad64 33605 Gop/s).
MADS_PER_FIELD_PRODUCT = 13 * 13 + 14 * 12
MAD_PEAK_GLANEOPS = 33605.0
ALU_PEAK_GLANEOPS = 57062.0  # 32-bit add / logic lane-ops per second at 8 waves/SIMD, same file (add32x3)


def isa_loop_mix(products):
    """VALU instructions and v_mad_u64_u32 per bucket addition, counted in the code object's accumulation loop by
    tools/isa_mix.py (profiles/rNN_*/isa_mix.json, newest first); None when no listing is committed."""
    prof = os.path.join(ROOT, "profiles")
    key = {7: "k_accumulate<TeDev,TeAffBase>", 8: "k_accumulate<TeDev>", 10: "k_accumulate<G1Dev>"}.get(products)
    for tag in sorted(os.listdir(prof), reverse=True) if os.path.isdir(prof) else []:
        path = os.path.join(prof, tag, "isa_mix.json")
        if key and os.path.exists(path):
            with open(path) as f:
                loop = json.load(f).get(key, {}).get("hottest_loop")
            if loop:
                return tag, loop["valu"], loop["v_mad_u64_u32"]
    return None, None, None


def launch_command(gpus, argv, port):
    """argv of the child that runs the N ranks: torch.distributed.run on 127.0.0.1 (the container hostname may not
    resolve), this very script and its own arguments."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def self_launch(args, argv):
    """--gpus N > 1 from a plain `python bench.py`: start the ranks as a child process group and relay.  Nothing in
    this process has initialised HIP at this point (torch is not even imported)."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = launch_command(args.gpus, argv, free_port())
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def latest_pmc_summary():
    """(tag, summary) of the newest committed rocprofv3 PMC summary (profiles/rNN_*/pmc_summary.json)."""
    prof = os.path.join(ROOT, "profiles")
    tags = [d for d in os.listdir(prof) if os.path.exists(os.path.join(prof, d, "pmc_summary.json"))] if os.path.isdir(prof) else []
    # newest first: by round number, and within a round the pass named in profiles/LATEST (else the name, descending)
    latest = ""
    if os.path.exists(os.path.join(prof, "LATEST")):
        with open(os.path.join(prof, "LATEST")) as f:
            latest = f.read().strip()
    tags.sort(key=lambda d: (d == latest, d.split("_")[0], d), reverse=True)
    for tag in tags:
        with open(os.path.join(prof, tag, "pmc_summary.json")) as f:
            return tag, json.load(f)
    return None, {}


def algorithmic_bytes(n, glv=False):
    """B_alg(n) = 32n + 96n + W*96n + 2*W*2^15*144 + 96 (whole MSM), and the
    share of the bucket-accumulation launch: W*96n gathered + W*2^15*144 written.  Behind the GLV front end the
    launch gathers the same 16n coordinates pairs (8 windows x 2n points) and writes 8 windows of buckets."""
    whole = 32 * n + 96 * n + NUM_WINDOWS * 96 * n + 2 * NUM_WINDOWS * NUM_BUCKETS * 144 + 96
    if glv:
        accumulate = 8 * 96 * 2 * n + 8 * NUM_BUCKETS * 144
    else:
        accumulate = NUM_WINDOWS * 96 * n + NUM_WINDOWS * NUM_BUCKETS * 144
    return whole, accumulate


def pmc_traffic_bytes(log_n):
    """(bytes, source) -- HBM bytes per k_accumulate launch from the newest committed rocprofv3 PMC passes (FETCH_SIZE
    and WRITE_SIZE in separate passes; FETCH_SIZE doubled per the gfx950 note in /opt/skills/guides/MI355X_MICROARCH.md).
    A STATIC figure from that profile, not a measurement of this run (counters need the profiler); only valid for the
    workload the passes were taken on (2^20).  (None, None) otherwise."""
    if log_n != 20:
        return None, None
    tag, summary = latest_pmc_summary()
    k = summary.get("k_accumulate", {})
    if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
        nbytes = int((2 * k["FETCH_SIZE"]["mean_per_launch"] + k["WRITE_SIZE"]["mean_per_launch"]) * 1024)
        return nbytes, "static: profiles/%s/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload; FETCH_SIZE x2 per the gfx950 note)" % tag
    return None, None


def seeded_scalars(seed, n):
    """n scalars uniform below r: four SplitMix64 outputs per scalar, reduced mod r (identical to
    tests/pyref.py:rand_scalars)."""
    import numpy as np

    with np.errstate(over="ignore"):
        idx = np.arange(1, 4 * n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    raw = z.astype("<u8").tobytes()
    out = bytearray(32 * n)
    for i in range(n):
        v = int.from_bytes(raw[32 * i : 32 * i + 32], "little") % R_ORDER
        out[32 * i : 32 * i + 32] = v.to_bytes(32, "little")
    return bytes(out)


def fixed64_scalar_sets(torch, scalars_host, n, batch):
    """`batch` scalar sets on the device: the seeded set rotated by b entries (distinct MSMs, no 2 GB of host bignum work)."""
    d_one = torch.frombuffer(bytearray(scalars_host), dtype=torch.uint8).cuda().view(n, 32)
    return torch.cat([torch.roll(d_one, shifts=b, dims=0) for b in range(batch)]).contiguous().view(-1)


def fixed64_expected(n, batch, scalars_host):
    """Closed form of every MSM of the fixed-base batch: P_i = [a_i]G with a_i = SplitMix64(0x377)_i and scalar set b = the
    seeded set rotated by b, so MSM_b = [sum_i k_((i - b) mod n) a_i mod r] G -- one oracle scalar multiplication of the
    generator each.  Checker only (bench.py after its timed region, tests/test_bench_gpu.py)."""
    import ctypes
    import operator

    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util

    oracle = util.load_oracle()
    with np.errstate(over="ignore"):
        z = np.uint64(0x377) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    a = [int(v) or 1 for v in z.tolist()]
    ks = [int.from_bytes(scalars_host[32 * i : 32 * i + 32], "little") for i in range(n)]
    gen = ctypes.create_string_buffer(96)
    oracle.oracle_g1_generator(ctypes.addressof(gen))
    out = []
    for b in range(batch):
        rot = ks[n - b :] + ks[: n - b] if b else ks  # torch.roll(shifts=b): element i of set b is k[(i - b) mod n]
        total = sum(map(operator.mul, rot, a)) % R_ORDER
        exp = ctypes.create_string_buffer(96)
        assert oracle.oracle_g1_scalar_mul(gen.raw, total.to_bytes(32, "little"), 32, ctypes.addressof(exp)) == 0
        out.append(exp.raw)
    return out


def side_workload(args, torch, msm, n):
    """Informational lines for BASELINE.json configs[2] (Edwards) and configs[4] (64 fixed-base MSMs)."""
    eng = msm.MsmEngine(n, device=0)
    scalars_host = seeded_scalars(0x5CA1A5, n)
    out = {"n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "unit": "ms", "higher_is_better": False, "data": "synthetic",
           "scaling": "strong", "vs_baseline": None}
    if args.workload == "ed":
        sub = 2111115437357092606062206234695386632838870926408408195193685246394721360383  # AleoConstants.ts:5
        ks = b"".join((int.from_bytes(scalars_host[32 * i : 32 * i + 32], "little") % sub).to_bytes(32, "little") for i in range(n))
        d_points = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
        eng.ed_generate_bases_device(0xED, n, d_points.data_ptr())
        d_scalars = torch.frombuffer(bytearray(ks), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        step = lambda: eng.ed_msm_device(d_points.data_ptr(), d_scalars.data_ptr(), n)  # noqa: E731
        per_step = 1
        out.update({"metric": "ms per 2^%d Twisted-Edwards BLS12 MSM" % args.log_n, "dtype": "u32 (9 x 29-bit limbs)",
                    "config": {"workload": "2^%d Edwards-BLS12 MSM (extended coordinates, add-2008-hwcd-3), 64-byte points, inputs resident" % args.log_n}})
        alg = 32 * n + 64 * n + NUM_WINDOWS * 64 * n + 2 * NUM_WINDOWS * NUM_BUCKETS * 128 + 64
    else:
        batch = 64
        d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
        eng.generate_bases_device(0x377, n, d_points.data_ptr())
        t_set = time.perf_counter()
        # Default: the plain affine table.  The precomputed-window table (MSM377_BENCH_PRECOMPUTE=1) saves the reduction of
        # 15 windows and most of the host tail, but its 16 x n records (2.7 GB) no longer sit in the 256 MB Infinity Cache
        # the way the 168 MB table does, and the gathers of the accumulation kernel pay for it: 2.49 vs 2.25 ms per MSM.
        pre = os.environ.get("MSM377_BENCH_PRECOMPUTE", "0")
        if pre in ("1", "16", "20"):
            bits = 20 if pre == "20" else 16
            eng.set_precompute_window(bits)
            eng.set_bases_precomputed_device(d_points.data_ptr(), n)  # [2^(c w)] P_i for all windows: one reduction per MSM
            out["table"] = "precomputed window multiples, %d-bit windows: %d x n affine records" % (bits, 13 if bits == 20 else 16)
        else:
            eng.set_bases_device(d_points.data_ptr(), n)
            out["table"] = "n affine records"
        out["set_bases_ms"] = round((time.perf_counter() - t_set) * 1e3, 2)
        d_scalars = fixed64_scalar_sets(torch, scalars_host, n, batch)
        torch.cuda.synchronize()
        step = lambda: eng.msm_fixed_base_batch_device(d_scalars.data_ptr(), n, batch)  # noqa: E731
        per_step = batch
        verify_fixed = not args.no_cpu_baseline
        out.update({"metric": "ms per 2^%d fixed-base BLS12-377 G1 MSM (batch of 64, resident bases)" % args.log_n,
                    "dtype": "u32 (29-bit limbs, 64-bit accumulate)",
                    "config": {"workload": "64 x 2^%d fixed-base G1 MSMs over one HBM-resident converted base set, host tail overlapped" % args.log_n}})
        alg = 32 * n + NUM_WINDOWS * 96 * n + 2 * NUM_WINDOWS * NUM_BUCKETS * 144 + 96
    for _ in range(args.warmup):
        res = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / max(1, args.steps) / per_step
    out.update({"value": round(ms, 4), "ms_per_step": round(ms * per_step, 4), "whole_job_hbm_GBps": round(alg / (ms * 1e-3) / 1e9, 2)})
    if args.workload == "fixed64" and verify_fixed:
        # Every one of the 64 results against a closed form (after the timed region; the oracle is the checker).
        exp = fixed64_expected(n, batch, scalars_host)
        for b in range(batch):
            if exp[b] != res[b]:
                raise SystemExit("PARITY FAILURE: fixed-base MSM %d of the batch differs from the closed form" % b)
        out["verified"] = "all %d results bit-exact against the closed form [sum_i k_i a_i]G (oracle scalar multiplication)" % batch
    if args.workload == "ed" and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import util

        oracle = util.load_oracle()
        ph = d_points.cpu().numpy().tobytes()
        t1 = time.perf_counter()
        cpu = util.oracle_ed_msm(oracle, ph, ks)
        cpu_ms = (time.perf_counter() - t1) * 1e3
        if cpu != res:
            raise SystemExit("PARITY FAILURE: HIP Edwards result differs from the CPU oracle")
        eng.ed_msm(ph, ks)  # host-buffer entry point (chunked upload): first call allocates the staging buffer
        samples = []
        for _ in range(5):
            t1 = time.perf_counter()
            r2 = eng.ed_msm(ph, ks)
            samples.append((time.perf_counter() - t1) * 1e3)
            assert r2 == res
        out["ms_incl_h2d"] = round(sorted(samples)[2], 3)
        out["cpu_baseline"] = {"value": round(cpu_ms, 1), "unit": "ms per 2^%d MSM" % args.log_n, "cores": int(oracle.oracle_omp_threads()),
                               "kind": "port", "sample": "the full workload, 1 run, same inputs; bit-exact with the GPU's"}
    print(json.dumps(out), flush=True)
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--setup-msms", type=int, default=30,
                    help="untimed MSMs before the --warmup steps (the clocks take ~0.1 s of load to settle after idle); 0 reproduces the round-1 protocol")
    # a step is ~3 ms: the defaults take a quarter of a second.  The first ~10 MSMs after idle run ~5 % slower (the
    # accumulation kernel 1.82 instead of 1.70 ms while the clocks settle), hence the longer warm-up.
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--log-n", type=int, default=LOG_N, help="log2 of the point count (default 20: the metric's workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument(
        "--workload",
        choices=["g1", "ed", "fixed64"],
        default="g1",
        help="g1 = the metric's workload (BASELINE.json configs[1], default); ed = configs[2] Twisted-Edwards MSM; "
        "fixed64 = configs[4] 64 fixed-base MSMs over one HBM-resident base set (informational lines, single GPU)",
    )
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))

    import torch

    import webgpu_msm_bls12_377_amd as msm
    from webgpu_msm_bls12_377_amd.host.sharding import ShardedMsm, choose_partition, points_for_rank, windows_for_rank

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # Rehearsal knobs for a one-GPU box (never set by the driver): MSM377_BENCH_SINGLE_DEVICE=1 puts every
    # rank on device 0 and MSM377_BENCH_BACKEND=gloo moves the gather to CPU tensors -- RCCL refuses two ranks
    # on one device.  The default is one rank per GPU over RCCL ("nccl").
    if os.environ.get("MSM377_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("MSM377_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    # MSM377_BENCH_FORCE_SHARDED=1 (rehearsal on a one-GPU box, never set by the driver): run the multi-rank code path --
    # window records left in HBM, RCCL all-gather, host combine -- with ONE rank that owns all 16 windows.
    force_sharded = world == 1 and os.environ.get("MSM377_BENCH_FORCE_SHARDED") == "1"
    saved_stdout = None
    if world > 1 or force_sharded:
        import torch.distributed as dist

        # RCCL prints a version banner on STDOUT when its first communicator comes up; the contract is ONE JSON line
        # there, so stdout points at stderr until the warm-up is over.
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    n = 1 << args.log_n
    if args.workload != "g1":
        if world != 1:
            raise SystemExit("--workload %s is a single-GPU informational run" % args.workload)
        return side_workload(args, torch, msm, n)
    eng = msm.MsmEngine(n, device=local_rank)
    d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
    eng.generate_bases_device(0x377, n, d_points.data_ptr())
    scalars_host = seeded_scalars(0x5CA1A5, n)
    d_scalars = torch.frombuffer(bytearray(scalars_host), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    pp, sp = d_points.data_ptr(), d_scalars.data_ptr()
    dev = torch.device("cuda", local_rank)
    xdev = dev if backend == "nccl" else torch.device("cpu")  # where the exchange buffers live
    # N > 1: the 16 windows are sharded, each rank in the twisted Edwards form (the engine's default).
    # MSM377_BENCH_GLV=1 shards the 8 windows of the GLV front end instead (Weierstrass form; opt-in because it is
    # only valid for prime-order subgroup points, which the synthetic inputs are); scalars outside the GLV range
    # make every rank fall back to the plain 16 windows (all ranks see the same scalars, so they agree).
    use_glv = 1 < world <= 8 and os.environ.get("MSM377_BENCH_GLV", "0") == "1"
    # the engine's defaults (include/msm377.h): twisted Edwards form, 16 plain windows; MSM377_G1_FORM=0 MSM377_GLV=1
    # selects the Weierstrass XYZZ path behind the GLV front end
    te_single = world == 1 and os.environ.get("MSM377_G1_FORM", "1") != "0"
    glv_single = world == 1 and not te_single and os.environ.get("MSM377_GLV", "0") == "1"
    # How the MSM is split over the ranks: "points" (every rank a complete MSM of its n / G points, one all-gather of
    # 96-byte results) or "windows" (BASELINE.json config 4's window subtasks: one all-gather of window records, host
    # combine).  Default from the measured per-rank times (host/sharding.py choose_partition);
    # MSM377_BENCH_PARTITION forces one.
    partition = os.environ.get("MSM377_BENCH_PARTITION") or choose_partition(n, world)
    if partition not in ("points", "windows") or use_glv:
        partition = "windows"
    sharder = ShardedMsm(rank, world, device=xdev, force_collective=force_sharded)
    sharder_glv = ShardedMsm(rank, world, device=xdev, num_windows=8) if use_glv else None
    resident = (world > 1 or force_sharded) and backend == "nccl" and not use_glv  # records stay in HBM until the all-gather

    def rerun_weierstrass(b, c, out_ptr):  # Edwards records that add up to an exceptional case: this rank's windows again in form 0
        eng.set_g1_form("weierstrass")
        try:
            eng.window_partials_resident(pp, sp, n, b, c, out_ptr)
        finally:
            eng.set_g1_form("edwards")

    def step():
        if world == 1 and not force_sharded:
            return eng.msm_device(pp, sp, n)
        if partition == "points":
            return sharder.run_points(lambda first, count: eng.msm_device(pp + 96 * first, sp + 32 * first, count), n)
        if resident:
            return sharder.run_resident(lambda b, c, out_ptr: eng.window_partials_resident(pp, sp, n, b, c, out_ptr), eng.combine_partials, rerun_weierstrass)
        if use_glv:
            try:
                return sharder_glv.run(lambda b, c: eng.glv_window_partials_device(pp, sp, n, b, c))
            except msm.MsmError as e:
                if e.code != -6:
                    raise
        return sharder.run(lambda b, c: eng.window_partials_device(pp, sp, n, b, c))

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Setup, before the W warm-up steps the contract asks for: the GPU's clocks take ~30 MSMs (0.1 s) of load to settle
    # after idle (accumulation kernel 1.82 -> 1.70 ms), so the first-call allocations and that ramp are not left to the
    # warm-up count the caller happens to pass.  Reported as config.setup_msms.
    setup_msms = max(0, args.setup_msms)
    result = None
    for _ in range(setup_msms):
        result = step()
    for _ in range(args.warmup):
        result = step()
    if saved_stdout is not None:
        if args.warmup == 0:
            fence()  # brings the communicator up (and its banner out) before stdout returns
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    # Timed region: HIP events around the accumulation kernel only (the roofline's launch duration, on the engine's own
    # stream).  Events around EVERY stage put a few microseconds of GPU idle time between the launches they separate
    # (~50 us per MSM), so the stage breakdown comes from a separate pass after the timed region.
    eng.set_timing(2)
    acc_sum = 0.0
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step()
        acc_sum += eng.stage_ms()["accumulate_kernel"]
    fence()
    elapsed = time.perf_counter() - t0
    eng.set_timing(True)
    stage_sum = {}
    stage_steps = max(1, min(args.steps, 5))
    for _ in range(stage_steps):
        step()
        for k, v in eng.stage_ms().items():
            stage_sum[k] = stage_sum.get(k, 0.0) + v
    fence()
    eng.set_timing(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / max(1, args.steps)
    stages = {k: v / stage_steps for k, v in stage_sum.items()}
    stages["accumulate_kernel"] = acc_sum / max(1, args.steps)  # the timed region's own average

    out = None
    if rank == 0:
        glv_path = glv_single or (world > 1 and use_glv)
        te_path = te_single or (world > 1 and not use_glv and os.environ.get("MSM377_G1_FORM", "1") != "0")
        whole_bytes, acc_bytes = algorithmic_bytes(n, glv_path)
        nwin = 8 if (world > 1 and use_glv) else NUM_WINDOWS
        _, my_windows = windows_for_rank(rank, world, nwin)
        sharded = world > 1 or force_sharded
        if sharded and partition == "points":  # this rank's launch: all windows over its slice of the points
            my_windows = nwin
            share = points_for_rank(rank, world, n)[1] / n
            _, acc_slice = algorithmic_bytes(points_for_rank(rank, world, n)[1], glv_path)
        else:
            share = my_windows / nwin
            acc_slice = acc_bytes * share
        acc_ms = stages.get("accumulate_kernel", 0.0)  # HIP events around the k_accumulate launch alone
        acc_bytes_launch = acc_slice  # one launch covers this rank's share of the windows (or of the points)
        achieved = acc_bytes_launch / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        traffic, traffic_source = pmc_traffic_bytes(args.log_n) if world == 1 else (None, None)
        # int32-mad roof: one launch adds every non-zero digit's point once (16 n additions less the 2^-16 share of
        # zero digits: negligible, not subtracted) at `products` field products of 337 multiply-adds each
        products = eng.accumulate_products()
        lane_mads = NUM_WINDOWS * n * share * products * MADS_PER_FIELD_PRODUCT
        mad_rate = lane_mads / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        # all VALU issue slots of the loop, each class at its own measured rate: the roof the kernel actually sits under
        isa_tag, valu_per_add, mads_per_add = isa_loop_mix(products)
        issue_frac = None
        if valu_per_add and acc_ms > 0:
            adds = NUM_WINDOWS * n * share
            issue_ms = adds * (mads_per_add / MAD_PEAK_GLANEOPS + (valu_per_add - mads_per_add) / ALU_PEAK_GLANEOPS) / 1e9 * 1e3
            issue_frac = round(issue_ms / acc_ms, 4)
        out = {
            "metric": "ms per 2^%d BLS12-377 G1 MSM" % args.log_n,
            "value": round(ms_per_step, 4),
            "unit": "ms",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": False,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32 (29-bit limbs, 64-bit accumulate)",
            "data": "synthetic",
            "config": {
                "workload": "2^%d BLS12-377 G1 (short Weierstrass) MSM, 16-bit signed windows (the top three: 15 bits, unsigned), inputs resident in HBM" % args.log_n,
                "front_end": "GLV: 8 windows over the 2n points {P_i, phi(P_i)}" if glv_path else "plain: 16 windows over n points",
                "coordinates": "twisted Edwards form of G1, extended coordinates (csrc/te377.hpp)" if te_path else "short Weierstrass, XYZZ",
                "points": "P_i=[a_i]G, a_i=SplitMix64(0x377)",
                "scalars": "uniform < r, SplitMix64(0x5ca1a5)",
                "setup_msms": setup_msms,  # untimed MSMs run as part of setup, before the warm-up steps (clock ramp after idle)
                "parallelism": (("points sharded over %d GPUs (a complete MSM of n / %d points per rank), one RCCL all-gather of 96-byte results" % (world, world))
                                if partition == "points" else "%s windows sharded over %d GPUs, one RCCL all-gather" % ("8 GLV" if use_glv else "16", world)) if sharded else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_accumulate (bucket accumulation)",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": int(acc_bytes_launch),
                "kernel_ms": round(acc_ms, 4),
                # the roof that actually binds this kernel: 32x32+64-bit multiply-adds (v_mad_u64_u32), not bytes
                "secondary": {
                    "bound": "int32-mad",
                    "achieved": round(mad_rate, 1),
                    "peak": MAD_PEAK_GLANEOPS,
                    "unit": "G lane-mad/s",
                    "frac": round(mad_rate / MAD_PEAK_GLANEOPS, 4),
                    "lane_mads_per_launch": int(lane_mads),
                    "derivation": "%d windows x n additions x %d field products x %d v_mad_u64_u32 (13x13 + 14x12, csrc/field29.hpp); peak = mad64 at 8 waves/SIMD, profiles/microbench_r01.txt"
                    % (my_windows, products, MADS_PER_FIELD_PRODUCT),
                    # fraction of the kernel's time that pure VALU issue accounts for: (mads / mad rate + other VALU / add rate) per addition
                    "valu_issue_frac": issue_frac,
                    "valu_issue_derivation": ("profiles/%s/isa_mix.json: %d VALU instructions per addition in the loop, %d of them v_mad_u64_u32; rates %g / %g G lane-op/s"
                                              % (isa_tag, valu_per_add, mads_per_add, MAD_PEAK_GLANEOPS, ALU_PEAK_GLANEOPS)) if issue_frac else None,
                },
            },
            "whole_job_hbm_GBps": round(whole_bytes / (ms_per_step * 1e-3) / 1e9, 2),
            "stages_ms": {k: round(v, 4) for k, v in stages.items()},
            "stages_note": "accumulate_kernel: HIP events inside the timed region; the other stages: %d more steps after it with events around every stage" % stage_steps,
            "result_x": hex(int.from_bytes(result[:48], "little")),
        }
        if world == 1 and not args.no_cpu_baseline:
            # PCIe-inclusive figure for DESIGN.md (never `value`): host buffers in, result out
            points_host = d_points.cpu().numpy().tobytes()
            eng.reserve_host_staging()  # pinned staging allocated up front (msm377_ctx_reserve_host_staging), not inside the first call
            t1 = time.perf_counter()
            r2 = eng.msm(points_host, scalars_host)
            out["ms_incl_h2d_first_call"] = round((time.perf_counter() - t1) * 1e3, 3)
            samples = []
            for _ in range(15):  # chunked upload overlapped with the computation (msm377_g1_msm); median of 15 calls
                t1 = time.perf_counter()
                r2 = eng.msm(points_host, scalars_host)
                samples.append((time.perf_counter() - t1) * 1e3)
                assert r2 == result
            out["ms_incl_h2d"] = round(sorted(samples)[len(samples) // 2], 3)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import util  # the CPU oracle: checker + cpu_baseline leg only

            oracle = util.load_oracle()
            t1 = time.perf_counter()
            cpu = util.oracle_msm(oracle, points_host, scalars_host)
            cpu_ms = (time.perf_counter() - t1) * 1e3
            if cpu != result:
                raise SystemExit("PARITY FAILURE: HIP result differs from the CPU oracle at n=2^%d" % args.log_n)
            out["cpu_baseline"] = {
                "value": round(cpu_ms, 1),
                "unit": "ms per 2^%d MSM" % args.log_n,
                "cores": int(oracle.oracle_omp_threads()),
                "kind": "port",
                "sample": "the full 2^%d workload, 1 run, same inputs; result bit-exact with the GPU's" % args.log_n,
            }
        if world > 1 and not args.no_cpu_baseline:
            # multi-rank parity, after the timed region: the closed form [sum_i k_i a_i]G of the synthetic inputs (one oracle
            # scalar multiplication of the generator; the oracle is the checker)
            if fixed64_expected(n, 1, scalars_host)[0] != result:
                raise SystemExit("PARITY FAILURE: the %d-rank result differs from the closed form at n=2^%d" % (world, args.log_n))
            out["verified"] = "bit-exact against the closed form [sum_i k_i a_i]G (oracle scalar multiplication)"
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
