"""Multi-GPU path on CPU: world_size-2 (and 3) gloo runs of the window sharding + all-gather +
host combine.  The per-rank window work is done by the ORACLE here (no GPU in
this container); on the GPU box the same sharded_msm() is driven by MsmEngine.window_partials_device
(tests/test_g1_parity_gpu.py, bench.py).  CPU only."""
import ctypes
import os
import random
import socket
import struct
import sys

import pytest
import torch.multiprocessing as mp

import util
import webgpu_msm_bls12_377_amd as msm


def test_windows_for_rank_covers_all_windows():
    for world in range(1, 20):
        seen = []
        for r in range(world):
            b, c = msm.windows_for_rank(r, world)
            seen += list(range(b, b + c))
        assert seen == list(range(16)), world
    assert msm.windows_for_rank(0, 1) == (0, 16)
    assert msm.windows_for_rank(7, 8) == (14, 2)
    with pytest.raises(ValueError):
        msm.windows_for_rank(3, 3)


def _worker(rank, world, port, case_name, q):
    import pyref as R
    import torch.distributed as dist
    from webgpu_msm_bls12_377_amd.host.sharding import sharded_msm

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oracle = util.load_oracle()
        case = util.load_golden()[case_name]
        _, ws = util.oracle_msm_params(oracle, case["points"], case["scalars"], 16, 256, want_windows=True)

        def partials_fn(begin, count):
            return b"".join(util.partial_record_from_window_sum(R.decode_result(ws[96 * w : 96 * w + 96])) for w in range(begin, begin + count))

        out = sharded_msm(partials_fn, rank, world, device="cpu")
        from webgpu_msm_bls12_377_amd.host.sharding import ShardedMsm

        sh = ShardedMsm(rank, world, device="cpu")
        out2 = sh.run(partials_fn)
        out3 = sh.run(partials_fn)  # buffers are reusable

        # the RCCL-path shape (records written straight into the exchange buffer, all-gather, one read-back, combine)
        def write_records(begin, count, d_out):
            data = partials_fn(begin, count)
            ctypes.memmove(d_out, data, len(data))

        out4 = sh.run_resident(write_records)

        # with a combine callback (MsmEngine.combine_partials on the GPU box): it gets the 16 records re-packed without the
        # per-rank status words
        seen = []

        def combine(parts):
            seen.append(type(parts).__name__)
            return msm.combine_partials(parts)

        out5 = sh.run_resident(write_records, combine)
        ok_kind = seen == ["bytes"]

        # POINT sharding: every rank a complete MSM of its slice (the oracle stands in for MsmEngine.msm_device), one
        # all-gather of 96-byte results, msm377_g1_add_points on every rank
        n = len(case["scalars"]) // 32
        from webgpu_msm_bls12_377_amd.host.sharding import points_for_rank

        slices = [points_for_rank(r, world, n) for r in range(world)]
        covered = sum(c for _, c in slices) == n and all(slices[r][0] + slices[r][1] == slices[r + 1][0] for r in range(world - 1))

        def msm_fn(first, count):
            return util.oracle_msm(oracle, case["points"][96 * first : 96 * (first + count)], case["scalars"][32 * first : 32 * (first + count)])

        out6 = sh.run_points(msm_fn, n)

        # a rank whose local work fails still enters the collective, and then EVERY rank raises (no rank is left waiting
        # for the collective's timeout)
        def failing(first, count):
            if rank == world - 1:
                raise msm.MsmError(-2, "injected")
            return msm_fn(first, count)

        def failing_records(begin, count, d_out):
            if rank == 0:
                raise msm.MsmError(-4, "injected")
            write_records(begin, count, d_out)

        agreed = True
        for call in (lambda: sh.run_points(failing, n), lambda: sh.run_resident(failing_records), lambda: sh.run(lambda b, c: failing(0, 0) and partials_fn(b, c))):
            try:
                call()
                agreed = False
            except msm.MsmError as e:
                agreed = agreed and e.code in (-2, -4)
        out7 = sh.run_points(msm_fn, n)  # and the exchange buffers are still usable afterwards
        q.put((rank, out == case["expected"] and out2 == out and out3 == out and out4 == out and out5 == out and ok_kind and covered and out6 == out
               and agreed and out7 == out))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,case", [(2, "g1_n33_random"), (3, "g1_n20_edge_scalars")])
def test_sharded_msm_gloo(world, case):
    util.load_oracle()  # build once before forking
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


def test_single_rank_needs_no_process_group(oracle, golden):
    import pyref as R
    from webgpu_msm_bls12_377_amd.host.sharding import sharded_msm

    case = golden["g1_n16_cuzk_test"]
    _, ws = util.oracle_msm_params(oracle, case["points"], case["scalars"], 16, 256, want_windows=True)
    out = sharded_msm(lambda b, c: b"".join(util.partial_record_from_window_sum(R.decode_result(ws[96 * w : 96 * w + 96])) for w in range(b, b + c)), 0, 1)
    assert out == case["expected"]


def _exceptional_worker(rank, world, port, q):
    """Edwards records that add up to an exceptional case of their law (tests/test_host_tail.py): every rank's combine
    reports it, every rank recomputes its windows in Weierstrass form, the second exchange succeeds."""
    import pyref as R
    import torch.distributed as dist
    from webgpu_msm_bls12_377_amd.host.sharding import ShardedMsm

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rnd = random.Random(5)
        c = R.mul(R.G, 4242)
        a = R.add(R.mul(c, 1 << 16), util.t_prime())
        win = [[None] * 16 for _ in range(16)]
        win[0][0], win[1][0] = a, c
        calls = []

        def pack(words):
            return struct.pack("<%dI" % len(words), *words)

        def edwards(begin, count, d_out):
            calls.append("edwards")
            data = b"".join(pack(sum((util.te_record_point_words(p, rnd.randrange(1, R.P), tag=(i == 0)) for i, p in enumerate(win[w])), [])) for w in range(begin, begin + count))
            ctypes.memmove(d_out, data, len(data))

        def weierstrass(begin, count, d_out):
            calls.append("weierstrass")
            data = b"".join(pack(sum((util.record_point_words(p, rnd.randrange(1, R.P)) for p in win[w]), [])) for w in range(begin, begin + count))
            ctypes.memmove(d_out, data, len(data))

        sh = ShardedMsm(rank, world, device="cpu")
        out = sh.run_resident(edwards, rerun_weierstrass=weierstrass)
        ok = out == R.encode_result(R.add(a, R.mul(c, 1 << 16))) and calls == ["edwards", "weierstrass"]
        try:
            sh.run_resident(edwards)  # no way to recompute: the error surfaces
            ok = False
        except msm.MsmError as e:
            ok = ok and e.code == -7
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_exceptional_records_are_recomputed_on_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_exceptional_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


def test_add_points_host_function():
    """msm377_g1_add_points (the last step of a point-sharded run), host-only: sums of affine wire points against Python
    big integers, the identity as the wire format writes it, P + P, P + (-P), and a coordinate that is not below p."""
    import pyref as R
    from webgpu_msm_bls12_377_amd.host.engine import add_points_bytes
    from webgpu_msm_bls12_377_amd.host.sharding import points_for_rank

    rnd = random.Random(11)
    pts = [R.mul(R.G, rnd.randrange(1, R.R_ORDER)) for _ in range(9)]
    ident = bytes(48) + b"\x01" + bytes(47)
    acc = None
    for p in pts:
        acc = p if acc is None else R.add(acc, p)
    wire = b"".join(R.encode_result(p) for p in pts)
    assert add_points_bytes(wire) == R.encode_result(acc)
    assert add_points_bytes(ident + wire[:96] + ident + wire[96:] + ident) == R.encode_result(acc)
    assert add_points_bytes(b"") == ident and add_points_bytes(ident * 3) == ident
    p0 = wire[:96]
    neg = p0[:48] + (R.P - int.from_bytes(p0[48:], "little")).to_bytes(48, "little")
    assert add_points_bytes(p0 + neg) == ident
    assert add_points_bytes(p0 + p0) == R.encode_result(R.add(pts[0], pts[0]))
    assert add_points_bytes(p0 + p0 + neg + neg) == ident
    with pytest.raises(msm.MsmError) as e:
        add_points_bytes(p0 + R.P.to_bytes(48, "little") + bytes(48))
    assert e.value.code == -1
    for world in range(1, 12):  # the slices tile [0, n) for every world size, including more ranks than points
        for n in (0, 1, 7, 100, 1 << 20):
            spans = [points_for_rank(r, world, n) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            assert all(spans[r][0] + spans[r][1] == spans[r + 1][0] for r in range(world - 1))
