"""GPU parity tests proper: the HIP engine, called through the C ABI (ctypes mirror), against the
CPU oracle on the same seeded inputs, the committed golden vectors, and -- at the full 2^20 size --
a closed-form property.  Run on an MI355X with `pytest -m gpu`.

Bar: bit-exact (integer work).  Shapes follow the reference's own checks: end-to-end result
(src/ui/Benchmark.tsx:41-48) and the per-stage debug read-backs (src/submission/submission.ts:466-520,
613-641, 724-798).
"""
import ctypes
import random

import numpy as np
import os

import pytest

import pyref as R
import util
import webgpu_msm_bls12_377_amd as msm

pytestmark = pytest.mark.gpu


def dev(buf: bytes):
    """bytes -> uint8 tensor in HBM (torch is plumbing for device memory only)."""
    import torch

    return torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()


def seeded_inputs(oracle, n, seed):
    rnd = random.Random(seed)
    pts = util.oracle_gen_points(oracle, n, rnd.randrange(1, 1 << 200), rnd.randrange(1, 1 << 200))
    ks = R.encode_scalars(R.rand_scalars(seed, n))
    return pts, ks


def test_golden_vectors(engine, golden):
    for name, case in golden.items():
        if name.startswith("g1_"):
            assert engine.msm(case["points"], case["scalars"]) == case["expected"], name


def test_compute_msm_entry_point(golden):
    """Same call shapes as the reference's callers (Benchmark.tsx:32; full_benchmarks.ts:62)."""
    case = golden["g1_n33_random"]
    exp = R.decode_result(case["expected"])
    res = msm.compute_msm(case["points"], case["scalars"], log_result=False)
    assert res == {"x": exp[0], "y": exp[1]}
    pts = [{"x": x, "y": y, "z": 1} for x, y in R.decode_points(case["points"])]
    assert msm.compute_msm(pts, R.decode_scalars(case["scalars"]), False) == res
    u32pts = [{"x": msm.bigIntToU32Array(p["x"], 384), "y": msm.bigIntToU32Array(p["y"], 384)} for p in pts]
    u32ks = [msm.bigIntToU32Array(k) for k in R.decode_scalars(case["scalars"])]
    assert msm.compute_msm(u32pts, u32ks, False) == res
    assert msm.compute_msm(b"", b"", False) == {"x": 0, "y": 1}  # submission.ts:93-95


@pytest.mark.parametrize("windows", ["narrow", "wide"])
@pytest.mark.parametrize("n", [1, 2, 3, 5, 63, 64, 65, 255, 257, 1000, 4097, 10007])
def test_ragged_sizes_against_oracle(engine, oracle, n, windows):
    """Both window geometries at every size: 11-bit windows (the small-input path, default up to 2^15 points; the
    reference's own small-input switch is submission.ts:97) and the 16-bit main path forced onto the same inputs."""
    pts, ks = seeded_inputs(oracle, n, 100 + n)
    engine.set_narrow_max(1 << 16 if windows == "narrow" else 0)
    try:
        exp = util.oracle_msm(oracle, pts, ks)
        assert engine.msm(pts, ks) == exp
        d_p, d_s = dev(pts), dev(ks)
        assert engine.msm_device(d_p.data_ptr(), d_s.data_ptr(), n) == exp
    finally:
        engine.set_narrow_max()


def test_narrow_windows_edge_cases(engine, oracle, golden):
    """The small-input path (eleven signed 12-bit windows + eleven unsigned 11-bit ones, the last from bit 242; 2^11
    buckets each; kernels/decompose.hpp k_decompose_geom) on the inputs that stress a recode: every golden vector (edge
    scalars, cancellations, repeated points), digits at the window boundaries of this geometry and of the one before it
    (22 signed 11-bit windows + an unsigned top one, still there as MSM377_NARROW_EVEN=0), scalars of 2^253 and more
    (their top digit does not fit: the call must rerun on the 16-bit path), one repeated base point, a skewed set that
    splits rows, and the scalar-overflow error, which must fire exactly as on the main path."""
    engine.set_narrow_max(1 << 15)
    try:
        for name, case in golden.items():
            if name.startswith("g1_"):
                assert engine.msm(case["points"], case["scalars"]) == case["expected"], name
        g = R.G
        p2 = R.mul(g, 2)
        pts = [g, g, R.neg(g), p2, R.neg(p2), g, p2, g, p2]
        full = lambda d: sum((d & 0x7FF) << (11 * w) for w in range(22))  # noqa: E731
        ks = [full(0x400), full(0x400), full(0x400), full(0x3FF), full(0x3FF), full(1), full(0x7FF), 0, R.R_ORDER - 1]
        assert engine.msm(R.encode_points(pts), R.encode_scalars(ks)) == R.encode_result(R.msm_naive(pts, ks))
        # the same for the even geometry: d12 in every signed window, d11 in every unsigned one
        even = lambda d12, d11: sum((d12 & 0xFFF) << (12 * w) for w in range(11)) + sum((d11 & 0x7FF) << (132 + 11 * w) for w in range(11))  # noqa: E731
        ke = [even(0x800, 0x400), even(0x800, 0x7FF), even(0x7FF, 0x7FF), even(0x7FF, 1), even(0x801, 0), even(0xFFF, 0x3FF), even(1, 0x400), 0, even(0xFFF, 0x7FF) % R.R_ORDER]
        assert engine.msm(R.encode_points(pts), R.encode_scalars(ke)) == R.encode_result(R.msm_naive(pts, ke))
        big = [(1 << 253) - 1, 1 << 253, (1 << 254) + 5, 1194 << 242, 2047 << 242, 2048 << 242, 12345, 1, R.R_ORDER - 2]
        assert engine.msm(R.encode_points(pts), R.encode_scalars(big)) == R.encode_result(R.msm_naive(pts, big))
        n = 3000
        rep = R.encode_points([R.FIXED_BASE]) * n
        kr = R.encode_scalars(R.rand_scalars(31338, n))
        assert engine.msm(rep, kr) == R.encode_result(R.mul(R.FIXED_BASE, sum(R.decode_scalars(kr)) % R.R_ORDER))
        pl, _ = seeded_inputs(oracle, 2048, 78)
        same = R.encode_scalars([R.rand_scalars(79, 1)[0]] * 2048)
        assert engine.msm(pl, same) == util.oracle_msm(oracle, pl, same)
        case = golden["g1_n1_gen"]
        for bad in ((1 << 256) - 1, (1 << 255) - (1 << 239)):
            with pytest.raises(msm.MsmError) as e:
                engine.msm(case["points"], bad.to_bytes(32, "little"))
            assert e.value.code == -3
        def final_carry16(k):  # cuzk/utils.ts:66-109 on 16-bit windows
            carry = 0
            for w in range(16):
                carry = 1 if ((k >> (16 * w)) & 0xFFFF) + carry >= 32768 else 0
            return carry

        assert final_carry16((1 << 255) - (1 << 239)) == 1 and final_carry16((1 << 256) - 1) == 1
        for ok in (0x7FFF << 240, (0x7FFF << 240) + (0x7FFF << 224) + 12345, (1 << 255) - (1 << 239) - (1 << 224)):
            assert final_carry16(ok) == 0
            assert engine.msm(case["points"], ok.to_bytes(32, "little")) == R.encode_result(R.mul(R.G, ok)), hex(ok)
        # resident table + small n
        engine.set_bases(pl)
        assert engine.msm_fixed_base(same) == util.oracle_msm(oracle, pl, same)
        # an exceptional pair of the Edwards law meeting at the FIRST reduction level of the narrow geometry
        # (buckets 0 and 1024 of the unsigned top window, bits 242..252 in both narrow geometries), in a bucket chain,
        # and in the tail
        tp = util.t_prime()
        p = R.mul(R.G, 4711)
        q = R.add(p, tp)
        c = R.mul(R.G, 99)
        a = R.add(R.mul(c, 1 << 12), tp)  # window 0 is 12 bits wide
        from webgpu_msm_bls12_377_amd.host.engine import FB_ACCUMULATE, FB_TAIL, FB_TREE

        for pts2, ks2, where in (([p, q], [1 << 242, 1025 << 242], FB_TREE), ([p, q], [7, 7], FB_ACCUMULATE), ([c, a], [1 << 12, 1], FB_TAIL)):
            before, _ = engine.fallback_info()
            assert engine.msm(R.encode_points(pts2), R.encode_scalars(ks2)) == R.encode_result(R.msm_naive(pts2, ks2))
            count, mask = engine.fallback_info()
            assert count == before + 1 and mask & where, (ks2, mask)
    finally:
        engine.set_narrow_max()


def test_even_window_geometry_edge_cases(engine, oracle, golden):
    """Whole MSMs on the main path recode scalars into thirteen signed 16-bit windows and three UNSIGNED 15-bit ones
    (bit offsets 208, 223, 238; kernels/decompose.hpp k_decompose `even`): every golden vector forced onto that path,
    digits at the boundaries of the short windows (2^15 - 1 with and without a carry coming in, carries running through
    all three), scalars whose top digit does not fit (2^253 - 2^238 + ... and everything from 2^253 on: the call reruns
    with sixteen equal windows), the scalar-overflow error unchanged, and the same inputs through a batch."""
    engine.set_narrow_max(0)
    try:
        for name, case in golden.items():
            if name.startswith("g1_"):
                assert engine.msm(case["points"], case["scalars"]) == case["expected"], name
        n = 300
        pts, ks = seeded_inputs(oracle, n, 4242)
        pl = R.decode_points(pts)
        kl = R.decode_scalars(ks)
        s15 = 0x7FFF
        edge = [
            s15 << 208, s15 << 223, s15 << 238,                      # the largest digit of each short window
            (s15 << 208) + (0x8000 << 192),                          # a carry into window 13 wraps it to 0, carry on
            (s15 << 208) + (s15 << 223) + (0x8000 << 192),           # ... through window 14 into the top window
            (s15 << 208) + (s15 << 223) + (s15 << 238) + (0x8000 << 192),  # ... and out of it: does not fit (< 2^253)
            (s15 << 238) + (s15 << 223) + (s15 << 208) + 0x7FFF,     # largest scalar that fits with no carry at all
            (1 << 253) - 1, 1 << 253, (1 << 254) + 5,                # from 2^253 on: never fits
            (1 << 208) - 1, 1 << 208, (1 << 223) - 1, 1 << 223, (1 << 238) - 1, 1 << 238,
            0x8000 << 192, 0x7FFF << 192, R.R_ORDER - 1, R.R_ORDER - 2, 0, 1,
        ]
        fits = lambda k: k < (1 << 253) and not (k >> 238 == s15 and (k >> 223) & s15 == s15 and (k >> 208) & s15 == s15 and (k >> 207) & 1)  # noqa: E731
        assert [fits(k) for k in edge[3:10]] == [True, True, False, True, False, False, False]
        base = R.decode_result(util.oracle_msm(oracle, pts, ks))
        for i, k in enumerate(edge):  # one edge scalar at a time, so that those that fit stay on the even windows
            kk = list(kl)
            kk[i] = k
            exp = R.add(R.add(base, R.neg(R.mul(pl[i], kl[i]))), R.mul(pl[i], k))
            assert engine.msm(pts, R.encode_scalars(kk)) == R.encode_result(exp), hex(k)
        kk = list(kl)
        kk[: len(edge)] = edge
        want = R.encode_result(R.msm_naive(pl, kk))
        assert engine.msm(pts, R.encode_scalars(kk)) == want
        d_p, d_s = dev(pts), dev(R.encode_scalars(kk))
        assert engine.msm_device(d_p.data_ptr(), d_s.data_ptr(), n) == want
        engine.set_bases(pts)
        d_b = dev(ks + R.encode_scalars(kk) + ks + R.encode_scalars(kk) + ks)
        plain = util.oracle_msm(oracle, pts, ks)
        assert engine.msm_fixed_base_batch_device(d_b.data_ptr(), n, 5) == [plain, want, plain, want, plain]
        case = golden["g1_n1_gen"]
        for bad in ((1 << 256) - 1, (1 << 255) - (1 << 239)):
            with pytest.raises(msm.MsmError) as e:
                engine.msm(case["points"], bad.to_bytes(32, "little"))
            assert e.value.code == -3
        for ok in (0x7FFF << 240, (1 << 255) - (1 << 239) - (1 << 224)):  # beyond 2^253, below the error threshold
            assert engine.msm(case["points"], ok.to_bytes(32, "little")) == R.encode_result(R.mul(R.G, ok)), hex(ok)
    finally:
        engine.set_narrow_max()


def test_2_16_against_reference_sized_oracle(engine, oracle):
    """BASELINE.json configs[0] size: the oracle runs the reference's production parameters
    (16-bit windows, 256 BPR threads) on 2^16 points."""
    n = 1 << 16
    pts, ks = seeded_inputs(oracle, n, 16)
    exp = util.oracle_msm(oracle, pts, ks)
    assert engine.msm(pts, ks) == exp
    d_p, d_s = dev(pts), dev(ks)
    assert engine.msm_device(d_p.data_ptr(), d_s.data_ptr(), n) == exp


def test_stage_parity(engine, oracle):
    """Decomposition, CSR and bucket sums against the oracle's stage functions."""
    n = 5000
    pts, ks = seeded_inputs(oracle, n, 4242)
    engine.set_stage_capture(True)
    engine.set_g1_form("weierstrass")  # the stage read-backs describe the plain 16-window XYZZ path
    engine.set_glv(False)
    try:
        res = engine.msm(pts, ks)
        assert res == util.oracle_msm(oracle, pts, ks)
        chunks = np.zeros(16 * n, dtype=np.uint32)
        assert oracle.oracle_decompose_scalars_signed(ks, n, 16, chunks.ctypes.data) == 0
        chunks = chunks.reshape(16, n)
        rp_o = np.zeros(16 * 65537, dtype=np.uint32)
        vi_o = np.zeros(16 * n, dtype=np.uint32)
        oracle.oracle_cpu_transpose(chunks.ctypes.data, n, 65536, 16, rp_o.ctypes.data, vi_o.ctypes.data)
        for slot in (0, 7, 15):
            st = engine.read_stage(slot, n)
            # a4/a5: biased signed digits identical to decompose_scalars_signed
            assert np.array_equal(st["digits"].astype(np.uint32), chunks[slot])
            # a6: row t of the folded CSR holds exactly the points of digits +t and -t
            rp, vi = st["row_ptr"], st["val_idx"]
            assert rp[0] == 0 and rp[-1] == n and np.all(np.diff(rp.astype(np.int64)) >= 0)
            rpo = rp_o[slot * 65537 : (slot + 1) * 65537]
            vio = vi_o[slot * n : (slot + 1) * n]
            for key in list(range(0, 40)) + [32767, 32768] + random.Random(slot).sample(range(40, 32767), 200):
                got = sorted((int(e) & 0x7FFFFFFF, int(e) >> 31) for e in vi[rp[key] : rp[key + 1]])
                pos = [(int(i), 0) for i in vio[rpo[32768 + key] : rpo[32768 + key + 1]]] if key < 32768 else []
                negs = [(int(i), 1) for i in vio[rpo[32768 - key] : rpo[32768 - key + 1]]] if key > 0 else []
                assert got == sorted(pos + negs), (slot, key)
            # a7: bucket sums equal the WGSL semantics (bucket t-1 here = thread id t there; id 0 = digit -2^15)
            bo = ctypes.create_string_buffer(96 * 32768)
            assert oracle.oracle_g1_smvp_window(pts, ks, n, 16, slot, ctypes.addressof(bo)) == 0
            bk = st["buckets"]
            nonempty = 0
            for t in range(1, 32769):
                exp = bo.raw[96 * (t % 32768) : 96 * (t % 32768) + 96]
                words = bk[t - 1]
                if not words[26:39].any():  # ZZ = 0: identity
                    assert exp == R.encode_result(None), (slot, t)
                    continue
                nonempty += 1
                if nonempty <= 400:  # full big-integer check on the first few hundred non-empty buckets
                    assert R.encode_result(util.affine_from_xyzz_words(words)) == exp, (slot, t)
            assert nonempty > 1000
    finally:
        engine.set_stage_capture(False)
        engine.set_glv("auto")
        engine.set_g1_form("edwards")


def test_stage_parity_edwards_form(engine, oracle):
    """The DEFAULT accumulation kernel (k_accumulate<TeDev>, twisted Edwards buckets) bucket by bucket against the
    oracle's SMVP (WGSL semantics, smvp_bls12_377.template.wgsl:72-160): the read-back returns the (X, Y, T, Z)
    words, decoded here with Python integers (extended-coordinate invariant and curve equation checked on the way)."""
    n = 5000
    pts, ks = seeded_inputs(oracle, n, 4343)
    engine.set_stage_capture(True)
    default_path(engine)
    try:
        assert engine.msm(pts, ks) == util.oracle_msm(oracle, pts, ks)
        assert engine.stage_form() == 1  # twisted Edwards, no fallback happened
        for slot in (0, 9, 15):
            bk = engine.read_stage(slot, n, want=("buckets",))["buckets"]
            bo = ctypes.create_string_buffer(96 * 32768)
            assert oracle.oracle_g1_smvp_window(pts, ks, n, 16, slot, ctypes.addressof(bo)) == 0
            nonempty = 0
            for t in range(1, 32769):
                exp = bo.raw[96 * (t % 32768) : 96 * (t % 32768) + 96]
                words = bk[t - 1]
                if not words[0:13].any() and np.array_equal(words[13:26], words[39:52]):  # (0 : c : 0 : c): the identity
                    assert exp == R.encode_result(None), (slot, t)
                    continue
                nonempty += 1
                if nonempty <= 300:
                    assert R.encode_result(util.affine_from_te_ext_words(words)) == exp, (slot, t)
            assert nonempty > (1000 if slot < 15 else 500)
    finally:
        engine.set_stage_capture(False)


def test_every_check_of_the_edwards_law_fires(engine):
    """the exceptional pairs of the a = -1 law are P, P + T' with T' = (-omega, 0) (util.t_prime) -- NOT the
    2-torsion point (-1, 0) the older tests used, whose Edwards image is an ordinary point.  One case per place the
    pair can meet: a bucket chain in k_accumulate, the merge of a split row, a thread-level and a quad-level step of
    the bucket reduction, and the host tail.  Each must (a) raise its flag -- read back through
    msm377_ctx_get_fallback_info -- and (b) still return the exact sum, computed here with Python integers."""
    from webgpu_msm_bls12_377_amd.host.engine import FB_ACCUMULATE, FB_MERGE, FB_TAIL, FB_TREE

    tp = util.t_prime()
    p = R.mul(R.G, 31337)
    q = R.add(p, tp)
    assert R.on_curve(q)
    c = R.mul(R.G, 777)
    a = R.add(R.mul(c, 1 << 16), tp)
    cases = [
        # same digit, same bucket: first(P) then madd(P + T')
        ("accumulate", [p, q], [5, 5], FB_ACCUMULATE),
        # opposite signs of one digit meet in the same row as well: 65527 = -9 + 2^16, so window 0 holds P - (P + T')
        ("accumulate, opposite signs", [p, q], [9, 65527], FB_ACCUMULATE),
        # one row of 32 entries = two work items of 16 (SEG_MIN): partial sums 16 P and 16 P + T' whatever the order
        ("split-row merge", [p] * 31 + [q], [3] * 32, FB_ACCUMULATE | FB_MERGE),
        # buckets 0 and 16384 of window 0 meet at reduction level 0 (one thread per addition)
        ("tree level 0", [p, q], [1, 16385], FB_TREE),
        # buckets 0 and 16 meet at level 10 (one lane quad per addition)
        ("tree level 10", [p, q], [1, 17], FB_TREE),
        # windows 1 and 0 meet in the host's Horner chain: [2^16] C + ([2^16] C + T')
        ("host tail", [c, a], [1 << 16, 1], FB_TAIL),
    ]
    default_path(engine)
    engine.set_narrow_max(0)  # the bucket placements below are those of the 16-bit geometry
    for name, pts, ks, where in cases:
        exp = R.encode_result(R.msm_naive(pts, ks))
        pb, sb = R.encode_points(pts), R.encode_scalars(ks)
        n = len(pts)
        before, _ = engine.fallback_info()
        assert engine.msm(pb, sb) == exp, name
        count, mask = engine.fallback_info()
        assert count == before + 1 and mask & where, (name, count - before, mask)
        # device entry point, resident table (projective or affine records), batch
        d_p, d_s = dev(pb), dev(sb + sb)
        assert engine.msm_device(d_p.data_ptr(), d_s.data_ptr(), n) == exp, name
        engine.set_bases(pb)
        before, _ = engine.fallback_info()
        assert engine.msm_fixed_base(sb) == exp, name
        assert engine.fallback_info()[0] == before + 1, name
        assert engine.msm_fixed_base(sb) == exp, name  # the table stays in the form it fell back to
        assert engine.fallback_info()[0] == before + 1, name
        engine.set_bases(pb)
        assert engine.msm_fixed_base_batch_device(d_s.data_ptr(), n, 2) == [exp, exp], name
        # window shards: the shard that owns the pair reruns alone (untagged records); a pair that only meets in the
        # tail is reported by the combine, and form 0 records combine to the right point
        parts = [engine.window_partials_device(d_p.data_ptr(), d_s.data_ptr(), n, *msm.windows_for_rank(r, 4)) for r in range(4)]
        if where == FB_TAIL:
            with pytest.raises(msm.MsmError) as e:
                msm.combine_partials(b"".join(parts))
            assert e.value.code == -7
            with pytest.raises(msm.MsmError) as e:
                engine.combine_partials(b"".join(parts))
            assert e.value.code == -7
            engine.set_g1_form("weierstrass")
            try:
                parts = [engine.window_partials_device(d_p.data_ptr(), d_s.data_ptr(), n, *msm.windows_for_rank(r, 4)) for r in range(4)]
            finally:
                default_path(engine)
        assert msm.combine_partials(b"".join(parts)) == exp, name
        assert engine.combine_partials(b"".join(parts)) == exp, name
    engine.set_narrow_max()


@pytest.mark.parametrize("world", [2, 8])
def test_window_records_left_in_device_memory(engine, oracle, world):
    """The RCCL path of bench.py --gpus N: msm377_g1_window_partials_resident leaves a rank's records in HBM (what the
    all-gather reads), the same points as the host-buffer variant; the context-threaded combine gives the oracle's result."""
    import torch

    n = 3000
    pts, ks = seeded_inputs(oracle, n, 123 + world)
    d_p, d_s = dev(pts), dev(ks)
    from webgpu_msm_bls12_377_amd.host.engine import WINDOW_PARTIAL_BYTES

    gathered = torch.zeros(16 * WINDOW_PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
    off = 0
    for r in range(world):
        b, c = msm.windows_for_rank(r, world)
        engine.window_partials_resident(d_p.data_ptr(), d_s.data_ptr(), n, b, c, gathered.data_ptr() + off)
        host = engine.window_partials_device(d_p.data_ptr(), d_s.data_ptr(), n, b, c)
        mine = gathered[off : off + c * WINDOW_PARTIAL_BYTES].cpu().numpy().tobytes()
        # same points, not the same bytes: the order of additions inside a bucket is free, so two runs may leave
        # different projective representatives
        wa = np.frombuffer(mine, dtype=np.uint32).reshape(c, 16, 48)
        wb = np.frombuffer(host, dtype=np.uint32).reshape(c, 16, 48)
        for w in range(c):
            assert wa[w, 0, 11] >> 31 == 1 and wb[w, 0, 11] >> 31 == 1  # tagged: twisted Edwards records
            for k in (0, 1, 8, 15):
                assert util.affine_from_te_record_words(wa[w, k]) == util.affine_from_te_record_words(wb[w, k]), (r, w, k)
        off += c * WINDOW_PARTIAL_BYTES
    rec = gathered.cpu().numpy().tobytes()
    exp = util.oracle_msm(oracle, pts, ks)
    assert engine.combine_partials(rec) == exp == msm.combine_partials(rec)


def select_path(engine, path):
    """The three internal paths of the G1 entry points: twisted Edwards form (default), Weierstrass XYZZ behind the
    GLV front end, Weierstrass XYZZ with the plain 16 windows."""
    engine.set_g1_form("edwards" if path == "edwards" else "weierstrass")
    engine.set_glv({"edwards": "auto", "glv": True, "plain": False}[path])


def default_path(engine):
    engine.set_g1_form("edwards")
    engine.set_glv("auto")


@pytest.mark.parametrize("path", ["edwards", "glv", "plain"])
def test_glv_and_plain_front_ends(engine, oracle, golden, path):
    """f4: the twisted Edwards form, the GLV front end (k = k1 + k2 LAMBDA, 8 windows over {P_i, phi(P_i)}) and the
    plain 16-window Weierstrass path give the oracle's result on the same inputs."""
    select_path(engine, path)
    try:
        for name, case in golden.items():
            if name.startswith("g1_"):
                assert engine.msm(case["points"], case["scalars"]) == case["expected"], name
        for n in (1, 2, 77, 1000, 20000):
            pts, ks = seeded_inputs(oracle, n, 31 + n)
            assert engine.msm(pts, ks) == util.oracle_msm(oracle, pts, ks), n
    finally:
        default_path(engine)


def test_scalars_outside_the_glv_range_fall_back(engine, oracle):
    """k >= ~2^254 does not split into two 127-bit halves: the call reruns on the plain path (which accepts
    scalars up to 2^255 - 2^239) and still returns sum k_i P_i."""
    n = 300
    pts, ks = seeded_inputs(oracle, n, 909)
    ks_int = R.decode_scalars(ks)
    ks_int[17] = (1 << 254) + 123456789
    ks_int[200] = (1 << 255) - (1 << 240)
    ks2 = R.encode_scalars(ks_int)
    exp = R.encode_result(R.msm_naive(R.decode_points(pts), ks_int))
    select_path(engine, "glv")
    try:
        assert engine.msm(pts, ks2) == exp
        engine.set_bases(pts)
        assert engine.msm_fixed_base(ks2) == exp
        assert engine.msm_fixed_base(ks) == util.oracle_msm(oracle, pts, ks)
    finally:
        default_path(engine)
    assert engine.msm(pts, ks2) == exp  # Edwards form: 16 plain windows, no range limit below 2^255 - 2^239


@pytest.mark.parametrize("path", ["edwards", "glv", "plain"])
def test_bucket_boundaries_and_signs(engine, oracle, path):
    """Digits hitting 0, +-1, +2^15-1, -2^15 in every window; repeated and opposite points."""
    select_path(engine, path)
    g = R.G
    p2 = R.mul(g, 2)
    pts = [g, g, R.neg(g), p2, R.neg(p2), g, p2, g]
    full = lambda d: sum((d & 0xFFFF) << (16 * w) for w in range(15))  # noqa: E731
    ks = [full(0x8000), full(0x8000), full(0x8000), full(0x7FFF), full(0x7FFF), full(1), full(0xFFFF) % R.R_ORDER, 0]
    pb, sb = R.encode_points(pts), R.encode_scalars(ks)
    exp = R.encode_result(R.msm_naive(pts, ks))
    try:
        assert engine.msm(pb, sb) == exp == util.oracle_msm(oracle, pb, sb, "oracle_g1_msm_naive")
    finally:
        default_path(engine)


@pytest.mark.parametrize("path", ["edwards", "glv", "plain"])
def test_one_repeated_base_point(engine, oracle, path):
    """The harness's 'random inputs' mode: ONE base point repeated (src/ui/AllBenchmarks.tsx:84-88),
    so every bucket with two entries doubles."""
    select_path(engine, path)
    n = 3000
    pts = R.encode_points([R.FIXED_BASE]) * n
    ks = R.encode_scalars(R.rand_scalars(31337, n))
    total = sum(R.decode_scalars(ks)) % R.R_ORDER
    exp = R.encode_result(R.mul(R.FIXED_BASE, total))
    try:
        assert engine.msm(pts, ks) == exp
    finally:
        default_path(engine)


@pytest.mark.parametrize("path", ["edwards", "plain"])
@pytest.mark.parametrize("n", [65, 130, 200, 2048])
def test_all_same_scalar(engine, oracle, n, path):
    """Maximally skewed buckets: every point lands in the same bucket of each window, so every row is
    split into work items and merged (2, 3, 4 and 32 segments)."""
    pts, _ = seeded_inputs(oracle, n, 77)
    k = R.rand_scalars(78, 1)[0]
    ks = R.encode_scalars([k] * n)
    select_path(engine, path)
    try:
        assert engine.msm(pts, ks) == util.oracle_msm(oracle, pts, ks)
    finally:
        default_path(engine)


def test_points_outside_the_prime_order_subgroup(engine, oracle):
    """The Edwards form's addition law has exceptional pairs, all involving points of even order (csrc/te377.hpp):
    the map does not cover the 2-torsion point (-1, 0), and P, Q with P - Q of order 2 hit a point at infinity of the
    model.  Valid curve points all the same: the engine must notice (conversion / per-addition Z = 0 check) and rerun
    on the Weierstrass path -- full MSM, resident table, batch."""
    t2 = (R.P - 1, 0)
    rnd_pts = R.decode_points(seeded_inputs(oracle, 40, 4040)[0])
    shifted = [R.add(p, t2) for p in rnd_pts[:10]]
    for a in shifted:
        assert (a[1] * a[1] - a[0] ** 3 - 1) % R.P == 0
    te = util.te_params()
    x4 = (-1 - pow(te["s"], -1, R.P)) % R.P  # s (x + 1) = -1: the order-4 point over (-1, 0), where u + 1 = 0
    import gen_consts  # tools/ (put on sys.path by util.te_params)

    t4 = (x4, gen_consts._sqrt_p((x4 ** 3 + 1) % R.P))
    assert R.add(t4, t4) == t2
    cases = {
        "order-4 input": rnd_pts[:3] + [t4] + rnd_pts[3:6] + [R.neg(t4)],
        "two-torsion input": rnd_pts[:5] + [t2] + rnd_pts[5:9],
        "P and P + T2 in one bucket": [rnd_pts[0], shifted[0]] + rnd_pts[1:4],
        "cofactor points only": shifted,
    }
    for name, pts in cases.items():
        n = len(pts)
        ks = R.rand_scalars(len(name), n)
        if name.startswith("P and"):
            ks[0], ks[1] = 5, 5  # same digits, same buckets: P + (P + T2)
        exp = R.encode_result(R.msm_naive(pts, ks))
        pb, sb = R.encode_points(pts), R.encode_scalars(ks)
        assert engine.msm(pb, sb) == exp, name
        engine.set_bases(pb)
        assert engine.msm_fixed_base(sb) == exp, name
        assert engine.msm_fixed_base(sb) == exp, name  # the table stays in the form it fell back to
        d_s = dev(sb + sb)
        assert engine.msm_fixed_base_batch_device(d_s.data_ptr(), n, 2) == [exp, exp], name
        # a fresh Edwards table and a batch large enough to run as two halves on the twin context: whichever half meets
        # the exceptional case, the whole batch reruns on the Weierstrass table; the same over both precomputed tables
        for bits in (0, 16, 20):
            if bits:
                engine.set_precompute_window(bits)
                engine.set_bases_precomputed(pb)
            else:
                engine.set_bases(pb)
            d_b = dev(sb * 5)
            assert engine.msm_fixed_base_batch_device(d_b.data_ptr(), n, 5) == [exp] * 5, (name, bits)
        engine.set_precompute_window(16)
    # P - (P + T2) = -T2: opposite signs of the same digit
    pts = [rnd_pts[0], shifted[0]]
    ks = [7, R.R_ORDER - 7]
    assert engine.msm(R.encode_points(pts), R.encode_scalars(ks)) == R.encode_result(R.msm_naive(pts, ks))


def test_scalar_overflow_is_an_error(engine, golden):
    """cuzk/utils.ts:95-98 throws "final carry is 1"; here MSM377_ESCALAR."""
    case = golden["g1_n1_gen"]
    with pytest.raises(msm.MsmError) as e:
        engine.msm(case["points"], ((1 << 256) - 1).to_bytes(32, "little"))
    assert e.value.code == -3
    assert engine.msm(case["points"], case["scalars"]) == case["expected"]  # context still usable


def test_argument_errors(engine, golden):
    case = golden["g1_n33_random"]
    with pytest.raises(ValueError):
        engine.msm(case["points"][:-1], case["scalars"])
    d_p, d_s = dev(case["points"]), dev(case["scalars"])
    with pytest.raises(msm.MsmError) as e:
        engine.msm_device(d_p.data_ptr() + 4, d_s.data_ptr(), 32)
    assert e.value.code == -1
    with pytest.raises(msm.MsmError) as e:
        engine.msm_device(d_p.data_ptr(), d_s.data_ptr(), engine.max_points + 1)
    assert e.value.code == -1
    with pytest.raises(msm.MsmError) as e:
        engine.window_partials_device(d_p.data_ptr(), d_s.data_ptr(), 33, 15, 2)
    assert e.value.code == -1


@pytest.mark.parametrize("window_bits", [16, 20])
def test_fixed_base_with_precomputed_window_multiples(engine, oracle, window_bits):
    """BASELINE.json config 5 ("precomputed-point reuse"): the resident table keeps [2^(c w)] P_i for every window.
    c = 16: the sixteen bucket sets are added on the GPU, one reduction and a 16-step tail follow.  c = 20
    (msm377_ctx_set_precompute_window): 13 windows feed ONE set of 2^19 buckets -- 13 n instead of 16 n bucket
    additions.  Same results as the oracle: single calls, a prefix of the bases, batches, and inputs that make the
    Edwards form fall back."""
    n = 3000
    pts, _ = seeded_inputs(oracle, n, 55)
    engine.set_precompute_window(window_bits)
    engine.set_bases_precomputed(pts)
    for s_ in range(3):
        ks = R.encode_scalars(R.rand_scalars(700 + s_, n))
        assert engine.msm_fixed_base(ks) == util.oracle_msm(oracle, pts, ks)
    ks = R.encode_scalars(R.rand_scalars(2, 100))  # a prefix of the bases
    assert engine.msm_fixed_base(ks) == util.oracle_msm(oracle, pts[: 96 * 100], ks)
    sets = [R.encode_scalars(R.rand_scalars(40 + b, n)) for b in range(5)]
    d_s = dev(b"".join(sets))
    assert engine.msm_fixed_base_batch_device(d_s.data_ptr(), n, 5) == [util.oracle_msm(oracle, pts, k) for k in sets]
    # edge scalars: 0, 1, r - 1, digits at the window boundaries
    pl = R.decode_points(pts)[:8]
    full = lambda d: sum((d & 0xFFFF) << (16 * w) for w in range(15))  # noqa: E731
    wide = lambda d: sum((d & 0xFFFFF) << (20 * w) for w in range(12))  # noqa: E731  (20-bit digits at their boundaries)
    kl = [0, 1, R.R_ORDER - 1, full(0x8000), full(0x7FFF), full(0xFFFF) % R.R_ORDER, 2, (1 << 252) + 5]
    if window_bits == 20:
        kl[3:6] = [wide(0x80000), wide(0x7FFFF), wide(0xFFFFF) % R.R_ORDER]
    engine.set_bases_precomputed(R.encode_points(pl))
    assert engine.msm_fixed_base(R.encode_scalars(kl)) == R.encode_result(R.msm_naive(pl, kl))
    # a point outside the prime-order subgroup somewhere in the table: conversion or a doubling flags it, the table
    # is rebuilt in Weierstrass form and the result is still exact
    tp = util.t_prime()
    pl2 = pl[:5] + [R.add(pl[5], tp), (R.P - 1, 0)]
    kl2 = R.rand_scalars(9, len(pl2))
    engine.set_bases_precomputed(R.encode_points(pl2))
    before, _ = engine.fallback_info()
    assert engine.msm_fixed_base(R.encode_scalars(kl2)) == R.encode_result(R.msm_naive(pl2, kl2))
    assert engine.fallback_info()[0] == before + 1
    engine.msm(pts[:96], ks[:32])  # a plain MSM invalidates the resident set
    with pytest.raises(msm.MsmError) as e:
        engine.msm_fixed_base(ks)
    assert e.value.code == -5
    engine.set_precompute_window(16)
    with pytest.raises(msm.MsmError) as e:
        engine.set_precompute_window(21)  # 12 x 21 = 252 bits: still 13 windows for a 253-bit scalar, not offered
    assert e.value.code == -1


def test_wide_windows_on_skewed_and_larger_inputs(engine, oracle):
    """The 20-bit-window table at a size where both partition passes of its sort run over several tiles, and with
    heavily repeated scalars (rows far longer than a work item, merged from their overflow records; sort regions
    longer than the LDS path holds)."""
    n = 40000
    pts, ks = seeded_inputs(oracle, n, 77)
    engine.set_precompute_window(20)
    engine.set_bases_precomputed(pts)
    want = util.oracle_msm(oracle, pts, ks)
    assert engine.msm_fixed_base(ks) == want
    few = R.rand_scalars(5, 3)
    skew = R.encode_scalars([few[i % 3] for i in range(n)])
    assert engine.msm_fixed_base(skew) == util.oracle_msm(oracle, pts, skew)
    # The top window of the table holds 19 bits (6 x 20 + 7 x 19 = 253): scalars of 2^253 and more rerun on the
    # 16-window path over the table's window 0 -- alone, and as one element of a batch; the largest scalar whose top
    # digit still fits (2^253 exactly lands on key 2^19) stays on the wide path.
    ks_int = R.decode_scalars(ks)
    big = list(ks_int)
    big[7], big[n - 1] = (1 << 254) + 12345, (1 << 253) + 1
    big_b = R.encode_scalars(big)
    want_big = util.oracle_msm(oracle, pts, big_b)
    assert engine.msm_fixed_base(big_b) == want_big
    edge = list(ks_int)
    edge[3], edge[4] = 1 << 253, (1 << 253) - 1
    edge_b = R.encode_scalars(edge)
    want_edge = util.oracle_msm(oracle, pts, edge_b)
    assert engine.msm_fixed_base(edge_b) == want_edge
    d_s = dev(ks + big_b + edge_b)
    assert engine.msm_fixed_base_batch_device(d_s.data_ptr(), n, 3) == [want, want_big, want_edge]
    engine.set_precompute_window(16)
    engine.set_bases(pts[:96])  # drops the table


def test_fixed_base_batches(engine, oracle):
    """BASELINE.json config 5 in small: bases converted once and kept in HBM, several scalar sets."""
    n = 3000
    pts, _ = seeded_inputs(oracle, n, 5)
    engine.set_bases(pts)
    for s in range(4):
        ks = R.encode_scalars(R.rand_scalars(900 + s, n))
        assert engine.msm_fixed_base(ks) == util.oracle_msm(oracle, pts, ks)
    ks = R.encode_scalars(R.rand_scalars(1, 100))  # a prefix of the bases
    assert engine.msm_fixed_base(ks) == util.oracle_msm(oracle, pts[: 96 * 100], ks)
    engine.msm(pts[:96], ks[:32])  # a plain MSM invalidates the resident set
    with pytest.raises(msm.MsmError) as e:
        engine.msm_fixed_base(ks)
    assert e.value.code == -5


@pytest.mark.parametrize("path", ["edwards", "glv", "plain"])
def test_fixed_base_batch_pipeline(engine, oracle, path):
    """BASELINE.json config 5 shape: one resident base set, a batch of scalar sets in ONE call (the host
    tail of each MSM overlaps the next MSM's GPU work); every result equals the oracle's."""
    n, batch = 2500, 5
    pts, _ = seeded_inputs(oracle, n, 55)
    sets = [R.encode_scalars(R.rand_scalars(7000 + b, n)) for b in range(batch)]
    select_path(engine, path)
    engine.set_bases(pts)
    default_path(engine)  # the resident table remembers how it was built
    d_s = dev(b"".join(sets))
    got = engine.msm_fixed_base_batch_device(d_s.data_ptr(), n, batch)
    assert got == [util.oracle_msm(oracle, pts, s) for s in sets]
    assert engine.msm_fixed_base_batch_device(d_s.data_ptr(), n, 1) == got[:1]
    # one element of the batch outside the GLV range: that element alone reruns on the plain path
    ks_int = R.decode_scalars(sets[2])
    ks_int[5] = (1 << 254) + 99
    mixed = list(sets)
    mixed[2] = R.encode_scalars(ks_int)
    d_m = dev(b"".join(mixed))
    got_m = engine.msm_fixed_base_batch_device(d_m.data_ptr(), n, batch)
    assert got_m[:2] == got[:2] and got_m[3:] == got[3:]
    assert got_m[2] == R.encode_result(R.msm_naive(R.decode_points(pts), ks_int))
    # an out-of-range scalar anywhere in the batch is reported, and the context stays usable
    bad = bytearray(b"".join(sets))
    bad[32 * (n + 3) : 32 * (n + 4)] = b"\xff" * 32
    d_bad = dev(bytes(bad))
    with pytest.raises(msm.MsmError) as e:
        engine.msm_fixed_base_batch_device(d_bad.data_ptr(), n, batch)
    assert e.value.code == -3
    assert engine.msm_fixed_base_batch_device(d_s.data_ptr(), n, 2) == got[:2]
    # ... also in the half of the batch that runs on the twin context (elements 3 and 4 of 5), whose error and message
    # the call hands on; and an element there that has to rerun alone
    bad = bytearray(b"".join(sets))
    bad[32 * (4 * n + 7) : 32 * (4 * n + 8)] = b"\xff" * 32
    d_bad = dev(bytes(bad))
    with pytest.raises(msm.MsmError) as e:
        engine.msm_fixed_base_batch_device(d_bad.data_ptr(), n, batch)
    assert e.value.code == -3 and "scalar" in str(e.value)
    assert engine.msm_fixed_base_batch_device(d_s.data_ptr(), n, batch) == got
    late = list(sets)
    late[4] = R.encode_scalars(ks_int)
    d_l = dev(b"".join(late))
    got_l = engine.msm_fixed_base_batch_device(d_l.data_ptr(), n, batch)
    assert got_l[:4] == got[:4] and got_l[4] == got_m[2]


@pytest.mark.parametrize("world", [1, 2, 4, 8, 3])
def test_window_sharding_on_one_gpu(engine, oracle, world):
    """BASELINE.json config 4's data path with every rank's window block run on this one GPU: the
    gathered partial records combine to the oracle's result (the RCCL gather itself is covered by
    tests/test_sharding_gloo.py and bench.py --gpus N)."""
    n = 6000
    pts, ks = seeded_inputs(oracle, n, 88)
    d_p, d_s = dev(pts), dev(ks)
    parts = []
    for r in range(world):
        b, c = msm.windows_for_rank(r, world)
        parts.append(engine.window_partials_device(d_p.data_ptr(), d_s.data_ptr(), n, b, c))
    assert msm.combine_partials(b"".join(parts)) == util.oracle_msm(oracle, pts, ks)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_point_sharding_on_one_gpu(engine, oracle, world):
    """The other partitioning of a multi-GPU run (host/sharding.py run_points): every rank's slice of the points run as
    a complete MSM on this one GPU -- slices of a few hundred points (narrow windows) and of thousands (16-bit windows),
    one of them holding the point at infinity's neighbours: an all-zero scalar slice gives the identity -- and the rank
    results added by msm377_g1_add_points equal the oracle's MSM of everything."""
    from webgpu_msm_bls12_377_amd.host.sharding import points_for_rank
    from webgpu_msm_bls12_377_amd.host.engine import add_points_bytes

    n = 70000 if world == 3 else 6001
    pts, ks = seeded_inputs(oracle, n, 89)
    first, count = points_for_rank(world - 1, world, n)
    ks = ks[: 32 * first] + b"\x00" * (32 * count)  # the last rank's scalars are all zero: its result is the identity (0, 1)
    d_p, d_s = dev(pts), dev(ks)
    results = []
    for r in range(world):
        f, c = points_for_rank(r, world, n)
        results.append(engine.msm_device(d_p.data_ptr() + 96 * f, d_s.data_ptr() + 32 * f, c))
    assert results[-1] == bytes(48) + b"\x01" + bytes(47)
    assert add_points_bytes(b"".join(results)) == util.oracle_msm(oracle, pts, ks)
    # P + (-P) and P + P through the host addition
    one = results[0]
    neg = one[:48] + ((R.P - int.from_bytes(one[48:], "little")) % R.P).to_bytes(48, "little")
    assert add_points_bytes(one + neg) == bytes(48) + b"\x01" + bytes(47)
    assert add_points_bytes(one + one) == R.encode_result(R.add(R.decode_result(one), R.decode_result(one)))
    with pytest.raises(msm.MsmError):
        add_points_bytes(b"\xff" * 96)  # a coordinate that is not below p


@pytest.mark.parametrize(
    "knobs",
    [
        {"MSM377_NARROW_QUAD_ACC": "0"},  # narrow path with a thread per work item (k_accumulate), not a lane quad
        {"MSM377_ZERO_COPY_OUT": "0"},  # D2H copies + event instead of zero-copy stores and a polled sequence number
        {"MSM377_TAIL_THREADS": "1"},
        {"MSM377_TAIL_THREADS": "8", "MSM377_TAIL_SPIN_US": "0", "MSM377_TAIL_NUMA": "0"},
        {"MSM377_TAIL_THREADS": "3"},
        {"MSM377_EVEN_WINDOWS": "0", "MSM377_TWIN_BATCH": "0"},  # sixteen equal windows everywhere; batches on one context
        {"MSM377_NARROW_EVEN": "0", "MSM377_TAIL_LDS": "0"},  # small inputs: 22 signed 11-bit windows + an unsigned top one; reduction tail in global memory
        {"MSM377_NARROW_TAIL_FROM": "7", "MSM377_COOP_THREADS": "65536", "MSM377_NARROW_SEG": "32"},
        {"MSM377_NARROW_TAIL_FROM": "1", "MSM377_COOP_THREADS": "100000000"},  # every level on lane quads, everything behind level 0 in one launch
    ],
    ids=lambda k: ",".join("%s=%s" % (a.replace("MSM377_", ""), b) for a, b in k.items()),
)
def test_alternative_settings_of_the_default_path(oracle, monkeypatch, knobs):
    """The default path's build-time-equal alternatives (context knobs read from the environment) against the oracle on
    both window geometries: the result does not depend on which of them runs.  Also the stage timing levels."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    eng = msm.MsmEngine(1 << 17)
    try:
        for n in (1, 7, 300, 4099, 40000, 100003):
            pts, ks = seeded_inputs(oracle, n, 900 + n)
            exp = util.oracle_msm(oracle, pts, ks)
            d_p, d_s = dev(pts), dev(ks)
            for timing in (False, 2, True):
                eng.set_timing(timing)
                assert eng.msm_device(d_p.data_ptr(), d_s.data_ptr(), n) == exp, (n, timing)
                st = eng.stage_ms()
                if timing:
                    assert st["accumulate_kernel"] > 0.0
                    assert (st["reduce"] > 0.0) == (timing is True)
            eng.set_timing(False)
            assert eng.msm(pts, ks) == exp, n
    finally:
        eng.close()


@pytest.mark.parametrize("schedule", ["front end per chunk", "sorted once", "sorted once, 7 chunks"])
def test_host_buffers_upload_in_chunks(oracle, monkeypatch, schedule):
    """msm377_g1_msm with large host buffers uploads and accumulates in chunks of points (later chunks on top of the
    earlier ones' buckets, one reduction): forced here at small sizes through MSM377_UPLOAD_CHUNK_MIN; both schedules
    (the default: every chunk its own decompose / sort; MSM377_UPLOAD_SORT_ONCE: scalars first, one sort with the rows
    filed by chunk, every chunk its own sub-rows), both coordinate forms, ragged sizes, skew, and an exceptional point
    in a later chunk (whole call reruns on the Weierstrass path)."""
    monkeypatch.setenv("MSM377_UPLOAD_CHUNK_MIN", "100")
    if schedule != "front end per chunk":
        monkeypatch.setenv("MSM377_UPLOAD_SORT_ONCE", "1")
    if schedule.endswith("7 chunks"):
        monkeypatch.setenv("MSM377_UPLOAD_CHUNKS", "7")
        monkeypatch.setenv("MSM377_UPLOAD_SPLIT", "9")
    eng = msm.MsmEngine(1 << 17)
    eng.reserve_host_staging()
    eng.reserve_host_staging()  # idempotent
    try:
        for n in (128, 131, 1000, 4097, 70001):
            pts, ks = seeded_inputs(oracle, n, 600 + n)
            exp = util.oracle_msm(oracle, pts, ks)
            for form in ("edwards", "weierstrass"):
                eng.set_g1_form(form)
                assert eng.msm(pts, ks) == exp, (n, form)
            eng.set_g1_form("edwards")
        n = 3000
        pts, _ = seeded_inputs(oracle, n, 77)
        ks = R.encode_scalars([R.rand_scalars(5, 1)[0]] * n)  # one bucket per window, split rows in both chunks
        assert eng.msm(pts, ks) == util.oracle_msm(oracle, pts, ks)
        kb = R.decode_scalars(ks)
        kb[n - 2] = (1 << 253) + 77  # does not fit the even windows: one rerun in one piece
        assert eng.msm(pts, R.encode_scalars(kb)) == util.oracle_msm(oracle, pts, R.encode_scalars(kb))
        pl = R.decode_points(pts)[:400]
        pl[333] = (R.P - 1, 0)
        kl = R.rand_scalars(9, 400)
        assert eng.msm(R.encode_points(pl), R.encode_scalars(kl)) == R.encode_result(R.msm_naive(pl, kl))
    finally:
        eng.close()


@pytest.mark.parametrize("world", [2, 5])
def test_window_sharding_with_mixed_record_forms(engine, oracle, world):
    """Partial records carry their coordinate system (twisted Edwards by default, Weierstrass after a fallback or
    in form 0): ranks need not agree, the combine accepts any mixture -- including an all-Weierstrass gather and
    a shard whose own windows hit an exceptional case (a 2-torsion input point)."""
    n = 1500
    pts, ks = seeded_inputs(oracle, n, 99)
    d_p, d_s = dev(pts), dev(ks)
    exp = util.oracle_msm(oracle, pts, ks)
    for pattern in ("alternate", "weierstrass"):
        parts = []
        try:
            for r in range(world):
                b, c = msm.windows_for_rank(r, world)
                engine.set_g1_form("weierstrass" if (pattern == "weierstrass" or r % 2) else "edwards")
                parts.append(engine.window_partials_device(d_p.data_ptr(), d_s.data_ptr(), n, b, c))
        finally:
            default_path(engine)
        assert msm.combine_partials(b"".join(parts)) == exp, pattern
    pl = R.decode_points(pts)[:300] + [(R.P - 1, 0)]
    kl = R.decode_scalars(ks)[:301]
    pb, sb = R.encode_points(pl), R.encode_scalars(kl)
    d_p2, d_s2 = dev(pb), dev(sb)
    parts = [engine.window_partials_device(d_p2.data_ptr(), d_s2.data_ptr(), 301, *msm.windows_for_rank(r, world)) for r in range(world)]
    assert msm.combine_partials(b"".join(parts)) == R.encode_result(R.msm_naive(pl, kl))


@pytest.mark.parametrize("world", [1, 2, 4, 8, 3])
def test_glv_window_sharding_on_one_gpu(engine, oracle, world):
    """The multi-GPU data path bench.py uses for N > 1: the 8 GLV windows in rank-sized blocks."""
    n = 6000
    pts, ks = seeded_inputs(oracle, n, 89)
    d_p, d_s = dev(pts), dev(ks)
    parts = []
    for r in range(world):
        b, c = msm.windows_for_rank(r, world, 8)
        parts.append(engine.glv_window_partials_device(d_p.data_ptr(), d_s.data_ptr(), n, b, c))
    from webgpu_msm_bls12_377_amd.host.engine import combine_partials_bytes

    assert combine_partials_bytes(b"".join(parts), 8) == util.oracle_msm(oracle, pts, ks)


def test_glv_window_sharding_reports_out_of_range_scalars(engine, oracle):
    n = 100
    pts, ks = seeded_inputs(oracle, n, 90)
    ks_int = R.decode_scalars(ks)
    ks_int[3] = (1 << 254) + 7
    d_p, d_s = dev(pts), dev(R.encode_scalars(ks_int))
    for b, c in ((0, 8), (2, 3), (7, 1)):  # every shard reaches the same verdict
        with pytest.raises(msm.MsmError) as e:
            engine.glv_window_partials_device(d_p.data_ptr(), d_s.data_ptr(), n, b, c)
        assert e.value.code == -6
    parts = engine.window_partials_device(d_p.data_ptr(), d_s.data_ptr(), n, 0, 16)  # the plain fallback
    assert msm.combine_partials(parts) == R.encode_result(R.msm_naive(R.decode_points(pts), ks_int))


def test_sharded_msm_single_rank(engine, oracle):
    from webgpu_msm_bls12_377_amd.host.sharding import sharded_msm

    n = 2000
    pts, ks = seeded_inputs(oracle, n, 99)
    d_p, d_s = dev(pts), dev(ks)
    out = sharded_msm(lambda b, c: engine.window_partials_device(d_p.data_ptr(), d_s.data_ptr(), n, b, c), 0, 1)
    assert out == util.oracle_msm(oracle, pts, ks)


def test_bucket_reduction_partials_against_oracle_window_sums(engine, oracle):
    """a10 (BPR): every window's partial record -- plain bucket sum + 15 bit-plane sums -- weighs up to
    the oracle's window sum (its BPR stage 1 + stage 2 + sum of the 256 partials, submission.ts:297-308)."""
    n = 3000
    pts, ks = seeded_inputs(oracle, n, 1010)
    d_p, d_s = dev(pts), dev(ks)
    _, ws = util.oracle_msm_params(oracle, pts, ks, 16, 256, want_windows=True)
    for form in ("edwards", "weierstrass"):  # the records carry their coordinate system in a tag bit
        engine.set_g1_form(form)
        try:
            rec = engine.window_partials_device(d_p.data_ptr(), d_s.data_ptr(), n, 0, 16)
        finally:
            default_path(engine)
        words = np.frombuffer(rec, dtype=np.uint32).reshape(16, 16, 48)
        for w in (0, 1, 8, 15):
            te = bool(words[w, 0, 11] >> 31)
            assert te == (form == "edwards")
            decode = util.affine_from_te_record_words if te else util.affine_from_record_words
            g = decode(words[w, 0])
            for l in range(15):
                g = R.add(g, R.mul(decode(words[w, 1 + l]), 1 << l))
            assert R.encode_result(g) == ws[96 * w : 96 * w + 96], (form, w)


def test_heavily_skewed_scalars_at_scale(engine, oracle):
    """f2: 2^18 points whose scalars take only 3 values -- rows of ~87k entries are split into work items
    and merged (the reference assumes uniform scalars, README.md:543-547).  Closed form: sum k_i P_i with
    P_i = [a0 + i d]G."""
    n = 1 << 18
    a0, d = 0x9999999999, 0x77777
    pts = util.oracle_gen_points(oracle, n, a0, d)
    vals = R.rand_scalars(4, 3)
    ks_int = [vals[i % 3] for i in range(n)]
    total = sum(k * (a0 + i * d) for i, k in enumerate(ks_int)) % R.R_ORDER
    exp = ctypes.create_string_buffer(96)
    gen = ctypes.create_string_buffer(96)
    oracle.oracle_g1_generator(ctypes.addressof(gen))
    assert oracle.oracle_g1_scalar_mul(gen.raw, total.to_bytes(32, "little"), 32, ctypes.addressof(exp)) == 0
    assert engine.msm(pts, R.encode_scalars(ks_int)) == exp.raw


def test_generate_bases(engine):
    """Synthetic inputs P_i = [a_i]G, a_i = SplitMix64(seed)."""
    import torch

    n, seed = 40, 0x377
    out = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
    engine.generate_bases_device(seed, n, out.data_ptr())
    got = R.decode_points(out.cpu().numpy().tobytes())
    g = R.splitmix64(seed)
    assert got == [R.mul(R.G, next(g)) for _ in range(n)]


def test_full_size_2_20_closed_form(engine, oracle):
    """BASELINE.json configs[1] size.  P_i = [a0 + i d]G, so sum k_i P_i = [sum k_i (a0 + i d) mod r]G:
    one oracle scalar multiplication checks the whole 2^20-point MSM."""
    n = 1 << 20
    a0, d = 0x1234567890ABCDEF1234567890ABCDEF, 0xFEDCBA0987654321FEDCBA
    pts = util.oracle_gen_points(oracle, n, a0, d)
    ks_int = R.rand_scalars(0x5CA1A5, n)
    ks = R.encode_scalars(ks_int)
    total = sum(k * (a0 + i * d) for i, k in enumerate(ks_int)) % R.R_ORDER
    exp = ctypes.create_string_buffer(96)
    gen = ctypes.create_string_buffer(96)
    oracle.oracle_g1_generator(ctypes.addressof(gen))
    assert oracle.oracle_g1_scalar_mul(gen.raw, total.to_bytes(32, "little"), 32, ctypes.addressof(exp)) == 0
    d_p, d_s = dev(pts), dev(ks)
    engine.set_timing(True)
    try:
        got = engine.msm_device(d_p.data_ptr(), d_s.data_ptr(), n)
        print("stage ms:", engine.stage_ms())
    finally:
        engine.set_timing(False)
    assert got == exp.raw
    # linearity on the same resident inputs: MSM(P, 2k) == 2 MSM(P, k) via the fixed-base path
    engine.set_bases_device(d_p.data_ptr(), n)
    ks2 = R.encode_scalars([(2 * k) % R.R_ORDER for k in ks_int])
    exp2 = ctypes.create_string_buffer(96)
    assert oracle.oracle_g1_scalar_mul(gen.raw, ((2 * total) % R.R_ORDER).to_bytes(32, "little"), 32, ctypes.addressof(exp2)) == 0
    assert engine.msm_fixed_base(ks2) == exp2.raw


def test_2_22_closed_form(oracle):
    """BASELINE.json configs[3] size on one GPU (the 8-GPU run shards the same problem by windows)."""
    import webgpu_msm_bls12_377_amd as msm
    import bench

    n = 1 << 22
    a0, d = 0xABCDEF0123456789ABCDEF, 0x1357924680ACE
    pts = util.oracle_gen_points(oracle, n, a0, d)
    ks = bench.seeded_scalars(0x5CA1A5 + 22, n)
    ks_int = R.decode_scalars(ks)
    total = sum(k * (a0 + i * d) for i, k in enumerate(ks_int)) % R.R_ORDER
    exp = ctypes.create_string_buffer(96)
    gen = ctypes.create_string_buffer(96)
    oracle.oracle_g1_generator(ctypes.addressof(gen))
    assert oracle.oracle_g1_scalar_mul(gen.raw, total.to_bytes(32, "little"), 32, ctypes.addressof(exp)) == 0
    with msm.MsmEngine(n) as eng:
        d_p, d_s = dev(pts), dev(ks)
        eng.set_timing(True)
        got = eng.msm_device(d_p.data_ptr(), d_s.data_ptr(), n)
        print("2^22 stage ms:", eng.stage_ms())
        assert got == exp.raw
        # the same problem as 8 window shards
        parts = [eng.window_partials_device(d_p.data_ptr(), d_s.data_ptr(), n, *msm.windows_for_rank(r, 8)) for r in range(8)]
        assert msm.combine_partials(b"".join(parts)) == exp.raw


@pytest.mark.parametrize("table", ["plain", "wide"])
def test_config5_full_batch_of_64_at_2_20(engine, table):
    """BASELINE.json configs[4] at its exact shape: 64 fixed-base MSMs of 2^20 scalars each over one resident base set
    (the plain affine table, and the 20-bit-window precomputed table), every result against its closed form
    [sum_i k_i a_i]G -- P_i = [a_i]G are the bench's synthetic bases, set b is the seeded scalar set rotated by b
    (bench.fixed64_expected: one oracle scalar multiplication of the generator per MSM)."""
    import torch

    import bench

    n, batch = 1 << 20, 64
    scalars_host = bench.seeded_scalars(0x5CA1A5, n)
    d_points = torch.empty(96 * n, dtype=torch.uint8, device="cuda")
    engine.generate_bases_device(0x377, n, d_points.data_ptr())
    if table == "wide":
        engine.set_precompute_window(20)
        engine.set_bases_precomputed_device(d_points.data_ptr(), n)
    else:
        engine.set_bases_device(d_points.data_ptr(), n)
    d_scalars = bench.fixed64_scalar_sets(torch, scalars_host, n, batch)
    torch.cuda.synchronize()
    got = engine.msm_fixed_base_batch_device(d_scalars.data_ptr(), n, batch)
    assert got == bench.fixed64_expected(n, batch, scalars_host)
    engine.set_precompute_window(16)
    engine.set_bases_device(d_points.data_ptr(), 1)  # drops the 2.2 GB table
