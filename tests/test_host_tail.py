"""Host-only entry points of the C ABI (no GPU needed): msm377_g1_combine_partials and
msm377_g1_xyzz_to_affine -- the replacement of the reference's CPU tail
(src/submission/submission.ts:290-321).  CPU only."""
import random
import struct

import pytest

import pyref as R
import util
import webgpu_msm_bls12_377_amd as msm
from webgpu_msm_bls12_377_amd.host import engine as E


def record(points16, rnd=None):
    words = []
    for pt in points16:
        words += util.record_point_words(pt, rnd.randrange(1, R.P) if rnd else 1)
    return struct.pack("<%dI" % len(words), *words)


def test_xyzz_to_affine():
    rnd = random.Random(2)
    for _ in range(10):
        pt = R.mul(R.G, rnd.randrange(1, R.R_ORDER))
        z = rnd.randrange(1, R.P)
        zz, zzz = z * z % R.P, z * z * z % R.P
        words = util.to_limbs29_mont(pt[0] * zz % R.P) + util.to_limbs29_mont(pt[1] * zzz % R.P) + util.to_limbs29_mont(zz) + util.to_limbs29_mont(zzz)
        assert E.xyzz_to_affine(words) == R.encode_result(pt)
    assert E.xyzz_to_affine(util.xyzz_words_from_affine(None)) == R.encode_result(None)


def test_combine_matches_definition():
    """result = sum_w 2^(16 w) * (Sum_w + sum_l 2^l Plane_{w,l})."""
    rnd = random.Random(3)
    base = [R.mul(R.G, rnd.randrange(1, R.R_ORDER)) for _ in range(9)]
    recs, expect = [], None
    for w in range(16):
        pts16 = [rnd.choice(base + [None]) for _ in range(16)]
        recs.append(record(pts16, rnd))
        g = pts16[0]
        for l in range(15):
            g = R.add(g, R.mul(pts16[1 + l], 1 << l))
        expect = R.add(expect, R.mul(g, 1 << (16 * w)))
    assert msm.combine_partials(b"".join(recs)) == R.encode_result(expect)


def test_combine_against_oracle_window_sums(oracle, golden):
    """Window sums from the oracle (submission.ts:297-308) fed through the product's host combine
    reproduce the oracle's Horner result (submission.ts:310-318)."""
    for name in ("g1_n33_random", "g1_n20_edge_scalars", "g1_n2_cancel"):
        case = golden[name]
        res, ws = util.oracle_msm_params(oracle, case["points"], case["scalars"], 16, 256, want_windows=True)
        recs = b"".join(util.partial_record_from_window_sum(R.decode_result(ws[96 * w : 96 * w + 96])) for w in range(16))
        assert msm.combine_partials(recs) == res == case["expected"]


def test_combine_rejects_wrong_length():
    with pytest.raises(ValueError):
        msm.combine_partials(b"\0" * 100)


def te_record(pts16, rnd):
    import struct

    words = []
    for i, p in enumerate(pts16):
        words += util.te_record_point_words(p, rnd.randrange(1, R.P), tag=(i == 0))
    return struct.pack("<%dI" % len(words), *words)


def test_combine_accepts_mixed_record_forms_and_folded_blocks():
    """Window records carry their coordinate system (csrc/fp64_host.hpp TE_RECORD_TAG): all Edwards, all Weierstrass
    and any mixture combine to the same point; a rank may first fold its consecutive windows
    (msm377_g1_fold_window_partials) -- the exchange then carries one real point per rank."""
    from webgpu_msm_bls12_377_amd.host.engine import WINDOW_PARTIAL_BYTES, fold_partials_bytes

    rnd = random.Random(11)
    base = [R.mul(R.G, rnd.randrange(1, R.R_ORDER)) for _ in range(7)] + [(R.P - 1, 0)]
    win_pts, expect = [], None
    for w in range(16):
        pts16 = [rnd.choice(base + [None, None]) for _ in range(16)]
        win_pts.append(pts16)
        g = pts16[0]
        for l in range(15):
            g = R.add(g, R.mul(pts16[1 + l], 1 << l))
        expect = R.add(expect, R.mul(g, 1 << (16 * w)))
    exp = R.encode_result(expect)
    for pattern in ("edwards", "weierstrass", "alternate", "blocks"):
        recs = []
        for w in range(16):
            te = {"edwards": True, "weierstrass": False, "alternate": w % 2 == 0, "blocks": (w // 3) % 2 == 0}[pattern]
            recs.append(te_record(win_pts[w], rnd) if te else record(win_pts[w], rnd))
        assert msm.combine_partials(b"".join(recs)) == exp, pattern
        for world in (2, 3, 8, 16):  # every rank folds its own block before the exchange
            folded = []
            for r in range(world):
                b, c = msm.windows_for_rank(r, world)
                mine = fold_partials_bytes(b"".join(recs[b : b + c]))
                assert len(mine) == c * WINDOW_PARTIAL_BYTES
                folded.append(mine)
            assert msm.combine_partials(b"".join(folded)) == exp, (pattern, world)


def test_edwards_tail_reports_exceptional_cases():
    """ADVICE r01 (fp64_host.hpp): the a = -1 law is not complete on this curve, and the host tail used to add without
    looking: C with scalar 2^16 and A = [2^16]C + T' with scalar 1 (T' = (-omega, 0), util.t_prime) make the Horner step
    [2^16]C + A exceptional (Z3 = 0) although the true sum is an ordinary point.  Every add / dbl of the tail now checks:
    the combine entry points return MSM377_EEXCEPTIONAL, the optional fold leaves its records alone, and the same
    records in Weierstrass form combine to the right point."""
    from webgpu_msm_bls12_377_amd.host.engine import EEXCEPTIONAL, fold_partials_bytes

    rnd = random.Random(21)
    tp = util.t_prime()
    c = R.mul(R.G, 777)
    a = R.add(R.mul(c, 1 << 16), tp)
    assert R.on_curve(a)
    win = [[None] * 16 for _ in range(16)]
    win[0][0], win[1][0] = a, c
    expect = R.encode_result(R.add(a, R.mul(c, 1 << 16)))
    te = b"".join(te_record(win[w], rnd) for w in range(16))
    with pytest.raises(msm.MsmError) as e:
        msm.combine_partials(te)
    assert e.value.code == EEXCEPTIONAL
    # the pair in two windows of one rank: folding would hit the same case, so the records stay as they are
    assert fold_partials_bytes(te[: 2 * E.WINDOW_PARTIAL_BYTES]) == te[: 2 * E.WINDOW_PARTIAL_BYTES]
    # mixed forms: the Edwards chain alone is still exceptional
    mixed = b"".join(te_record(win[w], rnd) if w < 2 else record(win[w], rnd) for w in range(16))
    with pytest.raises(msm.MsmError) as e:
        msm.combine_partials(mixed)
    assert e.value.code == EEXCEPTIONAL
    # one of the two in Weierstrass form: the chains no longer meet inside the Edwards law
    split = b"".join(record(win[w], rnd) if w == 0 else te_record(win[w], rnd) for w in range(16))
    assert msm.combine_partials(split) == expect
    assert msm.combine_partials(b"".join(record(win[w], rnd) for w in range(16))) == expect


def test_tail_in_pieces_matches_the_single_chain():
    """The threaded host tail cuts the 256-position Horner chain into balanced pieces (csrc/fp64_host.hpp tail_split,
    teh_tail_piece; doublings without T between the pieces) and adds them up.  msm377_g1_combine_partials_split runs that
    decomposition on the calling thread: every piece count gives the single chain's answer, which is the definition's;
    and an exceptional case inside one piece is still reported."""
    from webgpu_msm_bls12_377_amd.host.engine import EEXCEPTIONAL, combine_partials_split_bytes

    rnd = random.Random(31)
    base = [R.mul(R.G, rnd.randrange(1, R.R_ORDER)) for _ in range(9)]
    win_pts, expect = [], None
    for w in range(16):
        pts16 = [rnd.choice(base + [None]) for _ in range(16)]
        win_pts.append(pts16)
        g = pts16[0]
        for l in range(15):
            g = R.add(g, R.mul(pts16[1 + l], 1 << l))
        expect = R.add(expect, R.mul(g, 1 << (16 * w)))
    recs = b"".join(te_record(win_pts[w], rnd) for w in range(16))
    exp = R.encode_result(expect)
    assert msm.combine_partials(recs) == exp
    for pieces in (1, 2, 3, 4, 5, 6, 7, 8, 13, 64):
        assert combine_partials_split_bytes(recs, pieces) == exp, pieces
    # Weierstrass records: not this entry point's business
    with pytest.raises(msm.MsmError):
        combine_partials_split_bytes(b"".join(record(win_pts[w], rnd) for w in range(16)), 4)
    # the exceptional pair of test_edwards_tail_reports_exceptional_cases: caught whichever piece holds it
    tp = util.t_prime()
    c = R.mul(R.G, 777)
    a = R.add(R.mul(c, 1 << 16), tp)
    win = [[None] * 16 for _ in range(16)]
    win[0][0], win[1][0] = a, c
    bad = b"".join(te_record(win[w], rnd) for w in range(16))
    for pieces in (1, 2, 6, 8):
        with pytest.raises(msm.MsmError) as e:
            combine_partials_split_bytes(bad, pieces)
        assert e.value.code == EEXCEPTIONAL, pieces
