"""Pin the CPU oracle: the reference's in-tree constants / known answers, the committed golden
vectors (Python big integers) and the oracle's own cross-checks.  CPU only.

What the reference's tests hold for this path and where it is used here:
  cuzk/bls12_377.ts:10-12,21-29        modulus, generator            test_constants
  miscellaneous/tests/bls12_377.test.ts:8-35  projective->affine KAT, negation   test_bls12_377_test_ts_*
  miscellaneous/tests/cuzk.test.ts:26-114     16-point pipeline == naive sum     test_cuzk_test_ts_pipeline
  cuzk/utils.ts:448-533                Montgomery constants          test_montgomery_constants
  src/test-data/testCases.ts:14-26     2^16..2^20 answers (inputs absent): on-curve only
"""
import ctypes
import random

import numpy as np
import pytest

import pyref as R
import util


def test_constants(oracle):
    assert R.P.bit_length() == 377 and R.R_ORDER.bit_length() == 253
    g = ctypes.create_string_buffer(96)
    oracle.oracle_g1_generator(ctypes.addressof(g))
    assert R.decode_points(g.raw)[0] == R.G
    assert R.on_curve(R.G) and R.on_curve(R.FIXED_BASE)
    assert oracle.oracle_g1_on_curve(g.raw) == 1
    assert oracle.oracle_g1_on_curve(R.encode_points([R.FIXED_BASE])) == 1
    assert oracle.oracle_g1_on_curve(R.encode_points([(R.G[0], R.G[1] + 1)])) == 0


def test_montgomery_constants(oracle):
    # the reference's 13-bit-limb parameters (cuzk/utils.ts:435-533): num_words, n0, R
    num_words = 377 // 13
    while num_words * 13 <= 377:
        num_words += 1
    assert num_words == 30
    assert (-pow(R.P, -1, 1 << 13)) % (1 << 13) == 8191
    # this build's radices
    assert (-pow(R.P, -1, 1 << 64)) % (1 << 64) == 0x8508BFFFFFFFFFFF
    assert (-pow(R.P, -1, 1 << 29)) % (1 << 29) == (1 << 29) - 1
    r1, r2 = ctypes.create_string_buffer(48), ctypes.create_string_buffer(48)
    oracle.oracle_fp_mont_constants(ctypes.addressof(r1), ctypes.addressof(r2))
    assert int.from_bytes(r1.raw, "little") == (1 << 384) % R.P
    assert int.from_bytes(r2.raw, "little") == (1 << 768) % R.P


def test_field_ops_against_python(oracle):
    rnd = random.Random(20230807)
    vals = [0, 1, 2, R.P - 1, R.P - 2, (1 << 376), (1 << 377) - 1 - (1 << 377) + R.P - 3]
    vals += [rnd.randrange(R.P) for _ in range(300)]
    for _ in range(400):
        a, b = rnd.choice(vals), rnd.choice(vals)
        m, s, d = (ctypes.create_string_buffer(48) for _ in range(3))
        oracle.oracle_fp_ops(a.to_bytes(48, "little"), b.to_bytes(48, "little"), ctypes.addressof(m), ctypes.addressof(s), ctypes.addressof(d))
        assert int.from_bytes(m.raw, "little") == a * b % R.P
        assert int.from_bytes(s.raw, "little") == (a + b) % R.P
        assert int.from_bytes(d.raw, "little") == (a - b) % R.P  # fr_sub(a, a) is 0 here, not p


def test_bls12_377_test_ts_projective_to_affine(oracle):
    x = 256948617686061222151099205917657017678095364246733267446961262336245752294716969877488967765059276401456368208320
    y = 227105170010858909588581432763679191613848062092956907254157646977554095359559069986365111542605613311147866154268
    z = 200530079991103180348766713932858124680528544393104247922429137530783589900802654704895259718337700840462356557585
    ex = 100406495097683584255358201597988016233591504530780704496495127329407856735949558049179479777252726970296785896216
    ey = 63807138026163771468611662767681672353158802952448833583661885801557782260041754817382759731402212350042477212809
    out = ctypes.create_string_buffer(96)
    xyz = b"".join(v.to_bytes(48, "little") for v in (x, y, z))
    assert oracle.oracle_g1_proj_to_affine(xyz, ctypes.addressof(out)) == 0
    assert R.decode_points(out.raw)[0] == (ex, ey)
    # negation: createAffinePoint(x, p - y, z) == p.negate()
    xyz_neg = b"".join(v.to_bytes(48, "little") for v in (x, R.P - y, z))
    assert oracle.oracle_g1_proj_to_affine(xyz_neg, ctypes.addressof(out)) == 0
    assert R.decode_points(out.raw)[0] == R.neg((ex, ey))


def test_test_cases_ts_answers_are_on_the_curve(oracle):
    # src/test-data/testCases.ts:14-26; the inputs are not in the tree, so only this much can be checked
    answers = [
        (94006842082116618334698674554269938560504658220442275405704974851793018623976750030932275315377339755327327987799, 20373698276638985490622302772174938574967913528479846848006540077491753947648956036093654307050792702539840457541),
        (206224560584082546776307678440614275320062113355561962308721799926405988566792861311857124914191508657092244026797, 211505771810605149801236229583532591257930087722075039263647957125630724803810862016000585191202320499088754389346),
        (213590253091531711003295174396041900486736230199904022674226470027355022490783453188751023812621283421365133044335, 166168294849747437548140695864136486986897221068029518430368940173172785864820517559403857089626657281214248033436),
        (227918075012010659569854027573177112762469117095506192259456355647196733855535622181356473956903755312919537388289, 232048820726736272000228087347068589163288439026577981179126188061989792518064409423298246183820422050991578154066),
        (105645455159295492078411402285457085811978509815703136952786959329738979428758249440990135440135199333488003965024, 217434031274260429359512002379640961971443333898312105830518865556255108267359047513395163712830071551228264849716),
    ]
    for pt in answers:
        assert R.on_curve(pt)
        assert oracle.oracle_g1_on_curve(R.encode_points([pt])) == 1


def test_golden_vectors(oracle, golden):
    for name, case in golden.items():
        if not name.startswith("g1_"):
            continue
        for fn in ("oracle_g1_msm", "oracle_g1_msm_naive"):
            assert util.oracle_msm(oracle, case["points"], case["scalars"], fn) == case["expected"], (name, fn)


def test_cuzk_test_ts_pipeline(oracle, golden):
    """cuzk.test.ts:26-114: every bucket-reduction variant and window size agrees with the naive sum."""
    case = golden["g1_n16_cuzk_test"]
    naive = util.oracle_msm(oracle, case["points"], case["scalars"], "oracle_g1_msm_naive")
    assert naive == case["expected"]
    for c, T in ((4, 8), (4, 4), (4, 1), (8, 16), (8, 128), (16, 256), (16, 32768)):
        assert util.oracle_msm_params(oracle, case["points"], case["scalars"], c, T) == naive, (c, T)


def test_reference_sized_pipeline_on_small_inputs(oracle, golden):
    """The reference's production parameters (16-bit windows, 256 BPR threads, submission.ts:97,230)
    on the golden inputs, plus Horner over the window sums (submission.ts:310-318)."""
    for name in ("g1_n33_random", "g1_n20_edge_scalars", "g1_n64_same_point", "g1_n48_repeats_and_negs", "g1_n2_cancel"):
        case = golden[name]
        res, ws = util.oracle_msm_params(oracle, case["points"], case["scalars"], 16, 256, want_windows=True)
        assert res == case["expected"], name
        out = ctypes.create_string_buffer(96)
        assert oracle.oracle_g1_horner(ws, 16, 16, ctypes.addressof(out)) == 0
        assert out.raw == case["expected"], name


def test_decompose_scalars_signed(oracle):
    """cuzk/utils.ts:66-109: digits rebuild the scalar, lie in [-2^(c-1), 2^(c-1)), and a final carry is an error."""
    rnd = random.Random(7)
    ks = [0, 1, R.R_ORDER - 1, (1 << 15), (1 << 16) - 1] + [rnd.randrange(R.R_ORDER) for _ in range(200)]
    buf = R.encode_scalars(ks)
    for c in (4, 8, 16):
        W = (256 + c - 1) // c
        chunks = np.zeros(W * len(ks), dtype=np.uint32)
        assert oracle.oracle_decompose_scalars_signed(buf, len(ks), c, chunks.ctypes.data) == 0
        chunks = chunks.reshape(W, len(ks)).astype(np.int64) - (1 << (c - 1))
        assert chunks.min() >= -(1 << (c - 1)) and chunks.max() < (1 << (c - 1))
        for i, k in enumerate(ks):
            assert sum(int(chunks[w, i]) << (c * w) for w in range(W)) == k
    bad = R.encode_scalars([(1 << 256) - 1])
    chunks = np.zeros(16, dtype=np.uint32)
    assert oracle.oracle_decompose_scalars_signed(bad, 1, 16, chunks.ctypes.data) == -1


def test_cpu_transpose(oracle):
    """cuzk/transpose.ts:14-62: row_ptr are exclusive offsets of the per-digit counts and val_idx lists
    the point indices of each digit in input order (stable)."""
    rnd = np.random.RandomState(3)
    n, ncols, W = 1000, 16, 3
    chunks = rnd.randint(0, ncols, size=W * n).astype(np.uint32)
    rp = np.zeros(W * (ncols + 1), dtype=np.uint32)
    vi = np.zeros(W * n, dtype=np.uint32)
    oracle.oracle_cpu_transpose(chunks.ctypes.data, n, ncols, W, rp.ctypes.data, vi.ctypes.data)
    for w in range(W):
        col = chunks[w * n : (w + 1) * n]
        counts = np.bincount(col, minlength=ncols)
        assert list(rp[w * (ncols + 1) : (w + 1) * (ncols + 1)]) == [0] + list(np.cumsum(counts))
        expect = np.argsort(col, kind="stable")
        assert list(vi[w * n : (w + 1) * n]) == list(expect)


def test_gen_points_arith(oracle):
    n, a0, d = 300, 0x1234567, 0xABCDEF0123
    pts = R.decode_points(util.oracle_gen_points(oracle, n, a0, d))
    for i in (0, 1, 2, 16, 17, 18, 150, 299):
        assert pts[i] == R.mul(R.G, a0 + i * d)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 17, 100])
def test_pipeline_matches_naive_random(oracle, n):
    pts = [R.mul(R.G, a % R.R_ORDER or 1) for a in R.rand_scalars(1000 + n, n)]
    ks = R.rand_scalars(2000 + n, n)
    pb, sb = R.encode_points(pts), R.encode_scalars(ks)
    exp = R.encode_result(R.msm_naive(pts, ks))
    assert util.oracle_msm(oracle, pb, sb) == exp
    assert util.oracle_msm(oracle, pb, sb, "oracle_g1_msm_naive") == exp
