"""GPU parity for the Twisted-Edwards BLS12 path (BASELINE.json config 3):
HIP engine through the C ABI against the Edwards oracle, the golden vectors and a closed form at 2^20.
Bit-exact.  Run with `pytest -m gpu`."""
import ctypes
import random

import pytest

import pyref as R
import util
import webgpu_msm_bls12_377_amd as msm
from ed_vectors import ADD_GROUPS, GROUP_SCALAR_MUL, MULTIPLY

pytestmark = pytest.mark.gpu


def dev(buf: bytes):
    import torch

    return torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()


def test_golden_vectors(engine, golden):
    for name, case in golden.items():
        if name.startswith("ed_"):
            assert engine.ed_msm(case["points"], case["scalars"]) == case["expected"], name


# ---- the reference's own Edwards vectors through the HIP path (n = 1 and n = 2 MSMs in disguise) ----
# Expected values are the reference's published answers, not the oracle's: this is what pins the GPU path to the
# reference for this curve (missing #1).


def _xy(buf: bytes):
    return (int.from_bytes(buf[:32], "little"), int.from_bytes(buf[32:], "little"))


@pytest.mark.parametrize("case", range(len(MULTIPLY)))
def test_reference_multiply_vectors_as_one_point_msm(engine, case):
    """src/reference/utils/FieldMath.test.ts:4-62: [k]P as the MSM of one point, host and device entry points."""
    pt, k, exp = MULTIPLY[case]
    pts, ks = R.ed_encode_points([pt]), R.encode_scalars([k])
    assert _xy(engine.ed_msm(pts, ks)) == exp
    d_p, d_s = dev(pts), dev(ks)
    assert _xy(engine.ed_msm_device(d_p.data_ptr(), d_s.data_ptr(), 1)) == exp


def test_reference_multiply_vectors_as_one_composite_msm(engine):
    """All five MULTIPLY pairs in ONE MSM: the expectation is the sum (pure Python, complete affine law) of the
    five answers the reference publishes -- no oracle involved."""
    pts = R.ed_encode_points([pt for pt, _, _ in MULTIPLY])
    ks = R.encode_scalars([k for _, k, _ in MULTIPLY])
    exp = R.ED_ID
    for _, _, res in MULTIPLY:
        exp = R.ed_add(exp, res)
    assert _xy(engine.ed_msm(pts, ks)) == exp
    # ... and each pair repeated in 40 copies with the scalar split into 40 random shares (fills real buckets)
    rnd = random.Random(5)
    big_pts, big_ks = [], []
    for pt, k, _ in MULTIPLY:
        shares = [rnd.randrange(R.ED_SUBGROUP) for _ in range(39)]
        shares.append((k - sum(shares)) % R.ED_SUBGROUP)
        big_pts += [pt] * 40
        big_ks += shares
    assert _xy(engine.ed_msm(R.ed_encode_points(big_pts), R.encode_scalars(big_ks))) == exp


@pytest.mark.parametrize("case", range(len(ADD_GROUPS)))
def test_reference_add_groups_vectors_as_two_point_msm(engine, case):
    """src/reference/utils/wasmFunctions.test.ts:24-36: P1 + P2 as the MSM with scalars (1, 1).  The reference's
    "group" strings carry x only; y comes from the decompression the reference pins in FieldMath.test.ts:64-98
    (tests/test_ed_oracle_pins.py::test_fieldmath_point_from_x_vectors).  Cases 3 and 4 are doublings (P1 = P2)."""
    x1, x2, x3 = ADD_GROUPS[case]
    a, b = R.ed_point_from_x(x1), R.ed_point_from_x(x2)
    got = _xy(engine.ed_msm(R.ed_encode_points([a, b]), R.encode_scalars([1, 1])))
    assert got[0] == x3 and R.ed_on_curve(got)


@pytest.mark.parametrize("case", range(len(GROUP_SCALAR_MUL)))
def test_reference_group_scalar_mul_vectors_as_one_point_msm(engine, case):
    """src/reference/utils/wasmFunctions.test.ts:38-49."""
    x, k, xr = GROUP_SCALAR_MUL[case]
    got = _xy(engine.ed_msm(R.ed_encode_points([R.ed_point_from_x(x)]), R.encode_scalars([k])))
    assert got[0] == xr and R.ed_on_curve(got)


def test_empty_input_is_the_neutral_element(engine):
    assert engine.ed_msm(b"", b"") == R.ed_encode_result(R.ED_ID)


@pytest.mark.parametrize("n", [1, 2, 3, 65, 257, 1000, 4097])
def test_ragged_sizes_against_oracle(engine, oracle, n):
    rnd = random.Random(n)
    pts = util.oracle_ed_gen_points(oracle, n, rnd.randrange(1, 1 << 200), rnd.randrange(1, 1 << 200))
    ks = R.encode_scalars(R.rand_scalars(500 + n, n, R.ED_SUBGROUP))
    assert engine.ed_msm(pts, ks) == util.oracle_ed_msm(oracle, pts, ks)


def test_2_16_against_reference_sized_oracle(engine, oracle):
    n = 1 << 16
    pts = util.oracle_ed_gen_points(oracle, n, 0xABCDEF, 0x13579B)
    ks = R.encode_scalars(R.rand_scalars(16, n, R.ED_SUBGROUP))
    exp = util.oracle_ed_msm(oracle, pts, ks)  # 16-bit windows, 256 BPR threads
    assert engine.ed_msm(pts, ks) == exp
    d_p, d_s = dev(pts), dev(ks)
    assert engine.ed_msm_device(d_p.data_ptr(), d_s.data_ptr(), n) == exp


def test_host_buffers_upload_in_two_chunks(oracle, monkeypatch):
    """msm377_ed_msm with large host buffers: two chunks of points, the second accumulating on top of the first
    (forced at small sizes through MSM377_UPLOAD_CHUNK_MIN)."""
    monkeypatch.setenv("MSM377_UPLOAD_CHUNK_MIN", "100")
    eng = msm.MsmEngine(1 << 17)
    try:
        for n in (128, 131, 1000, 4097, 70001):
            rnd = random.Random(900 + n)
            pts = util.oracle_ed_gen_points(oracle, n, rnd.randrange(1, 1 << 200), rnd.randrange(1, 1 << 200))
            ks = R.encode_scalars([rnd.randrange(R.R_ORDER) for _ in range(n)])
            assert eng.ed_msm(pts, ks) == util.oracle_ed_msm(oracle, pts, ks), n
    finally:
        eng.close()


def test_even_window_geometry_edge_scalars(engine, oracle, monkeypatch):
    """The top three windows are 15 bits wide (kernels/decompose.hpp k_decompose `even`): digits at their boundaries,
    carries through them, and scalars that do not fit (the call reruns with sixteen equal windows) -- from device
    buffers, from host buffers, and through the chunked upload."""
    s15 = 0x7FFF
    edge = [
        s15 << 208, s15 << 223, s15 << 238, (s15 << 208) + (0x8000 << 192), (s15 << 208) + (s15 << 223) + (0x8000 << 192),
        (s15 << 208) + (s15 << 223) + (s15 << 238) + (0x8000 << 192), (s15 << 238) + (s15 << 223) + (s15 << 208) + 0x7FFF,
        (1 << 253) - 1, 1 << 253, (1 << 254) + 5, (1 << 223) - 1, 1 << 223, (1 << 238) - 1, 1 << 238, 0, 1, R.ED_SUBGROUP - 1,
    ]
    n = len(edge)
    pts_b = util.oracle_ed_gen_points(oracle, n, 0x1234567, 0x7654321)
    pts = [(int.from_bytes(pts_b[64 * i : 64 * i + 32], "little"), int.from_bytes(pts_b[64 * i + 32 : 64 * i + 64], "little")) for i in range(n)]
    small = R.rand_scalars(4, n, R.ED_SUBGROUP)
    for i, k in enumerate(edge):
        kk = list(small)
        kk[i] = k
        assert engine.ed_msm(pts_b, R.encode_scalars(kk)) == R.ed_encode_result(R.ed_msm_naive(pts, kk)), hex(k)
    want = R.ed_encode_result(R.ed_msm_naive(pts, edge))
    assert engine.ed_msm(pts_b, R.encode_scalars(edge)) == want
    d_p, d_s = dev(pts_b), dev(R.encode_scalars(edge))
    assert engine.ed_msm_device(d_p.data_ptr(), d_s.data_ptr(), n) == want
    monkeypatch.setenv("MSM377_UPLOAD_CHUNK_MIN", "100")
    eng = msm.MsmEngine(1 << 12)
    try:
        reps = 8  # 136 points: four chunks
        ks8 = [k * (r + 1) % (1 << 253) if k < (1 << 253) else k for r in range(reps) for k in edge]
        assert eng.ed_msm(pts_b * reps, R.encode_scalars(ks8)) == R.ed_encode_result(R.ed_msm_naive(pts * reps, ks8))
        assert eng.ed_msm(pts_b * reps, R.encode_scalars(small * reps)) == R.ed_encode_result(R.ed_msm_naive(pts * reps, small * reps))
    finally:
        eng.close()


def test_one_repeated_point_and_opposites(engine):
    """The complete addition law must cover P + P and P + (-P) inside buckets."""
    p = R.ed_mul(R.ED_G, 123456789)
    pts = [p] * 500 + [R.ed_neg(p)] * 500
    ks = R.rand_scalars(77, 1000, R.ED_SUBGROUP)
    total = (sum(ks[:500]) - sum(ks[500:])) % R.ED_SUBGROUP
    assert engine.ed_msm(R.ed_encode_points(pts), R.encode_scalars(ks)) == R.ed_encode_result(R.ed_mul(p, total))


def test_generate_bases(engine):
    import torch

    n, seed = 24, 0xED
    out = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
    engine.ed_generate_bases_device(seed, n, out.data_ptr())
    raw = out.cpu().numpy().tobytes()
    g = R.splitmix64(seed)
    for i in range(n):
        pt = (int.from_bytes(raw[64 * i : 64 * i + 32], "little"), int.from_bytes(raw[64 * i + 32 : 64 * i + 64], "little"))
        assert pt == R.ed_mul(R.ED_G, next(g))


def test_full_size_2_20_closed_form(engine, oracle):
    """P_i = [a0 + i d]G_ed: the 2^20-point MSM equals one scalar multiplication of the generator."""
    n = 1 << 20
    a0, d = 0x1234567890ABCDEF12345, 0xFEDCBA098765
    pts = util.oracle_ed_gen_points(oracle, n, a0, d)
    ks_int = R.rand_scalars(0x5CA1A5, n, R.ED_SUBGROUP)
    total = sum(k * (a0 + i * d) for i, k in enumerate(ks_int)) % R.ED_SUBGROUP
    exp = ctypes.create_string_buffer(64)
    assert oracle.oracle_ed_scalar_mul(R.ed_encode_points([R.ED_G]), total.to_bytes(32, "little"), 32, ctypes.addressof(exp)) == 0
    d_p, d_s = dev(pts), dev(R.encode_scalars(ks_int))
    engine.set_timing(True)
    try:
        got = engine.ed_msm_device(d_p.data_ptr(), d_s.data_ptr(), n)
        print("edwards stage ms:", engine.stage_ms())
    finally:
        engine.set_timing(False)
    assert got == exp.raw
