"""The product's device math headers (csrc/field29.hpp, g1_xyzz.hpp, fp64_host.hpp) compiled for the
host and checked against Python big integers.  CPU only.  The same headers compile into the HIP
kernels; csrc/microbench.hip checks device == host on the GPU."""
import ctypes
import os
import random
import struct
import sys
import subprocess

import pytest

import pyref as R
import util

ROOT = util.ROOT
CSRC = os.path.join(ROOT, "webgpu-msm-bls12-377_amd", "csrc")
SHIM_SRC = os.path.join(ROOT, "tests", "native", "field29_shim.cpp")
SHIM_SO = os.path.join(ROOT, "tests", "native", "_build", "libfield29_shim.so")


@pytest.fixture(scope="module")
def shim():
    deps = [SHIM_SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    if not os.path.exists(SHIM_SO) or any(os.path.getmtime(d) > os.path.getmtime(SHIM_SO) for d in deps):
        os.makedirs(os.path.dirname(SHIM_SO), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-I", CSRC, "-o", SHIM_SO, SHIM_SRC])
    return ctypes.CDLL(SHIM_SO)


def words12(v):
    return (ctypes.c_uint32 * 12)(*[(v >> (32 * i)) & 0xFFFFFFFF for i in range(12)])


def from_words(arr):
    return sum(int(w) << (32 * i) for i, w in enumerate(arr))


def test_field_ops(shim):
    rnd = random.Random(377)
    special = [0, 1, 2, R.P - 1, R.P - 2, (1 << 376), (1 << 29) - 1, 1 << 29, (R.P - 1) // 2, (1 << 348) - 1, R.P - (1 << 348)]
    vals = special + [rnd.randrange(R.P) for _ in range(400)]
    for it in range(1500):
        a, b = (rnd.choice(vals), rnd.choice(vals)) if it >= len(special) ** 2 else (special[it // len(special)], special[it % len(special)])
        outs = [(ctypes.c_uint32 * 12)() for _ in range(5)]
        shim.shim_fp_ops(words12(a), words12(b), *outs)
        got = [from_words(o) for o in outs]
        assert got == [a * b % R.P, (a + b) % R.P, (a - b) % R.P, a * a % R.P, (-a) % R.P], (a, b)


def test_fused_mul_sub_mul(shim):
    """a*b - c*d with one reduction (Y3 of every point addition): extremes maximise the unreduced value."""
    rnd = random.Random(99)
    special = [0, 1, R.P - 1, R.P - 2, (R.P - 1) // 2, 1 << 376]
    vals = special + [rnd.randrange(R.P) for _ in range(60)]
    for it in range(2000):
        a, b, c, d = (rnd.choice(vals) for _ in range(4)) if it >= 300 else (rnd.choice(special) for _ in range(4))
        out = (ctypes.c_uint32 * 12)()
        shim.shim_fp_mul_sub_mul(words12(a), words12(b), words12(c), words12(d), out)
        assert from_words(out) == (a * b - c * d) % R.P, (a, b, c, d)


def test_montgomery_limb_format(shim):
    rnd = random.Random(1)
    for v in [0, 1, R.P - 1] + [rnd.randrange(R.P) for _ in range(50)]:
        limbs = (ctypes.c_uint32 * 13)()
        shim.shim_fp_to_mont_limbs(words12(v), limbs)
        assert list(limbs) == util.to_limbs29_mont(v)
        assert all(x < (1 << 29) for x in limbs)


def xyzz_buf(pt, z=1):
    """XYZZ words of an affine point scaled by a projective factor z (ZZ = z^2, ZZZ = z^3)."""
    if pt is None:
        return (ctypes.c_uint32 * 52)(*util.xyzz_words_from_affine(None))
    x, y = pt
    zz, zzz = z * z % R.P, z * z * z % R.P
    w = util.to_limbs29_mont(x * zz % R.P) + util.to_limbs29_mont(y * zzz % R.P) + util.to_limbs29_mont(zz) + util.to_limbs29_mont(zzz)
    return (ctypes.c_uint32 * 52)(*w)


def xy24(pt):
    return (ctypes.c_uint32 * 24)(*([(pt[0] >> (32 * i)) & 0xFFFFFFFF for i in range(12)] + [(pt[1] >> (32 * i)) & 0xFFFFFFFF for i in range(12)]))


def test_curve_ops_including_special_cases(shim):
    rnd = random.Random(5)
    pts = [R.mul(R.G, rnd.randrange(1, R.R_ORDER)) for _ in range(12)]
    out = (ctypes.c_uint32 * 52)()
    cases = []
    for _ in range(60):
        cases.append((rnd.choice(pts), rnd.choice(pts)))
    cases += [(p, p) for p in pts[:4]] + [(p, R.neg(p)) for p in pts[:4]] + [(None, p) for p in pts[:3]]
    for a, b in cases:
        z1, z2 = rnd.randrange(1, R.P), rnd.randrange(1, R.P)
        # mixed add
        shim.shim_g1_madd(xyzz_buf(a, z1), xy24(b), out)
        assert util.affine_from_xyzz_words(list(out)) == R.add(a, b), ("madd", a, b)
        # general add, both operands with non-trivial ZZ/ZZZ, both orders
        shim.shim_g1_add(xyzz_buf(a, z1), xyzz_buf(b, z2), out)
        assert util.affine_from_xyzz_words(list(out)) == R.add(a, b), ("add", a, b)
        shim.shim_g1_add(xyzz_buf(b, z2), xyzz_buf(a, z1), out)
        assert util.affine_from_xyzz_words(list(out)) == R.add(a, b)
        shim.shim_g1_add(xyzz_buf(b, z2), xyzz_buf(None), out)
        assert util.affine_from_xyzz_words(list(out)) == b
        # doubling
        shim.shim_g1_dbl(xyzz_buf(b, z2), out)
        assert util.affine_from_xyzz_words(list(out)) == R.add(b, b)
        # host-tail field (64-bit words) through the same template
        w = ctypes.create_string_buffer(96)
        shim.shim_g1h_add_to_wire(xyzz_buf(a, z1), xyzz_buf(b, z2), w)
        assert w.raw == R.encode_result(R.add(a, b))
    shim.shim_g1_dbl(xyzz_buf(None), out)
    assert util.affine_from_xyzz_words(list(out)) is None


def test_lazy_accumulation_chains(shim):
    """k_accumulate's inner loop on the host: the accumulator stays in its stored form (X below 5p, not reduced
    mod p -- csrc/g1_xyzz.hpp madd_lz) across mixed additions with signs, repeated and opposite points, then the
    partial sums are added with the general formula.  Stored limbs must stay carry-normalised."""
    rnd = random.Random(2024)
    pts = [R.mul(R.G, rnd.randrange(1, R.R_ORDER)) for _ in range(10)]
    partials, expected_total = [], None
    for trial in range(12):
        count = rnd.choice([1, 2, 5, 40])
        seq = [rnd.choice(pts) for _ in range(count)]
        negs = [rnd.randrange(2) for _ in range(count)]
        if trial == 3:  # P, P, -P, -P ...: doubling and cancellation inside a chain
            seq, negs = [pts[0]] * 6, [0, 0, 1, 1, 1, 0]
        start = rnd.choice([None, pts[1]])
        z = rnd.randrange(1, R.P)
        q = (ctypes.c_uint32 * (24 * count))(*[w for pt in seq for w in xy24(pt)])
        out = (ctypes.c_uint32 * 52)()
        shim.shim_g1_madd_chain(xyzz_buf(start, z), q, (ctypes.c_uint8 * count)(*negs), count, out)
        exp = start
        for pt, ng in zip(seq, negs):
            exp = R.add(exp, R.neg(pt) if ng else pt)
        assert util.affine_from_xyzz_words(list(out)) == exp, trial
        words = list(out)
        for c in range(4):
            assert all(w < (1 << 29) for w in words[13 * c : 13 * c + 12]), "limbs 0..11 carry-normalised"
        assert sum(int(w) << (29 * i) for i, w in enumerate(words[0:13])) < 5 * R.P + (1 << 354)
        for c in range(1, 4):
            assert sum(int(w) << (29 * i) for i, w in enumerate(words[13 * c : 13 * c + 13])) < R.P + (1 << 354)
        partials.append(words)
        expected_total = R.add(expected_total, exp)
    flat = (ctypes.c_uint32 * (52 * len(partials)))(*[w for p in partials for w in p])
    out = (ctypes.c_uint32 * 52)()
    shim.shim_g1_add_chain(flat, len(partials), out)
    assert util.affine_from_xyzz_words(list(out)) == expected_total


def test_twisted_edwards_form_of_g1(shim):
    """csrc/te377.hpp on the host: wire point -> projective Edwards record -> mixed additions with signs (repeated
    and opposite points included: the unified law has no special cases) -> general additions -> back to the
    Weierstrass wire format; and the host tail's doubling / addition through a scalar multiplication."""
    rnd = random.Random(77)
    pts = [R.mul(R.G, rnd.randrange(1, R.R_ORDER)) for _ in range(9)]
    beta_pts = None
    for trial in range(14):
        count = rnd.choice([1, 2, 3, 7, 30])
        parts = rnd.choice([1, 2, 3])
        seq = [rnd.choice(pts) for _ in range(count)]
        negs = [rnd.randrange(2) for _ in range(count)]
        if trial == 0:
            seq, negs, count, parts = [pts[0]] * 5, [0, 0, 1, 0, 1], 5, 1  # P + P - P + P - P
        if trial == 1:
            seq, negs, count, parts = [pts[2], pts[2]], [0, 1], 2, 1  # sums to the identity
        phi = trial % 5 == 4
        q = (ctypes.c_uint32 * (24 * count))(*[w for pt in seq for w in xy24(pt)])
        out = ctypes.create_string_buffer(96)
        ext = (ctypes.c_uint32 * 52)()
        bad = shim.shim_te_sum(q, (ctypes.c_uint8 * count)(*negs), count, parts, int(phi), out, ext)
        assert bad == 0
        exp = None
        for pt, ng in zip(seq, negs):
            if phi:
                pt = R.mul(pt, 0x8508C00000000001 ** 2 - 1)  # phi(P) = [LAMBDA] P
            exp = R.add(exp, R.neg(pt) if ng else pt)
        assert out.raw == R.encode_result(exp), trial
        if not phi:  # affine records of resident tables: same sum with 7-product additions
            out2 = ctypes.create_string_buffer(96)
            assert shim.shim_te_sum_affine(q, (ctypes.c_uint8 * count)(*negs), count, out2) == 0
            assert out2.raw == out.raw
        words = list(ext)
        for c in range(4):  # stored coordinates: lazy products, carry-normalised, below p + 2^354
            assert all(w < (1 << 29) for w in words[13 * c : 13 * c + 13])
            assert sum(int(w) << (29 * i) for i, w in enumerate(words[13 * c : 13 * c + 13])) < R.P + (1 << 354)
    for k in [0, 1, 2, R.R_ORDER - 1, rnd.randrange(R.R_ORDER), (1 << 253) - 1]:
        out = ctypes.create_string_buffer(96)
        shim.shim_teh_scalar_mul(xy24(pts[3]), (ctypes.c_uint32 * 8)(*[(k >> (32 * i)) & 0xFFFFFFFF for i in range(8)]), out)
        assert out.raw == R.encode_result(R.mul(pts[3], k)), k


def test_twisted_edwards_exceptional_inputs_are_flagged(shim):
    """The two curve points the map does not cover -- (-1, 0) of order 2 and the order-4 points over it -- and
    sums that land on a point at infinity of the Edwards model must raise the flag (the engine then reruns on the
    Weierstrass path); they can only come from outside the prime-order subgroup."""
    two_torsion = (R.P - 1, 0)
    assert R.add(two_torsion, two_torsion) is None
    out = ctypes.create_string_buffer(96)
    ext = (ctypes.c_uint32 * 52)()
    assert shim.shim_te_sum(xy24(two_torsion), (ctypes.c_uint8 * 1)(0), 1, 1, 0, out, ext) == 1
    te = util.te_params()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_consts

    x4 = (-1 - pow(te["s"], -1, R.P)) % R.P  # the order-4 point with s (x + 1) = -1
    t4 = (x4, gen_consts._sqrt_p((x4 ** 3 + 1) % R.P))
    assert R.add(t4, t4) == two_torsion
    assert shim.shim_te_sum(xy24(t4), (ctypes.c_uint8 * 1)(0), 1, 1, 0, out, ext) == 1
    # Q = P + T2 for a subgroup point P: P - Q = T2 is exceptional for the Edwards law
    p = R.mul(R.G, 12345)
    qpt = R.add(p, two_torsion)
    q = (ctypes.c_uint32 * 48)(*(list(xy24(p)) + list(xy24(qpt))))
    bad = shim.shim_te_sum(q, (ctypes.c_uint8 * 2)(0, 1), 2, 1, 0, out, ext)
    assert bad == 1 or out.raw == R.encode_result(two_torsion)  # T2 = (-1, 0) is a finite point of the model: not exceptional
    # T' = (-omega, 0), the 2-torsion point the model sends to infinity: conversion flags T' itself, and P, P + T' are
    # an exceptional pair under BOTH signs -- the flag must be raised by the addition (is_bad), in a plain chain
    # (k_accumulate) and when the pair meets as partial sums of a general addition (merge / tree levels).
    tp = util.t_prime()
    assert shim.shim_te_sum(xy24(tp), (ctypes.c_uint8 * 1)(0), 1, 1, 0, out, ext) == 1
    qt = R.add(p, tp)
    assert R.on_curve(qt)
    q = (ctypes.c_uint32 * 48)(*(list(xy24(p)) + list(xy24(qt))))
    for negs in ((0, 0), (0, 1), (1, 0), (1, 1)):
        assert shim.shim_te_sum(q, (ctypes.c_uint8 * 2)(*negs), 2, 1, 0, out, ext) == 1, negs  # chain: P then +-(P + T')
        assert shim.shim_te_sum(q, (ctypes.c_uint8 * 2)(*negs), 2, 2, 0, out, ext) == 1, negs  # two partial sums, general add
    assert shim.shim_te_sum_affine(q, (ctypes.c_uint8 * 2)(0, 0), 2, out) == 1  # affine records (7-product additions)
    # three points where only the LAST addition is exceptional: (P + R) + (P + R + T')
    r2 = R.mul(R.G, 999)
    q3 = (ctypes.c_uint32 * 72)(*(list(xy24(p)) + list(xy24(r2)) + list(xy24(R.add(R.add(p, r2), tp)))))
    assert shim.shim_te_sum(q3, (ctypes.c_uint8 * 3)(0, 0, 0), 3, 1, 0, out, ext) == 1
    # ... and a non-exceptional sum over the same kind of points stays exact
    q3b = (ctypes.c_uint32 * 72)(*(list(xy24(p)) + list(xy24(qt)) + list(xy24(r2))))
    assert shim.shim_te_sum(q3b, (ctypes.c_uint8 * 3)(0, 0, 0), 3, 3, 0, out, ext) in (0, 1)
    q2 = (ctypes.c_uint32 * 48)(*(list(xy24(r2)) + list(xy24(qt))))
    assert shim.shim_te_sum(q2, (ctypes.c_uint8 * 2)(0, 0), 2, 1, 0, out, ext) == 0
    assert out.raw == R.encode_result(R.add(r2, qt))


def test_adx_multiplier_matches_the_portable_one(shim):
    """csrc/fp64_host.hpp mont_mul6_adx (mulx + adcx / adox, used by the host tail and the affine conversion's host
    inversion) against the portable unsigned __int128 multiplier AND Python integers: random operands, operands at and
    around p, zero, all-ones words."""
    rnd = random.Random(64)
    R64 = 1 << 384
    ri = pow(R64, -1, R.P)
    special = [0, 1, R.P - 1, R.P - 2, R.P, (1 << 384) - 1, (1 << 376) - 1, 1 << 383, R64 % R.P]
    cases = [(a, b) for a in special for b in special] + [(rnd.randrange(R.P), rnd.randrange(R.P)) for _ in range(20000)]
    have_adx = None
    for a, b in cases:
        av = (ctypes.c_uint64 * 6)(*[(a >> (64 * i)) & (2**64 - 1) for i in range(6)])
        bv = (ctypes.c_uint64 * 6)(*[(b >> (64 * i)) & (2**64 - 1) for i in range(6)])
        oa, op = (ctypes.c_uint64 * 6)(), (ctypes.c_uint64 * 6)()
        have_adx = shim.shim_fp64_mul_both(av, bv, oa, op)
        got_p = sum(int(w) << (64 * i) for i, w in enumerate(op))
        if a < R.P and b < R.P:
            assert got_p == a * b * ri % R.P, (hex(a), hex(b))
        if have_adx:
            assert list(oa) == list(op), (hex(a), hex(b))
    if not have_adx:
        pytest.skip("this CPU has no BMI2 / ADX: the portable multiplier is the only one")


def test_divstep_inversion_matches_python_and_fermat(shim):
    """csrc/fp64_host.hpp inv: the Bernstein-Yang divstep iteration (62 steps at a time, variable time) that replaced the
    a^(p-2) exponentiation in the host tail and in the affine conversion's block inversion.  Its plain integer result
    against Python's pow(a, -1, p), and the Montgomery-form inverse against the exponentiation's, for both host fields:
    0 (no inverse: inv(0) = 0 like the exponentiation), 1, p - 1, small values, values around 2^62 k limb boundaries,
    and random ones."""
    rnd = random.Random(62)
    for which, p, nw in ((0, R.P, 6), (1, R.Q, 4)):
        special = [0, 1, 2, 3, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, (1 << 62) - 1, 1 << 62, (1 << 62) + 1, (1 << 124) - 1, 1 << 124,
                   (1 << 186) + 5, (1 << 248) - 1, p - (1 << 62), p - (1 << 124) + 1]
        vals = [v % p for v in special] + [rnd.randrange(1, 1 << rnd.choice([8, 61, 63, 200, 250])) % p for _ in range(300)]
        vals += [rnd.randrange(p) for _ in range(3000)]
        mask = (1 << 64) - 1
        for a in vals:
            av = (ctypes.c_uint64 * nw)(*[(a >> (64 * i)) & mask for i in range(nw)])
            plain, mont, fermat = ((ctypes.c_uint64 * nw)() for _ in range(3))
            ok = shim.shim_fp64_inv(which, av, plain, mont, fermat)
            assert list(mont) == list(fermat), hex(a)
            if a == 0:
                assert ok == 0 and all(w == 0 for w in mont)
            else:
                assert ok == 1, hex(a)
                assert sum(int(w) << (64 * i) for i, w in enumerate(plain)) == pow(a, -1, p), hex(a)


def test_host_tail_in_the_even_window_geometries(shim):
    """csrc/fp64_host.hpp tail_position / teh_combine / g1h_combine with `short_from`: the host tail of a whole MSM on the
    main path (thirteen 16-bit windows, then three 15-bit ones: bit offsets 208, 223, 238), of the small-input path
    (eleven 12-bit windows, then eleven 11-bit ones, 11 bit planes) and of the uniform layouts, against the definition
    sum_w 2^(offset_w) (Sum_w + sum_l 2^l Plane_{w,l}) computed with Python integers.  Records as the GPU writes them
    (16 points per window; planes beyond the geometry's count hold garbage that must not be read)."""
    import struct

    rnd = random.Random(253)
    base = [R.mul(R.G, rnd.randrange(1, R.R_ORDER)) for _ in range(9)]

    def case(num_windows, cbits, planes, short_from, form):
        recs, expect, offset = [], None, 0
        for w in range(num_windows):
            width = cbits - 1 if (short_from and w >= short_from) else cbits
            pts16 = [rnd.choice(base + [None]) for _ in range(max(16, planes + 1))]  # (the wide table's one record has 20 points)
            g = pts16[0]
            for l in range(planes):
                if l < width:  # a plane at or beyond the window's width is never produced by the reduction: leave it out of the sum ...
                    g = R.add(g, R.mul(pts16[1 + l], 1 << l))
                else:
                    pts16[1 + l] = None  # ... and out of the record
            words = []
            for i, pt in enumerate(pts16):
                z = rnd.randrange(1, R.P)
                words += util.record_point_words(pt, z) if form == 1 else util.te_record_point_words(pt, z, tag=(i == 0))
            recs.append(struct.pack("<%dI" % len(words), *words))
            expect = R.add(expect, R.mul(g, 1 << offset))
            offset += width
        assert shim.shim_tail_positions(num_windows, cbits, short_from) == offset
        buf = b"".join(recs)
        arr = (ctypes.c_uint32 * (len(buf) // 4)).from_buffer_copy(buf)
        out = ctypes.create_string_buffer(96)
        assert shim.shim_tail_combine_geom(arr, num_windows, cbits, planes, short_from, form, out) == 0
        assert out.raw == R.encode_result(expect), (num_windows, cbits, planes, short_from, form)
        return offset

    assert case(16, 16, 15, 13, 0) == 253  # whole MSMs on the main path
    assert case(16, 16, 15, 13, 1) == 253  # the same chain over Weierstrass records
    assert case(22, 12, 11, 11, 0) == 253  # the small-input path
    assert case(16, 16, 15, 0, 0) == 256  # sixteen equal windows (window shards, the fallback for scalars >= 2^253)
    assert case(23, 11, 11, 0, 0) == 253  # the small-input path's first geometry
    assert case(1, 20, 19, 0, 0) == 20  # one record of the wide-window table


def test_lazy_bounds_proof():
    """tools/check_lazy_bounds.py: interval replay of the lazy formulas -- no 64-bit column can overflow, no limb
    of a limb-wise subtraction can go negative, results meet the storage invariant."""
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "check_lazy_bounds.py")])
