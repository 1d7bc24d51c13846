#!/usr/bin/env python3
"""Mint tests/golden/ : seeded inputs + expected MSM results from Python big integers.

TEST INFRASTRUCTURE.  The reference's own G1 oracle (Aleo.Address.bls12_377_msm,
src/reference/reference.ts:24,57) is a third-party WASM package that is not vendored, and the
inputs of its 2^16..2^20 known answers (src/test-data/testCases.ts:14-26) are not in the
tree, so the golden vectors here come from tests/pyref.py -- plain affine/Jacobian formulas
on Python ints, independent of both the C oracle and the HIP engine.  Run from the repo root:

    python3 oracle/gen_golden.py

Each case is tests/golden/<name>.bin = points (96 n bytes) || scalars (32 n) || expected (96),
listed in tests/golden/manifest.json.  Wire format: src/ui/AllBenchmarks.tsx:57-68.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import pyref as R  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")


def random_points(seed, n):
    """P_i = [a_i]G, a_i = i-th SplitMix64(seed) output."""
    g = R.splitmix64(seed)
    return [R.mul(R.G, next(g) or 1) for _ in range(n)]


def case_cuzk_test():
    """The 16-point shape of src/submission/miscellaneous/tests/cuzk.test.ts:26-63: the fixed base
    point, scalars i * 1111...1 mod p_edwards (the test pushes a second point per iteration but
    only the first 16 are paired with scalars; here exactly 16 pairs: pt, [1]pt .. as listed)."""
    p_ed = 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001
    v = int("1" * 76)
    pts, ks = [], []
    seq = []
    for i in range(16):
        seq.append(R.FIXED_BASE)
        seq.append(R.mul(R.FIXED_BASE, i + 1))
    for i in range(16):
        pts.append(seq[i])
        ks.append((i * v) % p_ed)
    return pts, ks


def edge_scalars():
    r = R.R_ORDER
    ks = [0, 1, 2, r - 1, r - 2, 1 << 15, (1 << 15) - 1, (1 << 15) + 1, (1 << 16) - 1, 1 << 16,
          0x8000800080008000800080008000800080008000800080008000800080008000 % r,
          0x7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF % r,
          0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF % r,
          0x12AB << 240, (0x12AB << 240) | 0x8000, (1 << 252), (1 << 239) | (1 << 15), 0x80000000, 0xFFFF0000,
          0x0001000100010001000100010001000100010001000100010001000100010001]
    return ks


def cases():
    out = {}
    out["g1_n1_gen"] = ([R.G], [1])
    out["g1_n1_big"] = ([R.G], [R.R_ORDER - 1])
    pts = random_points(0x377, 2)
    out["g1_n2_cancel"] = ([pts[0], R.neg(pts[0])], [12345678901234567890, 12345678901234567890])
    out["g1_n2_zero_scalars"] = (pts, [0, 0])
    out["g1_n16_cuzk_test"] = case_cuzk_test()
    out["g1_n33_random"] = (random_points(0x377 + 33, 33), R.rand_scalars(0x5CA1A5 + 33, 33))
    ks = edge_scalars()
    out["g1_n%d_edge_scalars" % len(ks)] = (random_points(0x377 + 64, len(ks)), ks)
    out["g1_n64_same_point"] = ([R.FIXED_BASE] * 64, R.rand_scalars(0x5CA1A5 + 64, 64))
    out["g1_n64_same_scalar"] = (random_points(0x377 + 65, 64), [R.rand_scalars(7, 1)[0]] * 64)
    p3 = random_points(0x377 + 66, 3)
    out["g1_n48_repeats_and_negs"] = ((p3 + [R.neg(q) for q in p3]) * 8, [0x123456789ABCDEF0123456789ABCDEF] * 24 + R.rand_scalars(99, 24))
    out["g1_n1024_random"] = (random_points(0x377, 1024), R.rand_scalars(0x5CA1A5, 1024))
    return out


def ed_random_points(seed, n):
    g = R.splitmix64(seed)
    return [R.ed_mul(R.ED_G, next(g) or 1) for _ in range(n)]


def ed_cases():
    """Edwards-BLS12 cases (64-byte points).  Scalars below the subgroup order
    (src/reference/params/AleoConstants.ts:5)."""
    out = {}
    out["ed_n1_gen"] = ([R.ED_G], [1])
    p = ed_random_points(0xED, 2)
    out["ed_n2_cancel"] = ([p[0], R.ed_neg(p[0])], [987654321987654321, 987654321987654321])
    out["ed_n24_random"] = (ed_random_points(0xED + 24, 24), R.rand_scalars(0x5CA1A5 + 24, 24, R.ED_SUBGROUP))
    ks = [0, 1, R.ED_SUBGROUP - 1, 1 << 15, (1 << 15) - 1, (1 << 16) - 1, 1 << 16,
          0x8000800080008000800080008000800080008000800080008000800080008000 % R.ED_SUBGROUP,
          0x7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF % R.ED_SUBGROUP, 1 << 250]
    out["ed_n%d_edge_scalars" % len(ks)] = (ed_random_points(0xED + 10, len(ks)), ks)
    out["ed_n32_same_point"] = ([p[1]] * 32, R.rand_scalars(0x5CA1A5 + 32, 32, R.ED_SUBGROUP))
    out["ed_n256_random"] = (ed_random_points(0xED, 256), R.rand_scalars(0x5CA1A5, 256, R.ED_SUBGROUP))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    manifest = {}
    for name, (pts, ks) in ed_cases().items():
        assert len(pts) == len(ks) and all(R.ed_on_curve(p) for p in pts)
        exp = R.ed_msm_naive(pts, ks)
        blob = R.ed_encode_points(pts) + R.encode_scalars(ks) + R.ed_encode_result(exp)
        with open(os.path.join(OUT, name + ".bin"), "wb") as f:
            f.write(blob)
        manifest[name] = {"n": len(pts), "expected_identity": exp == R.ED_ID}
        print(name, len(pts), hex(exp[0])[:18])
    for name, (pts, ks) in cases().items():
        assert len(pts) == len(ks)
        assert all(R.on_curve(p) for p in pts)
        exp = R.msm_naive(pts, ks)
        blob = R.encode_points(pts) + R.encode_scalars(ks) + R.encode_result(exp)
        with open(os.path.join(OUT, name + ".bin"), "wb") as f:
            f.write(blob)
        manifest[name] = {"n": len(pts), "expected_identity": exp is None}
        print(name, len(pts), "identity" if exp is None else hex(exp[0])[:18])
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
