"""Pin the Edwards oracle (oracle/ed_oracle.c) and tests/pyref.py with the reference's own Edwards
vectors.  CPU only.
  src/reference/utils/FieldMath.test.ts:5-62    multiply (5 scalar-multiplication vectors)
  src/reference/utils/FieldMath.test.ts:64-98   getPointFromX (6 decompression vectors)
  src/reference/utils/wasmFunctions.test.ts:24-36  addGroups (x-only "group" encoding)
  src/reference/utils/wasmFunctions.test.ts:38-49  groupScalarMul
  src/reference/params/AleoConstants.ts:2-5     q, a, d, subgroup order
"""
import ctypes

import pyref as R
import util

from ed_vectors import ADD_GROUPS, GROUP_SCALAR_MUL, MULTIPLY, POINT_FROM_X  # the reference-held vectors (data)


def o_mul(oracle, pt, k):
    out = ctypes.create_string_buffer(64)
    assert oracle.oracle_ed_scalar_mul(R.ed_encode_points([pt]), int(k).to_bytes(32, "little"), 32, ctypes.addressof(out)) == 0
    return (int.from_bytes(out.raw[:32], "little"), int.from_bytes(out.raw[32:], "little"))


def o_add(oracle, a, b):
    out = ctypes.create_string_buffer(64)
    assert oracle.oracle_ed_add_affine(R.ed_encode_points([a]), R.ed_encode_points([b]), ctypes.addressof(out)) == 0
    return (int.from_bytes(out.raw[:32], "little"), int.from_bytes(out.raw[32:], "little"))


def test_curve_parameters(oracle):
    assert R.Q == 8444461749428370424248824938781546531375899335154063827935233455917409239041
    assert R.ED_A == R.Q - 1 and R.ED_D == 3021
    assert R.ed_on_curve(R.ED_G) and oracle.oracle_ed_on_curve(R.ed_encode_points([R.ED_G])) == 1
    assert R.ed_mul(R.ED_G, R.ED_SUBGROUP) == R.ED_ID
    assert pow(R.ED_D, (R.Q - 1) // 2, R.Q) == R.Q - 1  # d non-square, a = -1 square: complete addition law
    assert pow(R.ED_A, (R.Q - 1) // 2, R.Q) == 1


def test_fieldmath_multiply_vectors(oracle):
    for pt, k, exp in MULTIPLY:
        assert R.ed_on_curve(pt)
        assert R.ed_mul(pt, k) == exp
        assert o_mul(oracle, pt, k) == exp


def test_fieldmath_point_from_x_vectors():
    for x, y in POINT_FROM_X:
        assert R.ed_point_from_x(x) == (x, y)


def test_wasm_add_groups_vectors(oracle):
    for x1, x2, x3 in ADD_GROUPS:
        a, b = R.ed_point_from_x(x1), R.ed_point_from_x(x2)
        assert R.ed_add(a, b)[0] == x3
        assert o_add(oracle, a, b) == R.ed_add(a, b)


def test_wasm_group_scalar_mul_vectors(oracle):
    for x, k, xr in GROUP_SCALAR_MUL:
        pt = R.ed_point_from_x(x)
        assert R.ed_mul(pt, k)[0] == xr
        assert o_mul(oracle, pt, k) == R.ed_mul(pt, k)


def test_golden_vectors(oracle, golden):
    names = [n for n in golden if n.startswith("ed_")]
    assert len(names) >= 5
    for name in names:
        case = golden[name]
        for fn in ("oracle_ed_msm", "oracle_ed_msm_naive"):
            assert util.oracle_ed_msm(oracle, case["points"], case["scalars"], fn) == case["expected"], (name, fn)
        out = ctypes.create_string_buffer(64)
        assert oracle.oracle_ed_msm_params(case["points"], case["scalars"], case["n"], 16, 256, ctypes.addressof(out)) == 0
        assert out.raw == case["expected"], name


def test_gen_points(oracle):
    n = 130
    buf = util.oracle_ed_gen_points(oracle, n, 11, 13)
    for i in (0, 1, 2, 3, 64, 65, 129):
        x = int.from_bytes(buf[64 * i : 64 * i + 32], "little")
        y = int.from_bytes(buf[64 * i + 32 : 64 * i + 64], "little")
        assert (x, y) == R.ed_mul(R.ED_G, 11 + 13 * i)
