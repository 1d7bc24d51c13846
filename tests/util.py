"""Shared helpers for the tests (TEST INFRASTRUCTURE): oracle loading, golden loading,
device-format encoders."""
import ctypes
import json
import os
import sys
import subprocess

import pyref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "libmsm_oracle.so")

LB = 29
LMASK = (1 << LB) - 1
R29 = 1 << (LB * 14)  # device Montgomery radix for Fp (14 reduction steps over 13 limbs, csrc/field29.hpp)


def load_oracle():
    """ctypes handle of the C oracle; built on demand (gcc) when the .so is missing or stale."""
    src = [os.path.join(ROOT, "oracle", f) for f in ("msm_oracle.c", "ed_oracle.c", "Makefile")]
    stale = not os.path.exists(ORACLE_SO) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in src if os.path.exists(s))
    if stale:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(ORACLE_SO)
    lib.oracle_g1_msm.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p]
    lib.oracle_g1_msm_naive.argtypes = lib.oracle_g1_msm.argtypes
    lib.oracle_g1_msm_params.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    lib.oracle_g1_smvp_window.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    lib.oracle_decompose_scalars_signed.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p]
    lib.oracle_cpu_transpose.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    lib.oracle_cpu_transpose.restype = None
    lib.oracle_g1_horner.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    lib.oracle_g1_scalar_mul.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p]
    lib.oracle_g1_add_affine.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p]
    lib.oracle_g1_proj_to_affine.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
    lib.oracle_g1_on_curve.argtypes = [ctypes.c_char_p]
    lib.oracle_g1_gen_points_arith.argtypes = [ctypes.c_uint64, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p]
    lib.oracle_g1_generator.argtypes = [ctypes.c_void_p]
    lib.oracle_g1_generator.restype = None
    for fn in ("oracle_ed_msm", "oracle_ed_msm_naive"):
        getattr(lib, fn).argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p]
    lib.oracle_ed_msm_params.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    lib.oracle_ed_scalar_mul.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p]
    lib.oracle_ed_add_affine.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p]
    lib.oracle_ed_on_curve.argtypes = [ctypes.c_char_p]
    lib.oracle_ed_gen_points_arith.argtypes = [ctypes.c_uint64, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p]
    lib.oracle_fp_ops.argtypes = [ctypes.c_char_p] * 2 + [ctypes.c_void_p] * 3
    lib.oracle_fp_ops.restype = None
    lib.oracle_fp_mont_constants.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.oracle_fp_mont_constants.restype = None
    return lib


def oracle_msm(lib, points: bytes, scalars: bytes, fn="oracle_g1_msm") -> bytes:
    n = len(scalars) // 32
    out = ctypes.create_string_buffer(96)
    rc = getattr(lib, fn)(points, scalars, n, ctypes.addressof(out))
    assert rc == 0, rc
    return out.raw


def oracle_msm_params(lib, points, scalars, c, T, want_windows=False):
    n = len(scalars) // 32
    W = (256 + c - 1) // c
    out = ctypes.create_string_buffer(96)
    ws = ctypes.create_string_buffer(96 * W) if want_windows else None
    rc = lib.oracle_g1_msm_params(points, scalars, n, c, T, ctypes.addressof(out), ctypes.addressof(ws) if ws else None)
    assert rc == 0, rc
    return (out.raw, ws.raw) if want_windows else out.raw


def oracle_gen_points(lib, n, a0, delta) -> bytes:
    out = ctypes.create_string_buffer(96 * n)
    rc = lib.oracle_g1_gen_points_arith(n, int(a0).to_bytes(32, "little"), int(delta).to_bytes(32, "little"), ctypes.addressof(out))
    assert rc == 0, rc
    return out.raw


def oracle_ed_msm(lib, points: bytes, scalars: bytes, fn="oracle_ed_msm") -> bytes:
    n = len(scalars) // 32
    out = ctypes.create_string_buffer(64)
    rc = getattr(lib, fn)(points, scalars, n, ctypes.addressof(out))
    assert rc == 0, rc
    return out.raw


def oracle_ed_gen_points(lib, n, a0, delta) -> bytes:
    out = ctypes.create_string_buffer(64 * n)
    rc = lib.oracle_ed_gen_points_arith(n, R.ed_encode_points([R.ED_G]), int(a0).to_bytes(32, "little"), int(delta).to_bytes(32, "little"), ctypes.addressof(out))
    assert rc == 0, rc
    return out.raw


def load_golden():
    with open(os.path.join(GOLDEN_DIR, "manifest.json")) as f:
        manifest = json.load(f)
    cases = {}
    for name, meta in manifest.items():
        n = meta["n"]
        psz = 64 if name.startswith("ed_") else 96  # Edwards cases carry 64-byte points / results
        with open(os.path.join(GOLDEN_DIR, name + ".bin"), "rb") as f:
            blob = f.read()
        assert len(blob) == (psz + 32) * n + psz
        cases[name] = {"n": n, "points": blob[: psz * n], "scalars": blob[psz * n : (psz + 32) * n], "expected": blob[(psz + 32) * n :]}
    return cases


# ---- device formats ----
def to_limbs29_mont(v: int):
    """canonical residue -> 13 x 29-bit limbs of v * 2^406 mod p (csrc/field29.hpp format)."""
    m = (v * R29) % R.P
    return [(m >> (LB * i)) & LMASK for i in range(13)]


def from_limbs29_mont(limbs) -> int:
    m = sum(int(x) << (LB * i) for i, x in enumerate(limbs))
    return (m * pow(R29, -1, R.P)) % R.P


def xyzz_words_from_affine(pt):
    """Affine point (or None) -> 52 device words X, Y, ZZ, ZZZ with ZZ = ZZZ = 1."""
    if pt is None:
        return to_limbs29_mont(0) + to_limbs29_mont(1) + [0] * 26
    return to_limbs29_mont(pt[0]) + to_limbs29_mont(pt[1]) + to_limbs29_mont(1) + to_limbs29_mont(1)


def affine_from_xyzz_words(words):
    """52 device words -> affine point via Python ints (None for the identity)."""
    X = from_limbs29_mont(words[0:13])
    Y = from_limbs29_mont(words[13:26])
    ZZ = from_limbs29_mont(words[26:39])
    ZZZ = from_limbs29_mont(words[39:52])
    if ZZ == 0:
        return None
    assert pow(ZZ, 3, R.P) == pow(ZZZ, 2, R.P), "XYZZ invariant ZZ^3 = ZZZ^2 violated"
    return (X * pow(ZZ, -1, R.P) % R.P, Y * pow(ZZZ, -1, R.P) % R.P)


R64 = 1 << 384  # Montgomery radix of the partial records (host-tail format)


def _words12(v):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(12)]


def record_point_words(pt, z=1):
    """One point of a window partial record (include/msm377.h): X, Y, ZZ, ZZZ as 12 u32 words each,
    Montgomery form with radix 2^384; None (the identity) has ZZ = ZZZ = 0."""
    if pt is None:
        return _words12(0) + _words12(R64 % R.P) + [0] * 24
    zz, zzz = z * z % R.P, z * z * z % R.P
    return (_words12(pt[0] * zz * R64 % R.P) + _words12(pt[1] * zzz * R64 % R.P) + _words12(zz * R64 % R.P) + _words12(zzz * R64 % R.P))


_TE = None


def te_params():
    """Constants of the twisted Edwards form of G1 (tools/gen_consts.py te_params: s, c, d)."""
    global _TE
    if _TE is None:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import gen_consts

        _TE = gen_consts.te_params()
    return _TE


def t_prime():
    """A 2-torsion point of y^2 = x^3 + 1 OTHER than (-1, 0): (-omega, 0) with omega a primitive cube root of unity.
    The twisted Edwards model (built on (-1, 0)) sends it to a point at infinity, so P and P + T' are an exceptional
    pair of the a = -1 addition law (csrc/te377.hpp): P + (P + T') and P - (P + T') both come out with Z3 = 0."""
    g = 2
    while pow(g, (R.P - 1) // 3, R.P) == 1:
        g += 1
    omega = pow(g, (R.P - 1) // 3, R.P)
    assert omega != 1 and pow(omega, 3, R.P) == 1
    pt = ((-omega) % R.P, 0)
    assert R.on_curve(pt) and R.add(pt, pt) is None
    return pt


def te_record_point_words(pt, z=1, tag=False):
    """Weierstrass affine point (or None) -> one twisted Edwards partial-record point (X, Y, T, Z as 12 u32 words each,
    radix-2^384 Montgomery form), scaled by a projective factor z; tag sets the record's coordinate-system bit."""
    te = te_params()
    if pt is None:
        xe, ye = 0, 1
    elif pt == (R.P - 1, 0):
        xe, ye = 0, R.P - 1
    else:
        u, v = te["s"] * (pt[0] + 1) % R.P, te["s"] * pt[1] % R.P
        xe, ye = te["c"] * u * pow(v, -1, R.P) % R.P, (u - 1) * pow(u + 1, -1, R.P) % R.P
    X, Y, T, Z = xe * z % R.P, ye * z % R.P, xe * ye * z % R.P, z % R.P
    w = _words12(X * R64 % R.P) + _words12(Y * R64 % R.P) + _words12(T * R64 % R.P) + _words12(Z * R64 % R.P)
    if tag:
        w[11] |= 0x80000000
    return w


def affine_from_te_record_words(words):
    """A twisted Edwards partial-record point (X, Y, T, Z; csrc/te377.hpp) -> Weierstrass affine point via Python ints."""
    ri = pow(R64, -1, R.P)
    w = [int(x) for x in words]
    w[11] &= 0x7FFFFFFF  # the record's coordinate-system tag (fp64_host.hpp TE_RECORD_TAG)
    X, Y, T, Z = (sum(w[12 * c + i] << (32 * i) for i in range(12)) * ri % R.P for c in range(4))
    assert Z != 0 and (X * Y - T * Z) % R.P == 0, "extended-coordinate invariant T Z = X Y violated"
    te = te_params()
    zi = pow(Z, -1, R.P)
    xe, ye = X * zi % R.P, Y * zi % R.P
    assert (-xe * xe + ye * ye - 1 - te["d"] * xe * xe * ye * ye) % R.P == 0, "not on the Edwards curve"
    if xe == 0:
        return None if ye == 1 else (R.P - 1, 0)
    u = (1 + ye) * pow(1 - ye, -1, R.P) % R.P
    v = te["c"] * u * pow(xe, -1, R.P) % R.P
    si = pow(te["s"], -1, R.P)
    return ((u * si - 1) % R.P, v * si % R.P)


def affine_from_te_ext_words(words):
    """A twisted Edwards BUCKET in the device format (X, Y, T, Z: 13 x 29-bit limbs each, Montgomery radix 2^406, lazy
    residues; csrc/te377.hpp) -> Weierstrass affine point via Python ints (None for the identity)."""
    X, Y, T, Z = (from_limbs29_mont(words[13 * c : 13 * c + 13]) for c in range(4))
    assert Z != 0 and (X * Y - T * Z) % R.P == 0, "extended-coordinate invariant T Z = X Y violated"
    te = te_params()
    zi = pow(Z, -1, R.P)
    xe, ye = X * zi % R.P, Y * zi % R.P
    assert (-xe * xe + ye * ye - 1 - te["d"] * xe * xe * ye * ye) % R.P == 0, "not on the Edwards curve"
    if xe == 0:
        return None if ye == 1 else (R.P - 1, 0)
    u = (1 + ye) * pow(1 - ye, -1, R.P) % R.P
    v = te["c"] * u * pow(xe, -1, R.P) % R.P
    si = pow(te["s"], -1, R.P)
    return ((u * si - 1) % R.P, v * si % R.P)


def affine_from_record_words(words):
    """Inverse of record_point_words via Python ints (either coordinate system, told apart by the caller's window tag)."""
    ri = pow(R64, -1, R.P)
    X, Y, ZZ, ZZZ = (sum(int(w) << (32 * i) for i, w in enumerate(words[12 * c : 12 * c + 12])) * ri % R.P for c in range(4))
    if ZZ == 0:
        return None
    assert pow(ZZ, 3, R.P) == pow(ZZZ, 2, R.P), "XYZZ invariant ZZ^3 = ZZZ^2 violated"
    return (X * pow(ZZ, -1, R.P) % R.P, Y * pow(ZZZ, -1, R.P) % R.P)


def partial_record_from_window_sum(pt) -> bytes:
    """A window partial record (include/msm377.h) whose point 0 is the window sum and whose 15
    bit-plane points are the identity."""
    import struct

    words = record_point_words(pt) + record_point_words(None) * 15
    return struct.pack("<%dI" % len(words), *words)
