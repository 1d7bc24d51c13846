#!/usr/bin/env python3
"""Emit csrc/consts_gen.hpp: field/curve constants in the 29-bit-limb device format.

Sources of the numbers (reference, read as text):
  p  (BLS12-377 base field)   src/submission/implementation/cuzk/bls12_377.ts:10-12
  r  (scalar field = Edwards base field q)  src/reference/params/AleoConstants.ts:2,8
  G1 generator                src/submission/implementation/cuzk/bls12_377.ts:21-29
  Edwards a = -1, d = 3021    src/reference/params/AleoConstants.ts:3-4
  Edwards generator           src/reference/utils/FieldMath.ts:108-109

Output: two structs of static constexpr tables, G1Consts (Fp) and EdConsts (Fq).
Device format: little-endian 29-bit limbs in u32 words.  Fp: 13 limbs, Montgomery
radix R = 2^406 (14 reduction steps over 13-limb operands: a product of two values
below 2^380 comes out below p + 2^354, so the hot formulas need no conditional
subtractions -- see field29.hpp).  Fq: 9 limbs, Montgomery radix R = 2^261.  Both moduli
are = 1 mod 2^29, so -p^-1 mod 2^29 = 2^29 - 1 and the Montgomery quotient digit is
just (-t0) mod 2^29 (no multiplication).
"""
import os

P = 0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001
Q = 8444461749428370424248824938781546531375899335154063827935233455917409239041
GX = 81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695
GY = 241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030
ED_D = 3021
ED_GX = 1540945439182663264862696551825005342995406165131907382295858612069623286213
ED_GY = 8003546896475222703853313610036801932325312921786952001586936882361378122196
LB = 29
MASK = (1 << LB) - 1


def limbs(v, n):
    assert 0 <= v < (1 << (LB * n))
    return [(v >> (LB * i)) & MASK for i in range(n)]


def limbs_top(v, n):
    return [(v >> (LB * i)) & MASK for i in range(n - 1)] + [v >> (LB * (n - 1))]


def arr(name, v, n):
    body = ", ".join("0x%08xu" % x for x in limbs(v, n))
    return "static constexpr uint32_t %s[%d] = {%s};\n" % (name, n, body)


def redundant(v, n, w):
    """k*p written with every limb but the top one raised by w*2^29 (borrowed from the limb above), so that
    a_j + c_j - (up to w limbs b_j < 2^29) never goes negative limb by limb."""
    m = limbs_top(v, n)
    c = [m[0] + w * (1 << LB)] + [m[j] + w * (1 << LB) - w for j in range(1, n - 1)] + [m[n - 1] - w]
    assert sum(x << (LB * j) for j, x in enumerate(c)) == v and all(0 <= x < (1 << 32) for x in c)
    assert all(x >= w * MASK for x in c[:-1])
    return c


def emit(ns, mod, n, extra, rs=None, lazy=False):
    rs = rs or n
    R = 1 << (LB * rs)
    assert mod & MASK == 1
    out = "struct %s {\n" % ns
    out += "static constexpr int NL = %d;\n" % n
    out += "static constexpr int RS = %d;  // Montgomery reduction steps: R = 2^(29 RS)\n" % rs
    if lazy:
        for name, k, w in (("KP2", 2, 1), ("KP6", 6, 1), ("KP4W3", 4, 3)):
            body = ", ".join("0x%08xu" % x for x in redundant(k * mod, n, w))
            out += "static constexpr uint32_t %s[%d] = {%s};  // %d p, limbs raised by %d * 2^29\n" % (name, n, body, k, w)
        for name, k in (("MOD2", 2), ("MOD4", 4)):
            body = ", ".join("0x%08xu" % x for x in limbs_top(k * mod, n))
            out += "static constexpr uint32_t %s[%d] = {%s};  // %d p, top limb unbounded\n" % (name, n, body, k)
    out += arr("MOD", mod, n)
    out += arr("ONE", R % mod, n)          # Montgomery form of 1
    out += arr("R2", (R * R) % mod, n)     # to-Montgomery multiplier
    for k, v in extra.items():
        out += arr(k, (v * R) % mod, n)    # Montgomery form
    nw64 = (mod.bit_length() + 63) // 64
    out += arr("TO64", pow(2, 64 * nw64, mod), n)  # mul(x, TO64) turns v * 2^(29 n) into v * 2^(64 nw64): the host tail's Montgomery form
    nw = (mod.bit_length() + 31) // 32
    words = ", ".join("0x%08xu" % (((mod - 2) >> (32 * i)) & 0xffffffff) for i in range(nw))
    out += "static constexpr int PM2_NW = %d;\n" % nw
    out += "static constexpr uint32_t PM2_W[%d] = {%s};  // p - 2, LE u32 words (Fermat inversion)\n" % (nw, words)
    out += "};\n\n"
    return out


def emit64(ns, mod, n, n29, extra=None):
    """Host-tail field constants: 64-bit words, Montgomery radix 2^(64n); n29 = Montgomery reduction
    steps (radix 2^(29 n29)) of the device format of the same field."""
    R = 1 << (64 * n)

    def arr64(name, v):
        body = ", ".join("0x%016xull" % ((v >> (64 * i)) & ((1 << 64) - 1)) for i in range(n))
        return "static constexpr uint64_t %s[%d] = {%s};\n" % (name, n, body)

    out = "struct %s {\n" % ns
    out += "static constexpr int NW = %d;\n" % n
    out += arr64("MOD", mod)
    out += arr64("ONE", R % mod)
    out += arr64("R2", (R * R) % mod)
    # x*2^(29*n29) mod p -> x*2^(64n) mod p is a Montgomery product with 2^(64n - 29*n29) * R
    out += arr64("FROM29", (pow(2, 64 * n - 29 * n29, mod) * R) % mod)
    for k, v in (extra or {}).items():
        out += arr64(k, (v * R) % mod)
    out += "static constexpr uint64_t N0 = 0x%016xull;  // -p^-1 mod 2^64\n" % ((-pow(mod, -1, 1 << 64)) % (1 << 64))
    out += "};\n\n"
    return out


# GLV endomorphism of BLS12-377 G1: phi(x, y) = (BETA x, y) = [LAMBDA](x, y), LAMBDA = x0^2 - 1 for the curve
# parameter x0 (r = x0^4 - x0^2 + 1, so LAMBDA^2 + LAMBDA + 1 = 0 mod r).  k = k1 + k2 LAMBDA with
# k2 = floor(k / LAMBDA), k1 = k mod LAMBDA: both non-negative and < 2^127 for k < r.
X0 = 0x8508C00000000001
GLV_LAMBDA = X0 * X0 - 1
GLV_BETA = None  # found at generation time by _find_beta()


def _find_beta():
    """The primitive cube root of unity in Fp with (beta x, y) = [LAMBDA](x, y) on the generator."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    import pyref as R

    t = 2
    while pow(t, (P - 1) // 3, P) == 1:
        t += 1
    b = pow(t, (P - 1) // 3, P)
    target = R.mul(R.G, GLV_LAMBDA)
    for beta in (b, b * b % P):
        if (beta * GX % P, GY) == target:
            return beta
    raise AssertionError("no beta matches LAMBDA")


# ---- BLS12-377 G1 as a twisted Edwards curve with a = -1 ----
# W: y^2 = x^3 + 1 has the 2-torsion point (-1, 0).  With s = 1/sqrt(3):  u = s (x + 1), v = s y is the Montgomery
# curve s v^2 = u^3 - 3 s u^2 + u, and  xe0 = u / v, ye = (u - 1) / (u + 1)  the twisted Edwards curve
# a0 xe0^2 + ye^2 = 1 + d0 xe0^2 ye^2 with a0 = (A + 2) / B, d0 = (A - 2) / B (A = -3 s, B = s).  -a0 is a square in
# Fp, so xe = c xe0 with c = sqrt(-a0) gives  -xe^2 + ye^2 = 1 + d xe^2 ye^2,  d = -d0 / a0.  d is a square: the
# addition law has exceptional pairs, all involving points of even order -- never inside the prime-order subgroup
# (see csrc/te377.hpp for how the engine detects and survives them).
def _sqrt_p(a):
    a %= P
    assert pow(a, (P - 1) // 2, P) == 1
    q, e = P - 1, 0
    while q % 2 == 0:
        q //= 2
        e += 1
    z = 2
    while pow(z, (P - 1) // 2, P) == 1:
        z += 1
    m, c, t, r = e, pow(z, q, P), pow(a, q, P), pow(a, (q + 1) // 2, P)
    while t != 1:
        i, tt = 0, t
        while tt != 1:
            tt = tt * tt % P
            i += 1
        b = pow(c, 1 << (m - i - 1), P)
        m, c, t, r = i, b * b % P, t * b * b % P, r * b % P
    return min(r, P - r)


def te_params():
    s = pow(_sqrt_p(3), -1, P)
    A, B = (-3 * s) % P, s
    a0 = (A + 2) * pow(B, -1, P) % P
    d0 = (A - 2) * pow(B, -1, P) % P
    c = _sqrt_p(-a0)
    d = (-d0) * pow(a0, -1, P) % P
    # self-check on the generator: the image satisfies the a = -1 curve equation
    u, v = s * (GX + 1) % P, s * GY % P
    xe, ye = c * u * pow(v, -1, P) % P, (u - 1) * pow(u + 1, -1, P) % P
    assert (-xe * xe + ye * ye - 1 - d * xe * xe * ye * ye) % P == 0
    return {"s": s, "c": c, "d": d}


def emit_glv():
    assert (GLV_LAMBDA * GLV_LAMBDA + GLV_LAMBDA + 1) % Q == 0 and GLV_LAMBDA.bit_length() == 127
    mu = (1 << 384) // GLV_LAMBDA

    def words(v, n):
        return ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xFFFFFFFF) for i in range(n))

    out = "struct GlvConsts {  // k = k1 + k2 LAMBDA; Barrett quotient with MU = floor(2^384 / LAMBDA)\n"
    out += "static constexpr uint32_t LAMBDA[4] = {%s};\n" % words(GLV_LAMBDA, 4)
    out += "static constexpr uint32_t MU[9] = {%s};\n" % words(mu, 9)
    out += "};\n\n"
    return out


def main():
    global GLV_BETA
    GLV_BETA = _find_beta()
    assert (GY * GY - GX ** 3 - 1) % P == 0
    assert (-ED_GX * ED_GX + ED_GY * ED_GY - 1 - ED_D * ED_GX * ED_GX * ED_GY * ED_GY) % Q == 0
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, "..", "webgpu-msm-bls12-377_amd", "csrc", "consts_gen.hpp")
    s = "// GENERATED by tools/gen_consts.py -- do not edit.\n#pragma once\n#include <stdint.h>\n\n"
    s += "namespace msm377 {\n\n"
    te = te_params()
    s_, c_, d_ = te["s"], te["c"], te["d"]
    R29 = 1 << (LB * 14)
    g1_extra = {"GEN_X": GX, "GEN_Y": GY, "B3": 3, "BETA": GLV_BETA,
                # twisted Edwards form (csrc/te377.hpp): Montgomery forms of s, c s, 2d; and the raw (non-Montgomery)
                # multipliers s R, c s R that take a RAW wire coordinate straight to the Montgomery form of s x, c s x
                "TE_S": s_, "TE_CS": c_ * s_, "TE_2D": 2 * d_, "TE_INV_D": pow(d_, -1, P), "TE_SR": s_ * R29, "TE_CSR": c_ * s_ * R29,
                "TE_SBR": s_ * GLV_BETA * R29, "TE_CSBR": c_ * s_ * GLV_BETA * R29}
    s += emit("G1Consts", P, 13, g1_extra, rs=14, lazy=True)
    s += emit_glv()
    # Fq keeps R = 2^261 (9 steps): q is 253 bits, so the radix already leaves 8 bits of slack for the lazy forms
    s += emit("EdConsts", Q, 9, {"ED_D": ED_D, "ED_2D": 2 * ED_D, "GEN_X": ED_GX, "GEN_Y": ED_GY, "TE_2D": 2 * ED_D}, lazy=True)
    # TO29: a host-format residue (radix 2^384) times this constant is the DEVICE Montgomery form (radix 2^406) as a plain
    # integer -- the way back for the block inverses of the batched affine conversion (msm377.hip affine_convert_finish)
    s += emit64("G1Consts64", P, 6, 14, {"TE_2D": 2 * d_, "TE_INV_S": pow(s_, -1, P), "TE_C_OVER_S": c_ * pow(s_, -1, P), "TO29": 1 << (29 * 14 - 64 * 6)})
    s += emit64("EdConsts64", Q, 4, 9, {"ED_D": ED_D, "ED_2D": 2 * ED_D})
    s += "}  // namespace msm377\n"
    with open(dst, "w") as f:
        f.write(s)
    print("wrote", os.path.normpath(dst))


if __name__ == "__main__":
    main()
