#!/usr/bin/env python3
"""Worst-case bounds of the lazy G1 formulas (csrc/g1_xyzz.hpp madd_lz / add_lz, msm377.hip g1_add_quad).

Every field value is tracked as (limb upper bounds[13], value upper bound, value lower bound); the script replays
the formulas on those intervals and asserts, for every product, that no 64-bit column accumulator of
csrc/field29.hpp (mul_lz, sqr_lz, mul_add_mul_lz) can overflow, for every add_kp_sub that no limb can go negative,
and that the results satisfy the storage invariant the next addition assumes:

    X: N-form, value < 5p + 2^354        Y, ZZ, ZZZ: N-form, value < p + 2^354

Run by tests/test_lazy_bounds.py; exits non-zero on any violation.
"""
import os
import re
import sys

P_FP = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
Q_FQ = 8444461749428370424248824938781546531375899335154063827935233455917409239041
LB = 29
BETA = 1 << LB
MASK = BETA - 1
# the field under test (use_field): modulus, limbs, reduction steps, slack of a lazy product above the modulus
P, N, RS, E, MOD, K = None, None, None, None, None, None


def load_consts(struct, end, n, rs):
    here = os.path.dirname(os.path.abspath(__file__))
    text = open(os.path.join(here, "..", "webgpu-msm-bls12-377_amd", "csrc", "consts_gen.hpp")).read()
    body = text[text.index("struct " + struct) : text.index(end)]
    out = {}
    for name in ("KP2", "KP6", "KP4W3", "MOD", "MOD2", "MOD4"):
        m = re.search(r"uint32_t %s\[%d\] = \{([^}]*)\}" % (name, n), body)
        out[name] = [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]
    assert "RS = %d" % rs in body
    return out


def use_field(which):
    global P, N, RS, E, MOD, K
    if which == "Fp":
        P, N, RS, E = P_FP, 13, 14, 1 << 354
        K = load_consts("G1Consts", "struct GlvConsts", N, RS)
    else:  # Fq: R = 2^261 = 438 q; a lazy product of values below 5q exceeds q by less than q / 16
        P, N, RS, E = Q_FQ, 9, 9, Q_FQ // 16
        K = load_consts("EdConsts", "struct G1Consts64", N, RS)
    MOD = [(P >> (LB * j)) & MASK for j in range(N)]
    assert K["MOD"] == MOD
    for name, k in (("KP2", 2), ("KP6", 6), ("KP4W3", 4), ("MOD2", 2), ("MOD4", 4)):
        assert sum(x << (LB * j) for j, x in enumerate(K[name])) == k * P, name


class V:
    """Interval model of one 13-limb value."""

    def __init__(self, limbs, hi, lo=0, name=""):
        self.limbs, self.hi, self.lo, self.name = list(limbs), hi, lo, name
        assert all(0 <= x < (1 << 32) for x in self.limbs), (name, [hex(x) for x in self.limbs])

    @property
    def normalised(self):
        return all(x <= MASK for x in self.limbs[:-1])


def nform(hi, lo=0, name=""):
    """Carry-normalised value below hi: limbs 0..11 <= 2^29 - 1, top limb <= (hi - 1) >> 348."""
    return V([MASK] * (N - 1) + [(hi - 1) >> (LB * (N - 1))], hi, lo, name)


def m1(name=""):
    return nform(P + E, 0, name)


def check_columns(pairs, what):
    """pairs: list of (a, b) products accumulated into the same columns, plus RS reduction rows."""
    worst = 0
    for k in range(2 * N):
        s = 0
        for a, b in pairs:
            for i in range(N):
                j = k - i
                if 0 <= j < N:
                    s += a.limbs[i] * b.limbs[j]
        for i in range(RS):  # q_i * MOD[j], q_i <= 2^29 - 1; j = 0 is the implicit + q
            j = k - i
            if 0 <= j < N:
                s += MASK * MOD[j]
        s += 1 << 40  # carry from the column below (a column is < 2^64, >> 29 leaves < 2^35)
        worst = max(worst, s)
    assert worst < (1 << 64), "%s: column sum 2^%.2f" % (what, __import__("math").log2(worst))
    return worst


def product_value(pairs):
    return sum(a.hi * b.hi for a, b in pairs) // (1 << (LB * RS)) + 1 + P


def mul_lz(a, b, name):
    check_columns([(a, b)], name)
    hi = product_value([(a, b)])
    assert hi <= P + E, (name, hi / P)
    return m1(name)


def sqr_lz(a, name):
    assert a.normalised, name  # 2 * a_i must fit 32 bits for i <= 11
    return mul_lz(a, a, name)


def mul_add_mul_lz(a, b, e, d, name):
    check_columns([(a, b), (e, d)], name)
    hi = product_value([(a, b), (e, d)])
    assert hi <= P + E, (name, hi / P)
    return m1(name)


def add_kp_sub(a, kname, k, b, name, b2=None):
    """a + K - b (- 2 b2): limb-wise; K = k p."""
    Kl = K[kname]
    limbs = []
    for j in range(N):
        sub = b.limbs[j] + (2 * b2.limbs[j] if b2 else 0)
        assert Kl[j] >= sub, "%s: limb %d of %s (%#x) below the subtrahend bound %#x" % (name, j, kname, Kl[j], sub)
        limbs.append(a.limbs[j] + Kl[j])
    sub_hi = b.hi + (2 * b2.hi if b2 else 0)
    sub_lo = b.lo + (2 * b2.lo if b2 else 0)
    return V(limbs, a.hi + k * P - sub_lo, max(0, a.lo + k * P - sub_hi), name)


def kp_sub(kname, k, b, name):
    Kl = K[kname]
    for j in range(N):
        assert Kl[j] >= b.limbs[j], (name, j)
    return V(list(Kl), k * P - b.lo + 1, max(0, k * P - b.hi), name)


def add_lz(a, b, name):
    return V([x + y for x, y in zip(a.limbs, b.limbs)], a.hi + b.hi, a.lo + b.lo, name)


def norm(a, name):
    assert a.hi < (1 << (LB * (N - 1) + 32)), (name, a.hi / P)  # the top limb must fit 32 bits
    return nform(a.hi, a.lo, name)


def canon_ok(a, name):
    assert a.normalised and a.hi <= 8 * P, (name, a.hi / P)


def point_formula(x1, y1, zz1, zzz1, u_in, s_in, kp_p, kname_p, p_mults, tag):
    """Shared tail of madd_lz (u_in = U2, X1 = x1, ...) and add_lz (x1 = U1, y1 = S1)."""
    p = norm(add_kp_sub(u_in, kname_p, kp_p, x1, tag + " P"), tag + " P")
    r = norm(add_kp_sub(s_in, "KP2", 2, y1, tag + " R"), tag + " R")
    # P = 0 mod p <=> P in {p, .., p_mults p}: the guard (P.l[0] - 1) < p_mults must cover the whole range
    assert p.lo > 0 and p.hi <= (p_mults + 1) * P and r.lo > 0, (tag, p.lo, p.hi / P)
    canon_ok(p, tag + " canon(P)")
    canon_ok(r, tag + " canon(R)")
    pp = sqr_lz(p, tag + " PP")
    ppp = mul_lz(p, pp, tag + " PPP")
    qq = mul_lz(x1, pp, tag + " Q")
    rr = sqr_lz(r, tag + " RR")
    x3 = norm(add_kp_sub(rr, "KP4W3", 4, ppp, tag + " X3", b2=qq), tag + " X3")
    assert x3.hi <= 5 * P + E, (tag, x3.hi / P)
    d = norm(add_kp_sub(qq, "KP6", 6, x3, tag + " D"), tag + " D")
    y3 = mul_add_mul_lz(r, d, kp_sub("KP2", 2, y1, tag + " -Y1"), ppp, tag + " Y3")
    return x3, y3, pp, ppp, r, d, qq


def te_formulas(canonical):
    """te377.hpp TeLazy: every stored coordinate is a lazy product (M1), base records canonical."""

    def te_finish(a, b, c, d, tag):
        e = norm(add_kp_sub(b, "KP2", 2, a, tag + " E"), tag + " E")
        f = norm(add_kp_sub(d, "KP2", 2, c, tag + " F"), tag + " F")
        g = norm(add_lz(d, c, tag + " G"), tag + " G")
        h = add_lz(b, a, tag + " H")
        for nm, (x, y) in (("X3", (e, f)), ("Y3", (h, g)), ("T3", (h, e)), ("Z3", (f, g))):
            mul_lz(x, y, "%s %s" % (tag, nm))

    PX, PY, PT, PZ = m1("PX"), m1("PY"), m1("PT"), m1("PZ")
    a = mul_lz(add_kp_sub(PY, "KP2", 2, PX, "te madd Y-X"), canonical, "te madd A")
    b = mul_lz(add_lz(PY, PX, "te madd Y+X"), canonical, "te madd B")
    c = mul_lz(kp_sub("KP2", 2, canonical, "te madd -kt"), PT, "te madd C")
    d = mul_lz(PZ, canonical, "te madd D")
    te_finish(a, b, c, d, "te madd")
    a = mul_lz(norm(add_kp_sub(PY, "KP2", 2, PX, "te add Y1-X1"), "te add Y1-X1"), norm(add_kp_sub(PY, "KP2", 2, PX, "te add Y2-X2"), "te add Y2-X2"), "te add A")
    b = mul_lz(norm(add_lz(PY, PX, "te add Y1+X1"), "te add Y1+X1"), norm(add_lz(PY, PX, "te add Y2+X2"), "te add Y2+X2"), "te add B")
    c = mul_lz(mul_lz(PT, PT, "te add T1T2"), canonical, "te add C")
    d = mul_lz(PZ, PZ, "te add D")
    te_finish(a, b, c, add_lz(d, d, "te add 2D"), "te add")
    # madd_affine: A, B, C as in madd, D = 2 Z1 limb-wise
    a = mul_lz(add_kp_sub(PY, "KP2", 2, PX, "te amadd Y-X"), canonical, "te amadd A")
    b = mul_lz(add_lz(PY, PX, "te amadd Y+X"), canonical, "te amadd B")
    c = mul_lz(kp_sub("KP2", 2, canonical, "te amadd -kt"), PT, "te amadd C")
    te_finish(a, b, c, add_lz(PZ, PZ, "te amadd 2Z"), "te amadd")
    # canonical operations on stored (lazy) coordinates: mul() = reduce_once(mul_lz()) needs mul_lz < 2p
    mul_lz(PX, canonical, "gather X*TO64")


def csub_mod(a, name):
    """field29.hpp csub(x, MOD): x N-form; subtracts p once if x >= p."""
    assert a.normalised, name
    return nform(max(P, a.hi - P), 0, name)


def conversion_formulas():
    """kernels/convert.hpp, the batched affine conversion in lazy forms (AffWireSource::load, k_affine_up, k_affine_down)."""
    canonical = nform(P, 0, "canonical")
    zero = V([0] * N, 1, 0, "zero")
    u = add_lz(mul_lz(canonical, canonical, "conv x*SR"), canonical, "conv u")
    v = mul_lz(canonical, canonical, "conv v")
    cu = csub_mod(norm(add_lz(mul_lz(canonical, canonical, "conv x*CSR"), canonical, "conv cu"), "conv cu"), "conv cu")
    assert cu.hi <= P + E, cu.hi / P
    up = add_lz(u, canonical, "conv u+1")
    assert max(up.limbs[:-1]) < 3 * BETA and up.limbs[-1] < (1 << 31)
    z = mul_lz(up, v, "conv z")
    n1 = mul_lz(up, cu, "conv n1")
    n2 = norm(add_kp_sub(z, "KP4W3", 4, zero, "conv n2", b2=v), "conv n2")
    assert n2.lo > 0 and n2.hi <= 5 * P + E, (n2.lo, n2.hi / P)
    c = mul_lz(m1("c"), z, "conv c*z")
    node = mul_lz(c, c, "conv tree node")
    root = mul_lz(node, canonical, "conv root*TO64")  # then reduce_once: needs < 2p
    assert root.hi <= 2 * P
    inv = mul_lz(canonical, node, "conv tree down")  # the host's inverse is canonical; further down N x N
    inv = mul_lz(inv, node, "conv tree down 2")
    zi = mul_lz(inv, c, "conv zi")
    inv = mul_lz(inv, z, "conv inv*Z")
    x = mul_lz(n1, zi, "conv x")  # then reduce_once -> canonical
    y = mul_lz(n2, zi, "conv y")
    assert x.hi <= 2 * P and y.hi <= 2 * P
    kt = mul_lz(mul_lz(canonical, canonical, "conv xy"), canonical, "conv 2dxy")
    # the record's kt is now a lazy product, its y -+ x stay canonical: replay the affine mixed addition with it
    PX, PY, PT, PZ = m1("PX"), m1("PY"), m1("PT"), m1("PZ")
    mul_lz(kp_sub("KP2", 2, kt, "conv-rec -kt"), PT, "conv-rec C neg")
    mul_lz(kt, PT, "conv-rec C")
    mul_lz(kt, canonical, "conv-rec first: kt / d")  # from_base_affine: mul(kt, TE_INV_D), reduce_once needs < 2p


def main():
    use_field("Fq")  # Edwards-BLS12 buckets (EdDev): same law over the 9-limb field
    te_formulas(nform(P, 0, "canonical"))
    use_field("Fp")
    canonical = nform(P, 0, "canonical")
    X1 = nform(5 * P + E, 0, "X1")
    Y1, ZZ1, ZZZ1 = m1("Y1"), m1("ZZ1"), m1("ZZZ1")

    # ---- madd_lz ----
    u2 = mul_lz(canonical, ZZ1, "madd U2")
    qy = kp_sub("KP2", 2, canonical, "madd -qy")  # negated base y (lazy); the plain one is canonical
    s2 = mul_lz(qy, ZZZ1, "madd S2")
    x3, y3, pp, ppp, *_ = point_formula(X1, Y1, ZZ1, ZZZ1, u2, s2, 6, "KP6", 7, "madd")
    mul_lz(ZZ1, pp, "madd ZZ3")
    mul_lz(ZZZ1, ppp, "madd ZZZ3")

    # ---- add_lz (thread) ----
    X2 = nform(5 * P + E, 0, "X2")
    u1 = mul_lz(X1, ZZ1, "add U1")
    u2 = mul_lz(X2, ZZ1, "add U2")
    s1 = mul_lz(Y1, ZZZ1, "add S1")
    s2 = mul_lz(Y1, ZZZ1, "add S2")
    x3, y3, pp, ppp, r, d, qq = point_formula(u1, s1, ZZ1, ZZZ1, u2, s2, 2, "KP2", 3, "add")
    mul_lz(mul_lz(ZZ1, ZZ1, "add ZZ1ZZ2"), pp, "add ZZ3")
    mul_lz(mul_lz(ZZZ1, ZZZ1, "add ZZZ1ZZZ2"), ppp, "add ZZZ3")

    # ---- g1_add_quad: round 2 squares P and R through mul_lz, round 4 forms Y3 from two separate products ----
    mul_lz(r, r, "quad RR")
    a = mul_lz(r, d, "quad R*D")
    b = mul_lz(s1, ppp, "quad S1*PPP")
    y = norm(add_kp_sub(a, "KP2", 2, b, "quad Y3"), "quad Y3")
    canon_ok(y, "quad canon(Y3)")

    te_formulas(canonical)
    conversion_formulas()

    # canonical operations on stored (lazy) coordinates: mul() = reduce_once(mul_lz()) needs mul_lz < 2p
    mul_lz(X1, canonical, "gather X*TO64")
    print("lazy bounds OK")


if __name__ == "__main__":
    main()
