"""Pure-Python big-integer BLS12-377 G1 / Edwards-BLS12 arithmetic (test infrastructure).

Independent of both the HIP engine and the C oracle: plain affine/Jacobian formulas on
Python ints.  Used to mint tests/golden/ (oracle/gen_golden.py) and to cross-check the
oracle at small sizes.  Constants are the reference's:
  p   src/submission/implementation/cuzk/bls12_377.ts:10-12
  r   src/reference/params/AleoConstants.ts:8        (scalar field; also the Edwards base field)
  G   src/submission/implementation/cuzk/bls12_377.ts:21-29
  Edwards a, d, generator  src/reference/params/AleoConstants.ts:3-4, src/reference/utils/FieldMath.ts:108-109
Wire format (src/ui/AllBenchmarks.tsx:57-68, src/reference/webgpu/utils.ts:63-72):
  points  = n x (x as 48-byte LE || y as 48-byte LE);  scalars = n x 32-byte LE.
"""
P = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
R_ORDER = 8444461749428370424248824938781546531375899335154063827935233455917409239041
GX = 81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695
GY = 241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030
G = (GX, GY)
# src/ui/AllBenchmarks.tsx:84-85 == src/submission/miscellaneous/tests/cuzk.test.ts:16-21
FIXED_BASE = (
    111871295567327857271108656266735188604298176728428155068227918632083036401841336689521497731900230387779623820740,
    76860045326390600098227152997486448974650822224305058012700629806287380625419427989664237630603922765089083164740,
)
MASK64 = (1 << 64) - 1


def splitmix64(seed):
    """SplitMix64 stream (BASELINE.md section 3 names it for the synthetic inputs)."""
    x = seed & MASK64
    while True:
        x = (x + 0x9E3779B97F4A7C15) & MASK64
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        yield z ^ (z >> 31)


def rand_scalars(seed, n, modulus=R_ORDER):
    g = splitmix64(seed)
    out = []
    for _ in range(n):
        v = 0
        for k in range(4):
            v |= next(g) << (64 * k)
        out.append(v % modulus)
    return out


# ---- short Weierstrass y^2 = x^3 + 1 over Fp; None is the identity ----
def on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - 1) % P == 0


def neg(pt):
    if pt is None:
        return None
    return (pt[0], (-pt[1]) % P)


def add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def _jac_dbl(X, Y, Z):
    if Y == 0 or Z == 0:
        return (1, 1, 0)
    A = X * X % P
    B = Y * Y % P
    C = B * B % P
    D = 2 * ((X + B) * (X + B) - A - C) % P
    E = 3 * A % P
    X3 = (E * E - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def _jac_madd(X1, Y1, Z1, x2, y2):
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % P
    U2 = x2 * Z1Z1 % P
    S2 = y2 * Z1 * Z1Z1 % P
    H = (U2 - X1) % P
    r = (S2 - Y1) % P
    if H == 0:
        if r == 0:
            return _jac_dbl(X1, Y1, Z1)
        return (1, 1, 0)
    HH = H * H % P
    HHH = H * HH % P
    V = X1 * HH % P
    X3 = (r * r - HHH - 2 * V) % P
    Y3 = (r * (V - X3) - Y1 * HHH) % P
    Z3 = Z1 * H % P
    return (X3, Y3, Z3)


def mul(pt, k):
    """[k]pt by left-to-right double-and-add in Jacobian coordinates."""
    if pt is None or k == 0:
        return None
    if k < 0:
        return mul(neg(pt), -k)
    x, y = pt
    X, Y, Z = 1, 1, 0
    for bit in bin(k)[2:]:
        X, Y, Z = _jac_dbl(X, Y, Z)
        if bit == "1":
            X, Y, Z = _jac_madd(X, Y, Z, x, y)
    if Z == 0:
        return None
    zi = pow(Z, -1, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def msm_naive(points, scalars):
    acc = None
    for pt, k in zip(points, scalars):
        acc = add(acc, mul(pt, k))
    return acc


# ---- wire format ----
def encode_points(points):
    return b"".join(x.to_bytes(48, "little") + y.to_bytes(48, "little") for x, y in points)


def decode_points(buf):
    n = len(buf) // 96
    return [
        (int.from_bytes(buf[96 * i : 96 * i + 48], "little"), int.from_bytes(buf[96 * i + 48 : 96 * i + 96], "little"))
        for i in range(n)
    ]


def encode_scalars(scalars):
    return b"".join(int(s).to_bytes(32, "little") for s in scalars)


def decode_scalars(buf):
    return [int.from_bytes(buf[32 * i : 32 * i + 32], "little") for i in range(len(buf) // 32)]


def encode_result(pt):
    """compute_msm's return value as x||y, 48-byte LE each; the identity is {x:0, y:1}
    (src/submission/submission.ts:93-95)."""
    if pt is None:
        return (0).to_bytes(48, "little") + (1).to_bytes(48, "little")
    return pt[0].to_bytes(48, "little") + pt[1].to_bytes(48, "little")


def decode_result(buf):
    x = int.from_bytes(buf[:48], "little")
    y = int.from_bytes(buf[48:96], "little")
    if x == 0 and y == 1:
        return None
    return (x, y)


# ---- twisted Edwards a*x^2 + y^2 = 1 + d*x^2*y^2 over Fq (q = R_ORDER), a = -1, d = 3021 ----
Q = R_ORDER
ED_A = Q - 1
ED_D = 3021
ED_G = (
    1540945439182663264862696551825005342995406165131907382295858612069623286213,
    8003546896475222703853313610036801932325312921786952001586936882361378122196,
)
ED_SUBGROUP = 2111115437357092606062206234695386632838870926408408195193685246394721360383
ED_ID = (0, 1)


def ed_on_curve(pt):
    x, y = pt
    return (ED_A * x * x + y * y - 1 - ED_D * x * x * y * y) % Q == 0


def ed_add(a, b):
    x1, y1 = a
    x2, y2 = b
    t = ED_D * x1 * x2 * y1 * y2 % Q
    x3 = (x1 * y2 + y1 * x2) * pow(1 + t, -1, Q) % Q
    y3 = (y1 * y2 - ED_A * x1 * x2) * pow(1 - t, -1, Q) % Q
    return (x3, y3)


def ed_neg(a):
    return ((-a[0]) % Q, a[1])


def ed_mul(pt, k):
    acc = ED_ID
    for bit in bin(k)[2:] if k else "":
        acc = ed_add(acc, acc)
        if bit == "1":
            acc = ed_add(acc, pt)
    return acc


def ed_msm_naive(points, scalars):
    acc = ED_ID
    for pt, k in zip(points, scalars):
        acc = ed_add(acc, ed_mul(pt, k))
    return acc


def ed_encode_points(points):
    return b"".join(x.to_bytes(32, "little") + y.to_bytes(32, "little") for x, y in points)


def ed_encode_result(pt):
    return pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little")


def _sqrt_mod_q(a):
    """Tonelli-Shanks in Fq (q - 1 = 2^47 * odd)."""
    a %= Q
    if a == 0:
        return 0
    if pow(a, (Q - 1) // 2, Q) != 1:
        return None
    s, t = 0, Q - 1
    while t % 2 == 0:
        s, t = s + 1, t // 2
    z = 2
    while pow(z, (Q - 1) // 2, Q) != Q - 1:
        z += 1
    m, c, tt, r = s, pow(z, t, Q), pow(a, t, Q), pow(a, (t + 1) // 2, Q)
    while tt != 1:
        i, x = 0, tt
        while x != 1:
            x = x * x % Q
            i += 1
        b = pow(c, 1 << (m - i - 1), Q)
        m, c = i, b * b % Q
        tt, r = tt * c % Q, r * b % Q
    return r


def ed_point_from_x(x):
    """getPointFromX (src/reference/utils/FieldMath.ts:31-55): y^2 = (a x^2 - 1)/(d x^2 - 1), the root
    that puts the point in the prime-order subgroup."""
    xx = x * x % Q
    y2 = (ED_A * xx - 1) * pow(ED_D * xx - 1, -1, Q) % Q
    y = _sqrt_mod_q(y2)
    assert y is not None
    for cand in (y, (-y) % Q):
        if ed_mul((x, cand), ED_SUBGROUP) == ED_ID:
            return (x, cand)
    raise ValueError("no subgroup point with this x")
