// Host-only prime fields on 64-bit words (Montgomery radix 2^(64 NW)) for the CPU tail of
// the pipeline: Horner over the window / bit-plane partial sums and the single field inversion
// of the result (replaces the reference's BigInt tail with 4096 inversions,
// src/submission/submission.ts:290-321 and cuzk/bls12_377.ts:41-63).  Same static interface
// as Field<> in field29.hpp so G1T<> / EdT<> work over it.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ed_ext.hpp"
#include "field29.hpp"
#include "g1_xyzz.hpp"

namespace msm377 {

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
// a * b * 2^-384 mod p on six 64-bit limbs with the two carry chains of BMI2 / ADX (mulx + adcx / adox), result below 2^384
// and not yet reduced below p: one round = t += a * b[i]; m = t0 * n0; t += m * p; t >>= 64.  The host tail is a serial
// chain of ~2 600 field multiplications (0.12-0.18 ms per MSM at every input size), so its multiplier is worth the
// assembly: 47 vs 64 ns here against the portable unsigned __int128 form (which stays the fallback and the test oracle:
// tests/test_field29_host.py::test_adx_multiplier_matches_the_portable_one).
__attribute__((target("bmi2,adx"))) inline void mont_mul6_adx(uint64_t t[7], const uint64_t* a, const uint64_t* b, const uint64_t* p, uint64_t n0) {
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0;
  for (int i = 0; i < 6; i++) {
    uint64_t lo, hi, zero = 0;
    const uint64_t bi = b[i];
    __asm__ volatile(
        "xorl %%eax, %%eax\n\t"  // clears CF and OF
        "mulxq 0(%[a]), %[lo], %[hi]\n\t  adcxq %[lo], %[t0]\n\t  adoxq %[hi], %[t1]\n\t"
        "mulxq 8(%[a]), %[lo], %[hi]\n\t  adcxq %[lo], %[t1]\n\t  adoxq %[hi], %[t2]\n\t"
        "mulxq 16(%[a]), %[lo], %[hi]\n\t adcxq %[lo], %[t2]\n\t  adoxq %[hi], %[t3]\n\t"
        "mulxq 24(%[a]), %[lo], %[hi]\n\t adcxq %[lo], %[t3]\n\t  adoxq %[hi], %[t4]\n\t"
        "mulxq 32(%[a]), %[lo], %[hi]\n\t adcxq %[lo], %[t4]\n\t  adoxq %[hi], %[t5]\n\t"
        "mulxq 40(%[a]), %[lo], %[hi]\n\t adcxq %[lo], %[t5]\n\t  adoxq %[hi], %[t6]\n\t"
        "adcxq %[z], %[t6]\n\t"  // the CF chain ends in t6,
        "adoxq %[z], %[t7]\n\t"  // the OF chain in t7,
        "adcxq %[z], %[t7]\n\t"  // which also takes the carry out of t6
        : [t0] "+r"(t0), [t1] "+r"(t1), [t2] "+r"(t2), [t3] "+r"(t3), [t4] "+r"(t4), [t5] "+r"(t5), [t6] "+r"(t6), [t7] "+r"(t7), [lo] "=&r"(lo), [hi] "=&r"(hi)
        : [a] "r"(a), "d"(bi), [z] "r"(zero)
        : "rax", "cc", "memory");
    const uint64_t m = t0 * n0;
    __asm__ volatile(
        "xorl %%eax, %%eax\n\t"
        "mulxq 0(%[p]), %[lo], %[hi]\n\t  adcxq %[lo], %[t0]\n\t  adoxq %[hi], %[t1]\n\t"
        "mulxq 8(%[p]), %[lo], %[hi]\n\t  adcxq %[lo], %[t1]\n\t  adoxq %[hi], %[t2]\n\t"
        "mulxq 16(%[p]), %[lo], %[hi]\n\t adcxq %[lo], %[t2]\n\t  adoxq %[hi], %[t3]\n\t"
        "mulxq 24(%[p]), %[lo], %[hi]\n\t adcxq %[lo], %[t3]\n\t  adoxq %[hi], %[t4]\n\t"
        "mulxq 32(%[p]), %[lo], %[hi]\n\t adcxq %[lo], %[t4]\n\t  adoxq %[hi], %[t5]\n\t"
        "mulxq 40(%[p]), %[lo], %[hi]\n\t adcxq %[lo], %[t5]\n\t  adoxq %[hi], %[t6]\n\t"
        "adcxq %[z], %[t6]\n\t"
        "adoxq %[z], %[t7]\n\t"
        "adcxq %[z], %[t7]\n\t"
        : [t0] "+r"(t0), [t1] "+r"(t1), [t2] "+r"(t2), [t3] "+r"(t3), [t4] "+r"(t4), [t5] "+r"(t5), [t6] "+r"(t6), [t7] "+r"(t7), [lo] "=&r"(lo), [hi] "=&r"(hi)
        : [p] "r"(p), "d"(m), [z] "r"(zero)
        : "rax", "cc", "memory");
    t0 = t1, t1 = t2, t2 = t3, t3 = t4, t4 = t5, t5 = t6, t6 = t7, t7 = 0;  // t0 has become zero: down one word
  }
  t[0] = t0, t[1] = t1, t[2] = t2, t[3] = t3, t[4] = t4, t[5] = t5, t[6] = t6;
}
inline bool host_has_adx() {
  static const bool yes = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx");
  return yes;
}
#define MSM377_HAVE_ADX_MUL 1
#endif
// MSM377_NO_ADX=1 in the environment keeps the portable multiplier (A/B and tests).
inline bool& adx_enabled() {
  static bool on = [] {
#if defined(MSM377_HAVE_ADX_MUL)
    const char* e = getenv("MSM377_NO_ADX");
    return host_has_adx() && !(e && e[0] == '1');
#else
    return false;
#endif
  }();
  return on;
}

// C: 64-bit constants (G1Consts64 / EdConsts64); F29: the device-format field of the same modulus.
template <class C, class F29>
struct FieldHost64 {
  static constexpr int NW = C::NW;
  struct El {
    uint64_t v[NW];
  };
  typedef unsigned __int128 u128;

  static El zero() {
    El r;
    memset(&r, 0, sizeof r);
    return r;
  }
  static El from_const(const uint64_t (&c)[NW]) {
    El r;
    for (int i = 0; i < NW; i++) r.v[i] = c[i];
    return r;
  }
  static El one() { return from_const(C::ONE); }
  static bool is_zero(const El& a) {
    uint64_t acc = 0;
    for (int i = 0; i < NW; i++) acc |= a.v[i];
    return acc == 0;
  }
  static bool eq(const El& a, const El& b) {
    uint64_t acc = 0;
    for (int i = 0; i < NW; i++) acc |= a.v[i] ^ b.v[i];
    return acc == 0;
  }
  static El select(bool c, const El& a, const El& b) { return c ? a : b; }
  static bool geq_p(const uint64_t* a) {
    for (int i = NW - 1; i >= 0; i--) {
      if (a[i] > C::MOD[i]) return true;
      if (a[i] < C::MOD[i]) return false;
    }
    return true;
  }
  static void sub_p(uint64_t* a) {
    uint64_t borrow = 0;
    for (int i = 0; i < NW; i++) {
      u128 d = (u128)a[i] - C::MOD[i] - borrow;
      a[i] = (uint64_t)d;
      borrow = (uint64_t)(d >> 64) & 1;
    }
  }
  static El add(const El& a, const El& b) {
    El r;
    uint64_t carry = 0;
    for (int i = 0; i < NW; i++) {
      u128 s = (u128)a.v[i] + b.v[i] + carry;
      r.v[i] = (uint64_t)s;
      carry = (uint64_t)(s >> 64);
    }
    if (carry || geq_p(r.v)) sub_p(r.v);
    return r;
  }
  static El dbl(const El& a) { return add(a, a); }
  static El sub(const El& a, const El& b) {
    El r;
    uint64_t borrow = 0;
    for (int i = 0; i < NW; i++) {
      u128 d = (u128)a.v[i] - b.v[i] - borrow;
      r.v[i] = (uint64_t)d;
      borrow = (uint64_t)(d >> 64) & 1;
    }
    if (borrow) {
      uint64_t carry = 0;
      for (int i = 0; i < NW; i++) {
        u128 s = (u128)r.v[i] + C::MOD[i] + carry;
        r.v[i] = (uint64_t)s;
        carry = (uint64_t)(s >> 64);
      }
    }
    return r;
  }
  static El neg(const El& a) { return is_zero(a) ? a : sub(zero(), a); }
  static El cneg(const El& a, bool c) { return c ? neg(a) : a; }
  static El mul_portable(const El& a, const El& b) {
    uint64_t t[NW + 2];
    for (int i = 0; i < NW + 2; i++) t[i] = 0;
    for (int i = 0; i < NW; i++) {
      uint64_t carry = 0;
      for (int j = 0; j < NW; j++) {
        u128 acc = (u128)a.v[j] * b.v[i] + t[j] + carry;
        t[j] = (uint64_t)acc;
        carry = (uint64_t)(acc >> 64);
      }
      u128 acc = (u128)t[NW] + carry;
      t[NW] = (uint64_t)acc;
      t[NW + 1] = (uint64_t)(acc >> 64);
      uint64_t m = t[0] * C::N0;
      acc = (u128)m * C::MOD[0] + t[0];
      carry = (uint64_t)(acc >> 64);
      for (int j = 1; j < NW; j++) {
        acc = (u128)m * C::MOD[j] + t[j] + carry;
        t[j - 1] = (uint64_t)acc;
        carry = (uint64_t)(acc >> 64);
      }
      acc = (u128)t[NW] + carry;
      t[NW - 1] = (uint64_t)acc;
      t[NW] = t[NW + 1] + (uint64_t)(acc >> 64);
    }
    if (t[NW] || geq_p(t)) sub_p(t);
    El r;
    memcpy(r.v, t, sizeof r.v);
    return r;
  }
  static El mul(const El& a, const El& b) {
#if defined(MSM377_HAVE_ADX_MUL)
    if (NW == 6 && adx_enabled()) {
      uint64_t t[7];
      mont_mul6_adx(t, a.v, b.v, C::MOD, C::N0);
      if (t[6] || geq_p(t)) sub_p(t);
      El r;
      memcpy(r.v, t, sizeof r.v);
      return r;
    }
#endif
    return mul_portable(a, b);
  }
  // ---- inversion ----
  // inv_fermat: a^(p-2), ~570 products (16 us with the ADX multiplier).  inv: the same value by the Bernstein-Yang
  // divstep iteration ("safegcd", variable time: nothing here is secret), 62 divsteps at a time on the low words of
  // (f, g) = (p, a), the 2 x 2 transition matrix then applied to the full-size f, g and -- modulo p, with a Montgomery
  // style exact division by 2^62 -- to the coefficients (d, e) that keep f = d a, g = e a (mod p): ~10 rounds of ~0.15 us
  // for the 377-bit field.  The result is CHECKED (one product: a inv = 1) and falls back to inv_fermat if the check
  // fails, so a mistake here can cost time, never a wrong result.  inv(0) = 0, like the exponentiation.
  static El inv_fermat(const El& a) {  // a^(p-2); p is odd and p = 1 mod 4 here, so only the low word changes
    El r = one();
    for (int i = 64 * NW - 1; i >= 0; i--) {
      r = sqr(r);
      uint64_t w = C::MOD[i >> 6] - ((i >> 6) == 0 ? 2 : 0);
      if ((w >> (i & 63)) & 1) r = mul(r, a);
    }
    return r;
  }
  static constexpr int N62 = (64 * NW + 2 + 61) / 62;  // signed 62-bit limbs that hold |value| < 2 p (7 for 384 bits, 5 for 256)
  typedef __int128 i128;
  struct S62 {
    int64_t v[N62];  // value = sum v[i] 2^(62 i); limbs 0 .. N62-2 in [0, 2^62), the top one carries the sign
  };
  static S62 to62(const uint64_t* w) {
    S62 r;
    for (int i = 0; i < N62; i++) {
      const int bit = 62 * i, word = bit >> 6, off = bit & 63;
      uint64_t x = word < NW ? w[word] >> off : 0;
      if (off > 2 && word + 1 < NW) x |= w[word + 1] << (64 - off);
      r.v[i] = (int64_t)(x & 0x3fffffffffffffffull);
    }
    return r;
  }
  static void from62(const S62& a, uint64_t* w) {  // a in [0, 2^(64 NW))
    for (int i = 0; i < NW; i++) w[i] = 0;
    for (int i = 0; i < N62; i++) {
      const int bit = 62 * i, word = bit >> 6, off = bit & 63;
      const uint64_t x = (uint64_t)a.v[i];
      if (word < NW) w[word] |= x << off;
      if (off > 2 && word + 1 < NW) w[word + 1] |= x >> (64 - off);
    }
  }
  static bool is_zero62(const S62& a) {
    int64_t acc = 0;
    for (int i = 0; i < N62; i++) acc |= a.v[i];
    return acc == 0;
  }
  static bool is_neg62(const S62& a) { return a.v[N62 - 1] < 0; }
  // a += sign m (sign = +1 / -1), carries resolved; m has non-negative limbs
  static void add_mod62(S62& a, const S62& m, int sign) {
    int64_t carry = 0;
    for (int i = 0; i < N62; i++) {
      int64_t t = a.v[i] + (sign > 0 ? m.v[i] : -m.v[i]) + carry;
      if (i + 1 < N62) {
        carry = t >> 62;  // arithmetic shift: floor division
        t &= 0x3fffffffffffffffll;
      }
      a.v[i] = t;
    }
  }
  static bool geq62(const S62& a, const S62& m) {  // both non-negative with canonical limbs
    for (int i = N62 - 1; i >= 0; i--) {
      if (a.v[i] > m.v[i]) return true;
      if (a.v[i] < m.v[i]) return false;
    }
    return true;
  }
  struct Trans {
    int64_t u, v, q, r;  // 2^62 (f', g') = (u f + v g, q f + r g)
  };
  // 62 divsteps on the low 64 bits of f (odd) and g; delta as in Bernstein-Yang (starts at 1).
  static int64_t divsteps62(int64_t delta, uint64_t f, uint64_t g, Trans& t) {
    uint64_t u = 1, v = 0, q = 0, r = 1;  // two's complement
    int i = 62;
    for (;;) {
      // a run of even g: g /= 2, the f row doubles (common scale 2^steps), delta += 1 each
      const int zeros = __builtin_ctzll(g | (~0ull << i));  // at most i
      g >>= zeros;
      u <<= zeros;
      v <<= zeros;
      delta += zeros;
      i -= zeros;
      if (i == 0) break;
      if (delta > 0) {  // (f, g) <- (g, (g - f) / 2)
        delta = 1 - delta;
        const uint64_t nf = g, ng = g - f, nu = q << 1, nv = r << 1, nq = q - u, nr = r - v;
        f = nf, g = ng >> 1, u = nu, v = nv, q = nq, r = nr;
      } else {  // g <- (g + f) / 2
        delta = 1 + delta;
        g = (g + f) >> 1;
        q += u, r += v;
        u <<= 1, v <<= 1;
      }
      i--;
    }
    t.u = (int64_t)u, t.v = (int64_t)v, t.q = (int64_t)q, t.r = (int64_t)r;
    return delta;
  }
  static void update_fg62(S62& f, S62& g, const Trans& t) {
    const int64_t M = 0x3fffffffffffffffll;
    i128 cf = (i128)t.u * f.v[0] + (i128)t.v * g.v[0];
    i128 cg = (i128)t.q * f.v[0] + (i128)t.r * g.v[0];
    cf >>= 62, cg >>= 62;  // the low 62 bits are zero by construction
    for (int i = 1; i < N62; i++) {
      cf += (i128)t.u * f.v[i] + (i128)t.v * g.v[i];
      cg += (i128)t.q * f.v[i] + (i128)t.r * g.v[i];
      f.v[i - 1] = (int64_t)cf & M, cf >>= 62;
      g.v[i - 1] = (int64_t)cg & M, cg >>= 62;
    }
    f.v[N62 - 1] = (int64_t)cf;
    g.v[N62 - 1] = (int64_t)cg;
  }
  // (d, e) <- (u d + v e, q d + r e) / 2^62 mod m, for d, e in [0, m): add the multiple of m that clears the low 62 bits,
  // shift, bring the result from (-m, 2 m) back into [0, m).
  static void update_de62(S62& d, S62& e, const Trans& t, const S62& m, uint64_t m_inv62) {
    const int64_t M = 0x3fffffffffffffffll;
    i128 cd = (i128)t.u * d.v[0] + (i128)t.v * e.v[0];
    i128 ce = (i128)t.q * d.v[0] + (i128)t.r * e.v[0];
    const int64_t md = (int64_t)((0 - m_inv62 * (uint64_t)cd) & (uint64_t)M);
    const int64_t me = (int64_t)((0 - m_inv62 * (uint64_t)ce) & (uint64_t)M);
    cd += (i128)md * m.v[0], ce += (i128)me * m.v[0];
    cd >>= 62, ce >>= 62;
    S62 nd, ne;
    for (int i = 1; i < N62; i++) {
      cd += (i128)t.u * d.v[i] + (i128)t.v * e.v[i] + (i128)md * m.v[i];
      ce += (i128)t.q * d.v[i] + (i128)t.r * e.v[i] + (i128)me * m.v[i];
      nd.v[i - 1] = (int64_t)cd & M, cd >>= 62;
      ne.v[i - 1] = (int64_t)ce & M, ce >>= 62;
    }
    nd.v[N62 - 1] = (int64_t)cd;
    ne.v[N62 - 1] = (int64_t)ce;
    if (is_neg62(nd)) add_mod62(nd, m, +1); else if (geq62(nd, m)) add_mod62(nd, m, -1);
    if (is_neg62(ne)) add_mod62(ne, m, +1); else if (geq62(ne, m)) add_mod62(ne, m, -1);
    d = nd, e = ne;
  }
  // out = x^-1 mod p as a plain integer; false if it did not come out (x = 0 mod p, or the round limit)
  static bool modinv62(const uint64_t* x, uint64_t* out) {
    const S62 m = to62(C::MOD);
    uint64_t m_inv62 = C::MOD[0];  // Newton: the inverse of the odd low word modulo 2^64, then modulo 2^62
    for (int k = 0; k < 6; k++) m_inv62 *= 2 - C::MOD[0] * m_inv62;
    S62 f = m, g = to62(x), d, e;
    for (int i = 0; i < N62; i++) d.v[i] = 0, e.v[i] = 0;
    e.v[0] = 1;
    if (is_zero62(g)) return false;
    int64_t delta = 1;
    for (int round = 0; round < 24 && !is_zero62(g); round++) {  // (49 x 384 + 57) / 17 divsteps bound any input: 18 rounds
      Trans t;
      const uint64_t f0 = (uint64_t)f.v[0] | ((uint64_t)f.v[1] << 62), g0 = (uint64_t)g.v[0] | ((uint64_t)g.v[1] << 62);
      delta = divsteps62(delta, f0, g0, t);
      update_de62(d, e, t, m, m_inv62);
      update_fg62(f, g, t);
    }
    if (!is_zero62(g)) return false;
    // f = +-1 (the gcd), d a = f (mod p)
    bool plus = f.v[0] == 1, minus = f.v[0] == 0x3fffffffffffffffll;
    for (int i = 1; i < N62; i++) {
      plus = plus && f.v[i] == 0;
      minus = minus && f.v[i] == (i + 1 < N62 ? 0x3fffffffffffffffll : -1);
    }
    if (!plus && !minus) return false;
    if (minus && !is_zero62(d)) {  // d <- m - d
      for (int i = 0; i < N62; i++) d.v[i] = -d.v[i];
      add_mod62(d, m, +1);
    }
    from62(d, out);
    return true;
  }
  static El inv(const El& a) {
    if (is_zero(a)) return a;
    static const El r3 = mul(from_const(C::R2), from_const(C::R2));  // R^3: (a R)^-1 = a^-1 R^-1 -> a^-1 R
    El t;
    if (modinv62(a.v, t.v)) {
      const El r = mul(t, r3);
      if (eq(mul(a, r), one())) return r;
    }
    return inv_fermat(a);
  }
  static El mul_sub_mul(const El& a, const El& b, const El& c, const El& d) { return sub(mul(a, b), mul(c, d)); }
  static El from_words32(const uint32_t* w) {  // 2 NW little-endian u32 words, already in this Montgomery form
    El x;
    for (int i = 0; i < NW; i++) x.v[i] = ((uint64_t)w[2 * i + 1] << 32) | w[2 * i];
    return x;
  }
  // Montgomery square: 21 + 36 word products instead of 72 for NW = 6.
  static El sqr(const El& a) {
#if defined(MSM377_HAVE_ADX_MUL)
    if (NW == 6 && adx_enabled()) return mul(a, a);  // the two-chain multiplier beats the portable squaring (47 vs 57 ns)
#endif
    uint64_t t[2 * NW + 1];
    for (int i = 0; i <= 2 * NW; i++) t[i] = 0;
    for (int i = 0; i < NW; i++) {  // off-diagonal products, once
      uint64_t carry = 0;
      for (int j = i + 1; j < NW; j++) {
        u128 acc = (u128)a.v[i] * a.v[j] + t[i + j] + carry;
        t[i + j] = (uint64_t)acc;
        carry = (uint64_t)(acc >> 64);
      }
      t[i + NW] = carry;
    }
    uint64_t top = 0;  // double
    for (int i = 0; i < 2 * NW; i++) {
      const uint64_t v = t[i];
      t[i] = (v << 1) | top;
      top = v >> 63;
    }
    uint64_t carry = 0;  // add the squares
    for (int i = 0; i < NW; i++) {
      u128 acc = (u128)a.v[i] * a.v[i] + t[2 * i] + carry;
      t[2 * i] = (uint64_t)acc;
      acc = (u128)t[2 * i + 1] + (uint64_t)(acc >> 64);
      t[2 * i + 1] = (uint64_t)acc;
      carry = (uint64_t)(acc >> 64);
    }
    for (int i = 0; i < NW; i++) {  // Montgomery reduction, one word per round
      const uint64_t m = t[i] * C::N0;
      uint64_t c = 0;
      for (int j = 0; j < NW; j++) {
        u128 acc = (u128)m * C::MOD[j] + t[i + j] + c;
        t[i + j] = (uint64_t)acc;
        c = (uint64_t)(acc >> 64);
      }
      for (int k = i + NW; c && k <= 2 * NW; k++) {
        u128 acc = (u128)t[k] + c;
        t[k] = (uint64_t)acc;
        c = (uint64_t)(acc >> 64);
      }
    }
    if (t[2 * NW] || geq_p(t + NW)) sub_p(t + NW);
    El r;
    memcpy(r.v, t + NW, sizeof r.v);
    return r;
  }
  // F29::N x 29-bit limbs in the device's Montgomery form -> this format.
  static El from_limbs29_mont(const uint32_t* l) {
    uint32_t w[2 * NW];
    typename F29::El e;
    for (int j = 0; j < F29::N; j++) e.l[j] = l[j];
    F29::template to_words<2 * NW>(e, w);
    El x;
    for (int i = 0; i < NW; i++) x.v[i] = ((uint64_t)w[2 * i + 1] << 32) | w[2 * i];
    return mul(x, from_const(C::FROM29));
  }
  // Montgomery -> canonical little-endian bytes (8 NW of them).
  static void to_wire(const El& a, uint8_t* out) {
    El o = zero();
    o.v[0] = 1;
    El t = mul(a, o);
    for (int i = 0; i < NW; i++)
      for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(t.v[i] >> (8 * k));
  }
};

using Fp64 = FieldHost64<G1Consts64, Fp>;
using Fq64 = FieldHost64<EdConsts64, Fq>;

// ---- G1 tail ----
using G1H = G1T<Fp64>;

// X, Y, ZZ, ZZZ (13 device limbs each, device Montgomery form) -> host point.
inline G1H::XYZZ g1h_from_device_words(const uint32_t* w52) {
  G1H::XYZZ p;
  p.x = Fp64::from_limbs29_mont(w52);
  p.y = Fp64::from_limbs29_mont(w52 + 13);
  p.zz = Fp64::from_limbs29_mont(w52 + 26);
  p.zzz = Fp64::from_limbs29_mont(w52 + 39);
  return p;
}
// One point of a partial record: X, Y, ZZ, ZZZ as 12 u32 words each, already in the host's
// Montgomery form (k_gather_partials re-bases on the GPU).
inline G1H::XYZZ g1h_from_record_words(const uint32_t* w48) {
  G1H::XYZZ p;
  p.x = Fp64::from_words32(w48);
  p.y = Fp64::from_words32(w48 + 12);
  p.zz = Fp64::from_words32(w48 + 24);
  p.zzz = Fp64::from_words32(w48 + 36);
  return p;
}

// Affine wire format of an XYZZ point: x = X/ZZ, y = Y/ZZZ with ONE inversion
// (1/ZZ = ZZ^2 / ZZZ^2 because ZZ^3 = ZZZ^2).  Identity -> x = 0, y = 1 (submission.ts:93-95).
inline void g1h_to_wire(const G1H::XYZZ& p, uint8_t out[96]) {
  memset(out, 0, 96);
  if (G1H::is_identity(p)) {
    out[48] = 1;
    return;
  }
  Fp64::El i3 = Fp64::inv(p.zzz);
  Fp64::El t = Fp64::mul(i3, p.zz);
  Fp64::El i2 = Fp64::sqr(t);
  Fp64::to_wire(Fp64::mul(p.x, i2), out);
  Fp64::to_wire(Fp64::mul(p.y, i3), out + 48);
}

// Horner over the 16 x 16 partial points of a full MSM.  Window w contributes
//   G_w = Sum_w + sum_l 2^l * Plane_{w,l}      (Plane_{w,l} = sum of buckets whose (t-1) has bit l)
// and the MSM is sum_w 2^(16 w) G_w (submission.ts:310-318), i.e. one (16 num_windows)-step
// double-and-add over bit positions b = 16 w + l (16 windows on the plain path, 8 behind the GLV
// front end).  partials layout: [window][point][48 words], point 0 =
// Sum_w, point 1 + l = Plane_{w,l}.
// Bit positions [lo, hi) of the chain only (position b = 16 w + l): sum_b 2^(b - lo) record_b.
// Window geometry of the chain (sequencer.hip Phase::even): short_from = 0 -- every window is cbits wide; otherwise the
// windows from short_from on are one bit shorter (common.hpp even_offset: thirteen 16-bit and three 15-bit windows).
inline void tail_position(int b, int cbits, int short_from, int& w, int& l) {
  const int full = cbits * short_from;
  if (short_from == 0 || b < full) {
    w = b / cbits, l = b % cbits;
  } else {
    w = short_from + (b - full) / (cbits - 1), l = (b - full) % (cbits - 1);
  }
}
inline int tail_positions(int num_windows, int cbits, int short_from) {
  return cbits * num_windows - (short_from && num_windows > short_from ? num_windows - short_from : 0);
}
inline G1H::XYZZ g1h_horner_bits(const uint32_t* partials, int lo, int hi, uint32_t skip_windows = 0, int short_from = 0) {
  G1H::XYZZ acc = G1H::identity();
  for (int b = hi - 1; b >= lo; b--) {
    acc = G1H::dbl(acc);
    int w, l;
    tail_position(b, 16, short_from, w, l);
    if ((skip_windows >> w) & 1u) continue;
    const uint32_t* base = partials + (size_t)w * 16 * 48;
    if (l < 15) acc = G1H::add(acc, g1h_from_record_words(base + (size_t)(1 + l) * 48));
    if (l == 0) acc = G1H::add(acc, g1h_from_record_words(base));
  }
  return acc;
}
inline G1H::XYZZ g1h_horner(const uint32_t* partials, int num_windows, uint32_t skip_windows = 0, int short_from = 0) {
  return g1h_horner_bits(partials, 0, tail_positions(num_windows, 16, short_from), skip_windows, short_from);
}
inline void g1h_combine(const uint32_t* partials, int num_windows, uint8_t out[96], int short_from = 0) {
  g1h_to_wire(g1h_horner(partials, num_windows, 0, short_from), out);
}

// ---- G1 tail in twisted Edwards form (csrc/te377.hpp) ----
struct TeK64 {
  static Fp64::El two_d() { return Fp64::from_const(G1Consts64::TE_2D); }
};
using TeH = EdT<Fp64, TeK64>;

// A window's partial record (16 points x 48 words) carries its coordinate system in the top bit of word 11 of
// its first coordinate (values are < 2^377, the bit is otherwise zero): set = twisted Edwards (X, Y, T, Z),
// clear = Weierstrass (X, Y, ZZ, ZZZ).  Lets a sharded MSM mix windows that fell back to the Weierstrass path.
constexpr uint32_t TE_RECORD_TAG = 0x80000000u;
inline bool window_record_is_te(const uint32_t* window_record) { return (window_record[11] & TE_RECORD_TAG) != 0; }

inline TeH::Ext teh_from_record_words(const uint32_t* w48) {  // X, Y, T, Z as 12 u32 words each, host Montgomery form
  TeH::Ext p;
  p.x = Fp64::from_words32(w48);
  p.x.v[5] &= ~((uint64_t)TE_RECORD_TAG << 32);
  p.y = Fp64::from_words32(w48 + 12);
  p.t = Fp64::from_words32(w48 + 24);
  p.z = Fp64::from_words32(w48 + 36);
  return p;
}
// Back to the Weierstrass wire format (Z != 0: the GPU side has checked every addition).  With xe = X/Z, ye = Y/Z:
//   u = (1 + ye) / (1 - ye),  v = c u / xe,  x = u / s - 1,  y = v / s
//   => x = (Z + Y) X / (s (Z - Y) X) - 1,   y = c (Z + Y) Z / (s (Z - Y) X)        one inversion.
// X = 0 is the identity (Y = Z; wire x = 0, y = 1 like submission.ts:93-95) or the 2-torsion point (-1, 0) (Y = -Z).
inline void teh_to_wire(const TeH::Ext& p, uint8_t out[96]) {
  memset(out, 0, 96);
  if (Fp64::is_zero(p.x)) {
    if (Fp64::is_zero(Fp64::sub(p.y, p.z))) {
      out[48] = 1;
    } else {
      Fp64::to_wire(Fp64::neg(Fp64::one()), out);  // x = -1, y = 0
    }
    return;
  }
  const Fp64::El zy = Fp64::add(p.z, p.y);
  const Fp64::El inv = Fp64::inv(Fp64::mul(Fp64::sub(p.z, p.y), p.x));
  const Fp64::El t = Fp64::mul(zy, inv);
  const Fp64::El x = Fp64::sub(Fp64::mul(Fp64::mul(t, p.x), Fp64::from_const(G1Consts64::TE_INV_S)), Fp64::one());
  const Fp64::El y = Fp64::mul(Fp64::mul(t, p.z), Fp64::from_const(G1Consts64::TE_C_OVER_S));
  Fp64::to_wire(x, out);
  Fp64::to_wire(y, out + 48);
}
// The a = -1 law on this curve is NOT complete (d is a square, te377.hpp): an addition or doubling whose true
// result is one of the points at infinity of the Edwards model comes out with Z3 = 0, and a later operation can
// lead back to Z != 0 with a wrong value.  The GPU kernels check every addition; so does the tail: every add / dbl
// below reports Z3 = 0 through the sticky flag and the caller reruns on the Weierstrass path.  (Inside the
// prime-order subgroup this never fires.)
// (One per thread, written on every operation: a cache line pair of its own, or the tail threads fight over the line.)
struct alignas(128) TeChecked {
  bool bad = false;
  TeH::Ext add(const TeH::Ext& a, const TeH::Ext& b) {
    const TeH::Ext r = TeH::add(a, b);
    bad |= Fp64::is_zero(r.z);
    return r;
  }
  TeH::Ext dbl(const TeH::Ext& a) {
    const TeH::Ext r = TeH::dbl(a);
    bad |= Fp64::is_zero(r.z);
    return r;
  }
  TeH::Ext dbl_nt(const TeH::Ext& a) {  // no T: only before another doubling
    const TeH::Ext r = TeH::dbl_nt(a);
    bad |= Fp64::is_zero(r.z);
    return r;
  }
};
// Same Horner as g1h_combine over Edwards partial records.
inline bool teh_is_identity(const TeH::Ext& p) { return Fp64::is_zero(p.x) && Fp64::is_zero(Fp64::sub(p.y, p.z)); }
// cbits: distance of two windows in bits (16 on the main path, 11 on the narrow-window path for small inputs);
// planes: bit planes per window record = log2 of its buckets (15 / 11), all inside the same 16-point record.
// Bit positions [lo, hi) of the chain only (position b = cbits w + l): sum_b 2^(b - lo) record_b.
inline TeH::Ext teh_horner_bits(const uint32_t* partials, int lo, int hi, TeChecked& chk, uint32_t skip_windows = 0, int cbits = 16, int planes = 15,
                                int short_from = 0) {
  TeH::Ext acc = TeH::identity();
  for (int b = hi - 1; b >= lo; b--) {
    acc = chk.dbl(acc);
    int w, l;
    tail_position(b, cbits, short_from, w, l);
    if ((skip_windows >> w) & 1u) continue;
    const uint32_t* base = partials + (size_t)w * 16 * 48;
    // identity points cost nothing: the records of a rank that folded its windows (g1_fold_tagged) are mostly that
    if (l < planes) {
      const TeH::Ext p = teh_from_record_words(base + (size_t)(1 + l) * 48);
      if (!teh_is_identity(p)) acc = chk.add(acc, p);
    }
    if (l == 0) {
      const TeH::Ext p = teh_from_record_words(base);
      if (!teh_is_identity(p)) acc = chk.add(acc, p);
    }
  }
  return acc;
}
inline TeH::Ext teh_horner(const uint32_t* partials, int num_windows, TeChecked& chk, uint32_t skip_windows = 0, int cbits = 16, int planes = 15,
                           int short_from = 0) {
  return teh_horner_bits(partials, 0, tail_positions(num_windows, cbits, short_from), chk, skip_windows, cbits, planes, short_from);
}
inline int imin(int a, int b) { return a < b ? a : b; }
// ---- the tail in pieces (msm377.hip tail_horner_mt runs them on threads) ----
// The Horner chain over `positions` bit positions cut into at most `chains` pieces: the piece that owns positions
// [lo, hi) costs (hi - lo) steps of doubling + addition (~17 field products each) and then `lo` doublings (~7 each);
// the cuts balance that sum, so the pieces shrink towards the top.  bounds[0 .. used]; returns used <= chains.
constexpr double TAIL_STEP_COST = 17.0, TAIL_DBL_COST = 7.0;
inline int tail_split(int positions, int chains, int* bounds) {
  double lo = 0.0, hi = positions * TAIL_STEP_COST;
  auto reach = [&](double t) {
    double o = 0.0;
    for (int k = 0; k < chains; k++) o += fmax(0.0, (t - TAIL_DBL_COST * o) / TAIL_STEP_COST);
    return o;
  };
  for (int it = 0; it < 48; it++) {
    const double t = 0.5 * (lo + hi);
    (reach(t) >= positions ? hi : lo) = t;
  }
  int used = 0;
  double o = 0.0;
  bounds[0] = 0;
  for (int k = 0; k < chains && bounds[used] < positions; k++) {
    o += fmax(0.0, (hi - TAIL_DBL_COST * o) / TAIL_STEP_COST);
    const int b = k + 1 == chains ? positions : imin(positions, (int)(o + 0.5));
    if (b > bounds[used]) bounds[++used] = b;
  }
  bounds[used] = positions;
  return used;
}

// One piece: sum of the records of positions [lo, hi) x 2^(position), i.e. its own Horner chain and then `lo` doublings
// (all but the last without T: nothing reads it before the next doubling).
inline TeH::Ext teh_tail_piece(const uint32_t* partials, int lo, int hi, TeChecked& chk, int cbits = 16, int planes = 15, int short_from = 0) {
  TeH::Ext acc = teh_horner_bits(partials, lo, hi, chk, 0, cbits, planes, short_from);
  for (int i = 0; i + 1 < lo; i++) acc = chk.dbl_nt(acc);
  if (lo > 0) acc = chk.dbl(acc);
  return acc;
}
// The same tail as teh_combine, computed the way the threaded tail does -- `chains` pieces, added up -- on the calling
// thread (msm377_g1_combine_partials_split: the CPU tests' view of that decomposition).
inline bool teh_combine_split(const uint32_t* partials, int num_windows, uint8_t out[96], int chains, int cbits = 16, int planes = 15) {
  int bounds[65];
  if (chains < 1) chains = 1;
  if (chains > 64) chains = 64;
  const int used = tail_split(cbits * num_windows, chains, bounds);
  TeChecked chk;
  TeH::Ext acc = teh_tail_piece(partials, bounds[used - 1], bounds[used], chk, cbits, planes);
  for (int k = used - 2; k >= 0; k--) acc = chk.add(acc, teh_tail_piece(partials, bounds[k], bounds[k + 1], chk, cbits, planes));
  if (chk.bad) return true;
  teh_to_wire(acc, out);
  return false;
}

// false: done; true: an exceptional case of the law (out untouched).
inline bool teh_combine(const uint32_t* partials, int num_windows, uint8_t out[96], int cbits = 16, int planes = 15, int short_from = 0) {
  TeChecked chk;
  const TeH::Ext r = teh_horner(partials, num_windows, chk, 0, cbits, planes, short_from);
  if (chk.bad) return true;
  teh_to_wire(r, out);
  return false;
}

inline void words_from_fp64(const Fp64::El& a, uint32_t* w12) {
  for (int i = 0; i < 6; i++) {
    w12[2 * i] = (uint32_t)a.v[i];
    w12[2 * i + 1] = (uint32_t)(a.v[i] >> 32);
  }
}
// A rank's own share of the host tail, before the exchange: the records of `count` CONSECUTIVE windows are replaced
// by records with the same total -- point 0 of the first one becomes sum_w 2^(16 (w - first)) G_w (one short Horner
// chain), every other point the identity -- so that the final combine, which skips identity points, is left with
// its doublings and one addition per rank.  Records of mixed kinds are left as they are.
// Records whose Horner chain hits an exceptional case of the Edwards law are left as they are too (folding is
// optional; the final combine then decides).
inline void g1_fold_tagged(uint32_t* partials, int count) {
  if (count <= 0) return;
  const bool te = window_record_is_te(partials);
  for (int w = 1; w < count; w++)
    if (window_record_is_te(partials + (size_t)w * 16 * 48) != te) return;
  uint32_t folded[48];
  if (te) {
    TeChecked chk;
    const TeH::Ext f = teh_horner(partials, count, chk);
    if (chk.bad) return;
    words_from_fp64(f.x, folded);
    words_from_fp64(f.y, folded + 12);
    words_from_fp64(f.t, folded + 24);
    words_from_fp64(f.z, folded + 36);
  } else {
    const G1H::XYZZ f = g1h_horner(partials, count);
    words_from_fp64(f.x, folded);
    words_from_fp64(f.y, folded + 12);
    words_from_fp64(f.zz, folded + 24);
    words_from_fp64(f.zzz, folded + 36);
  }
  uint32_t ident[48];
  memset(ident, 0, sizeof(ident));
  uint32_t one[12];
  words_from_fp64(Fp64::one(), one);
  memcpy(ident + 12, one, sizeof(one));          // Y = 1 in both systems
  if (te) memcpy(ident + 36, one, sizeof(one));  // Edwards identity (0, 1, 0, 1); Weierstrass: ZZ = ZZZ = 0
  for (int k = 0; k < count * 16; k++) memcpy(partials + (size_t)k * 48, ident, sizeof(ident));
  memcpy(partials, folded, sizeof(folded));
  if (te)
    for (int w = 0; w < count; w++) partials[(size_t)w * 16 * 48 + 11] |= TE_RECORD_TAG;
}

// Partial records of either kind, window by window (see TE_RECORD_TAG).  All of one kind: one Horner chain.  Mixed
// (some ranks of a sharded MSM fell back to the Weierstrass path): one chain per kind over its own windows, the
// Edwards sum is mapped back to the Weierstrass curve and the two sums are added there.
// Returns true (out untouched) when the Edwards records add up to an exceptional case of their law.
inline bool g1_combine_tagged(const uint32_t* partials, int num_windows, uint8_t out[96]) {
  uint32_t te_mask = 0;
  for (int w = 0; w < num_windows; w++)
    if (window_record_is_te(partials + (size_t)w * 16 * 48)) te_mask |= 1u << w;
  const uint32_t all = num_windows >= 32 ? 0xffffffffu : ((1u << num_windows) - 1u);
  if (te_mask == all) return teh_combine(partials, num_windows, out);
  if (te_mask == 0) {
    g1h_combine(partials, num_windows, out);
    return false;
  }
  uint8_t te_wire[96];
  TeChecked chk;
  const TeH::Ext te_sum = teh_horner(partials, num_windows, chk, all & ~te_mask);
  if (chk.bad) return true;
  G1H::XYZZ acc = g1h_horner(partials, num_windows, te_mask);
  if (!(Fp64::is_zero(te_sum.x) && Fp64::is_zero(Fp64::sub(te_sum.y, te_sum.z)))) {  // not the identity
    teh_to_wire(te_sum, te_wire);
    uint32_t w[24];
    memcpy(w, te_wire, 96);
    G1H::Affine q;
    q.x = Fp64::mul(Fp64::from_words32(w), Fp64::from_const(G1Consts64::R2));
    q.y = Fp64::mul(Fp64::from_words32(w + 12), Fp64::from_const(G1Consts64::R2));
    acc = G1H::madd(acc, q);
  }
  g1h_to_wire(acc, out);
  return false;
}

// Sum of `count` affine wire points (x || y, 48-byte little-endian each; the identity as the wire format writes it:
// x = 0, y = 1) -- the last step of a points-partitioned multi-GPU MSM, whose ranks each return the MSM of their own
// slice of the points (host/sharding.py).  false: a coordinate is not below p.
inline bool g1h_add_wire_points(const uint8_t* pts, uint32_t count, uint8_t out[96]) {
  G1H::XYZZ acc = G1H::identity();
  for (uint32_t i = 0; i < count; i++) {
    uint32_t w[24];
    memcpy(w, pts + (size_t)96 * i, 96);
    uint64_t lim[2][6];
    for (int c = 0; c < 2; c++)
      for (int k = 0; k < 6; k++) lim[c][k] = (uint64_t)w[12 * c + 2 * k] | ((uint64_t)w[12 * c + 2 * k + 1] << 32);
    if (Fp64::geq_p(lim[0]) || Fp64::geq_p(lim[1])) return false;
    bool x_zero = true, y_one = lim[1][0] == 1;
    for (int k = 0; k < 6; k++) x_zero = x_zero && lim[0][k] == 0;
    for (int k = 1; k < 6; k++) y_one = y_one && lim[1][k] == 0;
    if (x_zero && y_one) continue;  // the identity
    G1H::Affine q;
    q.x = Fp64::mul(Fp64::from_words32(w), Fp64::from_const(G1Consts64::R2));
    q.y = Fp64::mul(Fp64::from_words32(w + 12), Fp64::from_const(G1Consts64::R2));
    acc = G1H::madd(acc, q);
  }
  g1h_to_wire(acc, out);
  return true;
}

// ---- Edwards tail ----
struct EdK64 {
  static Fq64::El two_d() { return Fq64::from_const(EdConsts64::ED_2D); }
};
using EdH = EdT<Fq64, EdK64>;

inline EdH::Ext edh_from_record_words(const uint32_t* w32) {  // X, Y, T, Z as 8 u32 words each, host Montgomery form
  EdH::Ext p;
  p.x = Fq64::from_words32(w32);
  p.y = Fq64::from_words32(w32 + 8);
  p.t = Fq64::from_words32(w32 + 16);
  p.z = Fq64::from_words32(w32 + 24);
  return p;
}
inline void edh_to_wire(const EdH::Ext& p, uint8_t out[64]) {
  Fq64::El zi = Fq64::inv(p.z);
  Fq64::to_wire(Fq64::mul(p.x, zi), out);
  Fq64::to_wire(Fq64::mul(p.y, zi), out + 32);
}
// Same Horner as g1h_horner_bits; partials layout [window][point][32 words].  Bit positions [lo, hi) of the chain.
inline EdH::Ext edh_horner_bits(const uint32_t* partials, int lo, int hi, int short_from = 0) {
  EdH::Ext acc = EdH::identity();
  for (int b = hi - 1; b >= lo; b--) {
    acc = EdH::dbl(acc);
    int w, l;
    tail_position(b, 16, short_from, w, l);
    const uint32_t* base = partials + (size_t)w * 16 * 32;
    if (l < 15) acc = EdH::add(acc, edh_from_record_words(base + (size_t)(1 + l) * 32));
    if (l == 0) acc = EdH::add(acc, edh_from_record_words(base));
  }
  return acc;
}
inline void edh_combine(const uint32_t* partials, uint8_t out[64], int short_from = 0) {
  edh_to_wire(edh_horner_bits(partials, 0, tail_positions(16, 16, short_from), short_from), out);
}

}  // namespace msm377
