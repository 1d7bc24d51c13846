// Host-only BLS12-377 base field on 6 x 64-bit words (Montgomery radix 2^384) for the CPU
// tail of the pipeline: Horner over the window/bit-plane partial sums and the single field
// inversion of the result (replaces the reference's BigInt tail with 4096 inversions,
// src/submission/submission.ts:290-321 and cuzk/bls12_377.ts:41-63).  Same static interface
// as Field<> in field29.hpp so G1T<> works over it.
#pragma once
#include <stdint.h>
#include <string.h>

#include "field29.hpp"
#include "g1_xyzz.hpp"

namespace msm377 {

struct Fp64 {
  struct El {
    uint64_t v[6];
  };
  using C = G1Consts64;
  typedef unsigned __int128 u128;

  static El zero() {
    El r;
    memset(&r, 0, sizeof r);
    return r;
  }
  static El one() {
    El r;
    for (int i = 0; i < 6; i++) r.v[i] = C::ONE[i];
    return r;
  }
  static bool is_zero(const El& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3] | a.v[4] | a.v[5]) == 0; }
  static bool eq(const El& a, const El& b) {
    uint64_t acc = 0;
    for (int i = 0; i < 6; i++) acc |= a.v[i] ^ b.v[i];
    return acc == 0;
  }
  static bool geq_p(const uint64_t* a) {
    for (int i = 5; i >= 0; i--) {
      if (a[i] > C::MOD[i]) return true;
      if (a[i] < C::MOD[i]) return false;
    }
    return true;
  }
  static void sub_p(uint64_t* a) {
    uint64_t borrow = 0;
    for (int i = 0; i < 6; i++) {
      u128 d = (u128)a[i] - C::MOD[i] - borrow;
      a[i] = (uint64_t)d;
      borrow = (uint64_t)(d >> 64) & 1;
    }
  }
  static El add(const El& a, const El& b) {
    El r;
    uint64_t carry = 0;
    for (int i = 0; i < 6; i++) {
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
    for (int i = 0; i < 6; i++) {
      u128 d = (u128)a.v[i] - b.v[i] - borrow;
      r.v[i] = (uint64_t)d;
      borrow = (uint64_t)(d >> 64) & 1;
    }
    if (borrow) {
      uint64_t carry = 0;
      for (int i = 0; i < 6; i++) {
        u128 s = (u128)r.v[i] + C::MOD[i] + carry;
        r.v[i] = (uint64_t)s;
        carry = (uint64_t)(s >> 64);
      }
    }
    return r;
  }
  static El neg(const El& a) { return is_zero(a) ? a : sub(zero(), a); }
  static El mul(const El& a, const El& b) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; i++) {
      uint64_t carry = 0;
      for (int j = 0; j < 6; j++) {
        u128 acc = (u128)a.v[j] * b.v[i] + t[j] + carry;
        t[j] = (uint64_t)acc;
        carry = (uint64_t)(acc >> 64);
      }
      u128 acc = (u128)t[6] + carry;
      t[6] = (uint64_t)acc;
      t[7] = (uint64_t)(acc >> 64);
      uint64_t m = t[0] * C::N0;
      acc = (u128)m * C::MOD[0] + t[0];
      carry = (uint64_t)(acc >> 64);
      for (int j = 1; j < 6; j++) {
        acc = (u128)m * C::MOD[j] + t[j] + carry;
        t[j - 1] = (uint64_t)acc;
        carry = (uint64_t)(acc >> 64);
      }
      acc = (u128)t[6] + carry;
      t[5] = (uint64_t)acc;
      t[6] = t[7] + (uint64_t)(acc >> 64);
    }
    if (t[6] || geq_p(t)) sub_p(t);
    El r;
    memcpy(r.v, t, 48);
    return r;
  }
  static El sqr(const El& a) { return mul(a, a); }
  static El inv(const El& a) {  // a^(p-2)
    El r = one();
    for (int i = 383; i >= 0; i--) {
      r = sqr(r);
      uint64_t w = C::MOD[i >> 6] - ((i >> 6) == 0 ? 2 : 0);  // p - 2: only the low word changes
      if ((w >> (i & 63)) & 1) r = mul(r, a);
    }
    return r;
  }
  // 13 x 29-bit limbs, Montgomery radix 2^377 (the device format) -> this format.
  static El from_limbs29_mont(const uint32_t* l) {
    uint32_t w[12];
    Fp::El e;
    for (int j = 0; j < 13; j++) e.l[j] = l[j];
    Fp::to_words<12>(e, w);
    El x, c;
    for (int i = 0; i < 6; i++) {
      x.v[i] = ((uint64_t)w[2 * i + 1] << 32) | w[2 * i];
      c.v[i] = C::FROM29[i];
    }
    return mul(x, c);
  }
  // Montgomery -> canonical 48-byte little-endian.
  static void to_wire(const El& a, uint8_t* out) {
    El o = zero();
    o.v[0] = 1;
    El t = mul(a, o);
    for (int i = 0; i < 6; i++)
      for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(t.v[i] >> (8 * k));
  }
};

using G1H = G1T<Fp64>;

// X, Y, ZZ, ZZZ (13 device limbs each) -> host point.
inline G1H::XYZZ g1h_from_device_words(const uint32_t* w52) {
  G1H::XYZZ p;
  p.x = Fp64::from_limbs29_mont(w52);
  p.y = Fp64::from_limbs29_mont(w52 + 13);
  p.zz = Fp64::from_limbs29_mont(w52 + 26);
  p.zzz = Fp64::from_limbs29_mont(w52 + 39);
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
// and the MSM is sum_w 2^(16 w) G_w (submission.ts:310-318), i.e. one 256-step double-and-add
// over bit positions b = 16 w + l.  partials layout: [window][point][52 words], point 0 =
// Sum_w, point 1 + l = Plane_{w,l}.
inline void g1h_combine(const uint32_t* partials, uint8_t out[96]) {
  G1H::XYZZ acc = G1H::identity();
  for (int b = 255; b >= 0; b--) {
    acc = G1H::dbl(acc);
    const int w = b >> 4, l = b & 15;
    const uint32_t* base = partials + (size_t)w * 16 * 52;
    if (l < 15) acc = G1H::add(acc, g1h_from_device_words(base + (size_t)(1 + l) * 52));
    if (l == 0) acc = G1H::add(acc, g1h_from_device_words(base));
  }
  g1h_to_wire(acc, out);
}

}  // namespace msm377
