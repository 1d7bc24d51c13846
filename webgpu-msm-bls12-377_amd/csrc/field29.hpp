// Prime-field arithmetic on 29-bit limbs for gfx950 (and, for unit tests, the host).
//
// Replaces the reference's 13-bit-limb WGSL field library:
//   montgomery_product / conditional_reduce  src/submission/implementation/wgsl/montgomery/mont_pro_product.template.wgsl:15-62
//   fr_add / fr_sub / fr_reduce               src/submission/implementation/wgsl/field/field.template.wgsl:1-32
//   bigint_add / bigint_sub / bigint_gt       src/submission/implementation/wgsl/bigint/bigint.template.wgsl:1-45
//   Barrett field_mul (to-Montgomery only)    src/submission/implementation/wgsl/cuzk/barrett.template.wgsl:60-82
//
// Why 29 bits: gfx950's widest integer multiply is v_mad_u64_u32 (32x32+64 -> 64, no
// carry-in).  With 29-bit limbs a 13-limb product column holds at most 13 a*b terms and
// 12 q*p terms of < 2^58 each, so every column fits a 64-bit accumulator and the whole
// product is a carry-free stream of v_mad_u64_u32; carries are resolved once at the end.
// Both moduli are = 1 (mod 2^29), so the Montgomery quotient digit is (-t0) mod 2^29 and
// the q*p[0] term is a plain add: a product costs N*N + RS*(N-1) multiply-adds (RS = reduction
// steps: N for Fq, N + 1 for Fp -- see "Montgomery products" below).
//
// Unlike the reference (fr_sub(a,a) = p, conditional_reduce keeps p), every value
// returned by the plain operations (add, sub, mul, ...) is canonical: 0 <= x < p, limbs < 2^29.
// The *_lz operations used inside the G1 point formulas trade that for fewer instructions.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MSM_HD __host__ __device__ __forceinline__
#else
#define MSM_HD inline __attribute__((always_inline))
#endif

#include "consts_gen.hpp"

namespace msm377 {

constexpr int LB = 29;
constexpr uint32_t LMASK = (1u << LB) - 1u;

template <int N>
struct Limbs {
  uint32_t l[N];
};

template <class C>
struct Field {
  static constexpr int N = C::NL;
  static constexpr int RS = C::RS;  // Montgomery radix R = 2^(29 RS)
  using Consts = C;
  using El = Limbs<N>;

  static MSM_HD El zero() {
    El r;
#pragma unroll
    for (int j = 0; j < N; j++) r.l[j] = 0;
    return r;
  }
  static MSM_HD El from_const(const uint32_t (&c)[N]) {
    El r;
#pragma unroll
    for (int j = 0; j < N; j++) r.l[j] = c[j];
    return r;
  }
  static MSM_HD El one() { return from_const(C::ONE); }  // Montgomery form of 1

  static MSM_HD bool is_zero(const El& a) {
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < N; j++) acc |= a.l[j];
    return acc == 0;
  }
  static MSM_HD bool eq(const El& a, const El& b) {
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < N; j++) acc |= (a.l[j] ^ b.l[j]);
    return acc == 0;
  }
  static MSM_HD El select(bool c, const El& a, const El& b) {  // c ? a : b
    El r;
#pragma unroll
    for (int j = 0; j < N; j++) r.l[j] = c ? a.l[j] : b.l[j];
    return r;
  }

  // x has limbs < 2^29 except the top one (< 2^31), value < 2p.  Returns x mod p.
  static MSM_HD El reduce_once(const El& x) {
    El d;
    int32_t bw = 0;
#pragma unroll
    for (int j = 0; j < N - 1; j++) {
      int32_t v = (int32_t)x.l[j] - (int32_t)C::MOD[j] + bw;
      d.l[j] = (uint32_t)v & LMASK;
      bw = v >> 31;
    }
    int32_t top = (int32_t)x.l[N - 1] - (int32_t)C::MOD[N - 1] + bw;
    d.l[N - 1] = (uint32_t)top;
    return select(top < 0, x, d);
  }

  static MSM_HD El add(const El& a, const El& b) {
    El s;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < N - 1; j++) {
      uint32_t v = a.l[j] + b.l[j] + c;
      s.l[j] = v & LMASK;
      c = v >> LB;
    }
    s.l[N - 1] = a.l[N - 1] + b.l[N - 1] + c;
    return reduce_once(s);
  }
  static MSM_HD El dbl(const El& a) { return add(a, a); }

  static MSM_HD El sub(const El& a, const El& b) {
    El d, e;
    int32_t bw = 0, c2 = 0;
#pragma unroll
    for (int j = 0; j < N - 1; j++) {
      int32_t v = (int32_t)a.l[j] - (int32_t)b.l[j];
      int32_t v1 = v + bw;
      d.l[j] = (uint32_t)v1 & LMASK;
      bw = v1 >> 31;
      int32_t v2 = v + (int32_t)C::MOD[j] + c2;  // in (-2^29, 2^30)
      e.l[j] = (uint32_t)v2 & LMASK;
      c2 = v2 >> LB;  // arithmetic: -1, 0 or 1
    }
    int32_t v = (int32_t)a.l[N - 1] - (int32_t)b.l[N - 1];
    int32_t top = v + bw;
    d.l[N - 1] = (uint32_t)top;
    e.l[N - 1] = (uint32_t)(v + (int32_t)C::MOD[N - 1] + c2);
    return select(top < 0, e, d);
  }

  static MSM_HD El neg(const El& a) {  // p - a, and 0 for a = 0
    El r;
    int32_t bw = 0;
    uint32_t nz = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
      nz |= a.l[j];
      int32_t v = (int32_t)C::MOD[j] - (int32_t)a.l[j] + bw;
      r.l[j] = (j < N - 1) ? ((uint32_t)v & LMASK) : (uint32_t)v;
      bw = v >> 31;
    }
    return select(nz != 0, r, a);
  }
  static MSM_HD El cneg(const El& a, bool c) { return select(c, neg(a), a); }

  // ---- Montgomery products, R = 2^(29 RS) ----
  //
  // RS = N (Fq): the classic form, a product of canonical operands is < 2p.
  // RS = N + 1 (Fp: p fills all but half a bit of its 13 limbs, so R = 2^377 leaves no slack at
  // all): one more reduction step than the operands have limbs.  The operands may then be ANY
  // 13-limb values below 2^380 -- not reduced mod p -- and the product is below p + 2^354: the
  // "lazy" forms below (mul_lz, sqr_lz, mul_add_mul_lz) skip the conditional subtraction and the
  // point formulas replace modular add/sub by limb-wise ones against a multiple of p
  // (add_kp_sub).  Per mixed addition: 4400 instead of 5350 VALU instructions, 160 instead of
  // 248 VGPRs (hipcc 7.2, gfx950).
  //
  // Column bound (every column of the schoolbook product is ONE u64 accumulator): with limb
  // bounds A_i, B_j the worst column holds sum A_i B_j + 12 * 2^58 (the q p terms) + a carry
  // < 2^36.  Allowed operand shapes, checked per call site in g1_xyzz.hpp:
  //   "N-form"  limbs 0..11 < 2^29, top limb < 2^31.6   (norm() output, value < 7.1 p)
  //   "lazy"    limbs < 3 * 2^29, top limb < 2^31       (one add_kp_sub of N-form values with top limbs < 2^29.1)
  // N x N, lazy x N (top limb of the N operand < 2^29.1) and N^2 fit; lazy x lazy does not.

  // Lazy product: see above.  Output N-form with value < p + 2^354 (top limb <= MOD[N-1] + 64).
  static MSM_HD El mul_lz(const El& a, const El& b) {
    uint64_t t[N];
#pragma unroll
    for (int j = 0; j < N; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < RS; i++) {
      if (i < N) {
#pragma unroll
        for (int j = 0; j < N; j++) t[j] += (uint64_t)a.l[i] * b.l[j];
      }
      uint32_t q = (0u - (uint32_t)t[0]) & LMASK;
      uint64_t carry = (t[0] + q) >> LB;  // low 29 bits cancel exactly
#pragma unroll
      for (int j = 1; j < N; j++) t[j] += (uint64_t)q * C::MOD[j];
      t[1] += carry;
#pragma unroll
      for (int j = 0; j < N - 1; j++) t[j] = t[j + 1];
      t[N - 1] = 0;
    }
    El r;
#pragma unroll
    for (int j = 0; j < N - 1; j++) {
      r.l[j] = (uint32_t)t[j] & LMASK;
      t[j + 1] += t[j] >> LB;
    }
    r.l[N - 1] = (uint32_t)t[N - 1];
    return r;
  }

  // a*b + e*d in ONE reduction (Y3 = R (Q - X3) + (-Y1) PPP of every point addition): the quotient
  // digits serve both products.  Same output contract as mul_lz; the column bound covers both sums.
  static MSM_HD El mul_add_mul_lz(const El& a, const El& b, const El& e, const El& d) {
    uint64_t t[N];
#pragma unroll
    for (int j = 0; j < N; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < RS; i++) {
      if (i < N) {
#pragma unroll
        for (int j = 0; j < N; j++) t[j] += (uint64_t)a.l[i] * b.l[j];
#pragma unroll
        for (int j = 0; j < N; j++) t[j] += (uint64_t)e.l[i] * d.l[j];
      }
      uint32_t q = (0u - (uint32_t)t[0]) & LMASK;
      uint64_t carry = (t[0] + q) >> LB;
#pragma unroll
      for (int j = 1; j < N; j++) t[j] += (uint64_t)q * C::MOD[j];
      t[1] += carry;
#pragma unroll
      for (int j = 0; j < N - 1; j++) t[j] = t[j + 1];
      t[N - 1] = 0;
    }
    El r;
#pragma unroll
    for (int j = 0; j < N - 1; j++) {
      r.l[j] = (uint32_t)t[j] & LMASK;
      t[j + 1] += t[j] >> LB;
    }
    r.l[N - 1] = (uint32_t)t[N - 1];
    return r;
  }

  // Lazy square of an N-form value: off-diagonal terms once with a doubled operand (2 a_i < 2^30 for
  // i <= N-2; the top limb is never the doubled one).
  static MSM_HD El sqr_lz(const El& a) {
    uint64_t t[RS + N];
#pragma unroll
    for (int j = 0; j < RS + N; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      t[2 * i] += (uint64_t)a.l[i] * a.l[i];
      const uint32_t a2 = a.l[i] << 1;
#pragma unroll
      for (int j = i + 1; j < N; j++) t[i + j] += (uint64_t)a2 * a.l[j];
    }
#pragma unroll
    for (int i = 0; i < RS; i++) {
      uint32_t q = (0u - (uint32_t)t[i]) & LMASK;
      uint64_t carry = (t[i] + q) >> LB;
#pragma unroll
      for (int j = 1; j < N; j++) t[i + j] += (uint64_t)q * C::MOD[j];
      t[i + 1] += carry;
    }
    El r;
#pragma unroll
    for (int j = 0; j < N - 1; j++) {
      r.l[j] = (uint32_t)t[RS + j] & LMASK;
      t[RS + j + 1] += t[RS + j] >> LB;
    }
    r.l[N - 1] = (uint32_t)t[RS + N - 1];
    return r;
  }

  // Canonical forms: operands canonical (or anything the lazy forms accept), result in [0, p).
  static MSM_HD El mul(const El& a, const El& b) { return reduce_once(mul_lz(a, b)); }
  static MSM_HD El sqr(const El& a) { return reduce_once(sqr_lz(a)); }
  // a*b - c*d with a single Montgomery reduction.  RS = N: a*b + (p - c)*d is < 2.7 p after the
  // reduction (two conditional subtractions); RS = N + 1: < p + 2^354 (one).
  static MSM_HD El mul_sub_mul(const El& a, const El& b, const El& c, const El& d) {
    const El r = mul_add_mul_lz(a, b, neg(c), d);
    return RS > N ? reduce_once(r) : reduce_once(reduce_once(r));
  }

  // ---- limb-wise helpers of the lazy formulas (RS = N + 1 only) ----

  // a + K - b - s * b2 limb by limb, no carries: K is a multiple of p from consts_gen.hpp whose limbs
  // are raised by w * 2^29 (KP2, KP6: w = 1; KP4W3: w = 3), so no limb goes negative as long as
  // b (+ s b2) has at most w units of 2^29 per limb and a top limb below K's.  The value is
  // a - b - s b2 + K, congruent to a - b - s b2.
  static MSM_HD El add_kp_sub(const El& a, const uint32_t (&K)[N], const El& b) {
    El r;
#pragma unroll
    for (int j = 0; j < N; j++) r.l[j] = a.l[j] + K[j] - b.l[j];
    return r;
  }
  static MSM_HD El add_lz(const El& a, const El& b) {  // a + b, limb-wise
    El r;
#pragma unroll
    for (int j = 0; j < N; j++) r.l[j] = a.l[j] + b.l[j];
    return r;
  }
  static MSM_HD El kp_sub(const uint32_t (&K)[N], const El& b) {  // K - b
    El r;
#pragma unroll
    for (int j = 0; j < N; j++) r.l[j] = K[j] - b.l[j];
    return r;
  }
  static MSM_HD El add_kp_sub_sub2(const El& a, const uint32_t (&K)[N], const El& b, const El& b2) {  // a + K - b - 2 b2
    El r;
#pragma unroll
    for (int j = 0; j < N; j++) r.l[j] = a.l[j] + K[j] - b.l[j] - 2u * b2.l[j];
    return r;
  }
  // Carry propagation only: limbs 0..N-2 back below 2^29, the value (not reduced mod p) unchanged.
  static MSM_HD El norm(const El& a) {
    El r;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < N - 1; j++) {
      const uint32_t v = a.l[j] + c;
      r.l[j] = v & LMASK;
      c = v >> LB;
    }
    r.l[N - 1] = a.l[N - 1] + c;
    return r;
  }
  // x (N-form) >= m ? x - m : x, for m = MOD, MOD2, MOD4 (top limb of either up to 32 bits).
  static MSM_HD El csub(const El& x, const uint32_t (&m)[N]) {
    El d;
    int32_t bw = 0;
#pragma unroll
    for (int j = 0; j < N - 1; j++) {
      const int32_t v = (int32_t)x.l[j] - (int32_t)m[j] + bw;
      d.l[j] = (uint32_t)v & LMASK;
      bw = v >> 31;
    }
    const int64_t top = (int64_t)x.l[N - 1] - (int64_t)m[N - 1] + bw;
    d.l[N - 1] = (uint32_t)top;
    return select(top < 0, x, d);
  }
  // Any N-form value below 8 p -> [0, p).  Rare paths only (special cases of the point formulas, output).
  static MSM_HD El canon(const El& x) { return csub(csub(csub(x, C::MOD4), C::MOD2), C::MOD); }

  static MSM_HD El to_mont(const El& a) { return mul(a, from_const(C::R2)); }
  static MSM_HD El from_mont(const El& a) {
    El o = zero();
    o.l[0] = 1;
    return mul(a, o);
  }

  // Little-endian u32 words (the harness's LE byte buffers) -> limbs.  NW = 12 for the
  // 48-byte G1 coordinates, 8 for 32-byte Edwards coordinates.  Value must be < p.
  template <int NW>
  static MSM_HD El from_words(const uint32_t* w) {
    El r;
#pragma unroll
    for (int j = 0; j < N; j++) {
      const int bit = LB * j, wi = bit >> 5, off = bit & 31;
      uint32_t v = 0;
      if (wi < NW) v = w[wi] >> off;
      if (off > 32 - LB && wi + 1 < NW) v |= w[wi + 1] << (32 - off);
      r.l[j] = v & LMASK;
    }
    return r;
  }
  template <int NW>
  static MSM_HD void to_words(const El& a, uint32_t* w) {
#pragma unroll
    for (int k = 0; k < NW; k++) w[k] = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
      const int bit = LB * j, wi = bit >> 5, off = bit & 31;
      if (wi < NW) w[wi] |= a.l[j] << off;
      if (off > 32 - LB && wi + 1 < NW) w[wi + 1] |= a.l[j] >> (32 - off);
    }
  }
};

using Fp = Field<G1Consts>;  // BLS12-377 base field, 13 limbs, R = 2^406
using Fq = Field<EdConsts>;  // Edwards-BLS12 base field (= BLS12-377 scalar field), 9 limbs, R = 2^261

// a^e for a public exponent given as little-endian u32 words (host tail only: inversion).
template <class F, int NW>
MSM_HD typename F::El fe_pow(const typename F::El& a, const uint32_t* e) {
  typename F::El r = F::one();
  for (int i = NW * 32 - 1; i >= 0; i--) {
    r = F::sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1u) r = F::mul(r, a);
  }
  return r;
}

}  // namespace msm377
