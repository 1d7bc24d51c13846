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
// the q*p[0] term is a plain add: a product costs N*N + N*(N-1) multiply-adds.
//
// Unlike the reference (fr_sub(a,a) = p, conditional_reduce keeps p), every value
// returned here is canonical: 0 <= x < p, limbs < 2^29.
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

  // Montgomery product a*b*R^-1 mod p, R = 2^(29*N).  Operands canonical (< p).
  static MSM_HD El mul(const El& a, const El& b) {
    uint64_t t[N];
#pragma unroll
    for (int j = 0; j < N; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
      for (int j = 0; j < N; j++) t[j] += (uint64_t)a.l[i] * b.l[j];
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
    return reduce_once(r);
  }

  // a*b - c*d in ONE Montgomery reduction: a*b + (p - c)*d accumulates 26 product terms and 12
  // q*p terms per column (38 * 2^58 < 2^64), the quotient digits serve both products, and the
  // result (< 2.7 p) takes two conditional subtractions.  Saves the 156 multiply-adds of a second
  // reduction; used for Y3 = R (Q - X3) - Y1 PPP in every point addition.
  static MSM_HD El mul_sub_mul(const El& a, const El& b, const El& c, const El& d) {
    const El e = neg(c);
    uint64_t t[N];
#pragma unroll
    for (int j = 0; j < N; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
      for (int j = 0; j < N; j++) t[j] += (uint64_t)a.l[i] * b.l[j];
#pragma unroll
      for (int j = 0; j < N; j++) t[j] += (uint64_t)e.l[i] * d.l[j];
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
    return reduce_once(reduce_once(r));
  }

  // Montgomery square: off-diagonal terms once with a doubled operand (2a_i < 2^30).
  static MSM_HD El sqr(const El& a) {
    uint64_t t[2 * N];
#pragma unroll
    for (int j = 0; j < 2 * N; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      t[2 * i] += (uint64_t)a.l[i] * a.l[i];
      uint32_t a2 = a.l[i] << 1;
#pragma unroll
      for (int j = i + 1; j < N; j++) t[i + j] += (uint64_t)a2 * a.l[j];
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
      uint32_t q = (0u - (uint32_t)t[i]) & LMASK;
      uint64_t carry = (t[i] + q) >> LB;
#pragma unroll
      for (int j = 1; j < N; j++) t[i + j] += (uint64_t)q * C::MOD[j];
      t[i + 1] += carry;
    }
    El r;
#pragma unroll
    for (int j = 0; j < N - 1; j++) {
      r.l[j] = (uint32_t)t[N + j] & LMASK;
      t[N + j + 1] += t[N + j] >> LB;
    }
    r.l[N - 1] = (uint32_t)t[2 * N - 1];
    return reduce_once(r);
  }

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

using Fp = Field<G1Consts>;  // BLS12-377 base field, 13 limbs, R = 2^377
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
