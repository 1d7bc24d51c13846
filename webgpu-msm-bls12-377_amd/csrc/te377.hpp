// BLS12-377 G1 in twisted Edwards form (a = -1), extended coordinates (X : Y : T : Z).
//
// The reference accumulates G1 buckets with the 16-multiplication projective addition of
// src/submission/implementation/wgsl/curve/ec_bls12_377.template.wgsl:13-52; csrc/g1_xyzz.hpp cut that to
// 8M + 2S.  On MI355X the accumulation kernel runs against the socket power limit (clock drops to
// ~2.05-2.15 GHz at ~1.13 kW), so its time follows the number of 32x32-bit multiply-adds per addition, not the
// instruction count -- and those are set by the number of field products and Montgomery reductions.  The same
// group in twisted Edwards form needs 8 products / 8 reductions per addition of a PROJECTIVE input point (7 for an
// affine one) against 10 / 9 for XYZZ, a general addition 9 against 14, and no case distinctions at all: the
// unified formulas cover doubling, inverses and the identity (EFD add-2008-hwcd-3, k = 2d).  The reference's own
// Edwards code (src/submission/miscellaneous/wgsl/add_points_any_a.template.wgsl:24-71) is the same law for
// another curve; csrc/ed_ext.hpp holds that one.
//
// The map (constants from tools/gen_consts.py):  W: y^2 = x^3 + 1  ->  u = s (x + 1), v = s y  (Montgomery form,
// s = 1/sqrt(3), built on the 2-torsion point (-1, 0))  ->  xe = c u / v, ye = (u - 1) / (u + 1)  with
// -xe^2 + ye^2 = 1 + d xe^2 ye^2.  An input point is converted WITHOUT an inversion to the projective
// extended point
//     X = c u (u + 1),  Y = (u - 1) v,  T = c u (u - 1),  Z = v (u + 1)            (T = X Y / Z)
// and stored as (Y - X, Y + X, 2d T, 2Z): 6 products per point (T = X - 2cu and Y = Z - 2v cost none), then 8 per
// bucket addition.
//
// d is a square in Fp, so the law has exceptional pairs: the sum formula fails exactly when 1 +- d x1 x2 y1 y2 = 0,
// i.e. when Z3 = F G = 0, and that needs a point of even order (P +- Q must be one of the curve's points at
// infinity, orders 2 and 4).  Points of the prime-order subgroup -- everything the harness and the protocols above
// it ever feed an MSM -- never hit it.  For arbitrary curve points the engine stays correct anyway: every
// addition checks Z3 = 0 (is_bad), the conversion checks Z = 0 (the two input points the map does not cover), either
// one raises bit 2 of the error word and the call reruns on the Weierstrass path.
//
// Field arithmetic: the lazy forms of field29.hpp.  Every coordinate of a stored point is the output of a lazy
// product (< p + 2^354) or canonical; bounds of each line are replayed by tools/check_lazy_bounds.py.
#pragma once
#include "ed_ext.hpp"
#include "field29.hpp"

namespace msm377 {

// The lazy addition law itself, for any field with slack in its Montgomery radix and constants K (MOD, KP2, TE_2D):
// Fp / G1Consts (this file's curve) and Fq / EdConsts (Edwards-BLS12, ed_ext.hpp -- R = 2^261 over a 253-bit q
// already has the slack; its law is complete, is_bad never fires).
template <class F_, class K_>
struct TeLazy {
  using F = F_;
  using El = typename F::El;
  using K = K_;
  struct ABase {  // affine input point in precomputed form: y - x, y + x, 2d x y
    El ymx, ypx, kt;
  };
  struct PBase {  // projective input point in precomputed form, canonical coordinates
    El ymx, ypx, kt, z2;  // Y - X, Y + X, 2d T, 2 Z
  };
  struct Ext {
    El x, y, t, z;
  };
  static MSM_HD Ext identity() {
    Ext r;
    r.x = F::zero();
    r.y = F::one();
    r.t = F::zero();
    r.z = F::one();
    return r;
  }
  // A stored coordinate (lazy product, carry-normalised) that is 0 mod p: all limbs zero, or exactly p.
  static MSM_HD bool is_zero_mod_p(const El& a) { return F::is_zero(a) || F::eq(a, F::from_const(K::MOD)); }
  static MSM_HD bool is_bad(const Ext& p) { return is_zero_mod_p(p.z); }

  // p + q (neg: p - q; -(x, y) = (-x, y) swaps Y - X with Y + X and negates T): 8 products.
  //   A, B, C, D lazy products (< p + e);  E = B - A + 2p, F = D - C + 2p  in (p - e, 3p + e),
  //   G = D + C, H = B + A  < 2p + 2e;  E, F, G carry-normalised, H left limb-wise (limbs < 2^30).
  static MSM_HD Ext madd(const Ext& p, const PBase& q, bool neg) {
    const El a = F::mul_lz(F::add_kp_sub(p.y, K::KP2, p.x), F::select(neg, q.ypx, q.ymx));
    const El b = F::mul_lz(F::add_lz(p.y, p.x), F::select(neg, q.ymx, q.ypx));
    const El c = F::mul_lz(F::select(neg, F::kp_sub(K::KP2, q.kt), q.kt), p.t);
    const El d = F::mul_lz(p.z, q.z2);
    return finish(a, b, c, d);
  }
  // The same with an affine input point (Z2 = 1): 7 products; D = 2 Z1 limb-wise.
  static MSM_HD Ext madd_affine(const Ext& p, const ABase& q, bool neg) {
    const El a = F::mul_lz(F::add_kp_sub(p.y, K::KP2, p.x), F::select(neg, q.ypx, q.ymx));
    const El b = F::mul_lz(F::add_lz(p.y, p.x), F::select(neg, q.ymx, q.ypx));
    const El c = F::mul_lz(F::select(neg, F::kp_sub(K::KP2, q.kt), q.kt), p.t);
    return finish(a, b, c, F::add_lz(p.z, p.z));
  }
  // General addition: 9 products (one of them by the constant 2d).
  static MSM_HD Ext add(const Ext& p, const Ext& q) {
    const El a = F::mul_lz(F::norm(F::add_kp_sub(p.y, K::KP2, p.x)), F::norm(F::add_kp_sub(q.y, K::KP2, q.x)));
    const El b = F::mul_lz(F::norm(F::add_lz(p.y, p.x)), F::norm(F::add_lz(q.y, q.x)));
    const El c = F::mul_lz(F::mul_lz(p.t, q.t), F::from_const(K::TE_2D));
    const El d = F::mul_lz(p.z, q.z);
    return finish(a, b, c, F::add_lz(d, d));
  }
  // E = B - A, F = D - C, G = D + C, H = B + A;  X3 = E F, Y3 = G H, T3 = E H, Z3 = F G.
  // d may be limb-wise doubled (limbs < 2^30, value < 2p + 2e): F < 5p, G < 4p, still N-form after norm().
  static MSM_HD Ext finish(const El& a, const El& b, const El& c, const El& d) {
    const El e = F::norm(F::add_kp_sub(b, K::KP2, a));
    const El f = F::norm(F::add_kp_sub(d, K::KP2, c));
    const El g = F::norm(F::add_lz(d, c));
    const El h = F::add_lz(b, a);
    Ext r;
    r.x = F::mul_lz(e, f);
    r.y = F::mul_lz(h, g);
    r.t = F::mul_lz(h, e);
    r.z = F::mul_lz(f, g);
    return r;
  }
};

using EdLazy = TeLazy<Fq, EdConsts>;  // Edwards-BLS12 buckets (EdDev in msm377.hip)

struct Te377 : TeLazy<Fp, G1Consts> {
  using F = Fp;
  using El = Fp::El;
  using K = G1Consts;

  // Wire coordinates (12 little-endian u32 words each, canonical, NOT Montgomery) -> base record.  phi = true
  // converts the GLV image (beta x, y) of the same point.  b.z2 = 0 <=> the map is undefined at this point
  // (y = 0 or s (x + 1) = -1: the curve points of order 2 and 4).
  static MSM_HD PBase from_wire(const uint32_t* x12, const uint32_t* y12, bool phi) {
    const El xr = F::template from_words<12>(x12), yr = F::template from_words<12>(y12);
    const El u = F::add(F::mul(xr, F::from_const(phi ? K::TE_SBR : K::TE_SR)), F::from_const(K::TE_S));
    const El v = F::mul(yr, F::from_const(K::TE_SR));
    const El cu = F::add(F::mul(xr, F::from_const(phi ? K::TE_CSBR : K::TE_CSR)), F::from_const(K::TE_CS));
    // u - 1 = (u + 1) - 2, so only X and Z need a product: T = cu (u - 1) = X - 2 cu, Y = (u - 1) v = Z - 2 v.
    const El up = F::add(u, F::one());
    const El X = F::mul(cu, up), Z = F::mul(v, up);
    const El T = F::sub(X, F::dbl(cu)), Y = F::sub(Z, F::dbl(v));
    PBase b;
    b.ymx = F::sub(Y, X);
    b.ypx = F::add(Y, X);
    b.kt = F::mul(T, F::from_const(K::TE_2D));
    b.z2 = F::dbl(Z);
    return b;
  }

  // The first addition of a chain, identity + (+-)q, without the 8 products of the general law: the record holds
  // (Y - X, Y + X, 2d T, 2Z), and (2X : 2Y : 2T : 2Z) is the same projective point -- two additions and ONE product
  // (2T = (2d T) / d).  Canonical coordinates, as the accumulator invariant asks.
  static MSM_HD Ext from_base(const PBase& q, bool neg) {
    Ext r;
    const El x2 = F::sub(q.ypx, q.ymx), t2 = F::mul(q.kt, F::from_const(K::TE_INV_D));
    r.x = F::cneg(x2, neg);
    r.y = F::add(q.ypx, q.ymx);
    r.t = F::cneg(t2, neg);
    r.z = q.z2;
    return r;
  }
  static MSM_HD Ext from_base_affine(const ABase& q, bool neg) {  // Z = 1: (2x, 2y, 2dxy / d, 2)
    Ext r;
    const El x2 = F::sub(q.ypx, q.ymx), t2 = F::mul(q.kt, F::from_const(K::TE_INV_D));
    r.x = F::cneg(x2, neg);
    r.y = F::add(q.ypx, q.ymx);
    r.t = F::cneg(t2, neg);
    r.z = F::dbl(F::one());
    return r;
  }

  // The affine form of the same record: one Fermat inversion per point (~450 field products), affordable only
  // where the table is reused (msm377_g1_set_bases).  bad <=> the map is undefined at this point.
  static MSM_HD ABase affine_from_wire(const uint32_t* x12, const uint32_t* y12, bool& bad) {
    const El xr = F::template from_words<12>(x12), yr = F::template from_words<12>(y12);
    const El u = F::add(F::mul(xr, F::from_const(K::TE_SR)), F::from_const(K::TE_S));
    const El v = F::mul(yr, F::from_const(K::TE_SR));
    const El cu = F::add(F::mul(xr, F::from_const(K::TE_CSR)), F::from_const(K::TE_CS));
    const El up = F::add(u, F::one()), um = F::sub(u, F::one());
    const El Z = F::mul(v, up);
    bad = F::is_zero(Z);
    const El zi = fe_pow<F, K::PM2_NW>(Z, K::PM2_W);
    const El x = F::mul(F::mul(cu, up), zi), y = F::mul(F::mul(um, v), zi);
    ABase b;
    b.ymx = F::sub(y, x);
    b.ypx = F::add(y, x);
    b.kt = F::mul(F::mul(x, y), F::from_const(K::TE_2D));
    return b;
  }

};

}  // namespace msm377
