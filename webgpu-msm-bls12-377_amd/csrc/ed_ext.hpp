// Twisted Edwards BLS12 ("Edwards-BLS12", a = -1, d = 3021 over Fq = the BLS12-377 scalar
// field) in extended coordinates (X : Y : T : Z), x = X/Z, y = Y/Z, T = XY/Z.
//
// Replaces the reference's Edwards routines (orphaned under src/submission/miscellaneous at
// this commit; paths relative to /root/reference/):
//   add_points (add-2008-hwcd, any a)   src/submission/miscellaneous/wgsl/add_points_any_a.template.wgsl:24-71
//   a = -1 fast path                    src/submission/miscellaneous/add_points.ts:55-112
//   negate_point (-x, y, -t, z)         src/submission/miscellaneous/wgsl/scalar_mul.template.wgsl:66-75
//   identity (0, R, 0, R)               src/submission/miscellaneous/wgsl/horners_rule.template.wgsl:19-25
//   t = x*y on load                     src/submission/miscellaneous/wgsl/convert_inputs.template.wgsl:34-41
//   curve constants                     src/reference/params/AleoConstants.ts:2-5
// Here: the unified a = -1 formulas with k = 2d (EFD add-2008-hwcd-3 / madd-2008-hwcd-3,
// dbl-2008-hwcd).  a = -1 is a square and d a non-square in Fq, so the addition law is complete:
// no special cases for doubling, inverses or the identity.  Input points are kept in the
// "precomputed" form (y - x, y + x, 2d*x*y), which makes a mixed addition 7M.
#pragma once
#include "field29.hpp"

namespace msm377 {

// F: field policy (Fq on the device, Fq64 on the host); K: provides K::two_d() in F's Montgomery form.
template <class F, class K>
struct EdT {
  using El = typename F::El;

  struct Base {  // affine input point, precomputed form
    El ymx, ypx, kt;
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
  // (x, y) in Montgomery form -> precomputed base
  static MSM_HD Base make_base(const El& x, const El& y) {
    Base b;
    b.ymx = F::sub(y, x);
    b.ypx = F::add(y, x);
    b.kt = F::mul(F::mul(x, y), K::two_d());
    return b;
  }
  // -(x, y) = (-x, y): swaps y-x and y+x, negates 2d*x*y
  static MSM_HD Base cneg(const Base& b, bool c) {
    Base r;
    r.ymx = F::select(c, b.ypx, b.ymx);
    r.ypx = F::select(c, b.ymx, b.ypx);
    r.kt = F::cneg(b.kt, c);
    return r;
  }
  // madd-2008-hwcd-3: 7M
  static MSM_HD Ext madd(const Ext& p, const Base& q) {
    El a = F::mul(F::sub(p.y, p.x), q.ymx);
    El b = F::mul(F::add(p.y, p.x), q.ypx);
    El c = F::mul(p.t, q.kt);
    El d = F::dbl(p.z);
    El e = F::sub(b, a), f = F::sub(d, c), g = F::add(d, c), h = F::add(b, a);
    Ext r;
    r.x = F::mul(e, f);
    r.y = F::mul(g, h);
    r.t = F::mul(e, h);
    r.z = F::mul(f, g);
    return r;
  }
  // add-2008-hwcd-3: 8M + 1 multiplication by k = 2d
  static MSM_HD Ext add(const Ext& p, const Ext& q) {
    El a = F::mul(F::sub(p.y, p.x), F::sub(q.y, q.x));
    El b = F::mul(F::add(p.y, p.x), F::add(q.y, q.x));
    El c = F::mul(F::mul(p.t, K::two_d()), q.t);
    El d = F::dbl(F::mul(p.z, q.z));
    El e = F::sub(b, a), f = F::sub(d, c), g = F::add(d, c), h = F::add(b, a);
    Ext r;
    r.x = F::mul(e, f);
    r.y = F::mul(g, h);
    r.t = F::mul(e, h);
    r.z = F::mul(f, g);
    return r;
  }
  // dbl-2008-hwcd with a = -1: 4M + 4S
  static MSM_HD Ext dbl(const Ext& p) {
    El a = F::sqr(p.x);
    El b = F::sqr(p.y);
    El c = F::dbl(F::sqr(p.z));
    El d = F::neg(a);
    El xy = F::add(p.x, p.y);
    El e = F::sub(F::sub(F::sqr(xy), a), b);
    El g = F::add(d, b), f = F::sub(g, c), h = F::sub(d, b);
    Ext r;
    r.x = F::mul(e, f);
    r.y = F::mul(g, h);
    r.t = F::mul(e, h);
    r.z = F::mul(f, g);
    return r;
  }
  // The same without T3 (one product less): for a doubling that is followed by another doubling, which never reads T.
  static MSM_HD Ext dbl_nt(const Ext& p) {
    El a = F::sqr(p.x);
    El b = F::sqr(p.y);
    El c = F::dbl(F::sqr(p.z));
    El d = F::neg(a);
    El xy = F::add(p.x, p.y);
    El e = F::sub(F::sub(F::sqr(xy), a), b);
    El g = F::add(d, b), f = F::sub(g, c), h = F::sub(d, b);
    Ext r;
    r.x = F::mul(e, f);
    r.y = F::mul(g, h);
    r.t = F::zero();
    r.z = F::mul(f, g);
    return r;
  }
};

struct EdK29 {
  static MSM_HD Fq::El two_d() { return Fq::from_const(EdConsts::ED_2D); }
};
using Ed = EdT<Fq, EdK29>;

}  // namespace msm377
