// TEST INFRASTRUCTURE: exposes the product's field29.hpp / g1_xyzz.hpp (the very headers the
// HIP kernels compile) to ctypes on the host, so tests can check them against Python big
// integers without a GPU.  Built by tests/test_field29_host.py with g++; not shipped.
#include <stdint.h>

#include "fp64_host.hpp"
#include "g1_xyzz.hpp"
#include "te377.hpp"

using namespace msm377;

static Fp::El load(const uint32_t* w12) { return Fp::to_mont(Fp::from_words<12>(w12)); }
static void store(const Fp::El& a, uint32_t* w12) { Fp::to_words<12>(Fp::from_mont(a), w12); }
static void store_xyzz(const G1XYZZ& p, uint32_t* w52) {
  for (int j = 0; j < 13; j++) {
    w52[j] = p.x.l[j];
    w52[13 + j] = p.y.l[j];
    w52[26 + j] = p.zz.l[j];
    w52[39 + j] = p.zzz.l[j];
  }
}
static G1XYZZ load_xyzz(const uint32_t* w52) {
  G1XYZZ p;
  for (int j = 0; j < 13; j++) {
    p.x.l[j] = w52[j];
    p.y.l[j] = w52[13 + j];
    p.zz.l[j] = w52[26 + j];
    p.zzz.l[j] = w52[39 + j];
  }
  return p;
}

extern "C" {

// canonical 12-word little-endian operands -> canonical results
void shim_fp_ops(const uint32_t* a, const uint32_t* b, uint32_t* mul, uint32_t* add, uint32_t* sub, uint32_t* sqr, uint32_t* neg) {
  Fp::El x = load(a), y = load(b);
  store(Fp::mul(x, y), mul);
  store(Fp::add(x, y), add);
  store(Fp::sub(x, y), sub);
  store(Fp::sqr(x), sqr);
  store(Fp::neg(x), neg);
}
// a*b - c*d through the fused single-reduction path
void shim_fp_mul_sub_mul(const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* out) {
  store(Fp::mul_sub_mul(load(a), load(b), load(c), load(d)), out);
}
// raw Montgomery limbs of to_mont(a): checks the limb format itself
void shim_fp_to_mont_limbs(const uint32_t* a, uint32_t* limbs13) {
  Fp::El x = load(a);
  for (int j = 0; j < 13; j++) limbs13[j] = x.l[j];
}
// acc (52 device words) (+) affine q (24 canonical words): madd; acc (+) acc2: add; 2*acc: dbl
void shim_g1_madd(const uint32_t* acc52, const uint32_t* q24, uint32_t* out52) {
  G1Affine q;
  q.x = load(q24);
  q.y = load(q24 + 12);
  store_xyzz(g1_madd(load_xyzz(acc52), q), out52);
}
// acc += (+-) q_k for k = 0..count-1, the accumulator fed back in its stored (lazy) form like k_accumulate does
void shim_g1_madd_chain(const uint32_t* acc52, const uint32_t* q24s, const uint8_t* negs, uint32_t count, uint32_t* out52) {
  G1XYZZ acc = load_xyzz(acc52);
  for (uint32_t k = 0; k < count; k++) {
    G1Affine q;
    q.x = load(q24s + 24 * k);
    q.y = load(q24s + 24 * k + 12);
    acc = g1_madd(acc, q, negs[k] != 0);
  }
  store_xyzz(acc, out52);
}
// tree of general additions over stored (lazy) points: out = sum of count points
void shim_g1_add_chain(const uint32_t* pts52, uint32_t count, uint32_t* out52) {
  G1XYZZ acc = load_xyzz(pts52);
  for (uint32_t k = 1; k < count; k++) acc = g1_add(load_xyzz(pts52 + 52 * k), acc);
  store_xyzz(acc, out52);
}
// ---- twisted Edwards form of G1 (csrc/te377.hpp) ----
static TeH::Ext te_to_host(const Te377::Ext& p) {
  TeH::Ext h;
  h.x = Fp64::from_limbs29_mont(p.x.l);
  h.y = Fp64::from_limbs29_mont(p.y.l);
  h.t = Fp64::from_limbs29_mont(p.t.l);
  h.z = Fp64::from_limbs29_mont(p.z.l);
  return h;
}
// sum of (+-) wire points through from_wire + madd (k_accumulate's loop), optionally split into `parts` partial sums
// that are merged with the general addition; result in the Weierstrass wire format.  Returns 1 if any step was flagged
// exceptional (is_bad / unrepresentable input).
int shim_te_sum(const uint32_t* q24s, const uint8_t* negs, uint32_t count, uint32_t parts, int phi, uint8_t* out96, uint32_t* ext52) {
  int bad = 0;
  Te377::Ext total = Te377::identity();
  for (uint32_t part = 0; part < parts; part++) {
    Te377::Ext acc = Te377::identity();
    for (uint32_t k = part; k < count; k += parts) {
      const Te377::PBase b = Te377::from_wire(q24s + 24 * k, q24s + 24 * k + 12, phi != 0);
      bad |= Fp::is_zero(b.z2);
      acc = k == part ? Te377::from_base(b, negs[k] != 0) : Te377::madd(acc, b, negs[k] != 0);  // k_accumulate: the first entry needs no addition
      bad |= Te377::is_bad(acc);
    }
    total = Te377::add(total, acc);
    bad |= Te377::is_bad(total);
  }
  for (int j = 0; j < 13; j++) {
    ext52[j] = total.x.l[j];
    ext52[13 + j] = total.y.l[j];
    ext52[26 + j] = total.t.l[j];
    ext52[39 + j] = total.z.l[j];
  }
  teh_to_wire(te_to_host(total), out96);
  return bad;
}
// the same sum through the affine records of resident tables (affine_from_wire + madd_affine, 7 products)
int shim_te_sum_affine(const uint32_t* q24s, const uint8_t* negs, uint32_t count, uint8_t* out96) {
  int bad = 0;
  Te377::Ext acc = Te377::identity();
  for (uint32_t k = 0; k < count; k++) {
    bool b0;
    const Te377::ABase b = Te377::affine_from_wire(q24s + 24 * k, q24s + 24 * k + 12, b0);
    bad |= b0;
    acc = k == 0 ? Te377::from_base_affine(b, negs[k] != 0) : Te377::madd_affine(acc, b, negs[k] != 0);
    bad |= Te377::is_bad(acc);
  }
  teh_to_wire(te_to_host(acc), out96);
  return bad;
}
// [k] P on the host tail's field (TeH doubling and addition), P given in wire format
void shim_teh_scalar_mul(const uint32_t* q24, const uint32_t* k8, uint8_t* out96) {
  const Te377::PBase b = Te377::from_wire(q24, q24 + 12, false);
  const TeH::Ext p = te_to_host(Te377::madd(Te377::identity(), b, false));
  TeH::Ext acc = TeH::identity();
  for (int i = 255; i >= 0; i--) {
    acc = TeH::dbl(acc);
    if ((k8[i >> 5] >> (i & 31)) & 1u) acc = TeH::add(acc, p);
  }
  teh_to_wire(acc, out96);
}
// Host-tail multiplier: the BMI2 / ADX assembly (when the CPU has it) against the portable unsigned __int128 form, on raw
// 6 x 64-bit operands (any values below 2^384: both forms take the same inputs).  Returns 0 when the CPU lacks ADX.
int shim_fp64_mul_both(const uint64_t* a6, const uint64_t* b6, uint64_t* out_adx6, uint64_t* out_portable6) {
  Fp64::El a, b;
  for (int i = 0; i < 6; i++) {
    a.v[i] = a6[i];
    b.v[i] = b6[i];
  }
  const Fp64::El rp = Fp64::mul_portable(a, b);
  for (int i = 0; i < 6; i++) out_portable6[i] = rp.v[i];
#if defined(MSM377_HAVE_ADX_MUL)
  if (!host_has_adx()) return 0;
  uint64_t t[7];
  mont_mul6_adx(t, a.v, b.v, G1Consts64::MOD, G1Consts64::N0);
  if (t[6] || Fp64::geq_p(t)) Fp64::sub_p(t);
  for (int i = 0; i < 6; i++) out_adx6[i] = t[i];
  return 1;
#else
  return 0;
#endif
}
// Host-field inversion: the plain integer inverse from the divstep iteration (fp64_host.hpp modinv62; returns 0 if it
// did not come out), and both Montgomery-form inverses -- inv (divsteps, checked, Fermat as the fallback) and
// inv_fermat -- of the same operand.  which = 0: the 377-bit base field (6 words), 1: the Edwards-BLS12 base field (4).
int shim_fp64_inv(int which, const uint64_t* a, uint64_t* plain_inv, uint64_t* mont_inv, uint64_t* mont_inv_fermat) {
  if (which == 0) {
    Fp64::El x;
    for (int i = 0; i < 6; i++) x.v[i] = a[i];
    const int ok = Fp64::modinv62(x.v, plain_inv) ? 1 : 0;
    const Fp64::El r = Fp64::inv(x), f = Fp64::inv_fermat(x);
    for (int i = 0; i < 6; i++) mont_inv[i] = r.v[i], mont_inv_fermat[i] = f.v[i];
    return ok;
  }
  Fq64::El x;
  for (int i = 0; i < 4; i++) x.v[i] = a[i];
  const int ok = Fq64::modinv62(x.v, plain_inv) ? 1 : 0;
  const Fq64::El r = Fq64::inv(x), f = Fq64::inv_fermat(x);
  for (int i = 0; i < 4; i++) mont_inv[i] = r.v[i], mont_inv_fermat[i] = f.v[i];
  return ok;
}
// The host tail's Horner chain over window records in a given window geometry (fp64_host.hpp teh_combine / g1h_combine /
// tail_position): num_windows records of 16 points x 48 words, windows of cbits bits with `planes` bit-plane sums each,
// the windows from short_from on one bit shorter (0: all alike).  form 0: Edwards records (returns 1 on an exceptional
// case), 1: Weierstrass records (16-bit windows only).  pieces > 1: the chain cut the way the threaded tail cuts it.
int shim_tail_combine_geom(const uint32_t* partials, int num_windows, int cbits, int planes, int short_from, int form, uint8_t* out96) {
  if (form == 1) {
    g1h_combine(partials, num_windows, out96, short_from);
    return 0;
  }
  return teh_combine(partials, num_windows, out96, cbits, planes, short_from) ? 1 : 0;
}
int shim_tail_positions(int num_windows, int cbits, int short_from) { return tail_positions(num_windows, cbits, short_from); }
void shim_g1_add(const uint32_t* a52, const uint32_t* b52, uint32_t* out52) { store_xyzz(g1_add(load_xyzz(a52), load_xyzz(b52)), out52); }
void shim_g1_dbl(const uint32_t* a52, uint32_t* out52) { store_xyzz(g1_dbl(load_xyzz(a52)), out52); }
// the 64-bit host-tail field through the same curve template
void shim_g1h_add_to_wire(const uint32_t* a52, const uint32_t* b52, uint8_t* out96) {
  g1h_to_wire(G1H::add(g1h_from_device_words(a52), g1h_from_device_words(b52)), out96);
}
}
