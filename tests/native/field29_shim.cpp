// TEST INFRASTRUCTURE: exposes the product's field29.hpp / g1_xyzz.hpp (the very headers the
// HIP kernels compile) to ctypes on the host, so tests can check them against Python big
// integers without a GPU.  Built by tests/test_field29_host.py with g++; not shipped.
#include <stdint.h>

#include "fp64_host.hpp"
#include "g1_xyzz.hpp"

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
void shim_g1_add(const uint32_t* a52, const uint32_t* b52, uint32_t* out52) { store_xyzz(g1_add(load_xyzz(a52), load_xyzz(b52)), out52); }
void shim_g1_dbl(const uint32_t* a52, uint32_t* out52) { store_xyzz(g1_dbl(load_xyzz(a52)), out52); }
// the 64-bit host-tail field through the same curve template
void shim_g1h_add_to_wire(const uint32_t* a52, const uint32_t* b52, uint8_t* out96) {
  g1h_to_wire(G1H::add(g1h_from_device_words(a52), g1h_from_device_words(b52)), out96);
}
}
