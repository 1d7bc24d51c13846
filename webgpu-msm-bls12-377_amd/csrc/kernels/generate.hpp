// Synthetic inputs for bench.py and the tests: P_i = [a_i]G.
// Device code; included by sequencer.hip only.
#pragma once
#include "../curves.hpp"

namespace msm377 {
namespace {

// Synthetic bases: P_i = [a_i]G with a_i the (i+1)-th SplitMix64(seed) output, wire format.
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t i) {
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256, 2) k_generate_bases(uint64_t seed, uint64_t n, uint32_t* __restrict__ out_raw) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t a = splitmix64_at(seed, i);
  if (a == 0) a = 1;
  G1Affine gen;
  gen.x = Fp::from_const(G1Consts::GEN_X);
  gen.y = Fp::from_const(G1Consts::GEN_Y);
  G1XYZZ acc = g1_identity();
#pragma unroll 1
  for (int bit = 63; bit >= 0; bit--) {
    acc = g1_dbl(acc);
    if ((a >> bit) & 1) acc = g1_madd(acc, gen);
  }
  // affine: x = X * (ZZ/ZZZ)^2, y = Y / ZZZ
  Fp::El i3 = Fp::one();
#pragma unroll 1
  for (int b = G1Consts::PM2_NW * 32 - 1; b >= 0; b--) {
    i3 = Fp::sqr(i3);
    if ((G1Consts::PM2_W[b >> 5] >> (b & 31)) & 1u) i3 = Fp::mul(i3, acc.zzz);
  }
  Fp::El tt = Fp::mul(i3, acc.zz);
  Fp::El x = Fp::from_mont(Fp::mul(acc.x, Fp::sqr(tt)));
  Fp::El y = Fp::from_mont(Fp::mul(acc.y, i3));
  uint32_t w[24];
  Fp::to_words<12>(x, w);
  Fp::to_words<12>(y, w + 12);
  uint4* dst = reinterpret_cast<uint4*>(out_raw + i * 24);
#pragma unroll
  for (int k = 0; k < 6; k++) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}

// Edwards twin: P_i = [a_i]G_ed (generator: src/reference/utils/FieldMath.ts:108-109), 64-byte wire records.
__global__ void __launch_bounds__(256, 2) k_generate_bases_ed(uint64_t seed, uint64_t n, uint32_t* __restrict__ out_raw) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t a = splitmix64_at(seed, i);
  if (a == 0) a = 1;
  const Ed::Base gen = Ed::make_base(Fq::from_const(EdConsts::GEN_X), Fq::from_const(EdConsts::GEN_Y));
  Ed::Ext acc = Ed::identity();
#pragma unroll 1
  for (int bit = 63; bit >= 0; bit--) {
    acc = Ed::dbl(acc);
    if ((a >> bit) & 1) acc = Ed::madd(acc, gen);
  }
  Fq::El zi = Fq::one();
#pragma unroll 1
  for (int b = EdConsts::PM2_NW * 32 - 1; b >= 0; b--) {
    zi = Fq::sqr(zi);
    if ((EdConsts::PM2_W[b >> 5] >> (b & 31)) & 1u) zi = Fq::mul(zi, acc.z);
  }
  Fq::El x = Fq::from_mont(Fq::mul(acc.x, zi));
  Fq::El y = Fq::from_mont(Fq::mul(acc.y, zi));
  uint32_t w[16];
  Fq::to_words<8>(x, w);
  Fq::to_words<8>(y, w + 8);
  uint4* dst = reinterpret_cast<uint4*>(out_raw + i * 16);
#pragma unroll
  for (int k = 0; k < 4; k++) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}

}  // namespace
}  // namespace msm377
