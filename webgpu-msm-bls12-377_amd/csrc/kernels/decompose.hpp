// Stage 1: scalars -> signed window digits (k_decompose, k_decompose_narrow, k_decompose_glv).  Replaces
// wgsl/cuzk/convert_point_coords_and_decompose_scalars.template.wgsl:100-141; model cuzk/utils.ts:66-109.
// Device code; included by sequencer.hip only.
#pragma once
#include "../curves.hpp"

namespace msm377 {
namespace {

__device__ __forceinline__ void digit_key(uint32_t biased, uint32_t& key, uint32_t& sign) {
  int d = (int)biased - 32768;
  sign = d < 0 ? 1u : 0u;
  key = (uint32_t)(d < 0 ? -d : d);
}
// A window whose digits stay small (the top window: a 253-bit scalar leaves it 13 bits) would crowd all its
// elements into a few of the 256 ranges -- regions far beyond what k_local_sort keeps in LDS.  The decomposition
// records the largest key of window 15 and the sort narrows that window's ranges by a power of two (shift s:
// 128 >> s keys per range, s <= 5) so that the keys in use still spread over the 256 regions; every other
// window slot keeps the full width (key_max = NB).
__device__ __forceinline__ uint32_t win_shift(uint32_t key_max_word) {
  if (!(key_max_word & KEY_TRACKED)) return 0;
  const uint32_t max_key = key_max_word & ~KEY_TRACKED;
  uint32_t s = 0;
  while (s < 5 && max_key < (NB >> (s + 1))) s++;
  return s;
}
__device__ __forceinline__ uint32_t key_range(uint32_t key, uint32_t s) {
  return s == 0 ? (key >= NB ? NRANGE - 1 : key / KRANGE) : key >> (7 - s);
}

// One thread per scalar: 16 signed digits d_w in [-2^15, 2^15), stored biased (d + 2^15).
// Only windows [wb, wb + wc) are written (window sharding); the carry chain always runs over
// all 16.  A final carry (scalar >= 2^255 - 2^239) sets *err, as cuzk/utils.ts:95-98 throws.
//
// even != 0 (whole MSMs, sequencer.hip Phase::even): the three bits a 253-bit scalar leaves unused are spread over the
// top THREE windows instead of emptying the last one -- windows 0..12 as above, windows 13, 14, 15 take 15 bits each
// (bit offsets 208, 223, 238: common.hpp even_offset) as UNSIGNED digits in [0, 2^15) with a carry chain of their own
// (limb + carry = 2^15 -> digit 0, carry 1).  Every window then fills its 2^15 buckets with rows of n / 2^15 entries;
// with sixteen 16-bit windows the top one holds 13 significant bits: n entries in 4096 rows, each cut into several
// work items whose partial sums a merge kernel adds back (31 us at 2^20), and a sort with narrowed ranges for that
// window (win_shift).  A scalar of 2^253 - 2^238 and more, whose top digit does not fit, raises ERR_NARROW_RANGE and
// the call reruns with the sixteen equal windows; the ERROR condition stays the one of the 16-bit recode.
__global__ void __launch_bounds__(256) k_decompose(const uint32_t* __restrict__ scalars, uint16_t* __restrict__ digits, uint64_t n,
                                                   uint32_t wb, uint32_t wc, int* __restrict__ err, uint32_t* __restrict__ top_key_max, uint32_t prio,
                                                   uint32_t even) {
  if (prio) __builtin_amdgcn_s_setprio(3);  // sequencer.hip: front-end kernels outrank the conversion beside them
  // top_key_max (may be null): largest key of window 15, the one window that scalars below a 253-bit modulus leave
  // mostly empty; see win_shift.  One LDS atomic per thread at worst, one global atomic per block.
  __shared__ uint32_t wmax;
  if (threadIdx.x == 0) wmax = 0;
  __syncthreads();
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    uint32_t w[8];
    load_words16(scalars + i * 8, w, 2);
    uint32_t carry = 0;
    if (even) {
      uint32_t carry16 = 0;  // the 16-bit recode's carry chain: the error condition
#pragma unroll
      for (uint32_t win = 0; win < 16; win++) {
        const uint32_t limb = (w[win >> 1] >> (16 * (win & 1))) & 0xffffu;
        carry16 = limb + carry16 >= 32768u ? 1u : 0u;
        uint32_t biased;
        if (win < EVEN_FROM) {
          const uint32_t v = limb + carry;
          carry = v >= 32768u ? 1u : 0u;
          biased = (v + 32768u) & 0xffffu;
        } else {
          const uint32_t raw = win == 13 ? (w[6] >> 16) & 0x7fffu : win == 14 ? (w[6] >> 31) | ((w[7] & 0x3fffu) << 1) : (w[7] >> 14) & 0x7fffu;
          const uint32_t v = raw + carry;
          carry = v >> 15;
          biased = (v & 0x7fffu) + 32768u;
        }
        if (win >= wb && win < wb + wc) digits[(size_t)(win - wb) * n + i] = (uint16_t)biased;
      }
      if (carry | (w[7] >> 29)) atomicOr(err, ERR_NARROW_RANGE);
      if (carry16) atomicOr(err, ERR_SCALAR);
    } else {
#pragma unroll
      for (uint32_t win = 0; win < 16; win++) {
        uint32_t limb = (w[win >> 1] >> (16 * (win & 1))) & 0xffffu;
        uint32_t v = limb + carry;
        carry = v >= 32768u ? 1u : 0u;
        if (win >= wb && win < wb + wc) {
          const uint32_t biased = (v + 32768u) & 0xffffu;
          digits[(size_t)(win - wb) * n + i] = (uint16_t)biased;
          if (win == 15 && top_key_max) {
            uint32_t key, sign;
            digit_key(biased, key, sign);
            if ((key | KEY_TRACKED) > wmax) atomicMax(&wmax, key | KEY_TRACKED);
          }
        }
      }
      if (carry) atomicOr(err, 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && top_key_max && wmax > *top_key_max) atomicMax(top_key_max, wmax);
}

// ---- narrow windows for small inputs (the reference switches to 4-bit windows below
//      65 536 points, src/submission/submission.ts:97,173-186) ----
// Below ~2^15 points the 16 x 32 768 buckets of the main path are mostly empty and their reduction -- 15 levels, the
// first ones streaming 134 MB of identity records -- is most of the call.  With 11-bit windows (23 windows of 2 048
// buckets) the bucket array shrinks 11-fold and the reduction loses four levels (0.28 -> 0.115 ms); the additions grow
// from 16 n to 23 n, which a small input does not notice.  Everything behind the sort runs the same kernels with
// L = 11 as their run-time bucket geometry; decomposition and sort have their own small kernels here.

// One thread per scalar: W windows of c bits.  Windows 0 .. W-2 are signed digits with a carry (|d| <= 2^(c-1)); the TOP
// window takes everything that is left WITHOUT a carry out, as an unsigned digit: a signed top window would push
// its carry into one more window whose only digits are 0 and 1 -- a single row holding a seventh of all points,
// which no segmenting saves on a small input (measured: the merge of that row alone took 2.6 ms at 2^14).  With
// c = 11 the top window starts at bit 242, so scalars below r (253 bits) leave it digits below 1 195 < 2^L = 2 048;
// a larger top digit (scalars >= 2^253) raises ERR_NARROW_RANGE and the call reruns on the 16-bit path.  All digits
// are stored biased by 2^L: d + 2^L in [0, 2^(L+1)).
// The error condition is the one of the 16-bit recode for every input size (cuzk/utils.ts:95-98 throws when the recode
// ends with a carry; with 16-bit windows that is k >= 2^255 - 2^239), whichever window width runs here.  The reference
// itself switches to 4-bit windows below 65 536 points (submission.ts:97), where the same check fires for every
// k > 0x7777...7 -- a set that differs only in NON-canonical scalars (every k < r passes both): deliberately not
// reproduced, one error condition for all sizes (tests/test_g1_parity_gpu.py::test_narrow_windows_edge_cases).
__global__ void __launch_bounds__(256) k_decompose_narrow(const uint32_t* __restrict__ scalars, uint16_t* __restrict__ digits, uint64_t n, uint32_t c,
                                                          uint32_t L, uint32_t W, int* __restrict__ err) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  load_words16(scalars + i * 8, w, 2);
  const uint32_t half = 1u << (c - 1), mask = (1u << c) - 1u, bias = 1u << L;
  uint32_t carry = 0;
  for (uint32_t win = 0; win + 1 < W; win++) {
    const uint32_t bit = win * c, word = bit >> 5, off = bit & 31;
    uint32_t v = w[word] >> off;
    if (off + c > 32 && word + 1 < 8) v |= w[word + 1] << (32 - off);
    v = (v & mask) + carry;
    carry = v >= half ? 1u : 0u;
    const int d = (int)v - (int)(carry << c);  // in [-2^(c-1), 2^(c-1))
    digits[(size_t)win * n + i] = (uint16_t)(d + (int)bias);
  }
  {  // the top window: bits (W - 1) c .. 255, unsigned, no carry out (at most 32 bits wide for the widths in use)
    const uint32_t bit = (W - 1) * c, word = bit >> 5, off = bit & 31;
    uint64_t v = w[word] >> off;
    for (uint32_t k = word + 1, sh = 32 - off; k < 8; k++, sh += 32) v |= (uint64_t)w[k] << sh;
    v += carry;
    if (v >= bias) atomicOr(err, ERR_NARROW_RANGE);
    digits[(size_t)(W - 1) * n + i] = (uint16_t)((uint32_t)(v < bias ? v : 0) + bias);
  }
  uint32_t carry16 = 0;
#pragma unroll
  for (uint32_t win = 0; win < 16; win++) carry16 = (((w[win >> 1] >> (16 * (win & 1))) & 0xffffu) + carry16) >= 32768u ? 1u : 0u;
  if (carry16) atomicOr(err, ERR_SCALAR);
}

// The even geometry in general (k_decompose's `even` mode is the instance L = 15, a = 13, W = 16 with its fields at fixed
// places): W windows of 2^L buckets over the 253 bits of a scalar below r -- the first `a` windows are SIGNED digits of
// L + 1 bits (|d| <= 2^L, carry into the next window), the other W - a are UNSIGNED digits of L bits with a carry chain
// of their own (limb + carry = 2^L -> digit 0, carry 1), and a (L + 1) + (W - a) L = 253, so that every window fills
// its 2^L buckets with rows of n / 2^L entries.  The small-input path runs it with L = 11, a = 11, W = 22 (common.hpp
// NARROW_EVEN_*): against k_decompose_narrow's 22 signed 11-bit windows + an unsigned top one, whose signed digits reach
// only half of a window's 2048 buckets, that is one window less and rows half as long -- fewer rows cut into several
// work items, a shorter merge.  Digits are stored biased by `bias` (2^L for k_small_sort).  A scalar whose top digit does
// not fit -- everything from 2^253 on and the few below that carry out of the top window -- raises ERR_NARROW_RANGE and
// the call reruns with sixteen 16-bit windows; the ERROR condition stays the 16-bit recode's (cuzk/utils.ts:95-98).
__global__ void __launch_bounds__(256) k_decompose_geom(const uint32_t* __restrict__ scalars, uint16_t* __restrict__ digits, uint64_t n, uint32_t L,
                                                        uint32_t a, uint32_t W, uint32_t bias, int* __restrict__ err) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  load_words16(scalars + i * 8, w, 2);
  uint32_t carry = 0, bit = 0;
  for (uint32_t win = 0; win < W; win++) {
    const uint32_t c = win < a ? L + 1 : L, word = bit >> 5, off = bit & 31;
    uint32_t v = w[word] >> off;
    if (off + c > 32 && word + 1 < 8) v |= w[word + 1] << (32 - off);
    v = (v & ((1u << c) - 1u)) + carry;
    uint32_t stored;
    if (win < a) {
      carry = v >> L ? 1u : 0u;  // v >= 2^L: the digit is v - 2^(L+1), in [-2^L, 0]
      stored = v + bias - (carry << c);
    } else {
      carry = v >> L;  // v = 2^L: digit 0, carry on
      stored = (v & ((1u << L) - 1u)) + bias;
    }
    digits[(size_t)win * n + i] = (uint16_t)stored;
    bit += c;
  }
  if (carry | (w[7] >> 29)) atomicOr(err, ERR_NARROW_RANGE);  // bit == 253 here
  uint32_t carry16 = 0;
#pragma unroll
  for (uint32_t win = 0; win < 16; win++) carry16 = (((w[win >> 1] >> (16 * (win & 1))) & 0xffffu) + carry16) >= 32768u ? 1u : 0u;
  if (carry16) atomicOr(err, ERR_SCALAR);
}

// out[0..NA+NB) = a * b on 32-bit words (schoolbook, carries resolved per row).
template <int NA, int NB_>
__device__ __forceinline__ void mul_words(const uint32_t* a, const uint32_t* b, uint32_t* out) {
#pragma unroll
  for (int k = 0; k < NA + NB_; k++) out[k] = 0;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < NB_; j++) {
      const uint64_t t = (uint64_t)a[i] * b[j] + out[i + j] + carry;
      out[i + j] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
    out[i + NB_] = carry;
  }
}

// Eight signed 16-bit digits of a 128-bit value v < 2^127 (top digit stays non-negative); windows
// [wb, wb + wc) are written to slots 0..wc-1.  Returns non-zero if the value does not fit (top window
// reaches 2^15).
__device__ __forceinline__ uint32_t recode128(const uint32_t* v, uint16_t* __restrict__ digits, size_t stride, size_t col, uint32_t wb,
                                              uint32_t wc) {
  uint32_t carry = 0, bad = 0;
#pragma unroll
  for (uint32_t win = 0; win < 8; win++) {
    const uint32_t limb = (v[win >> 1] >> (16 * (win & 1))) & 0xffffu;
    const uint32_t t = limb + carry;
    carry = (win < 7 && t >= 32768u) ? 1u : 0u;
    if (win >= wb && win < wb + wc) digits[(size_t)(win - wb) * stride + col] = (uint16_t)((t + 32768u) & 0xffffu);
    if (win == 7 && t >= 32768u) bad = 1u;
  }
  return bad;
}

// One thread per scalar: k -> (k1, k2) by a Barrett quotient (MU = floor(2^384 / LAMBDA), at
// most one correction), then the signed digits of k1 into column i and of k2 into column n + i of
// the 8 x 2n digit matrix.  Scalars outside the GLV range (k2 >= 2^127, i.e. k >~ 2^254) set bit 1
// of *err: the host then reruns the call on the plain 16-window path.
__global__ void __launch_bounds__(256) k_decompose_glv(const uint32_t* __restrict__ scalars, uint16_t* __restrict__ digits, uint64_t n,
                                                       uint32_t wb, uint32_t wc, int* __restrict__ err) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t k[8];
  load_words16(scalars + i * 8, k, 2);
  uint32_t prod[17];
  mul_words<8, 9>(k, GlvConsts::MU, prod);
  uint32_t q[5];
#pragma unroll
  for (int j = 0; j < 5; j++) q[j] = prod[12 + j];  // floor(k MU / 2^384): the quotient or one less
  uint32_t ql[9];
  mul_words<5, 4>(q, GlvConsts::LAMBDA, ql);
  uint32_t rem[5];
  {
    uint32_t borrow = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const uint64_t d = (uint64_t)k[j] - ql[j] - borrow;
      rem[j] = (uint32_t)d;
      borrow = (uint32_t)(d >> 32) & 1u;
    }
  }
  // rem in [0, 2 LAMBDA): one conditional correction
  uint32_t sub[5];
  uint32_t borrow = 0;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const uint64_t d = (uint64_t)rem[j] - (j < 4 ? GlvConsts::LAMBDA[j] : 0u) - borrow;
    sub[j] = (uint32_t)d;
    borrow = (uint32_t)(d >> 32) & 1u;
  }
  if (!borrow) {
#pragma unroll
    for (int j = 0; j < 5; j++) rem[j] = sub[j];
    uint32_t c = 1;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const uint64_t t = (uint64_t)q[j] + c;
      q[j] = (uint32_t)t;
      c = (uint32_t)(t >> 32);
    }
  }
  uint32_t bad = q[4] | rem[4];
  bad |= recode128(rem, digits, (size_t)2 * n, (size_t)i, wb, wc);
  bad |= recode128(q, digits, (size_t)2 * n, (size_t)(n + i), wb, wc);
  if (bad) atomicOr(err, 2);
}

}  // namespace
}  // namespace msm377
