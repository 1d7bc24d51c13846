// Wide windows for fixed-base batches with precomputed window multiples (BASELINE.json config 5; the reference lists
// precomputation as future work, /root/reference README.md:558-563).  With the table T[w][i] = [2^(20 w)] P_i every
// window's points already carry the window's weight, so ALL windows share ONE bucket set -- and the window can then
// widen without multiplying buckets: 20-bit signed windows are 13 windows (13 n bucket additions instead of 16 n)
// over 2^19 buckets, exactly as many as the 16 x 2^15 of the main path.  (21 bits would still be 13 windows for a
// 253-bit scalar -- 12 x 21 = 252 -- and 22 bits quadruple the buckets.)
//
// The 13 digit columns form ONE flat list of N = 13 n entries; position w n + i names the table record w * stride + i to
// gather, so the sort is a single counting sort of N entries by key |d| in 0 .. 2^19: 4096 coarse ranges of 128 keys (the
// second level is the main path's k_local_sort_lds, one workgroup per range), chunk-major counters so that the scan
// over 4096 x chunks counters is three small coalesced kernels.
// Device code; included by sequencer.hip only.
#pragma once
#include "sort.hpp"

namespace msm377 {
namespace {

// One thread per scalar: 13 signed 20-bit digits d_w in [-2^19, 2^19), stored biased (d + 2^19) as u32, window-major.
// The top window holds bits 240..255 plus a carry: below 2^17, never a carry out.  The error condition stays the
// reference's (cuzk/utils.ts:95-98 throws when the 16-bit recode ends with a carry), whichever width runs.
__global__ void __launch_bounds__(256) k_decompose_wide(const uint32_t* __restrict__ scalars, uint32_t* __restrict__ digits, uint64_t n, int* __restrict__ err) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  load_words16(scalars + i * 8, w, 2);
  constexpr uint32_t C = WIDE_BITS, HALF = 1u << (C - 1), MASK = (1u << C) - 1u;
  uint32_t carry = 0;
#pragma unroll
  for (uint32_t win = 0; win < WIDE_WINDOWS; win++) {
    const uint32_t bit = win * C, word = bit >> 5, off = bit & 31;
    uint32_t v = w[word] >> off;
    if (off + C > 32 && word + 1 < 8) v |= w[word + 1] << (32 - off);
    v = (v & MASK) + carry;
    carry = v >= HALF ? 1u : 0u;
    digits[(size_t)win * n + i] = v + HALF - (carry << C);  // d + 2^19 with d = v - carry 2^20
  }
  uint32_t carry16 = 0;
#pragma unroll
  for (uint32_t win = 0; win < 16; win++) carry16 = (((w[win >> 1] >> (16 * (win & 1))) & 0xffffu) + carry16) >= 32768u ? 1u : 0u;
  if (carry16) atomicOr(err, ERR_SCALAR);
}

__device__ __forceinline__ void wide_key(uint32_t biased, uint32_t& key, uint32_t& sign) {
  const int d = (int)biased - (int)(1u << WIDE_LOG);
  sign = d < 0 ? 1u : 0u;
  key = (uint32_t)(d < 0 ? -d : d);
}
__device__ __forceinline__ uint32_t wide_range(uint32_t key) { return key >= (1u << WIDE_LOG) ? WIDE_NRANGE - 1 : key >> 7; }

// f(position, biased digit) for positions [beg, end) of the flat digit list, four digits per 16-byte load.
template <class F>
__device__ __forceinline__ void for_each_digit32(const uint32_t* __restrict__ dg, uint64_t beg, uint64_t end, uint32_t tid, uint32_t nthreads, F f) {
  uint64_t head = beg;
  while (head < end && (((uintptr_t)(dg + head)) & 15)) head++;
  for (uint64_t i = beg + tid; i < head; i += nthreads) f(i, dg[i]);
  const uint64_t groups = (end - head) / 4;
  const uint4* v = reinterpret_cast<const uint4*>(dg + head);
  for (uint64_t g = tid; g < groups; g += nthreads) {
    const uint4 q = v[g];
    f(head + g * 4 + 0, q.x);
    f(head + g * 4 + 1, q.y);
    f(head + g * 4 + 2, q.z);
    f(head + g * 4 + 3, q.w);
  }
  for (uint64_t i = head + groups * 4 + tid; i < end; i += nthreads) f(i, dg[i]);
}

// Workgroup per chunk: LDS histogram over the 4096 ranges; counts[chunk][range].
__global__ void __launch_bounds__(1024) k_wide_count(const uint32_t* __restrict__ digits, uint32_t* __restrict__ counts, uint64_t N, uint64_t per_chunk) {
  __shared__ uint32_t cnt[WIDE_NRANGE];
  const uint32_t c = blockIdx.x, tid = threadIdx.x;
  for (uint32_t r = tid; r < WIDE_NRANGE; r += 1024) cnt[r] = 0;
  __syncthreads();
  const uint64_t beg = (uint64_t)c * per_chunk;
  const uint64_t end = (beg + per_chunk < N) ? beg + per_chunk : N;
  for_each_digit32(digits, beg, end, tid, 1024, [&](uint64_t, uint32_t biased) {
    uint32_t key, sign;
    wide_key(biased, key, sign);
    atomicAdd(&cnt[wide_range(key)], 1u);
  });
  __syncthreads();
  for (uint32_t r = tid; r < WIDE_NRANGE; r += 1024) counts[(size_t)c * WIDE_NRANGE + r] = cnt[r];
}

// Thread per range: its total over the chunks (coalesced across the ranges).
__global__ void __launch_bounds__(256) k_wide_total(const uint32_t* __restrict__ counts, uint32_t* __restrict__ tot, uint32_t chunks) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  uint32_t t = 0;
  for (uint32_t c = 0; c < chunks; c++) t += counts[(size_t)c * WIDE_NRANGE + r];
  tot[r] = t;
}
// One workgroup: region_base[r] = entries in smaller ranges; region_base[WIDE_NRANGE] = N.
__global__ void __launch_bounds__(1024) k_wide_scan(const uint32_t* __restrict__ tot, uint32_t* __restrict__ region_base) {
  __shared__ uint32_t part[1024];
  const uint32_t tid = threadIdx.x;
  constexpr uint32_t PER = WIDE_NRANGE / 1024;
  uint32_t v[PER], sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    v[k] = tot[tid * PER + k];
    sum += v[k];
  }
  part[tid] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    const uint32_t x = tid >= off ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += x;
    __syncthreads();
  }
  uint32_t run = part[tid] - sum;
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    region_base[tid * PER + k] = run;
    run += v[k];
  }
  if (tid == 1023) region_base[WIDE_NRANGE] = run;
}
// Thread per range: counts[c][r] becomes the write offset of chunk c inside region r (absolute).
__global__ void __launch_bounds__(256) k_wide_offsets(uint32_t* __restrict__ counts, const uint32_t* __restrict__ region_base, uint32_t chunks) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  uint32_t run = region_base[r];
  for (uint32_t c = 0; c < chunks; c++) {
    const uint32_t v = counts[(size_t)c * WIDE_NRANGE + r];
    counts[(size_t)c * WIDE_NRANGE + r] = run;
    run += v;
  }
}

// Workgroup per chunk: appends each entry (table record | sign << 31, key) to its range's region.  Position w n + i of
// the digit list is window w of scalar i; the record to gather is w * stride + i (stride = points in the resident
// table, >= n: a call may use a prefix of the bases).
__global__ void __launch_bounds__(1024) k_wide_partition(const uint32_t* __restrict__ digits, const uint32_t* __restrict__ counts, SortElem* __restrict__ temp, uint64_t N,
                                                         uint64_t per_chunk, uint32_t n, uint32_t stride) {
  __shared__ uint32_t cur[WIDE_NRANGE];
  const uint32_t c = blockIdx.x, tid = threadIdx.x;
  for (uint32_t r = tid; r < WIDE_NRANGE; r += 1024) cur[r] = counts[(size_t)c * WIDE_NRANGE + r];
  __syncthreads();
  const uint64_t beg = (uint64_t)c * per_chunk;
  const uint64_t end = (beg + per_chunk < N) ? beg + per_chunk : N;
  for_each_digit32(digits, beg, end, tid, 1024, [&](uint64_t i, uint32_t biased) {
    uint32_t key, sign;
    wide_key(biased, key, sign);
    const uint32_t w = (uint32_t)i / n;
    temp[atomicAdd(&cur[wide_range(key)], 1u)] = SortElem{((uint32_t)i + w * (stride - n)) | (sign << 31), key};
  });
}

}  // namespace
}  // namespace msm377
