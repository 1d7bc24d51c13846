// Wide windows for fixed-base batches with precomputed window multiples (BASELINE.json config 5; the reference lists
// precomputation as future work, /root/reference README.md:558-563).  With the table T[w][i] = [2^(20 w)] P_i every
// window's points already carry the window's weight, so ALL windows share ONE bucket set -- and the window can then
// widen without multiplying buckets: 20-bit signed windows are 13 windows (13 n bucket additions instead of 16 n)
// over 2^19 buckets, exactly as many as the 16 x 2^15 of the main path.  (21 bits would still be 13 windows for a
// 253-bit scalar -- 12 x 21 = 252 -- and 22 bits quadruple the buckets.)
//
// The 13 digit columns form ONE flat list of N = 13 n entries; position w n + i names the table record w * stride + i to
// gather, so the sort is a single counting sort of N entries by key |d| in 0 .. 2^19: 4096 ranges of 128 keys reached by
// two radix-64 partition passes, then the main path's k_local_sort_lds, one workgroup per range.
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

// ---- the sort of the N = 13 n entries into 4096 ranges of 128 keys: two radix-64 passes, each staged through LDS ----
//
// A single pass straight into 4096 write streams per workgroup (the first version) wrote every 8-byte element as a
// partial line: sort stage 0.33 ms at 2^20 against 0.11 for the main path's sort of 16 n entries.  Two passes of 64
// streams each keep the tile-staged scatter of k_partition_staged -- a tile of 8192 elements holds ~128 per stream, so
// every run is a kilobyte of whole lines:
//   pass A  digits -> (record | sign, key) elements, partitioned by the top 6 bits of the range (coarse regions of 8192
//           keys); workgroup per chunk of the digit list
//   pass B  every coarse region partitioned by the low 6 bits of the range; workgroup per (quarter of a region, region)
//   then    k_local_sort_lds over the 4096 regions, as on the main path.
constexpr uint32_t WP_R = 64;         // streams per pass
constexpr uint32_t WP_TILE = 8192;    // elements per tile, 8 per thread
constexpr uint32_t WP_BCHUNKS = 4;    // workgroups per coarse region in pass B
static_assert(WP_R * WP_R == WIDE_NRANGE, "two radix-64 passes make the 4096 ranges");

__device__ __forceinline__ uint32_t wide_fine_range(uint32_t key) { return key >= (1u << WIDE_LOG) ? WIDE_NRANGE - 1 : key >> 7; }

// Where a pass takes its elements from.  load8(i0, end, e, valid): elements i0 .. i0+7 (those below `end`).
struct WideFromDigits {  // pass A: position w n + i of the digit list -> record w * stride + i (a call may use a prefix of the table)
  const uint32_t* dg;
  uint32_t n, stride;
  static constexpr uint32_t SHIFT = 6;  // stream = fine range >> 6
  __device__ __forceinline__ void load8(uint64_t i0, uint64_t end, SortElem* e, uint32_t& valid) const {
    uint32_t d[8];
    valid = 0;
    if (i0 + 8 <= end && (((uintptr_t)(dg + i0)) & 15) == 0) {
      const uint4 a = *reinterpret_cast<const uint4*>(dg + i0), b = *reinterpret_cast<const uint4*>(dg + i0 + 4);
      d[0] = a.x, d[1] = a.y, d[2] = a.z, d[3] = a.w, d[4] = b.x, d[5] = b.y, d[6] = b.z, d[7] = b.w;
      valid = 0xffu;
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        d[j] = 1u << WIDE_LOG;
        if (i0 + j < end) {
          d[j] = dg[i0 + j];
          valid |= 1u << j;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      uint32_t key, sign;
      wide_key(d[j], key, sign);
      const uint32_t pos = (uint32_t)(i0 + j), w = pos / n;
      e[j] = SortElem{(pos + w * (stride - n)) | (sign << 31), key};
    }
  }
};
struct WideFromElems {  // pass B: the elements pass A wrote
  const SortElem* in;
  static constexpr uint32_t SHIFT = 0;  // stream = fine range & 63
  __device__ __forceinline__ void load8(uint64_t i0, uint64_t end, SortElem* e, uint32_t& valid) const {
    valid = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      e[j] = SortElem{0u, 0u};
      if (i0 + j < end) {
        e[j] = in[i0 + j];
        valid |= 1u << j;
      }
    }
  }
};
template <class SRC>
__device__ __forceinline__ uint32_t wide_stream(uint32_t key) {
  const uint32_t f = wide_fine_range(key);
  return SRC::SHIFT ? f >> SRC::SHIFT : f & (WP_R - 1);
}
// The elements [beg, end) a workgroup of a pass owns: pass A cuts the digit list into gridDim.x chunks, pass B cuts
// coarse region blockIdx.y (regionA: the 65 region bounds pass A's scan left) into WP_BCHUNKS pieces.
__device__ __forceinline__ void wide_block_span(const uint32_t* __restrict__ regionA, uint64_t N, uint64_t per_chunk, uint64_t& beg, uint64_t& end) {
  if (regionA) {
    const uint64_t rb = regionA[blockIdx.y], re = regionA[blockIdx.y + 1];
    const uint64_t pc = (((re - rb) + WP_BCHUNKS - 1) / WP_BCHUNKS + 7) & ~7ull;
    beg = rb + blockIdx.x * pc;
    end = beg + pc < re ? beg + pc : re;
    if (beg > re) beg = end = re;
  } else {
    beg = (uint64_t)blockIdx.x * per_chunk;
    end = beg + per_chunk < N ? beg + per_chunk : N;
    if (beg > N) beg = end = N;
  }
}

// counts[(blockIdx.y * gridDim.x + blockIdx.x) * 64 + stream]: elements of this workgroup's span per stream.
template <class SRC>
__global__ void __launch_bounds__(1024) k_wide_count(SRC src, const uint32_t* __restrict__ regionA, uint32_t* __restrict__ counts, uint64_t N, uint64_t per_chunk) {
  __shared__ uint32_t cnt[WP_R];
  const uint32_t tid = threadIdx.x;
  if (tid < WP_R) cnt[tid] = 0;
  __syncthreads();
  uint64_t beg, end;
  wide_block_span(regionA, N, per_chunk, beg, end);
  for (uint64_t t0 = beg; t0 < end; t0 += WP_TILE) {
    SortElem e[8];
    uint32_t valid;
    src.load8(t0 + (uint64_t)tid * 8, end, e, valid);
#pragma unroll
    for (int j = 0; j < 8; j++)
      if ((valid >> j) & 1u) atomicAdd(&cnt[wide_stream<SRC>(e[j].key)], 1u);
  }
  __syncthreads();
  if (tid < WP_R) counts[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * WP_R + tid] = cnt[tid];
}

// Pass A's scan, one workgroup: regionA[r] = entries in coarser streams (65 bounds), counts[c][r] -> chunk c's write
// offset in stream r.  Thread (q, r): 16 chunk groups x 64 streams.
__global__ void __launch_bounds__(1024) k_wide_scan_a(uint32_t* __restrict__ counts, uint32_t* __restrict__ regionA, uint32_t chunks) {
  __shared__ uint32_t part[16][WP_R];
  const uint32_t tid = threadIdx.x, r = tid & 63, q = tid >> 6;
  const uint32_t per = (chunks + 15) / 16, c0 = q * per, c1 = (c0 + per < chunks) ? c0 + per : chunks;
  uint32_t sum = 0;
  for (uint32_t c = c0; c < c1; c++) sum += counts[(size_t)c * WP_R + r];
  part[q][r] = sum;
  __syncthreads();
  if (q == 0) {  // one wave: totals per stream, exclusive scan over the streams, then each group's start
    uint32_t tot = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) tot += part[k][r];
    uint32_t incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = __shfl_up((int)incl, off, 64);
      if (r >= (uint32_t)off) incl += v;
    }
    uint32_t run = incl - tot;
    regionA[r] = run;
    if (r == 63) regionA[WP_R] = incl;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const uint32_t v = part[k][r];
      part[k][r] = run;
      run += v;
    }
  }
  __syncthreads();
  uint32_t run = part[q][r];
  for (uint32_t c = c0; c < c1; c++) {
    const uint32_t v = counts[(size_t)c * WP_R + r];
    counts[(size_t)c * WP_R + r] = run;
    run += v;
  }
}
// Pass B's scan, one wave per coarse region s: region_base[s * 64 + r] (and the end sentinel), counts[s][c][r] -> offsets.
__global__ void __launch_bounds__(WP_R) k_wide_scan_b(uint32_t* __restrict__ counts, const uint32_t* __restrict__ regionA, uint32_t* __restrict__ region_base) {
  const uint32_t s = blockIdx.x, r = threadIdx.x;
  uint32_t v[WP_BCHUNKS], tot = 0;
#pragma unroll
  for (uint32_t c = 0; c < WP_BCHUNKS; c++) {
    v[c] = counts[((size_t)s * WP_BCHUNKS + c) * WP_R + r];
    tot += v[c];
  }
  uint32_t incl = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t x = __shfl_up((int)incl, off, 64);
    if (r >= (uint32_t)off) incl += x;
  }
  uint32_t run = regionA[s] + incl - tot;
  region_base[s * WP_R + r] = run;
  if (s == WP_R - 1 && r == WP_R - 1) region_base[WIDE_NRANGE] = regionA[WP_R];
#pragma unroll
  for (uint32_t c = 0; c < WP_BCHUNKS; c++) {
    counts[((size_t)s * WP_BCHUNKS + c) * WP_R + r] = run;
    run += v[c];
  }
}

// The scatter of a pass, a tile at a time through LDS (k_partition_staged with 64 streams).
template <class SRC>
__global__ void __launch_bounds__(1024) k_wide_scatter(SRC src, const uint32_t* __restrict__ regionA, const uint32_t* __restrict__ counts, SortElem* __restrict__ out, uint64_t N,
                                                       uint64_t per_chunk) {
  __shared__ uint32_t cur[WP_R], cnt[WP_R], toff[WP_R], gdelta[WP_R];
  __shared__ SortElem stage[WP_TILE];
  const uint32_t tid = threadIdx.x;
  if (tid < WP_R) {
    cur[tid] = counts[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * WP_R + tid];
    cnt[tid] = 0;
  }
  __syncthreads();
  uint64_t beg, end;
  wide_block_span(regionA, N, per_chunk, beg, end);
  for (uint64_t t0 = beg; t0 < end; t0 += WP_TILE) {
    SortElem e[8];
    uint32_t valid, rg[8], rk[8];
    src.load8(t0 + (uint64_t)tid * 8, end, e, valid);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      rg[j] = wide_stream<SRC>(e[j].key);
      rk[j] = 0;
      if ((valid >> j) & 1u) rk[j] = atomicAdd(&cnt[rg[j]], 1u);
    }
    __syncthreads();
    if (tid < WP_R) {  // one wave: exclusive scan of the tile's 64 counts
      const uint32_t mine = cnt[tid];
      uint32_t incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up((int)incl, off, 64);
        if (tid >= (uint32_t)off) incl += v;
      }
      toff[tid] = incl - mine;
      gdelta[tid] = cur[tid] - (incl - mine);
      cur[tid] += mine;
      cnt[tid] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++)
      if ((valid >> j) & 1u) stage[toff[rg[j]] + rk[j]] = e[j];
    __syncthreads();
    const uint32_t tile_len = (uint32_t)((end - t0 < WP_TILE) ? end - t0 : WP_TILE);
    for (uint32_t p = tid; p < tile_len; p += 1024) {
      const SortElem x = stage[p];
      out[gdelta[wide_stream<SRC>(x.key)] + p] = x;
    }
    __syncthreads();
  }
}

}  // namespace
}  // namespace msm377
