// Wide windows for fixed-base batches with precomputed window multiples (BASELINE.json config 5; the reference lists
// precomputation as future work, /root/reference README.md:558-563).  With the table T[w][i] = [2^(off_w)] P_i every
// window's points already carry the window's weight, so ALL windows share ONE bucket set -- and the window can then
// widen without multiplying buckets: 13 windows (13 n bucket additions instead of 16 n) over 2^19 buckets, exactly as
// many as the 16 x 2^15 of the main path.  (21 bits would still be 13 windows for a 253-bit scalar -- 12 x 21 = 252 --
// and 22 bits quadruple the buckets.)
//
// Window widths: SIX 20-bit and SEVEN 19-bit windows, 6 x 20 + 7 x 19 = 253 bits (common.hpp wide_width / wide_offset).
// Thirteen 20-bit windows would leave the top one 13 significant bits: a twelfth of all entries in 1/64 of the key space
// -- rows of ~280 entries beside rows of ~24, sort regions five times the size the LDS path holds, and (measured) a
// partition pass whose workgroups for that corner ran six times longer than the rest.  With the uneven widths every
// window fills its key range; the top window is unsigned (no carry out, keys up to 2^19), and a scalar of 2^253 and
// more, whose top digit does not fit, raises ERR_NARROW_RANGE: the call reruns on the plain 16-window path over the
// table's window 0, which is the affine record of P_i itself.
//
// The sort of the N = 13 n (window, scalar) entries into 4096 fine ranges of 128 keys, two levels PER WINDOW so that
// both partition passes keep the tile-staged scatter of the main path (whole runs out of LDS, kernels/sort.hpp) -- one
// pass straight into 4096 write streams (the first version) wrote every 8-byte element as a partial line: 163 us of
// the 310 us sort at 2^20:
//   k_wide_count     workgroup (chunk, window): LDS histogram over the 4096 fine ranges -> cnt[w][c][f]
//   k_wide_sums      totals per (window, fine range), per (window, chunk, coarse range)
//   k_wide_scan      one workgroup: region_base[f] over all windows; fin[f][w] = where window w's share of range f goes
//   k_wide_offsets1  workgroup per window: start of every coarse region of the window, chunk offsets inside it
//   k_wide_part1     workgroup (chunk, window): digits -> (table record | sign, key), 256 coarse streams per window
//   k_wide_part2     workgroup (coarse range, window): its region -> 16 fine streams, each straight to fin[f][w]
//   k_local_sort_lds workgroup per fine range, as on the main path (its region is the 13 windows' shares, contiguous)
// Device code; included by sequencer.hip only.
#pragma once
#include "sort.hpp"

namespace msm377 {
namespace {

constexpr uint32_t WS_CHUNKS = 16;                    // chunks per window in the count / first partition pass
constexpr uint32_t WS_COARSE = 256;                   // coarse ranges per window: 2048 keys each
constexpr uint32_t WS_SUB = WIDE_NRANGE / WS_COARSE;  // 16 fine ranges per coarse one
constexpr uint32_t WS_TILE = 8192;                    // elements per staged tile, 8 per thread
static_assert(WS_SUB == 16 && (1u << WIDE_LOG) / WS_COARSE == 2048, "coarse range = key >> 11, fine range = key >> 7");

// Layout of the sort's counter buffer (u32 words; ctx->d_wide_counts).
constexpr size_t WC_CNT = 0;                                                        // cnt[w][c][f]
constexpr size_t WC_CNTF = WC_CNT + (size_t)WIDE_WINDOWS * WS_CHUNKS * WIDE_NRANGE;  // cntF[w][f]
constexpr size_t WC_CNT1 = WC_CNTF + (size_t)WIDE_WINDOWS * WIDE_NRANGE;             // cnt1[w][c][r1] -> off1: chunk c's cursor in coarse region (w, r1)
constexpr size_t WC_START1 = WC_CNT1 + (size_t)WIDE_WINDOWS * WS_CHUNKS * WS_COARSE;  // start1[w][r1], 257 per window
constexpr size_t WC_FIN = WC_START1 + (size_t)WIDE_WINDOWS * (WS_COARSE + 1);         // fin[f][w]
constexpr size_t WC_WORDS = WC_FIN + (size_t)WIDE_NRANGE * WIDE_WINDOWS;

// One thread per scalar: 13 digits (widths wide_width), stored biased (d + 2^19) as u32, window-major.  Windows 0..11
// are signed with a carry; the top window takes everything from bit 234 on plus the carry, unsigned.
// The error condition stays the reference's (cuzk/utils.ts:95-98 throws when the 16-bit recode ends with a carry),
// whichever width runs.
__global__ void __launch_bounds__(256) k_decompose_wide(const uint32_t* __restrict__ scalars, uint32_t* __restrict__ digits, uint64_t n, int* __restrict__ err) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  load_words16(scalars + i * 8, w, 2);
  constexpr uint32_t BIAS = 1u << WIDE_LOG;
  uint32_t carry = 0;
#pragma unroll
  for (uint32_t win = 0; win + 1 < WIDE_WINDOWS; win++) {
    const uint32_t c = wide_width(win), bit = wide_offset(win), word = bit >> 5, off = bit & 31;
    uint32_t v = w[word] >> off;
    if (off + c > 32 && word + 1 < 8) v |= w[word + 1] << (32 - off);
    v = (v & ((1u << c) - 1u)) + carry;
    carry = v >= (1u << (c - 1)) ? 1u : 0u;
    digits[(size_t)win * n + i] = v + BIAS - (carry << c);  // d + 2^19 with d = v - carry 2^c
  }
  {
    constexpr uint32_t bit = wide_offset(WIDE_WINDOWS - 1), word = bit >> 5, off = bit & 31;  // bit 234: word 7, offset 10
    static_assert(word == 7, "the top window lives in the last word");
    const uint32_t v = (w[word] >> off) + carry;  // < 2^22 + 1
    if (v > BIAS) atomicOr(err, ERR_NARROW_RANGE);  // scalars of 2^253 and more: rerun on the 16-window path
    digits[(size_t)(WIDE_WINDOWS - 1) * n + i] = (v > BIAS ? 0u : v) + BIAS;
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
__device__ __forceinline__ uint32_t wide_fine(uint32_t key) { return key >= (1u << WIDE_LOG) ? WIDE_NRANGE - 1 : key >> 7; }

// The positions [beg, end) of window w's digit column that chunk c owns.
__device__ __forceinline__ void wide_chunk_span(uint64_t n, uint32_t c, uint64_t& beg, uint64_t& end) {
  const uint64_t pc = ((n + WS_CHUNKS - 1) / WS_CHUNKS + 7) & ~7ull;
  beg = (uint64_t)c * pc;
  end = beg + pc < n ? beg + pc : n;
  if (beg > n) beg = end = n;
}
// Eight consecutive digits of a column from position i0 (those below `end`).
__device__ __forceinline__ void wide_load8(const uint32_t* __restrict__ col, uint64_t i0, uint64_t end, uint32_t* d, uint32_t& valid) {
  valid = 0;
  if (i0 + 8 <= end && (((uintptr_t)(col + i0)) & 15) == 0) {
    const uint4 a = *reinterpret_cast<const uint4*>(col + i0), b = *reinterpret_cast<const uint4*>(col + i0 + 4);
    d[0] = a.x, d[1] = a.y, d[2] = a.z, d[3] = a.w, d[4] = b.x, d[5] = b.y, d[6] = b.z, d[7] = b.w;
    valid = 0xffu;
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      d[j] = 1u << WIDE_LOG;
      if (i0 + j < end) {
        d[j] = col[i0 + j];
        valid |= 1u << j;
      }
    }
  }
}

// Workgroup (chunk c, window w): LDS histogram of the chunk over the 4096 fine ranges.
__global__ void __launch_bounds__(1024) k_wide_count(const uint32_t* __restrict__ digits, uint32_t* __restrict__ wc, uint64_t n) {
  __shared__ uint32_t cnt[WIDE_NRANGE];
  const uint32_t c = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
  for (uint32_t r = tid; r < WIDE_NRANGE; r += 1024) cnt[r] = 0;
  __syncthreads();
  uint64_t beg, end;
  wide_chunk_span(n, c, beg, end);
  const uint32_t* col = digits + (size_t)w * n;
  for (uint64_t t0 = beg; t0 < end; t0 += WS_TILE) {
    uint32_t d[8], valid;
    wide_load8(col, t0 + (uint64_t)tid * 8, end, d, valid);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      uint32_t key, sign;
      wide_key(d[j], key, sign);
      if ((valid >> j) & 1u) atomicAdd(&cnt[wide_fine(key)], 1u);
    }
  }
  __syncthreads();
  uint32_t* out = wc + WC_CNT + ((size_t)w * WS_CHUNKS + c) * WIDE_NRANGE;
  for (uint32_t r = tid; r < WIDE_NRANGE; r += 1024) out[r] = cnt[r];
}

// Workgroup (tile of 256 fine ranges, window w): cntF[w][f] = sum over the chunks; cnt1[w][c][r1] = sum over the 16 fine
// ranges of coarse range r1 (16-lane groups of the wave).
__global__ void __launch_bounds__(256) k_wide_sums(uint32_t* __restrict__ wc) {
  const uint32_t f = blockIdx.x * 256 + threadIdx.x, w = blockIdx.y;
  uint32_t tot = 0;
  for (uint32_t c = 0; c < WS_CHUNKS; c++) {
    const uint32_t v = wc[WC_CNT + ((size_t)w * WS_CHUNKS + c) * WIDE_NRANGE + f];
    tot += v;
    uint32_t s = v;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) s += (uint32_t)__shfl_xor((int)s, off, 64);  // sum of the 16-lane group
    if ((threadIdx.x & 15) == 0) wc[WC_CNT1 + ((size_t)w * WS_CHUNKS + c) * WS_COARSE + (f >> 4)] = s;
  }
  wc[WC_CNTF + (size_t)w * WIDE_NRANGE + f] = tot;
}

// One workgroup: region_base[f] = entries of all windows in smaller fine ranges (4097 bounds), fin[f][w] = region_base[f]
// + the shares of the windows before w.
__global__ void __launch_bounds__(1024) k_wide_scan(uint32_t* __restrict__ wc, uint32_t* __restrict__ region_base) {
  __shared__ uint32_t sums[1024];
  const uint32_t tid = threadIdx.x;
  constexpr uint32_t PER = WIDE_NRANGE / 1024;
  uint32_t tot[PER], sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    uint32_t t = 0;
    for (uint32_t w = 0; w < WIDE_WINDOWS; w++) t += wc[WC_CNTF + (size_t)w * WIDE_NRANGE + tid * PER + k];
    tot[k] = t;
    sum += t;
  }
  sums[tid] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    const uint32_t x = tid >= off ? sums[tid - off] : 0u;
    __syncthreads();
    sums[tid] += x;
    __syncthreads();
  }
  uint32_t run = sums[tid] - sum;
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    const uint32_t f = tid * PER + k;
    region_base[f] = run;
    uint32_t at = run;
    for (uint32_t w = 0; w < WIDE_WINDOWS; w++) {
      wc[WC_FIN + (size_t)f * WIDE_WINDOWS + w] = at;
      at += wc[WC_CNTF + (size_t)w * WIDE_NRANGE + f];
    }
    run += tot[k];
  }
  if (tid == 1023) region_base[WIDE_NRANGE] = run;
}

// Workgroup per window, thread per coarse range: start1[w][r1] (position in the first pass's output, window w's part
// starting at w n), cnt1[w][c][r1] -> chunk c's write cursor inside that region.
__global__ void __launch_bounds__(WS_COARSE) k_wide_offsets1(uint32_t* __restrict__ wc, uint64_t n) {
  __shared__ uint32_t part[WS_COARSE];
  const uint32_t w = blockIdx.x, r1 = threadIdx.x;
  uint32_t tot = 0;
  for (uint32_t c = 0; c < WS_CHUNKS; c++) tot += wc[WC_CNT1 + ((size_t)w * WS_CHUNKS + c) * WS_COARSE + r1];
  part[r1] = tot;
  __syncthreads();
  for (uint32_t off = 1; off < WS_COARSE; off <<= 1) {
    const uint32_t v = r1 >= off ? part[r1 - off] : 0u;
    __syncthreads();
    part[r1] += v;
    __syncthreads();
  }
  uint32_t run = (uint32_t)((uint64_t)w * n) + part[r1] - tot;
  wc[WC_START1 + (size_t)w * (WS_COARSE + 1) + r1] = run;
  if (r1 == WS_COARSE - 1) wc[WC_START1 + (size_t)w * (WS_COARSE + 1) + WS_COARSE] = run + tot;
  for (uint32_t c = 0; c < WS_CHUNKS; c++) {
    uint32_t* p = wc + WC_CNT1 + ((size_t)w * WS_CHUNKS + c) * WS_COARSE + r1;
    const uint32_t v = *p;
    *p = run;
    run += v;
  }
}

// One tile of the staged scatter (k_partition_staged's, for NS streams): ranks inside the tile from an LDS histogram,
// an exclusive scan over the streams, the elements parked in LDS in stream order, then written out by consecutive
// lanes.  All 1024 threads call it; e / rg / valid: the thread's up to 8 elements, their streams, which of them exist.
// cur: the workgroup's running write cursor per stream (LDS).
template <uint32_t NS, class StreamOf>
__device__ __forceinline__ void wide_scatter_tile(const SortElem* e, const uint32_t* rg, uint32_t valid, uint32_t tile_len, uint32_t* cur, uint32_t* cnt, uint32_t* toff,
                                                  uint32_t* gdelta, uint32_t* wsum, SortElem* stage, SortElem* __restrict__ out, StreamOf stream_of) {
  const uint32_t tid = threadIdx.x;
  uint32_t rk[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    rk[j] = 0;
    if ((valid >> j) & 1u) rk[j] = atomicAdd(&cnt[rg[j]], 1u);
  }
  __syncthreads();
  constexpr uint32_t SCAN_THREADS = NS < 64 ? 64 : NS;  // whole waves run the scan
  uint32_t mine = 0, incl = 0;
  if (tid < SCAN_THREADS) {
    mine = tid < NS ? cnt[tid] : 0u;
    incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64);
      if ((tid & 63) >= (uint32_t)off) incl += v;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
  }
  __syncthreads();
  if (tid < NS) {
    uint32_t before = 0;
    for (uint32_t k = 0; k < (tid >> 6); k++) before += wsum[k];
    const uint32_t excl = before + incl - mine;
    toff[tid] = excl;
    gdelta[tid] = cur[tid] - excl;
    cur[tid] += mine;
    cnt[tid] = 0;  // for the next tile (its atomics come after this tile's last barrier)
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; j++)
    if ((valid >> j) & 1u) stage[toff[rg[j]] + rk[j]] = e[j];
  __syncthreads();
  for (uint32_t p = tid; p < tile_len; p += 1024) {
    const SortElem x = stage[p];
    out[gdelta[stream_of(x.key)] + p] = x;
  }
  __syncthreads();
}

// First partition pass, workgroup (chunk c, window w): the chunk's digits become (table record | sign, key) elements in
// the window's 256 coarse regions.  Record = w * stride + i (stride = points in the resident table, >= n: a call may use a
// prefix of the bases).
__global__ void __launch_bounds__(1024) k_wide_part1(const uint32_t* __restrict__ digits, const uint32_t* __restrict__ wc, SortElem* __restrict__ temp1, uint64_t n,
                                                     uint32_t stride) {
  __shared__ uint32_t cur[WS_COARSE], cnt[WS_COARSE], toff[WS_COARSE], gdelta[WS_COARSE], wsum[WS_COARSE / 64];
  __shared__ SortElem stage[WS_TILE];
  const uint32_t c = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
  if (tid < WS_COARSE) {
    cur[tid] = wc[WC_CNT1 + ((size_t)w * WS_CHUNKS + c) * WS_COARSE + tid];
    cnt[tid] = 0;
  }
  __syncthreads();
  uint64_t beg, end;
  wide_chunk_span(n, c, beg, end);
  const uint32_t* col = digits + (size_t)w * n;
  auto coarse_of = [](uint32_t key) { return wide_fine(key) >> 4; };
  for (uint64_t t0 = beg; t0 < end; t0 += WS_TILE) {
    uint32_t d[8], valid, rg[8];
    SortElem e[8];
    const uint64_t i0 = t0 + (uint64_t)tid * 8;
    wide_load8(col, i0, end, d, valid);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      uint32_t key, sign;
      wide_key(d[j], key, sign);
      e[j] = SortElem{(w * stride + (uint32_t)(i0 + j)) | (sign << 31), key};
      rg[j] = coarse_of(key);
    }
    const uint32_t tile_len = (uint32_t)((end - t0 < WS_TILE) ? end - t0 : WS_TILE);
    wide_scatter_tile<WS_COARSE>(e, rg, valid, tile_len, cur, cnt, toff, gdelta, wsum, stage, temp1, coarse_of);
  }
}

// Second partition pass, workgroup (coarse range r1, window w): the region's elements go to their fine range's slot for
// this window, fin[f][w] -- this workgroup is the only writer of those 16 slots.
__global__ void __launch_bounds__(1024) k_wide_part2(const SortElem* __restrict__ temp1, const uint32_t* __restrict__ wc, SortElem* __restrict__ temp2) {
  __shared__ uint32_t cur[WS_SUB], cnt[WS_SUB], toff[WS_SUB], gdelta[WS_SUB], wsum[1];
  __shared__ SortElem stage[WS_TILE];
  const uint32_t r1 = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
  if (tid < WS_SUB) {
    cur[tid] = wc[WC_FIN + (size_t)(r1 * WS_SUB + tid) * WIDE_WINDOWS + w];
    cnt[tid] = 0;
  }
  __syncthreads();
  const uint64_t beg = wc[WC_START1 + (size_t)w * (WS_COARSE + 1) + r1], end = wc[WC_START1 + (size_t)w * (WS_COARSE + 1) + r1 + 1];
  auto sub_of = [](uint32_t key) { return wide_fine(key) & (WS_SUB - 1); };
  for (uint64_t t0 = beg; t0 < end; t0 += WS_TILE) {
    uint32_t valid = 0, rg[8];
    SortElem e[8];
    const uint64_t i0 = t0 + (uint64_t)tid * 8;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      e[j] = SortElem{0u, 0u};
      if (i0 + j < end) {
        e[j] = temp1[i0 + j];
        valid |= 1u << j;
      }
      rg[j] = sub_of(e[j].key);
    }
    const uint32_t tile_len = (uint32_t)((end - t0 < WS_TILE) ? end - t0 : WS_TILE);
    wide_scatter_tile<WS_SUB>(e, rg, valid, tile_len, cur, cnt, toff, gdelta, wsum, stage, temp2, sub_of);
  }
}

}  // namespace
}  // namespace msm377
