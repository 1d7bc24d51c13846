// Stage 2: per-window counting sort of the digits -> CSR rows (the reference's transpose, wgsl/cuzk/transpose_serial.wgsl:34-76;
// model cuzk/transpose.ts:14-62).  digit_key / win_shift / key_range live in decompose.hpp.
// Device code; included by sequencer.hip only.
#pragma once
#include "../curves.hpp"
#include "decompose.hpp"

namespace msm377 {
namespace {

// ---- per-window counting sort (the reference's transpose, transpose_serial.wgsl:34-76) ----
//
// Two-level (MSD) counting sort; every pass touches each (window, point) element once:
//   k_range_count  block (chunk, window): LDS histogram of the chunk over 256 coarse key ranges
//   k_range_scan   block per window: region bases per range, per-chunk write offsets
//   k_partition_staged  block (chunk, window): appends each element (index|sign, key) to its
//                  range's region at LDS-ranked offsets, a tile at a time -- contiguous runs, no global atomics
//   k_local_sort_lds  block (range, window): counting sort of the region's <= 129 keys in LDS;
//                  writes its row_ptr slice and its val_idx slice, a CONTIGUOUS output owned by
//                  one block, so the 4-byte stores combine in that XCD's L2.
// (The first version scattered straight from the digit columns: every 4-byte store then left
// L2 as a partial write, 8x the payload, 213 us at n = 2^20.)  Order inside a bucket is free:
// group addition commutes (the reference's transpose is stable only because it is serial).


// Calls f(i, biased_digit) for every i in [beg, end) of a digit column, eight digits per
// 16-byte load where the address allows it (keeps 8x more bytes in flight per thread).
template <class F>
__device__ __forceinline__ void for_each_digit(const uint16_t* __restrict__ dg, uint64_t beg, uint64_t end, uint32_t tid, uint32_t nthreads, F f) {
  uint64_t head = beg;
  while (head < end && (((uintptr_t)(dg + head)) & 15)) head++;
  for (uint64_t i = beg + tid; i < head; i += nthreads) f(i, (uint32_t)dg[i]);
  const uint64_t groups = (end - head) / 8;
  const uint4* v = reinterpret_cast<const uint4*>(dg + head);
  for (uint64_t g = tid; g < groups; g += nthreads) {
    const uint4 q = v[g];
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      f(head + g * 8 + 2 * k, w[k] & 0xffffu);
      f(head + g * 8 + 2 * k + 1, w[k] >> 16);
    }
  }
  for (uint64_t i = head + groups * 8 + tid; i < end; i += nthreads) f(i, (uint32_t)dg[i]);
}

__global__ void __launch_bounds__(1024) k_range_count(const uint16_t* __restrict__ digits, uint32_t* __restrict__ counts /* [ws][r][c] */,
                                                      uint64_t n, uint32_t chunks, uint64_t per_chunk, const uint32_t* __restrict__ key_max, uint32_t prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);  // sequencer.hip: front-end kernels outrank the conversion beside them
  __shared__ uint32_t cnt[NRANGE];
  const uint32_t c = blockIdx.x, ws = blockIdx.y, tid = threadIdx.x;
  const uint32_t shift = win_shift(key_max[ws]);
  if (tid < NRANGE) cnt[tid] = 0;
  __syncthreads();
  const uint64_t beg = (uint64_t)c * per_chunk;
  const uint64_t end = (beg + per_chunk < n) ? beg + per_chunk : n;
  for_each_digit(digits + (size_t)ws * n, beg, end, tid, 1024, [&](uint64_t, uint32_t biased) {
    uint32_t key, sign;
    digit_key(biased, key, sign);
    atomicAdd(&cnt[key_range(key, shift)], 1u);
  });
  __syncthreads();
  if (tid < NRANGE) counts[((size_t)ws * NRANGE + tid) * chunks + c] = cnt[tid];
}

// Block per window slot, thread per range: region_base[r] = elements in smaller ranges;
// counts[ws][r][c] becomes the write offset of chunk c inside region r (absolute).
__global__ void __launch_bounds__(NRANGE) k_range_scan(uint32_t* __restrict__ counts, uint32_t* __restrict__ region_base, uint32_t chunks, uint32_t prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);  // sequencer.hip: front-end kernels outrank the conversion beside them
  __shared__ uint32_t part[NRANGE];
  const uint32_t ws = blockIdx.x, r = threadIdx.x;
  uint32_t* cr = counts + ((size_t)ws * NRANGE + r) * chunks;
  uint32_t tot = 0;
  for (uint32_t c = 0; c < chunks; c++) tot += cr[c];
  part[r] = tot;
  __syncthreads();
  for (uint32_t off = 1; off < NRANGE; off <<= 1) {
    const uint32_t v = r >= off ? part[r - off] : 0u;
    __syncthreads();
    part[r] += v;
    __syncthreads();
  }
  const uint32_t base = part[r] - tot;
  region_base[ws * (NRANGE + 1) + r] = base;
  if (r == NRANGE - 1) region_base[ws * (NRANGE + 1) + NRANGE] = part[r];
  uint32_t run = base;
  for (uint32_t c = 0; c < chunks; c++) {
    const uint32_t v = cr[c];
    cr[c] = run;
    run += v;
  }
}

// The partition pass, its scatter staged through LDS (round 3).  Round 2's k_partition appended every element straight to
// one of 256 write streams per workgroup, 8 bytes at a time; a stream advances by 2 KB over the workgroup's whole life,
// so its lines left L2 half-written: WRITE_SIZE 184 MB for 131 MB of sort_temp, 87 us.  Here a workgroup takes its
// chunk in tiles of PT_TILE elements: ranks inside the tile from an LDS histogram (the one atomic per element that is
// left), an exclusive scan over the 256 ranges, the elements parked in LDS in range order, and then the tile is written
// out by consecutive lanes -- a range's ~32 elements of the tile are one contiguous run of its stream, so whole lines go
// out in one piece and only the two ends of a run are partial.  Interleaved A/B at 2^20: sort stage 0.344 -> 0.314 ms.
constexpr uint32_t PT_TILE = 8192;  // 8 digits (one 16-byte load) per thread; 64 KB of staging: two workgroups per CU
__global__ void __launch_bounds__(1024) k_partition_staged(const uint16_t* __restrict__ digits, const uint32_t* __restrict__ counts,
                                                           SortElem* __restrict__ temp, uint64_t n, uint32_t chunks, uint64_t per_chunk,
                                                           const uint32_t* __restrict__ key_max, uint32_t prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);  // sequencer.hip: front-end kernels outrank the conversion beside them
  __shared__ uint32_t cur[NRANGE];     // this workgroup's write cursor in every range's region
  __shared__ uint32_t cnt[NRANGE];     // elements of the tile per range
  __shared__ uint32_t toff[NRANGE];    // first slot of the range in the staged tile
  __shared__ uint32_t gdelta[NRANGE];  // region position of a staged element = its slot + gdelta[range]
  __shared__ uint32_t wsum[NRANGE / 64];
  __shared__ SortElem stage[PT_TILE];
  const uint32_t c = blockIdx.x, ws = blockIdx.y, tid = threadIdx.x;
  const uint32_t shift = win_shift(key_max[ws]);
  if (tid < NRANGE) {
    cur[tid] = counts[((size_t)ws * NRANGE + tid) * chunks + c];
    cnt[tid] = 0;
  }
  __syncthreads();
  const uint64_t beg = (uint64_t)c * per_chunk;
  const uint64_t end = (beg + per_chunk < n) ? beg + per_chunk : n;
  const uint16_t* dg = digits + (size_t)ws * n;
  SortElem* out = temp + (size_t)ws * n;
  for (uint64_t tile0 = beg; tile0 < end; tile0 += PT_TILE) {
    const uint64_t i0 = tile0 + (uint64_t)tid * 8;
    uint32_t d[8];
    uint32_t valid = 0;  // bit j: digit j exists
    if (i0 + 8 <= end && (((uintptr_t)(dg + i0)) & 15) == 0) {
      const uint4 q = *reinterpret_cast<const uint4*>(dg + i0);
      const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        d[2 * k] = w[k] & 0xffffu;
        d[2 * k + 1] = w[k] >> 16;
      }
      valid = 0xffu;
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        d[j] = 0;
        if (i0 + j < end) {
          d[j] = dg[i0 + j];
          valid |= 1u << j;
        }
      }
    }
    uint32_t key[8], sign[8], rg[8], rk[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      digit_key(d[j], key[j], sign[j]);
      rg[j] = key_range(key[j], shift);
      rk[j] = 0;
      if ((valid >> j) & 1u) rk[j] = atomicAdd(&cnt[rg[j]], 1u);
    }
    __syncthreads();
    // exclusive scan of cnt over the 256 ranges: wave scans + the four wave totals
    uint32_t mine = 0, incl = 0;
    if (tid < NRANGE) {
      mine = cnt[tid];
      incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up((int)incl, off, 64);
        if ((tid & 63) >= (uint32_t)off) incl += v;
      }
      if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < NRANGE) {
      uint32_t before = 0;
      for (uint32_t w = 0; w < (tid >> 6); w++) before += wsum[w];
      const uint32_t excl = before + incl - mine;
      toff[tid] = excl;
      gdelta[tid] = cur[tid] - excl;
      cur[tid] += mine;
      cnt[tid] = 0;  // for the next tile (its atomics come after this tile's last barrier)
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++)
      if ((valid >> j) & 1u) stage[toff[rg[j]] + rk[j]] = SortElem{(uint32_t)(i0 + j) | (sign[j] << 31), key[j]};
    __syncthreads();
    const uint32_t tile_len = (uint32_t)((end - tile0 < PT_TILE) ? end - tile0 : PT_TILE);
    for (uint32_t p = tid; p < tile_len; p += 1024) {
      const SortElem e = stage[p];
      out[gdelta[key_range(e.key, shift)] + p] = e;
    }
    __syncthreads();
  }
}

// Block (range r, window slot ws), 256 threads: the region holds exactly the elements with keys in [r KRANGE, (r + 1) KRANGE)
// (plus key NBK for the last range).  It was written by other CUs, so every load misses L2: a region of up to LS_CACHE
// elements is read ONCE into REGISTERS (24 per thread, all their loads in flight at once), histogrammed, scattered into a
// 24 KB LDS array in key order and written to val_idx by consecutive lanes -- whole lines instead of 4-byte stores at
// random places of the slice, and 25 KB of LDS per workgroup (round 2 kept the region in 48 KB of LDS and scattered
// straight to val_idx: three workgroups per CU and no room for the base conversion that runs beside the sort; interleaved
// A/B at 2^20: sort stage 0.344 -> 0.296 ms, whole MSM 2.60 -> 2.55).  Longer regions (skewed scalars) are streamed twice.
constexpr uint32_t LS_CACHE = 6144;           // region length (n / 256 = 4096 on average at n = 2^20) the register / LDS path holds
constexpr uint32_t LS_REG = LS_CACHE / 256;  // elements per thread
// CH: the rows are filed by upload chunk as well (common.hpp RowView / ChunkCuts; the host-buffer entry point): bin
// (key, chunk of the point index) instead of key, K sub-row bounds per key in row_ptr.  CH = false compiles to the
// plain layout.
template <bool CH>
__global__ void __launch_bounds__(256) k_local_sort_lds(const SortElem* __restrict__ temp, const uint32_t* __restrict__ region_base,
                                                        uint32_t* __restrict__ row_ptr, uint32_t* __restrict__ val_idx, uint64_t n,
                                                        const uint32_t* __restrict__ key_max, uint32_t NR, uint32_t NBK, ChunkCuts cuts, uint32_t prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);  // sequencer.hip: front-end kernels outrank the conversion beside them
  // NR ranges per window over keys 0 .. NBK (256 x 2^15 on the main path; 4096 x 2^19, one window, no key_max, for the
  // wide windows of kernels/wide.hpp); the last range also owns key NBK.
  constexpr uint32_t MAXK = CH ? MAX_UPLOAD_CHUNKS : 1;
  constexpr uint32_t NBINS = (KRANGE + 1) * MAXK, PER = (NBINS + 255) / 256;  // bins a thread owns in the scan
  __shared__ uint32_t bins[NBINS];
  __shared__ uint32_t part[256];
  __shared__ uint32_t sorted[LS_CACHE];
  const uint32_t K = CH ? cuts.k : 1u;
  const uint32_t RPW = (NBK + 1) * K + 1;  // row_ptr entries per window
  const uint32_t r = blockIdx.x, ws = blockIdx.y, tid = threadIdx.x;
  const uint32_t shift = key_max ? win_shift(key_max[ws]) : 0u;
  const uint32_t KR = KRANGE >> shift;  // keys per range in this window
  const uint32_t lo = r * KR;
  const bool last = shift == 0 && r == NR - 1;  // only the full-width layout reaches key NBK
  const uint32_t nb = (KR + (last ? 1u : 0u)) * K;  // bins in use: (key - lo, chunk)
  const uint32_t rbeg = region_base[ws * (NR + 1) + r], rend = region_base[ws * (NR + 1) + r + 1];
  const uint32_t len = rend - rbeg;
  const bool cached = len <= LS_CACHE;
  const SortElem* in = temp + (size_t)ws * n + rbeg;
  auto bin_of = [&](const SortElem& e) {
    uint32_t b = (e.key - lo) * K;
    if (CH) {
      const uint32_t idx = e.idx_sign & 0x7fffffffu;
#pragma unroll
      for (uint32_t j = 1; j < MAXK; j++) b += (j < K && idx >= cuts.cut[j]) ? 1u : 0u;
    }
    return b;
  };
  for (uint32_t b = tid; b < NBINS; b += 256) bins[b] = 0;
  __syncthreads();
  SortElem e[LS_REG];
  if (cached) {
#pragma unroll
    for (uint32_t u = 0; u < LS_REG; u++) {
      const uint32_t i = u * 256 + tid;
      e[u].key = lo;
      e[u].idx_sign = 0;
      if (i < len) e[u] = in[i];
    }
#pragma unroll
    for (uint32_t u = 0; u < LS_REG; u++)
      if (u * 256 + tid < len) atomicAdd(&bins[bin_of(e[u])], 1u);
  } else {
    for (uint32_t i0 = 0; i0 < len; i0 += 2048) {
      SortElem f[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t i = i0 + u * 256 + tid;
        f[u].key = lo;
        f[u].idx_sign = 0;
        if (i < len) f[u] = in[i];
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (i0 + u * 256 + tid < len) atomicAdd(&bins[bin_of(f[u])], 1u);
    }
  }
  __syncthreads();
  // exclusive scan over the bins in use: a thread owns PER consecutive bins
  uint32_t own[PER], sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    const uint32_t b = tid * PER + k;
    own[k] = b < nb ? bins[b] : 0u;
    sum += own[k];
  }
  part[tid] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 256; off <<= 1) {
    const uint32_t v = tid >= off ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t* rp = row_ptr + (size_t)ws * RPW + (size_t)lo * K;  // bin b of this range is offset lo K + b of the window
  uint32_t start = rbeg + part[tid] - sum;
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < PER; k++) {
    const uint32_t b = tid * PER + k;
    if (b < nb) {
      bins[b] = start;  // becomes the write cursor of the bin
      rp[b] = start;
      start += own[k];
    }
  }
  if (tid == 0 && last) rp[nb] = rend;  // the end sentinel behind key NBK
  if (shift) {  // narrowed ranges cover keys below NR * KR only: every row above is empty and starts at the end
    const uint32_t covered = NR * KR * K, total = region_base[ws * (NR + 1) + NR];
    const uint32_t per_block = (RPW - covered + NR - 1) / NR;
    uint32_t* rp_w = row_ptr + (size_t)ws * RPW;
    for (uint32_t j = tid; j < per_block; j += 256) {
      const uint32_t idx = covered + r * per_block + j;
      if (idx < RPW) rp_w[idx] = total;
    }
  }
  __syncthreads();
  uint32_t* vi = val_idx + (size_t)ws * n;
  if (cached) {
#pragma unroll
    for (uint32_t u = 0; u < LS_REG; u++)
      if (u * 256 + tid < len) sorted[atomicAdd(&bins[bin_of(e[u])], 1u) - rbeg] = e[u].idx_sign;
    __syncthreads();
    for (uint32_t i = tid; i < len; i += 256) vi[rbeg + i] = sorted[i];
  } else {
    for (uint32_t i0 = 0; i0 < len; i0 += 2048) {
      SortElem f[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t i = i0 + u * 256 + tid;
        f[u].key = lo;
        f[u].idx_sign = 0;
        if (i < len) f[u] = in[i];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t i = i0 + u * 256 + tid;
        if (i < len) vi[atomicAdd(&bins[bin_of(f[u])], 1u)] = f[u].idx_sign;
      }
    }
  }
}

// One workgroup per window: counting sort of the window's n <= SMALL_SORT_MAX digits by key |d| in LDS, straight to the
// CSR form the accumulation reads (row_ptr: 2^L + 2 offsets per window over keys 0 .. 2^L; val_idx: index | sign << 31).
__global__ void __launch_bounds__(1024) k_small_sort(const uint16_t* __restrict__ digits, uint32_t* __restrict__ row_ptr, uint32_t* __restrict__ val_idx,
                                                     uint32_t n, uint32_t L) {
  __shared__ uint32_t bins[SMALL_BINS_MAX + 1];
  __shared__ uint32_t part[1024];
  const uint32_t ws = blockIdx.x, tid = threadIdx.x;
  const uint32_t half = 1u << L, nbins = half + 1;
  const uint16_t* dg = digits + (size_t)ws * n;
  for (uint32_t b = tid; b <= nbins; b += 1024) bins[b] = 0;
  __syncthreads();
  for (uint32_t i = tid; i < n; i += 1024) {
    const int d = (int)dg[i] - (int)half;
    atomicAdd(&bins[d < 0 ? -d : d], 1u);
  }
  __syncthreads();
  // exclusive scan over the bins: each thread owns `per` consecutive bins
  const uint32_t per = (nbins + 1023) / 1024;
  uint32_t local = 0;
  for (uint32_t k = 0; k < per; k++) {
    const uint32_t b = tid * per + k;
    if (b < nbins) local += bins[b];
  }
  part[tid] = local;
  __syncthreads();
  for (uint32_t off = 1; off < 1024; off <<= 1) {
    const uint32_t v = tid >= off ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = part[tid] - local;
  uint32_t* rp = row_ptr + (size_t)ws * (half + 2);
  for (uint32_t k = 0; k < per; k++) {
    const uint32_t b = tid * per + k;
    if (b < nbins) {
      const uint32_t cnt = bins[b];
      bins[b] = run;  // becomes the write cursor of the bin
      rp[b] = run;
      run += cnt;
    }
  }
  if (tid == 0) rp[nbins] = n;
  __syncthreads();
  uint32_t* vi = val_idx + (size_t)ws * n;
  for (uint32_t i = tid; i < n; i += 1024) {
    const int d = (int)dg[i] - (int)half;
    vi[atomicAdd(&bins[d < 0 ? -d : d], 1u)] = i | (d < 0 ? 0x80000000u : 0u);
  }
}

}  // namespace
}  // namespace msm377
