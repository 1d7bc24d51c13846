// Stage 3: bucket accumulation -- work list (k_work_hist / scan / scatter), k_accumulate (the dominant kernel),
// k_accumulate_quad (small inputs), k_merge_split_rows_quad.  Replaces wgsl/cuzk/smvp_bls12_377.template.wgsl:72-160.
// Device code; included by sequencer.hip only.
#pragma once
#include "../curves.hpp"
#include "reduce.hpp"  // quad_bcast, sel4, add_quad, coord4

namespace msm377 {
namespace {

// ---- bucket accumulation (the reference's SMVP, smvp_bls12_377.template.wgsl:72-160) ----
//
// A bucket's CSR row (key t: +P for digit +t, -P for digit -t; the reference's thread walks
// rows t+h and h-t and negates the second sum, :96-133; its bucket 0 = digit -2^15 is bucket
// 32768 here) is cut into work items of at most SEG entries, one thread each.  Row lengths
// are Poisson(n/2^15) in 15 windows but ~7x longer in the top window (13 significant bits),
// and a wave runs as long as its longest lane, so the items are counting-sorted by length,
// longest first, across ALL window slots (k_work_hist / k_work_scan / k_work_scatter): the
// lanes of a wave then finish together and no serial chain exceeds SEG mixed additions.
// Item 0 of a row writes the bucket; items s >= 1 write overflow partials that
// k_merge_split_rows adds back (rows longer than SEG: the top window always, any window under
// skewed scalars -- the load balancing the reference left out, README.md:543-547).


// [0], [1]: begin and end of bucket t's row in window slot ws -- of its sub-row rv.c when the rows are filed by upload
// chunk (common.hpp RowView).
__device__ __forceinline__ const uint32_t* row_bounds(const uint32_t* __restrict__ row_ptr, uint32_t L, uint32_t ws, uint32_t t, RowView rv) {
  return row_ptr + (size_t)ws * (((1u << L) + 1) * rv.k + 1) + (size_t)(t + 1) * rv.k + rv.c;
}
__device__ __forceinline__ uint32_t row_len(const uint32_t* __restrict__ row_ptr, uint32_t L, uint32_t row, RowView rv) {
  const uint32_t* b = row_bounds(row_ptr, L, row >> L, row & ((1u << L) - 1), rv);
  return b[1] - b[0];
}

// A row of `len` entries becomes `nseg` work items of `seglen` entries (the last one `lastlen`): equal parts of at
// most SEG entries, so a row just over a multiple of SEG does not leave one full-length chain beside a stub.
struct RowSplit {
  uint32_t nseg, seglen, lastlen;
};
__device__ __forceinline__ RowSplit row_split(uint32_t len, uint32_t SEG) {
  RowSplit r;
  if (len <= SEG) {
    r.nseg = 1;
    r.seglen = r.lastlen = len;
    return r;
  }
  const uint32_t parts = (len + SEG - 1) / SEG;
  r.seglen = (len + parts - 1) / parts;
  r.nseg = (len + r.seglen - 1) / r.seglen;  // <= parts
  r.lastlen = len - (r.nseg - 1) * r.seglen;  // 1..seglen
  return r;
}

// Thread per row, 1024 rows per block: length histogram of its work items (LDS, then one global
// atomic per bin and block -- the ~60 hot counters serialise, hence the large blocks); rows with more than one item reserve overflow slots and join the split-row list.
__global__ void __launch_bounds__(1024) k_work_hist(const uint32_t* __restrict__ row_ptr, uint32_t L, uint32_t rows, uint32_t SEG, uint32_t* __restrict__ work_hist,
                                                   uint32_t* __restrict__ row_ovf_base, uint32_t* __restrict__ counters /* [0]=split rows, [1]=overflow slots */,
                                                   uint32_t* __restrict__ split_rows, RowView rv, uint32_t prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);  // sequencer.hip: front-end kernels outrank the conversion beside them
  __shared__ uint32_t lh[SEG_BINS];
  __shared__ uint32_t blk[4];  // split rows, overflow slots of this block; then their bases in the global lists
  const uint32_t tid = threadIdx.x, row = blockIdx.x * 1024 + tid;
  if (tid < SEG_BINS) lh[tid] = 0;
  if (tid < 2) blk[tid] = 0;
  __syncthreads();
  RowSplit sp = {1, 0, 0};
  uint32_t my_split = 0, my_ovf = 0;
  if (row < rows) {
    sp = row_split(row_len(row_ptr, L, row, rv), SEG);
    atomicAdd(&lh[sp.lastlen], 1u);
    if (sp.nseg > 1) {
      atomicAdd(&lh[sp.seglen], sp.nseg - 1);
      my_split = atomicAdd(&blk[0], 1u);
      my_ovf = atomicAdd(&blk[1], sp.nseg - 1);
    }
  }
  __syncthreads();
  if (tid < SEG_BINS && lh[tid]) atomicAdd(&work_hist[tid], lh[tid]);
  // One pair of global atomics per block (a row's slots stay contiguous, a block's rows stay together in the
  // split-row list, so the merge pass touches neighbouring buckets).
  if (tid == 0 && blk[0]) {
    blk[2] = atomicAdd(&counters[0], blk[0]);
    blk[3] = atomicAdd(&counters[1], blk[1]);
  }
  __syncthreads();
  if (sp.nseg > 1) {
    row_ovf_base[row] = blk[3] + my_ovf;
    split_rows[blk[2] + my_split] = row;
  }
}

// Thread per row again: claims its slots in the work list, which is sorted by item length, longest first: the items of
// length b start behind all longer ones, at first[b] = sum of work_hist[b' > b] -- every workgroup works that out for
// itself from the finished histogram (129 words; a launch of its own for this scan cost 5 us of every call) -- and
// cursor[b] (zero at the start: the meta block's clear) counts the slots of that length handed out so far.
__global__ void __launch_bounds__(1024) k_work_scatter(const uint32_t* __restrict__ row_ptr, uint32_t L, uint32_t rows, uint32_t SEG,
                                                      const uint32_t* __restrict__ work_hist, uint32_t* __restrict__ cursor, uint32_t* __restrict__ total,
                                                      WorkItem* __restrict__ work, RowView rv, uint32_t prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);  // sequencer.hip: front-end kernels outrank the conversion beside them
  __shared__ uint32_t lh[SEG_BINS];
  __shared__ uint32_t lbase[SEG_BINS];
  __shared__ uint32_t gh[SEG_BINS];
  const uint32_t tid = threadIdx.x, row = blockIdx.x * 1024 + tid;
  if (tid < SEG_BINS) {
    lh[tid] = 0;
    gh[tid] = work_hist[tid];
  }
  __syncthreads();
  uint32_t first = 0;  // of this thread's bin
  if (tid < SEG_BINS) {
    for (uint32_t b = tid + 1; b < SEG_BINS; b++) first += gh[b];
    if (tid == 0 && blockIdx.x == 0) *total = first + gh[0];  // the item count k_accumulate reads
  }
  RowSplit sp = {0, 0, 0};
  uint32_t rank_full = 0, rank_last = 0;
  if (row < rows) {
    sp = row_split(row_len(row_ptr, L, row, rv), SEG);
    if (sp.nseg > 1) rank_full = atomicAdd(&lh[sp.seglen], sp.nseg - 1);
    rank_last = atomicAdd(&lh[sp.lastlen], 1u);
  }
  __syncthreads();
  if (tid < SEG_BINS && lh[tid]) lbase[tid] = first + atomicAdd(&cursor[tid], lh[tid]);
  __syncthreads();
  if (row < rows) {
    for (uint32_t s = 0; s + 1 < sp.nseg; s++) work[lbase[sp.seglen] + rank_full + s] = WorkItem{row, s};
    work[lbase[sp.lastlen] + rank_last] = WorkItem{row, sp.nseg - 1};
  }
}

// One thread per work item.  OCC = waves per SIMD the register allocator must allow.
template <class CV, int OCC, class BP = CV, int PF = 0>  // BP: where the input points come from (CV itself, or TeAffBase); PF: extra records in flight
__global__ void __launch_bounds__(256, OCC) k_accumulate(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ val_idx,
                                                       const uint32_t* __restrict__ bases, uint32_t* __restrict__ buckets, uint64_t n,
                                                       const WorkItem* __restrict__ work, const uint32_t* __restrict__ work_total,
                                                       const uint32_t* __restrict__ row_ovf_base, uint32_t* __restrict__ ovf, uint32_t SEG,
                                                       int* __restrict__ err, const int* __restrict__ conv_err, uint32_t into, uint64_t table_stride, uint32_t L, RowView rv) {
  const uint32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v == 0 && *conv_err) atomicOr(err, *conv_err);  // the table holds a point its coordinate system cannot represent
  if (v >= *work_total) return;
  const WorkItem it = work[v];
  const uint32_t ws = it.row >> L, t = it.row & ((1u << L) - 1);
  const uint32_t* rb = row_bounds(row_ptr, L, ws, t, rv);
  const uint32_t* vi = val_idx + (size_t)ws * n;
  bases += (size_t)ws * table_stride * BP::REC_WORDS;  // precomputed-window tables: window slot ws gathers from its own copy, [2^(16 ws)] P_i
  const uint32_t row_beg = rb[0], row_end = rb[1];
  const uint32_t seglen = row_split(row_end - row_beg, SEG).seglen;
  uint32_t k = row_beg + it.seg * seglen;
  const uint32_t end = (row_end - k > seglen) ? k + seglen : row_end;
  // into: the buckets already hold the sums of an earlier chunk of the same MSM (host-buffer entry point, chunked
  // upload): the row's first item continues from there.
  typename CV::Pt acc = (into && it.seg == 0) ? load_bucket<CV>(buckets, L, ws, t) : CV::identity();
  const bool start_fresh = !(into && it.seg == 0);
  bool bad = false;  // an exceptional pair of the twisted Edwards law (te377.hpp): sticky, the caller falls back
  if (k < end) {
    // Software pipeline: while entry k is added the record of entry k+PF+1 is on its way (its gather is issued at the
    // start of the stage), the records of entries k+1 .. k+PF have been requested one to PF additions ago, and the index
    // of entry k+PF+2 is being read -- so neither the val_idx -> bases address dependency nor the gather latency stalls
    // the wave.  PF = 0 everywhere: one addition, ~10 us, covers a gather even from the 2.2 GB tables of precomputed
    // window multiples, which miss the Infinity Cache -- PF = 1 (one more record in flight, 252 VGPRs) was measured on
    // the 20-bit-window table: kernel 1.388 -> 1.435 ms, 64 x 2^20 batch 2.199 -> 2.274 ms per MSM.
    typename BP::Base rec[PF + 1];  // rec[0]: the entry being added
    uint32_t e[PF + 2];             // e[j]: entry of rec[j]; e[PF + 1]: the next index, already loaded
#pragma unroll
    for (int j = 0; j <= PF + 1; j++) e[j] = (k + j < end) ? vi[k + j] : 0u;
#pragma unroll
    for (int j = 0; j <= PF; j++) rec[j] = BP::load_base(bases, e[j] & 0x7fffffffu);  // (past the end: record 0, never used)
    bool more = true;  // rec[0] / e[0] hold an entry that has not been added yet
    // One stage: start the gathers for the entries further down, add entry rec[0], rotate.  FIRST is a compile-time
    // switch so that the chain's first entry (BP::first: a copy, one product) is PEELED off the loop -- written as
    // `start_fresh ? first(cur) : madd(acc, cur)` inside the loop the compiler evaluated both sides every
    // iteration and selected: 8 products per iteration instead of 7 (9 instead of 8 with projective records; 2697
    // v_mad_u64_u32 in the loop body instead of 2360 -- tools/isa_mix.py, profiles/r02_final/isa_mix.json).
    auto stage = [&](auto first_tag) {
      constexpr bool FIRST = decltype(first_tag)::value;
      k++;
      more = k < end;
      typename BP::Base fresh = rec[PF];
      uint32_t e_new = 0u;
      if (k + PF < end) {
        fresh = BP::load_base(bases, e[PF + 1] & 0x7fffffffu);
        if (k + PF + 1 < end) e_new = vi[k + PF + 1];
      }
      if constexpr (FIRST)
        acc = BP::first(rec[0], (e[0] >> 31) != 0);
      else
        acc = BP::madd(acc, rec[0], (e[0] >> 31) != 0);
      bad |= CV::is_bad(acc);
#pragma unroll
      for (int j = 0; j < PF; j++) rec[j] = rec[j + 1];
      rec[PF] = fresh;
#pragma unroll
      for (int j = 0; j <= PF; j++) e[j] = e[j + 1];
      e[PF + 1] = e_new;
    };
    if (start_fresh) stage(std::true_type{});  // a chain that starts from the identity: its first entry needs no addition
    while (more) stage(std::false_type{});
  }
  if (bad) atomicOr(err, ERR_TE_EXCEPTIONAL);
  if (it.seg == 0) {
    store_bucket<CV>(buckets, L, ws, t, acc);
  } else {
    store_record<CV>(ovf + (size_t)(row_ovf_base[it.row] + it.seg - 1) * CV::BKT_WORDS, acc);
  }
}

// A mixed addition acc + (+-)base on a lane quad: every lane holds the whole accumulator and ONE coordinate of the
// base record -- lane 0 the factor of Y1 - X1, lane 1 that of Y1 + X1 (the two swap for a negated point, which the
// caller does by loading the other one), lane 2 (+-) 2d T2, lane 3 2 Z2 -- and computes one product of each of the two
// rounds of te377.hpp madd: 2 products deep instead of 8.
template <class F, class K>
__device__ __forceinline__ typename TeLazy<F, K>::Ext te_madd_quad(const typename TeLazy<F, K>::Ext& a, const typename F::El& mine, uint32_t q) {
  using El = typename F::El;
  const El m1 = F::mul_lz(sel4(q, F::add_kp_sub(a.y, K::KP2, a.x), F::add_lz(a.y, a.x), a.t, a.z), mine);
  const El pa = quad_bcast<0>(m1), pb = quad_bcast<1>(m1), c = quad_bcast<2>(m1), d = quad_bcast<3>(m1);
  const El e = F::norm(F::add_kp_sub(pb, K::KP2, pa)), f = F::norm(F::add_kp_sub(d, K::KP2, c));
  const El g = F::norm(F::add_lz(d, c)), h = F::add_lz(pb, pa);
  const El m3 = F::mul_lz(sel4(q, e, h, h, f), sel4(q, f, g, e, g));
  typename TeLazy<F, K>::Ext o;
  o.x = quad_bcast<0>(m3);
  o.y = quad_bcast<1>(m3);
  o.t = quad_bcast<2>(m3);
  o.z = quad_bcast<3>(m3);
  return o;
}

// k_accumulate with a lane quad per work item, for inputs so small that the launch is one chain's latency (the
// narrow-window path: a few thousand short chains, a thread-level addition is ~8 us, a quad-level one ~3).  Same work
// list, same buckets and overflow records; projective base records (TeDev) only.
template <class CV>
__global__ void __launch_bounds__(256, 2) k_accumulate_quad(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ val_idx,
                                                          const uint32_t* __restrict__ bases, uint32_t* __restrict__ buckets, uint64_t n,
                                                          const WorkItem* __restrict__ work, const uint32_t* __restrict__ work_total,
                                                          const uint32_t* __restrict__ row_ovf_base, uint32_t* __restrict__ ovf, uint32_t SEG,
                                                          int* __restrict__ err, const int* __restrict__ conv_err, uint32_t into, uint32_t L) {
  using El = typename CV::F::El;
  using K = typename CV::Pt_K;
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  const uint32_t v = gid >> 2, q = threadIdx.x & 3;
  if (gid == 0 && *conv_err) atomicOr(err, *conv_err);
  if (v >= *work_total) return;  // whole quads leave together
  const WorkItem it = work[v];
  const uint32_t ws = it.row >> L, t = it.row & ((1u << L) - 1);
  const uint32_t* rb = row_bounds(row_ptr, L, ws, t, RowView{});
  const uint32_t* vi = val_idx + (size_t)ws * n;
  const uint32_t row_beg = rb[0], row_end = rb[1];
  const uint32_t seglen = row_split(row_end - row_beg, SEG).seglen;
  uint32_t k = row_beg + it.seg * seglen;
  const uint32_t end = (row_end - k > seglen) ? k + seglen : row_end;
  const bool continues = into && it.seg == 0;  // see k_accumulate
  typename CV::Pt acc = continues ? load_bucket<CV>(buckets, L, ws, t) : CV::identity();
  bool bad = false;
  // the coordinate of entry e's record this lane multiplies by (lane 2: negated below for a negative digit)
  auto load_mine = [&](uint32_t e) {
    const bool neg = (e >> 31) != 0;
    const uint32_t comp = q < 2 ? (q ^ (neg ? 1u : 0u)) : q;
    const uint32_t* src = bases + (size_t)(e & 0x7fffffffu) * CV::REC_WORDS + comp * CV::NL;
    El r;
#pragma unroll
    for (int j = 0; j < (int)CV::NL; j++) r.l[j] = src[j];
    return r;
  };
  if (k < end && !continues) {  // a chain that starts from the identity: its first entry is a copy (one product), on every lane
    const uint32_t e = vi[k];
    acc = CV::first(CV::load_base(bases, e & 0x7fffffffu), (e >> 31) != 0);
    bad |= CV::is_bad(acc);
    k++;
  }
  if (k < end) {
    // entry k's coordinate and the index of entry k + 1 are in flight while entry k - 1 is added (as in k_accumulate)
    uint32_t e_cur = vi[k];
    uint32_t e_nxt = (k + 1 < end) ? vi[k + 1] : 0u;
    El cur = load_mine(e_cur);
    for (;;) {
      k++;
      const bool more = k < end;
      El nxt = cur;
      uint32_t e_nn = 0u;
      if (more) {
        nxt = load_mine(e_nxt);
        if (k + 1 < end) e_nn = vi[k + 1];
      }
      const El mine = (q == 2 && (e_cur >> 31)) ? CV::F::kp_sub(K::KP2, cur) : cur;
      acc = te_madd_quad<typename CV::F, K>(acc, mine, q);
      bad |= CV::is_bad(acc);
      if (!more) break;
      cur = nxt;
      e_cur = e_nxt;
      e_nxt = e_nn;
    }
  }
  if (bad) atomicOr(err, ERR_TE_EXCEPTIONAL);
  uint32_t* dst = it.seg == 0 ? bucket_ptr<CV>(buckets, L, ws, t) : ovf + (size_t)(row_ovf_base[it.row] + it.seg - 1) * CV::BKT_WORDS;
  store_coord<CV>(dst + q * CV::COORD_WORDS, coord4(q, acc).l);  // each lane stores one coordinate
}

// Quad per split row: bucket += its overflow partials.
template <class CV>
__global__ void __launch_bounds__(256, 2) k_merge_split_rows_quad(const uint32_t* __restrict__ row_ptr, uint32_t* __restrict__ buckets,
                                                                  const uint32_t* __restrict__ counters, const uint32_t* __restrict__ split_rows,
                                                                  const uint32_t* __restrict__ row_ovf_base, const uint32_t* __restrict__ ovf, uint32_t SEG,
                                                                  int* __restrict__ err, uint32_t L, RowView rv, uint32_t* __restrict__ host_flag, uint32_t seq) {
  // The first launch behind the accumulation kernel tells the host that it is through (sequencer.hip TailArm: the
  // tail's helper threads start polling for their jobs now).  An event record between the two kernels did the same and
  // cost a ~6 us bubble in the stream.
  if (host_flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const uint32_t count = counters[0];
  const uint32_t q = threadIdx.x & 3;
  for (uint32_t i = (blockIdx.x * 256 + threadIdx.x) >> 2; i < count; i += gridDim.x * 64) {
    const uint32_t row = split_rows[i];
    const uint32_t len = row_len(row_ptr, L, row, rv);
    const uint32_t nseg = row_split(len, SEG).nseg;
    const uint32_t ws = row >> L, t = row & ((1u << L) - 1);
    typename CV::Pt acc = load_bucket_quad<CV>(buckets, L, ws, t, q);  // (all four lanes of a quad share i: they are all here)
    const uint32_t* src = ovf + (size_t)row_ovf_base[row] * CV::BKT_WORDS;
    typename CV::Pt nxt = load_record_quad<CV>(src, q);
    bool bad = false;
    for (uint32_t s = 1; s < nseg; s++) {
      const typename CV::Pt cur = nxt;
      if (s + 1 < nseg) nxt = load_record_quad<CV>(src + (size_t)s * CV::BKT_WORDS, q);
      acc = add_quad(acc, cur, q);
      bad |= CV::is_bad(acc);
    }
    if (bad) atomicOr(err, ERR_TE_MERGE);
    const typename CV::F::El c = coord4(q, acc);
    store_coord<CV>(bucket_ptr<CV>(buckets, L, ws, t) + q * CV::COORD_WORDS, c.l);
  }
}

}  // namespace
}  // namespace msm377
