// msm377: BLS12-377 G1 multi-scalar multiplication for MI355X (gfx950), C ABI in include/msm377.h.
// This translation unit is the stage sequencer: it owns every kernel launch (kernels/*.hpp are compiled here and only
// here), the stream / event choreography of a call and the entry points' control flow (fallbacks, chunked uploads,
// batches, window shards).  The C ABI itself is capi.hip, the host tail host_tail.hip, the context context.hpp.
//
// Pipeline (each stage names the reference code it replaces; paths relative to /root/reference/src/submission/):
//   kernels/convert.hpp     k_affine_up / host inversion / k_affine_down (n >= 2^20, resident tables), k_convert_bases (otherwise)
//                           wire x||y -> Montgomery records            wgsl/cuzk/convert_point_coords_and_decompose_scalars.template.wgsl:41-99 + barrett.template.wgsl:60-82
//   kernels/decompose.hpp   k_decompose (16 windows), k_decompose_geom / k_decompose_narrow (inputs <= 2^16 points: 22 / 23 windows of 2^11 buckets, submission.ts:97)
//                           scalars -> signed digits                   same file :100-141; model cuzk/utils.ts:66-109
//   kernels/sort.hpp        k_range_count / k_range_scan / k_partition / k_local_sort (k_small_sort on the narrow path)
//                           per-window counting sort -> CSR            wgsl/cuzk/transpose_serial.wgsl:34-76 (16 serial threads there); model cuzk/transpose.ts:14-62
//   kernels/accumulate.hpp  k_accumulate: bucket sums (the dominant kernel)   wgsl/cuzk/smvp_bls12_377.template.wgsl:72-160
//   kernels/reduce.hpp      k_tree_step / k_tree_step_quad / k_reduce_tail / k_gather_partials
//                           bucket reduction, log-depth bit planes     wgsl/cuzk/bpr.template.wgsl:69-173; models cuzk/bpr.ts:5-126
//   host_tail.hip           Horner over windows + one inversion        submission.ts:290-321
// The kernels are templates over a curve policy (curves.hpp): TeDev (default: G1 in twisted Edwards form, te377.hpp -- 7 field
// products per bucket addition on affine base records (TeAffBase), 8 on projective ones, unified law, exceptional cases
// detected and rerun), G1Dev (G1 in Weierstrass XYZZ coordinates, g1_xyzz.hpp: the fallback, the GLV front end, the
// stage read-backs) and EdDev (Edwards-BLS12 over the scalar field, ed_ext.hpp).  Everything behind the sort takes the
// bucket geometry as a run-time argument L (2^L buckets per window: 15 on the main path, 11 on the narrow one).
// HBM layout: DESIGN.md section 3.
#include "sequencer.hpp"

#include <hip/hip_runtime.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <thread>
#include <type_traits>
#include <vector>

#include "context.hpp"
#include "host_tail.hpp"
#include "kernels/accumulate.hpp"
#include "kernels/convert.hpp"
#include "kernels/decompose.hpp"
#include "kernels/generate.hpp"
#include "kernels/reduce.hpp"
#include "kernels/sort.hpp"
#include "kernels/wide.hpp"

namespace msm377 {
namespace eng {

bool hip_ok(msm377_ctx* ctx, int hip_error, const char* what) {
  const hipError_t e = (hipError_t)hip_error;
  if (e == hipSuccess) return true;
  if (ctx) ctx->err = std::string(what) + ": " + hipGetErrorString(e);
  return false;
}

namespace {

void note_fallback(msm377_ctx* ctx, uint32_t mask) {
  ctx->fallback_count++;
  ctx->fallback_mask = mask;
}
#define HIP_TRY(ctx, call)                            \
  do {                                                \
    if (!msm377::eng::hip_ok((ctx), (int)(call), #call)) return MSM377_EHIP; \
  } while (0)

// Pageable host memory -> device through a pinned staging buffer: four workers copy ~4 MB pieces
// into it and queue the DMA of each piece on their own stream, so the CPU copy of one piece
// overlaps the DMA of the others.  Measured on the MI355X box for 160 MB: 4.2 ms, against 28 ms
// for a first hipMemcpy from fresh pageable pages (4.4 ms once the runtime has pinned them) and
// 3.3 + 2.9 ms for hipHostRegister + copy.  Returns when the data is on the device.
int h2d_staged(msm377_ctx* ctx, void* d_dst, const uint8_t* src, size_t bytes, size_t stage_off) {
  using msm377::eng::reserve_host_staging;
  constexpr int NT_MAX = 8;
  constexpr int NT = 4;  // copy workers: 2, 4, 6 or 8 all moved 128 MB in 2.9-3.0 ms (round 2) -- the DMA sets the pace
  constexpr size_t SMALL = 8u << 20, PIECE = 4u << 20;
  if (bytes < SMALL) {  // not worth four threads
    HIP_TRY(ctx, hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return MSM377_OK;
  }
  if (!ctx->h_stage && reserve_host_staging(ctx) != MSM377_OK) {
    (void)hipGetLastError();
    HIP_TRY(ctx, hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));  // fall back to the runtime's pageable path
    return MSM377_OK;
  }
  // Pieces of about 4 MB, their number a multiple of the worker count (with fixed 8 MB pieces a 48 MB upload took as
  // long as a 64 MB one); a thread's host copy of its next piece overlaps the DMA of the one before.  The pieces are
  // CLAIMED from a counter, by the NT workers and by the calling thread alike: every thread ends up with the same
  // amount when all run, and a worker that has lost its CPU leaves its pieces to the others instead of holding up the call.
  size_t npieces = (bytes + PIECE - 1) / PIECE;
  npieces = (npieces + NT - 1) / NT * NT;
  const size_t piece = ((bytes + npieces - 1) / npieces + 4095) & ~(size_t)4095;
  uint8_t* stage = ctx->h_stage + stage_off;
  hipError_t errs[NT_MAX + 1];
  std::thread workers[NT_MAX];
  const int device = ctx->device;
  std::atomic<size_t> next{0};
  auto copy_pieces = [&, device](int t) {  // t: this thread's copy stream
    hipError_t e = hipSetDevice(device);
    for (;;) {
      const size_t c = next.fetch_add(1, std::memory_order_relaxed);
      const size_t off = c * piece;
      if (c >= npieces || off >= bytes || e != hipSuccess) break;
      const size_t len = (bytes - off < piece) ? bytes - off : piece;
      memcpy(stage + off, src + off, len);
      e = hipMemcpyAsync((uint8_t*)d_dst + off, stage + off, len, hipMemcpyHostToDevice, ctx->copy_stream[t]);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_stream[t]);
    errs[t] = e;
  };
  for (int t = 0; t <= NT; t++) errs[t] = hipSuccess;
  for (int t = 0; t < NT; t++) workers[t] = std::thread([&copy_pieces, t] { copy_pieces(t); });
  copy_pieces(NT);  // the caller takes pieces too (its own stream)
  for (int t = 0; t < NT; t++) workers[t].join();
  for (int t = 0; t <= NT; t++) HIP_TRY(ctx, errs[t]);
  return MSM377_OK;
}

void identity_wire(uint8_t out[96]) {
  memset(out, 0, 96);
  out[48] = 1;
}

struct StageTimer {  // HIP events around one stage of one part, on the part's own stream
  msm377_ctx* c;
  int s;
  hipStream_t st;
  uint32_t part;
  bool on() const { return c->timing == 1 || (c->timing == 2 && s == MSM377_STAGE_ACC_KERNEL); }
  StageTimer(msm377_ctx* ctx, int stage, hipStream_t stream, uint32_t part_) : c(ctx), s(stage), st(stream), part(part_) {
    if (on()) (void)hipEventRecord(c->ev[part][s][0], st);
  }
  ~StageTimer() {
    if (on()) (void)hipEventRecord(c->ev[part][s][1], st);
  }
};

template <class CV>
int convert_bases(msm377_ctx* ctx, const uint32_t* d_raw, uint64_t n, uint64_t first = 0, bool clear_err = true) {
  // Runs on the side stream: it depends on the points only, while decomposition and the sort
  // depend on the scalars only, so the two overlap (HBM-bound vs LDS/latency-bound);
  // k_accumulate waits for `bases_ready`.  The previous call's readers of d_bases are done: every entry point ends with
  // a host-side wait -- for the main stream's completion event, or (zero-copy output) for the sequence number that the
  // gather kernel publishes, and that kernel is the LAST launch of a call and reads the buckets only (the invariant
  // is spelled out at publish_to_host, kernels/reduce.hpp).
  if (n == 0) return MSM377_OK;
  if (ctx->timing == 1) (void)hipEventRecord(ctx->ev[0][MSM377_STAGE_CONVERT][0], ctx->stream2);
  if (clear_err) hipLaunchKernelGGL(k_clear_words, dim3(1), dim3(256), 0, ctx->stream2, (uint32_t*)(ctx->d_err + 2), 1u, (uint32_t*)nullptr, 0u);
  hipLaunchKernelGGL(k_convert_bases<CV>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream2, d_raw, ctx->d_bases + first * CV::REC_WORDS, n,
                     ctx->d_err + 2);
  HIP_TRY(ctx, hipGetLastError());
  if (ctx->timing == 1) (void)hipEventRecord(ctx->ev[0][MSM377_STAGE_CONVERT][1], ctx->stream2);
  HIP_TRY(ctx, hipEventRecord(ctx->bases_ready, ctx->stream2));
  return MSM377_OK;
}

// Phase 1, queued on the side stream: products up to one value per workgroup, delivered into pinned host memory.
// (Round 3 tried the conversion as TWO launches per direction, blocks [0, h) and [h, nblk), so that the host inverts the
// first half's products while the second half is still on its way up and the ~50 us round trip idles nothing: slower,
// 2.60 -> 2.72 ms at 2^20.  The conversion is latency-bound at two workgroups per CU -- a half-size launch is one
// workgroup per CU and takes 104 us where the full one takes 171 -- so the chain grew from 171 + 50 + 170 to 4 x ~110 us.)
int affine_convert_begin(msm377_ctx* ctx, const uint32_t* d_raw, uint64_t n, const uint32_t* prev_window_records = nullptr, bool clear_err = true) {
  if (n == 0) return MSM377_OK;
  const uint32_t nblk = affine_blocks(n);
  if (ctx->timing == 1) (void)hipEventRecord(ctx->ev[0][MSM377_STAGE_CONVERT][0], ctx->stream2);
  __atomic_store_n(ctx->h_aff_flag, 0u, __ATOMIC_RELEASE);
  if (clear_err) hipLaunchKernelGGL(k_clear_words, dim3(1), dim3(256), 0, ctx->stream2, (uint32_t*)(ctx->d_err + 2), 1u, (uint32_t*)nullptr, 0u);
  if (prev_window_records)
    hipLaunchKernelGGL(k_affine_up<AffDoublingSource>, dim3(nblk), dim3(AFF_THREADS), 0, ctx->stream2, AffDoublingSource{prev_window_records, ctx->table_doublings}, n,
                       ctx->d_aff_stash, ctx->d_aff_trees, ctx->dm_aff_prod, ctx->dm_aff_flag, ctx->d_aff_count, ctx->d_err + 2, ctx->conv_wave_prio);
  else
    hipLaunchKernelGGL(k_affine_up<AffWireSource>, dim3(nblk), dim3(AFF_THREADS), 0, ctx->stream2, AffWireSource{d_raw}, n, ctx->d_aff_stash, ctx->d_aff_trees,
                       ctx->dm_aff_prod, ctx->dm_aff_flag, ctx->d_aff_count, ctx->d_err + 2, ctx->conv_wave_prio);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ctx->aff_up_done, ctx->stream2));
  if (ctx->tail_threads > 1 && nblk >= 32) ctx->tail_pool.prewake(ctx->aff_prewake_us, std::min(ctx->tail_threads, TailPool::WORKERS + 1) - 1);
  return MSM377_OK;
}

// Phase 2: waits for phase 1 (the main stream keeps the GPU busy meanwhile), inverts the block products on the tail
// threads, queues the way down and signals `bases_ready`.
int affine_convert_finish(msm377_ctx* ctx, uint32_t* d_records_out, uint64_t n, bool behind_sort = false) {
  if (n == 0) return MSM377_OK;
  const uint32_t nblk = affine_blocks(n);
  // Poll the flag in pinned memory (no runtime calls: they would contend with nothing, but they are not free either);
  // after 20 ms fall back to the event, which also surfaces a failed kernel.
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t spins = 0; __atomic_load_n(ctx->h_aff_flag, __ATOMIC_ACQUIRE) != nblk; spins++) {
    __builtin_ia32_pause();
    if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
      HIP_TRY(ctx, hipEventSynchronize(ctx->aff_up_done));
      break;
    }
  }
  if (__atomic_load_n(ctx->h_aff_flag, __ATOMIC_ACQUIRE) != nblk) {
    ctx->err = "batched affine conversion: the block products did not arrive";
    return MSM377_EHIP;
  }
  {
    const int inv_rc = invert_block_products_mt(ctx, 0, nblk);
    ctx->tail_pool.disarm();  // (armed by affine_convert_begin; the tail arms them again once the accumulation is through)
    if (inv_rc) return inv_rc;
  }
  // The way down starts as soon as the host has inverted the block products, beside whatever the sort is doing (letting
  // it wait for the sort was measured both ways in round 2: 2^20 2.62 -> 2.59 ms, 2^22 10.39 -> 10.21 without the wait).
  hipLaunchKernelGGL(k_affine_down, dim3(nblk), dim3(AFF_THREADS), 0, ctx->stream2, n, ctx->d_aff_stash, ctx->d_aff_trees, ctx->dm_aff_inv, d_records_out, ctx->conv_wave_prio);
  HIP_TRY(ctx, hipGetLastError());
  if (ctx->timing == 1) (void)hipEventRecord(ctx->ev[0][MSM377_STAGE_CONVERT][1], ctx->stream2);
  HIP_TRY(ctx, hipEventRecord(ctx->bases_ready, ctx->stream2));
  return MSM377_OK;
}

// Entries per accumulation work item.  The kernel is a list of ~(rows + entries / SEG) independent serial chains
// handed out longest-first to 2048 resident waves: too few, too long chains leave the last round of waves
// half-empty (GLV at 2^20 with SEG 96: 4400 waves, 2.83 ms; SEG 64: 6100 waves, 2.47 ms), too short ones pay an
// overflow record and a merge addition per extra chain.  Interleaved A/B runs (tools/ab_knobs.py) put the best
// length near entries / 2^18 for both front ends (entries = windows in this call x points per window): 32 at
// n = 2^19, 64 at 2^20, 96-128 at 2^21; a rank that owns one or two windows of a sharded MSM gets short chains,
// so that its few rows still fill the GPU.
uint32_t auto_seg(const msm377_ctx* ctx, uint64_t entries, bool glv) {
  const uint32_t forced = glv ? ctx->seg_glv : ctx->seg_plain;
  if (forced) return forced;
  const uint64_t s = ((entries >> 18) + 7) & ~7ull;
  return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(s, SEG_MIN), SEG_MAX);
}



// Enqueue stages decompose .. gather for windows [wb, wb + wc) against ctx->d_bases, the D2H of
// the partial records into slot `slot` of ctx->h_partials and that slot's completion event.
// Nothing here waits for the GPU.
// One part of a call's windows: window slots [ws0, ws0 + wc) of every window-indexed buffer, windows
// [wb, wb + wc) of the scalars, on its own stream.
// Which stages of a call to enqueue (all of them, except for the chunked host-buffer entry point).
struct Phase {
  bool clear_err = true;   // first chunk of a call
  bool front = true;       // decompose .. merge
  bool into = false;       // accumulate on top of the buckets of an earlier chunk
  bool back = true;        // bucket reduction, gather, D2H, completion event
  bool zc_out = false;     // the gather kernel writes the records and the error word into pinned host memory itself (k_gather_partials)
  uint64_t base_first = 0; // first record of ctx->d_bases this chunk's indices refer to
  // Precomputed-window tables (msm377_g1_set_bases_precomputed): window slot ws gathers from record ws * table_stride + i
  // of `table`, and because the table already carries the 2^(16 ws) weights the 16 bucket sets are ADDED together
  // before the reduction: one window's reduction, one window's partial record, a 16-step host tail.
  const uint32_t* table = nullptr;
  uint64_t table_stride = 0;
  // Window width of the call: 16 (the main path: 16 windows x 2^15 buckets, the two-level sort) or NARROW_BITS (small
  // inputs: k_decompose_narrow + k_small_sort, 22 windows x 2^11 buckets); everything behind the sort takes
  // L = cbits - 1 as a run-time argument.
  uint32_t cbits = MSM377_WINDOW_BITS;
  uint32_t bucket_log = MSM377_WINDOW_BITS - 1;  // L: 2^L buckets per window (NARROW_LOG on the small-input path)
  // Wide windows over a precomputed table (kernels/wide.hpp): `table` holds [2^(20 w)] P_i for 13 windows, the call has
  // ONE window slot of 2^19 buckets fed by the flat list of 13 n digits (cbits = WIDE_BITS, bucket_log = WIDE_LOG).
  bool wide = false;
  bool even = false;  // whole MSMs on 16 windows: the top three windows 15 bits wide (kernels/decompose.hpp k_decompose); the tail gets short_from = EVEN_FROM
  const uint32_t* bases_override = nullptr;  // base records of the call if not ctx->d_bases (the wide table's window 0 = the plain affine records)
  // Points that arrive in chunks (run_sorted_upload): ONE decomposition and sort of all scalars files every row's entries
  // by chunk (`cuts`), then each chunk's accumulation phase walks its own sub-rows (`chunk`) once its points are on the
  // device.  sort / accumulate select which half of the front phase a call enqueues.
  bool sort = true;        // decompose + sort
  bool accumulate = true;  // work list, accumulation, merge of split rows
  ChunkCuts cuts;          // cuts.k > 1: rows filed by chunk
  uint32_t chunk = 0;      // the chunk this phase accumulates
  uint64_t chunk_points = 0;  // its points (work-item length), 0 = all n
};

struct PartView {
  hipStream_t st;
  uint32_t part, ws0, wb, wc;
  size_t work_off, ovf_off;  // first work item / overflow slot of this part
};

template <class CV, class BP>
int enqueue_part(msm377_ctx* ctx, const uint32_t* d_scalars, uint64_t n_scalars, uint64_t n, const PartView& pv, int* d_err, uint32_t* d_partials, bool glv,
                 uint32_t sort_blocks, const Phase& ph) {
  hipStream_t st = pv.st;
  const uint32_t wc = pv.wc, part = pv.part;
  const uint32_t L = ph.bucket_log, NB = 1u << L;  // this call's bucket geometry (shadows the main path's constant)
  const RowView rv{ph.cuts.k, ph.chunk};
  const uint32_t wprio = ctx->front_wave_prio;  // s_setprio in the decompose / sort / work-list kernels
  const uint32_t RP = (NB + 1) * rv.k + 1;  // row_ptr entries per window slot (NB + 2 for plain rows)
  const bool wide = ph.wide;
  const bool narrow = !wide && ph.cbits != MSM377_WINDOW_BITS;
  const uint64_t entries = wide ? (uint64_t)WIDE_WINDOWS * n : (uint64_t)wc * (ph.chunk_points ? ph.chunk_points : n);  // (window, point) pairs this phase accumulates
  static_assert((uint64_t)NARROW_WINDOWS * SMALL_SORT_MAX / NARROW_SEG + NARROW_WINDOWS * (1u << NARROW_LOG) <= (uint64_t)MSM377_NUM_WINDOWS * 32768, "narrow work items fit the work-item buffer");
  // per launch: each part must fill the GPU on its own.  Narrow windows: a small input is all latency -- a work item is
  // a serial chain of ~10 us additions -- so its chains are cut at 8 entries (the buffers, sized for 16 windows of
  // 2^15 rows plus entries / SEG_MIN items, hold the 23 x 2^11 rows and 23 n / 8 items of an input this small easily).
  // (With the even geometry and quad-cooperative record loads in the merge, 16 entries win from 2^14 points on and 12
  // at 2^13 -- accumulate + merge 0.288 / 0.277 / 0.284 / 0.270 ms for 8 / 10 / 12 / 16 at 2^16, 0.077 / 0.077 / 0.074 /
  // 0.085 at 2^13, profiles/r03_final/sweep_narrow_seg_even.txt; below that a row has two entries on average.)
  const uint32_t SEG = (narrow && !ctx->seg_plain) ? (ctx->narrow_seg ? ctx->narrow_seg : (n > 8192 ? 16u : 12u)) : auto_seg(ctx, entries, glv);
  uint16_t* digits = ctx->d_digits + (size_t)pv.ws0 * n;
  uint32_t* range_counts = ctx->d_range_counts + (size_t)part * NRANGE * (MAX_SORT_BLOCKS / 2);
  uint32_t* region_base = ctx->d_region_base + (size_t)pv.ws0 * (NRANGE + 1);
  SortElem* sort_temp = ctx->d_sort_temp + (size_t)pv.ws0 * n;
  uint32_t* row_ptr = (rv.k > 1 ? ctx->d_row_ptr_chunks : ctx->d_row_ptr) + (size_t)pv.ws0 * RP;
  uint32_t* val_idx = ctx->d_val_idx + (size_t)pv.ws0 * n;
  uint32_t* buckets = ctx->d_buckets + (size_t)pv.ws0 * CV::BKT_WORDS * NB;
  uint32_t* row_ovf_base = ctx->d_row_ovf_base + (size_t)pv.ws0 * NB;
  uint32_t* split_rows = ctx->d_split_rows + (size_t)pv.ws0 * NB;
  WorkItem* work = ctx->d_work + pv.work_off;
  uint32_t* ovf = ctx->d_ovf + pv.ovf_off * CV::BKT_WORDS;
  const uint32_t* bases = ph.table ? ph.table : ph.bases_override ? ph.bases_override : ctx->d_bases + ph.base_first * BP::REC_WORDS;
  uint32_t* meta_block = ctx->d_work_meta + (size_t)part * META_BLOCK_WORDS;  // [work-list counters | key_max[16]]
  uint32_t* key_max = meta_block + (2 * SEG_BINS + 4);
  if (ph.front) {
  // One memset clears this part's work-list counters AND its key_max words (0 = full-width ranges); k_decompose
  // then measures window 15 of the plain front end.
  hipLaunchKernelGGL(k_clear_words, dim3(1), dim3(256), 0, st, meta_block, META_BLOCK_WORDS, (uint32_t*)d_err, (ph.clear_err && part == 0) ? 1u : 0u);
  uint32_t* top_key_max = (!glv && !ph.even && pv.wb + wc == MSM377_NUM_WINDOWS) ? key_max + (wc - 1) : nullptr;
  const uint64_t max_items = (uint64_t)wc * NB + entries / SEG;  // every row has an item; extra ones are full segments
  // a lane quad per work item while the launch is one chain's latency (up to 2^14 points: ~94 k items); beyond that
  // the quads are VALU-bound like threads and only add their exchange instructions (kernel at 2^16: 0.216 / 0.183 ms)
  const bool quad_acc = std::is_same<BP, CV>::value && std::is_same<CV, TeDev>::value && narrow && ctx->narrow_quad_acc && !ph.table && max_items <= ctx->narrow_quad_items;
  // (One launch for the whole front end of such a call -- each window's workgroup recoding, sorting and listing its
  // work items itself, one global atomic per list and workgroup -- was built and dropped: 0.271 -> 0.293 ms at 2^12,
  // 0.342 -> 0.373 at 2^14.  Saving four dispatch latencies did not pay for a work list that is sorted by length
  // only within each window: the accumulation kernel went from 0.038 to 0.054 ms at 2^12.)
  if (ph.sort) {
  {
    StageTimer t(ctx, MSM377_STAGE_DECOMPOSE, st, part);
    if (wide)
      hipLaunchKernelGGL(k_decompose_wide, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, ctx->d_wide_digits, n, d_err);
    else if (narrow && ph.even)
      hipLaunchKernelGGL(k_decompose_geom, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, digits, n, L, NARROW_EVEN_SIGNED, wc, 1u << L, d_err);
    else if (narrow)
      hipLaunchKernelGGL(k_decompose_narrow, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, digits, n, ph.cbits, L, wc, d_err);
    else if (glv)
      hipLaunchKernelGGL(k_decompose_glv, dim3((unsigned)((n_scalars + 255) / 256)), dim3(256), 0, st, d_scalars, digits, n_scalars, pv.wb, wc, d_err);
    else
      hipLaunchKernelGGL(k_decompose, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, digits, n, pv.wb, wc, d_err, top_key_max, wprio, ph.even ? 1u : 0u);
    HIP_TRY(ctx, hipGetLastError());
  }

  if (wide) {  // the 13 n entries into 4096 fine ranges: two staged partition passes per window (kernels/wide.hpp), then k_local_sort_lds
    StageTimer t(ctx, MSM377_STAGE_SORT, st, part);
    uint32_t* wc = ctx->d_wide_counts;
    hipLaunchKernelGGL(k_wide_count, dim3(WS_CHUNKS, WIDE_WINDOWS), dim3(1024), 0, st, (const uint32_t*)ctx->d_wide_digits, wc, n);
    hipLaunchKernelGGL(k_wide_sums, dim3(WIDE_NRANGE / 256, WIDE_WINDOWS), dim3(256), 0, st, wc);
    hipLaunchKernelGGL(k_wide_scan, dim3(1), dim3(1024), 0, st, wc, region_base);
    hipLaunchKernelGGL(k_wide_offsets1, dim3(WIDE_WINDOWS), dim3(WS_COARSE), 0, st, wc, n);
    hipLaunchKernelGGL(k_wide_part1, dim3(WS_CHUNKS, WIDE_WINDOWS), dim3(1024), 0, st, (const uint32_t*)ctx->d_wide_digits, (const uint32_t*)wc, sort_temp, n, (uint32_t)ph.table_stride);
    hipLaunchKernelGGL(k_wide_part2, dim3(WS_COARSE, WIDE_WINDOWS), dim3(1024), 0, st, (const SortElem*)sort_temp, (const uint32_t*)wc, ctx->d_wide_temp);
    hipLaunchKernelGGL(k_local_sort_lds<false>, dim3(WIDE_NRANGE, 1), dim3(256), 0, st, (const SortElem*)ctx->d_wide_temp, region_base, row_ptr, val_idx, entries,
                       (const uint32_t*)nullptr, WIDE_NRANGE, NB, ChunkCuts{}, wprio);
    HIP_TRY(ctx, hipGetLastError());
  } else if (narrow) {
    StageTimer t(ctx, MSM377_STAGE_SORT, st, part);
    hipLaunchKernelGGL(k_small_sort, dim3(wc), dim3(1024), 0, st, digits, row_ptr, val_idx, (uint32_t)n, L);
    HIP_TRY(ctx, hipGetLastError());
  } else {
    StageTimer t(ctx, MSM377_STAGE_SORT, st, part);
    uint32_t chunks = sort_blocks / wc;
    const uint64_t want = (n + 4095) / 4096;  // at least ~4096 elements per block
    if (chunks > want) chunks = (uint32_t)(want ? want : 1);
    const uint64_t per_chunk = (n + chunks - 1) / chunks;
    hipLaunchKernelGGL(k_range_count, dim3(chunks, wc), dim3(1024), 0, st, digits, range_counts, n, chunks, per_chunk, key_max, wprio);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_range_scan, dim3(wc), dim3(NRANGE), 0, st, range_counts, region_base, chunks, wprio);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_partition_staged, dim3(chunks, wc), dim3(1024), 0, st, digits, range_counts, sort_temp, n, chunks, per_chunk, key_max, wprio);
    HIP_TRY(ctx, hipGetLastError());
    if (rv.k > 1)  // rows filed by upload chunk: K sub-row bounds per key
      hipLaunchKernelGGL(k_local_sort_lds<true>, dim3(NRANGE, wc), dim3(256), 0, st, sort_temp, region_base, row_ptr, val_idx, n, key_max, NRANGE, NB, ph.cuts, wprio);
    else
      hipLaunchKernelGGL(k_local_sort_lds<false>, dim3(NRANGE, wc), dim3(256), 0, st, sort_temp, region_base, row_ptr, val_idx, n, key_max, NRANGE, NB, ChunkCuts{}, wprio);
    HIP_TRY(ctx, hipGetLastError());
  }
  }  // ph.sort
  if (ph.accumulate) {
    StageTimer t(ctx, MSM377_STAGE_ACCUMULATE, st, part);
    const uint32_t rows = wc * NB;
    uint32_t* meta = meta_block;
    uint32_t* work_hist = meta;
    uint32_t* cursor = meta + SEG_BINS;
    uint32_t* total = meta + 2 * SEG_BINS;
    uint32_t* counters = meta + 2 * SEG_BINS + 1;  // [0] split rows, [1] overflow slots
    hipLaunchKernelGGL(k_work_hist, dim3((rows + 1023) / 1024), dim3(1024), 0, st, row_ptr, L, rows, SEG, work_hist, row_ovf_base, counters, split_rows, rv, wprio);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_work_scatter, dim3((rows + 1023) / 1024), dim3(1024), 0, st, row_ptr, L, rows, SEG, (const uint32_t*)work_hist, cursor, total, work, rv, wprio);
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->before_accumulate) {  // must run before the wait below is queued: the wait binds to the event's latest record
      std::function<int()> f;
      f.swap(ctx->before_accumulate);
      const int hook_rc = f();
      if (hook_rc) return hook_rc;
    }
    HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->bases_ready, 0));
    ctx->last_products = BP::MADD_PRODUCTS;
    {
      StageTimer tk(ctx, MSM377_STAGE_ACC_KERNEL, st, part);
      const dim3 grid((unsigned)((max_items + 255) / 256));
      bool launched = false;
      if constexpr (std::is_same<BP, CV>::value && std::is_same<CV, TeDev>::value) {
        if (quad_acc) {
          hipLaunchKernelGGL(k_accumulate_quad<CV>, dim3((unsigned)((4 * max_items + 255) / 256)), dim3(256), 0, st, row_ptr, val_idx, bases, buckets, n, work, total,
                             row_ovf_base, ovf, SEG, d_err, ctx->d_err + 2, ph.into ? 1u : 0u, L);
          launched = true;
        }
      }
      if (launched) {
      } else if constexpr (!std::is_same<BP, CV>::value)
        hipLaunchKernelGGL((k_accumulate<CV, 2, BP>), grid, dim3(256), 0, st, row_ptr, val_idx, bases, buckets, n, work, total, row_ovf_base, ovf, SEG, d_err,
                           ctx->d_err + 2, ph.into ? 1u : 0u, ph.table_stride, L, rv);
      else
        hipLaunchKernelGGL((k_accumulate<CV, 2>), grid, dim3(256), 0, st, row_ptr, val_idx, bases, buckets, n, work, total, row_ovf_base, ovf, SEG, d_err,
                           ctx->d_err + 2, ph.into ? 1u : 0u, ph.table_stride, L, rv);
    }
    HIP_TRY(ctx, hipGetLastError());
    if (part == 0) ctx->acc_seq++;
    static_assert(CV::HAS_QUAD, "every curve policy has the quad-cooperative addition");
    // (grid-stride over the split-row list: with the even windows few rows split, and 8192 workgroups that only read the
    // count cost 11 us at 2^20)
    hipLaunchKernelGGL(k_merge_split_rows_quad<CV>, dim3(std::min<uint32_t>((rows + 63) / 64, 1024u)), dim3(256), 0, st, row_ptr, buckets, counters, split_rows, row_ovf_base, ovf, SEG, d_err, L, rv,
                       part == 0 ? ctx->dm_out_flag + ACC_FLAG_WORD : (uint32_t*)nullptr, ctx->acc_seq);
    HIP_TRY(ctx, hipGetLastError());
  }  // ph.accumulate
  }  // ph.front
  if (!ph.back) return MSM377_OK;
  if (ctx->capture) {
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_buckets_snap, buckets, (size_t)wc * CV::BKT_WORDS * NB * 4, hipMemcpyDeviceToDevice, st));
  }
  {
    StageTimer t(ctx, MSM377_STAGE_REDUCE, st, part);
    const uint32_t wc_acc = wc;  // window slots the accumulation filled
    uint32_t wc = wc_acc;        // window slots left to reduce (shadows the parameter copy on purpose)
    if (ph.table) {
      for (uint32_t half = wc_acc / 2; half >= 1; half /= 2) {  // wc_acc = 16: a power of two
        hipLaunchKernelGGL(k_fold_windows<CV>, dim3(half * NB / 256), dim3(256), 0, st, buckets, L, half, d_err);
        HIP_TRY(ctx, hipGetLastError());
      }
      wc = 1;
    }
    const uint32_t levels = L;  // log2 of the buckets per window
    const uint32_t first_level = 0;
    uint32_t coop_from = 0;
    for (coop_from = 0; coop_from < levels && 4ull * (coop_from + 1) * (NB >> (coop_from + 1)) * wc > ctx->coop_threads; coop_from++) {
      }
    // Levels [0, coop_from): one thread per addition (VALU-bound: 2^18 additions per level at first); [coop_from,
    // tail_from): one lane quad per addition, one launch per level; [tail_from, levels): k_reduce_tail, one launch.
    // (wide windows: 2^19 buckets in one window -- the lists of the single-launch tail must be down to one round of
    // 128 lane quads, which is level L - 8)
    const uint32_t tail_from = CV::HAS_QUAD ? std::min(wide ? levels - 8 : narrow ? ctx->narrow_tail_from : ctx->tail_from, levels) : levels;
    // (Fusing pairs of thread-level levels -- four buckets a quarter-list apart per thread, four additions, three
    // stores -- halves their HBM traffic and was slower all the same: reduce 0.290 -> 0.310 ms at 2^20, 0.278 -> 0.296
    // at 2^16.  The first levels are VALU-bound at two waves per SIMD, the later ones cost one addition's latency
    // per launch; a thread with four serial additions only lengthens that.)
    // More waves do not help either: k_tree_step at 3 / 4 waves per SIMD (132 VGPRs, no scratch) reduces in 0.300 /
    // 0.32 ms against 0.298 at 2; lane quads for levels 0-4 (MSM377_COOP_THREADS up to 2^20 threads) in 0.36.
    for (uint32_t r = first_level; r < tail_from; r++) {
      const uint32_t ops = (r + 1) * (NB >> (r + 1));
      bool done = false;
      if constexpr (CV::HAS_QUAD) {
        if (r >= coop_from) {
          hipLaunchKernelGGL(k_tree_step_quad<CV>, dim3((4 * ops + 255) / 256, wc), dim3(256), 0, st, buckets, L, r, ops, d_err);
          done = true;
        }
      }
      if (!done) hipLaunchKernelGGL(k_tree_step<CV>, dim3((ops + 255) / 256, wc), dim3(256), 0, st, buckets, L, r, ops, d_err);
      HIP_TRY(ctx, hipGetLastError());
    }
    if constexpr (CV::HAS_QUAD) {
      if (tail_from < levels) {
        const size_t lds_bytes = (size_t)(NB >> tail_from) * CV::PT_WORDS * 4;  // the stretch of buckets a workgroup works on
        if (ctx->tail_lds && lds_bytes > 64 * 1024 && lds_bytes <= TAIL_LDS_BYTES_MAX)  // (only with MSM377_TAIL_FROM below its default; per device, so every time)
          HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_reduce_tail_lds<CV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TAIL_LDS_BYTES_MAX));
        if (ctx->tail_lds && lds_bytes <= TAIL_LDS_BYTES_MAX)
          hipLaunchKernelGGL(k_reduce_tail_lds<CV>, dim3(tail_from + 1, wc), dim3(TAIL_THREADS), lds_bytes, st, buckets, L, tail_from, d_err);
        else
          hipLaunchKernelGGL(k_reduce_tail<CV>, dim3(tail_from + 1, wc), dim3(TAIL_THREADS), 0, st, buckets, L, tail_from, d_err);
        HIP_TRY(ctx, hipGetLastError());
      }
    }
    const uint32_t pp = wide ? WIDE_POINTS : (uint32_t)MSM377_G1_PARTIAL_POINTS;  // points per window record
    if (ctx->zc_active)  // set by enqueue_windows for this call: one part, slot 0
      hipLaunchKernelGGL(k_gather_partials<CV>, dim3((wc * pp * 4 + 63) / 64), dim3(64), 0, st, buckets, d_partials, wc, L, ctx->dm_partials,
                         ctx->dm_out_flag, ctx->d_out_count, (const int*)d_err, ctx->out_seq, pp);
    else
      hipLaunchKernelGGL(k_gather_partials<CV>, dim3((wc * pp * 4 + 63) / 64), dim3(64), 0, st, buckets,
                         d_partials + (size_t)pv.ws0 * MSM377_G1_PARTIAL_POINTS * CV::OUT_WORDS, wc, L, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                         (const int*)nullptr, 0u, pp);
    HIP_TRY(ctx, hipGetLastError());
  }
  return MSM377_OK;
}

// Enqueue stages decompose .. gather for windows [wb, wb + wc) against ctx->d_bases, the D2H of
// the partial records into slot `slot` of ctx->h_partials and that slot's completion event.
// Nothing here waits for the GPU.
//
// (Rounds 1 and 2 could run a large call as TWO parts of half the windows on two streams, so that one part's sort and
// reduction hid under the other's accumulation: 3.19 vs 3.17 ms at 2^20, then 2.72 -> 2.89 on round 2's kernels -- the
// accumulation kernel owns every VGPR of the chip, kernels of another stream do not become co-resident.  Removed in
// round 3; PartView keeps the stream and the window slots of a call together.)
template <class CV, class BP = CV>
int enqueue_windows(msm377_ctx* ctx, const uint32_t* d_scalars, uint64_t n_scalars, uint32_t wb, uint32_t wc, int slot, bool glv = false,
                    const Phase& ph = Phase()) {
  // GLV front end: n_scalars scalars become 2 n_scalars (point, half-scalar) columns over 8 windows.
  const uint64_t n = glv ? 2 * n_scalars : n_scalars;
  hipStream_t st = ctx->stream;
  int* d_err = ctx->d_err + slot;
  uint32_t* d_partials = ctx->d_partials + (size_t)slot * SLOT_WORDS;
  // the error word is cleared by the call's first kernel together with its counters -- unless there is no such kernel
  // (back phase only)
  const bool clear_here = ph.clear_err && !ph.front;
  if (clear_here) hipLaunchKernelGGL(k_clear_words, dim3(1), dim3(256), 0, st, (uint32_t*)d_err, 1u, (uint32_t*)nullptr, 0u);
  Phase part_phase = ph;
  part_phase.clear_err = ph.clear_err && !clear_here;
  const PartView pv{st, 0, 0, wb, wc, 0, 0};
  ctx->zc_active = ph.zc_out && ph.back && ctx->zc_out && slot == 0 && !ph.table;
  if (ctx->zc_active) ctx->out_seq++;
  {
    const int rc = enqueue_part<CV, BP>(ctx, d_scalars, n_scalars, n, pv, d_err, d_partials, glv, MAX_SORT_BLOCKS, part_phase);
    if (rc) return rc;
  }
  if (!ph.back) return MSM377_OK;
  const uint32_t wc_out = ph.table ? 1u : wc;  // precomputed-window tables fold the windows on the GPU
  const uint32_t pp_out = ph.wide ? WIDE_POINTS : (uint32_t)MSM377_G1_PARTIAL_POINTS;
  if (!ctx->zc_active) {
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_partials + (size_t)slot * SLOT_WORDS, d_partials, (size_t)wc_out * pp_out * CV::OUT_WORDS * 4,
                                 hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_err + slot, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(ctx, hipEventRecord(ctx->done_ev[slot], st));
  ctx->last_n = n;
  ctx->last_wc = wc;
  ctx->last_glv = glv;
  ctx->last_form = CV::FORM_ID;
  return MSM377_OK;
}

// Wait for slot `slot`; its partial records are then in ctx->h_partials + slot * SLOT_WORDS.
// Zero-copy output (Phase::zc_out): poll the sequence number the gather kernel's last block writes behind the records;
// the stream's completion event is waited for only when stage timing needs it (or after 50 ms without the flag, which
// then also surfaces a failed kernel).
int wait_zero_copy_out(msm377_ctx* ctx) {
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t spins = 0; __atomic_load_n(&ctx->h_out_flag[0], __ATOMIC_ACQUIRE) != ctx->out_seq; spins++) {
    __builtin_ia32_pause();
    if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) {
      HIP_TRY(ctx, hipEventSynchronize(ctx->done_ev[0]));
      break;
    }
  }
  if (__atomic_load_n(&ctx->h_out_flag[0], __ATOMIC_ACQUIRE) != ctx->out_seq) {
    ctx->err = "zero-copy output: the window records did not arrive";
    return MSM377_EHIP;
  }
  ctx->h_err[0] = (int)__atomic_load_n(&ctx->h_out_flag[1], __ATOMIC_RELAXED);
  return MSM377_OK;
}

int finish_windows(msm377_ctx* ctx, int slot) {
  if (ctx->zc_active && slot == 0) {
    const int rc = wait_zero_copy_out(ctx);
    if (rc) return rc;
    if (ctx->timing) HIP_TRY(ctx, hipEventSynchronize(ctx->done_ev[slot]));
  } else {
    HIP_TRY(ctx, hipEventSynchronize(ctx->done_ev[slot]));
  }
  if (ctx->timing) {
    for (int s = 0; s < MSM377_NUM_STAGES; s++) {
      if (s == MSM377_STAGE_TAIL) continue;  // host wall time, set by the caller
      if (ctx->timing == 2 && s != MSM377_STAGE_ACC_KERNEL) {
        ctx->stage_ms[s] = 0.0;
        continue;
      }
      double sum = 0.0;  // a pipelined call reports the sum over its two parts (they overlap each other in wall time)
      for (uint32_t p = 0; p < 1u; p++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev[p][s][0], ctx->ev[p][s][1]) == hipSuccess) sum += ms;
      }
      ctx->stage_ms[s] = sum;
    }
    (void)hipGetLastError();  // a stage that did not run in this call must not leave its error for the next launch check
  }
  if (ctx->h_err[slot] & ERR_SCALAR) {
    ctx->err = "a scalar overflows the signed 16-bit window recode (final carry)";
    return MSM377_ESCALAR;
  }
  return MSM377_OK;
}

template <class CV>
int run_windows(msm377_ctx* ctx, const uint32_t* d_scalars, uint64_t n, uint32_t wb, uint32_t wc, bool glv = false) {
  int rc = enqueue_windows<CV>(ctx, d_scalars, n, wb, wc, 0, glv);
  if (rc) return rc;
  return finish_windows(ctx, 0);
}

inline bool use_glv(const msm377_ctx* ctx, uint64_t) { return ctx->glv_mode == 1; }  // see msm377_ctx::glv_mode

// Base conversion for the G1 entry points: with the GLV front end the table also gets phi(P_i).
int convert_bases_g1(msm377_ctx* ctx, const uint32_t* d_raw, uint64_t n, bool glv) {
  if (!glv) return convert_bases<G1Dev>(ctx, d_raw, n);
  if (n == 0) return MSM377_OK;
  if (ctx->timing == 1) (void)hipEventRecord(ctx->ev[0][MSM377_STAGE_CONVERT][0], ctx->stream2);
  hipLaunchKernelGGL(k_clear_words, dim3(1), dim3(256), 0, ctx->stream2, (uint32_t*)(ctx->d_err + 2), 1u, (uint32_t*)nullptr, 0u);
  hipLaunchKernelGGL(k_convert_bases_glv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream2, d_raw, ctx->d_bases, n);
  HIP_TRY(ctx, hipGetLastError());
  if (ctx->timing == 1) (void)hipEventRecord(ctx->ev[0][MSM377_STAGE_CONVERT][1], ctx->stream2);
  HIP_TRY(ctx, hipEventRecord(ctx->bases_ready, ctx->stream2));
  return MSM377_OK;
}

// What ctx->d_bases holds for the G1 entry points.
enum TableForm { TABLE_XYZZ = 0, TABLE_XYZZ_GLV = 1, TABLE_TE = 2, TABLE_TE_AFFINE = 3, TABLE_TE_PRECOMP = 4 };
inline bool form_is_te(int form) { return form == TABLE_TE || form == TABLE_TE_AFFINE || form == TABLE_TE_PRECOMP; }
constexpr int RC_TE_FALLBACK = 1;  // internal: an exceptional case of the twisted Edwards law, rerun on the Weierstrass path

// A prefix of a GLV table (records 0..n-1 = the plain points) serves the plain path; the phi half needs all of it.
inline int resident_form(const msm377_ctx* ctx, uint64_t n) {
  return (ctx->bases_form == TABLE_XYZZ_GLV && n != ctx->bases_n) ? TABLE_XYZZ : ctx->bases_form;
}
inline int weierstrass_form(const msm377_ctx* ctx, uint64_t n) { return use_glv(ctx, n) ? TABLE_XYZZ_GLV : TABLE_XYZZ; }
inline int pick_form(const msm377_ctx* ctx, uint64_t n) { return ctx->g1_form == 1 ? TABLE_TE : weierstrass_form(ctx, n); }

int convert_table(msm377_ctx* ctx, const uint32_t* d_raw, uint64_t n, int form) {
  if (form == TABLE_TE) return convert_bases<TeDev>(ctx, d_raw, n);
  if (form == TABLE_TE_AFFINE) {  // resident tables: both phases back to back (the caller waits for the side stream anyway)
    const int rc = affine_convert_begin(ctx, d_raw, n);
    return rc ? rc : affine_convert_finish(ctx, ctx->d_bases, n);
  }
  return convert_bases_g1(ctx, d_raw, n, form == TABLE_XYZZ_GLV);
}

void time_tail(msm377_ctx* ctx, std::chrono::steady_clock::time_point t0) {
  ctx->stage_ms[MSM377_STAGE_TAIL] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// Full G1 MSM of n scalars against ctx->d_bases in form `form` (already converted or being converted on the
// side stream).  TABLE_TE: 16 windows in twisted Edwards form; RC_TE_FALLBACK when an addition or an input point
// hit an exceptional case (the caller reconverts and reruns).  TABLE_XYZZ_GLV: the GLV front end; a scalar outside
// its range (bit 1 of the error word) reruns on the plain 16-window path, whose records 0..n-1 of the table are
// the plain points either way.
// Arms the tail workers of one call once its accumulation kernel is through (TailPool::prewake: they poll for their
// jobs while the GPU reduces the buckets); disarmed when the tail is done.
struct TailArm {
  msm377_ctx* c;
  bool armed = false;
  explicit TailArm(msm377_ctx* ctx) : c(ctx) {}
  void arm() {
    if (armed || c->tail_threads <= 1 || c->tail_spin_us <= 0) return;
    armed = true;
    c->tail_pool.prewake(c->tail_spin_us, std::min(c->tail_threads, TailPool::WORKERS + 1) - 1);
  }
  // A small input is over in a few hundred microseconds, its bucket reduction in 0.1 ms -- no longer than a sleeping
  // worker may take to come back -- so its call arms the workers before it enqueues anything.
  void at_start(uint64_t n) {
    if (n <= (1ull << 17)) arm();
  }
  void after_accumulation() {
    if (armed || c->tail_threads <= 1 || c->tail_spin_us <= 0) return;
    // k_merge_split_rows_quad, the launch behind the accumulation kernel, writes the call's sequence number; without it
    // within 100 ms the caller's own wait reports whatever went wrong
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; __atomic_load_n(&c->h_out_flag[ACC_FLAG_WORD], __ATOMIC_ACQUIRE) != c->acc_seq; spins++) {
      __builtin_ia32_pause();
      if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(100)) return;
    }
    arm();
  }
  ~TailArm() { c->tail_pool.disarm(); }
};

int g1_table_msm(msm377_ctx* ctx, const uint32_t* d_scalars, uint64_t n, int form, uint8_t out_xy[96]) {
  TailArm arm(ctx);
  arm.at_start(n);
  if (form_is_te(form)) {
    Phase ph;
    if (form == TABLE_TE_PRECOMP) {
      ph.table = ctx->d_table;
      ph.table_stride = ctx->bases_n;
    }
    // Small inputs: narrow windows (k_decompose_narrow); the window-indexed buffers are sized for them too
    // (msm377_ctx_create: wcap).  Stage read-backs describe the 16-bit geometry.
    bool narrow = form != TABLE_TE_PRECOMP && n <= ctx->narrow_max_points && n <= SMALL_SORT_MAX && !ctx->capture;
    bool wide = form == TABLE_TE_PRECOMP && ctx->table_window_bits == WIDE_BITS;
    bool table0 = false;  // the 16-window path over window 0 of the wide table (= the affine records of the points themselves)
    bool even = ctx->even_windows && form != TABLE_TE_PRECOMP && !ctx->capture;  // (stage read-backs describe sixteen equal windows)
    for (;;) {
      uint32_t windows = MSM377_NUM_WINDOWS;
      int cbits = MSM377_WINDOW_BITS, planes = MSM377_WINDOW_BITS - 1, short_from = 0;  // the host tail's view of the geometry
      ph.cbits = MSM377_WINDOW_BITS;
      ph.bucket_log = MSM377_WINDOW_BITS - 1;
      if (wide) {  // one window slot of 2^19 buckets over the 13-window table
        ph.wide = true;
        ph.cbits = WIDE_BITS;
        ph.bucket_log = WIDE_LOG;
        windows = 1;
        cbits = WIDE_BITS;
        planes = WIDE_LOG;
      } else if (narrow && even && ctx->narrow_even) {  // eleven signed 12-bit + eleven unsigned 11-bit windows
        ph.cbits = NARROW_EVEN_BITS;
        ph.bucket_log = NARROW_LOG;
        windows = NARROW_EVEN_WINDOWS;
        cbits = NARROW_EVEN_BITS;
        planes = NARROW_LOG;
        short_from = NARROW_EVEN_SIGNED;
      } else if (narrow) {  // 22 signed 11-bit windows + an unsigned top one
        ph.cbits = NARROW_BITS;
        ph.bucket_log = NARROW_LOG;
        windows = NARROW_WINDOWS;
        cbits = NARROW_BITS;
        planes = NARROW_LOG;
      }
      ph.even = even && !wide && !table0 && (!narrow || ctx->narrow_even);
      if (ph.even && !narrow) short_from = EVEN_FROM;
      ph.zc_out = true;
      int rc = form == TABLE_TE ? enqueue_windows<TeDev>(ctx, d_scalars, n, 0, windows, 0, false, ph)
                                : enqueue_windows<TeDev, TeAffBase>(ctx, d_scalars, n, 0, windows, 0, false, ph);
      if (rc) return rc;
      arm.after_accumulation();
      if (ctx->zc_active) {
        rc = wait_zero_copy_out(ctx);
        if (rc) return rc;
      } else {
        HIP_TRY(ctx, hipEventSynchronize(ctx->done_ev[0]));
      }
      if (narrow && (ctx->h_err[0] & ERR_NARROW_RANGE) && !(ctx->h_err[0] & ERR_SCALAR)) {  // a scalar >= 2^253: sixteen 16-bit windows take it
        narrow = false;
        even = false;
        continue;
      }
      if (ph.even && (ctx->h_err[0] & ERR_NARROW_RANGE) && !(ctx->h_err[0] & ERR_SCALAR)) {  // likewise: sixteen equal windows take it
        even = false;
        continue;
      }
      if (wide && (ctx->h_err[0] & ERR_NARROW_RANGE) && !(ctx->h_err[0] & ERR_SCALAR)) {  // likewise: its top window holds 19 bits
        wide = false;
        table0 = true;
        ph.wide = false;
        ph.table = nullptr;
        ph.table_stride = 0;  // every window slot gathers from the same records
        ph.bases_override = ctx->d_table;
        continue;
      }
      if (ctx->h_err[0] & ERR_TE_ANY) {
        note_fallback(ctx, (uint32_t)(ctx->h_err[0] & ERR_TE_ANY));
        return RC_TE_FALLBACK;
      }
      rc = finish_windows(ctx, 0);
      if (rc) return rc;
      auto t0 = std::chrono::steady_clock::now();
      const int tr = form == TABLE_TE_PRECOMP && !table0 ? (teh_combine(ctx->h_partials, 1, out_xy, cbits, planes) ? TAIL_EXCEPTIONAL : TAIL_OK)
                                              : te_tail(ctx, ctx->h_partials, out_xy, (int)windows, cbits, planes, short_from);
      time_tail(ctx, t0);
      if (tr < 0) return tr;
      if (tr == TAIL_EXCEPTIONAL) note_fallback(ctx, MSM377_FB_TAIL);
      return tr == TAIL_EXCEPTIONAL ? RC_TE_FALLBACK : MSM377_OK;
    }
  }
  if (form == TABLE_XYZZ_GLV) {
    int rc = enqueue_windows<G1Dev>(ctx, d_scalars, n, 0, GLV_WINDOWS, 0, true);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventSynchronize(ctx->done_ev[0]));
    if ((ctx->h_err[0] & ERR_GLV_RANGE) == 0) {
      rc = finish_windows(ctx, 0);
      if (rc) return rc;
      auto t0 = std::chrono::steady_clock::now();
      g1h_combine(ctx->h_partials, GLV_WINDOWS, out_xy);
      time_tail(ctx, t0);
      return MSM377_OK;
    }
  }
  int rc = run_windows<G1Dev>(ctx, d_scalars, n, 0, MSM377_NUM_WINDOWS);
  if (rc) return rc;
  auto t0 = std::chrono::steady_clock::now();
  rc = xyzz_tail(ctx, ctx->h_partials, out_xy);
  time_tail(ctx, t0);
  return rc;
}

// The resident table hit an exceptional case of the Edwards law: rebuild it in Weierstrass form from the raw
// copy kept by msm377_g1_set_bases_device.
int resident_table_to_weierstrass(msm377_ctx* ctx) {
  const int form = TABLE_XYZZ;  // points outside the prime-order subgroup: never the GLV front end
  int rc = convert_table(ctx, ctx->d_raw_points, ctx->bases_n, form);
  if (rc) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2));
  ctx->bases_form = form;
  return MSM377_OK;
}

// Host-buffer entry points with large inputs: the upload (3.0-3.6 ms for the 128 MB of a 2^20-point G1 input from
// pageable memory, box to box) is longer than the whole computation, so the two overlap: the MSM runs as K chunks of
// points, later chunks accumulate on top of the buckets the earlier ones left (Phase::into), reduction, gather and D2H
// are queued once, with the last chunk.  Two schedules:
//
// run_chunked_upload (DEFAULT): a chunk's scalars AND points go up together and the chunk runs the whole front end --
// decompose .. accumulate .. merge -- while the next one is on its way.
//
// run_sorted_upload (MSM377_UPLOAD_SORT_ONCE=1): all scalars first, ONE decomposition and sort with
// the rows filed by chunk of the point index (k_local_sort_lds<true>: K sub-row bounds per key, common.hpp RowView), then
// per chunk of points only its base conversion, a work list over ITS sub-rows, the accumulation and the merge.  Built,
// parity-green and NOT faster: same box, interleaved (profiles/r03_final/ab_upload.txt) 4.44 / 4.45 / 4.72 ms for the
// per-chunk front ends against 4.67 / 4.67 / 4.74 sorted once.  The trace (MSM377_UPLOAD_TRACE=1,
// profiles/r03_final/upload_trace.txt) says why: the call is bound by the GPU, not by the upload -- the chunks'
// accumulations with projective records cost ~2.7 ms per 2^20 points (0.65 ms per 24 % chunk: short rows, a bucket
// load and store per item and chunk, work list + merge per chunk) plus reduction and tail, and that work cannot start
// before the first points are on the device; the per-chunk sorts the new schedule saves (~0.1 ms each) were hidden
// behind the upload anyway, while its scalars-first head (0.78 ms of upload + 0.18 ms of sort before the first
// accumulation) is not.
// The chunks' decompositions use the even window geometry in the Edwards forms (g1_msm / ed_msm hand EVEN_FROM to the
// tail and rerun in one piece when a scalar does not fit).
template <class CV>
inline bool upload_even(const msm377_ctx* ctx) {
  return (std::is_same<CV, TeDev>::value || std::is_same<CV, EdDev>::value) && ctx->even_windows;
}

template <class CV>
int run_chunked_upload(msm377_ctx* ctx, const uint8_t* points, const uint8_t* scalars, uint64_t n) {
  constexpr size_t PB = CV::RAW_WORDS * 4;  // bytes per wire point
  uint64_t cut[10];  // chunk c = points [cut[c], cut[c + 1]): the first one upload_split_pct of n, the rest even
  uint32_t K = 0;
  cut[0] = 0;
  for (uint32_t c = 1; c < ctx->upload_chunks; c++) {
    const uint64_t first_end = std::max<uint64_t>(64, (n * ctx->upload_split_pct / 100) & ~63ull);
    const uint64_t b = c == 1 ? first_end : (first_end + (n - first_end) * (c - 1) / (ctx->upload_chunks - 1)) & ~63ull;
    if (b > cut[K] && b < n) cut[++K] = b;  // no empty chunks (small n)
  }
  cut[++K] = n;
  const size_t sc_stage = (size_t)ctx->cap * 96;
  auto upload_chunk = [&](uint32_t c) -> int {
    const uint64_t first = cut[c], cnt = cut[c + 1] - cut[c];
    int r = h2d_staged(ctx, (uint8_t*)ctx->d_raw_scalars + first * 32, scalars + first * 32, cnt * 32, sc_stage + first * 32);
    if (r == MSM377_OK) r = h2d_staged(ctx, (uint8_t*)ctx->d_raw_points + first * PB, points + first * PB, cnt * PB, first * PB);
    return r;
  };
  int rc = upload_chunk(0);
  if (rc) return rc;
  int up_rc = MSM377_OK;
  std::atomic<uint32_t> uploaded{1};  // chunks on the device so far
  std::atomic<bool> upload_done{false};
  std::thread upload([&] {
    if (hipSetDevice(ctx->device) != hipSuccess) up_rc = MSM377_EHIP;
    for (uint32_t c = 1; c < K && up_rc == MSM377_OK; c++) {
      up_rc = upload_chunk(c);
      if (up_rc == MSM377_OK) uploaded.store(c + 1, std::memory_order_release);
    }
    upload_done.store(true, std::memory_order_release);
  });
  for (uint32_t c = 0; c < K && rc == MSM377_OK; c++) {
    while (uploaded.load(std::memory_order_acquire) <= c && !upload_done.load(std::memory_order_acquire)) std::this_thread::yield();
    if (uploaded.load(std::memory_order_acquire) <= c) {  // the upload thread stopped on an error
      rc = up_rc ? up_rc : MSM377_EHIP;
      break;
    }
    const uint64_t first = cut[c], cnt = cut[c + 1] - cut[c];
    Phase ph;
    ph.clear_err = c == 0;
    ph.into = c > 0;
    ph.back = c + 1 == K;
    ph.base_first = first;
    ph.even = upload_even<CV>(ctx);
    rc = convert_bases<CV>(ctx, ctx->d_raw_points + first * CV::RAW_WORDS, cnt, first, c == 0);
    if (rc == MSM377_OK) rc = enqueue_windows<CV>(ctx, ctx->d_raw_scalars + first * 8, cnt, 0, MSM377_NUM_WINDOWS, 0, false, ph);
  }
  upload.join();
  if (rc) (void)hipStreamSynchronize(ctx->stream);
  return rc;
}

template <class CV>
int run_sorted_upload(msm377_ctx* ctx, const uint8_t* points, const uint8_t* scalars, uint64_t n) {
  constexpr size_t PB = CV::RAW_WORDS * 4;  // bytes per wire point
  // chunk c = points [cut[c], cut[c + 1]): the first one upload_split_pct of n (it should land when the sort is through),
  // the last one half a share (only ITS accumulation trails the upload), the rest even; multiples of 64 points
  ChunkCuts cuts;
  {
    const uint32_t want = std::min<uint32_t>(ctx->upload_chunks + 1, MAX_UPLOAD_CHUNKS);  // one chunk more than the default schedule: its last one is half a share
    const uint64_t first_end = std::max<uint64_t>(64, (n * (ctx->upload_split_pct * 5 / 9) / 100) & ~63ull);  // 30 % -> 16 %: the first chunk should land when the sort is through
    const double rest = (double)(n - std::min(n, first_end)), shares = want > 2 ? (double)(want - 2) + 0.5 : 1.0;
    uint32_t K = 0;
    cuts.cut[0] = 0;
    double at = (double)first_end;
    for (uint32_t c = 1; c < want; c++) {
      const uint64_t bnd = c == 1 ? first_end : ((uint64_t)at) & ~63ull;
      if (bnd > cuts.cut[K] && bnd < n) cuts.cut[++K] = (uint32_t)bnd;  // no empty chunks (small n)
      at += rest / shares;
    }
    cuts.k = K + 1;
    for (uint32_t j = cuts.k; j <= MAX_UPLOAD_CHUNKS; j++) cuts.cut[j] = (uint32_t)n;
  }
  const uint32_t K = cuts.k;
  if (K > 1 && !ctx->d_row_ptr_chunks) {  // K sub-row bounds per key: allocated the first time a call needs them
    if (hipMalloc((void**)&ctx->d_row_ptr_chunks, (size_t)MSM377_NUM_WINDOWS * ((size_t)(NB + 1) * MAX_UPLOAD_CHUNKS + 1) * 4) != hipSuccess) {
      ctx->d_row_ptr_chunks = nullptr;
      (void)hipGetLastError();
      ctx->err = "chunked upload: out of device memory for the row bounds";
      return MSM377_ENOMEM;
    }
  }
  // MSM377_UPLOAD_TRACE=1: host timestamps (us after the call) of the upload and enqueue steps and GPU timestamps of the
  // phases' ends (HIP events on the main stream), printed by upload_trace_report once the call is through.
  UploadTrace& tr = ctx->upload_trace;
  tr.begin(ctx->upload_trace_on, K);
  int rc = h2d_staged(ctx, ctx->d_raw_scalars, scalars, n * 32, (size_t)ctx->cap * 96);
  if (rc) return rc;
  tr.host("scalars up");
  int up_rc = MSM377_OK;
  std::atomic<uint32_t> uploaded{0};  // chunks of points on the device so far
  std::atomic<bool> upload_done{false};
  std::thread upload([&] {
    if (hipSetDevice(ctx->device) != hipSuccess) up_rc = MSM377_EHIP;
    for (uint32_t c = 0; c < K && up_rc == MSM377_OK; c++) {
      const uint64_t first = cuts.cut[c], cnt = cuts.cut[c + 1] - first;
      up_rc = h2d_staged(ctx, (uint8_t*)ctx->d_raw_points + first * PB, points + first * PB, cnt * PB, first * PB);
      tr.chunk_up(c);
      if (up_rc == MSM377_OK) uploaded.store(c + 1, std::memory_order_release);
    }
    upload_done.store(true, std::memory_order_release);
  });
  {  // phase 1: decompose + sort, once
    Phase ph;
    ph.accumulate = false;
    ph.back = false;
    ph.cuts = cuts;
    ph.even = upload_even<CV>(ctx);
    tr.gpu(ctx->stream, 0);
    rc = enqueue_windows<CV>(ctx, ctx->d_raw_scalars, n, 0, MSM377_NUM_WINDOWS, 0, false, ph);
    tr.gpu(ctx->stream, 1);
    tr.host("sort enqueued");
  }
  for (uint32_t c = 0; c < K && rc == MSM377_OK; c++) {
    while (uploaded.load(std::memory_order_acquire) <= c && !upload_done.load(std::memory_order_acquire)) std::this_thread::yield();
    if (uploaded.load(std::memory_order_acquire) <= c) {  // the upload thread stopped on an error
      rc = up_rc ? up_rc : MSM377_EHIP;
      break;
    }
    const uint64_t first = cuts.cut[c], cnt = cuts.cut[c + 1] - first;
    Phase ph;
    ph.clear_err = false;
    ph.sort = false;
    ph.into = c > 0;
    ph.back = c + 1 == K;
    ph.cuts = cuts;
    ph.chunk = c;
    ph.chunk_points = cnt;
    rc = convert_bases<CV>(ctx, ctx->d_raw_points + first * CV::RAW_WORDS, cnt, first, c == 0);
    if (rc == MSM377_OK) rc = enqueue_windows<CV>(ctx, ctx->d_raw_scalars, n, 0, MSM377_NUM_WINDOWS, 0, false, ph);
    tr.gpu(ctx->stream, 2 + c);
    tr.chunk_enqueued(c);
  }
  upload.join();
  if (rc) (void)hipStreamSynchronize(ctx->stream);
  return rc;
}

int check_args(msm377_ctx* ctx, const void* a, const void* b, uint64_t n, bool need_a) {
  if (!ctx) return MSM377_EINVAL;
  ctx->err.clear();
  if (n > ctx->cap) {
    ctx->err = "n exceeds the context capacity";
    return MSM377_EINVAL;
  }
  if (n && ((need_a && !a) || !b)) {
    ctx->err = "null input pointer";
    return MSM377_EINVAL;
  }
  if (((uintptr_t)a & 15) || ((uintptr_t)b & 15)) {
    ctx->err = "device input pointers must be 16-byte aligned";
    return MSM377_EINVAL;
  }
  return MSM377_OK;
}



}  // namespace

// ---- entry points (C ABI: capi.hip forwards) ----

// The pinned staging buffer (128 bytes per point of capacity) and the copy streams of the host-buffer entry points.
// h2d_staged allocates them on first use -- which made the FIRST host-buffer call of a context ~35 ms; a caller that
// cares calls this right after msm377_ctx_create.
int reserve_host_staging(msm377_ctx* ctx) {
  if (!ctx) return MSM377_EINVAL;
  if (ctx->h_stage) return MSM377_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (hipHostMalloc((void**)&ctx->h_stage, (size_t)ctx->cap * 128) != hipSuccess) {
    ctx->h_stage = nullptr;
    ctx->err = "host staging buffer: out of pinned memory";
    return MSM377_ENOMEM;
  }
  for (int t = 0; t < 8; t++)
    if (!ctx->copy_stream[t]) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream[t], hipStreamNonBlocking));
  return MSM377_OK;
}

int g1_msm_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint8_t out_xy[96]) {
  if (!out_xy) return MSM377_EINVAL;
  int rc = check_args(ctx, d_points, d_scalars, n, true);
  if (rc) return rc;
  if (n == 0) {
    identity_wire(out_xy);
    return MSM377_OK;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  ctx->bases_n = 0;
  int form = pick_form(ctx, n);
  // (Queueing the conversion after k_decompose instead was measured: decompose 77 -> 23 us, sort 272 -> 386 us.)
  if (form == TABLE_TE && ctx->te_affine_msm && n >= ctx->affine_min_points) {
    // Affine records (7-product additions) by the batched conversion: its way up is queued now, the host's inversion
    // and the way down happen from the hook, once decompose .. work list are queued on the main stream.
    form = TABLE_TE_AFFINE;
    rc = affine_convert_begin(ctx, (const uint32_t*)d_points, n);
    if (rc) return rc;
    ctx->before_accumulate = [ctx, n]() -> int { return affine_convert_finish(ctx, ctx->d_bases, n, true); };
    rc = g1_table_msm(ctx, (const uint32_t*)d_scalars, n, form, out_xy);
    ctx->before_accumulate = nullptr;
  } else {
    rc = convert_table(ctx, (const uint32_t*)d_points, n, form);
    if (rc) return rc;
    rc = g1_table_msm(ctx, (const uint32_t*)d_scalars, n, form, out_xy);
  }
  if (rc != RC_TE_FALLBACK) return rc;
  form = TABLE_XYZZ;  // an exceptional case means points outside the prime-order subgroup: never the GLV front end
  rc = convert_table(ctx, (const uint32_t*)d_points, n, form);
  if (rc) return rc;
  return g1_table_msm(ctx, (const uint32_t*)d_scalars, n, form, out_xy);
}

int g1_msm(msm377_ctx* ctx, const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out_xy[96]) {
  if (!ctx || !out_xy) return MSM377_EINVAL;
  ctx->err.clear();
  if (n > ctx->cap || (n && (!points || !scalars))) {
    ctx->err = "bad arguments";
    return MSM377_EINVAL;
  }
  if (n == 0) {
    identity_wire(out_xy);
    return MSM377_OK;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  ctx->bases_n = 0;
  int form = pick_form(ctx, n);
  const uint32_t* d_sc = ctx->d_raw_scalars;
  const uint32_t* d_pt = ctx->d_raw_points;
  int rc;
  if (n >= ctx->upload_chunk_min && form != TABLE_XYZZ_GLV && !ctx->capture) {  // (stage read-backs describe the plain row layout)
    const bool te = form == TABLE_TE;
    if (ctx->upload_sort_once)
      rc = te ? run_sorted_upload<TeDev>(ctx, points, scalars, n) : run_sorted_upload<G1Dev>(ctx, points, scalars, n);
    else
      rc = te ? run_chunked_upload<TeDev>(ctx, points, scalars, n) : run_chunked_upload<G1Dev>(ctx, points, scalars, n);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventSynchronize(ctx->done_ev[0]));
    ctx->upload_trace.report();
    const bool even = te && upload_even<TeDev>(ctx);
    if (even && (ctx->h_err[0] & ERR_NARROW_RANGE) && !(ctx->h_err[0] & (ERR_SCALAR | ERR_TE_ANY))) {
      // a scalar of 2^253 and more: everything is on the device by now, rerun in one piece (g1_table_msm falls back to
      // sixteen equal windows by itself)
      rc = convert_table(ctx, d_pt, n, form);
      if (rc) return rc;
      rc = g1_table_msm(ctx, d_sc, n, form, out_xy);
      if (rc != RC_TE_FALLBACK) return rc;
    } else if (!(te && (ctx->h_err[0] & ERR_TE_ANY))) {
      rc = finish_windows(ctx, 0);
      if (rc) return rc;
      auto t0 = std::chrono::steady_clock::now();
      const int tr = te ? te_tail(ctx, ctx->h_partials, out_xy, 16, 16, 15, even ? (int)EVEN_FROM : 0) : xyzz_tail(ctx, ctx->h_partials, out_xy);
      time_tail(ctx, t0);
      if (tr != TAIL_EXCEPTIONAL) return tr;
      note_fallback(ctx, MSM377_FB_TAIL);
    } else {
      note_fallback(ctx, (uint32_t)(ctx->h_err[0] & ERR_TE_ANY));
    }
    // exceptional case of the Edwards law: everything is on the device by now, rerun in one piece below
  } else {
    // Scalars first: decomposition, sort and the work lists need nothing else, so they run while the points (three
    // quarters of the bytes) are still on their way; the conversion is launched when the upload lands, right before
    // the accumulation is queued.
    rc = h2d_staged(ctx, ctx->d_raw_scalars, scalars, n * 32, (size_t)ctx->cap * 96);
    if (rc) return rc;
    int up_rc = MSM377_OK;
    std::thread upload([&] {
      up_rc = hipSetDevice(ctx->device) == hipSuccess ? h2d_staged(ctx, ctx->d_raw_points, points, n * 96, 0) : MSM377_EHIP;
    });
    ctx->before_accumulate = [&]() -> int {
      if (upload.joinable()) upload.join();
      if (up_rc) return up_rc;
      return convert_table(ctx, d_pt, n, form);
    };
    rc = g1_table_msm(ctx, d_sc, n, form, out_xy);
    ctx->before_accumulate = nullptr;
    if (upload.joinable()) upload.join();  // an error before the hook ran
    if (rc != RC_TE_FALLBACK) return rc;
  }
  form = TABLE_XYZZ;
  rc = convert_table(ctx, d_pt, n, form);
  if (rc) return rc;
  return g1_table_msm(ctx, d_sc, n, form, out_xy);
}

// ---- Edwards-BLS12 (BASELINE.json config 3): same pipeline, EdDev policy ----
int ed_msm_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint8_t out_xy[64]) {
  if (!out_xy) return MSM377_EINVAL;
  int rc = check_args(ctx, d_points, d_scalars, n, true);
  if (rc) return rc;
  if (n == 0) {  // the neutral element (0, 1)
    memset(out_xy, 0, 64);
    out_xy[32] = 1;
    return MSM377_OK;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  ctx->bases_n = 0;
  rc = convert_bases<EdDev>(ctx, (const uint32_t*)d_points, n);
  if (rc) return rc;
  TailArm arm(ctx);
  arm.at_start(n);
  bool even = ctx->even_windows && !ctx->capture && !ctx->ed_equal_windows_once;
  ctx->ed_equal_windows_once = false;
  for (;;) {
    Phase ph;
    ph.zc_out = true;
    ph.even = even;
    rc = enqueue_windows<EdDev>(ctx, (const uint32_t*)d_scalars, n, 0, MSM377_NUM_WINDOWS, 0, false, ph);
    if (rc) return rc;
    arm.after_accumulation();
    if (ctx->zc_active) {
      rc = wait_zero_copy_out(ctx);
      if (rc) return rc;
    } else {
      HIP_TRY(ctx, hipEventSynchronize(ctx->done_ev[0]));
    }
    if (even && (ctx->h_err[0] & ERR_NARROW_RANGE) && !(ctx->h_err[0] & ERR_SCALAR)) {  // a scalar of 2^253 and more: sixteen equal windows
      even = false;
      continue;
    }
    rc = finish_windows(ctx, 0);
    if (rc) return rc;
    auto t0 = std::chrono::steady_clock::now();
    rc = ed_tail(ctx, ctx->h_partials, out_xy, even ? (int)EVEN_FROM : 0);
    time_tail(ctx, t0);
    return rc;
  }
}

int ed_msm(msm377_ctx* ctx, const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out_xy[64]) {
  if (!ctx || !out_xy) return MSM377_EINVAL;
  ctx->err.clear();
  if (n > ctx->cap || (n && (!points || !scalars))) {
    ctx->err = "bad arguments";
    return MSM377_EINVAL;
  }
  if (n == 0) return ed_msm_device(ctx, nullptr, nullptr, 0, out_xy);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (n >= ctx->upload_chunk_min) {  // chunks of points, like g1_msm: a chunk computes while the next one uploads
    ctx->bases_n = 0;
    int rc = ctx->upload_sort_once ? run_sorted_upload<EdDev>(ctx, points, scalars, n) : run_chunked_upload<EdDev>(ctx, points, scalars, n);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventSynchronize(ctx->done_ev[0]));
    const bool even = upload_even<EdDev>(ctx);
    if (even && (ctx->h_err[0] & ERR_NARROW_RANGE) && !(ctx->h_err[0] & ERR_SCALAR)) {  // everything is on the device: once more in one piece
      ctx->ed_equal_windows_once = true;
      return ed_msm_device(ctx, ctx->d_raw_points, ctx->d_raw_scalars, n, out_xy);
    }
    rc = finish_windows(ctx, 0);
    if (rc) return rc;
    auto t0 = std::chrono::steady_clock::now();
    rc = ed_tail(ctx, ctx->h_partials, out_xy, even ? (int)EVEN_FROM : 0);
    time_tail(ctx, t0);
    return rc;
  }
  int rc = h2d_staged(ctx, ctx->d_raw_points, points, n * 64, 0);
  if (rc == MSM377_OK) rc = h2d_staged(ctx, ctx->d_raw_scalars, scalars, n * 32, (size_t)ctx->cap * 96);
  if (rc) return rc;
  return ed_msm_device(ctx, ctx->d_raw_points, ctx->d_raw_scalars, n, out_xy);
}

int ed_generate_bases_device(msm377_ctx* ctx, uint64_t seed, uint64_t n, void* d_points_out) {
  if (!ctx || (n && !d_points_out) || ((uintptr_t)d_points_out & 15)) return MSM377_EINVAL;
  if (n == 0) return MSM377_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_generate_bases_ed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, seed, n, (uint32_t*)d_points_out);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MSM377_OK;
}

// The precomputed-window table and its wide-window work buffers (allocated on demand, 2.2-2.7 GB at 2^20 points).
void free_table(msm377_ctx* ctx) {
  for (void* p : {(void*)ctx->d_table, (void*)ctx->d_wide_digits, (void*)ctx->d_wide_counts, (void*)ctx->d_wide_temp})
    if (p) (void)hipFree(p);
  ctx->d_table = nullptr;
  ctx->d_wide_digits = nullptr;
  ctx->d_wide_counts = nullptr;
  ctx->d_wide_temp = nullptr;
  ctx->table_cap = 0;
  ctx->table_windows = 0;
}

int g1_set_bases_device(msm377_ctx* ctx, const void* d_points, uint64_t n) {
  int rc = check_args(ctx, d_points, d_points, n, true);
  if (rc) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int form = pick_form(ctx, n);
  if (form == TABLE_TE) form = TABLE_TE_AFFINE;  // resident: affine records by the batched inversion, once
  rc = convert_table(ctx, (const uint32_t*)d_points, n, form);
  if (rc) return rc;
  // raw copy for the (never expected) fallback from the Edwards form: see resident_table_to_weierstrass
  if (form_is_te(form) && d_points != ctx->d_raw_points)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_raw_points, d_points, n * 96, hipMemcpyDeviceToDevice, ctx->stream2));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2));
  if (ctx->d_table) {  // a plain table replaces a precomputed one: give its gigabytes back
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    free_table(ctx);
  }
  ctx->bases_n = n;
  ctx->bases_form = form;
  return MSM377_OK;
}

int g1_set_bases(msm377_ctx* ctx, const uint8_t* points, uint64_t n) {
  if (!ctx || n > ctx->cap || (n && !points)) return MSM377_EINVAL;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = h2d_staged(ctx, ctx->d_raw_points, points, n * 96, 0);
  if (rc) return rc;
  return g1_set_bases_device(ctx, ctx->d_raw_points, n);
}

// Precomputed-window tables (BASELINE.json config 5 "precomputed-point reuse"; the reference lists precomputation as
// future work, README.md:558-563): T[w][i] = [2^(c w)] P_i as affine Edwards records.  Window 0 is the batched
// conversion of the input; every further window doubles the previous one c times (unified law) and runs through the
// same batched inversion (k_affine_up<AffDoublingSource> -> host -> k_affine_down).
//   c = 16 (round 2): 16 windows, the main path's geometry; the 16 bucket sets are folded after the accumulation.
//   c = 20 (round 3, msm377_ctx_set_precompute_window): 13 windows over ONE set of 2^19 buckets -- 13 n bucket
//          additions per MSM instead of 16 n (kernels/wide.hpp).
int g1_set_bases_precomputed_device(msm377_ctx* ctx, const void* d_points, uint64_t n) {
  int rc = check_args(ctx, d_points, d_points, n, true);
  if (rc) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->g1_form != 1 || n == 0) return g1_set_bases_device(ctx, d_points, n);  // Weierstrass form: no precomputation
  const bool wide = ctx->precomp_bits == (int)WIDE_BITS;
  const uint32_t windows = wide ? WIDE_WINDOWS : (uint32_t)MSM377_NUM_WINDOWS;
  if (ctx->table_cap < n || ctx->table_windows != windows) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // nothing in flight reads the old table
    free_table(ctx);
    bool ok = hipMalloc((void**)&ctx->d_table, (size_t)windows * n * TeAffBase::REC_WORDS * 4) == hipSuccess;
    if (ok && wide)
      ok = hipMalloc((void**)&ctx->d_wide_digits, (size_t)WIDE_WINDOWS * n * 4) == hipSuccess &&
           hipMalloc((void**)&ctx->d_wide_temp, (size_t)WIDE_WINDOWS * n * sizeof(SortElem)) == hipSuccess &&
           hipMalloc((void**)&ctx->d_wide_counts, WC_WORDS * 4) == hipSuccess;
    if (!ok) {
      free_table(ctx);
      (void)hipGetLastError();
      ctx->err = "precomputed-window table: out of device memory";
      return MSM377_ENOMEM;
    }
    ctx->table_cap = n;
    ctx->table_windows = windows;
  }
  if (wide && (uint64_t)WIDE_WINDOWS * n >= (1ull << 31)) {
    ctx->err = "precomputed-window table: too many points for 20-bit windows (13 n must stay below 2^31)";
    return MSM377_EINVAL;
  }
  ctx->table_window_bits = wide ? WIDE_BITS : (uint32_t)MSM377_WINDOW_BITS;
  for (uint32_t w = 0; w < windows && rc == MSM377_OK; w++) {
    ctx->table_doublings = !wide ? (uint32_t)MSM377_WINDOW_BITS : w ? wide_width(w - 1) : 0u;  // from the previous window's multiple to this one's
    uint32_t* mine = ctx->d_table + (size_t)w * n * TeAffBase::REC_WORDS;
    rc = affine_convert_begin(ctx, (const uint32_t*)d_points, n, w == 0 ? nullptr : mine - (size_t)n * TeAffBase::REC_WORDS, w == 0);
    if (rc == MSM377_OK) rc = affine_convert_finish(ctx, mine, n);
  }
  if (rc) return rc;
  if (d_points != ctx->d_raw_points) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_raw_points, d_points, n * 96, hipMemcpyDeviceToDevice, ctx->stream2));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2));
  ctx->bases_n = n;
  ctx->bases_form = TABLE_TE_PRECOMP;
  return MSM377_OK;
}

int g1_set_bases_precomputed(msm377_ctx* ctx, const uint8_t* points, uint64_t n) {
  if (!ctx || n > ctx->cap || (n && !points)) return MSM377_EINVAL;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = h2d_staged(ctx, ctx->d_raw_points, points, n * 96, 0);
  if (rc) return rc;
  return g1_set_bases_precomputed_device(ctx, ctx->d_raw_points, n);
}

int g1_msm_fixed_base_device(msm377_ctx* ctx, const void* d_scalars, uint64_t n, uint8_t out_xy[96]) {
  if (!out_xy) return MSM377_EINVAL;
  int rc = check_args(ctx, nullptr, d_scalars, n, false);
  if (rc) return rc;
  if (n > ctx->bases_n) {
    ctx->err = "fixed-base MSM needs g1_set_bases with at least n points first";
    return MSM377_ESTATE;
  }
  if (n == 0) {
    identity_wire(out_xy);
    return MSM377_OK;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->timing == 1) {  // no conversion in this mode
    (void)hipEventRecord(ctx->ev[0][MSM377_STAGE_CONVERT][0], ctx->stream);
    (void)hipEventRecord(ctx->ev[0][MSM377_STAGE_CONVERT][1], ctx->stream);
  }
  rc = g1_table_msm(ctx, (const uint32_t*)d_scalars, n, resident_form(ctx, n), out_xy);
  if (rc != RC_TE_FALLBACK) return rc;
  rc = resident_table_to_weierstrass(ctx);
  if (rc) return rc;
  return g1_table_msm(ctx, (const uint32_t*)d_scalars, n, resident_form(ctx, n), out_xy);
}

// `batch` MSMs of n scalars each against the table resident in (or lent to, twin_borrow) `ctx`, on ctx's own stream and
// buffers.  RC_TE_FALLBACK: an exceptional case of the Edwards law; nothing of out_xy is valid then.
static int fixed_base_batch_share(msm377_ctx* ctx, const void* d_scalars, uint64_t n, uint32_t batch, uint8_t* out_xy) {
  int rc = MSM377_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint32_t* sc = (const uint32_t*)d_scalars;
  const int form = resident_form(ctx, n);
  const bool glv = form == TABLE_XYZZ_GLV, te = form_is_te(form);
  const bool wide = form == TABLE_TE_PRECOMP && ctx->table_window_bits == WIDE_BITS;
  const uint32_t W = wide ? 1u : glv ? GLV_WINDOWS : (uint32_t)MSM377_NUM_WINDOWS;  // window slots of a call
  Phase table_phase;
  if (form == TABLE_TE_PRECOMP) {
    table_phase.table = ctx->d_table;
    table_phase.table_stride = ctx->bases_n;
  }
  if (wide) {
    table_phase.wide = true;
    table_phase.cbits = WIDE_BITS;
    table_phase.bucket_log = WIDE_LOG;
  }
  const bool even = te && form != TABLE_TE_PRECOMP && ctx->even_windows;  // (the Weierstrass forms keep sixteen equal windows)
  table_phase.even = even;
  const int tail_cbits = wide ? (int)WIDE_BITS : 16, tail_planes = wide ? (int)WIDE_LOG : 15, tail_short = even ? (int)EVEN_FROM : 0;
  const int W_tail = form == TABLE_TE_PRECOMP ? 1 : (int)W;  // window records the host combines per MSM
  std::vector<uint32_t> redo;  // elements whose scalars fall outside the GLV range: rerun plain afterwards
  std::vector<uint32_t> redo_wide;  // elements with a scalar of 2^253 and more on the wide table or the even windows: rerun one by one (g1_table_msm falls back)
  bool te_fallback = false;
  // Software pipeline over the batch: while the GPU runs MSM b, the host finishes MSM b-1
  // (Horner + inversion on the other slot's partial records).
  for (uint32_t b = 0; b <= batch; b++) {
    if (b < batch && !te_fallback) {
      rc = (form == TABLE_TE_AFFINE || form == TABLE_TE_PRECOMP) ? enqueue_windows<TeDev, TeAffBase>(ctx, sc + (size_t)b * n * 8, n, 0, W, (int)(b & 1), false, table_phase)
           : te                    ? enqueue_windows<TeDev>(ctx, sc + (size_t)b * n * 8, n, 0, W, (int)(b & 1), false, table_phase)
                                   : enqueue_windows<G1Dev>(ctx, sc + (size_t)b * n * 8, n, 0, W, (int)(b & 1), glv);
      if (rc) return rc;
    }
    if (b > 0) {
      const int slot = (int)((b - 1) & 1);
      HIP_TRY(ctx, hipEventSynchronize(ctx->done_ev[slot]));
      if (te && !te_fallback && (ctx->h_err[slot] & ERR_TE_ANY)) {
        te_fallback = true;
        note_fallback(ctx, (uint32_t)(ctx->h_err[slot] & ERR_TE_ANY));
      }
      if (te_fallback) continue;
      if (glv && (ctx->h_err[slot] & ERR_GLV_RANGE)) {
        redo.push_back(b - 1);
        continue;
      }
      if ((wide || even) && (ctx->h_err[slot] & ERR_NARROW_RANGE) && !(ctx->h_err[slot] & ERR_SCALAR)) {
        redo_wide.push_back(b - 1);
        continue;
      }
      rc = finish_windows(ctx, slot);
      if (rc) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
      }
      if (te) {
        if (teh_combine(ctx->h_partials + (size_t)slot * SLOT_WORDS, W_tail, out_xy + (size_t)96 * (b - 1), tail_cbits, tail_planes, tail_short)) {
          te_fallback = true;
          note_fallback(ctx, MSM377_FB_TAIL);
        }
      } else
        g1h_combine(ctx->h_partials + (size_t)slot * SLOT_WORDS, W, out_xy + (size_t)96 * (b - 1));
    }
  }
  if (te_fallback) {  // an exceptional case of the Edwards law somewhere in the batch: the caller rebuilds the table and reruns
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RC_TE_FALLBACK;
  }
  for (uint32_t b : redo) {
    rc = g1_table_msm(ctx, sc + (size_t)b * n * 8, n, TABLE_XYZZ, out_xy + (size_t)96 * b);
    if (rc) return rc;
  }
  for (uint32_t b : redo_wide) {
    rc = g1_table_msm(ctx, sc + (size_t)b * n * 8, n, form, out_xy + (size_t)96 * b);
    if (rc) return rc;  // RC_TE_FALLBACK included
  }
  return MSM377_OK;
}

static void twin_return(msm377_ctx* ctx) {
  msm377_ctx* tw = ctx->twin;
  if (!tw) return;
  tw->d_bases = nullptr;
  tw->d_table = nullptr;
  tw->bases_n = 0;
}

// The twin of a context: a second context on the same device -- own streams, work buffers, pinned records -- that BORROWS
// the resident table for the length of one batch call.  Created with the first batch that is large enough; its cost
// (the work buffers a second time, ~0.7 GB at 2^20 points) is why small batches do not ask for it.
constexpr uint32_t TWIN_MIN_BATCH = 4;

static int twin_prepare(msm377_ctx* ctx) {
  if (!ctx->twin) {
    if (ctx->twin_failed) return MSM377_ENOMEM;
    msm377_ctx* tw = nullptr;
    if (msm377_ctx_create(ctx->device, ctx->cap, &tw) != MSM377_OK) {
      ctx->twin_failed = true;  // out of memory for a second set: batches run on one
      (void)hipGetLastError();
      return MSM377_ENOMEM;
    }
    (void)hipFree(tw->d_bases);  // it only ever borrows
    tw->d_bases = nullptr;
    (void)hipFree(tw->d_raw_points);
    tw->d_raw_points = nullptr;
    tw->twin_batches = false;
    ctx->twin = tw;
  }
  msm377_ctx* tw = ctx->twin;
  const bool wide = ctx->bases_form == TABLE_TE_PRECOMP && ctx->table_window_bits == WIDE_BITS;
  if (wide && tw->wide_cap < ctx->bases_n) {
    for (void* p : {(void*)tw->d_wide_digits, (void*)tw->d_wide_counts, (void*)tw->d_wide_temp})
      if (p) (void)hipFree(p);
    tw->d_wide_digits = nullptr, tw->d_wide_counts = nullptr, tw->d_wide_temp = nullptr, tw->wide_cap = 0;
    const uint64_t n = ctx->bases_n;
    if (hipMalloc((void**)&tw->d_wide_digits, (size_t)WIDE_WINDOWS * n * 4) != hipSuccess ||
        hipMalloc((void**)&tw->d_wide_temp, (size_t)WIDE_WINDOWS * n * sizeof(SortElem)) != hipSuccess ||
        hipMalloc((void**)&tw->d_wide_counts, WC_WORDS * 4) != hipSuccess) {
      (void)hipGetLastError();
      return MSM377_ENOMEM;
    }
    tw->wide_cap = n;
  }
  // lend the table, and the conversion's verdict that travels with it (d_err[2], read by the accumulation kernels)
  tw->d_bases = ctx->d_bases;
  tw->d_table = ctx->d_table;
  tw->bases_n = ctx->bases_n;
  tw->bases_form = ctx->bases_form;
  tw->table_window_bits = ctx->table_window_bits;
  tw->table_windows = ctx->table_windows;
  tw->seg_plain = ctx->seg_plain, tw->seg_glv = ctx->seg_glv, tw->tail_from = ctx->tail_from;
  if (hipMemcpyAsync(tw->d_err + 2, ctx->d_err + 2, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {
    twin_return(ctx);
    return MSM377_EHIP;
  }
  return MSM377_OK;
}

int g1_msm_fixed_base_batch_device(msm377_ctx* ctx, const void* d_scalars, uint64_t n, uint32_t batch, uint8_t* out_xy) {
  if (!out_xy) return MSM377_EINVAL;
  int rc = check_args(ctx, nullptr, d_scalars, n, false);
  if (rc) return rc;
  if (n > ctx->bases_n) {
    ctx->err = "fixed-base MSM needs g1_set_bases with at least n points first";
    return MSM377_ESTATE;
  }
  if (n == 0) {
    for (uint32_t b = 0; b < batch; b++) identity_wire(out_xy + (size_t)96 * b);
    return MSM377_OK;
  }
  // Batches run as two halves on two sets of streams and buffers (twin_prepare): the low-occupancy ends of one MSM --
  // the last tree levels and the tail of its reduction, the drain of its accumulation kernel, its memory-bound sort --
  // fill with the other half's kernels.  Two contexts side by side measured 2.01 -> 1.89 ms per MSM on the 20-bit table
  // and 2.19 -> 2.09 on the plain one (tools/twin_probe.py, profiles/r03_final/twin_probe.txt).
  const bool split = batch >= TWIN_MIN_BATCH && ctx->twin_batches && !ctx->timing && !ctx->capture && twin_prepare(ctx) == MSM377_OK;
  if (!split) {
    rc = fixed_base_batch_share(ctx, d_scalars, n, batch, out_xy);
  } else {
    msm377_ctx* tw = ctx->twin;
    const uint32_t mine = batch - batch / 2;
    int rc2 = MSM377_OK;
    std::thread other([&] { rc2 = fixed_base_batch_share(tw, (const uint32_t*)d_scalars + (size_t)mine * n * 8, n, batch - mine, out_xy + (size_t)96 * mine); });
    rc = fixed_base_batch_share(ctx, d_scalars, n, mine, out_xy);
    other.join();
    twin_return(ctx);
    if (rc2 && !rc) {  // the first half's error wins; RC_TE_FALLBACK of either half reruns the whole batch
      rc = rc2;
      if (rc2 != RC_TE_FALLBACK) ctx->err = tw->err;
    }
    if (tw->fallback_count) {
      ctx->fallback_count += tw->fallback_count;
      ctx->fallback_mask = tw->fallback_mask;
      tw->fallback_count = 0;
    }
  }
  if (rc != RC_TE_FALLBACK) return rc;
  rc = resident_table_to_weierstrass(ctx);  // whole batch again on the Weierstrass table
  if (rc) return rc;
  return g1_msm_fixed_base_batch_device(ctx, d_scalars, n, batch, out_xy);
}

int g1_msm_fixed_base(msm377_ctx* ctx, const uint8_t* scalars, uint64_t n, uint8_t out_xy[96]) {
  if (!ctx || !out_xy || n > ctx->cap || (n && !scalars)) return MSM377_EINVAL;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = h2d_staged(ctx, ctx->d_raw_scalars, scalars, n * 32, (size_t)ctx->cap * 96);
  if (rc) return rc;
  return g1_msm_fixed_base_device(ctx, ctx->d_raw_scalars, n, out_xy);
}

// Windows [win_begin, win_begin + win_count) of a G1 MSM; the records go to a host buffer, a device buffer, or both.
int window_partials(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint32_t win_begin, uint32_t win_count,
                           uint8_t* host_out, void* dev_out) {
  int rc = check_args(ctx, d_points, d_scalars, n, true);
  if (rc) return rc;
  if (win_count == 0 || win_begin >= MSM377_NUM_WINDOWS || win_count > MSM377_NUM_WINDOWS - win_begin) {
    ctx->err = "window range outside 0..16";
    return MSM377_EINVAL;
  }
  if ((uintptr_t)dev_out & 15) {
    ctx->err = "device output pointer must be 16-byte aligned";
    return MSM377_EINVAL;
  }
  const size_t bytes = (size_t)win_count * MSM377_G1_WINDOW_PARTIAL_BYTES;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (n == 0) {  // identity partials: ZZ = 0 everywhere
    if (host_out) memset(host_out, 0, bytes);
    if (dev_out) HIP_TRY(ctx, hipMemset(dev_out, 0, bytes));
    return MSM377_OK;
  }
  ctx->bases_n = 0;
  // The records are complete in ctx->d_partials (slot 0) once the call's completion event has fired; the copy
  // to the caller's device buffer rides the same stream and the call returns with that stream idle, so a
  // collective on any other stream may read the buffer.
  auto deliver = [&]() -> int {
    if (host_out) memcpy(host_out, ctx->h_partials, bytes);
    if (dev_out) {
      HIP_TRY(ctx, hipMemcpyAsync(dev_out, ctx->d_partials, bytes, hipMemcpyDeviceToDevice, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MSM377_OK;
  };
  if (ctx->g1_form == 1) {  // twisted Edwards form; k_gather_partials tags the records (fp64_host.hpp TE_RECORD_TAG)
    // Affine base records (7-product additions, batched inversion) once a point takes part in enough additions to pay for
    // its ~9 extra conversion products: windows x points >= 2^24 -- e.g. the 8 windows a rank of a 2-GPU run owns at 2^21
    // points and more, the 2 of an 8-GPU run at 2^23 (msm377_g1_msm_device: 16 windows, n >= 2^20).
    const bool affine = ctx->te_affine_msm && n >= (1ull << 18) && (uint64_t)win_count * n >= (1ull << 24);
    if (affine) {
      rc = affine_convert_begin(ctx, (const uint32_t*)d_points, n);
      if (rc) return rc;
      ctx->before_accumulate = [ctx, n]() -> int { return affine_convert_finish(ctx, ctx->d_bases, n, true); };
      rc = enqueue_windows<TeDev, TeAffBase>(ctx, (const uint32_t*)d_scalars, n, win_begin, win_count, 0);
      ctx->before_accumulate = nullptr;
    } else {
      rc = convert_bases<TeDev>(ctx, (const uint32_t*)d_points, n);
      if (rc) return rc;
      rc = enqueue_windows<TeDev>(ctx, (const uint32_t*)d_scalars, n, win_begin, win_count, 0);
    }
    if (rc) return rc;
    HIP_TRY(ctx, hipEventSynchronize(ctx->done_ev[0]));
    if ((ctx->h_err[0] & ERR_TE_ANY) == 0) {
      rc = finish_windows(ctx, 0);
      if (rc) return rc;
      return deliver();
    }
    note_fallback(ctx, (uint32_t)(ctx->h_err[0] & ERR_TE_ANY));
    // an exceptional case of the Edwards law in THESE windows: they alone rerun below, untagged
  }
  rc = convert_bases<G1Dev>(ctx, (const uint32_t*)d_points, n);
  if (rc) return rc;
  rc = run_windows<G1Dev>(ctx, (const uint32_t*)d_scalars, n, win_begin, win_count);
  if (rc) return rc;
  return deliver();
}

int g1_glv_window_partials_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint32_t win_begin,
                                         uint32_t win_count, uint8_t* partials_out) {
  if (!partials_out) return MSM377_EINVAL;
  int rc = check_args(ctx, d_points, d_scalars, n, true);
  if (rc) return rc;
  if (win_count == 0 || win_begin >= GLV_WINDOWS || win_count > GLV_WINDOWS - win_begin) {
    ctx->err = "GLV window range outside 0..8";
    return MSM377_EINVAL;
  }
  if (n == 0) {
    memset(partials_out, 0, (size_t)win_count * MSM377_G1_WINDOW_PARTIAL_BYTES);
    return MSM377_OK;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  ctx->bases_n = 0;
  rc = convert_bases_g1(ctx, (const uint32_t*)d_points, n, true);
  if (rc) return rc;
  rc = enqueue_windows<G1Dev>(ctx, (const uint32_t*)d_scalars, n, win_begin, win_count, 0, true);
  if (rc) return rc;
  HIP_TRY(ctx, hipEventSynchronize(ctx->done_ev[0]));
  if (ctx->h_err[0] & 2) {
    ctx->err = "a scalar is outside the GLV range; use the plain window path";
    return MSM377_EGLVRANGE;
  }
  rc = finish_windows(ctx, 0);
  if (rc) return rc;
  memcpy(partials_out, ctx->h_partials, (size_t)win_count * MSM377_G1_WINDOW_PARTIAL_BYTES);
  return MSM377_OK;
}

int g1_generate_bases_device(msm377_ctx* ctx, uint64_t seed, uint64_t n, void* d_points_out) {
  if (!ctx || (n && !d_points_out) || ((uintptr_t)d_points_out & 15)) return MSM377_EINVAL;
  if (n == 0) return MSM377_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_generate_bases, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, seed, n, (uint32_t*)d_points_out);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MSM377_OK;
}

}  // namespace eng
}  // namespace msm377
