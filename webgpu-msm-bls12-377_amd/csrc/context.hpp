// The engine context behind the opaque msm377_ctx of include/msm377.h: streams, events, device workspace (layout: DESIGN.md
// section 3), pinned host buffers, tuning state.  Shared by sequencer.hip, host_tail.hip and capi.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "common.hpp"
#include "fp64_host.hpp"
#include "tail_pool.hpp"

using msm377::SortElem;
using msm377::WorkItem;
using msm377::TailPool;
using msm377::Fp64;
using msm377::MAX_WINDOW_SLOTS;
using msm377::NARROW_SEG;

// MSM377_UPLOAD_TRACE=1: where the time of a host-buffer call goes (sequencer.hip run_sorted_upload).
struct UploadTrace {
  bool on = false;
  uint32_t k = 0;
  std::chrono::steady_clock::time_point t0;
  std::vector<std::pair<const char*, double>> marks;
  double up_us[9] = {}, enq_us[9] = {};
  hipEvent_t ev[12] = {};
  uint32_t nev = 0;
  double now_us() const { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
  void begin(bool enabled, uint32_t chunks) {
    on = enabled;
    if (!on) return;
    k = chunks;
    nev = 0;
    marks.clear();
    t0 = std::chrono::steady_clock::now();
    for (hipEvent_t& e : ev)
      if (!e) (void)hipEventCreate(&e);
  }
  void host(const char* what) {
    if (on) marks.emplace_back(what, now_us());
  }
  void chunk_up(uint32_t c) {
    if (on) up_us[c] = now_us();
  }
  void chunk_enqueued(uint32_t c) {
    if (on) enq_us[c] = now_us();
  }
  void gpu(hipStream_t st, uint32_t slot) {  // slot 0: before the sort; 1: behind it; 2 + c: behind chunk c's phase
    if (on && slot < 12) {
      (void)hipEventRecord(ev[slot], st);
      if (slot + 1 > nev) nev = slot + 1;
    }
  }
  void report() {  // after the call's completion event
    if (!on) return;
    fprintf(stderr, "upload trace (us after the call; GPU times relative to the sort's launch):");
    for (auto& m : marks) fprintf(stderr, "  %s %.0f", m.first, m.second);
    for (uint32_t c = 0; c < k; c++) fprintf(stderr, "  chunk %u up %.0f enq %.0f", c, up_us[c], enq_us[c]);
    for (uint32_t s = 1; s < nev; s++) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev[0], ev[s]) == hipSuccess) fprintf(stderr, "  gpu[%s%u] +%.0f", s == 1 ? "sort" : "chunk ", s == 1 ? 0u : s - 2, ms * 1e3);
    }
    fprintf(stderr, "  done %.0f\n", now_us());
    (void)hipGetLastError();
  }
};

struct msm377_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;      // base conversion, overlapped with decompose + sort
  hipEvent_t bases_ready = nullptr;
  uint64_t cap = 0;
  std::string err;
  // device buffers
  uint32_t* d_raw_points = nullptr;   // cap x 24 words (host-buffer API staging)
  uint32_t* d_raw_scalars = nullptr;  // cap x 8 words
  uint32_t* d_bases = nullptr;        // cap x 32 words
  uint16_t* d_digits = nullptr;       // 16 x cap
  uint32_t* d_range_counts = nullptr; // 16 x NRANGE x chunks: per-chunk range counts, then write offsets
  uint32_t* d_region_base = nullptr;  // 16 x (NRANGE + 1)
  SortElem* d_sort_temp = nullptr;    // 16 x cap partitioned (index|sign, key) pairs
  uint32_t* d_row_ptr = nullptr;      // 16 x RP
  uint32_t* d_row_ptr_chunks = nullptr; // 16 x ((NB + 1) x 8 + 1): rows filed by upload chunk (run_sorted_upload; allocated on first use)
  uint32_t* d_val_idx = nullptr;      // 16 x cap
  uint32_t* d_buckets = nullptr;      // 16 x 52 x NB
  uint32_t* d_buckets_snap = nullptr; // stage capture only
  uint32_t* d_partials = nullptr;     // 2 slots x 16 x 16 x 52 (double-buffered for batches)
  WorkItem* d_work = nullptr;         // sorted accumulation work items (<= 16 NB + 16 cap / SEG)
  uint32_t* d_work_meta = nullptr;    // [0..SEG] length histogram, [SEG_BINS..] cursors, then total, split-row count, overflow count
  uint32_t* d_row_ovf_base = nullptr; // 16 x NB
  uint32_t* d_split_rows = nullptr;   // 16 x NB
  uint32_t* d_ovf = nullptr;          // overflow partial points, 52 words each (<= 16 cap / SEG)
  uint32_t* d_table = nullptr;        // precomputed-window table: 16 x table_cap affine records, [2^(16 w)] P_i at record w * bases_n + i
  uint64_t table_cap = 0;
  uint32_t table_windows = 0;         // windows the allocated table holds (16, or WIDE_WINDOWS)
  int precomp_bits = MSM377_WINDOW_BITS;  // window width msm377_g1_set_bases_precomputed builds its next table for: 16 or 20 (msm377_ctx_set_precompute_window, MSM377_PRECOMP_BITS)
  uint32_t* d_wide_digits = nullptr;  // wide windows: 13 x n u32 biased 20-bit digits, the flat list the sort reads
  SortElem* d_wide_temp = nullptr;    // wide windows: output of the second partition pass (the first one writes d_sort_temp)
  uint32_t* d_wide_counts = nullptr;  // wide windows: the sort's counters and offsets (kernels/wide.hpp WC_*)
  uint32_t* d_aff_stash = nullptr;    // cap x 52 words: N1, N2, Z, running product per point (k_affine_up -> k_affine_down)
  uint32_t* d_aff_trees = nullptr;    // one product tree (2 x 256 nodes x 13 words) per AFF_BLOCK_POINTS points
  uint32_t* h_aff_prod = nullptr;     // pinned + coherent host memory the kernels access in place (dm_* = its device address)
  uint32_t* h_aff_inv = nullptr;
  uint32_t* h_aff_flag = nullptr;     // workgroups of k_affine_up that have delivered their product
  uint32_t *dm_aff_prod = nullptr, *dm_aff_inv = nullptr, *dm_aff_flag = nullptr;
  uint32_t* d_aff_count = nullptr;    // workgroups of k_affine_up that have delivered (device memory; the last one resets it)
  hipEvent_t aff_up_done = nullptr;
  uint32_t table_window_bits = MSM377_WINDOW_BITS;  // window width of the resident precomputed table: 16, or WIDE_BITS (six 20-bit + seven 19-bit windows)
  uint32_t table_doublings = MSM377_WINDOW_BITS;    // doublings from the previous window's multiple to the one being built (AffDoublingSource)
  std::vector<Fp64::El> aff_scratch;  // prefix products of the host's share of Montgomery's trick
  bool te_affine_msm = true;          // MSM377_TE_AFFINE_MSM=0: msm377_g1_msm_device keeps projective records (A/B knob)
  // Below this the batched conversion does not pay: it costs ~9 more products per point than the projective record and
  // saves 16, but its two kernels and the host round trip sit in front of the accumulation, which they cannot hide
  // under the (short) sort of a small input.  Interleaved A/B, projective / affine ms per MSM (tools/ab_knobs.py):
  // 2^15 0.69 / 0.88, 2^16 0.75 / 0.89, 2^17 0.88 / 0.97, 2^18 1.19 / 1.26, 2^19 1.76 / 1.78, 2^20 2.89 / 2.79.  MSM377_AFFINE_MIN.
  uint64_t affine_min_points = 1ull << 20;
  int* d_err = nullptr;               // 2 slots
  // pinned host
  uint32_t* h_partials = nullptr;     // 2 slots
  int* h_err = nullptr;               // 2 slots
  hipEvent_t done_ev[2] = {};
  // host-buffer entry points: pinned staging + copy workers (allocated on first use)
  uint8_t* h_stage = nullptr;  // cap x 128 bytes
  hipStream_t copy_stream[8] = {};
  // state
  uint64_t bases_n = 0;  // resident base count (fixed-base mode)
  uint64_t last_n = 0;
  uint32_t last_wc = 0;
  int last_form = -1;  // MSM377_STAGE_FORM_* of the buckets the last call left (stage read-backs)
  bool capture = false;
  // Zero-copy output of the full-MSM path (k_gather_partials, wait_zero_copy_out); MSM377_ZERO_COPY_OUT=0: D2H copies + event.
  int zc_out = 1;
  bool zc_active = false;         // the call being enqueued / waited for uses it
  uint32_t out_seq = 0;           // sequence number of the last zero-copy call
  uint32_t* h_out_flag = nullptr; // pinned: [0] sequence number, [1] error word
  uint32_t* dm_out_flag = nullptr;
  uint32_t* dm_partials = nullptr;  // device address of h_partials
  uint32_t* d_out_count = nullptr;
  int timing = 0;  // msm377_ctx_set_timing: 0 off, 1 every stage, 2 the accumulation kernel only
  // First reduction level run with one addition per lane quad: the first level whose 4 lanes x additions x windows fit
  // coop_threads -- level 7 for 16 windows (measured: 18-24 -> 13-18 us per level from there on, slower before), 6 for 8, 4
  // for the 2 windows a rank of an 8-GPU window-sharded run owns.
  // MSM377_COOP_THREADS: a tree level runs one lane quad per addition once that takes at most this many threads.  65536 / 131072 /
  // 262144 make no difference on the main path (2^20: 2.74 ms each); on the narrow path 131072 moves its levels 0-2 to quads.
  uint32_t coop_threads = 131072;
  // MSM377_NARROW_TAIL_FROM: tail_from of the narrow-window path (2048 buckets per window).  Reduce stage at 2^12 with
  // 7 / 5 / 4 / 3 / 2: 0.106 / 0.099 / 0.096 / 0.101 / 0.122 ms (profiles/r02_final/ab_narrow_tree.txt).
  uint32_t narrow_tail_from = 4;
  uint32_t narrow_seg = 0;  // MSM377_NARROW_SEG (>= NARROW_SEG: the buffers are sized for that); 0 = by input size (enqueue_part)
  uint64_t narrow_quad_items = 100000;  // MSM377_NARROW_QUAD_ITEMS: most work items k_accumulate_quad is used for
  int narrow_quad_acc = 1;         // MSM377_NARROW_QUAD_ACC=0: the narrow-window path accumulates with a thread per work item, like the main path
  // First level of the single-launch tail of the reduction (k_reduce_tail); MSM377_TAIL_FROM, 15 = one launch per level throughout.
  uint32_t tail_from = 7;  // measured (tools/ab_knobs.py, 2^20): 15: 2.874 ms, 7: 2.842, 6: 2.885, 5: 2.916, 4: 3.062
  // GLV front end of the Weierstrass path: 0 = off (default), 1 = on.  phi(P) = [lambda] P holds only for points of
  // the prime-order subgroup, so it is an opt-in: the caller vouches for the inputs (every protocol use does).
  // Interleaved A/B on one MI355X (tools/ab_knobs.py), Weierstrass plain vs GLV ms per MSM: 2^18 1.39 / 1.24,
  // 2^19 2.08 / 2.00, 2^20 3.56 / 3.51, 2^22 12.56 / 12.39 (halved bucket reduction and host tail).
  int glv_mode = 0;
  int bases_form = 0;      // TableForm of the resident base table (fixed-base mode)
  int g1_form = 1;         // G1 full-MSM entry points: 1 = twisted Edwards form (te377.hpp, default), 0 = Weierstrass XYZZ (MSM377_G1_FORM)
  bool last_glv = false;
  uint32_t seg_plain = 0, seg_glv = 0;  // MSM377_SEG_PLAIN / MSM377_SEG_GLV: force the work-item length (SEG_MIN..SEG_MAX), 0 = auto_seg()
  hipEvent_t ev[2][MSM377_NUM_STAGES][2] = {};  // [part][stage][begin, end]
  uint64_t upload_chunk_min = 1ull << 18;  // msm377_g1_msm: inputs of at least this many points upload and run as two chunks (MSM377_UPLOAD_CHUNK_MIN)
  UploadTrace upload_trace;
  bool upload_trace_on = false;           // MSM377_UPLOAD_TRACE=1
  uint32_t upload_chunks = 4;              // chunks of the host-buffer upload (MSM377_UPLOAD_CHUNKS, 2..7): 2: 5.07, 3: 4.89, 4-6: 4.70, 8: 4.95 ms at 2^20 (round 2)
  uint32_t upload_split_pct = 30;          // share of the points in the first chunk (MSM377_UPLOAD_SPLIT, 5..90)
  bool upload_sort_once = false;           // MSM377_UPLOAD_SORT_ONCE=1: scalars first, one sort, chunks of points accumulate through their sub-rows (run_sorted_upload)
  std::function<int()> before_accumulate;  // host-buffer entry point: joins the point upload and launches the base conversion (enqueue_part)
  TailPool tail_pool;
  int tail_threads = 6;               // MSM377_TAIL_THREADS: threads of the host tail (1..8, tail_horner_mt)
  // MSM377_TAIL_SPIN_US: how long at most the tail workers poll for their job after a call has armed them (TailPool;
  // 0 = they sleep until the job is posted).  Tail stage at 2^20, interleaved (tools/ab_knobs.py): one thread 0.140 ms,
  // six sleeping workers 0.124, six polling ones 0.089.  (Round 2 first measured no difference: the per-thread
  // exceptional-case flags shared a cache line then and the threads fought over it -- TeChecked is padded now.)
  // MSM377_AFF_PREWAKE_US: the helper threads that invert the conversion's block products are woken when the way up is
  // queued and poll for their share (at most this long) instead of being woken from their condition variable when the
  // products arrive -- the wake-up (20-60 us) sat in the middle of the front end's critical path.  0 = off.
  int64_t aff_prewake_us = 600;
  // MSM377_FRONT_WAVE_PRIO: the memory-bound front-end kernels (decompose, sort, work list) raise their waves' issue priority
  // (s_setprio 3) over the VALU-bound base conversion that runs beside them on the side stream.
  uint32_t front_wave_prio = 0;
  uint32_t conv_wave_prio = 1;  // MSM377_CONV_WAVE_PRIO: the same for the conversion kernels (k_affine_up / k_affine_down)
  int64_t tail_spin_us = 1000;
  bool tail_trace = false;  // MSM377_TAIL_TRACE=1
  double stage_ms[MSM377_NUM_STAGES] = {};
  int last_products = 0;        // field products per bucket addition of the last accumulation launch (bench.py's int32-mad roof)
  // Inputs of at most this many points run the narrow-window path (11-bit windows: 23 x 2048 buckets instead of
  // 16 x 32768; MSM377_NARROW_MAX, 0 = never).  Interleaved A/B, 16-bit / narrow ms per MSM (tools/ab_knobs.py):
  // 2^10 0.64 / 0.46, 2^13 0.67 / 0.53, 2^14 0.68 / 0.51, 2^15 0.73 / 0.60, 2^16 0.74 / 0.71 (first version); at the end of
  // round 2: 2^16 0.605 / 0.56 (its bucket reduction 0.28 / 0.10 ms, its accumulation kernel 0.14 / 0.18), hence 2^16.
  uint64_t narrow_max_points = 1ull << 16;
  // Batches on two sets of streams and buffers (sequencer.hip twin_prepare)
  bool tail_lds = true;         // MSM377_TAIL_LDS=0: the single-launch reduction tail works in global memory (k_reduce_tail)
  bool narrow_even = true;      // MSM377_NARROW_EVEN=0: the small-input path recodes into 22 signed 11-bit windows + an unsigned top one (k_decompose_narrow)
  bool even_windows = true;     // MSM377_EVEN_WINDOWS=0: sixteen 16-bit windows on every path (kernels/decompose.hpp k_decompose)
  bool ed_equal_windows_once = false;  // ed_msm -> ed_msm_device: this call reruns a chunked upload whose scalars did not fit
  uint32_t acc_seq = 0;  // calls' accumulation kernels so far; h_out_flag[ACC_FLAG_WORD] follows it (k_merge_split_rows_quad)
  msm377_ctx* twin = nullptr;   // owned; borrows d_bases / d_table for the length of a batch call
  bool twin_batches = true;     // MSM377_TWIN_BATCH=0: batches run on this context alone
  bool twin_failed = false;
  uint64_t wide_cap = 0;        // points the twin's wide-window buffers hold
  uint64_t fallback_count = 0;  // reruns on the Weierstrass path after an exceptional case of the Edwards law
  uint32_t fallback_mask = 0;   // MSM377_FB_* bits of the last one
};
