// The C ABI of include/msm377.h: context life cycle, settings, the host-only combine functions, the stage read-back,
// and one forwarding line per entry point that enqueues GPU work (those live in sequencer.hip under the same name
// without the msm377_ prefix).  No kernels in this translation unit.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "context.hpp"
#include "host_tail.hpp"
#include "sequencer.hpp"

using namespace msm377;

#define HIP_TRY(ctx, call)                                                  \
  do {                                                                      \
    if (!eng::hip_ok((ctx), (int)(call), #call)) return MSM377_EHIP;        \
  } while (0)

// --------------------------------------------------------------------------- C ABI ----

extern "C" {

const char* msm377_version(void) { return "msm377 0.1 gfx950"; }

const char* msm377_strerror(int code) {
  switch (code) {
    case MSM377_OK: return "ok";
    case MSM377_EINVAL: return "invalid argument";
    case MSM377_EHIP: return "HIP runtime error";
    case MSM377_ESCALAR: return "scalar out of range for the signed window recode";
    case MSM377_ENOMEM: return "out of memory";
    case MSM377_ESTATE: return "call sequence error";
    case MSM377_EGLVRANGE: return "scalar outside the GLV range";
    case MSM377_EEXCEPTIONAL: return "exceptional case of the twisted Edwards law while combining partial records";
    default: return "unknown error";
  }
}

int msm377_ctx_create(int device, uint64_t max_points, msm377_ctx** out) {
  if (!out || max_points == 0 || max_points > (1ull << 30)) return MSM377_EINVAL;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return MSM377_EHIP;
  msm377_ctx* ctx = new (std::nothrow) msm377_ctx();
  if (!ctx) return MSM377_ENOMEM;
  ctx->device = device;
  ctx->cap = max_points;
  if (const char* e = getenv("MSM377_GLV")) ctx->glv_mode = atoi(e);
  if (const char* e = getenv("MSM377_G1_FORM")) ctx->g1_form = atoi(e) != 0;
  if (const char* e = getenv("MSM377_UPLOAD_SORT_ONCE")) ctx->upload_sort_once = atoi(e) != 0;
  if (const char* e = getenv("MSM377_CONV_WAVE_PRIO")) ctx->conv_wave_prio = atoi(e) != 0;
  if (const char* e = getenv("MSM377_FRONT_WAVE_PRIO")) ctx->front_wave_prio = atoi(e) != 0;
  if (const char* e = getenv("MSM377_AFF_PREWAKE_US")) ctx->aff_prewake_us = atoll(e);
  if (const char* e = getenv("MSM377_UPLOAD_TRACE")) ctx->upload_trace_on = atoi(e) != 0;
  if (const char* e = getenv("MSM377_UPLOAD_CHUNKS")) ctx->upload_chunks = (uint32_t)std::min(std::max(atoi(e), 2), 7);
  if (const char* e = getenv("MSM377_UPLOAD_SPLIT")) ctx->upload_split_pct = (uint32_t)std::min(std::max(atoi(e), 5), 90);
  if (const char* e = getenv("MSM377_UPLOAD_CHUNK_MIN")) ctx->upload_chunk_min = strtoull(e, nullptr, 10);
  if (const char* e = getenv("MSM377_TAIL_THREADS")) ctx->tail_threads = std::min(std::max(atoi(e), 1), TailPool::WORKERS + 1);
  if (const char* e = getenv("MSM377_TAIL_WAIT_MS")) ctx->tail_pool.wait_limit_ns = (int64_t)std::max(atoi(e), 1) * 1000000ll;
  if (const char* e = getenv("MSM377_TAIL_SPIN_US")) ctx->tail_spin_us = atoll(e);
  if (const char* e = getenv("MSM377_TAIL_TRACE")) ctx->tail_trace = atoi(e) != 0;
  if (const char* e = getenv("MSM377_TAIL_NUMA")) ctx->tail_pool.numa_local = atoi(e) != 0;
  if (const char* e = getenv("MSM377_TE_AFFINE_MSM")) ctx->te_affine_msm = atoi(e) != 0;
  if (const char* e = getenv("MSM377_NARROW_MAX")) ctx->narrow_max_points = strtoull(e, nullptr, 10);
  if (const char* e = getenv("MSM377_PRECOMP_BITS")) ctx->precomp_bits = atoi(e) == (int)WIDE_BITS ? (int)WIDE_BITS : MSM377_WINDOW_BITS;
  if (const char* e = getenv("MSM377_AFFINE_MIN")) ctx->affine_min_points = strtoull(e, nullptr, 10);
  if (const char* e = getenv("MSM377_SEG_PLAIN")) ctx->seg_plain = std::min(std::max(atoi(e), (int)SEG_MIN), (int)SEG_MAX);
  if (const char* e = getenv("MSM377_SEG_GLV")) ctx->seg_glv = std::min(std::max(atoi(e), (int)SEG_MIN), (int)SEG_MAX);
  if (const char* e = getenv("MSM377_ZERO_COPY_OUT")) ctx->zc_out = atoi(e);
  if (const char* e = getenv("MSM377_NARROW_SEG")) ctx->narrow_seg = (uint32_t)std::min(std::max(atoi(e), (int)NARROW_SEG), (int)SEG_BINS - 1);
  if (const char* e = getenv("MSM377_NARROW_QUAD_ITEMS")) ctx->narrow_quad_items = strtoull(e, nullptr, 10);
  if (const char* e = getenv("MSM377_NARROW_QUAD_ACC")) ctx->narrow_quad_acc = atoi(e);
  if (const char* e = getenv("MSM377_COOP_THREADS")) ctx->coop_threads = (uint32_t)atoi(e);
  if (const char* e = getenv("MSM377_NARROW_TAIL_FROM")) ctx->narrow_tail_from = (uint32_t)std::min(std::max(atoi(e), 1), (int)TREE_LEVELS);
  if (const char* e = getenv("MSM377_TAIL_LDS")) ctx->tail_lds = atoi(e) != 0;
  if (const char* e = getenv("MSM377_NARROW_EVEN")) ctx->narrow_even = atoi(e) != 0;
  if (const char* e = getenv("MSM377_EVEN_WINDOWS")) ctx->even_windows = atoi(e) != 0;
  if (const char* e = getenv("MSM377_TWIN_BATCH")) ctx->twin_batches = atoi(e) != 0;
  if (const char* e = getenv("MSM377_TAIL_FROM")) ctx->tail_from = (uint32_t)std::min(std::max(atoi(e), 1), (int)TREE_LEVELS);
  const uint64_t cap = max_points;
  // The main stream outranks the side stream: the base conversion (VALU-heavy, ~0.2 ms) only has to finish before
  // the accumulation starts, decompose + sort on the main stream are the critical path (k_decompose: 16 us alone,
  // ~100 us when it competes with the conversion at equal priority).
  int prio_least = 0, prio_greatest = 0;
  bool ok = hipSetDevice(device) == hipSuccess;
  if (ok && hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) prio_least = prio_greatest = 0;
  // (Round 3 re-measured the priorities with the faster sort, now that the conversion chain is the front end's critical
  // path: main stream high 2.554 ms, side stream high 2.583, equal 2.603 at 2^20 -- profiles/r03_final/ab_prio_prewake.txt.)
  ok = ok && hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_greatest) == hipSuccess &&
            hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, prio_least) == hipSuccess &&
            hipEventCreateWithFlags(&ctx->bases_ready, hipEventDisableTiming) == hipSuccess;
  auto dalloc = [&](void** p, size_t bytes) { ok = ok && hipMalloc(p, bytes) == hipSuccess; };
  dalloc((void**)&ctx->d_raw_points, cap * 96);
  dalloc((void**)&ctx->d_raw_scalars, cap * 32);
  dalloc((void**)&ctx->d_bases, 2 * cap * G1_REC_WORDS * 4);  // 128-byte records of P_i and phi(P_i) (GLV front end), or 256-byte twisted Edwards records of P_i
  // (window, point) entries the window-indexed buffers hold: 16 windows of `cap` points, or the 23 windows of the
  // narrow path over a small input when that is more (small contexts)
  const uint64_t wcap = std::max<uint64_t>((uint64_t)MSM377_NUM_WINDOWS * cap, (uint64_t)NARROW_WINDOWS * std::min<uint64_t>(cap, SMALL_SORT_MAX));
  dalloc((void**)&ctx->d_digits, wcap * 2);
  dalloc((void**)&ctx->d_range_counts, (size_t)NRANGE * MAX_SORT_BLOCKS * 4);  // chunks * wc <= MAX_SORT_BLOCKS
  dalloc((void**)&ctx->d_region_base, (size_t)MSM377_NUM_WINDOWS * (NRANGE + 1) * 4);
  dalloc((void**)&ctx->d_sort_temp, cap * MSM377_NUM_WINDOWS * sizeof(SortElem));
  dalloc((void**)&ctx->d_row_ptr, (size_t)MSM377_NUM_WINDOWS * RP * 4);
  dalloc((void**)&ctx->d_val_idx, wcap * 4);
  dalloc((void**)&ctx->d_buckets, (size_t)MSM377_NUM_WINDOWS * BKT_WORDS * NB * 4);
  dalloc((void**)&ctx->d_partials, (size_t)2 * SLOT_WORDS * 4);
  // extra work items / overflow slots beyond one per row: entries / SEG_MIN on the main path, entries / NARROW_SEG on the narrow one
  const uint64_t extra_items = std::max<uint64_t>((uint64_t)MSM377_NUM_WINDOWS * cap / SEG_MIN, (uint64_t)NARROW_WINDOWS * std::min<uint64_t>(cap, SMALL_SORT_MAX) / NARROW_SEG) + 2;
  dalloc((void**)&ctx->d_work, ((size_t)MSM377_NUM_WINDOWS * NB + extra_items) * sizeof(WorkItem));
  dalloc((void**)&ctx->d_work_meta, (size_t)2 * META_BLOCK_WORDS * 4);  // one block per pipeline part
  dalloc((void**)&ctx->d_row_ovf_base, (size_t)MSM377_NUM_WINDOWS * NB * 4);
  dalloc((void**)&ctx->d_split_rows, (size_t)MSM377_NUM_WINDOWS * NB * 4);
  dalloc((void**)&ctx->d_ovf, (size_t)extra_items * BKT_WORDS * 4);
  const size_t aff_blocks = (size_t)affine_blocks(cap) + 1;
  dalloc((void**)&ctx->d_aff_count, 64);
  ok = ok && hipMemset(ctx->d_aff_count, 0, 64) == hipSuccess;
  dalloc((void**)&ctx->d_aff_stash, (size_t)affine_blocks(cap) * AFF_BLOCK_POINTS * AFF_STASH_WORDS * 4);  // whole workgroups: the stash is piece-major per workgroup
  dalloc((void**)&ctx->d_aff_trees, aff_blocks * 2 * AFF_THREADS * 13 * 4);
  const unsigned host_flags = hipHostMallocMapped | hipHostMallocCoherent;
  ok = ok && hipHostMalloc((void**)&ctx->h_aff_prod, aff_blocks * 48, host_flags) == hipSuccess &&
       hipHostMalloc((void**)&ctx->h_aff_inv, aff_blocks * 48, host_flags) == hipSuccess &&
       hipHostMalloc((void**)&ctx->h_aff_flag, 64, host_flags) == hipSuccess &&
       hipHostGetDevicePointer((void**)&ctx->dm_aff_prod, ctx->h_aff_prod, 0) == hipSuccess &&
       hipHostGetDevicePointer((void**)&ctx->dm_aff_inv, ctx->h_aff_inv, 0) == hipSuccess &&
       hipHostGetDevicePointer((void**)&ctx->dm_aff_flag, ctx->h_aff_flag, 0) == hipSuccess &&
       hipEventCreateWithFlags(&ctx->aff_up_done, hipEventDisableTiming) == hipSuccess;
  if (ok) ctx->aff_scratch.resize(aff_blocks);
  dalloc((void**)&ctx->d_err, 4 * sizeof(int));  // [0], [1]: the two pipeline slots; [2]: base conversion (lives with the table)
  ok = ok && hipHostMalloc((void**)&ctx->h_partials, (size_t)2 * SLOT_WORDS * 4, host_flags) == hipSuccess &&
       hipHostGetDevicePointer((void**)&ctx->dm_partials, ctx->h_partials, 0) == hipSuccess &&
       hipHostMalloc((void**)&ctx->h_out_flag, 64, host_flags) == hipSuccess &&
       hipHostGetDevicePointer((void**)&ctx->dm_out_flag, ctx->h_out_flag, 0) == hipSuccess;
  if (ok) memset(ctx->h_out_flag, 0, 64);
  dalloc((void**)&ctx->d_out_count, 64);
  ok = ok && hipMemset(ctx->d_out_count, 0, 64) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&ctx->h_err, 2 * sizeof(int)) == hipSuccess;
  for (int k = 0; ok && k < 2; k++) ok = ok && hipEventCreateWithFlags(&ctx->done_ev[k], hipEventDisableTiming) == hipSuccess;
  for (int s = 0; ok && s < MSM377_NUM_STAGES; s++)
    for (int k = 0; k < 4; k++) ok = ok && hipEventCreate(&ctx->ev[k >> 1][s][k & 1]) == hipSuccess;
  if (!ok) {
    msm377_ctx_destroy(ctx);
    return MSM377_ENOMEM;
  }
  *out = ctx;
  return MSM377_OK;
}

void msm377_ctx_destroy(msm377_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
  if (ctx->twin) {  // it owns everything but the table it borrows during a batch call
    ctx->twin->d_bases = nullptr;
    ctx->twin->d_table = nullptr;
    msm377_ctx_destroy(ctx->twin);
    ctx->twin = nullptr;
  }
  void* bufs[] = {ctx->d_raw_points, ctx->d_raw_scalars, ctx->d_bases, ctx->d_digits, ctx->d_range_counts, ctx->d_region_base, ctx->d_sort_temp,
                  ctx->d_row_ptr, ctx->d_val_idx, ctx->d_buckets, ctx->d_buckets_snap, ctx->d_partials, ctx->d_work, ctx->d_work_meta, ctx->d_row_ovf_base, ctx->d_split_rows, ctx->d_ovf, ctx->d_err, ctx->d_aff_stash, ctx->d_aff_trees, ctx->d_aff_count, ctx->d_out_count, ctx->d_table, ctx->d_wide_digits, ctx->d_wide_counts, ctx->d_wide_temp, ctx->d_row_ptr_chunks};
  for (void* p : bufs)
    if (p) (void)hipFree(p);
  if (ctx->h_partials) (void)hipHostFree(ctx->h_partials);
  if (ctx->h_err) (void)hipHostFree(ctx->h_err);
  if (ctx->h_out_flag) (void)hipHostFree(ctx->h_out_flag);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  if (ctx->h_aff_prod) (void)hipHostFree(ctx->h_aff_prod);
  if (ctx->h_aff_inv) (void)hipHostFree(ctx->h_aff_inv);
  if (ctx->h_aff_flag) (void)hipHostFree(ctx->h_aff_flag);
  if (ctx->aff_up_done) (void)hipEventDestroy(ctx->aff_up_done);
  for (int t = 0; t < 8; t++)
    if (ctx->copy_stream[t]) (void)hipStreamDestroy(ctx->copy_stream[t]);
  for (int k = 0; k < 2; k++)
    if (ctx->done_ev[k]) (void)hipEventDestroy(ctx->done_ev[k]);
  for (int s = 0; s < MSM377_NUM_STAGES; s++)
    for (int k = 0; k < 2; k++)
      for (int p = 0; p < 2; p++)
        if (ctx->ev[p][s][k]) (void)hipEventDestroy(ctx->ev[p][s][k]);
  if (ctx->bases_ready) (void)hipEventDestroy(ctx->bases_ready);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* msm377_last_error(const msm377_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }














int msm377_g1_window_partials_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint32_t win_begin,
                                     uint32_t win_count, uint8_t* partials_out) {
  if (!partials_out) return MSM377_EINVAL;
  return eng::window_partials(ctx, d_points, d_scalars, n, win_begin, win_count, partials_out, nullptr);
}

int msm377_g1_window_partials_resident(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint32_t win_begin,
                                       uint32_t win_count, void* d_partials_out) {
  if (!d_partials_out) return MSM377_EINVAL;
  return eng::window_partials(ctx, d_points, d_scalars, n, win_begin, win_count, nullptr, d_partials_out);
}


int msm377_g1_combine_window_partials(const uint8_t* partials, uint32_t num_windows, uint8_t out_xy[96]) {
  if (!partials || !out_xy || ((uintptr_t)partials & 3) || num_windows == 0 || num_windows > MSM377_NUM_WINDOWS) return MSM377_EINVAL;
  return g1_combine_tagged(reinterpret_cast<const uint32_t*>(partials), (int)num_windows, out_xy) ? MSM377_EEXCEPTIONAL : MSM377_OK;
}

int msm377_g1_combine_partials_ctx(msm377_ctx* ctx, const uint8_t* partials, uint8_t out_xy[96]) {
  if (!ctx || !partials || !out_xy || ((uintptr_t)partials & 3)) return MSM377_EINVAL;
  const uint32_t* rec = reinterpret_cast<const uint32_t*>(partials);
  bool all_te = true, all_w = true;
  for (int w = 0; w < MSM377_NUM_WINDOWS; w++) {
    const bool te = window_record_is_te(rec + (size_t)w * MSM377_G1_PARTIAL_POINTS * MSM377_G1_POINT_WORDS);
    all_te = all_te && te;
    all_w = all_w && !te;
  }
  int rc = MSM377_OK;
  // (The workers are not pre-woken for this call: between a rank's window_partials call and this combine lie the
  // all-gather and a D2H copy, longer than any spin deadline -- round 2 armed them from window_partials and they spun
  // for nothing, on every rank of the host.)
  if (all_te)
    {
    const int tr = eng::te_tail(ctx, rec, out_xy);
    rc = tr == eng::TAIL_EXCEPTIONAL ? MSM377_EEXCEPTIONAL : tr;
  }
  else if (all_w)
    rc = eng::xyzz_tail(ctx, rec, out_xy);
  else
    rc = g1_combine_tagged(rec, MSM377_NUM_WINDOWS, out_xy) ? MSM377_EEXCEPTIONAL : MSM377_OK;
  if (rc == MSM377_EEXCEPTIONAL) ctx->err = "the partial records add up to an exceptional case of the twisted Edwards law (points outside the prime-order subgroup): recompute them in form 0";
  return rc;
}

int msm377_g1_fold_window_partials(uint8_t* partials, uint32_t win_count) {
  if (!partials || ((uintptr_t)partials & 3) || win_count > MSM377_NUM_WINDOWS) return MSM377_EINVAL;
  g1_fold_tagged(reinterpret_cast<uint32_t*>(partials), (int)win_count);
  return MSM377_OK;
}

int msm377_g1_combine_partials(const uint8_t* partials, uint8_t out_xy[96]) {
  if (!partials || !out_xy || ((uintptr_t)partials & 3)) return MSM377_EINVAL;
  return g1_combine_tagged(reinterpret_cast<const uint32_t*>(partials), MSM377_NUM_WINDOWS, out_xy) ? MSM377_EEXCEPTIONAL : MSM377_OK;
}

int msm377_g1_add_points(const uint8_t* points_xy, uint32_t count, uint8_t out_xy[96]) {
  if (!out_xy || (count && !points_xy)) return MSM377_EINVAL;
  return g1h_add_wire_points(points_xy, count, out_xy) ? MSM377_OK : MSM377_EINVAL;
}

int msm377_g1_combine_partials_split(const uint8_t* partials, uint32_t pieces, uint8_t out_xy[96]) {
  if (!partials || !out_xy || ((uintptr_t)partials & 3) || pieces < 1 || pieces > 64) return MSM377_EINVAL;
  const uint32_t* p = reinterpret_cast<const uint32_t*>(partials);
  for (int w = 0; w < MSM377_NUM_WINDOWS; w++)
    if (!window_record_is_te(p + (size_t)w * 16 * 48)) return MSM377_EINVAL;  // the decomposition of the Edwards tail only
  return teh_combine_split(p, MSM377_NUM_WINDOWS, out_xy, (int)pieces) ? MSM377_EEXCEPTIONAL : MSM377_OK;
}


int msm377_ctx_set_stage_capture(msm377_ctx* ctx, int enabled) {
  if (!ctx) return MSM377_EINVAL;
  if (enabled && !ctx->d_buckets_snap) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (hipMalloc((void**)&ctx->d_buckets_snap, (size_t)MSM377_NUM_WINDOWS * BKT_WORDS * NB * 4) != hipSuccess) return MSM377_ENOMEM;
  }
  ctx->capture = enabled != 0;
  return MSM377_OK;
}

int msm377_g1_read_stage(msm377_ctx* ctx, uint32_t slot, uint16_t* digits, uint32_t* row_ptr, uint32_t* val_idx, uint32_t* buckets) {
  if (!ctx) return MSM377_EINVAL;
  if (!ctx->capture || ctx->last_n == 0 || slot >= ctx->last_wc || ctx->last_form < 0 || ctx->last_glv) {
    ctx->err = "no captured stage data for that window slot";
    return MSM377_ESTATE;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint64_t n = ctx->last_n;
  if (digits) HIP_TRY(ctx, hipMemcpy(digits, ctx->d_digits + (size_t)slot * n, n * 2, hipMemcpyDeviceToHost));
  if (row_ptr) HIP_TRY(ctx, hipMemcpy(row_ptr, ctx->d_row_ptr + (size_t)slot * RP, RP * 4, hipMemcpyDeviceToHost));
  if (val_idx) HIP_TRY(ctx, hipMemcpy(val_idx, ctx->d_val_idx + (size_t)slot * n, n * 4, hipMemcpyDeviceToHost));
  if (buckets) {
    uint32_t* tmp = (uint32_t*)malloc((size_t)BKT_WORDS * NB * 4);
    if (!tmp) return MSM377_ENOMEM;
    hipError_t e = hipMemcpy(tmp, ctx->d_buckets_snap + (size_t)slot * BKT_WORDS * NB, (size_t)BKT_WORDS * NB * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess)  // 64-word records (four 16-word coordinate slots) -> the 52 packed words of the ABI
      for (uint32_t t = 0; t < NB; t++)
        for (uint32_t c = 0; c < 4; c++)
          for (uint32_t j = 0; j < 13; j++) buckets[(size_t)t * PT_WORDS + c * 13 + j] = tmp[(size_t)t * BKT_WORDS + c * 16 + j];
    free(tmp);
    HIP_TRY(ctx, e);
  }
  return MSM377_OK;
}

int msm377_ctx_get_stage_form(const msm377_ctx* ctx) { return (ctx && ctx->capture && ctx->last_n) ? ctx->last_form : -1; }

int msm377_g1_xyzz_to_affine(const uint32_t xyzz[52], uint8_t out_xy[96]) {
  if (!xyzz || !out_xy) return MSM377_EINVAL;
  g1h_to_wire(g1h_from_device_words(xyzz), out_xy);
  return MSM377_OK;
}

int msm377_ctx_set_glv(msm377_ctx* ctx, int mode) {
  if (!ctx || mode < 0 || mode > 2) return MSM377_EINVAL;
  ctx->glv_mode = mode == 1 ? 1 : 0;  // 2 ("the library's choice") is off: see msm377_ctx::glv_mode
  return MSM377_OK;
}

int msm377_ctx_set_g1_form(msm377_ctx* ctx, int form) {
  if (!ctx || form < 0 || form > 1) return MSM377_EINVAL;
  ctx->g1_form = form;
  return MSM377_OK;
}

int msm377_ctx_get_products_per_addition(const msm377_ctx* ctx) { return ctx ? ctx->last_products : 0; }

int msm377_ctx_get_fallback_info(const msm377_ctx* ctx, uint64_t* count, uint32_t* last_mask) {
  if (!ctx) return MSM377_EINVAL;
  if (count) *count = ctx->fallback_count;
  if (last_mask) *last_mask = ctx->fallback_mask;
  return MSM377_OK;
}

int msm377_ctx_set_precompute_window(msm377_ctx* ctx, int window_bits) {
  if (!ctx || (window_bits != MSM377_WINDOW_BITS && window_bits != MSM377_WIDE_WINDOW_BITS)) return MSM377_EINVAL;
  ctx->precomp_bits = window_bits;
  return MSM377_OK;
}

int msm377_ctx_set_narrow_max(msm377_ctx* ctx, uint64_t max_points) {
  if (!ctx) return MSM377_EINVAL;
  ctx->narrow_max_points = max_points;
  return MSM377_OK;
}

int msm377_ctx_set_timing(msm377_ctx* ctx, int enabled) {
  if (!ctx) return MSM377_EINVAL;
  ctx->timing = enabled == 2 ? 2 : (enabled != 0);
  return MSM377_OK;
}

int msm377_ctx_get_stage_ms(msm377_ctx* ctx, double* ms_out) {
  if (!ctx || !ms_out) return MSM377_EINVAL;
  for (int s = 0; s < MSM377_NUM_STAGES; s++) ms_out[s] = ctx->stage_ms[s];
  return MSM377_OK;
}

// ---- entry points that enqueue GPU work: sequencer.hip ----
int msm377_ctx_reserve_host_staging(msm377_ctx* ctx) { return eng::reserve_host_staging(ctx); }
int msm377_g1_msm_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint8_t out_xy[96]) { return eng::g1_msm_device(ctx, d_points, d_scalars, n, out_xy); }
int msm377_g1_msm(msm377_ctx* ctx, const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out_xy[96]) { return eng::g1_msm(ctx, points, scalars, n, out_xy); }
int msm377_ed_msm_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint8_t out_xy[64]) { return eng::ed_msm_device(ctx, d_points, d_scalars, n, out_xy); }
int msm377_ed_msm(msm377_ctx* ctx, const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out_xy[64]) { return eng::ed_msm(ctx, points, scalars, n, out_xy); }
int msm377_ed_generate_bases_device(msm377_ctx* ctx, uint64_t seed, uint64_t n, void* d_points_out) { return eng::ed_generate_bases_device(ctx, seed, n, d_points_out); }
int msm377_g1_set_bases_device(msm377_ctx* ctx, const void* d_points, uint64_t n) { return eng::g1_set_bases_device(ctx, d_points, n); }
int msm377_g1_set_bases(msm377_ctx* ctx, const uint8_t* points, uint64_t n) { return eng::g1_set_bases(ctx, points, n); }
int msm377_g1_set_bases_precomputed_device(msm377_ctx* ctx, const void* d_points, uint64_t n) { return eng::g1_set_bases_precomputed_device(ctx, d_points, n); }
int msm377_g1_set_bases_precomputed(msm377_ctx* ctx, const uint8_t* points, uint64_t n) { return eng::g1_set_bases_precomputed(ctx, points, n); }
int msm377_g1_msm_fixed_base_device(msm377_ctx* ctx, const void* d_scalars, uint64_t n, uint8_t out_xy[96]) { return eng::g1_msm_fixed_base_device(ctx, d_scalars, n, out_xy); }
int msm377_g1_msm_fixed_base_batch_device(msm377_ctx* ctx, const void* d_scalars, uint64_t n, uint32_t batch, uint8_t* out_xy) { return eng::g1_msm_fixed_base_batch_device(ctx, d_scalars, n, batch, out_xy); }
int msm377_g1_msm_fixed_base(msm377_ctx* ctx, const uint8_t* scalars, uint64_t n, uint8_t out_xy[96]) { return eng::g1_msm_fixed_base(ctx, scalars, n, out_xy); }
int msm377_g1_glv_window_partials_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint32_t win_begin, uint32_t win_count, uint8_t* partials_out) { return eng::g1_glv_window_partials_device(ctx, d_points, d_scalars, n, win_begin, win_count, partials_out); }
int msm377_g1_generate_bases_device(msm377_ctx* ctx, uint64_t seed, uint64_t n, void* d_points_out) { return eng::g1_generate_bases_device(ctx, seed, n, d_points_out); }

}  // extern "C"

