/*
 * msm377 -- C ABI of the MI355X (gfx950) multi-scalar-multiplication engine.
 *
 * This is the drop-in boundary for the reference's hot path
 *     compute_msm(baseAffinePoints: Buffer, scalars: Buffer) -> {x, y}
 *     (/root/reference: src/submission/submission.ts:85-90, called from src/ui/Benchmark.tsx:32
 *      and src/submission/miscellaneous/full_benchmarks.ts:62,99).
 * An N-API shim (webgpu-msm-bls12-377_amd/node/) and a ctypes mirror
 * (webgpu-msm-bls12-377_amd/host/) bind exactly these entry points; see INTEGRATION.md.
 *
 * Wire format (unchanged from the reference harness, src/ui/AllBenchmarks.tsx:57-68 and
 * src/reference/webgpu/utils.ts:63-72):
 *   points   n x 96 bytes : x as 48-byte little-endian || y as 48-byte little-endian,
 *                           canonical residues < p, affine, never the point at infinity
 *   scalars  n x 32 bytes : little-endian integers < 2^255 - 2^239 (the reference requires
 *                           "no final carry" in the signed recode, cuzk/utils.ts:95-98)
 *   result   96 bytes     : affine x || y, 48-byte little-endian each; the identity (and the
 *                           empty input) is x = 0, y = 1 (submission.ts:93-95)
 *
 * All functions return 0 on success or a negative MSM377_E* code; none of them throws or
 * aborts.  A context is not thread-safe: one call in flight per context (the reference has one
 * caller thread).  There is NO CPU fallback: without a usable HIP
 * device every entry point that needs one fails with MSM377_EHIP.
 */
#ifndef MSM377_H
#define MSM377_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSM377_OK 0
#define MSM377_EINVAL (-1)    /* bad argument (null pointer, n over capacity, window range) */
#define MSM377_EHIP (-2)      /* a HIP runtime call failed; see msm377_last_error() */
#define MSM377_ESCALAR (-3)   /* a scalar overflowed the signed 16-bit recode (final carry) */
#define MSM377_ENOMEM (-4)    /* device or host allocation failed */
#define MSM377_ESTATE (-5)    /* call sequence error (e.g. fixed-base MSM before set_bases) */
#define MSM377_EGLVRANGE (-6) /* GLV window sharding only: a scalar >~ 2^254; repeat with the plain window path */
#define MSM377_EEXCEPTIONAL (-7) /* combine of window records only: the twisted-Edwards records add up to an exceptional
                                    case of their (incomplete) addition law -- possible only with input points outside the
                                    prime-order subgroup; recompute the windows in form 0 (msm377_ctx_set_g1_form) and combine
                                    again.  The full-MSM entry points handle this themselves and never return it. */

#define MSM377_NUM_WINDOWS 16          /* ceil(256 / 16): submission.ts:108-109 */
#define MSM377_WINDOW_BITS 16          /* chunk_size for n >= 2^16: submission.ts:97 */
/* One window's partial result: 16 points (plain bucket sum + 15 bit-plane sums), each four
 * coordinates (X, Y, ZZ, ZZZ) of 12 little-endian u32 words in Montgomery form, radix 2^384. */
#define MSM377_G1_PARTIAL_POINTS 16
#define MSM377_G1_POINT_WORDS 48
#define MSM377_G1_WINDOW_PARTIAL_BYTES (MSM377_G1_PARTIAL_POINTS * MSM377_G1_POINT_WORDS * 4)

typedef struct msm377_ctx msm377_ctx;

/* Library / build identification, e.g. "msm377 0.1 gfx950". */
const char* msm377_version(void);
/* Text for a MSM377_E* code. */
const char* msm377_strerror(int code);

/* Create a context on HIP device `device` with workspace for up to `max_points` inputs
 * (replaces get_device + the per-call buffer creation, cuzk/gpu.ts:2-52; unlike the reference
 * the device state persists across calls until msm377_ctx_destroy). */
int msm377_ctx_create(int device, uint64_t max_points, msm377_ctx** out);
void msm377_ctx_destroy(msm377_ctx* ctx);
/* Last HIP / argument error text for this context ("" if none). */
const char* msm377_last_error(const msm377_ctx* ctx);

/* ---- BLS12-377 G1 (short Weierstrass y^2 = x^3 + 1) ------------------------------------ */

/* compute_msm with host buffers: uploads, runs the pipeline, returns the affine result.
 * Replaces submission.ts:85-327 end to end.  Inputs of 2^18 points and more are uploaded in chunks
 * of points that accumulate into the same buckets, so the transfer overlaps the computation
 * (4.3-4.5 ms for 2^20 points from pageable memory, 2.45-2.55 ms with the inputs already on the device).
 * A context is used by one thread at a time. */
int msm377_g1_msm(msm377_ctx* ctx, const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out_xy[96]);

/* Optional, right after msm377_ctx_create: allocate the pinned staging buffer (128 bytes per point of capacity) and the
 * copy streams the host-buffer entry points upload through.  They are otherwise allocated by the first such call, which
 * then takes ~35 ms instead of ~4.5 (the reference's harness times the first call like any other, src/ui/Benchmark.tsx:31-34). */
int msm377_ctx_reserve_host_staging(msm377_ctx* ctx);

/* Same with inputs already in device memory (same wire format).  This is the variant timed
 * by bench.py ("inputs resident in HBM"). */
int msm377_g1_msm_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint8_t out_xy[96]);

/* Fixed-base batches (BASELINE.json config 5): convert and keep a base set in HBM once ... */
int msm377_g1_set_bases(msm377_ctx* ctx, const uint8_t* points, uint64_t n);
int msm377_g1_set_bases_device(msm377_ctx* ctx, const void* d_points, uint64_t n);
/* The same with precomputed window multiples (BASELINE.json config 5, "precomputed-point reuse"; the reference lists
 * precomputation among its own future improvements, /root/reference README.md:558-563): additionally keeps
 * [2^(16 w)] P_i for all 16 windows (16 n affine records: 2.7 GB at n = 2^20, allocated on demand, ~38 ms once).  Every
 * window then gathers points that already carry its weight, so the sixteen bucket sets are simply added together on the
 * GPU: ONE bucket reduction, one partial record and a 16-step host tail per MSM instead of 16 and 256.  Results are
 * identical.  In the Weierstrass form (msm377_ctx_set_g1_form 0) this is msm377_g1_set_bases. */
/* Window width of the tables the next msm377_g1_set_bases_precomputed* call builds: MSM377_WINDOW_BITS (16, default: the
 * layout above) or MSM377_WIDE_WINDOW_BITS (20): 13 windows -- six of 20 bits, then seven of 19, 253 bits in all -- with
 * [2^(offset of window w)] P_i in the table (13 n affine records, 2.2 GB at n = 2^20).
 * Because every window's points already carry its weight, all windows share ONE bucket set, so the window can widen
 * without multiplying buckets: 13 n bucket additions per MSM instead of 16 n over 2^19 buckets -- as many as the
 * 16 x 2^15 of the plain path -- one 19-level reduction, a 20-step host tail.  (21-bit windows are still 13 for a
 * 253-bit scalar; 22-bit ones quadruple the buckets.)  A scalar of 2^253 and more (none below the group order) reruns
 * on the 16-window path over the table's first window.  Results are identical. */
#define MSM377_WIDE_WINDOW_BITS 20
int msm377_ctx_set_precompute_window(msm377_ctx* ctx, int window_bits);
int msm377_g1_set_bases_precomputed(msm377_ctx* ctx, const uint8_t* points, uint64_t n);
int msm377_g1_set_bases_precomputed_device(msm377_ctx* ctx, const void* d_points, uint64_t n);
/* ... then run any number of MSMs of n scalars (host or device pointer) against it. */
int msm377_g1_msm_fixed_base(msm377_ctx* ctx, const uint8_t* scalars, uint64_t n, uint8_t out_xy[96]);
int msm377_g1_msm_fixed_base_device(msm377_ctx* ctx, const void* d_scalars, uint64_t n, uint8_t out_xy[96]);
/* `batch` MSMs over the resident bases in one call: d_scalars holds batch x n x 32 bytes, out_xy
 * receives batch x 96 bytes.  The host tail of MSM b overlaps the GPU work of MSM b+1.  From batch = 4 on the call runs
 * as two halves side by side on two sets of streams and work buffers (a "twin" of the context's buffers, created with
 * the first such call: the context's device memory without the tables, ~0.7 GB at 2^20 points, a second time; if that
 * allocation fails, or with MSM377_TWIN_BATCH=0, the batch runs on one set) -- the low-occupancy ends of one MSM fill
 * with the other half's kernels: 2.16 -> 2.05 ms per MSM at 2^20, 2.00 -> 1.89 on the 20-bit-window table. */
int msm377_g1_msm_fixed_base_batch_device(msm377_ctx* ctx, const void* d_scalars, uint64_t n, uint32_t batch, uint8_t* out_xy);

/* Window sharding for multi-GPU runs (the reference already treats the
 * 16 window subtasks as independent, submission.ts:199-224).  Computes windows
 * [win_begin, win_begin + win_count) only and writes win_count partial records of
 * MSM377_G1_WINDOW_PARTIAL_BYTES each to the HOST buffer partials_out.  The records are opaque: they carry
 * their own coordinate-system tag (twisted Edwards by default; windows that hit an exceptional case of
 * that form, and contexts in form 0, produce Weierstrass records), and the combine functions accept any
 * mixture, so ranks never have to agree on a form. */
int msm377_g1_window_partials_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n,
                                     uint32_t win_begin, uint32_t win_count, uint8_t* partials_out);
/* The same, but the records are left in DEVICE memory (d_partials_out: win_count records, 16-byte aligned): the
 * exchange of a multi-GPU run (one RCCL all-gather over xGMI) reads them where they are, no host round trip.
 * Returns with the context's stream idle, so any other stream may consume the buffer. */
int msm377_g1_window_partials_resident(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n,
                                       uint32_t win_begin, uint32_t win_count, void* d_partials_out);
/* Optional, before the exchange: a rank folds the records of its own win_count CONSECUTIVE windows (in place,
 * same size, same total: one short Horner chain over them; every point but one becomes the identity), which
 * leaves the final combine on every rank with its doublings and one addition per rank instead of 16 per
 * window -- the host tail is a fixed cost that does not shrink with the number of GPUs.  Host-only. */
int msm377_g1_fold_window_partials(uint8_t* partials, uint32_t win_count);
/* Combine the partial records of all MSM377_NUM_WINDOWS windows (window-major, gathered from
 * the ranks) into the final affine result: Horner over the windows, one field inversion.
 * Host-only; needs no context and no device (replaces the CPU tail, submission.ts:290-321). */
int msm377_g1_combine_partials(const uint8_t* partials, uint8_t out_xy[96]);
/* The same result for records in twisted Edwards form, computed the way a context's tail threads compute it -- the
 * Horner chain cut into `pieces` (1..64) balanced pieces, each doubled up to its position, then added -- but on the
 * calling thread: the decomposition of msm377_g1_combine_partials_ctx / the full-MSM entry points, checkable without a
 * GPU (tests/test_host_tail.py).  MSM377_EINVAL for records in Weierstrass form. */
int msm377_g1_combine_partials_split(const uint8_t* partials, uint32_t pieces, uint8_t out_xy[96]);
/* The same on the context's tail threads (the Horner chain cut into up to eight balanced pieces, as inside msm377_g1_msm:
 * MSM377_TAIL_THREADS, default 6): 0.08 instead of 0.14 ms.  MSM377_EHIP if a helper thread does not answer in time. */
int msm377_g1_combine_partials_ctx(msm377_ctx* ctx, const uint8_t* partials, uint8_t out_xy[96]);

/* Point sharding, the other partitioning of a multi-GPU run (SURVEY.md section 8e names it as the fallback): rank g runs a
 * COMPLETE MSM -- msm377_g1_msm_device, all 16 windows, its own host tail -- over its slice [g n / G, (g + 1) n / G) of the
 * points and scalars, the ranks all-gather their 96-byte results and every rank adds them up with this function: the
 * sum of `count` affine wire points (the identity as the wire format writes it, x = 0 and y = 1, is accepted).  Nothing
 * is replicated (conversion, decomposition, sort, reduction and tail all shrink with n / G), so it scales further than
 * window sharding, which replicates the base conversion and whose per-window fixed costs stay.  Host-only, no context.
 * MSM377_EINVAL for a coordinate that is not below p. */
int msm377_g1_add_points(const uint8_t* points_xy, uint32_t count, uint8_t out_xy[96]);

/* The same sharding behind the GLV front end (opt-in, prime-order subgroup points only -- see
 * msm377_ctx_set_glv; Weierstrass form): MSM377_GLV_WINDOWS = 8 windows over {P_i, phi(P_i)};
 * win_begin / win_count index those 8.  Returns MSM377_EGLVRANGE when a scalar does not split into
 * two 127-bit halves (every rank sees the same scalars, so every rank gets the same verdict and the
 * job repeats on the plain 16-window path).  Halving the windows halves the per-rank fixed costs
 * (bucket reduction, host tail), which dominate once the additions are spread over several GPUs. */
#define MSM377_GLV_WINDOWS 8
int msm377_g1_glv_window_partials_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n,
                                         uint32_t win_begin, uint32_t win_count, uint8_t* partials_out);
/* Combine `num_windows` gathered partial records (16 plain, 8 GLV).  Host-only. */
int msm377_g1_combine_window_partials(const uint8_t* partials, uint32_t num_windows, uint8_t out_xy[96]);

/* Synthetic inputs: P_i = [a_i]G, a_i the i-th SplitMix64(seed)
 * output, written in wire format to device memory d_points_out (n x 96 bytes). */
int msm377_g1_generate_bases_device(msm377_ctx* ctx, uint64_t seed, uint64_t n, void* d_points_out);

/* ---- Twisted-Edwards BLS12 ("Edwards-BLS12": a = -1, d = 3021 over the BLS12-377 scalar field;
 *      BASELINE.json config 3; the reference's orphaned Edwards shaders,
 *      src/submission/miscellaneous/wgsl/add_points_any_a.template.wgsl:24-71,
 *      src/reference/params/AleoConstants.ts:2-5) --------------------------------------------
 * Wire format (README.md:299-301): points n x 64 bytes = x || y, 32-byte little-endian each;
 * scalars as above; result 64 bytes x || y; the neutral element (and the empty input) is
 * x = 0, y = 1.  Same pipeline and workspace as G1 (extended coordinates, add-2008-hwcd-3). */
int msm377_ed_msm(msm377_ctx* ctx, const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out_xy[64]);
int msm377_ed_msm_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint8_t out_xy[64]);
/* Synthetic Edwards inputs: P_i = [a_i]G_ed (src/reference/utils/FieldMath.ts:108-109), 64 bytes each. */
int msm377_ed_generate_bases_device(msm377_ctx* ctx, uint64_t seed, uint64_t n, void* d_points_out);

/* ---- stage access for parity tests (the reference's debug=true read-backs,
 *      submission.ts:466-520, 613-641, 724-798) ------------------------------------------- */

/* After any g1 MSM call: copy stage outputs of window slot `slot` (0-based within the windows
 * computed by the last call) to host buffers; any pointer may be NULL.
 *   digits   n u16          biased signed digits, d + 2^15
 *   row_ptr  32770 u32      CSR offsets over keys |d| = 0..32768 (key 0 = digit 0)
 *   val_idx  n u32          point index | (sign << 31), grouped by key
 *   buckets  32768 x 52 u32 bucket t = 1..32768 at row t-1: X, Y, ZZ, ZZZ (13 Montgomery words
 *                           each) as left by bucket accumulation
 * Only valid when the context was created with stage capture enabled.
 *
 * The bucket words describe the coordinate system the call ended in (msm377_ctx_get_stage_form): XYZZ as above for
 * form 0 and for calls that fell back to it, (X, Y, T, Z) extended twisted Edwards coordinates (csrc/te377.hpp: lazy
 * residues below p + 2^354, the identity stored as (0, c, 0, c)) for the default form. */
#define MSM377_STAGE_FORM_XYZZ 0
#define MSM377_STAGE_FORM_TE 1
int msm377_ctx_get_stage_form(const msm377_ctx* ctx); /* -1: nothing captured */
int msm377_ctx_set_stage_capture(msm377_ctx* ctx, int enabled);
int msm377_g1_read_stage(msm377_ctx* ctx, uint32_t slot, uint16_t* digits, uint32_t* row_ptr, uint32_t* val_idx, uint32_t* buckets);
/* Convert one Montgomery XYZZ point (52 words) to the affine wire format (host-only). */
int msm377_g1_xyzz_to_affine(const uint32_t xyzz[52], uint8_t out_xy[96]);

/* GLV front end of the Weierstrass form (msm377_ctx_set_g1_form 0) of the G1 full-MSM entry points:
 * k = k1 + k2 LAMBDA, 8 windows over the 2n points {P_i, phi(P_i)}.  OPT-IN:
 * phi(P) = [LAMBDA] P holds only in the prime-order subgroup, so mode 1 is a promise by the caller that every
 * input point lies in it (true for every protocol use; the reference makes no such assumption, hence the
 * default 0; 2 = the library's choice = 0).  Scalars outside the GLV range (>~ 2^254) rerun on the plain
 * 16-window path automatically.  1.24 vs 1.39 ms at 2^18, 3.51 vs 3.56 ms at 2^20, 12.4 vs 12.6 ms at 2^22. */
int msm377_ctx_set_glv(msm377_ctx* ctx, int mode);

/* Internal coordinate system of the G1 full-MSM entry points (msm, msm_device, set_bases + fixed_base*); results
 * are identical.  form 1 (default): the twisted Edwards form of BLS12-377 G1 (csrc/te377.hpp) -- 7 field products per
 * bucket addition on affine base records (inputs of 2^20 points and more, resident tables), 8 on projective ones,
 * instead of the 10 of form 0, no case distinctions; inputs that hit an exceptional case of its addition law (only
 * possible with points outside the prime-order subgroup) rerun in form 0 automatically.  form 0: short Weierstrass
 * XYZZ coordinates behind the GLV front end selected by msm377_ctx_set_glv.  The window-partials entry points follow
 * the same setting and tag their records with the form they are in (form 1: twisted Edwards records; a shard that hit
 * an exceptional case, or form 0: Weierstrass records); the stage read-backs report theirs (msm377_ctx_get_stage_form). */
int msm377_ctx_set_g1_form(msm377_ctx* ctx, int form);

/* Small inputs: G1 full-MSM calls of at most `max_points` points (default and at most 2^16; 0 = never) run with
 * narrow windows -- 22 windows of 2 048 buckets (eleven signed 12-bit and eleven unsigned 11-bit digits per scalar)
 * instead of 16 of 32 768: 0.22-0.53 instead of 0.52-0.60 ms -- the
 * counterpart of the reference's switch to narrower windows for small inputs (src/submission/submission.ts:97: 4-bit
 * below 65 536 points).  Same results, and the error condition of the 16-bit recode for every input size (the reference's own
 * 4-bit branch rejects more NON-canonical scalars, k > 0x777...7; every k < r passes both); scalars of 2^253 and more rerun
 * on the 16-bit path. */
int msm377_ctx_set_narrow_max(msm377_ctx* ctx, uint64_t max_points);

/* How often this context had to rerun (part of) a call on the Weierstrass path because the twisted Edwards form hit
 * an exceptional case of its addition law, and where the last one surfaced (MSM377_FB_* bits).  Zero for inputs in
 * the prime-order subgroup; the parity tests use it to prove that each check fires. */
#define MSM377_FB_ACCUMULATE 4  /* a bucket addition in k_accumulate */
#define MSM377_FB_MERGE 8       /* the merge of a split row's partial sums */
#define MSM377_FB_TREE 16       /* a bucket-reduction level */
#define MSM377_FB_TAIL 32       /* the host tail (Horner over the partial records) */
#define MSM377_FB_CONVERT 64    /* an input point the Edwards model cannot represent (order 2 or 4) */
int msm377_ctx_get_fallback_info(const msm377_ctx* ctx, uint64_t* count, uint32_t* last_mask);

/* ---- measurement ------------------------------------------------------------------------ */

#define MSM377_STAGE_CONVERT 0     /* points -> Montgomery records */
#define MSM377_STAGE_DECOMPOSE 1   /* scalars -> signed 16-bit digits */
#define MSM377_STAGE_SORT 2        /* histogram + scan + scatter (CSR build) */
#define MSM377_STAGE_ACCUMULATE 3  /* bucket accumulation (SMVP) -- the dominant kernel */
#define MSM377_STAGE_REDUCE 4      /* bucket reduction tree */
#define MSM377_STAGE_TAIL 5        /* D2H of the partial records + host Horner/inversion */
#define MSM377_STAGE_ACC_KERNEL 6  /* the k_accumulate launch alone (inside STAGE_ACCUMULATE, which also
                                      covers the work-list kernels and the split-row merge) */
#define MSM377_NUM_STAGES 7
/* HIP-event timing on the context's streams (off by default).  enabled = 1: every stage; 2: the accumulation
 * kernel alone (MSM377_STAGE_ACC_KERNEL; the other entries read 0).  Every event pair costs a few microseconds of
 * GPU idle time between the launches it separates -- ~50 us per MSM with all stages on -- so a timed loop that only
 * needs the kernel's duration uses 2. */
int msm377_ctx_set_timing(msm377_ctx* ctx, int enabled);
/* Durations in milliseconds of the last call's stages (MSM377_NUM_STAGES entries; the TAIL
 * entry is host wall time). */
int msm377_ctx_get_stage_ms(msm377_ctx* ctx, double* ms_out);
/* Field products per bucket addition of the last accumulation launch (10 Weierstrass XYZZ, 8 twisted Edwards with
 * projective base records, 7 with affine ones): bench.py prices the int32 multiply-add roof with it. */
int msm377_ctx_get_products_per_addition(const msm377_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* MSM377_H */
