/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, 6x64-bit limbs) of the reference's BLS12-377 G1 MSM hot path,
 * used only as the checker in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg.  Nothing under webgpu-msm-bls12-377_amd/ may link, import or call this file.
 *
 * PARITY STATUS: "parity unpinned" at the reference's own oracle boundary.  The reference's
 * CPU path is Aleo.Address.bls12_377_msm from @demox-labs/gpu-wasm-expose 0.3.25
 * (src/reference/reference.ts:24,57; yarn.lock:1320-1323) and its host tail / CPU models use
 * @celo/bls12377js 0.1.1 (src/submission/implementation/cuzk/bls12_377.ts:1); neither is
 * vendored, and the inputs of the 2^16..2^20 known answers (src/test-data/testCases.ts:14-26)
 * live in another repository (README.md:28-35).  What pins this file instead:
 *   - the in-tree constants and KATs it reproduces (tests/test_oracle_pins.py): modulus
 *     (cuzk/bls12_377.ts:10-12), generator on-curve (cuzk/bls12_377.ts:21-29), the fixed base
 *     point (src/ui/AllBenchmarks.tsx:84-85 = cuzk.test.ts:16-21), projective->affine KAT and
 *     negation (miscellaneous/tests/bls12_377.test.ts:8-35), Montgomery constants
 *     (cuzk/utils.ts:448-533), the 16-point pipeline self-consistency test
 *     (miscellaneous/tests/cuzk.test.ts:26-114), the 2^16..2^20 answers being on the curve;
 *   - an independent Python big-integer implementation (oracle/gen_golden.py) whose outputs
 *     are committed under tests/golden/.
 *
 * Every function cites the reference lines it follows.  Paths are relative to
 * /root/reference/src/submission/ unless they start with src/.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ field Fp ---- */
/* p: implementation/cuzk/bls12_377.ts:10-12.  Montgomery radix here is 2^384 (the reference
 * uses 2^390 with 13-bit limbs, cuzk/utils.ts:448-533; the radix is internal). */
typedef struct { uint64_t v[6]; } fp;

static const uint64_t P[6] = {
    0x8508c00000000001ULL, 0x170b5d4430000000ULL, 0x1ef3622fba094800ULL,
    0x1a22d9f300f5138fULL, 0xc63b05c06ca1493bULL, 0x01ae3a4617c510eaULL};
#define N0 0x8508bfffffffffffULL /* -p^-1 mod 2^64 */

static fp FP_ONE, FP_R2; /* R mod p, R^2 mod p; filled by oracle_init() */
static int g_init = 0;

static int fp_is_zero(const fp* a) {
  uint64_t acc = 0;
  for (int i = 0; i < 6; i++) acc |= a->v[i];
  return acc == 0;
}
static int fp_eq(const fp* a, const fp* b) {
  uint64_t acc = 0;
  for (int i = 0; i < 6; i++) acc |= a->v[i] ^ b->v[i];
  return acc == 0;
}
static int geq_p(const uint64_t* a) {
  for (int i = 5; i >= 0; i--) {
    if (a[i] > P[i]) return 1;
    if (a[i] < P[i]) return 0;
  }
  return 1;
}
static void sub_p(uint64_t* a) {
  uint64_t borrow = 0;
  for (int i = 0; i < 6; i++) {
    u128 d = (u128)a[i] - P[i] - borrow;
    a[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
}
/* fr_add: wgsl/field/field.template.wgsl:1-10 (result always canonical here) */
static void fp_add(fp* r, const fp* a, const fp* b) {
  uint64_t carry = 0, t[6];
  for (int i = 0; i < 6; i++) {
    u128 s = (u128)a->v[i] + b->v[i] + carry;
    t[i] = (uint64_t)s;
    carry = (uint64_t)(s >> 64);
  }
  if (carry || geq_p(t)) sub_p(t);
  memcpy(r->v, t, sizeof t);
}
/* fr_sub: wgsl/field/field.template.wgsl:12-32 -- NOT copying its quirk fr_sub(a,a) = p */
static void fp_sub(fp* r, const fp* a, const fp* b) {
  uint64_t borrow = 0, t[6];
  for (int i = 0; i < 6; i++) {
    u128 d = (u128)a->v[i] - b->v[i] - borrow;
    t[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  if (borrow) {
    uint64_t carry = 0;
    for (int i = 0; i < 6; i++) {
      u128 s = (u128)t[i] + P[i] + carry;
      t[i] = (uint64_t)s;
      carry = (uint64_t)(s >> 64);
    }
  }
  memcpy(r->v, t, sizeof t);
}
static void fp_neg(fp* r, const fp* a) {
  fp z;
  memset(&z, 0, sizeof z);
  fp_sub(r, &z, a);
}
/* montgomery_product: wgsl/montgomery/mont_pro_product.template.wgsl:15-62 computes
 * a*b*R^-1 mod p; same contract here (CIOS on 64-bit words). */
static void fp_mul(fp* r, const fp* a, const fp* b) {
  uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 6; i++) {
    uint64_t carry = 0;
    for (int j = 0; j < 6; j++) {
      u128 acc = (u128)a->v[j] * b->v[i] + t[j] + carry;
      t[j] = (uint64_t)acc;
      carry = (uint64_t)(acc >> 64);
    }
    u128 acc = (u128)t[6] + carry;
    t[6] = (uint64_t)acc;
    t[7] = (uint64_t)(acc >> 64);
    uint64_t m = t[0] * N0;
    acc = (u128)m * P[0] + t[0];
    carry = (uint64_t)(acc >> 64);
    for (int j = 1; j < 6; j++) {
      acc = (u128)m * P[j] + t[j] + carry;
      t[j - 1] = (uint64_t)acc;
      carry = (uint64_t)(acc >> 64);
    }
    acc = (u128)t[6] + carry;
    t[5] = (uint64_t)acc;
    t[6] = t[7] + (uint64_t)(acc >> 64);
  }
  if (t[6] || geq_p(t)) sub_p(t);
  memcpy(r->v, t, 48);
}
static void fp_sqr(fp* r, const fp* a) { fp_mul(r, a, a); }

static void fp_from_bytes(fp* r, const uint8_t* b) { /* 48-byte LE canonical -> Montgomery */
  fp t;
  for (int i = 0; i < 6; i++) {
    uint64_t w = 0;
    for (int k = 7; k >= 0; k--) w = (w << 8) | b[8 * i + k];
    t.v[i] = w;
  }
  fp_mul(r, &t, &FP_R2);
}
static void fp_to_bytes(uint8_t* b, const fp* a) { /* Montgomery -> 48-byte LE canonical */
  fp one, t;
  memset(&one, 0, sizeof one);
  one.v[0] = 1;
  fp_mul(&t, a, &one);
  for (int i = 0; i < 6; i++)
    for (int k = 0; k < 8; k++) b[8 * i + k] = (uint8_t)(t.v[i] >> (8 * k));
}
static void fp_inv(fp* r, const fp* a) { /* a^(p-2); createAffinePoint's z_b.inverse(), cuzk/bls12_377.ts:54 */
  uint64_t e[6];
  memcpy(e, P, sizeof e);
  e[0] -= 2;
  fp acc = FP_ONE;
  for (int i = 383; i >= 0; i--) {
    fp_sqr(&acc, &acc);
    if ((e[i >> 6] >> (i & 63)) & 1) fp_mul(&acc, &acc, a);
  }
  *r = acc;
}

static void oracle_init(void) {
  if (g_init) return;
  /* R mod p by 384 doublings of 1; R^2 mod p by 384 more doublings of R. */
  fp x;
  memset(&x, 0, sizeof x);
  x.v[0] = 1;
  for (int i = 0; i < 384; i++) fp_add(&x, &x, &x);
  FP_ONE = x;
  for (int i = 0; i < 384; i++) fp_add(&x, &x, &x);
  FP_R2 = x;
  g_init = 1;
}

/* ------------------------------------------------- G1, projective XYZ (reference) ---- */
typedef struct { fp x, y, z; } g1p;

/* get_paf: wgsl/cuzk/smvp_bls12_377.template.wgsl:29-33 -- (0 : R : 0) */
static void g1_set_identity(g1p* r) {
  memset(r, 0, sizeof *r);
  r->y = FP_ONE;
}
/* is_zero(Z): wgsl/curve/ec_bls12_377.template.wgsl:1-8,17-22 */
static int g1_is_identity(const g1p* a) { return fp_is_zero(&a->z); }

/* add_points: wgsl/curve/ec_bls12_377.template.wgsl:13-52 (add-2002-bj, unified, a = 0) */
static void g1_add(g1p* r, const g1p* p1, const g1p* p2) {
  if (g1_is_identity(p1)) { *r = *p2; return; }
  if (g1_is_identity(p2)) { *r = *p1; return; }
  fp U1, U2, S1, S2, ZZ, T, M, U1U2, TT, R, F, L, G, RR, W, FW, X3, W2, GW2, RGW2, LL, Y3, FF, FFF, Z3;
  fp_mul(&U1, &p1->x, &p2->z);
  fp_mul(&U2, &p2->x, &p1->z);
  fp_mul(&S1, &p1->y, &p2->z);
  fp_mul(&S2, &p2->y, &p1->z);
  fp_mul(&ZZ, &p1->z, &p2->z);
  fp_add(&T, &U1, &U2);
  fp_add(&M, &S1, &S2);
  fp_mul(&U1U2, &U1, &U2);
  fp_mul(&TT, &T, &T);
  fp_sub(&R, &TT, &U1U2);
  fp_mul(&F, &ZZ, &M);
  fp_mul(&L, &M, &F);
  fp_mul(&G, &T, &L);
  fp_mul(&RR, &R, &R);
  fp_sub(&W, &RR, &G);
  fp_mul(&FW, &F, &W);
  fp_add(&X3, &FW, &FW);
  fp_add(&W2, &W, &W);
  fp_sub(&GW2, &G, &W2);
  fp_mul(&RGW2, &R, &GW2);
  fp_mul(&LL, &L, &L);
  fp_sub(&Y3, &RGW2, &LL);
  fp_mul(&FF, &F, &F);
  fp_mul(&FFF, &FF, &F);
  fp_add(&Z3, &FFF, &FFF);
  r->x = X3; r->y = Y3; r->z = Z3;
}
/* double_point: wgsl/curve/ec_bls12_377.template.wgsl:55-80 (dbl-2007-bl, a = 0) */
static void g1_dbl(g1p* r, const g1p* p1) {
  fp XX, w, y1z1, s, ss, sss, R, RR, X1R, X1RX1R, xxrr, B, ww, bb, h, X3, bh, RRRR, wbh, Y3;
  fp_mul(&XX, &p1->x, &p1->x);
  fp_add(&w, &XX, &XX);
  fp_add(&w, &w, &XX);
  fp_mul(&y1z1, &p1->y, &p1->z);
  fp_add(&s, &y1z1, &y1z1);
  fp_mul(&ss, &s, &s);
  fp_mul(&sss, &ss, &s);
  fp_mul(&R, &p1->y, &s);
  fp_mul(&RR, &R, &R);
  fp_add(&X1R, &p1->x, &R);
  fp_mul(&X1RX1R, &X1R, &X1R);
  fp_add(&xxrr, &XX, &RR);
  fp_sub(&B, &X1RX1R, &xxrr);
  fp_mul(&ww, &w, &w);
  fp_add(&bb, &B, &B);
  fp_sub(&h, &ww, &bb);
  fp_mul(&X3, &h, &s);
  fp_sub(&bh, &B, &h);
  fp_add(&RRRR, &RR, &RR);
  fp_mul(&wbh, &w, &bh);
  fp_sub(&Y3, &wbh, &RRRR);
  r->x = X3; r->y = Y3; r->z = sss;
}
/* negate_point: wgsl/cuzk/smvp_bls12_377.template.wgsl:60-68 */
static void g1_neg(g1p* r, const g1p* a) {
  *r = *a;
  fp_neg(&r->y, &a->y);
}
/* double_and_add: wgsl/cuzk/bpr.template.wgsl:42-57 (LSB first); scalarMult, cuzk/bls12_377.ts:68-70 */
static void g1_mul_u64(g1p* r, const g1p* p, uint64_t s) {
  g1p result, temp = *p;
  g1_set_identity(&result);
  while (s) {
    if (s & 1) g1_add(&result, &result, &temp);
    g1_dbl(&temp, &temp);
    s >>= 1;
  }
  *r = result;
}
static void g1_mul_bytes(g1p* r, const g1p* p, const uint8_t* k, int nbytes) { /* LE scalar */
  g1p result, temp = *p;
  g1_set_identity(&result);
  for (int i = 0; i < nbytes * 8; i++) {
    if ((k[i >> 3] >> (i & 7)) & 1) g1_add(&result, &result, &temp);
    g1_dbl(&temp, &temp);
  }
  *r = result;
}
static void g1_from_affine_bytes(g1p* r, const uint8_t* xy) { /* x||y, 48-byte LE each: src/ui/AllBenchmarks.tsx:57-63 */
  fp_from_bytes(&r->x, xy);
  fp_from_bytes(&r->y, xy + 48);
  r->z = FP_ONE;
}
/* createAffinePoint + get_bigint_x_y: cuzk/bls12_377.ts:41-79; identity -> {x:0, y:1} as
 * compute_msm's empty-input return (submission.ts:93-95). */
static void g1_to_affine_bytes(uint8_t* out, const g1p* a) {
  memset(out, 0, 96);
  if (g1_is_identity(a)) { out[48] = 1; return; }
  fp zi, x, y;
  fp_inv(&zi, &a->z);
  fp_mul(&x, &a->x, &zi);
  fp_mul(&y, &a->y, &zi);
  fp_to_bytes(out, &x);
  fp_to_bytes(out + 48, &y);
}

/* -------------------------------------------------------------- pipeline stages ---- */

/* decompose_scalars_signed: cuzk/utils.ts:66-109 and the GPU recode
 * wgsl/cuzk/convert_point_coords_and_decompose_scalars.template.wgsl:100-141.
 * chunks[w*n + i] = signed digit + 2^(c-1).  Returns -1 on a final carry (utils.ts:95-98). */
int oracle_decompose_scalars_signed(const uint8_t* scalars, uint64_t n, uint32_t c, uint32_t* chunks) {
  const uint32_t W = (256 + c - 1) / c, l = 1u << c, h = l >> 1;
  for (uint64_t i = 0; i < n; i++) {
    const uint8_t* s = scalars + 32 * i;
    uint32_t carry = 0;
    for (uint32_t w = 0; w < W; w++) {
      uint32_t limb = 0;
      for (uint32_t b = 0; b < c; b++) {
        uint32_t bit = w * c + b;
        if (bit < 256) limb |= (uint32_t)((s[bit >> 3] >> (bit & 7)) & 1) << b;
      }
      int64_t v = (int64_t)limb + carry;
      if (v >= (int64_t)h) { v -= l; carry = 1; } else carry = 0;
      chunks[(uint64_t)w * n + i] = (uint32_t)(v + h);
    }
    if (carry) return -1;
  }
  return 0;
}

/* cpu_transpose: cuzk/transpose.ts:14-62; GPU twin wgsl/cuzk/transpose_serial.wgsl:34-76.
 * Per subtask: counting sort of point indices by biased digit.  row_ptr has W*(ncols+1)
 * entries, val_idx W*n. */
void oracle_cpu_transpose(const uint32_t* chunks, uint64_t n, uint32_t ncols, uint32_t W, uint32_t* row_ptr, uint32_t* val_idx) {
  uint32_t* curr = (uint32_t*)malloc(sizeof(uint32_t) * ncols);
  for (uint32_t w = 0; w < W; w++) {
    uint32_t* rp = row_ptr + (uint64_t)w * (ncols + 1);
    const uint32_t* col = chunks + (uint64_t)w * n;
    uint32_t* vi = val_idx + (uint64_t)w * n;
    memset(rp, 0, sizeof(uint32_t) * (ncols + 1));
    memset(curr, 0, sizeof(uint32_t) * ncols);
    for (uint64_t j = 0; j < n; j++) rp[col[j] + 1]++;
    for (uint32_t i = 1; i < ncols + 1; i++) rp[i] += rp[i - 1];
    for (uint64_t j = 0; j < n; j++) {
      uint32_t loc = rp[col[j]] + curr[col[j]]++;
      vi[loc] = (uint32_t)j;
    }
  }
  free(curr);
}

/* SMVP, following the WGSL (wgsl/cuzk/smvp_bls12_377.template.wgsl:72-160), NOT cuzk/smvp.ts
 * whose negate() result is discarded (smvp.ts:51).  buckets[id], id in [0,h): id >= 1 holds
 * sum(digit +id) - sum(digit -id); id = 0 holds -sum(digit -h). */
static void smvp_signed(const g1p* pts, uint64_t n, uint32_t ncols, const uint32_t* rp, const uint32_t* vi, g1p* buckets) {
  const uint32_t h = ncols / 2;
#pragma omp parallel for schedule(dynamic, 64)
  for (uint32_t id = 0; id < h; id++) {
    g1p bucket;
    g1_set_identity(&bucket);
    for (int j = 0; j < 2; j++) {
      uint32_t row_idx = id + h;
      if (j == 1) row_idx = h - id;
      if (j == 0 && id == 0) row_idx = 0;
      g1p sum;
      g1_set_identity(&sum);
      for (uint32_t k = rp[row_idx]; k < rp[row_idx + 1]; k++) g1_add(&sum, &sum, &pts[vi[k]]);
      uint32_t bucket_idx;
      if (h > row_idx) { bucket_idx = h - row_idx; g1_neg(&sum, &sum); }
      else bucket_idx = row_idx - h;
      if (bucket_idx > 0) {
        if (j == 1) g1_add(&sum, &bucket, &sum);
        bucket = sum;
      }
    }
    buckets[id] = bucket;
  }
  (void)n;
}

/* bpr stage_1 + stage_2: wgsl/cuzk/bpr.template.wgsl:69-173; models
 * parallel_bucket_reduction_1/2, cuzk/bpr.ts:66-126.  T simulated threads per subtask. */
static void bpr(const g1p* buckets, uint32_t h, uint32_t T, g1p* g_points) {
  const uint32_t bpt = h / T;
#pragma omp parallel for schedule(dynamic, 1)
  for (uint32_t tid = 0; tid < T; tid++) {
    uint32_t idx = (tid == 0) ? 0 : (T - tid) * bpt;
    g1p m = buckets[idx], g = m;
    for (uint32_t i = 0; i + 1 < bpt; i++) {
      uint32_t bi = (T - tid) * bpt - 1 - i;
      g1_add(&m, &m, &buckets[bi]);
      g1_add(&g, &g, &m);
    }
    uint32_t s = bpt * (T - tid - 1);
    g1p ms;
    g1_mul_u64(&ms, &m, s);
    g1_add(&g, &g, &ms);
    g_points[tid] = g;
  }
}

static void load_points(g1p* pts, const uint8_t* points, uint64_t n) {
#pragma omp parallel for
  for (uint64_t i = 0; i < n; i++) g1_from_affine_bytes(&pts[i], points + 96 * i);
}

/* One subtask (window) of compute_msm, submission.ts:199-308: SMVP, BPR, sum of the T
 * partials.  Result in window_sum. */
static void window_sum(const g1p* pts, uint64_t n, uint32_t c, uint32_t T, const uint32_t* rp, const uint32_t* vi, g1p* out) {
  const uint32_t ncols = 1u << c, h = ncols / 2;
  g1p* buckets = (g1p*)malloc(sizeof(g1p) * h);
  g1p* gp = (g1p*)malloc(sizeof(g1p) * T);
  smvp_signed(pts, n, ncols, rp, vi, buckets);
  bpr(buckets, h, T, gp);
  g1p acc;
  g1_set_identity(&acc);
  for (uint32_t j = 0; j < T; j++) g1_add(&acc, &acc, &gp[j]); /* submission.ts:297-308 */
  *out = acc;
  free(buckets);
  free(gp);
}

/* The reference's compute_msm, restated end to end (submission.ts:85-327) for window size c
 * and T bucket-reduction threads.  window_sums_out (optional) receives W affine x||y pairs. */
static int msm_pipeline(const uint8_t* points, const uint8_t* scalars, uint64_t n, uint32_t c, uint32_t T, uint8_t* out_xy, uint8_t* window_sums_out) {
  oracle_init();
  if (n == 0) { memset(out_xy, 0, 96); out_xy[48] = 1; return 0; } /* submission.ts:93-95 */
  const uint32_t W = (256 + c - 1) / c, ncols = 1u << c;
  uint32_t* chunks = (uint32_t*)malloc(sizeof(uint32_t) * W * n);
  uint32_t* rp = (uint32_t*)malloc(sizeof(uint32_t) * (uint64_t)W * (ncols + 1));
  uint32_t* vi = (uint32_t*)malloc(sizeof(uint32_t) * W * n);
  g1p* pts = (g1p*)malloc(sizeof(g1p) * n);
  g1p* ws = (g1p*)malloc(sizeof(g1p) * W);
  int rc = oracle_decompose_scalars_signed(scalars, n, c, chunks);
  if (rc == 0) {
    load_points(pts, points, n);
    oracle_cpu_transpose(chunks, n, ncols, W, rp, vi);
    for (uint32_t w = 0; w < W; w++)
      window_sum(pts, n, c, T, rp + (uint64_t)w * (ncols + 1), vi + (uint64_t)w * n, &ws[w]);
    /* Horner, most significant window first: submission.ts:310-318 */
    g1p result = ws[W - 1];
    for (int w = (int)W - 2; w >= 0; w--) {
      for (uint32_t k = 0; k < c; k++) g1_dbl(&result, &result);
      g1_add(&result, &result, &ws[w]);
    }
    g1_to_affine_bytes(out_xy, &result);
    if (window_sums_out)
      for (uint32_t w = 0; w < W; w++) g1_to_affine_bytes(window_sums_out + 96 * w, &ws[w]);
  }
  free(chunks); free(rp); free(vi); free(pts); free(ws);
  return rc;
}

/* ------------------------------------------------------------------- exported ---- */

/* compute_msm(points, scalars): submission.ts:85-327.  chunk_size rule submission.ts:97; the
 * reference's n < 65536 branch (4-bit windows) cannot run as written (bpr.template.wgsl:85:
 * buckets_per_thread = 8 / 256 = 0 underflows the loop bound), so for it the same pipeline
 * runs with T = 8 threads of one bucket each. */
int oracle_g1_msm(const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out_xy[96]) {
  if (n >= 65536) return msm_pipeline(points, scalars, n, 16, 256, out_xy, 0);
  return msm_pipeline(points, scalars, n, 4, 8, out_xy, 0);
}
/* Same pipeline with explicit parameters (c in {4,8,16}; T divides 2^(c-1)). */
int oracle_g1_msm_params(const uint8_t* points, const uint8_t* scalars, uint64_t n, uint32_t c, uint32_t T, uint8_t out_xy[96], uint8_t* window_sums_out) {
  if (c < 2 || c > 16 || T == 0 || ((1u << (c - 1)) % T) != 0) return -2;
  return msm_pipeline(points, scalars, n, c, T, out_xy, window_sums_out);
}
/* Independent route: sum of k_i * P_i by double-and-add (the "expected" of cuzk.test.ts:106-113). */
int oracle_g1_msm_naive(const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out_xy[96]) {
  oracle_init();
  g1p acc;
  g1_set_identity(&acc);
  for (uint64_t i = 0; i < n; i++) {
    g1p p, kp;
    g1_from_affine_bytes(&p, points + 96 * i);
    g1_mul_bytes(&kp, &p, scalars + 32 * i, 32);
    g1_add(&acc, &acc, &kp);
  }
  g1_to_affine_bytes(out_xy, &acc);
  return 0;
}
/* Horner over W affine window sums (x||y; x=0,y=1 encodes the identity): submission.ts:310-318 */
int oracle_g1_horner(const uint8_t* window_sums, uint32_t W, uint32_t c, uint8_t out_xy[96]) {
  oracle_init();
  static const uint8_t zero48[48] = {0};
  g1p result;
  g1_set_identity(&result);
  for (int w = (int)W - 1; w >= 0; w--) {
    for (uint32_t k = 0; k < c; k++) g1_dbl(&result, &result);
    const uint8_t* xy = window_sums + 96 * w;
    if (memcmp(xy, zero48, 48) == 0 && xy[48] == 1 && memcmp(xy + 49, zero48, 47) == 0) continue;
    g1p q;
    g1_from_affine_bytes(&q, xy);
    g1_add(&result, &result, &q);
  }
  g1_to_affine_bytes(out_xy, &result);
  return 0;
}

/* Stage outputs for stage-level parity tests (the reference's debug=true checks,
 * submission.ts:466-520, 613-641, 724-798). */
int oracle_g1_smvp_window(const uint8_t* points, const uint8_t* scalars, uint64_t n, uint32_t c, uint32_t w, uint8_t* buckets_xy /* 2^(c-1) * 96 */) {
  oracle_init();
  const uint32_t W = (256 + c - 1) / c, ncols = 1u << c, h = ncols / 2;
  if (w >= W) return -2;
  uint32_t* chunks = (uint32_t*)malloc(sizeof(uint32_t) * W * n);
  uint32_t* rp = (uint32_t*)malloc(sizeof(uint32_t) * (uint64_t)W * (ncols + 1));
  uint32_t* vi = (uint32_t*)malloc(sizeof(uint32_t) * W * n);
  g1p* pts = (g1p*)malloc(sizeof(g1p) * n);
  g1p* buckets = (g1p*)malloc(sizeof(g1p) * h);
  int rc = oracle_decompose_scalars_signed(scalars, n, c, chunks);
  if (rc == 0) {
    load_points(pts, points, n);
    oracle_cpu_transpose(chunks, n, ncols, W, rp, vi);
    smvp_signed(pts, n, ncols, rp + (uint64_t)w * (ncols + 1), vi + (uint64_t)w * n, buckets);
    for (uint32_t t = 0; t < h; t++) g1_to_affine_bytes(buckets_xy + 96 * (uint64_t)t, &buckets[t]);
  }
  free(chunks); free(rp); free(vi); free(pts); free(buckets);
  return rc;
}

/* --- helpers for tests and input generation (no reference counterpart) --- */

/* [k]G, G = createGeneratorPoint (cuzk/bls12_377.ts:21-29); k is nbytes little-endian. */
static const char GEN_X_HEX[] = "008848defe740a67c8fc6225bf87ff5485951e2caa9d41bb188282c8bd37cb5cd5481512ffcd394eeab9b16eb21be9ef";
static const char GEN_Y_HEX[] = "01914a69c5102eff1f674f5d30afeec4bd7fb348ca3e52d96d182ad44fb82305c2fe3d3634a9591afd82de55559c8ea6";
static void hex_be_to_le48(uint8_t* out, const char* hex) {
  for (int i = 0; i < 48; i++) {
    unsigned v = 0;
    for (int k = 0; k < 2; k++) {
      char ch = hex[2 * i + k];
      v = v * 16 + (unsigned)((ch <= '9') ? ch - '0' : ch - 'a' + 10);
    }
    out[47 - i] = (uint8_t)v;
  }
}
void oracle_g1_generator(uint8_t out_xy[96]) {
  hex_be_to_le48(out_xy, GEN_X_HEX);
  hex_be_to_le48(out_xy + 48, GEN_Y_HEX);
}
int oracle_g1_scalar_mul(const uint8_t p_xy[96], const uint8_t* k, uint32_t nbytes, uint8_t out_xy[96]) {
  oracle_init();
  g1p p, r;
  g1_from_affine_bytes(&p, p_xy);
  g1_mul_bytes(&r, &p, k, (int)nbytes);
  g1_to_affine_bytes(out_xy, &r);
  return 0;
}
int oracle_g1_add_affine(const uint8_t a_xy[96], const uint8_t b_xy[96], uint8_t out_xy[96]) {
  oracle_init();
  g1p a, b, r;
  g1_from_affine_bytes(&a, a_xy);
  g1_from_affine_bytes(&b, b_xy);
  g1_add(&r, &a, &b);
  g1_to_affine_bytes(out_xy, &r);
  return 0;
}
/* projective (x, y, z canonical 48-byte LE each) -> affine: createAffinePoint, cuzk/bls12_377.ts:41-63 */
int oracle_g1_proj_to_affine(const uint8_t xyz[144], uint8_t out_xy[96]) {
  oracle_init();
  g1p a;
  fp_from_bytes(&a.x, xyz);
  fp_from_bytes(&a.y, xyz + 48);
  fp_from_bytes(&a.z, xyz + 96);
  g1_to_affine_bytes(out_xy, &a);
  return 0;
}
int oracle_g1_on_curve(const uint8_t xy[96]) { /* y^2 = x^3 + 1 */
  oracle_init();
  fp x, y, y2, x3;
  fp_from_bytes(&x, xy);
  fp_from_bytes(&y, xy + 48);
  fp_sqr(&y2, &y);
  fp_sqr(&x3, &x);
  fp_mul(&x3, &x3, &x);
  fp_add(&x3, &x3, &FP_ONE);
  return fp_eq(&y2, &x3);
}
/* a*b mod p, a+b, a-b on canonical 48-byte LE operands (unit tests against Python). */
void oracle_fp_ops(const uint8_t a[48], const uint8_t b[48], uint8_t mul[48], uint8_t add[48], uint8_t sub[48]) {
  oracle_init();
  fp x, y, r;
  fp_from_bytes(&x, a);
  fp_from_bytes(&y, b);
  fp_mul(&r, &x, &y); fp_to_bytes(mul, &r);
  fp_add(&r, &x, &y); fp_to_bytes(add, &r);
  fp_sub(&r, &x, &y); fp_to_bytes(sub, &r);
}
/* Montgomery constants as canonical bytes: R mod p and R^2 mod p for R = 2^384. */
void oracle_fp_mont_constants(uint8_t r1[48], uint8_t r2[48]) {
  oracle_init();
  for (int i = 0; i < 6; i++)
    for (int k = 0; k < 8; k++) {
      r1[8 * i + k] = (uint8_t)(FP_ONE.v[i] >> (8 * k));
      r2[8 * i + k] = (uint8_t)(FP_R2.v[i] >> (8 * k));
    }
}

/* Test-input generator: P_i = [a0 + i*delta]G for i < n, as n affine x||y records.  Points
 * are built as S_j + k*D (S_j by scalar multiplication, D = [delta]G) with one shared
 * inversion per step (Montgomery's trick), so 2^20 points take about a second. */
static void add256(uint8_t* r, const uint8_t* a, const uint8_t* b) {
  unsigned c = 0;
  for (int i = 0; i < 32; i++) { unsigned s = a[i] + b[i] + c; r[i] = (uint8_t)s; c = s >> 8; }
}
int oracle_g1_gen_points_arith(uint64_t n, const uint8_t a0[32], const uint8_t delta[32], uint8_t* out_points) {
  oracle_init();
  if (n == 0) return 0;
  uint64_t strands = 1;
  while (strands * strands < n && strands < 4096) strands <<= 1; /* ~sqrt(n) strands */
  uint64_t steps = (n + strands - 1) / strands;
  uint8_t gen[96];
  oracle_g1_generator(gen);
  g1p G, D;
  g1_from_affine_bytes(&G, gen);
  g1_mul_bytes(&D, &G, delta, 32);
  /* strand j covers indices j*steps .. j*steps+steps-1; start scalar a0 + j*steps*delta (mod 2^256;
   * callers keep a0 + n*delta < r so no wrap matters) */
  fp* xs = (fp*)malloc(sizeof(fp) * strands);
  fp* ys = (fp*)malloc(sizeof(fp) * strands);
  fp* den = (fp*)malloc(sizeof(fp) * strands);
  fp* pre = (fp*)malloc(sizeof(fp) * strands);
  uint8_t sd[32], cur[32];
  memset(sd, 0, 32); /* steps*delta */
  for (uint64_t s = 0; s < steps; s++) add256(sd, sd, delta);
  memcpy(cur, a0, 32);
  for (uint64_t j = 0; j < strands; j++) {
    g1p S;
    uint8_t aff[96];
    g1_mul_bytes(&S, &G, cur, 32);
    g1_to_affine_bytes(aff, &S);
    fp_from_bytes(&xs[j], aff);
    fp_from_bytes(&ys[j], aff + 48);
    add256(cur, cur, sd);
  }
  uint8_t daff[96];
  fp dx, dy;
  g1_to_affine_bytes(daff, &D);
  fp_from_bytes(&dx, daff);
  fp_from_bytes(&dy, daff + 48);
  for (uint64_t s = 0; s < steps; s++) {
    for (uint64_t j = 0; j < strands; j++) {
      uint64_t idx = j * steps + s;
      if (idx < n) { fp_to_bytes(out_points + 96 * idx, &xs[j]); fp_to_bytes(out_points + 96 * idx + 48, &ys[j]); }
    }
    if (s + 1 == steps) break;
    /* batched affine add of D to every strand: lambda = (dy - y)/(dx - x) */
    fp acc = FP_ONE;
    for (uint64_t j = 0; j < strands; j++) {
      fp_sub(&den[j], &dx, &xs[j]);
      if (fp_is_zero(&den[j])) { free(xs); free(ys); free(den); free(pre); return -3; } /* astronomically unlikely */
      pre[j] = acc;
      fp_mul(&acc, &acc, &den[j]);
    }
    fp inv;
    fp_inv(&inv, &acc);
    for (uint64_t jj = strands; jj-- > 0;) {
      fp dinv, lam, l2, x3, y3, t;
      fp_mul(&dinv, &inv, &pre[jj]);
      fp_mul(&inv, &inv, &den[jj]);
      fp_sub(&t, &dy, &ys[jj]);
      fp_mul(&lam, &t, &dinv);
      fp_sqr(&l2, &lam);
      fp_sub(&x3, &l2, &xs[jj]);
      fp_sub(&x3, &x3, &dx);
      fp_sub(&t, &xs[jj], &x3);
      fp_mul(&y3, &lam, &t);
      fp_sub(&y3, &y3, &ys[jj]);
      xs[jj] = x3;
      ys[jj] = y3;
    }
  }
  free(xs); free(ys); free(den); free(pre);
  return 0;
}

int oracle_omp_threads(void) {
#ifdef _OPENMP
  extern int omp_get_max_threads(void);
  return omp_get_max_threads();
#else
  return 1;
#endif
}
