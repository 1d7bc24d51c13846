/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, 4x64-bit limbs) of the reference's Twisted-Edwards BLS12 variant of
 * the MSM hot path (BASELINE.json config 3).  At the reference commit
 * the Edwards shaders are orphaned under src/submission/miscellaneous/ and not wired into
 * compute_msm, so this file restates the SAME cuZK pipeline as msm_oracle.c (decompose ->
 * transpose -> SMVP -> BPR -> Horner; src/submission/submission.ts:85-327) with the Edwards point
 * arithmetic of those shaders.  Paths relative to /root/reference/.
 *
 * PARITY STATUS: pinned by the reference's own Edwards vectors, reproduced in
 * tests/test_ed_oracle_pins.py: scalar multiplication (src/reference/utils/FieldMath.test.ts:5-62),
 * x -> y decompression (FieldMath.test.ts:64-98), group addition and group scalar multiplication
 * (src/reference/utils/wasmFunctions.test.ts:24-49), curve parameters
 * (src/reference/params/AleoConstants.ts:2-5, src/reference/utils/FieldMath.ts:104-137); and by
 * Python big integers (tests/pyref.py) through tests/golden/ed_*.bin.  The vendored
 * src/reference/wasm-loader/aleo_wasm_bg.wasm is prebuilt code shipped with the reference and is
 * never loaded.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* exported by msm_oracle.c */
int oracle_decompose_scalars_signed(const uint8_t* scalars, uint64_t n, uint32_t c, uint32_t* chunks);
void oracle_cpu_transpose(const uint32_t* chunks, uint64_t n, uint32_t ncols, uint32_t W, uint32_t* row_ptr, uint32_t* val_idx);

/* ----------------------------------------------------------------- field Fq ---- */
/* q = ALEO_FIELD_MODULUS, src/reference/params/AleoConstants.ts:2 (= BLS12-377 scalar field) */
typedef struct { uint64_t v[4]; } fq;
static const uint64_t Q[4] = {0x0a11800000000001ULL, 0x59aa76fed0000001ULL, 0x60b44d1e5c37b001ULL, 0x12ab655e9a2ca556ULL};
#define QN0 0x0a117fffffffffffULL /* -q^-1 mod 2^64 */
static fq FQ_ONE, FQ_R2, FQ_D; /* Montgomery forms of 1, R, d = 3021 (AleoConstants.ts:4) */
static int q_init = 0;

static int q_geq(const uint64_t* a) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] > Q[i]) return 1;
    if (a[i] < Q[i]) return 0;
  }
  return 1;
}
static void q_sub(uint64_t* a) {
  uint64_t borrow = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a[i] - Q[i] - borrow;
    a[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
}
static void fq_add(fq* r, const fq* a, const fq* b) {
  uint64_t carry = 0, t[4];
  for (int i = 0; i < 4; i++) {
    u128 s = (u128)a->v[i] + b->v[i] + carry;
    t[i] = (uint64_t)s;
    carry = (uint64_t)(s >> 64);
  }
  if (carry || q_geq(t)) q_sub(t);
  memcpy(r->v, t, sizeof t);
}
static void fq_sub(fq* r, const fq* a, const fq* b) {
  uint64_t borrow = 0, t[4];
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a->v[i] - b->v[i] - borrow;
    t[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  if (borrow) {
    uint64_t carry = 0;
    for (int i = 0; i < 4; i++) {
      u128 s = (u128)t[i] + Q[i] + carry;
      t[i] = (uint64_t)s;
      carry = (uint64_t)(s >> 64);
    }
  }
  memcpy(r->v, t, sizeof t);
}
static void fq_mul(fq* r, const fq* a, const fq* b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    uint64_t carry = 0;
    for (int j = 0; j < 4; j++) {
      u128 acc = (u128)a->v[j] * b->v[i] + t[j] + carry;
      t[j] = (uint64_t)acc;
      carry = (uint64_t)(acc >> 64);
    }
    u128 acc = (u128)t[4] + carry;
    t[4] = (uint64_t)acc;
    t[5] = (uint64_t)(acc >> 64);
    uint64_t m = t[0] * QN0;
    acc = (u128)m * Q[0] + t[0];
    carry = (uint64_t)(acc >> 64);
    for (int j = 1; j < 4; j++) {
      acc = (u128)m * Q[j] + t[j] + carry;
      t[j - 1] = (uint64_t)acc;
      carry = (uint64_t)(acc >> 64);
    }
    acc = (u128)t[4] + carry;
    t[3] = (uint64_t)acc;
    t[4] = t[5] + (uint64_t)(acc >> 64);
  }
  if (t[4] || q_geq(t)) q_sub(t);
  memcpy(r->v, t, 32);
}
static void fq_from_bytes(fq* r, const uint8_t* b) {
  fq t;
  for (int i = 0; i < 4; i++) {
    uint64_t w = 0;
    for (int k = 7; k >= 0; k--) w = (w << 8) | b[8 * i + k];
    t.v[i] = w;
  }
  fq_mul(r, &t, &FQ_R2);
}
static void fq_to_bytes(uint8_t* b, const fq* a) {
  fq one, t;
  memset(&one, 0, sizeof one);
  one.v[0] = 1;
  fq_mul(&t, a, &one);
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 8; k++) b[8 * i + k] = (uint8_t)(t.v[i] >> (8 * k));
}
static void fq_inv(fq* r, const fq* a) {
  uint64_t e[4];
  memcpy(e, Q, sizeof e);
  e[0] -= 2;
  fq acc = FQ_ONE;
  for (int i = 255; i >= 0; i--) {
    fq_mul(&acc, &acc, &acc);
    if ((e[i >> 6] >> (i & 63)) & 1) fq_mul(&acc, &acc, a);
  }
  *r = acc;
}
static void ed_init(void) {
  if (q_init) return;
  fq x;
  memset(&x, 0, sizeof x);
  x.v[0] = 1;
  for (int i = 0; i < 256; i++) fq_add(&x, &x, &x);
  FQ_ONE = x;
  for (int i = 0; i < 256; i++) fq_add(&x, &x, &x);
  FQ_R2 = x;
  fq d;
  memset(&d, 0, sizeof d);
  d.v[0] = 3021;
  fq_mul(&FQ_D, &d, &FQ_R2);
  q_init = 1;
}

/* ------------------------------------------- extended twisted Edwards points ---- */
typedef struct { fq x, y, t, z; } edp;

/* identity (0, R, 0, R): src/submission/miscellaneous/wgsl/horners_rule.template.wgsl:19-25 */
static void ed_set_identity(edp* r) {
  memset(r, 0, sizeof *r);
  r->y = FQ_ONE;
  r->z = FQ_ONE;
}
/* add_points: src/submission/miscellaneous/wgsl/add_points_any_a.template.wgsl:24-71
 * (add-2008-hwcd; with a = -1 the shader's h = b - (p - a) is B + A) */
static void ed_add(edp* r, const edp* p1, const edp* p2) {
  fq a, b, t2, c, d, xpy, xpy2, e, f, g, h;
  fq_mul(&a, &p1->x, &p2->x);
  fq_mul(&b, &p1->y, &p2->y);
  fq_mul(&t2, &p1->t, &p2->t);
  fq_mul(&c, &FQ_D, &t2);
  fq_mul(&d, &p1->z, &p2->z);
  fq_add(&xpy, &p1->x, &p1->y);
  fq_add(&xpy2, &p2->x, &p2->y);
  fq_mul(&e, &xpy, &xpy2);
  fq_sub(&e, &e, &a);
  fq_sub(&e, &e, &b);
  fq_sub(&f, &d, &c);
  fq_add(&g, &d, &c);
  fq_add(&h, &b, &a);
  fq_mul(&r->x, &e, &f);
  fq_mul(&r->y, &g, &h);
  fq_mul(&r->t, &e, &h);
  fq_mul(&r->z, &f, &g);
}
/* negate_point (-x, y, -t, z): src/submission/miscellaneous/wgsl/scalar_mul.template.wgsl:66-75 */
static void ed_neg(edp* r, const edp* a) {
  fq zero;
  memset(&zero, 0, sizeof zero);
  *r = *a;
  fq_sub(&r->x, &zero, &a->x);
  fq_sub(&r->t, &zero, &a->t);
}
/* double_and_add, LSB first (horners_rule.template.wgsl:27-44); doubling by the unified add */
static void ed_mul_u64(edp* r, const edp* p, uint64_t s) {
  edp result, temp = *p;
  ed_set_identity(&result);
  while (s) {
    if (s & 1) ed_add(&result, &result, &temp);
    ed_add(&temp, &temp, &temp);
    s >>= 1;
  }
  *r = result;
}
static void ed_mul_bytes(edp* r, const edp* p, const uint8_t* k, int nbytes) {
  edp result, temp = *p;
  ed_set_identity(&result);
  for (int i = 0; i < nbytes * 8; i++) {
    if ((k[i >> 3] >> (i & 7)) & 1) ed_add(&result, &result, &temp);
    ed_add(&temp, &temp, &temp);
  }
  *r = result;
}
/* x || y (32-byte LE each, README.md:299-301) -> (x, y, t = x*y, z = 1):
 * src/submission/miscellaneous/wgsl/convert_inputs.template.wgsl:34-41 */
static void ed_from_affine_bytes(edp* r, const uint8_t* xy) {
  fq_from_bytes(&r->x, xy);
  fq_from_bytes(&r->y, xy + 32);
  fq_mul(&r->t, &r->x, &r->y);
  r->z = FQ_ONE;
}
static void ed_to_affine_bytes(uint8_t* out, const edp* a) {
  fq zi, x, y;
  fq_inv(&zi, &a->z);
  fq_mul(&x, &a->x, &zi);
  fq_mul(&y, &a->y, &zi);
  fq_to_bytes(out, &x);
  fq_to_bytes(out + 32, &y);
}

/* ------------------------------------------------------------------- pipeline ---- */
/* Same stage semantics as msm_oracle.c (smvp_bls12_377.template.wgsl:72-160, bpr.template.wgsl:69-173)
 * with Edwards points. */
static void ed_smvp(const edp* pts, uint32_t ncols, const uint32_t* rp, const uint32_t* vi, edp* buckets) {
  const uint32_t h = ncols / 2;
#pragma omp parallel for schedule(dynamic, 64)
  for (uint32_t id = 0; id < h; id++) {
    edp bucket;
    ed_set_identity(&bucket);
    for (int j = 0; j < 2; j++) {
      uint32_t row_idx = id + h;
      if (j == 1) row_idx = h - id;
      if (j == 0 && id == 0) row_idx = 0;
      edp sum;
      ed_set_identity(&sum);
      for (uint32_t k = rp[row_idx]; k < rp[row_idx + 1]; k++) ed_add(&sum, &sum, &pts[vi[k]]);
      uint32_t bucket_idx;
      if (h > row_idx) { bucket_idx = h - row_idx; ed_neg(&sum, &sum); }
      else bucket_idx = row_idx - h;
      if (bucket_idx > 0) {
        if (j == 1) ed_add(&sum, &bucket, &sum);
        bucket = sum;
      }
    }
    buckets[id] = bucket;
  }
}
static void ed_bpr(const edp* buckets, uint32_t h, uint32_t T, edp* g_points) {
  const uint32_t bpt = h / T;
#pragma omp parallel for schedule(dynamic, 1)
  for (uint32_t tid = 0; tid < T; tid++) {
    uint32_t idx = (tid == 0) ? 0 : (T - tid) * bpt;
    edp m = buckets[idx], g = m;
    for (uint32_t i = 0; i + 1 < bpt; i++) {
      uint32_t bi = (T - tid) * bpt - 1 - i;
      ed_add(&m, &m, &buckets[bi]);
      ed_add(&g, &g, &m);
    }
    edp ms;
    ed_mul_u64(&ms, &m, (uint64_t)bpt * (T - tid - 1));
    ed_add(&g, &g, &ms);
    g_points[tid] = g;
  }
}

int oracle_ed_msm_params(const uint8_t* points, const uint8_t* scalars, uint64_t n, uint32_t c, uint32_t T, uint8_t out_xy[64]) {
  ed_init();
  if (c < 2 || c > 16 || T == 0 || ((1u << (c - 1)) % T) != 0) return -2;
  edp result;
  ed_set_identity(&result);
  if (n == 0) { ed_to_affine_bytes(out_xy, &result); return 0; }
  const uint32_t W = (256 + c - 1) / c, ncols = 1u << c, h = ncols / 2;
  uint32_t* chunks = (uint32_t*)malloc(sizeof(uint32_t) * W * n);
  uint32_t* rp = (uint32_t*)malloc(sizeof(uint32_t) * (uint64_t)W * (ncols + 1));
  uint32_t* vi = (uint32_t*)malloc(sizeof(uint32_t) * W * n);
  edp* pts = (edp*)malloc(sizeof(edp) * n);
  edp* buckets = (edp*)malloc(sizeof(edp) * h);
  edp* gp = (edp*)malloc(sizeof(edp) * T);
  edp* ws = (edp*)malloc(sizeof(edp) * W);
  int rc = oracle_decompose_scalars_signed(scalars, n, c, chunks);
  if (rc == 0) {
#pragma omp parallel for
    for (uint64_t i = 0; i < n; i++) ed_from_affine_bytes(&pts[i], points + 64 * i);
    oracle_cpu_transpose(chunks, n, ncols, W, rp, vi);
    for (uint32_t w = 0; w < W; w++) {
      ed_smvp(pts, ncols, rp + (uint64_t)w * (ncols + 1), vi + (uint64_t)w * n, buckets);
      ed_bpr(buckets, h, T, gp);
      edp acc;
      ed_set_identity(&acc);
      for (uint32_t j = 0; j < T; j++) ed_add(&acc, &acc, &gp[j]);
      ws[w] = acc;
    }
    result = ws[W - 1];
    for (int w = (int)W - 2; w >= 0; w--) {
      for (uint32_t k = 0; k < c; k++) ed_add(&result, &result, &result);
      ed_add(&result, &result, &ws[w]);
    }
    ed_to_affine_bytes(out_xy, &result);
  }
  free(chunks); free(rp); free(vi); free(pts); free(buckets); free(gp); free(ws);
  return rc;
}
/* chunk_size rule of submission.ts:97; see oracle_g1_msm for the n < 65536 branch */
int oracle_ed_msm(const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out_xy[64]) {
  if (n >= 65536) return oracle_ed_msm_params(points, scalars, n, 16, 256, out_xy);
  return oracle_ed_msm_params(points, scalars, n, 4, 8, out_xy);
}
int oracle_ed_msm_naive(const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out_xy[64]) {
  ed_init();
  edp acc;
  ed_set_identity(&acc);
  for (uint64_t i = 0; i < n; i++) {
    edp p, kp;
    ed_from_affine_bytes(&p, points + 64 * i);
    ed_mul_bytes(&kp, &p, scalars + 32 * i, 32);
    ed_add(&acc, &acc, &kp);
  }
  ed_to_affine_bytes(out_xy, &acc);
  return 0;
}
int oracle_ed_scalar_mul(const uint8_t p_xy[64], const uint8_t* k, uint32_t nbytes, uint8_t out_xy[64]) {
  ed_init();
  edp p, r;
  ed_from_affine_bytes(&p, p_xy);
  ed_mul_bytes(&r, &p, k, (int)nbytes);
  ed_to_affine_bytes(out_xy, &r);
  return 0;
}
int oracle_ed_add_affine(const uint8_t a_xy[64], const uint8_t b_xy[64], uint8_t out_xy[64]) {
  ed_init();
  edp a, b, r;
  ed_from_affine_bytes(&a, a_xy);
  ed_from_affine_bytes(&b, b_xy);
  ed_add(&r, &a, &b);
  ed_to_affine_bytes(out_xy, &r);
  return 0;
}
int oracle_ed_on_curve(const uint8_t xy[64]) { /* -x^2 + y^2 = 1 + d x^2 y^2 */
  ed_init();
  fq x, y, x2, y2, l, r, t;
  fq_from_bytes(&x, xy);
  fq_from_bytes(&y, xy + 32);
  fq_mul(&x2, &x, &x);
  fq_mul(&y2, &y, &y);
  fq_sub(&l, &y2, &x2);
  fq_mul(&t, &x2, &y2);
  fq_mul(&t, &t, &FQ_D);
  fq_add(&r, &FQ_ONE, &t);
  return memcmp(&l, &r, sizeof l) == 0;
}
/* Test inputs P_i = [a0 + i*delta]G_ed for i < n (generator: src/reference/utils/FieldMath.ts:108-109),
 * 64 strands walked independently, every point normalised by its own inversion. */
static void add256(uint8_t* r, const uint8_t* a, const uint8_t* b) {
  unsigned c = 0;
  for (int i = 0; i < 32; i++) { unsigned s = a[i] + b[i] + c; r[i] = (uint8_t)s; c = s >> 8; }
}
int oracle_ed_gen_points_arith(uint64_t n, const uint8_t gen_xy[64], const uint8_t a0[32], const uint8_t delta[32], uint8_t* out_points) {
  ed_init();
  if (n == 0) return 0;
  const uint64_t strands = n < 64 ? 1 : 64;
  const uint64_t steps = (n + strands - 1) / strands;
  edp G, D;
  ed_from_affine_bytes(&G, gen_xy);
  ed_mul_bytes(&D, &G, delta, 32);
  uint8_t sd[32];
  memset(sd, 0, 32);
  for (uint64_t s = 0; s < steps; s++) add256(sd, sd, delta);
  uint8_t (*starts)[32] = (uint8_t(*)[32])malloc(32 * strands);
  memcpy(starts[0], a0, 32);
  for (uint64_t j = 1; j < strands; j++) add256(starts[j], starts[j - 1], sd);
#pragma omp parallel for schedule(static, 1)
  for (uint64_t j = 0; j < strands; j++) {
    edp cur;
    ed_mul_bytes(&cur, &G, starts[j], 32);
    for (uint64_t s = 0; s < steps; s++) {
      uint64_t idx = j * steps + s;
      if (idx >= n) break;
      ed_to_affine_bytes(out_points + 64 * idx, &cur);
      ed_add(&cur, &cur, &D);
    }
  }
  free(starts);
  return 0;
}
