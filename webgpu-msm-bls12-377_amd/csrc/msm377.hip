// msm377: BLS12-377 G1 multi-scalar multiplication for MI355X (gfx950), C ABI in
// include/msm377.h.  One translation unit: device kernels, the stage sequencer and the
// host tail.
//
// Pipeline (each stage names the reference code it replaces; paths relative to
// /root/reference/src/submission/):
//   k_affine_up / host inversion / k_affine_down (n >= 2^20, resident tables), k_convert_bases (otherwise)
//                     wire x||y -> Montgomery records            wgsl/cuzk/convert_point_coords_and_decompose_scalars.template.wgsl:41-99 + barrett.template.wgsl:60-82
//   k_decompose (16-bit windows), k_decompose_narrow (inputs <= 2^16 points: 11-bit windows, submission.ts:97)
//                     scalars -> signed digits                   same file :100-141; model cuzk/utils.ts:66-109
//   k_range_count / k_range_scan / k_partition / k_local_sort (k_small_sort on the narrow path)
//                     per-window counting sort -> CSR            wgsl/cuzk/transpose_serial.wgsl:34-76 (16 serial threads there); model cuzk/transpose.ts:14-62
//   k_accumulate      bucket sums (the dominant kernel)          wgsl/cuzk/smvp_bls12_377.template.wgsl:72-160
//   k_tree_step / k_tree_step_quad / k_reduce_tail
//                     bucket reduction, log-depth bit planes     wgsl/cuzk/bpr.template.wgsl:69-173; models cuzk/bpr.ts:5-126
//   host tail         Horner over windows + one inversion        submission.ts:290-321
// The kernels are templates over a curve policy: TeDev (default: G1 in twisted Edwards form, te377.hpp -- 7 field
// products per bucket addition on affine base records (TeAffBase), 8 on projective ones, unified law, exceptional cases
// detected and rerun), G1Dev (G1 in Weierstrass XYZZ coordinates, g1_xyzz.hpp: the fallback, the GLV front end, the
// stage read-backs) and EdDev (Edwards-BLS12 over the scalar field, ed_ext.hpp).  Everything behind the sort takes the
// bucket geometry as a run-time argument L (2^L buckets per window: 15 on the main path, 11 on the narrow one).
//
// HBM layout (n points, W window slots, NB = 32768 buckets per window on the main path):
//   bases    TeDev: n x 160 B affine records (y-x)[13] (y+x)[13] (2dxy)[13] pad[1] u32 (TeAffBase) or n x 256 B
//            projective records (Y-X)[13] (Y+X)[13] (2dT)[13] (2Z)[13] pad[12]; G1Dev: n x 128 B x[13] y[13] pad[6]
//            (29-bit limbs, Montgomery R = 2^406); a gather touches two (one) 128-byte lines
//   digits   W x n u16, window-major: biased digit d + 2^15 (the reference's chunks[] as u32)
//   row_ptr  W x 32770 u32: CSR offsets over keys |d| in 0..32768 (the reference keeps 65537
//            signed rows; here +t and -t share row t and the sign rides in val_idx bit 31)
//   val_idx  W x n u32: point index | sign << 31
//   buckets  W x NB records of 256 bytes, point-major: four 64-byte coordinate slots (13 limbs + 3 zero words);
//            bucket t (key t + 1) of window slot w at record w * NB + t -- one lane writes whole lines
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/msm377.h"
#include "fp64_host.hpp"
#include "te377.hpp"
#include "g1_xyzz.hpp"

using namespace msm377;

namespace {

constexpr uint32_t NB = 32768;     // buckets per window: |d| = 1..32768
constexpr uint32_t NBIN = NB + 1;  // sort keys 0..32768 (key 0 = digit 0, never accumulated)
constexpr uint32_t RP = NBIN + 1;  // row_ptr entries per window
constexpr uint32_t PT_WORDS = 52;  // X, Y, ZZ, ZZZ
constexpr uint32_t BKT_WORDS = 64; // the largest bucket record (four 64-byte coordinate slots), sizes the shared buffers
constexpr uint32_t MAX_SORT_BLOCKS = 256;  // (window slot, chunk) blocks of the partition pass
constexpr uint32_t TREE_LEVELS = 15;       // log2(NB)
constexpr uint32_t SEG_MIN = 16;           // entries per accumulation work item (one thread), see auto_seg(); the work-item and overflow buffers are sized for SEG_MIN
constexpr uint32_t SEG_MAX = 128;
constexpr uint32_t SEG_BINS = SEG_MAX + 1; // work items are counting-sorted by length 0..seg
// Bits of the device error word.  The twisted Edwards form reports an exceptional case of its addition law (te377.hpp)
// with one bit per place it can surface, so that tests can tell which check fired (msm377_ctx_get_fallback_info);
// any of them makes the call rerun on the Weierstrass path.  MSM377_FB_TAIL is raised by the host tail (fp64_host.hpp
// TeChecked) and never lives in the device word.
constexpr int ERR_SCALAR = 1, ERR_GLV_RANGE = 2, ERR_NARROW_RANGE = 128;  // 128: a scalar's top digit does not fit the narrow-window path
constexpr int ERR_TE_EXCEPTIONAL = MSM377_FB_ACCUMULATE, ERR_TE_MERGE = MSM377_FB_MERGE, ERR_TE_TREE = MSM377_FB_TREE, ERR_TE_CONVERT = MSM377_FB_CONVERT;
constexpr int ERR_TE_ANY = ERR_TE_EXCEPTIONAL | ERR_TE_MERGE | ERR_TE_TREE | ERR_TE_CONVERT;
constexpr uint32_t MERGE_GRID = 64;        // workgroups sweeping the list of split rows
constexpr uint32_t NARROW_BITS = 11;       // digit width of the small-input path ...
constexpr uint32_t NARROW_LOG = 11;        // ... whose windows have 2^11 buckets (the unsigned top digit needs the room: k_decompose_narrow)
constexpr uint32_t NARROW_SEG = 8;         // entries per accumulation work item on that path
constexpr uint32_t NARROW_WINDOWS = 23;    // 22 signed 11-bit windows + the top window from bit 242 on
constexpr uint32_t MAX_WINDOW_SLOTS = NARROW_WINDOWS > MSM377_NUM_WINDOWS ? NARROW_WINDOWS : MSM377_NUM_WINDOWS;  // partial-record slots

// ------------------------------------------------------------------ device helpers ----

__device__ __forceinline__ void load_words16(const uint32_t* __restrict__ p, uint32_t* w, int nvec) {
  const uint4* s = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int k = 0; k < nvec; k++) {
    uint4 v = s[k];
    w[4 * k + 0] = v.x;
    w[4 * k + 1] = v.y;
    w[4 * k + 2] = v.z;
    w[4 * k + 3] = v.w;
  }
}

// ---- curve policies: what the curve-agnostic pipeline kernels need from a curve ----
// Base = affine input point as kept in a 128-byte `bases` record; Pt = bucket point.
struct G1Dev {
  static constexpr uint32_t REC_WORDS = 32;  // one base record, 128 bytes
  static constexpr bool HAS_QUAD = true;     // quad-cooperative additions (add_quad below)
  static constexpr uint32_t RAW_WORDS = 24;  // wire: x || y, 48 bytes each
  static constexpr uint32_t PT_WORDS = 52;   // X, Y, ZZ, ZZZ
  static constexpr uint32_t COORD_WORDS = 16;  // a coordinate's slot in a bucket record (13 limbs + 3 pad: 64 bytes)
  static constexpr uint32_t BKT_WORDS = 4 * COORD_WORDS;
  static constexpr uint32_t OUT_WORDS = 48;  // a partial-record point: 4 coordinates x 12 u32 (host-tail format)
  static constexpr uint32_t RECORD_TAG = 0;  // Weierstrass records carry no tag (fp64_host.hpp TE_RECORD_TAG)
  static constexpr int MADD_PRODUCTS = 10;   // field products per bucket addition (8M + 2S)
  static constexpr int FORM_ID = MSM377_STAGE_FORM_XYZZ;
  using F = Fp;
  static constexpr uint32_t NL = 13, NW32 = 12;
  static __device__ __forceinline__ Fp::El to64() { return Fp::from_const(G1Consts::TO64); }
  using Base = G1Affine;
  using Pt = G1XYZZ;
  static __device__ __forceinline__ bool convert(const uint32_t* raw, uint32_t* rec) {  // true: point not representable
    Fp::El x = Fp::to_mont(Fp::from_words<12>(raw));
    Fp::El y = Fp::to_mont(Fp::from_words<12>(raw + 12));
#pragma unroll
    for (int j = 0; j < 13; j++) {
      rec[j] = x.l[j];
      rec[13 + j] = y.l[j];
    }
#pragma unroll
    for (int j = 26; j < 32; j++) rec[j] = 0;
    return false;
  }
  static __device__ __forceinline__ Base load_base(const uint32_t* __restrict__ bases, uint32_t idx) {
    uint32_t w[28];
    load_words16(bases + (size_t)idx * REC_WORDS, w, 7);
    Base p;
#pragma unroll
    for (int j = 0; j < 13; j++) {
      p.x.l[j] = w[j];
      p.y.l[j] = w[13 + j];
    }
    return p;
  }
  static __device__ __forceinline__ bool is_bad(const Pt&) { return false; }  // every case is handled inside the formulas
  static __device__ __forceinline__ bool is_stored_identity(const Pt& p) { return G1::is_identity(p); }
  static __device__ __forceinline__ Pt identity() { return G1::identity(); }
  static __device__ __forceinline__ Pt madd(const Pt& a, const Base& q, bool negq) { return G1::madd(a, q, negq); }  // a + q or a - q
  static __device__ __forceinline__ Pt first(const Base& q, bool negq) { return G1::madd(G1::identity(), q, negq); }  // identity + q: a copy
  static __device__ __forceinline__ Pt add(const Pt& a, const Pt& b) { return G1::add(a, b); }
  static __device__ __forceinline__ void to_words(const Pt& p, uint32_t* w) {
#pragma unroll
    for (int j = 0; j < 13; j++) {
      w[j] = p.x.l[j];
      w[13 + j] = p.y.l[j];
      w[26 + j] = p.zz.l[j];
      w[39 + j] = p.zzz.l[j];
    }
  }
  static __device__ __forceinline__ Pt from_words(const uint32_t* w) {
    Pt p;
#pragma unroll
    for (int j = 0; j < 13; j++) {
      p.x.l[j] = w[j];
      p.y.l[j] = w[13 + j];
      p.zz.l[j] = w[26 + j];
      p.zzz.l[j] = w[39 + j];
    }
    return p;
  }
};

struct EdDev {
  static constexpr uint32_t REC_WORDS = 32;
  static constexpr bool HAS_QUAD = true;
  static constexpr uint32_t RAW_WORDS = 16;  // wire: x || y, 32 bytes each
  static constexpr uint32_t PT_WORDS = 36;   // X, Y, T, Z
  static constexpr uint32_t COORD_WORDS = 12;  // 9 limbs + 3 pad: 48 bytes
  static constexpr uint32_t BKT_WORDS = 4 * COORD_WORDS;
  static constexpr uint32_t OUT_WORDS = 32;  // a partial-record point: 4 coordinates x 8 u32 (host-tail format)
  static constexpr uint32_t RECORD_TAG = 0;
  static constexpr int MADD_PRODUCTS = 7;
  static constexpr int FORM_ID = -1;  // no stage read-back for the Edwards-BLS12 curve
  using F = Fq;
  static constexpr uint32_t NL = 9, NW32 = 8;
  static __device__ __forceinline__ Fq::El to64() { return Fq::from_const(EdConsts::TO64); }
  using Base = EdLazy::ABase;  // buckets and additions in the lazy forms (te377.hpp TeLazy); the law is complete
  using Pt = EdLazy::Ext;
  // record: (y - x)[9] (y + x)[9] (2d x y)[9] pad[5]
  static __device__ __forceinline__ bool convert(const uint32_t* raw, uint32_t* rec) {
    Fq::El x = Fq::to_mont(Fq::from_words<8>(raw));
    Fq::El y = Fq::to_mont(Fq::from_words<8>(raw + 8));
    const Ed::Base b = Ed::make_base(x, y);
#pragma unroll
    for (int j = 0; j < 9; j++) {
      rec[j] = b.ymx.l[j];
      rec[9 + j] = b.ypx.l[j];
      rec[18 + j] = b.kt.l[j];
    }
#pragma unroll
    for (int j = 27; j < 32; j++) rec[j] = 0;
    return false;
  }
  static __device__ __forceinline__ Base load_base(const uint32_t* __restrict__ bases, uint32_t idx) {
    uint32_t w[28];
    load_words16(bases + (size_t)idx * REC_WORDS, w, 7);
    Base p;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      p.ymx.l[j] = w[j];
      p.ypx.l[j] = w[9 + j];
      p.kt.l[j] = w[18 + j];
    }
    return p;
  }

  static __device__ __forceinline__ bool is_bad(const Pt&) { return false; }  // complete addition law
  static __device__ __forceinline__ bool is_stored_identity(const Pt& p) { return Fq::is_zero(p.x) && Fq::eq(p.y, p.z); }  // (0 : c : 0 : c)
  static __device__ __forceinline__ Pt identity() { return EdLazy::identity(); }
  static __device__ __forceinline__ Pt madd(const Pt& a, const Base& q, bool negq) { return EdLazy::madd_affine(a, q, negq); }
  static __device__ __forceinline__ Pt first(const Base& q, bool negq) { return EdLazy::madd_affine(EdLazy::identity(), q, negq); }
  static __device__ __forceinline__ Pt add(const Pt& a, const Pt& b) { return EdLazy::add(a, b); }
  static __device__ __forceinline__ void to_words(const Pt& p, uint32_t* w) {
#pragma unroll
    for (int j = 0; j < 9; j++) {
      w[j] = p.x.l[j];
      w[9 + j] = p.y.l[j];
      w[18 + j] = p.t.l[j];
      w[27 + j] = p.z.l[j];
    }
  }
  static __device__ __forceinline__ Pt from_words(const uint32_t* w) {
    Pt p;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      p.x.l[j] = w[j];
      p.y.l[j] = w[9 + j];
      p.t.l[j] = w[18 + j];
      p.z.l[j] = w[27 + j];
    }
    return p;
  }
};

// BLS12-377 G1 in twisted Edwards form (csrc/te377.hpp): 256-byte records (Y-X, Y+X, 2dT, 2Z), extended buckets.
struct TeDev {
  static constexpr uint32_t REC_WORDS = 64;
  static constexpr bool HAS_QUAD = true;
  static constexpr uint32_t RAW_WORDS = 24;
  static constexpr uint32_t PT_WORDS = 52;   // X, Y, T, Z
  static constexpr uint32_t COORD_WORDS = 16;
  static constexpr uint32_t BKT_WORDS = 4 * COORD_WORDS;
  static constexpr uint32_t OUT_WORDS = 48;
  static constexpr uint32_t RECORD_TAG = TE_RECORD_TAG;  // set in word 11 of every window record's first coordinate
  static constexpr int MADD_PRODUCTS = 8;
  static constexpr int FORM_ID = MSM377_STAGE_FORM_TE;
  using F = Fp;
  static constexpr uint32_t NL = 13, NW32 = 12;
  static __device__ __forceinline__ Fp::El to64() { return Fp::from_const(G1Consts::TO64); }
  using Base = Te377::PBase;
  using Pt = Te377::Ext;
  using Pt_K = G1Consts;  // the curve constants of the lazy law (k_accumulate_quad)
  static __device__ __forceinline__ bool convert(const uint32_t* raw, uint32_t* rec) {
    const Base b = Te377::from_wire(raw, raw + 12, false);
#pragma unroll
    for (int j = 0; j < 13; j++) {
      rec[j] = b.ymx.l[j];
      rec[13 + j] = b.ypx.l[j];
      rec[26 + j] = b.kt.l[j];
      rec[39 + j] = b.z2.l[j];
    }
#pragma unroll
    for (int j = 52; j < 64; j++) rec[j] = 0;
    return Fp::is_zero(b.z2);
  }
  static __device__ __forceinline__ Base load_base(const uint32_t* __restrict__ bases, uint32_t idx) {
    uint32_t w[52];
    load_words16(bases + (size_t)idx * REC_WORDS, w, 13);
    Base p;
#pragma unroll
    for (int j = 0; j < 13; j++) {
      p.ymx.l[j] = w[j];
      p.ypx.l[j] = w[13 + j];
      p.kt.l[j] = w[26 + j];
      p.z2.l[j] = w[39 + j];
    }
    return p;
  }
  static __device__ __forceinline__ bool is_bad(const Pt& p) { return Te377::is_bad(p); }
  static __device__ __forceinline__ bool is_stored_identity(const Pt& p) { return Fp::is_zero(p.x) && Fp::eq(p.y, p.z); }  // (0 : c : 0 : c), c != 0
  static __device__ __forceinline__ Pt identity() { return Te377::identity(); }
  static __device__ __forceinline__ Pt madd(const Pt& a, const Base& q, bool negq) { return Te377::madd(a, q, negq); }
  static __device__ __forceinline__ Pt first(const Base& q, bool negq) { return Te377::from_base(q, negq); }  // 1 product instead of 8
  static __device__ __forceinline__ Pt add(const Pt& a, const Pt& b) { return Te377::add(a, b); }
  static __device__ __forceinline__ void to_words(const Pt& p, uint32_t* w) {
#pragma unroll
    for (int j = 0; j < 13; j++) {
      w[j] = p.x.l[j];
      w[13 + j] = p.y.l[j];
      w[26 + j] = p.t.l[j];
      w[39 + j] = p.z.l[j];
    }
  }
  static __device__ __forceinline__ Pt from_words(const uint32_t* w) {
    Pt p;
#pragma unroll
    for (int j = 0; j < 13; j++) {
      p.x.l[j] = w[j];
      p.y.l[j] = w[13 + j];
      p.t.l[j] = w[26 + j];
      p.z.l[j] = w[39 + j];
    }
    return p;
  }
};

// Affine twisted Edwards records (the batched conversion k_affine_up / k_affine_down; msm377_g1_msm_device and resident
// tables): 7 instead of 8 products per bucket addition and 160-byte records.  Only the base-facing half of the policy
// differs; buckets, reduction and tail are TeDev's.
struct TeAffBase {
  static constexpr uint32_t REC_WORDS = 40;  // (y-x)[13] (y+x)[13] (2dxy)[13] pad[1]; written by k_affine_down
  static constexpr int MADD_PRODUCTS = 7;
  using Base = Te377::ABase;
  using Pt = Te377::Ext;
  static __device__ __forceinline__ Base load_base(const uint32_t* __restrict__ bases, uint32_t idx) {
    uint32_t w[40];
    load_words16(bases + (size_t)idx * REC_WORDS, w, 10);
    Base p;
#pragma unroll
    for (int j = 0; j < 13; j++) {
      p.ymx.l[j] = w[j];
      p.ypx.l[j] = w[13 + j];
      p.kt.l[j] = w[26 + j];
    }
    return p;
  }
  static __device__ __forceinline__ Pt madd(const Pt& a, const Base& q, bool negq) { return Te377::madd_affine(a, q, negq); }
  static __device__ __forceinline__ Pt first(const Base& q, bool negq) { return Te377::from_base_affine(q, negq); }
};

// Bucket records are point-major: coordinate c of a point sits in its own 16-byte-aligned slot of COORD_WORDS words
// (13 limbs + 3 zero words = 64 bytes for the 377-bit field: a G1 bucket is exactly two 128-byte lines), bucket t of
// window slot ws at record ws * NB + t.  Round 1 kept the buckets limb-major so that the thread-per-bucket reduction
// levels were unit-stride -- but the accumulation kernel, which writes every bucket once, hands its work items out
// sorted by LENGTH, so adjacent lanes hold unrelated buckets and each of its 52 four-byte stores per bucket left L2
// as a 32-byte partial write: 0.69 GB written per launch for 0.075 GB of buckets (rocprofv3 WRITE_SIZE,
// profiles/r01_te).  Here a lane writes its bucket as 16 full 16-byte stores into its own two lines.  The overflow
// partials of split rows use the same record.
template <class CV>
__device__ __forceinline__ typename CV::Pt load_record(const uint32_t* __restrict__ p) {
  uint32_t w[CV::PT_WORDS];
#pragma unroll
  for (uint32_t c = 0; c < 4; c++) {
    const uint4* s = reinterpret_cast<const uint4*>(p + c * CV::COORD_WORDS);
#pragma unroll
    for (uint32_t k = 0; k < CV::NL / 4; k++) {
      const uint4 v = s[k];
      w[c * CV::NL + 4 * k + 0] = v.x;
      w[c * CV::NL + 4 * k + 1] = v.y;
      w[c * CV::NL + 4 * k + 2] = v.z;
      w[c * CV::NL + 4 * k + 3] = v.w;
    }
    static_assert(CV::NL % 4 == 1, "one limb beyond the 16-byte groups");
    w[c * CV::NL + CV::NL - 1] = p[c * CV::COORD_WORDS + CV::NL - 1];
  }
  return CV::from_words(w);
}
// One coordinate (NL limbs, the slot's pad words written as zero so that whole 16-byte groups -- whole lines -- go out).
template <class CV>
__device__ __forceinline__ void store_coord(uint32_t* __restrict__ slot, const uint32_t* l) {
  uint4* d = reinterpret_cast<uint4*>(slot);
#pragma unroll
  for (uint32_t k = 0; k < CV::NL / 4; k++) d[k] = make_uint4(l[4 * k], l[4 * k + 1], l[4 * k + 2], l[4 * k + 3]);
  d[CV::NL / 4] = make_uint4(l[CV::NL - 1], 0u, 0u, 0u);
  static_assert(CV::COORD_WORDS == (CV::NL / 4 + 1) * 4, "slot = limbs rounded up to 16 bytes");
}
template <class CV>
__device__ __forceinline__ void store_record(uint32_t* __restrict__ p, const typename CV::Pt& r) {
  uint32_t w[CV::PT_WORDS];
  CV::to_words(r, w);
#pragma unroll
  for (uint32_t c = 0; c < 4; c++) store_coord<CV>(p + c * CV::COORD_WORDS, w + c * CV::NL);
}
// L = log2 of the buckets per window: 15 for the 16-bit windows of the main path, less on the narrow-window path
// for small inputs (a run-time value in every kernel behind the sort: `geometry` in the host code).
template <class CV>
__device__ __forceinline__ uint32_t* bucket_ptr(uint32_t* b, uint32_t L, uint32_t ws, uint32_t t) { return b + (((size_t)ws << L) + t) * CV::BKT_WORDS; }
template <class CV>
__device__ __forceinline__ const uint32_t* bucket_ptr(const uint32_t* b, uint32_t L, uint32_t ws, uint32_t t) { return b + (((size_t)ws << L) + t) * CV::BKT_WORDS; }
template <class CV>
__device__ __forceinline__ typename CV::Pt load_bucket(const uint32_t* __restrict__ b, uint32_t L, uint32_t ws, uint32_t t) {
  return load_record<CV>(bucket_ptr<CV>(b, L, ws, t));
}
template <class CV>
__device__ __forceinline__ void store_bucket(uint32_t* __restrict__ b, uint32_t L, uint32_t ws, uint32_t t, const typename CV::Pt& r) {
  store_record<CV>(bucket_ptr<CV>(b, L, ws, t), r);
}

// ------------------------------------------------------------------------ kernels ----

// Zeroes a few words (error words, work-list counters).  A kernel, not hipMemsetAsync: the runtime's fill kernel took
// 8-18 us per call in the kernel trace (profiles/r02_*), three of them in front of every MSM.
__global__ void __launch_bounds__(256) k_clear_words(uint32_t* __restrict__ a, uint32_t na, uint32_t* __restrict__ b, uint32_t nb) {
  for (uint32_t i = threadIdx.x; i < na; i += 256) a[i] = 0;
  for (uint32_t i = threadIdx.x; i < nb; i += 256) b[i] = 0;
}

// One thread per point: wire record (96 bytes G1, 64 bytes Edwards) -> 128-byte Montgomery record.
template <class CV>  // CV: a curve policy or a base policy (RAW_WORDS, REC_WORDS, convert)
__global__ void __launch_bounds__(256) k_convert_bases(const uint32_t* __restrict__ raw, uint32_t* __restrict__ bases, uint64_t n, int* __restrict__ err) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t w[CV::RAW_WORDS];
  load_words16(raw + i * CV::RAW_WORDS, w, CV::RAW_WORDS / 4);
  uint32_t o[CV::REC_WORDS];
  if (CV::convert(w, o)) atomicOr(err, ERR_TE_CONVERT);
  uint4* dst = reinterpret_cast<uint4*>(bases + i * CV::REC_WORDS);
#pragma unroll
  for (int k = 0; k < (int)CV::REC_WORDS / 4; k++) dst[k] = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
}

// ---- batched conversion to AFFINE twisted Edwards records (7-product bucket additions) ----
//
// An affine record needs 1 / Z_i per point (Z_i = v (u + 1), te377.hpp); a Fermat inversion each is ~450 products, so
// the inverses come from ONE inversion by Montgomery's trick, arranged for width instead of depth:
//   k_affine_up    a thread walks its AFF_K points (numerators N1, N2, denominator Z, running product C of the Z's --
//                  all four go to a stash in HBM, which idles during this phase), a product tree in LDS multiplies the
//                  thread totals of the workgroup (AFF_BLOCK_POINTS points), the tree goes to HBM as well and its root
//                  to the host, in the host's field format
//   host           inverts the n / AFF_BLOCK_POINTS block products (Montgomery's trick again, on the tail threads: one
//                  Fermat inversion per thread, ~25 us, instead of 0.5 ms of serial squarings on one GPU wave)
//   k_affine_down  walks the stored tree down with the block inverse (a node's inverse = its parent's times its
//                  sibling's product), then each thread unfolds its points backwards from the stash and writes the
//                  160-byte records
// 15 field products per point against 6 for the projective record -- spent while the VALUs idle anyway (the conversion
// runs on the side stream beside decomposition and the sort, which are HBM / LDS bound) -- and it takes one product
// off each of the 16 bucket additions the point takes part in.  (A first version kept two points per thread in
// registers and recomputed instead of stashing: 17 products per point, but most of them in tree levels with idle
// lanes -- 0.63 ms, longer than the sort it was meant to hide under.)  Points the Edwards model cannot represent
// (Z = 0: order 2 or 4) enter the product as 1 and raise ERR_TE_CONVERT.
// K = 4 (twice the waves, twice the host's share) measured the same; workgroups of 128 / 64 threads (smaller trees, 2 / 4 times
// the host's share) stretch the conversion stage of a 2^20 MSM from 0.47 to 0.64 / 0.69 ms.
constexpr uint32_t AFF_THREADS = 256, AFF_K = 8, AFF_BLOCK_POINTS = AFF_THREADS * AFF_K;
constexpr uint32_t AFF_STASH_WORDS = 52;  // N1, N2, Z, C (exclusive running product): 13 limbs each, 208 bytes per point

__device__ __forceinline__ void put13(uint32_t* w, const Fp::El& e) {
#pragma unroll
  for (int j = 0; j < 13; j++) w[j] = e.l[j];
}
__device__ __forceinline__ Fp::El get13(const uint32_t* w) {
  Fp::El e;
#pragma unroll
  for (int j = 0; j < 13; j++) e.l[j] = w[j];
  return e;
}
// Where k_affine_up takes point i from: numerators and denominator of its affine Edwards coordinates, x = n1 / z, y = n2 / z.
struct AffWireSource {  // wire format (x || y, canonical Weierstrass coordinates): the map of te377.hpp
  const uint32_t* raw;
  __device__ __forceinline__ bool load(uint64_t i, Fp::El& n1, Fp::El& n2, Fp::El& z) const {
    using K = G1Consts;
    uint32_t w[24];
    load_words16(raw + i * 24, w, 6);
    const Fp::El xr = Fp::from_words<12>(w), yr = Fp::from_words<12>(w + 12);
    const Fp::El u = Fp::add(Fp::mul(xr, Fp::from_const(K::TE_SR)), Fp::from_const(K::TE_S));
    const Fp::El v = Fp::mul(yr, Fp::from_const(K::TE_SR));
    const Fp::El cu = Fp::add(Fp::mul(xr, Fp::from_const(K::TE_CSR)), Fp::from_const(K::TE_CS));
    const Fp::El up = Fp::add(u, Fp::one());
    z = Fp::mul(v, up);
    n1 = Fp::mul(cu, up);
    n2 = Fp::sub(z, Fp::dbl(v));  // (u - 1) v = (u + 1) v - 2 v
    return Fp::is_zero(z);
  }
};
struct AffDoublingSource {  // [2^16] of the point in an affine record of the previous window's table (precomputed-window tables)
  const uint32_t* prev;
  __device__ __forceinline__ bool load(uint64_t i, Fp::El& n1, Fp::El& n2, Fp::El& z) const {
    Te377::Ext p = Te377::from_base_affine(TeAffBase::load_base(prev, (uint32_t)i), false);
    bool bad = false;
#pragma unroll 1
    for (int k = 0; k < MSM377_WINDOW_BITS; k++) {
      p = Te377::add(p, p);  // the unified law doubles
      bad |= Te377::is_bad(p);
    }
    n1 = Fp::canon(p.x);
    n2 = Fp::canon(p.y);
    z = Fp::canon(p.z);
    return bad;
  }
};
// Heap-shaped product tree over the workgroup's thread totals: leaves at AFF_THREADS + tid, root at 1, 13 words a node.
template <class SRC>
__global__ void __launch_bounds__(AFF_THREADS, 4) k_affine_up(SRC src, uint64_t n, uint32_t* __restrict__ stash,
                                                              uint32_t* __restrict__ trees, uint32_t* __restrict__ block_prod, uint32_t* __restrict__ host_flag, uint32_t* __restrict__ dev_count,
                                                              int* __restrict__ err) {
  // block_prod and host_flag live in pinned, coherent HOST memory: the host polls the flag and starts inverting the
  // moment the last workgroup has delivered (a D2H copy queued behind this kernel took 60 us to get through beside the
  // sort, and an event wait adds its wake-up latency on top).  Workgroups count themselves in DEVICE memory -- a
  // system-scope atomic on host memory is a PCIe round trip each, 0.6 ms for 512 of them -- and the last one raises
  // the flag with a plain store.
  using K = G1Consts;
  __shared__ uint32_t tree[2 * AFF_THREADS][13];
  const uint32_t tid = threadIdx.x;
  const uint64_t base = (uint64_t)blockIdx.x * AFF_BLOCK_POINTS + tid;
  bool bad = false;
  Fp::El c = Fp::one();
#pragma unroll 1
  for (uint32_t j = 0; j < AFF_K; j++) {
    const uint64_t i = base + (uint64_t)j * AFF_THREADS;
    if (i >= n) break;
    Fp::El n1, n2, z;
    if (src.load(i, n1, n2, z)) {
      bad = true;
      z = Fp::one();
    }
    uint32_t o[AFF_STASH_WORDS];
    put13(o, n1);
    put13(o + 13, n2);
    put13(o + 26, z);
    put13(o + 39, c);
    uint4* dst = reinterpret_cast<uint4*>(stash + i * AFF_STASH_WORDS);
#pragma unroll
    for (int k = 0; k < (int)AFF_STASH_WORDS / 4; k++) dst[k] = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
    c = Fp::mul(c, z);
  }
  if (bad) atomicOr(err, ERR_TE_CONVERT);
  put13(tree[AFF_THREADS + tid], c);
  for (uint32_t size = AFF_THREADS / 2; size >= 1; size >>= 1) {
    __syncthreads();
    if (tid < size) put13(tree[size + tid], Fp::mul(get13(tree[2 * (size + tid)]), get13(tree[2 * (size + tid) + 1])));
  }
  __syncthreads();
  uint32_t* out = trees + (size_t)blockIdx.x * (2 * AFF_THREADS * 13);
  const uint32_t* flat = &tree[0][0];
  for (uint32_t k = tid; k < 2 * AFF_THREADS * 13; k += AFF_THREADS) out[k] = flat[k];
  if (tid == 0) {  // the root in the host's field format (radix 2^384), like the partial records
    const Fp::El root = Fp::mul(get13(tree[1]), Fp::from_const(K::TO64));
    uint32_t w[12];
    Fp::to_words<12>(root, w);
#pragma unroll
    for (int j = 0; j < 12; j++) block_prod[(size_t)blockIdx.x * 12 + j] = w[j];
    __threadfence_system();  // the product is on its way before this workgroup counts itself
    if (atomicAdd(dev_count, 1u) == gridDim.x - 1) {
      *dev_count = 0;  // ready for the next conversion
      __threadfence_system();
      *reinterpret_cast<volatile uint32_t*>(host_flag) = gridDim.x;
    }
  }
}

// block_inv: 12 words per workgroup, the inverse of its product as a DEVICE Montgomery residue (the host re-bases);
// read straight from pinned host memory.
__global__ void __launch_bounds__(AFF_THREADS, 2) k_affine_down(uint64_t n, const uint32_t* __restrict__ stash, const uint32_t* __restrict__ trees,
                                                                const uint32_t* __restrict__ block_inv, uint32_t* __restrict__ bases) {
  __shared__ uint32_t tree[2 * AFF_THREADS][13];
  const uint32_t tid = threadIdx.x;
  const uint32_t* in = trees + (size_t)blockIdx.x * (2 * AFF_THREADS * 13);
  uint32_t* flat = &tree[0][0];
  for (uint32_t k = tid; k < 2 * AFF_THREADS * 13; k += AFF_THREADS) flat[k] = in[k];
  __syncthreads();
  if (tid == 0) {
    uint32_t w[12];
#pragma unroll
    for (int j = 0; j < 12; j++) w[j] = block_inv[(size_t)blockIdx.x * 12 + j];
    put13(tree[1], Fp::from_words<12>(w));
  }
  // downwards: a node's slot turns from the product of its leaves into the inverse of that product
  for (uint32_t size = 1; size < AFF_THREADS; size <<= 1) {
    __syncthreads();
    if (tid < size) {
      const uint32_t k = size + tid;
      const Fp::El inv_k = get13(tree[k]), a = get13(tree[2 * k]), b = get13(tree[2 * k + 1]);
      put13(tree[2 * k], Fp::mul(inv_k, b));
      put13(tree[2 * k + 1], Fp::mul(inv_k, a));
    }
  }
  __syncthreads();
  Fp::El inv = get13(tree[AFF_THREADS + tid]);  // 1 / (the product of this thread's Z's)
  const uint64_t base = (uint64_t)blockIdx.x * AFF_BLOCK_POINTS + tid;
#pragma unroll 1
  for (int j = (int)AFF_K - 1; j >= 0; j--) {
    const uint64_t i = base + (uint64_t)j * AFF_THREADS;
    if (i >= n) continue;
    uint32_t w[AFF_STASH_WORDS];
    load_words16(stash + i * AFF_STASH_WORDS, w, AFF_STASH_WORDS / 4);
    const Fp::El zi = Fp::mul(inv, get13(w + 39));  // 1 / Z_j = (1 / C_j) C_(j-1)
    inv = Fp::mul(inv, get13(w + 26));              // 1 / C_(j-1)
    const Fp::El x = Fp::mul(get13(w), zi), y = Fp::mul(get13(w + 13), zi);
    const Fp::El ymx = Fp::sub(y, x), ypx = Fp::add(y, x), kt = Fp::mul(Fp::mul(x, y), Fp::from_const(G1Consts::TE_2D));
    uint32_t o[TeAffBase::REC_WORDS];
    put13(o, ymx);
    put13(o + 13, ypx);
    put13(o + 26, kt);
    o[39] = 0;
    uint4* dst = reinterpret_cast<uint4*>(bases + i * TeAffBase::REC_WORDS);
#pragma unroll
    for (int k = 0; k < (int)TeAffBase::REC_WORDS / 4; k++) dst[k] = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
  }
}

// Sort keys: |d| in 0..32768 with the sign carried separately; coarse range = key / 128
// (256 ranges; the last one also owns key 32768).
constexpr uint32_t NRANGE = 256;
constexpr uint32_t KRANGE = NB / NRANGE;  // 128 keys per range
__device__ __forceinline__ void digit_key(uint32_t biased, uint32_t& key, uint32_t& sign) {
  int d = (int)biased - 32768;
  sign = d < 0 ? 1u : 0u;
  key = (uint32_t)(d < 0 ? -d : d);
}
// A window whose digits stay small (the top window: a 253-bit scalar leaves it 13 bits) would crowd all its
// elements into a few of the 256 ranges -- regions far beyond what k_local_sort keeps in LDS.  The decomposition
// records the largest key of window 15 and the sort narrows that window's ranges by a power of two (shift s:
// 128 >> s keys per range, s <= 5) so that the keys in use still spread over the 256 regions; every other
// window slot keeps the full width (key_max = NB).
constexpr uint32_t KEY_TRACKED = 0x80000000u;  // key_max word: bit 31 = "measured", low bits = largest key; 0 = not measured
__device__ __forceinline__ uint32_t win_shift(uint32_t key_max_word) {
  if (!(key_max_word & KEY_TRACKED)) return 0;
  const uint32_t max_key = key_max_word & ~KEY_TRACKED;
  uint32_t s = 0;
  while (s < 5 && max_key < (NB >> (s + 1))) s++;
  return s;
}
__device__ __forceinline__ uint32_t key_range(uint32_t key, uint32_t s) {
  return s == 0 ? (key >= NB ? NRANGE - 1 : key / KRANGE) : key >> (7 - s);
}

// One thread per scalar: 16 signed digits d_w in [-2^15, 2^15), stored biased (d + 2^15).
// Only windows [wb, wb + wc) are written (window sharding); the carry chain always runs over
// all 16.  A final carry (scalar >= 2^255 - 2^239) sets *err, as cuzk/utils.ts:95-98 throws.
__global__ void __launch_bounds__(256) k_decompose(const uint32_t* __restrict__ scalars, uint16_t* __restrict__ digits, uint64_t n,
                                                   uint32_t wb, uint32_t wc, int* __restrict__ err, uint32_t* __restrict__ top_key_max) {
  // top_key_max (may be null): largest key of window 15, the one window that scalars below a 253-bit modulus leave
  // mostly empty; see win_shift.  One LDS atomic per thread at worst, one global atomic per block.
  __shared__ uint32_t wmax;
  if (threadIdx.x == 0) wmax = 0;
  __syncthreads();
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    uint32_t w[8];
    load_words16(scalars + i * 8, w, 2);
    uint32_t carry = 0;
#pragma unroll
    for (uint32_t win = 0; win < 16; win++) {
      uint32_t limb = (w[win >> 1] >> (16 * (win & 1))) & 0xffffu;
      uint32_t v = limb + carry;
      carry = v >= 32768u ? 1u : 0u;
      if (win >= wb && win < wb + wc) {
        const uint32_t biased = (v + 32768u) & 0xffffu;
        digits[(size_t)(win - wb) * n + i] = (uint16_t)biased;
        if (win == 15 && top_key_max) {
          uint32_t key, sign;
          digit_key(biased, key, sign);
          if ((key | KEY_TRACKED) > wmax) atomicMax(&wmax, key | KEY_TRACKED);
        }
      }
    }
    if (carry) atomicOr(err, 1);
  }
  __syncthreads();
  if (threadIdx.x == 0 && top_key_max && wmax > *top_key_max) atomicMax(top_key_max, wmax);
}

// ---- narrow windows for small inputs (the reference switches to 4-bit windows below
//      65 536 points, src/submission/submission.ts:97,173-186) ----
// Below ~2^15 points the 16 x 32 768 buckets of the main path are mostly empty and their reduction -- 15 levels, the
// first ones streaming 134 MB of identity records -- is most of the call.  With 11-bit windows (23 windows of 2 048
// buckets) the bucket array shrinks 11-fold and the reduction loses four levels (0.28 -> 0.115 ms); the additions grow
// from 16 n to 23 n, which a small input does not notice.  Everything behind the sort runs the same kernels with
// L = 11 as their run-time bucket geometry; decomposition and sort have their own small kernels here.

// One thread per scalar: W windows of c bits.  Windows 0 .. W-2 are signed digits with a carry (|d| <= 2^(c-1)); the TOP
// window takes everything that is left WITHOUT a carry out, as an unsigned digit: a signed top window would push
// its carry into one more window whose only digits are 0 and 1 -- a single row holding a seventh of all points,
// which no segmenting saves on a small input (measured: the merge of that row alone took 2.6 ms at 2^14).  With
// c = 11 the top window starts at bit 242, so scalars below r (253 bits) leave it digits below 1 195 < 2^L = 2 048;
// a larger top digit (scalars >= 2^253) raises ERR_NARROW_RANGE and the call reruns on the 16-bit path.  All digits
// are stored biased by 2^L: d + 2^L in [0, 2^(L+1)).
// The error condition stays the reference's (cuzk/utils.ts:95-98 throws when the 16-bit recode ends with a carry):
// the same inputs are rejected whichever window width runs.
__global__ void __launch_bounds__(256) k_decompose_narrow(const uint32_t* __restrict__ scalars, uint16_t* __restrict__ digits, uint64_t n, uint32_t c,
                                                          uint32_t L, uint32_t W, int* __restrict__ err) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  load_words16(scalars + i * 8, w, 2);
  const uint32_t half = 1u << (c - 1), mask = (1u << c) - 1u, bias = 1u << L;
  uint32_t carry = 0;
  for (uint32_t win = 0; win + 1 < W; win++) {
    const uint32_t bit = win * c, word = bit >> 5, off = bit & 31;
    uint32_t v = w[word] >> off;
    if (off + c > 32 && word + 1 < 8) v |= w[word + 1] << (32 - off);
    v = (v & mask) + carry;
    carry = v >= half ? 1u : 0u;
    const int d = (int)v - (int)(carry << c);  // in [-2^(c-1), 2^(c-1))
    digits[(size_t)win * n + i] = (uint16_t)(d + (int)bias);
  }
  {  // the top window: bits (W - 1) c .. 255, unsigned, no carry out (at most 32 bits wide for the widths in use)
    const uint32_t bit = (W - 1) * c, word = bit >> 5, off = bit & 31;
    uint64_t v = w[word] >> off;
    for (uint32_t k = word + 1, sh = 32 - off; k < 8; k++, sh += 32) v |= (uint64_t)w[k] << sh;
    v += carry;
    if (v >= bias) atomicOr(err, ERR_NARROW_RANGE);
    digits[(size_t)(W - 1) * n + i] = (uint16_t)((uint32_t)(v < bias ? v : 0) + bias);
  }
  uint32_t carry16 = 0;
#pragma unroll
  for (uint32_t win = 0; win < 16; win++) carry16 = (((w[win >> 1] >> (16 * (win & 1))) & 0xffffu) + carry16) >= 32768u ? 1u : 0u;
  if (carry16) atomicOr(err, ERR_SCALAR);
}

// One workgroup per window: counting sort of the window's n <= SMALL_SORT_MAX digits by key |d| in LDS, straight to the
// CSR form the accumulation reads (row_ptr: 2^L + 2 offsets per window over keys 0 .. 2^L; val_idx: index | sign << 31).
constexpr uint32_t SMALL_SORT_MAX = 1u << 16;
constexpr uint32_t SMALL_BINS_MAX = (1u << 12) + 1;  // keys 0 .. 2^L for L <= 12
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

// ---- GLV front end (the reference lists it as future work, README.md:562) ----
// phi(x, y) = (BETA x, y) = [LAMBDA](x, y) on G1, LAMBDA = x0^2 - 1 (127 bits).  A scalar
// k < r splits as k = k1 + k2 LAMBDA with k2 = floor(k / LAMBDA), k1 = k mod LAMBDA, both
// non-negative and < 2^127, so sum k_i P_i = sum k1_i P_i + sum k2_i phi(P_i): 2n points with
// 128-bit scalars, i.e. EIGHT 16-bit windows over 2n points instead of sixteen over n.  The
// bucket additions are the same 16n, but there are half as many buckets to reduce, half as many
// Horner steps on the host, and no short top window (both halves fill their top window to
// ~2^14: no 219-entry rows, no split rows).  Everything after this front end is the unchanged
// pipeline run with wc = 8 window slots over 2n points.

// Record i = P_i, record n + i = phi(P_i).
__global__ void __launch_bounds__(256) k_convert_bases_glv(const uint32_t* __restrict__ raw, uint32_t* __restrict__ bases, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t w[24];
  load_words16(raw + i * 24, w, 6);
  const Fp::El x = Fp::to_mont(Fp::from_words<12>(w));
  const Fp::El y = Fp::to_mont(Fp::from_words<12>(w + 12));
  const Fp::El bx = Fp::mul(x, Fp::from_const(G1Consts::BETA));
  constexpr uint32_t REC_WORDS = G1Dev::REC_WORDS;
  uint32_t o[REC_WORDS];
#pragma unroll
  for (int j = 0; j < 13; j++) {
    o[j] = x.l[j];
    o[13 + j] = y.l[j];
  }
#pragma unroll
  for (int j = 26; j < 32; j++) o[j] = 0;
  uint4* dst = reinterpret_cast<uint4*>(bases + i * REC_WORDS);
#pragma unroll
  for (int k = 0; k < 8; k++) dst[k] = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
#pragma unroll
  for (int j = 0; j < 13; j++) o[j] = bx.l[j];
  dst = reinterpret_cast<uint4*>(bases + (n + i) * REC_WORDS);
#pragma unroll
  for (int k = 0; k < 8; k++) dst[k] = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
}

// out[0..NA+NB) = a * b on 32-bit words (schoolbook, carries resolved per row).
template <int NA, int NB_>
__device__ __forceinline__ void mul_words(const uint32_t* a, const uint32_t* b, uint32_t* out) {
#pragma unroll
  for (int k = 0; k < NA + NB_; k++) out[k] = 0;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < NB_; j++) {
      const uint64_t t = (uint64_t)a[i] * b[j] + out[i + j] + carry;
      out[i + j] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
    out[i + NB_] = carry;
  }
}

// Eight signed 16-bit digits of a 128-bit value v < 2^127 (top digit stays non-negative); windows
// [wb, wb + wc) are written to slots 0..wc-1.  Returns non-zero if the value does not fit (top window
// reaches 2^15).
__device__ __forceinline__ uint32_t recode128(const uint32_t* v, uint16_t* __restrict__ digits, size_t stride, size_t col, uint32_t wb,
                                              uint32_t wc) {
  uint32_t carry = 0, bad = 0;
#pragma unroll
  for (uint32_t win = 0; win < 8; win++) {
    const uint32_t limb = (v[win >> 1] >> (16 * (win & 1))) & 0xffffu;
    const uint32_t t = limb + carry;
    carry = (win < 7 && t >= 32768u) ? 1u : 0u;
    if (win >= wb && win < wb + wc) digits[(size_t)(win - wb) * stride + col] = (uint16_t)((t + 32768u) & 0xffffu);
    if (win == 7 && t >= 32768u) bad = 1u;
  }
  return bad;
}

// One thread per scalar: k -> (k1, k2) by a Barrett quotient (MU = floor(2^384 / LAMBDA), at
// most one correction), then the signed digits of k1 into column i and of k2 into column n + i of
// the 8 x 2n digit matrix.  Scalars outside the GLV range (k2 >= 2^127, i.e. k >~ 2^254) set bit 1
// of *err: the host then reruns the call on the plain 16-window path.
__global__ void __launch_bounds__(256) k_decompose_glv(const uint32_t* __restrict__ scalars, uint16_t* __restrict__ digits, uint64_t n,
                                                       uint32_t wb, uint32_t wc, int* __restrict__ err) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t k[8];
  load_words16(scalars + i * 8, k, 2);
  uint32_t prod[17];
  mul_words<8, 9>(k, GlvConsts::MU, prod);
  uint32_t q[5];
#pragma unroll
  for (int j = 0; j < 5; j++) q[j] = prod[12 + j];  // floor(k MU / 2^384): the quotient or one less
  uint32_t ql[9];
  mul_words<5, 4>(q, GlvConsts::LAMBDA, ql);
  uint32_t rem[5];
  {
    uint32_t borrow = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const uint64_t d = (uint64_t)k[j] - ql[j] - borrow;
      rem[j] = (uint32_t)d;
      borrow = (uint32_t)(d >> 32) & 1u;
    }
  }
  // rem in [0, 2 LAMBDA): one conditional correction
  uint32_t sub[5];
  uint32_t borrow = 0;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const uint64_t d = (uint64_t)rem[j] - (j < 4 ? GlvConsts::LAMBDA[j] : 0u) - borrow;
    sub[j] = (uint32_t)d;
    borrow = (uint32_t)(d >> 32) & 1u;
  }
  if (!borrow) {
#pragma unroll
    for (int j = 0; j < 5; j++) rem[j] = sub[j];
    uint32_t c = 1;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const uint64_t t = (uint64_t)q[j] + c;
      q[j] = (uint32_t)t;
      c = (uint32_t)(t >> 32);
    }
  }
  uint32_t bad = q[4] | rem[4];
  bad |= recode128(rem, digits, (size_t)2 * n, (size_t)i, wb, wc);
  bad |= recode128(q, digits, (size_t)2 * n, (size_t)(n + i), wb, wc);
  if (bad) atomicOr(err, 2);
}

// ---- per-window counting sort (the reference's transpose, transpose_serial.wgsl:34-76) ----
//
// Two-level (MSD) counting sort; every pass touches each (window, point) element once:
//   k_range_count  block (chunk, window): LDS histogram of the chunk over 256 coarse key ranges
//   k_range_scan   block per window: region bases per range, per-chunk write offsets
//   k_partition    block (chunk, window): appends each element (index|sign, key) to its
//                  range's region at LDS-ranked offsets -- contiguous runs, no global atomics
//   k_local_sort   block (range, window): counting sort of the region's <= 129 keys in LDS;
//                  writes its row_ptr slice and its val_idx slice, a CONTIGUOUS output owned by
//                  one block, so the 4-byte stores combine in that XCD's L2.
// (The first version scattered straight from the digit columns: every 4-byte store then left
// L2 as a partial write, 8x the payload, 213 us at n = 2^20.)  Order inside a bucket is free:
// group addition commutes (the reference's transpose is stable only because it is serial).

struct SortElem {
  uint32_t idx_sign;  // point index | sign << 31
  uint32_t key;       // |d|
};

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
                                                      uint64_t n, uint32_t chunks, uint64_t per_chunk, const uint32_t* __restrict__ key_max) {
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
__global__ void __launch_bounds__(NRANGE) k_range_scan(uint32_t* __restrict__ counts, uint32_t* __restrict__ region_base, uint32_t chunks) {
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

__global__ void __launch_bounds__(1024) k_partition(const uint16_t* __restrict__ digits, const uint32_t* __restrict__ counts,
                                                    SortElem* __restrict__ temp, uint64_t n, uint32_t chunks, uint64_t per_chunk,
                                                    const uint32_t* __restrict__ key_max) {
  __shared__ uint32_t cur[NRANGE];
  const uint32_t c = blockIdx.x, ws = blockIdx.y, tid = threadIdx.x;
  const uint32_t shift = win_shift(key_max[ws]);
  if (tid < NRANGE) cur[tid] = counts[((size_t)ws * NRANGE + tid) * chunks + c];
  __syncthreads();
  const uint64_t beg = (uint64_t)c * per_chunk;
  const uint64_t end = (beg + per_chunk < n) ? beg + per_chunk : n;
  SortElem* out = temp + (size_t)ws * n;
  for_each_digit(digits + (size_t)ws * n, beg, end, tid, 1024, [&](uint64_t i, uint32_t biased) {
    uint32_t key, sign;
    digit_key(biased, key, sign);
    out[atomicAdd(&cur[key_range(key, shift)], 1u)] = SortElem{(uint32_t)i | (sign << 31), key};
  });
}

// Block (range r, window slot ws), 256 threads: the region holds exactly the elements with keys
// in [r KRANGE, (r + 1) KRANGE) (plus key 32768 for the last range).  The region was written by
// other CUs, so every load misses L2: a region of up to LS_CACHE elements (n / 256 = 4096 on
// average at n = 2^20) is read ONCE, eight 8-byte loads in flight per thread, and kept in LDS for
// the scatter pass; longer regions are streamed twice.
constexpr uint32_t LS_CACHE = 6144;
__global__ void __launch_bounds__(256) k_local_sort(const SortElem* __restrict__ temp, const uint32_t* __restrict__ region_base,
                                                    uint32_t* __restrict__ row_ptr, uint32_t* __restrict__ val_idx, uint64_t n,
                                                    const uint32_t* __restrict__ key_max) {
  __shared__ uint32_t bins[KRANGE + 1];
  __shared__ uint32_t part[256];
  __shared__ SortElem cache[LS_CACHE];
  const uint32_t r = blockIdx.x, ws = blockIdx.y, tid = threadIdx.x;
  const uint32_t shift = win_shift(key_max[ws]);
  const uint32_t KR = KRANGE >> shift;  // keys per range in this window
  const uint32_t lo = r * KR;
  const bool last = shift == 0 && r == NRANGE - 1;  // only the full-width layout reaches key 32768
  const uint32_t rbeg = region_base[ws * (NRANGE + 1) + r], rend = region_base[ws * (NRANGE + 1) + r + 1];
  const uint32_t len = rend - rbeg;
  const bool cached = len <= LS_CACHE;
  const SortElem* in = temp + (size_t)ws * n + rbeg;
  if (tid <= KRANGE) bins[tid] = 0;
  __syncthreads();
  for (uint32_t i0 = 0; i0 < len; i0 += 2048) {
    SortElem e[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t i = i0 + u * 256 + tid;
      e[u].key = 0xffffffffu;
      if (i < len) e[u] = in[i];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t i = i0 + u * 256 + tid;
      if (i < len) {
        atomicAdd(&bins[e[u].key - lo], 1u);
        if (cached) cache[i] = e[u];
      }
    }
  }
  __syncthreads();
  const uint32_t own = tid < KR ? bins[tid] : 0u;
  part[tid] = own;
  __syncthreads();
  for (uint32_t off = 1; off < 256; off <<= 1) {
    const uint32_t v = tid >= off ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t* rp = row_ptr + (size_t)ws * RP + lo;
  const uint32_t start = rbeg + part[tid] - own;
  __syncthreads();
  if (tid < KR) {
    bins[tid] = start;
    rp[tid] = start;
  }
  if (tid == KRANGE - 1 && last) {  // key 32768 and the end sentinel
    bins[KRANGE] = start + own;
    rp[KRANGE] = start + own;
    rp[KRANGE + 1] = rend;
  }
  if (shift) {  // narrowed ranges cover keys below NRANGE * KR only: every row above is empty and starts at the end
    const uint32_t covered = NRANGE * KR, total = region_base[ws * (NRANGE + 1) + NRANGE];
    const uint32_t per_block = (RP - covered + NRANGE - 1) / NRANGE;
    uint32_t* rp_w = row_ptr + (size_t)ws * RP;
    for (uint32_t j = tid; j < per_block; j += 256) {
      const uint32_t idx = covered + r * per_block + j;
      if (idx < RP) rp_w[idx] = total;
    }
  }
  __syncthreads();
  uint32_t* vi = val_idx + (size_t)ws * n;
  if (cached) {
    for (uint32_t i = tid; i < len; i += 256) {
      const SortElem e = cache[i];
      vi[atomicAdd(&bins[e.key - lo], 1u)] = e.idx_sign;
    }
  } else {
    for (uint32_t i0 = 0; i0 < len; i0 += 2048) {
      SortElem e[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t i = i0 + u * 256 + tid;
        if (i < len) e[u] = in[i];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t i = i0 + u * 256 + tid;
        if (i < len) vi[atomicAdd(&bins[e[u].key - lo], 1u)] = e[u].idx_sign;
      }
    }
  }
}

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

struct WorkItem {
  uint32_t row;  // ws * NB + t   (bucket index t <-> key t + 1)
  uint32_t seg;  // entries [seg * seglen, seg * seglen + seglen) of the row, see row_split
};

__device__ __forceinline__ uint32_t row_len(const uint32_t* __restrict__ row_ptr, uint32_t L, uint32_t row) {
  const uint32_t* rp = row_ptr + (size_t)(row >> L) * ((1u << L) + 2) + (row & ((1u << L) - 1));
  return rp[2] - rp[1];
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
                                                   uint32_t* __restrict__ split_rows) {
  __shared__ uint32_t lh[SEG_BINS];
  __shared__ uint32_t blk[4];  // split rows, overflow slots of this block; then their bases in the global lists
  const uint32_t tid = threadIdx.x, row = blockIdx.x * 1024 + tid;
  if (tid < SEG_BINS) lh[tid] = 0;
  if (tid < 2) blk[tid] = 0;
  __syncthreads();
  RowSplit sp = {1, 0, 0};
  uint32_t my_split = 0, my_ovf = 0;
  if (row < rows) {
    sp = row_split(row_len(row_ptr, L, row), SEG);
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

// One block (SEG_BINS <= 256): cursor[b] = number of items longer than b (descending order), total item count.
__global__ void __launch_bounds__(256) k_work_scan(const uint32_t* __restrict__ work_hist, uint32_t* __restrict__ cursor, uint32_t* __restrict__ total) {
  __shared__ uint32_t h[SEG_BINS];
  const uint32_t tid = threadIdx.x;
  if (tid < SEG_BINS) h[tid] = work_hist[tid];
  __syncthreads();
  if (tid < SEG_BINS) {
    uint32_t c = 0;
    for (uint32_t b = tid + 1; b < SEG_BINS; b++) c += h[b];
    cursor[tid] = c;
    if (tid == 0) *total = c + h[0];
  }
}

// Thread per row again: claims its slots in the sorted work list.
__global__ void __launch_bounds__(1024) k_work_scatter(const uint32_t* __restrict__ row_ptr, uint32_t L, uint32_t rows, uint32_t SEG, uint32_t* __restrict__ cursor,
                                                      WorkItem* __restrict__ work) {
  __shared__ uint32_t lh[SEG_BINS];
  __shared__ uint32_t lbase[SEG_BINS];
  const uint32_t tid = threadIdx.x, row = blockIdx.x * 1024 + tid;
  if (tid < SEG_BINS) lh[tid] = 0;
  __syncthreads();
  RowSplit sp = {0, 0, 0};
  uint32_t rank_full = 0, rank_last = 0;
  if (row < rows) {
    sp = row_split(row_len(row_ptr, L, row), SEG);
    if (sp.nseg > 1) rank_full = atomicAdd(&lh[sp.seglen], sp.nseg - 1);
    rank_last = atomicAdd(&lh[sp.lastlen], 1u);
  }
  __syncthreads();
  if (tid < SEG_BINS && lh[tid]) lbase[tid] = atomicAdd(&cursor[tid], lh[tid]);
  __syncthreads();
  if (row < rows) {
    for (uint32_t s = 0; s + 1 < sp.nseg; s++) work[lbase[sp.seglen] + rank_full + s] = WorkItem{row, s};
    work[lbase[sp.lastlen] + rank_last] = WorkItem{row, sp.nseg - 1};
  }
}

// One thread per work item.  OCC = waves per SIMD the register allocator must allow.
template <class CV, int OCC, class BP = CV>  // BP: where the input points come from (CV itself, or TeAffBase)
__global__ void __launch_bounds__(256, OCC) k_accumulate(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ val_idx,
                                                       const uint32_t* __restrict__ bases, uint32_t* __restrict__ buckets, uint64_t n,
                                                       const WorkItem* __restrict__ work, const uint32_t* __restrict__ work_total,
                                                       const uint32_t* __restrict__ row_ovf_base, uint32_t* __restrict__ ovf, uint32_t SEG,
                                                       int* __restrict__ err, const int* __restrict__ conv_err, uint32_t into, uint64_t table_stride, uint32_t L) {
  const uint32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v == 0 && *conv_err) atomicOr(err, *conv_err);  // the table holds a point its coordinate system cannot represent
  if (v >= *work_total) return;
  const WorkItem it = work[v];
  const uint32_t ws = it.row >> L, t = it.row & ((1u << L) - 1);
  const uint32_t* rp = row_ptr + (size_t)ws * ((1u << L) + 2);
  const uint32_t* vi = val_idx + (size_t)ws * n;
  bases += (size_t)ws * table_stride * BP::REC_WORDS;  // precomputed-window tables: window slot ws gathers from its own copy, [2^(16 ws)] P_i
  const uint32_t row_beg = rp[t + 1], row_end = rp[t + 2];
  const uint32_t seglen = row_split(row_end - row_beg, SEG).seglen;
  uint32_t k = row_beg + it.seg * seglen;
  const uint32_t end = (row_end - k > seglen) ? k + seglen : row_end;
  // into: the buckets already hold the sums of an earlier chunk of the same MSM (host-buffer entry point, chunked
  // upload): the row's first item continues from there.
  typename CV::Pt acc = (into && it.seg == 0) ? load_bucket<CV>(buckets, L, ws, t) : CV::identity();
  const bool start_fresh = !(into && it.seg == 0);
  bool bad = false;  // an exceptional pair of the twisted Edwards law (te377.hpp): sticky, the caller falls back
  if (k < end) {
    // Software pipeline: the index of entry k+2 and the record of entry k+1 are in flight while
    // entry k is added, so neither the val_idx -> bases address dependency nor the gather
    // latency stalls the wave.
    uint32_t e_cur = vi[k];
    uint32_t e_nxt = (k + 1 < end) ? vi[k + 1] : 0u;
    typename BP::Base cur = BP::load_base(bases, e_cur & 0x7fffffffu);
    bool more = true;  // cur / e_cur hold an entry that has not been added yet
    // One stage: start the gathers for the next two entries, add entry `cur`, rotate.  FIRST is a compile-time
    // switch so that the chain's first entry (BP::first: a copy, one product) is PEELED off the loop -- written as
    // `start_fresh ? first(cur) : madd(acc, cur)` inside the loop the compiler evaluated both sides every
    // iteration and selected: 8 products per iteration instead of 7 (9 instead of 8 with projective records; 2697
    // v_mad_u64_u32 in the loop body instead of 2360 -- tools/isa_mix.py, profiles/r02_final/isa_mix.json).
    auto stage = [&](auto first_tag) {
      constexpr bool FIRST = decltype(first_tag)::value;
      k++;
      more = k < end;
      typename BP::Base nxt = cur;
      uint32_t e_nn = 0u;
      if (more) {
        nxt = BP::load_base(bases, e_nxt & 0x7fffffffu);
        if (k + 1 < end) e_nn = vi[k + 1];
      }
      if constexpr (FIRST)
        acc = BP::first(cur, (e_cur >> 31) != 0);
      else
        acc = BP::madd(acc, cur, (e_cur >> 31) != 0);
      bad |= CV::is_bad(acc);
      cur = nxt;
      e_cur = e_nxt;
      e_nxt = e_nn;
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

// Thread per split row: bucket += its overflow partials (serial; 3 additions per row of the
// top window at n = 2^20, more only under heavy skew).
template <class CV>
__global__ void __launch_bounds__(256, 2) k_merge_split_rows(const uint32_t* __restrict__ row_ptr, uint32_t* __restrict__ buckets,
                                                             const uint32_t* __restrict__ counters, const uint32_t* __restrict__ split_rows,
                                                             const uint32_t* __restrict__ row_ovf_base, const uint32_t* __restrict__ ovf, uint32_t SEG,
                                                             int* __restrict__ err, uint32_t L) {
  const uint32_t count = counters[0];
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) {
    const uint32_t row = split_rows[i];
    const uint32_t len = row_len(row_ptr, L, row);
    const uint32_t nseg = row_split(len, SEG).nseg;
    const uint32_t ws = row >> L, t = row & ((1u << L) - 1);
    typename CV::Pt acc = load_bucket<CV>(buckets, L, ws, t);
    const uint32_t* src = ovf + (size_t)row_ovf_base[row] * CV::BKT_WORDS;
    bool bad = false;
    for (uint32_t s = 1; s < nseg; s++) {
      acc = CV::add(acc, load_record<CV>(src + (size_t)(s - 1) * CV::BKT_WORDS));
      bad |= CV::is_bad(acc);
    }
    if (bad) atomicOr(err, ERR_TE_MERGE);
    store_bucket<CV>(buckets, L, ws, t, acc);
  }
}

// Bucket reduction.  Per window the buckets B[0..NB) (B[i] has weight i+1) are reduced in
// place to  B[0] = sum of all buckets  and  B[2^b] = sum of the buckets whose index has bit b
// set  (b = 0..14), so that  sum_i (i+1) B[i] = B[0] + sum_b 2^b B[2^b]  -- the weights are
// applied by the host's Horner pass, not by per-thread double-and-add as in the reference
// (bpr.template.wgsl:125-173).  Bits are peeled from the top: level r (r = 0..14) folds the
// upper half of the running block [0, NB/2^r) onto its lower half,
//     B[k] += B[k + NB/2^(r+1)],            k < NB/2^(r+1),
// which leaves the untouched upper half [NB/2^(r+1), NB/2^r) = "index bit 14-r set" as a
// contiguous list that later levels halve the same way,
//     B[lo + k] += B[lo + k + NB/2^(r+1)],  lo = NB/2^(r'+1) for every earlier level r' < r.
// Adjacent lanes touch adjacent 256-byte bucket records.  Total work 2 NB
// additions per window -- the same as the reference's running sum (bpr.template.wgsl:99-107) --
// at depth 15 instead of 2 * 128 serial additions plus a 15-bit scalar multiplication.

// One level r of the reduction (see above): (r + 1) lists of NB/2^(r+1) pair-additions.
template <class CV>
__global__ void __launch_bounds__(256, 2) k_tree_step(uint32_t* __restrict__ buckets, uint32_t L, uint32_t r, uint32_t ops_per_window, int* __restrict__ err) {
  const uint32_t NB = 1u << L;  // buckets per window of this call (shadows the main path's constant)
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  const uint32_t ws = blockIdx.y;
  if (g >= ops_per_window) return;
  const uint32_t half = NB >> (r + 1);
  const uint32_t oi = g / half, kk = g % half;
  const uint32_t lo = oi == 0 ? 0u : (NB >> oi);  // list of level r' = oi - 1 starts at NB/2^(r'+1)
  const uint32_t x = lo + kk;
  const uint32_t y = x + half;
  typename CV::Pt a = load_bucket<CV>(buckets, L, ws, x);
  typename CV::Pt b = load_bucket<CV>(buckets, L, ws, y);
  // Empty buckets hold the identity exactly as identity() wrote it (the top window of a 253-bit scalar uses a seventh
  // of its buckets; small inputs leave most of every window empty): adjacent lanes see adjacent buckets, so whole
  // waves skip the addition.
  if (CV::is_stored_identity(b)) return;
  const typename CV::Pt sum = CV::is_stored_identity(a) ? b : CV::add(a, b);
  if (CV::is_bad(sum)) atomicOr(err, ERR_TE_TREE);
  store_bucket<CV>(buckets, L, ws, x, sum);
}

// Precomputed-window tables: bucket t of window slot ws += bucket t of slot ws + half (the table of slot ws already
// carries the weight 2^(16 ws), so the sixteen bucket sets simply add up); log2(16) launches leave the sum in slot 0.
template <class CV>
__global__ void __launch_bounds__(256, 2) k_fold_windows(uint32_t* __restrict__ buckets, uint32_t L, uint32_t half, int* __restrict__ err) {
  const uint32_t NB = 1u << L;
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;  // < half * NB
  const uint32_t ws = g / NB, t = g % NB;
  if (ws >= half) return;
  const typename CV::Pt b = load_bucket<CV>(buckets, L, ws + half, t);
  if (CV::is_stored_identity(b)) return;
  const typename CV::Pt a = load_bucket<CV>(buckets, L, ws, t);
  const typename CV::Pt sum = CV::is_stored_identity(a) ? b : CV::add(a, b);
  if (CV::is_bad(sum)) atomicOr(err, ERR_TE_TREE);
  store_bucket<CV>(buckets, L, ws, t, sum);
}

// ---- latency-bound levels: one XYZZ addition per QUAD of lanes ----
// From level ~5 on a reduction level has fewer additions than the chip has lanes, and its
// duration is one serial addition (14 field multiplications, ~14 us).  Here four adjacent
// lanes share one addition: every lane holds both operands, each lane performs ONE of the
// independent multiplications of a round (operands chosen by lane role, so all lanes run the
// same instruction stream) and the products are exchanged inside the quad with wave
// shuffles.  Four rounds instead of fourteen multiplications:
//   1: U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1        P = U2 - U1, R = S2 - S1
//   2: PP = P P    | RR = R R    | ZZ1 ZZ2      | ZZZ1 ZZZ2
//   3: PPP = P PP  | Q = U1 PP   | ZZ3 = (ZZ1 ZZ2) PP | -               X3 = RR - PPP - 2Q
//   4: R (Q - X3)  | S1 PPP      | -            | ZZZ3 = (ZZZ1 ZZZ2) PPP   Y3 = lane0 - lane1
// Identity operands and P = 0 (equal / opposite points) fall back to the generic addition on
// lane 0 of the quad.
// Broadcast lane K of every quad to its four lanes.  ds_bpermute (__shfl), not DPP quad_perm:
// the DPP form produced wrong sums inside the looped merge kernel on ROCm 7.2 (the shuffle form
// is bit-exact everywhere), and the crossbar cost is invisible next to a field multiplication.
template <int K, int NLIMB>
__device__ __forceinline__ Limbs<NLIMB> quad_bcast(const Limbs<NLIMB>& v) {
  Limbs<NLIMB> r;
#if defined(MSM377_QUAD_DPP)
  // Build-time variant for the root-cause hunt (tools/dpp_repro.sh): v_mov_b32_dpp quad_perm:[K,K,K,K].
  // MSM377_QUAD_DPP = 1: bound_ctrl off, `old` = the lane's own value; 2: bound_ctrl on (reads of disabled lanes give 0).
  constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);
#pragma unroll
  for (int j = 0; j < NLIMB; j++) r.l[j] = (uint32_t)__builtin_amdgcn_update_dpp((int)v.l[j], (int)v.l[j], ctrl, 0xf, 0xf, MSM377_QUAD_DPP == 2);
#else
  const int src = (int)(((threadIdx.x & 63u) & ~3u) | (uint32_t)K);
#pragma unroll
  for (int j = 0; j < NLIMB; j++) r.l[j] = (uint32_t)__shfl((int)v.l[j], src, 64);
#endif
  return r;
}
template <int NLIMB>
__device__ __forceinline__ Limbs<NLIMB> sel4(uint32_t q, const Limbs<NLIMB>& a0, const Limbs<NLIMB>& a1, const Limbs<NLIMB>& a2, const Limbs<NLIMB>& a3) {
  Limbs<NLIMB> r;
#pragma unroll
  for (int j = 0; j < NLIMB; j++) {
    const uint32_t lo = (q & 1) ? a1.l[j] : a0.l[j];
    const uint32_t hi = (q & 1) ? a3.l[j] : a2.l[j];
    r.l[j] = (q & 2) ? hi : lo;
  }
  return r;
}

// a + b computed by the four lanes of a quad (q = lane & 3); every lane passes the same a, b
// and every lane receives the full sum.
__device__ __forceinline__ G1XYZZ g1_add_quad(const G1XYZZ& a, const G1XYZZ& b, uint32_t q) {
  using K = G1Consts;  // lazy field forms and their bounds: g1_xyzz.hpp add_lz
  const Fp::El m1 = Fp::mul_lz(sel4(q, a.x, b.x, a.y, b.y), sel4(q, b.zz, a.zz, b.zzz, a.zzz));
  const Fp::El u1 = quad_bcast<0>(m1), u2 = quad_bcast<1>(m1), s1 = quad_bcast<2>(m1), s2 = quad_bcast<3>(m1);
  const Fp::El p = Fp::norm(Fp::add_kp_sub(u2, K::KP2, u1)), rr0 = Fp::norm(Fp::add_kp_sub(s2, K::KP2, s1));
  if (G1::is_identity(a) || G1::is_identity(b) || ((p.l[0] - 1u) < 3u && Fp::is_zero(Fp::canon(p)))) return G1::add(a, b);  // uniform inside the quad
  const Fp::El m2 = Fp::mul_lz(sel4(q, p, rr0, a.zz, a.zzz), sel4(q, p, rr0, b.zz, b.zzz));
  const Fp::El pp = quad_bcast<0>(m2), rsq = quad_bcast<1>(m2);
  const Fp::El m3 = Fp::mul_lz(sel4(q, p, u1, m2, p), pp);  // lane 3 idles on a copy of lane 0's product
  const Fp::El ppp = quad_bcast<0>(m3), qq = quad_bcast<1>(m3);
  G1XYZZ o;
  o.x = Fp::norm(Fp::add_kp_sub_sub2(rsq, K::KP4W3, ppp, qq));
  const Fp::El m4 = Fp::mul_lz(sel4(q, rr0, s1, rr0, m2), sel4(q, Fp::norm(Fp::add_kp_sub(qq, K::KP6, o.x)), ppp, ppp, ppp));
  // Y3 = lane 0 - lane 1 is a difference of two reduced products here (no fused form across lanes): bring it
  // back below p so that the stored Y keeps the < p + 2^354 invariant.
  o.y = Fp::canon(Fp::norm(Fp::add_kp_sub(quad_bcast<0>(m4), K::KP2, quad_bcast<1>(m4))));
  o.zz = quad_bcast<2>(m3);
  o.zzz = quad_bcast<3>(m4);
  return o;
}

__device__ __forceinline__ G1XYZZ add_quad(const G1XYZZ& a, const G1XYZZ& b, uint32_t q) { return g1_add_quad(a, b, q); }
__device__ __forceinline__ Fp::El coord4(uint32_t q, const G1XYZZ& p) { return sel4(q, p.x, p.y, p.zz, p.zzz); }

// The same for the twisted Edwards form (te377.hpp add): 3 rounds instead of 9 multiplications.
//   1: A = (Y1-X1)(Y2-X2) | B = (Y1+X1)(Y2+X2) | T1 T2 | Z1 Z2
//   2: C = 2d (T1 T2)   (every lane: no exchange needed)
//   3: X3 = E F | Y3 = H G | T3 = H E | Z3 = F G           -- the coordinate lane q stores
template <class F, class K>
__device__ __forceinline__ typename TeLazy<F, K>::Ext te_add_quad(const typename TeLazy<F, K>::Ext& a, const typename TeLazy<F, K>::Ext& b, uint32_t q) {
  using El = typename F::El;
  const El m1 = F::mul_lz(sel4(q, F::norm(F::add_kp_sub(a.y, K::KP2, a.x)), F::norm(F::add_lz(a.y, a.x)), a.t, a.z),
                          sel4(q, F::norm(F::add_kp_sub(b.y, K::KP2, b.x)), F::norm(F::add_lz(b.y, b.x)), b.t, b.z));
  const El pa = quad_bcast<0>(m1), pb = quad_bcast<1>(m1), tt = quad_bcast<2>(m1), zz = quad_bcast<3>(m1);
  const El c = F::mul_lz(tt, F::from_const(K::TE_2D));
  const El d = F::add_lz(zz, zz);
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
__device__ __forceinline__ Te377::Ext add_quad(const Te377::Ext& a, const Te377::Ext& b, uint32_t q) { return te_add_quad<Fp, G1Consts>(a, b, q); }
__device__ __forceinline__ Fp::El coord4(uint32_t q, const Te377::Ext& p) { return sel4(q, p.x, p.y, p.t, p.z); }
__device__ __forceinline__ EdLazy::Ext add_quad(const EdLazy::Ext& a, const EdLazy::Ext& b, uint32_t q) { return te_add_quad<Fq, EdConsts>(a, b, q); }
__device__ __forceinline__ Fq::El coord4(uint32_t q, const EdLazy::Ext& p) { return sel4(q, p.x, p.y, p.t, p.z); }

// One reduction level r (same index scheme as k_tree_step) with a quad per addition.
template <class CV>
__global__ void __launch_bounds__(256, 2) k_tree_step_quad(uint32_t* __restrict__ buckets, uint32_t L, uint32_t r, uint32_t ops_per_window, int* __restrict__ err) {
  const uint32_t NB = 1u << L;
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  const uint32_t g = gid >> 2, q = threadIdx.x & 3;
  const uint32_t ws = blockIdx.y;
  if (g >= ops_per_window) return;  // whole quads leave together
  const uint32_t half = NB >> (r + 1);
  const uint32_t oi = g / half, kk = g % half;
  const uint32_t lo = oi == 0 ? 0u : (NB >> oi);
  const uint32_t x = lo + kk, y = x + half;
  const typename CV::Pt sum = add_quad(load_bucket<CV>(buckets, L, ws, x), load_bucket<CV>(buckets, L, ws, y), q);
  if (CV::is_bad(sum)) atomicOr(err, ERR_TE_TREE);
  // each lane stores one coordinate
  const typename CV::F::El c = coord4(q, sum);
  store_coord<CV>(bucket_ptr<CV>(buckets, L, ws, x) + q * CV::COORD_WORDS, c.l);
}

// One coordinate of one of a window's 16 partial points (point 0 = B[0], point 1 + l = B[2^l]) in the HOST TAIL's format:
// re-based from the device's Montgomery radix 2^(29 NL) to 2^(32 NW32) and written as NW32 little-endian u32 words, so
// the host does no conversion multiplications.  Used by k_gather_partials and by k_reduce_tail's own output stage.
template <class CV>
__device__ __forceinline__ void pack_partial(const uint32_t* __restrict__ buckets, uint32_t L, uint32_t ws, uint32_t pt, uint32_t coord,
                                             uint32_t* __restrict__ out, uint32_t* __restrict__ host_out) {
  const uint32_t x = pt == 0 ? 0u : (1u << (pt - 1));
  typename CV::F::El v;
#pragma unroll
  for (uint32_t j = 0; j < CV::NL; j++) v.l[j] = bucket_ptr<CV>(buckets, L, ws, x)[coord * CV::COORD_WORDS + j];
  v = CV::F::mul(v, CV::to64());
  uint32_t w[CV::NW32];
  CV::F::template to_words<CV::NW32>(v, w);
  if (pt == 0 && coord == 0) w[CV::NW32 - 1] |= CV::RECORD_TAG;  // the record names its coordinate system (values are < 2^377: the bit is free)
  const size_t at = ((size_t)(ws * MSM377_G1_PARTIAL_POINTS + pt) * 4 + coord) * CV::NW32;
#pragma unroll
  for (uint32_t j = 0; j < CV::NW32; j++) out[at + j] = w[j];
  if (host_out) {
#pragma unroll
    for (uint32_t j = 0; j < CV::NW32; j++) host_out[at + j] = w[j];
  }
}
// The block that finishes last (a device-memory counter) hands the call over to the host: error word, then the sequence
// number the host is polling for (wait_zero_copy_out).  Call with every store of the block issued; all threads.
__device__ __forceinline__ void publish_to_host(uint32_t blocks, uint32_t* host_flag, uint32_t* dev_count, const int* d_err, uint32_t seq) {
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(dev_count, 1u) == blocks - 1) {  // every other block's records are on their way
    __threadfence_system();
    __hip_atomic_store(&host_flag[1], (uint32_t)__hip_atomic_load(d_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __hip_atomic_store(&host_flag[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    *dev_count = 0u;  // for the next call (stream order)
  }
}

// The last levels of the reduction in ONE launch.  After level L - 1 every window holds L lists of M = NB >> L buckets
// (list j, created at level j, starts at bucket NB >> (j + 1)) plus the running block [0, M).  Nothing connects the
// lists any more: each one only has to be summed, and only the running block keeps spawning new lists (levels L..14).
// So a workgroup of 128 lane quads takes one list -- or the running block with everything it spawns -- through all
// its remaining levels with a workgroup barrier between levels, instead of one kernel launch per level for the whole
// grid: (L + 1) workgroups per window, 15 - L barriers each.  Every level of a list halves it in place exactly as
// k_tree_step_quad does (same bucket pairs, so the partial records come out the same).
constexpr uint32_t TAIL_THREADS = 512;  // 128 lane quads; 2 waves per SIMD, so an addition may use 256 VGPRs
// (An output stage of its own -- every workgroup packing the partial points it ends up owning, the last one publishing
// to the host, no k_gather_partials launch -- was built and dropped: the pack is one more field product of latency
// at the end of every workgroup, 0.274 -> 0.282 ms at 2^12, 0.341 -> 0.348 at 2^14, 2.60 -> 2.61 at 2^20.)
template <class CV>
__global__ void __launch_bounds__(TAIL_THREADS, 1) k_reduce_tail(uint32_t* __restrict__ buckets, uint32_t L, uint32_t first, int* __restrict__ err) {
  const uint32_t NB = 1u << L;
  const uint32_t ws = blockIdx.y, job = blockIdx.x;  // job < first: list `job`; job == first: the running block
  const uint32_t q = threadIdx.x & 3, quad = threadIdx.x >> 2;
  bool bad = false;
  for (uint32_t r = first; r < L; r++) {
    const uint32_t half = NB >> (r + 1);
    // lists this workgroup halves at level r: its own one, or the running block (list index 0) and the lists the
    // block has spawned since level L (created at levels L .. r - 1: list indices L + 1 .. r in k_tree_step's scheme)
    const uint32_t nlists = job < first ? 1u : 1u + (r - first);
    for (uint32_t op = quad; op < nlists * half; op += TAIL_THREADS / 4) {
      const uint32_t li = op / half, kk = op % half;
      const uint32_t oi = job < first ? job + 1 : (li == 0 ? 0u : first + li);
      const uint32_t lo = oi == 0 ? 0u : (NB >> oi);
      const uint32_t x = lo + kk, y = x + half;
      const typename CV::Pt sum = add_quad(load_bucket<CV>(buckets, L, ws, x), load_bucket<CV>(buckets, L, ws, y), q);
      bad |= CV::is_bad(sum);
      store_coord<CV>(bucket_ptr<CV>(buckets, L, ws, x) + q * CV::COORD_WORDS, coord4(q, sum).l);
    }
    __syncthreads();  // workgroup-scope fence + barrier: the next level reads what this one wrote
  }
  if (bad) atomicOr(err, ERR_TE_TREE);
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
  const uint32_t* rp = row_ptr + (size_t)ws * ((1u << L) + 2);
  const uint32_t* vi = val_idx + (size_t)ws * n;
  const uint32_t row_beg = rp[t + 1], row_end = rp[t + 2];
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
                                                                  int* __restrict__ err, uint32_t L) {
  const uint32_t count = counters[0];
  const uint32_t q = threadIdx.x & 3;
  for (uint32_t i = (blockIdx.x * 256 + threadIdx.x) >> 2; i < count; i += gridDim.x * 64) {
    const uint32_t row = split_rows[i];
    const uint32_t len = row_len(row_ptr, L, row);
    const uint32_t nseg = row_split(len, SEG).nseg;
    const uint32_t ws = row >> L, t = row & ((1u << L) - 1);
    typename CV::Pt acc = load_bucket<CV>(buckets, L, ws, t);
    const uint32_t* src = ovf + (size_t)row_ovf_base[row] * CV::BKT_WORDS;
    typename CV::Pt nxt = load_record<CV>(src);
    bool bad = false;
    for (uint32_t s = 1; s < nseg; s++) {
      const typename CV::Pt cur = nxt;
      if (s + 1 < nseg) nxt = load_record<CV>(src + (size_t)s * CV::BKT_WORDS);
      acc = add_quad(acc, cur, q);
      bad |= CV::is_bad(acc);
    }
    if (bad) atomicOr(err, ERR_TE_MERGE);
    const typename CV::F::El c = coord4(q, acc);
    store_coord<CV>(bucket_ptr<CV>(buckets, L, ws, t) + q * CV::COORD_WORDS, c.l);
  }
}

// Pack the 16 partial points of every window slot (point 0 = B[0], point 1 + l = B[2^l]) in the
// HOST TAIL's format: each coordinate re-based from the device's Montgomery radix 2^(29 NL) to
// 2^(32 NW32) and written as NW32 little-endian u32 words, so the host does no conversion
// multiplications.  One thread per (window slot, point, coordinate).
//
// host_out != nullptr: the records ALSO go straight into the caller's pinned host buffer (zero-copy stores over PCIe,
// 50-80 KB), and the block that finishes last copies the call's error word next to a sequence number the host is
// polling for (host_flag[0] = seq, host_flag[1] = error word) -- instead of two hipMemcpyAsync and an event, whose
// copy-engine hand-over and completion signal cost ~25 us at the very end of every MSM.
template <class CV>
__global__ void __launch_bounds__(64) k_gather_partials(const uint32_t* __restrict__ buckets, uint32_t* __restrict__ out, uint32_t wc, uint32_t L,
                                                      uint32_t* __restrict__ host_out = nullptr, uint32_t* host_flag = nullptr,
                                                      uint32_t* dev_count = nullptr, const int* d_err = nullptr, uint32_t seq = 0) {
  const uint32_t g = blockIdx.x * 64 + threadIdx.x;
  const uint32_t coord = g & 3, pt = (g >> 2) % MSM377_G1_PARTIAL_POINTS, ws = g / (4 * MSM377_G1_PARTIAL_POINTS);
  // narrow windows have fewer bit planes; the host tail never reads the unused points
  if (g < wc * MSM377_G1_PARTIAL_POINTS * 4 && pt <= L) pack_partial<CV>(buckets, L, ws, pt, coord, out, host_out);
  if (host_out) publish_to_host(gridDim.x, host_flag, dev_count, d_err, seq);  // a kernel argument: uniform
}

// Synthetic bases: P_i = [a_i]G with a_i the (i+1)-th SplitMix64(seed) output, wire format.
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t i) {
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256, 2) k_generate_bases(uint64_t seed, uint64_t n, uint32_t* __restrict__ out_raw) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t a = splitmix64_at(seed, i);
  if (a == 0) a = 1;
  G1Affine gen;
  gen.x = Fp::from_const(G1Consts::GEN_X);
  gen.y = Fp::from_const(G1Consts::GEN_Y);
  G1XYZZ acc = g1_identity();
#pragma unroll 1
  for (int bit = 63; bit >= 0; bit--) {
    acc = g1_dbl(acc);
    if ((a >> bit) & 1) acc = g1_madd(acc, gen);
  }
  // affine: x = X * (ZZ/ZZZ)^2, y = Y / ZZZ
  Fp::El i3 = Fp::one();
#pragma unroll 1
  for (int b = G1Consts::PM2_NW * 32 - 1; b >= 0; b--) {
    i3 = Fp::sqr(i3);
    if ((G1Consts::PM2_W[b >> 5] >> (b & 31)) & 1u) i3 = Fp::mul(i3, acc.zzz);
  }
  Fp::El tt = Fp::mul(i3, acc.zz);
  Fp::El x = Fp::from_mont(Fp::mul(acc.x, Fp::sqr(tt)));
  Fp::El y = Fp::from_mont(Fp::mul(acc.y, i3));
  uint32_t w[24];
  Fp::to_words<12>(x, w);
  Fp::to_words<12>(y, w + 12);
  uint4* dst = reinterpret_cast<uint4*>(out_raw + i * 24);
#pragma unroll
  for (int k = 0; k < 6; k++) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}

// Edwards twin: P_i = [a_i]G_ed (generator: src/reference/utils/FieldMath.ts:108-109), 64-byte wire records.
__global__ void __launch_bounds__(256, 2) k_generate_bases_ed(uint64_t seed, uint64_t n, uint32_t* __restrict__ out_raw) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t a = splitmix64_at(seed, i);
  if (a == 0) a = 1;
  const Ed::Base gen = Ed::make_base(Fq::from_const(EdConsts::GEN_X), Fq::from_const(EdConsts::GEN_Y));
  Ed::Ext acc = Ed::identity();
#pragma unroll 1
  for (int bit = 63; bit >= 0; bit--) {
    acc = Ed::dbl(acc);
    if ((a >> bit) & 1) acc = Ed::madd(acc, gen);
  }
  Fq::El zi = Fq::one();
#pragma unroll 1
  for (int b = EdConsts::PM2_NW * 32 - 1; b >= 0; b--) {
    zi = Fq::sqr(zi);
    if ((EdConsts::PM2_W[b >> 5] >> (b & 31)) & 1u) zi = Fq::mul(zi, acc.z);
  }
  Fq::El x = Fq::from_mont(Fq::mul(acc.x, zi));
  Fq::El y = Fq::from_mont(Fq::mul(acc.y, zi));
  uint32_t w[16];
  Fq::to_words<8>(x, w);
  Fq::to_words<8>(y, w + 8);
  uint4* dst = reinterpret_cast<uint4*>(out_raw + i * 16);
#pragma unroll
  for (int k = 0; k < 4; k++) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}

}  // namespace

// ------------------------------------------------------------------------- context ----

// Three helper threads for the host tail of a single MSM (the Horner pass is 0.18 ms of serial field arithmetic --
// 5 % of a 2^20 MSM -- and it is the one stage nothing else can hide).  Workers sleep on a condition variable
// between calls; a call publishes up to three jobs and collects them in the order it needs them.
struct TailPool {
  static constexpr int WORKERS = 7;
  // Workers asleep on their condition variable take 20-60 us to come back -- as long as their whole job (a piece of the
  // Horner chain is ~55 us) -- so a call that will need them ARMS the pool (prewake) once its accumulation kernel has
  // finished: the first `count` workers wake up while the GPU reduces the buckets (0.1-0.3 ms) and poll for their
  // job until the deadline, then go back to sleep.  Costs that many spinning cores for the length of the bucket
  // reduction, at most `spin_us` per call (MSM377_TAIL_SPIN_US, 0 = never spin).
  // Every worker has its own slot (job, generation counters, mutex, condition variable) on its own cache lines: posting
  // a job to a polling worker is two stores, no lock and no system call; only a sleeping worker is notified.
  struct alignas(128) Slot {
    std::function<void()> job;
    std::atomic<uint64_t> posted{0};  // generation of the last job handed to this worker
    std::atomic<uint64_t> done{0};    // generation it has finished
    std::atomic<bool> asleep{false};
    std::mutex mu;
    std::condition_variable cv;
    std::thread th;
  };
  Slot slot[WORKERS];
  std::atomic<int64_t> armed_until_ns{0};
  std::atomic<int> armed_count{0};
  std::atomic<bool> stop{false};
  bool started = false;
  static int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  bool armed(int k) const { return k < armed_count.load(std::memory_order_relaxed) && now_ns() < armed_until_ns.load(std::memory_order_relaxed); }
  // The logical CPUs of the NUMA node the calling thread runs on (false: unknown).  The workers are kept on that node:
  // on a two-socket host a worker on the far socket reads the records and its job across the socket link.  (Why: the
  // tail stage was bimodal from one context to the next on some boxes, 0.077 / 0.112 ms; an A/B of 8 contexts each
  // way on another box showed 0.075-0.079 for all of them, so the cause is a hypothesis, not a measurement.)
  static bool local_node_cpus(cpu_set_t* set) {
    const int cpu = sched_getcpu();
    if (cpu < 0) return false;
    for (int node = 0; node < 64; node++) {
      char path[96];
      snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
      FILE* f = fopen(path, "r");
      if (!f) break;
      char buf[4096];
      const bool got = fgets(buf, sizeof buf, f) != nullptr;
      fclose(f);
      if (!got) continue;
      CPU_ZERO(set);
      bool mine = false;
      for (char* p = buf; *p;) {  // "0-63,128-191"
        char* e;
        const long a = strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) CPU_SET((int)c, set);
        mine |= cpu >= a && cpu <= b;
        p = *e == ',' ? e + 1 : e;
        if (*e != ',') break;
      }
      if (mine) return true;
    }
    return false;
  }
  bool numa_local = true;  // MSM377_TAIL_NUMA=0: leave the workers where the scheduler puts them
  void start() {
    if (started) return;
    started = true;
    cpu_set_t node_cpus;
    const bool pin = numa_local && local_node_cpus(&node_cpus);
    for (int k = 0; k < WORKERS; k++) {
      slot[k].th = std::thread([this, k] {
        Slot& me = slot[k];
        uint64_t seen = 0;
        for (;;) {
          while (me.posted.load() == seen) {  // (sequentially consistent against post(): one of the two sides sees the other)
            if (stop.load()) return;
            if (armed(k)) {
              __builtin_ia32_pause();
              continue;
            }
            std::unique_lock<std::mutex> lk(me.mu);
            me.asleep.store(true);
            me.cv.wait(lk, [&] { return stop.load() || me.posted.load() != seen || armed(k); });
            me.asleep.store(false);
          }
          seen = me.posted.load();
          me.job();
          me.done.store(seen, std::memory_order_release);
        }
      });
      if (pin) (void)pthread_setaffinity_np(slot[k].th.native_handle(), sizeof node_cpus, &node_cpus);
    }
  }
  void wake(Slot& sl) {
    if (!sl.asleep.load()) return;
    std::lock_guard<std::mutex> lk(sl.mu);  // with the lock: a worker between its predicate and its sleep must not miss this
    sl.cv.notify_one();
  }
  void prewake(int64_t spin_us, int count = WORKERS) {
    if (spin_us <= 0 || count <= 0) return;
    start();
    armed_count.store(std::min(count, (int)WORKERS));
    armed_until_ns.store(now_ns() + spin_us * 1000);
    for (int k = 0; k < std::min(count, (int)WORKERS); k++) wake(slot[k]);
  }
  void disarm() { armed_until_ns.store(0, std::memory_order_relaxed); }
  // The previous job of worker k must have been waited for (wait(k)): the slot's job is not read any more.
  void post(int k, std::function<void()> f) {
    Slot& sl = slot[k];
    sl.job = std::move(f);
    sl.posted.fetch_add(1);
    wake(sl);
  }
  void wait(int k) {  // short: the job is a few tens of microseconds
    const uint64_t want = slot[k].posted.load(std::memory_order_acquire);
    while (slot[k].done.load(std::memory_order_acquire) != want) __builtin_ia32_pause();
  }
  ~TailPool() {
    if (!started) return;
    stop.store(true);
    armed_until_ns.store(0);
    for (Slot& sl : slot) {
      {
        std::lock_guard<std::mutex> lk(sl.mu);
        sl.cv.notify_one();
      }
      if (sl.th.joinable()) sl.th.join();
    }
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
  uint32_t* d_aff_stash = nullptr;    // cap x 52 words: N1, N2, Z, running product per point (k_affine_up -> k_affine_down)
  uint32_t* d_aff_trees = nullptr;    // one product tree (2 x 256 nodes x 13 words) per AFF_BLOCK_POINTS points
  uint32_t* h_aff_prod = nullptr;     // pinned + coherent host memory the kernels access in place (dm_* = its device address)
  uint32_t* h_aff_inv = nullptr;
  uint32_t* h_aff_flag = nullptr;     // workgroups of k_affine_up that have delivered their product
  uint32_t *dm_aff_prod = nullptr, *dm_aff_inv = nullptr, *dm_aff_flag = nullptr;
  uint32_t* d_aff_count = nullptr;    // workgroups of k_affine_up that have delivered (device memory; the last one resets it)
  hipEvent_t aff_up_done = nullptr;
  hipEvent_t sort_done = nullptr;     // recorded behind k_local_sort of the current call (main stream)
  int aff_down_after_sort = 0;        // MSM377_AFF_AFTER_SORT=1: k_affine_down waits for the sort (see affine_convert_finish)
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
  int h2d_threads = 4;                // copy workers of the host-buffer entry points (MSM377_H2D_THREADS, 1..8)
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
  // First reduction level run with one addition per lane quad.  0 = automatic: the first level whose 4 lanes x additions
  // x windows fit one wave per SIMD (65536 lanes) -- level 7 for 16 windows (measured: 18-24 -> 13-18 us per level from
  // there on, slower before), 6 for 8, 4 for the 2 windows a rank of an 8-GPU run owns.  MSM377_COOP_FROM forces it (15 = never).
  uint32_t coop_from = 0;
  // MSM377_COOP_THREADS: a tree level runs one lane quad per addition once that takes at most this many threads.  65536 / 131072 /
  // 262144 make no difference on the main path (2^20: 2.74 ms each); on the narrow path 131072 moves its levels 0-2 to quads.
  uint32_t coop_threads = 131072;
  // MSM377_NARROW_TAIL_FROM: tail_from of the narrow-window path (2048 buckets per window).  Reduce stage at 2^12 with
  // 7 / 5 / 4 / 3 / 2: 0.106 / 0.099 / 0.096 / 0.101 / 0.122 ms (profiles/r02_final/ab_narrow_tree.txt).
  uint32_t narrow_tail_from = 4;
  uint32_t narrow_seg = NARROW_SEG;  // MSM377_NARROW_SEG (>= NARROW_SEG: the buffers are sized for that)
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
  bool te_affine_table = true;  // MSM377_TE_AFFINE_TABLE=0: resident Edwards tables stay projective (A/B knob)
  int g1_form = 1;         // G1 full-MSM entry points: 1 = twisted Edwards form (te377.hpp, default), 0 = Weierstrass XYZZ (MSM377_G1_FORM)
  bool last_glv = false;
  bool merge_full_grid = true;  // MSM377_MERGE_FULL_GRID=0: fixed 64-workgroup sweep of the split-row list (A/B knob)
  bool merge_quad = true;  // MSM377_MERGE_QUAD=0: thread-per-row merge of split rows
  uint32_t seg_plain = 0, seg_glv = 0;  // MSM377_SEG_PLAIN / MSM377_SEG_GLV: force the work-item length (SEG_MIN..SEG_MAX), 0 = auto_seg()
  hipEvent_t ev[2][MSM377_NUM_STAGES][2] = {};  // [part][stage][begin, end]
  hipStream_t stream3 = nullptr;      // second part of a pipelined call (enqueue_windows)
  hipEvent_t part_fork = nullptr, part_join = nullptr, acc_done = nullptr;
  uint64_t upload_chunk_min = 1ull << 18;  // msm377_g1_msm: inputs of at least this many points upload and run as two chunks (MSM377_UPLOAD_CHUNK_MIN)
  uint32_t upload_chunks = 4;              // chunks of the host-buffer upload (MSM377_UPLOAD_CHUNKS, 2..8): 2: 5.07, 3: 4.89, 4-6: 4.70, 8: 4.95 ms at 2^20
  uint32_t upload_split_pct = 30;          // share of the points in the first chunk (MSM377_UPLOAD_SPLIT, 10..90)
  std::function<int()> before_accumulate;  // host-buffer entry point: joins the point upload and launches the base conversion (enqueue_part)
  bool key_shift = true;              // MSM377_KEY_SHIFT=0: full-width key ranges in every window (A/B knob)
  TailPool tail_pool;
  int tail_threads = 6;               // MSM377_TAIL_THREADS: threads of the host tail (1..8, tail_horner_mt)
  // MSM377_TAIL_SPIN_US: how long at most the tail workers poll for their job after a call has armed them (TailPool;
  // 0 = they sleep until the job is posted).  Tail stage at 2^20, interleaved (tools/ab_knobs.py): one thread 0.140 ms,
  // six sleeping workers 0.124, six polling ones 0.089.  (Round 2 first measured no difference: the per-thread
  // exceptional-case flags shared a cache line then and the threads fought over it -- TeChecked is padded now.)
  int64_t tail_spin_us = 1000;
  bool tail_trace = false;  // MSM377_TAIL_TRACE=1
  int pipeline_parts = 1;             // MSM377_PIPELINE=2: two parts on two streams.  Measured: no gain at 2^20 / 2^21 (3.19 vs 3.17 ms), 2 % at 2^22 -- the accumulation kernel owns every VGPR of the chip, so kernels of the other part cannot become co-resident
  uint32_t last_parts = 1;
  double stage_ms[MSM377_NUM_STAGES] = {};
  int last_products = 0;        // field products per bucket addition of the last accumulation launch (bench.py's int32-mad roof)
  // Inputs of at most this many points run the narrow-window path (11-bit windows: 23 x 2048 buckets instead of
  // 16 x 32768; MSM377_NARROW_MAX, 0 = never).  Interleaved A/B, 16-bit / narrow ms per MSM (tools/ab_knobs.py):
  // 2^10 0.64 / 0.46, 2^13 0.67 / 0.53, 2^14 0.68 / 0.51, 2^15 0.73 / 0.60, 2^16 0.74 / 0.71 (first version); at the end of
  // round 2: 2^16 0.605 / 0.56 (its bucket reduction 0.28 / 0.10 ms, its accumulation kernel 0.14 / 0.18), hence 2^16.
  uint64_t narrow_max_points = 1ull << 16;
  uint64_t fallback_count = 0;  // reruns on the Weierstrass path after an exceptional case of the Edwards law
  uint32_t fallback_mask = 0;   // MSM377_FB_* bits of the last one
};

namespace {

bool hip_ok(msm377_ctx* ctx, hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  if (ctx) ctx->err = std::string(what) + ": " + hipGetErrorString(e);
  return false;
}
void note_fallback(msm377_ctx* ctx, uint32_t mask) {
  ctx->fallback_count++;
  ctx->fallback_mask = mask;
}
#define HIP_TRY(ctx, call)                            \
  do {                                                \
    if (!hip_ok((ctx), (call), #call)) return MSM377_EHIP; \
  } while (0)

// Pageable host memory -> device through a pinned staging buffer: four workers copy ~4 MB pieces
// into it and queue the DMA of each piece on their own stream, so the CPU copy of one piece
// overlaps the DMA of the others.  Measured on the MI355X box for 160 MB: 4.2 ms, against 28 ms
// for a first hipMemcpy from fresh pageable pages (4.4 ms once the runtime has pinned them) and
// 3.3 + 2.9 ms for hipHostRegister + copy.  Returns when the data is on the device.
int h2d_staged(msm377_ctx* ctx, void* d_dst, const uint8_t* src, size_t bytes, size_t stage_off) {
  constexpr int NT_MAX = 8;
  const int NT = ctx->h2d_threads;
  constexpr size_t SMALL = 8u << 20, PIECE = 4u << 20;
  if (bytes < SMALL) {  // not worth four threads
    HIP_TRY(ctx, hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return MSM377_OK;
  }
  if (!ctx->h_stage) {
    if (hipHostMalloc((void**)&ctx->h_stage, (size_t)ctx->cap * 128) != hipSuccess) {
      ctx->h_stage = nullptr;
      HIP_TRY(ctx, hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));  // fall back to the runtime's pageable path
      return MSM377_OK;
    }
    for (int t = 0; t < NT_MAX; t++) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream[t], hipStreamNonBlocking));
  }
  // Pieces of about 4 MB, their number a multiple of the worker count: every worker copies the same amount (with
  // fixed 8 MB pieces a 48 MB upload took as long as a 64 MB one), and a worker's host copy of piece k+1 overlaps
  // the DMA of piece k.
  size_t npieces = (bytes + PIECE - 1) / PIECE;
  npieces = (npieces + NT - 1) / NT * NT;
  const size_t piece = ((bytes + npieces - 1) / npieces + 4095) & ~(size_t)4095;
  uint8_t* stage = ctx->h_stage + stage_off;
  hipError_t errs[NT_MAX];
  std::thread workers[NT_MAX];
  const int device = ctx->device;
  for (int t = 0; t < NT; t++) {
    errs[t] = hipSuccess;
    workers[t] = std::thread([=, &errs] {
      hipError_t e = hipSetDevice(device);
      for (size_t c = t; c < npieces && e == hipSuccess; c += NT) {
        const size_t off = c * piece;
        if (off >= bytes) break;
        const size_t len = (bytes - off < piece) ? bytes - off : piece;
        memcpy(stage + off, src + off, len);
        e = hipMemcpyAsync((uint8_t*)d_dst + off, stage + off, len, hipMemcpyHostToDevice, ctx->copy_stream[t]);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_stream[t]);
      errs[t] = e;
    });
  }
  for (int t = 0; t < NT; t++) workers[t].join();
  for (int t = 0; t < NT; t++) HIP_TRY(ctx, errs[t]);
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
  // k_accumulate waits for `bases_ready`.  Every entry point ends with a host-side wait for the
  // main stream, so the previous call's readers of d_bases are done.
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

// ---- batched affine conversion, host side (kernels: k_affine_up / k_affine_down) ----
// Inverses of block products [b0, b1) by Montgomery's trick with one Fermat inversion; results re-based to the device's
// Montgomery radix (G1Consts64::TO29) as 12 plain words each.
void invert_block_products(msm377_ctx* ctx, uint32_t b0, uint32_t b1) {
  if (b0 >= b1) return;
  Fp64::El* pre = ctx->aff_scratch.data();
  Fp64::El acc = Fp64::one();
  for (uint32_t b = b0; b < b1; b++) {
    pre[b] = acc;
    acc = Fp64::mul(acc, Fp64::from_words32(ctx->h_aff_prod + (size_t)b * 12));
  }
  Fp64::El inv = Fp64::inv(acc);
  const Fp64::El to29 = Fp64::from_const(G1Consts64::TO29);
  for (uint32_t b = b1; b-- > b0;) {
    const Fp64::El mine = Fp64::mul(Fp64::mul(inv, pre[b]), to29);  // a plain integer now: x 2^406 mod p
    inv = Fp64::mul(inv, Fp64::from_words32(ctx->h_aff_prod + (size_t)b * 12));
    words_from_fp64(mine, ctx->h_aff_inv + (size_t)b * 12);
  }
}

inline uint32_t affine_blocks(uint64_t n) { return (uint32_t)((n + AFF_BLOCK_POINTS - 1) / AFF_BLOCK_POINTS); }

// Phase 1, queued on the side stream: products up to one value per workgroup, delivered into pinned host memory.
int affine_convert_begin(msm377_ctx* ctx, const uint32_t* d_raw, uint64_t n, const uint32_t* prev_window_records = nullptr, bool clear_err = true) {
  if (n == 0) return MSM377_OK;
  const uint32_t nblk = affine_blocks(n);
  if (ctx->timing == 1) (void)hipEventRecord(ctx->ev[0][MSM377_STAGE_CONVERT][0], ctx->stream2);
  __atomic_store_n(ctx->h_aff_flag, 0u, __ATOMIC_RELEASE);
  if (clear_err) hipLaunchKernelGGL(k_clear_words, dim3(1), dim3(256), 0, ctx->stream2, (uint32_t*)(ctx->d_err + 2), 1u, (uint32_t*)nullptr, 0u);
  if (prev_window_records)
    hipLaunchKernelGGL(k_affine_up<AffDoublingSource>, dim3(nblk), dim3(AFF_THREADS), 0, ctx->stream2, AffDoublingSource{prev_window_records}, n, ctx->d_aff_stash,
                       ctx->d_aff_trees, ctx->dm_aff_prod, ctx->dm_aff_flag, ctx->d_aff_count, ctx->d_err + 2);
  else
    hipLaunchKernelGGL(k_affine_up<AffWireSource>, dim3(nblk), dim3(AFF_THREADS), 0, ctx->stream2, AffWireSource{d_raw}, n, ctx->d_aff_stash, ctx->d_aff_trees,
                       ctx->dm_aff_prod, ctx->dm_aff_flag, ctx->d_aff_count, ctx->d_err + 2);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ctx->aff_up_done, ctx->stream2));
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
  if (nblk >= 32 && ctx->tail_threads > 1) {
    TailPool& pool = ctx->tail_pool;
    pool.start();
    const int parts = std::min(ctx->tail_threads, TailPool::WORKERS + 1);
    const uint32_t per = (nblk + parts - 1) / parts;
    for (int k = 0; k + 1 < parts; k++)
      pool.post(k, [ctx, k, per, nblk] { invert_block_products(ctx, std::min(nblk, (uint32_t)(k + 1) * per), std::min(nblk, (uint32_t)(k + 2) * per)); });
    invert_block_products(ctx, 0, std::min(nblk, per));
    for (int k = 0; k + 1 < parts; k++) pool.wait(k);
  } else {
    invert_block_products(ctx, 0, nblk);
  }
  // k_affine_down beside k_local_sort: each stretches the other (they fight over the memory system), and when the way
  // down was first built letting it wait for the sort was the faster order.  At the end of round 2 -- shorter front
  // end, zero-copy products, polled flag -- the interleaved A/B says the opposite: 2^20 2.62 -> 2.59 ms, 2^21 5.08 ->
  // 5.02, 2^22 10.39 -> 10.21 without the wait (two contexts each way), so the way down starts as soon as the host
  // has inverted the block products.  MSM377_AFF_AFTER_SORT=1 restores the wait.
  if (behind_sort && ctx->aff_down_after_sort) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, ctx->sort_done, 0));
  hipLaunchKernelGGL(k_affine_down, dim3(nblk), dim3(AFF_THREADS), 0, ctx->stream2, n, ctx->d_aff_stash, ctx->d_aff_trees, ctx->dm_aff_inv, d_records_out);
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

constexpr uint32_t META_BLOCK_WORDS = 2 * SEG_BINS + 4 + MSM377_NUM_WINDOWS;  // per pipeline part: work-list counters + key_max words
constexpr uint64_t PIPELINE_MIN_ENTRIES = 1ull << 21;  // (windows x points) below which a call stays in one part

constexpr size_t SLOT_WORDS = (size_t)MAX_WINDOW_SLOTS * MSM377_G1_PARTIAL_POINTS * MSM377_G1_POINT_WORDS;  // per double-buffer slot

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
  const uint32_t L = ph.bucket_log, NB = 1u << L, RP = NB + 2;  // this call's bucket geometry (shadows the main path's constants)
  const bool narrow = ph.cbits != MSM377_WINDOW_BITS;
  static_assert((uint64_t)NARROW_WINDOWS * SMALL_SORT_MAX / NARROW_SEG + NARROW_WINDOWS * (1u << NARROW_LOG) <= (uint64_t)MSM377_NUM_WINDOWS * 32768, "narrow work items fit the work-item buffer");
  // per launch: each part must fill the GPU on its own.  Narrow windows: a small input is all latency -- a work item is
  // a serial chain of ~10 us additions -- so its chains are cut at 8 entries (the buffers, sized for 16 windows of
  // 2^15 rows plus entries / SEG_MIN items, hold the 23 x 2^11 rows and 23 n / 8 items of an input this small easily).
  const uint32_t SEG = (narrow && !ctx->seg_plain) ? ctx->narrow_seg : auto_seg(ctx, (uint64_t)wc * n, glv);
  uint16_t* digits = ctx->d_digits + (size_t)pv.ws0 * n;
  uint32_t* range_counts = ctx->d_range_counts + (size_t)part * NRANGE * (MAX_SORT_BLOCKS / 2);
  uint32_t* region_base = ctx->d_region_base + (size_t)pv.ws0 * (NRANGE + 1);
  SortElem* sort_temp = ctx->d_sort_temp + (size_t)pv.ws0 * n;
  uint32_t* row_ptr = ctx->d_row_ptr + (size_t)pv.ws0 * RP;
  uint32_t* val_idx = ctx->d_val_idx + (size_t)pv.ws0 * n;
  uint32_t* buckets = ctx->d_buckets + (size_t)pv.ws0 * CV::BKT_WORDS * NB;
  uint32_t* row_ovf_base = ctx->d_row_ovf_base + (size_t)pv.ws0 * NB;
  uint32_t* split_rows = ctx->d_split_rows + (size_t)pv.ws0 * NB;
  WorkItem* work = ctx->d_work + pv.work_off;
  uint32_t* ovf = ctx->d_ovf + pv.ovf_off * CV::BKT_WORDS;
  const uint32_t* bases = ph.table ? ph.table : ctx->d_bases + ph.base_first * BP::REC_WORDS;
  uint32_t* meta_block = ctx->d_work_meta + (size_t)part * META_BLOCK_WORDS;  // [work-list counters | key_max[16]]
  uint32_t* key_max = meta_block + (2 * SEG_BINS + 4);
  if (ph.front) {
  // One memset clears this part's work-list counters AND its key_max words (0 = full-width ranges); k_decompose
  // then measures window 15 of the plain front end.
  hipLaunchKernelGGL(k_clear_words, dim3(1), dim3(256), 0, st, meta_block, META_BLOCK_WORDS, (uint32_t*)d_err, (ph.clear_err && part == 0) ? 1u : 0u);
  uint32_t* top_key_max = (ctx->key_shift && !glv && pv.wb + wc == MSM377_NUM_WINDOWS) ? key_max + (wc - 1) : nullptr;
  const uint64_t max_items = (uint64_t)wc * NB + (uint64_t)wc * n / SEG;  // every row has an item; extra ones are full segments
  // a lane quad per work item while the launch is one chain's latency (up to 2^14 points: ~94 k items); beyond that
  // the quads are VALU-bound like threads and only add their exchange instructions (kernel at 2^16: 0.216 / 0.183 ms)
  const bool quad_acc = std::is_same<BP, CV>::value && std::is_same<CV, TeDev>::value && narrow && ctx->narrow_quad_acc && !ph.table && max_items <= ctx->narrow_quad_items;
  // (One launch for the whole front end of such a call -- each window's workgroup recoding, sorting and listing its
  // work items itself, one global atomic per list and workgroup -- was built and dropped: 0.271 -> 0.293 ms at 2^12,
  // 0.342 -> 0.373 at 2^14.  Saving four dispatch latencies did not pay for a work list that is sorted by length
  // only within each window: the accumulation kernel went from 0.038 to 0.054 ms at 2^12.)
  {
    StageTimer t(ctx, MSM377_STAGE_DECOMPOSE, st, part);
    if (narrow)
      hipLaunchKernelGGL(k_decompose_narrow, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, digits, n, ph.cbits, L, wc, d_err);
    else if (glv)
      hipLaunchKernelGGL(k_decompose_glv, dim3((unsigned)((n_scalars + 255) / 256)), dim3(256), 0, st, d_scalars, digits, n_scalars, pv.wb, wc, d_err);
    else
      hipLaunchKernelGGL(k_decompose, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, digits, n, pv.wb, wc, d_err, top_key_max);
    HIP_TRY(ctx, hipGetLastError());
  }

  if (narrow) {
    StageTimer t(ctx, MSM377_STAGE_SORT, st, part);
    hipLaunchKernelGGL(k_small_sort, dim3(wc), dim3(1024), 0, st, digits, row_ptr, val_idx, (uint32_t)n, L);
    HIP_TRY(ctx, hipGetLastError());
    if (part == 0) HIP_TRY(ctx, hipEventRecord(ctx->sort_done, st));
  } else {
    StageTimer t(ctx, MSM377_STAGE_SORT, st, part);
    uint32_t chunks = sort_blocks / wc;
    const uint64_t want = (n + 4095) / 4096;  // at least ~4096 elements per block
    if (chunks > want) chunks = (uint32_t)(want ? want : 1);
    const uint64_t per_chunk = (n + chunks - 1) / chunks;
    hipLaunchKernelGGL(k_range_count, dim3(chunks, wc), dim3(1024), 0, st, digits, range_counts, n, chunks, per_chunk, key_max);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_range_scan, dim3(wc), dim3(NRANGE), 0, st, range_counts, region_base, chunks);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_partition, dim3(chunks, wc), dim3(1024), 0, st, digits, range_counts, sort_temp, n, chunks, per_chunk, key_max);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_local_sort, dim3(NRANGE, wc), dim3(256), 0, st, sort_temp, region_base, row_ptr, val_idx, n, key_max);
    HIP_TRY(ctx, hipGetLastError());
    if (part == 0) HIP_TRY(ctx, hipEventRecord(ctx->sort_done, st));
  }
  {
    StageTimer t(ctx, MSM377_STAGE_ACCUMULATE, st, part);
    const uint32_t rows = wc * NB;
    uint32_t* meta = meta_block;
    uint32_t* work_hist = meta;
    uint32_t* cursor = meta + SEG_BINS;
    uint32_t* total = meta + 2 * SEG_BINS;
    uint32_t* counters = meta + 2 * SEG_BINS + 1;  // [0] split rows, [1] overflow slots
    hipLaunchKernelGGL(k_work_hist, dim3((rows + 1023) / 1024), dim3(1024), 0, st, row_ptr, L, rows, SEG, work_hist, row_ovf_base, counters, split_rows);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_work_scan, dim3(1), dim3(256), 0, st, work_hist, cursor, total);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_work_scatter, dim3((rows + 1023) / 1024), dim3(1024), 0, st, row_ptr, L, rows, SEG, cursor, work);
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->before_accumulate) {  // must run before the wait below is queued: the wait binds to the event's latest record
      std::function<int()> f;
      f.swap(ctx->before_accumulate);
      const int hook_rc = f();
      if (hook_rc) return hook_rc;
    }
    HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->bases_ready, 0));
    // The accumulation launches of the two parts run one after the other (the second waits for the first): they
    // are the power-limited kernels, sharing the GPU would only stretch both.
    if (part == 1) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->acc_done, 0));
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
                           ctx->d_err + 2, ph.into ? 1u : 0u, ph.table_stride, L);
      else
        hipLaunchKernelGGL((k_accumulate<CV, 2>), grid, dim3(256), 0, st, row_ptr, val_idx, bases, buckets, n, work, total, row_ovf_base, ovf, SEG, d_err,
                           ctx->d_err + 2, ph.into ? 1u : 0u, ph.table_stride, L);
    }
    HIP_TRY(ctx, hipGetLastError());
    if (part == 0) HIP_TRY(ctx, hipEventRecord(ctx->acc_done, st));
    bool merged = false;
    if constexpr (CV::HAS_QUAD) {
      if (ctx->merge_quad) {
        hipLaunchKernelGGL(k_merge_split_rows_quad<CV>, dim3(ctx->merge_full_grid ? (rows + 63) / 64 : 4 * MERGE_GRID), dim3(256), 0, st, row_ptr, buckets, counters, split_rows,
                           row_ovf_base, ovf, SEG, d_err, L);
        merged = true;
      }
    }
    if (!merged)
      hipLaunchKernelGGL(k_merge_split_rows<CV>, dim3(ctx->merge_full_grid ? (rows + 255) / 256 : MERGE_GRID), dim3(256), 0, st, row_ptr, buckets, counters, split_rows,
                         row_ovf_base, ovf, SEG, d_err, L);
    HIP_TRY(ctx, hipGetLastError());
  }
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
    uint32_t coop_from = ctx->coop_from;
    if (coop_from == 0)
      for (coop_from = 0; coop_from < levels && 4ull * (coop_from + 1) * (NB >> (coop_from + 1)) * wc > ctx->coop_threads; coop_from++) {
      }
    // Levels [0, coop_from): one thread per addition (VALU-bound: 2^18 additions per level at first); [coop_from,
    // tail_from): one lane quad per addition, one launch per level; [tail_from, levels): k_reduce_tail, one launch.
    const uint32_t tail_from = CV::HAS_QUAD ? std::min(narrow ? ctx->narrow_tail_from : ctx->tail_from, levels) : levels;
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
        hipLaunchKernelGGL(k_reduce_tail<CV>, dim3(tail_from + 1, wc), dim3(TAIL_THREADS), 0, st, buckets, L, tail_from, d_err);
        HIP_TRY(ctx, hipGetLastError());
      }
    }
    if (ctx->zc_active)  // set by enqueue_windows for this call: one part, slot 0
      hipLaunchKernelGGL(k_gather_partials<CV>, dim3((wc * MSM377_G1_PARTIAL_POINTS * 4 + 63) / 64), dim3(64), 0, st, buckets, d_partials, wc, L, ctx->dm_partials,
                         ctx->dm_out_flag, ctx->d_out_count, (const int*)d_err, ctx->out_seq);
    else
      hipLaunchKernelGGL(k_gather_partials<CV>, dim3((wc * MSM377_G1_PARTIAL_POINTS * 4 + 63) / 64), dim3(64), 0, st, buckets,
                         d_partials + (size_t)pv.ws0 * MSM377_G1_PARTIAL_POINTS * CV::OUT_WORDS, wc, L);
    HIP_TRY(ctx, hipGetLastError());
  }
  return MSM377_OK;
}

// Enqueue stages decompose .. gather for windows [wb, wb + wc) against ctx->d_bases, the D2H of
// the partial records into slot `slot` of ctx->h_partials and that slot's completion event.
// Nothing here waits for the GPU.
//
// Large calls run as TWO parts (half the windows each) on two streams: while the power-limited accumulation
// kernel of part 0 runs, the GPU also sorts part 1's digit columns and builds its work list (LDS / latency
// bound), and part 0's merge and bucket reduction (short launches, latency bound from level 5 on) overlap
// part 1's accumulation.  Only the second part's reduction stays exposed.
template <class CV, class BP = CV>
int enqueue_windows(msm377_ctx* ctx, const uint32_t* d_scalars, uint64_t n_scalars, uint32_t wb, uint32_t wc, int slot, bool glv = false,
                    const Phase& ph = Phase()) {
  // GLV front end: n_scalars scalars become 2 n_scalars (point, half-scalar) columns over 8 windows.
  const uint64_t n = glv ? 2 * n_scalars : n_scalars;
  hipStream_t st = ctx->stream;
  int* d_err = ctx->d_err + slot;
  uint32_t* d_partials = ctx->d_partials + (size_t)slot * SLOT_WORDS;
  const bool whole = ph.front && ph.back;  // the two-stream pipeline only for calls enqueued in one piece
  const uint32_t parts = (whole && ctx->pipeline_parts == 2 && wc >= 2 && !ctx->capture && !ph.table && (uint64_t)wc * n >= PIPELINE_MIN_ENTRIES) ? 2u : 1u;
  // the error word is cleared by part 0's first kernel together with its counters -- unless there is no such kernel
  // (back phase only) or a second part on another stream could raise a bit before that kernel has run
  const bool clear_here = ph.clear_err && (!ph.front || parts == 2);
  if (clear_here) hipLaunchKernelGGL(k_clear_words, dim3(1), dim3(256), 0, st, (uint32_t*)d_err, 1u, (uint32_t*)nullptr, 0u);
  Phase part_phase = ph;
  part_phase.clear_err = ph.clear_err && !clear_here;
  PartView pv[2];
  const uint32_t wc0 = parts == 2 ? (wc + 1) / 2 : wc;
  pv[0] = PartView{st, 0, 0, wb, wc0, 0, 0};
  pv[1] = PartView{ctx->stream3, 1, wc0, wb + wc0, wc - wc0, (size_t)wc0 * NB + (size_t)wc0 * n / SEG_MIN + 1, (size_t)wc0 * n / SEG_MIN + 1};
  if (parts == 2) {
    HIP_TRY(ctx, hipEventRecord(ctx->part_fork, st));  // after the error word is cleared and everything queued before this call
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream3, ctx->part_fork, 0));
  }
  ctx->last_parts = parts;
  ctx->zc_active = ph.zc_out && ph.back && ctx->zc_out && parts == 1 && slot == 0 && !ph.table;
  if (ctx->zc_active) ctx->out_seq++;
  for (uint32_t p = 0; p < parts; p++) {
    int rc = enqueue_part<CV, BP>(ctx, d_scalars, n_scalars, n, pv[p], d_err, d_partials, glv, parts == 2 ? MAX_SORT_BLOCKS / 2 : MAX_SORT_BLOCKS, part_phase);
    if (rc) return rc;
  }
  if (!ph.back) return MSM377_OK;
  if (parts == 2) {
    HIP_TRY(ctx, hipEventRecord(ctx->part_join, ctx->stream3));
    HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->part_join, 0));
  }
  const uint32_t wc_out = ph.table ? 1u : wc;  // precomputed-window tables fold the windows on the GPU
  if (!ctx->zc_active) {
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_partials + (size_t)slot * SLOT_WORDS, d_partials, (size_t)wc_out * MSM377_G1_PARTIAL_POINTS * CV::OUT_WORDS * 4,
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
      for (uint32_t p = 0; p < (s == MSM377_STAGE_CONVERT ? 1u : ctx->last_parts); p++) {
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

constexpr uint32_t GLV_WINDOWS = 8;
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

// Host tail of ONE MSM on up to 8 threads (MSM377_TAIL_THREADS).  The Horner chain over the `positions` = windows x
// cbits bit positions is cut into one piece per thread; the thread that owns positions [lo, hi) runs its own chain
// (hi - lo steps of doubling + addition, ~17 field products each) and then doubles its result `lo` times (7 products
// each: `dbl_nt`, a doubling whose result is only doubled again, skips T in the Edwards form), and the caller adds
// the pieces up.  The cuts balance (hi - lo) x 17 + lo x 7 over the threads, so the pieces shrink towards the top:
// 256 positions on 6 threads are 110 / 65 / 38 / 22 / 13 / 8 positions and ~1 900 products on the critical path,
// against 2 496 for six equal blocks stitched by the caller and ~4 350 for one thread.
// piece(lo, hi, k) -> the sum of the positions' records x 2^position (fp64_host.hpp teh_tail_piece: its own Horner chain,
// then lo doublings); add(a, b, k) adds two pieces; k is the piece's index (the Edwards form keeps one exceptional-case
// record per piece).  The caller runs the top piece itself and adds the others up as they finish.
template <class Pt, class PieceFn, class AddFn>
Pt tail_horner_mt(msm377_ctx* ctx, PieceFn piece, AddFn add, int positions) {
  constexpr int MAXC = TailPool::WORKERS + 1;
  int bounds[MAXC + 1];
  const int used = tail_split(positions, std::max(1, std::min(ctx->tail_threads, MAXC)), bounds);
  TailPool& pool = ctx->tail_pool;
  if (used > 1) pool.start();
  Pt part[MAXC];
  // MSM377_TAIL_TRACE=1: per-piece start / end (us after the call) and CPU, on stderr
  const bool trace = ctx->tail_trace;
  const int64_t t_call = trace ? TailPool::now_ns() : 0;
  struct alignas(128) Mark {
    int64_t t0, t1;
    int cpu;
  } mark[MAXC];
  auto chain = [&part, &bounds, &mark, trace, piece](int k) {
    if (trace) mark[k].t0 = TailPool::now_ns(), mark[k].cpu = sched_getcpu();
    part[k] = piece(bounds[k], bounds[k + 1], k);
    if (trace) mark[k].t1 = TailPool::now_ns();
  };
  for (int k = 0; k + 1 < used; k++) pool.post(k, [&chain, k] { chain(k); });
  chain(used - 1);  // the top piece: the fewest positions, the most doublings
  Pt acc = part[used - 1];
  for (int k = used - 2; k >= 0; k--) {
    pool.wait(k);
    acc = add(acc, part[k], used - 1);
  }
  if (trace) {
    fprintf(stderr, "tail trace:");
    for (int k = 0; k < used; k++) fprintf(stderr, "  [%d cpu %d: %.1f..%.1f]", k, mark[k].cpu, (mark[k].t0 - t_call) / 1e3, (mark[k].t1 - t_call) / 1e3);
    fprintf(stderr, "  done %.1f us\n", (TailPool::now_ns() - t_call) / 1e3);
  }
  return acc;
}

// true: an addition or doubling of the tail hit an exceptional case of the Edwards law (fp64_host.hpp TeChecked;
// out_xy untouched) -- the caller reruns on the Weierstrass path, exactly as for the GPU-side flag.
bool te_tail(msm377_ctx* ctx, const uint32_t* partials, uint8_t out_xy[96], int num_windows = MSM377_NUM_WINDOWS, int cbits = 16, int planes = 15) {
  if (ctx->tail_threads <= 1 || num_windows < 8) return teh_combine(partials, num_windows, out_xy, cbits, planes);
  TeChecked chk[TailPool::WORKERS + 1];  // one per piece
  const TeH::Ext r = tail_horner_mt<TeH::Ext>(
      ctx, [&chk, partials, cbits, planes](int lo, int hi, int k) { return teh_tail_piece(partials, lo, hi, chk[k], cbits, planes); },
      [&chk](const TeH::Ext& a, const TeH::Ext& b, int k) { return chk[k].add(a, b); }, cbits * num_windows);
  for (const TeChecked& c : chk)
    if (c.bad) return true;
  teh_to_wire(r, out_xy);
  return false;
}

void xyzz_tail(msm377_ctx* ctx, const uint32_t* partials, uint8_t out_xy[96]) {
  if (ctx->tail_threads <= 1) return g1h_combine(partials, MSM377_NUM_WINDOWS, out_xy);
  const G1H::XYZZ r = tail_horner_mt<G1H::XYZZ>(
      ctx,
      [partials](int lo, int hi, int) {
        G1H::XYZZ acc = g1h_horner_bits(partials, lo, hi);
        for (int i = 0; i < lo; i++) acc = G1H::dbl(acc);
        return acc;
      },
      [](const G1H::XYZZ& a, const G1H::XYZZ& b, int) { return G1H::add(a, b); }, 16 * MSM377_NUM_WINDOWS);
  g1h_to_wire(r, out_xy);
}

void ed_tail(msm377_ctx* ctx, const uint32_t* partials, uint8_t out_xy[64]) {  // Edwards-BLS12: a complete law, nothing to check
  if (ctx->tail_threads <= 1) return edh_combine(partials, out_xy);
  const EdH::Ext r = tail_horner_mt<EdH::Ext>(
      ctx,
      [partials](int lo, int hi, int) {
        EdH::Ext acc = edh_horner_bits(partials, lo, hi);
        for (int i = 0; i + 1 < lo; i++) acc = EdH::dbl_nt(acc);
        if (lo > 0) acc = EdH::dbl(acc);
        return acc;
      },
      [](const EdH::Ext& a, const EdH::Ext& b, int) { return EdH::add(a, b); }, 16 * MSM377_NUM_WINDOWS);
  edh_to_wire(r, out_xy);
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
    if (hipEventSynchronize(c->acc_done) != hipSuccess) return;  // the caller's own wait reports the error
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
    for (;;) {
      uint32_t windows = MSM377_NUM_WINDOWS;
      int cbits = MSM377_WINDOW_BITS, planes = MSM377_WINDOW_BITS - 1;
      ph.cbits = MSM377_WINDOW_BITS;
      ph.bucket_log = MSM377_WINDOW_BITS - 1;
      if (narrow) {
        ph.cbits = NARROW_BITS;
        ph.bucket_log = NARROW_LOG;
        windows = NARROW_WINDOWS;
        cbits = NARROW_BITS;
        planes = NARROW_LOG;
      }
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
      if (narrow && (ctx->h_err[0] & ERR_NARROW_RANGE) && !(ctx->h_err[0] & ERR_SCALAR)) {  // a scalar >= 2^253: the 16-bit path takes it
        narrow = false;
        continue;
      }
      if (ctx->h_err[0] & ERR_TE_ANY) {
        note_fallback(ctx, (uint32_t)(ctx->h_err[0] & ERR_TE_ANY));
        return RC_TE_FALLBACK;
      }
      rc = finish_windows(ctx, 0);
      if (rc) return rc;
      auto t0 = std::chrono::steady_clock::now();
      const bool bad = form == TABLE_TE_PRECOMP ? teh_combine(ctx->h_partials, 1, out_xy) : te_tail(ctx, ctx->h_partials, out_xy, (int)windows, cbits, planes);
      time_tail(ctx, t0);
      if (bad) note_fallback(ctx, MSM377_FB_TAIL);
      return bad ? RC_TE_FALLBACK : MSM377_OK;
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
  xyzz_tail(ctx, ctx->h_partials, out_xy);
  time_tail(ctx, t0);
  return MSM377_OK;
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

// Host-buffer entry points with large inputs: the upload (2.9 ms for 2^20 G1 points from pageable memory) is as
// long as the whole computation, so the MSM runs as K chunks of points: a chunk's decompose .. accumulate .. merge
// runs while the next one is on its way, later chunks accumulate on top of the buckets (Phase::into), and reduction,
// gather and D2H are queued once, with the last chunk.  Returns with everything enqueued (slot 0).
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
    rc = convert_bases<CV>(ctx, ctx->d_raw_points + first * CV::RAW_WORDS, cnt, first, c == 0);
    if (rc == MSM377_OK) rc = enqueue_windows<CV>(ctx, ctx->d_raw_scalars + first * 8, cnt, 0, MSM377_NUM_WINDOWS, 0, false, ph);
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
  if (const char* e = getenv("MSM377_MERGE_QUAD")) ctx->merge_quad = atoi(e) != 0;
  if (const char* e = getenv("MSM377_GLV")) ctx->glv_mode = atoi(e);
  if (const char* e = getenv("MSM377_G1_FORM")) ctx->g1_form = atoi(e) != 0;
  if (const char* e = getenv("MSM377_H2D_THREADS")) ctx->h2d_threads = std::min(std::max(atoi(e), 1), 8);
  if (const char* e = getenv("MSM377_UPLOAD_CHUNKS")) ctx->upload_chunks = (uint32_t)std::min(std::max(atoi(e), 2), 8);
  if (const char* e = getenv("MSM377_UPLOAD_SPLIT")) ctx->upload_split_pct = (uint32_t)std::min(std::max(atoi(e), 10), 90);
  if (const char* e = getenv("MSM377_UPLOAD_CHUNK_MIN")) ctx->upload_chunk_min = strtoull(e, nullptr, 10);
  if (const char* e = getenv("MSM377_KEY_SHIFT")) ctx->key_shift = atoi(e) != 0;
  if (const char* e = getenv("MSM377_TAIL_THREADS")) ctx->tail_threads = atoi(e);
  if (const char* e = getenv("MSM377_TAIL_SPIN_US")) ctx->tail_spin_us = atoll(e);
  if (const char* e = getenv("MSM377_TAIL_TRACE")) ctx->tail_trace = atoi(e) != 0;
  if (const char* e = getenv("MSM377_TAIL_NUMA")) ctx->tail_pool.numa_local = atoi(e) != 0;
  if (const char* e = getenv("MSM377_PIPELINE")) ctx->pipeline_parts = atoi(e) == 2 ? 2 : 1;
  if (const char* e = getenv("MSM377_TE_AFFINE_TABLE")) ctx->te_affine_table = atoi(e) != 0;
  if (const char* e = getenv("MSM377_TE_AFFINE_MSM")) ctx->te_affine_msm = atoi(e) != 0;
  if (const char* e = getenv("MSM377_NARROW_MAX")) ctx->narrow_max_points = strtoull(e, nullptr, 10);
  if (const char* e = getenv("MSM377_AFF_AFTER_SORT")) ctx->aff_down_after_sort = atoi(e);
  if (const char* e = getenv("MSM377_AFFINE_MIN")) ctx->affine_min_points = strtoull(e, nullptr, 10);
  if (const char* e = getenv("MSM377_MERGE_FULL_GRID")) ctx->merge_full_grid = atoi(e) != 0;
  if (const char* e = getenv("MSM377_SEG_PLAIN")) ctx->seg_plain = std::min(std::max(atoi(e), (int)SEG_MIN), (int)SEG_MAX);
  if (const char* e = getenv("MSM377_SEG_GLV")) ctx->seg_glv = std::min(std::max(atoi(e), (int)SEG_MIN), (int)SEG_MAX);
  if (const char* e = getenv("MSM377_COOP_FROM")) ctx->coop_from = (uint32_t)atoi(e);
  if (const char* e = getenv("MSM377_ZERO_COPY_OUT")) ctx->zc_out = atoi(e);
  if (const char* e = getenv("MSM377_NARROW_SEG")) ctx->narrow_seg = (uint32_t)std::min(std::max(atoi(e), (int)NARROW_SEG), (int)SEG_BINS - 1);
  if (const char* e = getenv("MSM377_NARROW_QUAD_ITEMS")) ctx->narrow_quad_items = strtoull(e, nullptr, 10);
  if (const char* e = getenv("MSM377_NARROW_QUAD_ACC")) ctx->narrow_quad_acc = atoi(e);
  if (const char* e = getenv("MSM377_COOP_THREADS")) ctx->coop_threads = (uint32_t)atoi(e);
  if (const char* e = getenv("MSM377_NARROW_TAIL_FROM")) ctx->narrow_tail_from = (uint32_t)std::min(std::max(atoi(e), 1), (int)TREE_LEVELS);
  if (const char* e = getenv("MSM377_TAIL_FROM")) ctx->tail_from = (uint32_t)std::min(std::max(atoi(e), 1), (int)TREE_LEVELS);
  const uint64_t cap = max_points;
  // The main stream outranks the side stream: the base conversion (VALU-heavy, ~0.2 ms) only has to finish before
  // the accumulation starts, decompose + sort on the main stream are the critical path (k_decompose: 16 us alone,
  // ~100 us when it competes with the conversion at equal priority).
  int prio_least = 0, prio_greatest = 0;
  bool ok = hipSetDevice(device) == hipSuccess;
  if (ok && hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) prio_least = prio_greatest = 0;
  const bool use_prio = getenv("MSM377_STREAM_PRIORITY") ? atoi(getenv("MSM377_STREAM_PRIORITY")) != 0 : true;
  ok = ok && hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, use_prio ? prio_greatest : prio_least) == hipSuccess &&
            hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, prio_least) == hipSuccess &&
            hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&ctx->bases_ready, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ctx->part_fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ctx->part_join, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ctx->acc_done, hipEventDisableTiming) == hipSuccess;
  auto dalloc = [&](void** p, size_t bytes) { ok = ok && hipMalloc(p, bytes) == hipSuccess; };
  dalloc((void**)&ctx->d_raw_points, cap * 96);
  dalloc((void**)&ctx->d_raw_scalars, cap * 32);
  dalloc((void**)&ctx->d_bases, 2 * cap * G1Dev::REC_WORDS * 4);  // 128-byte records of P_i and phi(P_i) (GLV front end), or 256-byte twisted Edwards records of P_i
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
  dalloc((void**)&ctx->d_aff_stash, cap * AFF_STASH_WORDS * 4);
  dalloc((void**)&ctx->d_aff_trees, aff_blocks * 2 * AFF_THREADS * 13 * 4);
  const unsigned host_flags = hipHostMallocMapped | hipHostMallocCoherent;
  ok = ok && hipHostMalloc((void**)&ctx->h_aff_prod, aff_blocks * 48, host_flags) == hipSuccess &&
       hipHostMalloc((void**)&ctx->h_aff_inv, aff_blocks * 48, host_flags) == hipSuccess &&
       hipHostMalloc((void**)&ctx->h_aff_flag, 64, host_flags) == hipSuccess &&
       hipHostGetDevicePointer((void**)&ctx->dm_aff_prod, ctx->h_aff_prod, 0) == hipSuccess &&
       hipHostGetDevicePointer((void**)&ctx->dm_aff_inv, ctx->h_aff_inv, 0) == hipSuccess &&
       hipHostGetDevicePointer((void**)&ctx->dm_aff_flag, ctx->h_aff_flag, 0) == hipSuccess &&
       hipEventCreateWithFlags(&ctx->aff_up_done, hipEventDisableTiming) == hipSuccess &&
       hipEventCreateWithFlags(&ctx->sort_done, hipEventDisableTiming) == hipSuccess;
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
  if (ctx->stream3) (void)hipStreamSynchronize(ctx->stream3);
  void* bufs[] = {ctx->d_raw_points, ctx->d_raw_scalars, ctx->d_bases, ctx->d_digits, ctx->d_range_counts, ctx->d_region_base, ctx->d_sort_temp,
                  ctx->d_row_ptr, ctx->d_val_idx, ctx->d_buckets, ctx->d_buckets_snap, ctx->d_partials, ctx->d_work, ctx->d_work_meta, ctx->d_row_ovf_base, ctx->d_split_rows, ctx->d_ovf, ctx->d_err, ctx->d_aff_stash, ctx->d_aff_trees, ctx->d_aff_count, ctx->d_out_count, ctx->d_table};
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
  if (ctx->sort_done) (void)hipEventDestroy(ctx->sort_done);
  for (int t = 0; t < 8; t++)
    if (ctx->copy_stream[t]) (void)hipStreamDestroy(ctx->copy_stream[t]);
  for (int k = 0; k < 2; k++)
    if (ctx->done_ev[k]) (void)hipEventDestroy(ctx->done_ev[k]);
  for (int s = 0; s < MSM377_NUM_STAGES; s++)
    for (int k = 0; k < 2; k++)
      for (int p = 0; p < 2; p++)
        if (ctx->ev[p][s][k]) (void)hipEventDestroy(ctx->ev[p][s][k]);
  if (ctx->bases_ready) (void)hipEventDestroy(ctx->bases_ready);
  for (hipEvent_t e : {ctx->part_fork, ctx->part_join, ctx->acc_done})
    if (e) (void)hipEventDestroy(e);
  if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* msm377_last_error(const msm377_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int msm377_g1_msm_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint8_t out_xy[96]) {
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

int msm377_g1_msm(msm377_ctx* ctx, const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out_xy[96]) {
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
  if (n >= ctx->upload_chunk_min && form != TABLE_XYZZ_GLV) {
    const bool te = form == TABLE_TE;
    rc = te ? run_chunked_upload<TeDev>(ctx, points, scalars, n) : run_chunked_upload<G1Dev>(ctx, points, scalars, n);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventSynchronize(ctx->done_ev[0]));
    if (!(te && (ctx->h_err[0] & ERR_TE_ANY))) {
      rc = finish_windows(ctx, 0);
      if (rc) return rc;
      auto t0 = std::chrono::steady_clock::now();
      bool bad = false;
      if (te)
        bad = te_tail(ctx, ctx->h_partials, out_xy);
      else
        xyzz_tail(ctx, ctx->h_partials, out_xy);
      time_tail(ctx, t0);
      if (!bad) return MSM377_OK;
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
int msm377_ed_msm_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint8_t out_xy[64]) {
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
  Phase ph;
  ph.zc_out = true;
  rc = enqueue_windows<EdDev>(ctx, (const uint32_t*)d_scalars, n, 0, MSM377_NUM_WINDOWS, 0, false, ph);
  if (rc) return rc;
  arm.after_accumulation();
  rc = finish_windows(ctx, 0);
  if (rc) return rc;
  auto t0 = std::chrono::steady_clock::now();
  ed_tail(ctx, ctx->h_partials, out_xy);
  time_tail(ctx, t0);
  return MSM377_OK;
}

int msm377_ed_msm(msm377_ctx* ctx, const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out_xy[64]) {
  if (!ctx || !out_xy) return MSM377_EINVAL;
  ctx->err.clear();
  if (n > ctx->cap || (n && (!points || !scalars))) {
    ctx->err = "bad arguments";
    return MSM377_EINVAL;
  }
  if (n == 0) return msm377_ed_msm_device(ctx, nullptr, nullptr, 0, out_xy);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (n >= ctx->upload_chunk_min) {  // chunks of points, like msm377_g1_msm: a chunk computes while the next one uploads
    ctx->bases_n = 0;
    int rc = run_chunked_upload<EdDev>(ctx, points, scalars, n);
    if (rc) return rc;
    rc = finish_windows(ctx, 0);
    if (rc) return rc;
    auto t0 = std::chrono::steady_clock::now();
    ed_tail(ctx, ctx->h_partials, out_xy);
    time_tail(ctx, t0);
    return MSM377_OK;
  }
  int rc = h2d_staged(ctx, ctx->d_raw_points, points, n * 64, 0);
  if (rc == MSM377_OK) rc = h2d_staged(ctx, ctx->d_raw_scalars, scalars, n * 32, (size_t)ctx->cap * 96);
  if (rc) return rc;
  return msm377_ed_msm_device(ctx, ctx->d_raw_points, ctx->d_raw_scalars, n, out_xy);
}

int msm377_ed_generate_bases_device(msm377_ctx* ctx, uint64_t seed, uint64_t n, void* d_points_out) {
  if (!ctx || (n && !d_points_out) || ((uintptr_t)d_points_out & 15)) return MSM377_EINVAL;
  if (n == 0) return MSM377_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_generate_bases_ed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, seed, n, (uint32_t*)d_points_out);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MSM377_OK;
}

int msm377_g1_set_bases_device(msm377_ctx* ctx, const void* d_points, uint64_t n) {
  int rc = check_args(ctx, d_points, d_points, n, true);
  if (rc) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int form = pick_form(ctx, n);
  if (form == TABLE_TE && ctx->te_affine_table) form = TABLE_TE_AFFINE;  // resident: one inversion per point, once
  rc = convert_table(ctx, (const uint32_t*)d_points, n, form);
  if (rc) return rc;
  // raw copy for the (never expected) fallback from the Edwards form: see resident_table_to_weierstrass
  if (form_is_te(form) && d_points != ctx->d_raw_points)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_raw_points, d_points, n * 96, hipMemcpyDeviceToDevice, ctx->stream2));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2));
  ctx->bases_n = n;
  ctx->bases_form = form;
  return MSM377_OK;
}

int msm377_g1_set_bases(msm377_ctx* ctx, const uint8_t* points, uint64_t n) {
  if (!ctx || n > ctx->cap || (n && !points)) return MSM377_EINVAL;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = h2d_staged(ctx, ctx->d_raw_points, points, n * 96, 0);
  if (rc) return rc;
  return msm377_g1_set_bases_device(ctx, ctx->d_raw_points, n);
}

// Precomputed-window tables (BASELINE.json config 5 "precomputed-point reuse"; the reference lists precomputation as
// future work, README.md:558-563): T[w][i] = [2^(16 w)] P_i for the 16 windows, as affine Edwards records.  Window 0
// is the batched conversion of the input; every further window doubles the previous one 16 times (unified law) and
// runs through the same batched inversion (k_affine_up<AffDoublingSource> -> host -> k_affine_down).
int msm377_g1_set_bases_precomputed_device(msm377_ctx* ctx, const void* d_points, uint64_t n) {
  int rc = check_args(ctx, d_points, d_points, n, true);
  if (rc) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->g1_form != 1 || n == 0) return msm377_g1_set_bases_device(ctx, d_points, n);  // Weierstrass form: no precomputation
  if (ctx->table_cap < n) {
    if (ctx->d_table) (void)hipFree(ctx->d_table);
    ctx->d_table = nullptr;
    ctx->table_cap = 0;
    if (hipMalloc((void**)&ctx->d_table, (size_t)MSM377_NUM_WINDOWS * n * TeAffBase::REC_WORDS * 4) != hipSuccess) {
      ctx->err = "precomputed-window table: out of device memory";
      return MSM377_ENOMEM;
    }
    ctx->table_cap = n;
  }
  for (uint32_t w = 0; w < MSM377_NUM_WINDOWS && rc == MSM377_OK; w++) {
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

int msm377_g1_set_bases_precomputed(msm377_ctx* ctx, const uint8_t* points, uint64_t n) {
  if (!ctx || n > ctx->cap || (n && !points)) return MSM377_EINVAL;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = h2d_staged(ctx, ctx->d_raw_points, points, n * 96, 0);
  if (rc) return rc;
  return msm377_g1_set_bases_precomputed_device(ctx, ctx->d_raw_points, n);
}

int msm377_g1_msm_fixed_base_device(msm377_ctx* ctx, const void* d_scalars, uint64_t n, uint8_t out_xy[96]) {
  if (!out_xy) return MSM377_EINVAL;
  int rc = check_args(ctx, nullptr, d_scalars, n, false);
  if (rc) return rc;
  if (n > ctx->bases_n) {
    ctx->err = "fixed-base MSM needs msm377_g1_set_bases with at least n points first";
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

int msm377_g1_msm_fixed_base_batch_device(msm377_ctx* ctx, const void* d_scalars, uint64_t n, uint32_t batch, uint8_t* out_xy) {
  if (!out_xy) return MSM377_EINVAL;
  int rc = check_args(ctx, nullptr, d_scalars, n, false);
  if (rc) return rc;
  if (n > ctx->bases_n) {
    ctx->err = "fixed-base MSM needs msm377_g1_set_bases with at least n points first";
    return MSM377_ESTATE;
  }
  if (n == 0) {
    for (uint32_t b = 0; b < batch; b++) identity_wire(out_xy + (size_t)96 * b);
    return MSM377_OK;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint32_t* sc = (const uint32_t*)d_scalars;
  const int form = resident_form(ctx, n);
  const bool glv = form == TABLE_XYZZ_GLV, te = form_is_te(form);
  const uint32_t W = glv ? GLV_WINDOWS : MSM377_NUM_WINDOWS;
  Phase table_phase;
  if (form == TABLE_TE_PRECOMP) {
    table_phase.table = ctx->d_table;
    table_phase.table_stride = ctx->bases_n;
  }
  const int W_tail = form == TABLE_TE_PRECOMP ? 1 : (int)W;  // window records the host combines per MSM
  std::vector<uint32_t> redo;  // elements whose scalars fall outside the GLV range: rerun plain afterwards
  bool te_fallback = false;
  // Software pipeline over the batch: while the GPU runs MSM b, the host finishes MSM b-1
  // (Horner + inversion on the other slot's partial records).
  for (uint32_t b = 0; b <= batch; b++) {
    if (b < batch && !te_fallback) {
      rc = (form == TABLE_TE_AFFINE || form == TABLE_TE_PRECOMP) ? enqueue_windows<TeDev, TeAffBase>(ctx, sc + (size_t)b * n * 8, n, 0, W, (int)(b & 1), false, table_phase)
           : te                    ? enqueue_windows<TeDev>(ctx, sc + (size_t)b * n * 8, n, 0, W, (int)(b & 1))
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
      rc = finish_windows(ctx, slot);
      if (rc) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
      }
      if (te) {
        if (teh_combine(ctx->h_partials + (size_t)slot * SLOT_WORDS, W_tail, out_xy + (size_t)96 * (b - 1))) {
          te_fallback = true;
          note_fallback(ctx, MSM377_FB_TAIL);
        }
      } else
        g1h_combine(ctx->h_partials + (size_t)slot * SLOT_WORDS, W, out_xy + (size_t)96 * (b - 1));
    }
  }
  if (te_fallback) {  // an exceptional case of the Edwards law somewhere in the batch: Weierstrass table, whole batch again
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    rc = resident_table_to_weierstrass(ctx);
    if (rc) return rc;
    return msm377_g1_msm_fixed_base_batch_device(ctx, d_scalars, n, batch, out_xy);
  }
  for (uint32_t b : redo) {
    rc = g1_table_msm(ctx, sc + (size_t)b * n * 8, n, TABLE_XYZZ, out_xy + (size_t)96 * b);
    if (rc) return rc;
  }
  return MSM377_OK;
}

int msm377_g1_msm_fixed_base(msm377_ctx* ctx, const uint8_t* scalars, uint64_t n, uint8_t out_xy[96]) {
  if (!ctx || !out_xy || n > ctx->cap || (n && !scalars)) return MSM377_EINVAL;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = h2d_staged(ctx, ctx->d_raw_scalars, scalars, n * 32, (size_t)ctx->cap * 96);
  if (rc) return rc;
  return msm377_g1_msm_fixed_base_device(ctx, ctx->d_raw_scalars, n, out_xy);
}

// Windows [win_begin, win_begin + win_count) of a G1 MSM; the records go to a host buffer, a device buffer, or both.
static int window_partials(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint32_t win_begin, uint32_t win_count,
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
  if (ctx->tail_threads > 1) ctx->tail_pool.prewake(ctx->tail_spin_us, std::min(ctx->tail_threads, TailPool::WORKERS + 1) - 1);  // the combine of the gathered records follows (msm377_g1_combine_partials_ctx disarms)
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
    rc = convert_bases<TeDev>(ctx, (const uint32_t*)d_points, n);
    if (rc) return rc;
    rc = enqueue_windows<TeDev>(ctx, (const uint32_t*)d_scalars, n, win_begin, win_count, 0);
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

int msm377_g1_window_partials_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint32_t win_begin,
                                     uint32_t win_count, uint8_t* partials_out) {
  if (!partials_out) return MSM377_EINVAL;
  return window_partials(ctx, d_points, d_scalars, n, win_begin, win_count, partials_out, nullptr);
}

int msm377_g1_window_partials_resident(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint32_t win_begin,
                                       uint32_t win_count, void* d_partials_out) {
  if (!d_partials_out) return MSM377_EINVAL;
  return window_partials(ctx, d_points, d_scalars, n, win_begin, win_count, nullptr, d_partials_out);
}

int msm377_g1_glv_window_partials_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint32_t win_begin,
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
  struct Disarm {
    msm377_ctx* c;
    ~Disarm() { c->tail_pool.disarm(); }
  } disarm{ctx};
  if (all_te)
    rc = te_tail(ctx, rec, out_xy) ? MSM377_EEXCEPTIONAL : MSM377_OK;
  else if (all_w)
    xyzz_tail(ctx, rec, out_xy);
  else
    rc = g1_combine_tagged(rec, MSM377_NUM_WINDOWS, out_xy) ? MSM377_EEXCEPTIONAL : MSM377_OK;
  if (rc) ctx->err = "the partial records add up to an exceptional case of the twisted Edwards law (points outside the prime-order subgroup): recompute them in form 0";
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

int msm377_g1_combine_partials_split(const uint8_t* partials, uint32_t pieces, uint8_t out_xy[96]) {
  if (!partials || !out_xy || ((uintptr_t)partials & 3) || pieces < 1 || pieces > 64) return MSM377_EINVAL;
  const uint32_t* p = reinterpret_cast<const uint32_t*>(partials);
  for (int w = 0; w < MSM377_NUM_WINDOWS; w++)
    if (!window_record_is_te(p + (size_t)w * 16 * 48)) return MSM377_EINVAL;  // the decomposition of the Edwards tail only
  return teh_combine_split(p, MSM377_NUM_WINDOWS, out_xy, (int)pieces) ? MSM377_EEXCEPTIONAL : MSM377_OK;
}

int msm377_g1_generate_bases_device(msm377_ctx* ctx, uint64_t seed, uint64_t n, void* d_points_out) {
  if (!ctx || (n && !d_points_out) || ((uintptr_t)d_points_out & 15)) return MSM377_EINVAL;
  if (n == 0) return MSM377_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_generate_bases, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, seed, n, (uint32_t*)d_points_out);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MSM377_OK;
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

}  // extern "C"
