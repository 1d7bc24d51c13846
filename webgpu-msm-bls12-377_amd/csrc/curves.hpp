// Curve policies: what the curve-agnostic pipeline kernels need from a curve, and the bucket-record layout.
//   G1Dev  BLS12-377 G1, Weierstrass XYZZ (g1_xyzz.hpp): the fallback, the GLV front end, the stage read-backs
//   TeDev  the same group in twisted Edwards form (te377.hpp): the default; TeAffBase = its affine base records
//   EdDev  Edwards-BLS12 over the scalar field (ed_ext.hpp), BASELINE.json config 3
// Device code; included by sequencer.hip only (through the kernel headers).
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "fp64_host.hpp"
#include "g1_xyzz.hpp"
#include "te377.hpp"

namespace msm377 {
namespace {

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

static_assert(G1Dev::REC_WORDS == G1_REC_WORDS, "common.hpp sizes the base table with it");

}  // namespace
}  // namespace msm377
