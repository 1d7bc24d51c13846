// msm377 engine: constants and plain types shared by the kernels (kernels/*.hpp), the stage sequencer
// (sequencer.hip), the context (context.hpp) and the C ABI (capi.hip).  No device code here.
#pragma once
#include <stdint.h>

#include "../../include/msm377.h"

namespace msm377 {

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
constexpr uint32_t NARROW_BITS = 11;       // digit width of the small-input path ...
constexpr uint32_t NARROW_LOG = 11;        // ... whose windows have 2^11 buckets (the unsigned top digit needs the room: k_decompose_narrow)
constexpr uint32_t NARROW_SEG = 8;         // entries per accumulation work item on that path
constexpr uint32_t NARROW_WINDOWS = 23;    // 22 signed 11-bit windows + the top window from bit 242 on
// The same path in the even geometry (kernels/decompose.hpp k_decompose_geom; the default, MSM377_NARROW_EVEN): eleven signed
// 12-bit windows, then eleven unsigned 11-bit ones, 11 x 12 + 11 x 11 = 253 bits, 2^11 buckets each.
constexpr uint32_t NARROW_EVEN_WINDOWS = 22, NARROW_EVEN_SIGNED = 11, NARROW_EVEN_BITS = NARROW_LOG + 1;
static_assert(NARROW_EVEN_SIGNED * NARROW_EVEN_BITS + (NARROW_EVEN_WINDOWS - NARROW_EVEN_SIGNED) * NARROW_LOG == 253, "the windows cover a 253-bit scalar");
constexpr uint32_t MAX_WINDOW_SLOTS = NARROW_WINDOWS > MSM377_NUM_WINDOWS ? NARROW_WINDOWS : MSM377_NUM_WINDOWS;  // partial-record slots

// ---- sort geometry (kernels/sort.hpp) ----
// Sort keys: |d| in 0..32768 with the sign carried separately; coarse range = key / 128
// (256 ranges; the last one also owns key 32768).
constexpr uint32_t NRANGE = 256;
constexpr uint32_t KRANGE = NB / NRANGE;  // 128 keys per range
constexpr uint32_t KEY_TRACKED = 0x80000000u;  // key_max word: bit 31 = "measured", low bits = largest key; 0 = not measured
constexpr uint32_t SMALL_SORT_MAX = 1u << 16;            // most points k_small_sort holds in LDS (narrow-window path)
constexpr uint32_t SMALL_BINS_MAX = (1u << 12) + 1;      // keys 0 .. 2^L for L <= 12
struct SortElem {
  uint32_t idx_sign;  // point index | sign << 31
  uint32_t key;       // |d|
};

// ---- accumulation work list (kernels/accumulate.hpp) ----
struct WorkItem {
  uint32_t row;  // ws * NB + t   (bucket index t <-> key t + 1)
  uint32_t seg;  // entries [seg * seglen, seg * seglen + seglen) of the row, see row_split
};
// The CSR rows of a call whose points arrive in K chunks (host-buffer entry point, sequencer.hip run_sorted_upload): the
// sort files every key's entries by chunk, row_ptr holds K sub-row bounds per key -- entry (key, c) at index key K + c of
// a window's ((2^L + 1) K + 1) offsets -- and a launch walks ONE chunk's sub-rows.  K = 1, c = 0: the plain layout
// (2^L + 2 offsets per window, row t = key t + 1 at [t + 1, t + 2)).
struct RowView {
  uint32_t k = 1, c = 0;
};
constexpr uint32_t MAX_UPLOAD_CHUNKS = 8;
struct ChunkCuts {  // chunk j of the points = indices [cut[j], cut[j + 1]); cut[0] = 0, entries from k on = n
  uint32_t k = 1;
  uint32_t cut[MAX_UPLOAD_CHUNKS + 1] = {};
};
constexpr uint32_t META_BLOCK_WORDS = 2 * SEG_BINS + 4 + MSM377_NUM_WINDOWS;  // per pipeline part: work-list counters + key_max words
constexpr size_t SLOT_WORDS = (size_t)MAX_WINDOW_SLOTS * MSM377_G1_PARTIAL_POINTS * MSM377_G1_POINT_WORDS;  // per double-buffer slot of partial records

// ---- batched affine conversion (kernels/convert.hpp) ----
// K = 4 (twice the waves, twice the host's share) measured the same; workgroups of 128 / 64 threads (smaller trees, 2 / 4 times
// the host's share) stretch the conversion stage of a 2^20 MSM from 0.47 to 0.64 / 0.69 ms.
// (Round 3, build-time A/B of the conversion alone, msm377_g1_set_bases_device wall time at 2^20: AFF_K = 8: 0.401 ms,
// AFF_K = 4: 0.421, AFF_K = 8 with the point loops unrolled by 2: 0.408 -- gpurun_out/r03_conv.txt.)
constexpr uint32_t AFF_THREADS = 256, AFF_K = 8, AFF_BLOCK_POINTS = AFF_THREADS * AFF_K;
constexpr uint32_t AFF_STASH_WORDS = 52;  // N1, N2, Z, C (exclusive running product): 13 limbs each, 208 bytes per point
inline uint32_t affine_blocks(uint64_t n) { return (uint32_t)((n + AFF_BLOCK_POINTS - 1) / AFF_BLOCK_POINTS); }

// ---- wide windows over a precomputed table (kernels/wide.hpp) ----
// Whole MSMs on the 16-window path: windows 13, 14, 15 are 15 bits wide (kernels/decompose.hpp k_decompose, `even`).
constexpr uint32_t ACC_FLAG_WORD = 4;  // word of the pinned flag line (ctx->h_out_flag) that says "the accumulation kernel of call #seq is through"
constexpr uint32_t EVEN_FROM = 13;
constexpr uint32_t even_offset(uint32_t w) { return 16 * w - (w > EVEN_FROM ? w - EVEN_FROM : 0u); }
static_assert(even_offset(13) == 208 && even_offset(14) == 223 && even_offset(15) == 238 && even_offset(16) == 253, "13 x 16 + 3 x 15 = 253 bits");
constexpr uint32_t WIDE_BITS = MSM377_WIDE_WINDOW_BITS;  // the widest window: signed 20-bit digits
constexpr uint32_t WIDE_LOG = 19;        // 2^19 buckets, one set for all windows
constexpr uint32_t WIDE_WINDOWS = 13;    // windows of the table: [2^(wide_offset(w))] P_i
// Six 20-bit windows, then seven 19-bit ones: 6 x 20 + 7 x 19 = 253 bits, every window fills its key range (kernels/wide.hpp).
constexpr uint32_t WIDE_FULL = 6;
constexpr uint32_t wide_width(uint32_t w) { return w < WIDE_FULL ? WIDE_BITS : WIDE_BITS - 1; }
constexpr uint32_t wide_offset(uint32_t w) { return w <= WIDE_FULL ? WIDE_BITS * w : WIDE_BITS * WIDE_FULL + (WIDE_BITS - 1) * (w - WIDE_FULL); }
static_assert(wide_offset(WIDE_WINDOWS - 1) == 234 && wide_offset(WIDE_WINDOWS - 1) + wide_width(WIDE_WINDOWS - 1) == 253, "the windows cover a 253-bit scalar");
constexpr uint32_t WIDE_NRANGE = 4096;   // coarse sort ranges of KRANGE = 128 keys
constexpr uint32_t WIDE_POINTS = WIDE_LOG + 1;  // partial points of the single window record: bucket sum + 19 bit planes
static_assert((1u << WIDE_LOG) == WIDE_NRANGE * KRANGE, "ranges of KRANGE keys cover the wide bucket set");
static_assert((1u << WIDE_LOG) <= MSM377_NUM_WINDOWS * NB, "the wide bucket set fits the main path's bucket, row and work-list buffers");
static_assert(WIDE_POINTS * 1 <= MAX_WINDOW_SLOTS * MSM377_G1_PARTIAL_POINTS, "the wide window record fits a partial-record slot");

constexpr uint32_t G1_REC_WORDS = 32;     // a Weierstrass base record, 128 bytes (curves.hpp G1Dev::REC_WORDS)
constexpr uint32_t GLV_WINDOWS = MSM377_GLV_WINDOWS;

}  // namespace msm377
