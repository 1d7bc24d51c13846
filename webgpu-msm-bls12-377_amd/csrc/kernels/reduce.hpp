// Stage 4: bucket reduction (k_tree_step, k_tree_step_quad, k_reduce_tail, k_fold_windows), the quad-cooperative additions
// and the output stage k_gather_partials.  Replaces wgsl/cuzk/bpr.template.wgsl:69-173; models cuzk/bpr.ts:5-126.
// Device code; included by sequencer.hip only.
#pragma once
#include "../curves.hpp"

namespace msm377 {
namespace {

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
// A bucket record read by a lane quad: lane q fetches coordinate q (one 64-byte slot) and the four lanes exchange what they
// got -- a quarter of the memory traffic of four lanes each reading the whole record (the merge of split rows moved
// 150 MB that way at 2^16 points, 90 us for 135 k additions).  All four lanes of the quad must be here.
template <class CV>
__device__ __forceinline__ typename CV::Pt load_record_quad(const uint32_t* __restrict__ p, uint32_t q) {
  constexpr uint32_t NL = CV::NL;
  typename CV::F::El mine;
  const uint32_t* slot = p + q * CV::COORD_WORDS;
  const uint4* s4 = reinterpret_cast<const uint4*>(slot);
#pragma unroll
  for (uint32_t k = 0; k < NL / 4; k++) {
    const uint4 v = s4[k];
    mine.l[4 * k + 0] = v.x;
    mine.l[4 * k + 1] = v.y;
    mine.l[4 * k + 2] = v.z;
    mine.l[4 * k + 3] = v.w;
  }
  mine.l[NL - 1] = slot[NL - 1];
  const typename CV::F::El c0 = quad_bcast<0>(mine), c1 = quad_bcast<1>(mine), c2 = quad_bcast<2>(mine), c3 = quad_bcast<3>(mine);
  uint32_t w[CV::PT_WORDS];
#pragma unroll
  for (uint32_t j = 0; j < NL; j++) {
    w[j] = c0.l[j];
    w[NL + j] = c1.l[j];
    w[2 * NL + j] = c2.l[j];
    w[3 * NL + j] = c3.l[j];
  }
  return CV::from_words(w);
}
template <class CV>
__device__ __forceinline__ typename CV::Pt load_bucket_quad(const uint32_t* __restrict__ b, uint32_t L, uint32_t ws, uint32_t t, uint32_t q) {
  return load_record_quad<CV>(bucket_ptr<CV>(b, L, ws, t), q);
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
  const typename CV::Pt sum = add_quad(load_bucket_quad<CV>(buckets, L, ws, x, q), load_bucket_quad<CV>(buckets, L, ws, y, q), q);
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
                                             uint32_t* __restrict__ out, uint32_t* __restrict__ host_out, uint32_t pp = MSM377_G1_PARTIAL_POINTS) {
  const uint32_t x = pt == 0 ? 0u : (1u << (pt - 1));
  typename CV::F::El v;
#pragma unroll
  for (uint32_t j = 0; j < CV::NL; j++) v.l[j] = bucket_ptr<CV>(buckets, L, ws, x)[coord * CV::COORD_WORDS + j];
  v = CV::F::mul(v, CV::to64());
  uint32_t w[CV::NW32];
  CV::F::template to_words<CV::NW32>(v, w);
  if (pt == 0 && coord == 0) w[CV::NW32 - 1] |= CV::RECORD_TAG;  // the record names its coordinate system (values are < 2^377: the bit is free)
  const size_t at = ((size_t)(ws * pp + pt) * 4 + coord) * CV::NW32;  // pp points per window record: 16, or WIDE_POINTS for the one wide window
#pragma unroll
  for (uint32_t j = 0; j < CV::NW32; j++) out[at + j] = w[j];
  if (host_out) {
#pragma unroll
    for (uint32_t j = 0; j < CV::NW32; j++) host_out[at + j] = w[j];
  }
}
// The block that finishes last (a device-memory counter) hands the call over to the host: error word, then the sequence
// number the host is polling for (wait_zero_copy_out).  Call with every store of the block issued; all threads.
// INVARIANT: the host returns to its caller the moment it sees the sequence number -- while this kernel is still
// finishing (the counter reset below, the grid's last waves) and before the stream's completion event fires.  So
// nothing in or behind this kernel may touch a buffer that the host or another stream reuses for the next call: it reads
// the buckets and writes the partial records, the error word and its own counter, all of which the next call touches
// only in main-stream order, and it must stay the last launch of a call.  (The next call's base conversion rewrites
// d_bases from the side stream without waiting for the main stream -- fine only because of this.)
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
      const typename CV::Pt sum = add_quad(load_bucket_quad<CV>(buckets, L, ws, x, q), load_bucket_quad<CV>(buckets, L, ws, y, q), q);
      bad |= CV::is_bad(sum);
      store_coord<CV>(bucket_ptr<CV>(buckets, L, ws, x) + q * CV::COORD_WORDS, coord4(q, sum).l);
    }
    __syncthreads();  // workgroup-scope fence + barrier: the next level reads what this one wrote
  }
  if (bad) atomicOr(err, ERR_TE_TREE);
}

// The same with the workgroup's stretch of the bucket array in LDS.  At level `first` every list -- and the running
// block -- is R = NB >> first buckets long and everything a workgroup touches from there on lies inside its R buckets
// (a list only shrinks; the lists the running block spawns are its own upper halves), so the workgroup reads them once
// (R x 208 bytes, 52 KB for R = 256: records packed to their 52 words, which also spreads 16 quads' reads over the
// banks), runs all remaining levels out of LDS -- a level is then one quad addition plus a barrier instead of that plus
// an L2 round trip for loads and stores -- and writes back only the heads the gather kernel reads (B[0] and B[2^l]).
template <class CV>
__device__ __forceinline__ typename CV::Pt lds_point(const uint32_t* p) {
  uint32_t w[CV::PT_WORDS];
  const uint4* s = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (uint32_t k = 0; k < CV::PT_WORDS / 4; k++) {
    const uint4 v = s[k];
    w[4 * k + 0] = v.x;
    w[4 * k + 1] = v.y;
    w[4 * k + 2] = v.z;
    w[4 * k + 3] = v.w;
  }
  return CV::from_words(w);
}
constexpr uint32_t TAIL_LDS_BYTES_MAX = 160 * 1024;  // a workgroup may take all of a CU's LDS on gfx950 (beyond 64 KB: hipFuncSetAttribute, sequencer.hip)
template <class CV>
__global__ void __launch_bounds__(TAIL_THREADS, 1) k_reduce_tail_lds(uint32_t* __restrict__ buckets, uint32_t L, uint32_t first, int* __restrict__ err) {
  extern __shared__ uint4 tail_lds4[];
  uint32_t* lds = reinterpret_cast<uint32_t*>(tail_lds4);
  constexpr uint32_t PW = CV::PT_WORDS, NL = CV::NL;
  static_assert(PW % 4 == 0 && PW == 4 * NL, "a packed record is four coordinates of NL limbs, a multiple of 16 bytes");
  const uint32_t NB = 1u << L, R = NB >> first;
  const uint32_t ws = blockIdx.y, job = blockIdx.x;  // job < first: list `job`; job == first: the running block
  const uint32_t q = threadIdx.x & 3, quad = threadIdx.x >> 2;
  const uint32_t base = job < first ? (NB >> (job + 1)) : 0u;  // first bucket of this workgroup's stretch
  for (uint32_t i = threadIdx.x; i < R * 4; i += TAIL_THREADS) {  // (bucket, coordinate)
    const uint32_t* src = bucket_ptr<CV>(buckets, L, ws, base + (i >> 2)) + (i & 3) * CV::COORD_WORDS;
    uint32_t* dst = lds + (size_t)i * NL;
#pragma unroll
    for (uint32_t j = 0; j < NL; j++) dst[j] = src[j];
  }
  __syncthreads();
  bool bad = false;
  for (uint32_t r = first; r < L; r++) {
    const uint32_t half = NB >> (r + 1);
    const uint32_t nlists = job < first ? 1u : 1u + (r - first);
    for (uint32_t op = quad; op < nlists * half; op += TAIL_THREADS / 4) {
      const uint32_t li = op / half, kk = op % half;
      const uint32_t lo = (job < first || li == 0) ? 0u : (NB >> (first + li));  // inside the stretch
      const uint32_t x = lo + kk, y = x + half;
      const typename CV::Pt sum = add_quad(lds_point<CV>(lds + (size_t)x * PW), lds_point<CV>(lds + (size_t)y * PW), q);
      bad |= CV::is_bad(sum);
      const typename CV::F::El c = coord4(q, sum);
      uint32_t* d = lds + (size_t)x * PW + q * NL;  // the quad's four lanes have read x before any of them writes (one wave, in order)
#pragma unroll
      for (uint32_t j = 0; j < NL; j++) d[j] = c.l[j];
    }
    __syncthreads();
  }
  if (bad) atomicOr(err, ERR_TE_TREE);
  // heads: bucket 0 of the stretch, and for the running block the lists it spawned, at 2^k for k < L - first
  const uint32_t heads = job < first ? 1u : 1u + (L - first);
  if (threadIdx.x < heads * 4) {
    const uint32_t h = threadIdx.x >> 2, c = threadIdx.x & 3;
    const uint32_t idx = h == 0 ? 0u : (1u << (h - 1));
    store_coord<CV>(bucket_ptr<CV>(buckets, L, ws, base + idx) + c * CV::COORD_WORDS, lds + (size_t)idx * PW + c * NL);
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
                                                      uint32_t* dev_count = nullptr, const int* d_err = nullptr, uint32_t seq = 0, uint32_t pp = MSM377_G1_PARTIAL_POINTS) {
  const uint32_t g = blockIdx.x * 64 + threadIdx.x;
  const uint32_t coord = g & 3, pt = (g >> 2) % pp, ws = g / (4 * pp);
  // narrow windows have fewer bit planes; the host tail never reads the unused points
  if (g < wc * pp * 4 && pt <= L) pack_partial<CV>(buckets, L, ws, pt, coord, out, host_out, pp);
  if (host_out) publish_to_host(gridDim.x, host_flag, dev_count, d_err, seq);  // a kernel argument: uniform
}

}  // namespace
}  // namespace msm377
