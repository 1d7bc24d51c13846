// Stage 0: wire points -> Montgomery base records (k_convert_bases, the batched affine conversion k_affine_up / k_affine_down,
// the GLV table) and small utilities (k_clear_words).  Replaces wgsl/cuzk/convert_point_coords_and_decompose_scalars.template.wgsl:41-99
// + barrett.template.wgsl:60-82 of the reference.
// Device code; included by sequencer.hip only.
#pragma once
#include "../curves.hpp"

namespace msm377 {
namespace {

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
    // The lazy field forms of field29.hpp (round 3; bounds replayed by tools/check_lazy_bounds.py conversion_formulas):
    // products without the conditional subtraction, sums limb-wise.  ~630 of the ~3 700 instructions per point less, and
    // none of the 13-step borrow chains of the canonical add / sub, which is what two waves per SIMD could not hide.
    const Fp::El xr = Fp::from_words<12>(w), yr = Fp::from_words<12>(w + 12);                      // canonical
    const Fp::El u = Fp::add_lz(Fp::mul_lz(xr, Fp::from_const(K::TE_SR)), Fp::from_const(K::TE_S));  // limbs < 2^30
    const Fp::El v = Fp::mul_lz(yr, Fp::from_const(K::TE_SR));                                      // < p + 2^354
    // c u is the one factor that has to come back below p + 2^354 (n1 multiplies it by the limb-wise u + 1)
    const Fp::El cu = Fp::csub(Fp::norm(Fp::add_lz(Fp::mul_lz(xr, Fp::from_const(K::TE_CSR)), Fp::from_const(K::TE_CS))), K::MOD);
    const Fp::El up = Fp::add_lz(u, Fp::one());                                                     // limbs < 3 * 2^29
    z = Fp::mul_lz(up, v);
    n1 = Fp::mul_lz(up, cu);
    n2 = Fp::norm(Fp::add_kp_sub_sub2(z, K::KP4W3, Fp::zero(), v));  // (u - 1) v = (u + 1) v - 2 v (+ 4p): N-form below 5p + 2^354
    return Te377::is_zero_mod_p(z);
  }
};
struct AffDoublingSource {  // [2^bits] of the point in an affine record of the previous window's table (precomputed-window tables)
  const uint32_t* prev;
  uint32_t bits;  // window width of the table: 16, or 20 for the wide-window table
  __device__ __forceinline__ bool load(uint64_t i, Fp::El& n1, Fp::El& n2, Fp::El& z) const {
    Te377::Ext p = Te377::from_base_affine(TeAffBase::load_base(prev, (uint32_t)i), false);
    bool bad = false;
#pragma unroll 1
    for (uint32_t k = 0; k < bits; k++) {
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
// Stash layout (round 3): 16-byte piece k of point (blk, j, tid) at piece index ((blk AFF_K + j) 13 + k) AFF_THREADS + tid, so
// that every store (and every load on the way down) of a wave is 1 KB of contiguous memory.  Round 2 kept a point's
// 208 bytes together: each of a wave's 13 stores then touched 64 different lines, and whichever sort kernel ran beside
// the conversion queued behind those requests (k_decompose 16 -> 107 us, k_local_sort 84 -> 224 us).
template <class SRC>
__global__ void __launch_bounds__(AFF_THREADS, 4) k_affine_up(SRC src, uint64_t n, uint32_t* __restrict__ stash,
                                                              uint32_t* __restrict__ trees, uint32_t* __restrict__ block_prod, uint32_t* __restrict__ host_flag, uint32_t* __restrict__ dev_count,
                                                              int* __restrict__ err, uint32_t prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);  // sequencer.hip: the conversion is the front end's critical path
  // block_prod and host_flag live in pinned, coherent HOST memory: the host polls the flag and starts inverting the
  // moment the last workgroup has delivered (a D2H copy queued behind this kernel took 60 us to get through beside the
  // sort, and an event wait adds its wake-up latency on top).  Workgroups count themselves in DEVICE memory -- a
  // system-scope atomic on host memory is a PCIe round trip each, 0.6 ms for 512 of them -- and the last one raises
  // the flag with a plain store.
  using K = G1Consts;
  __shared__ uint32_t tree[2 * AFF_THREADS][13];
  const uint32_t tid = threadIdx.x, blk = blockIdx.x;
  const uint64_t base = (uint64_t)blk * AFF_BLOCK_POINTS + tid;
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
    uint4* dst = reinterpret_cast<uint4*>(stash) + ((size_t)blk * AFF_K + j) * (AFF_STASH_WORDS / 4) * AFF_THREADS + tid;
#pragma unroll
    for (int k = 0; k < (int)AFF_STASH_WORDS / 4; k++) dst[(size_t)k * AFF_THREADS] = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
    c = Fp::mul_lz(c, z);
  }
  if (bad) atomicOr(err, ERR_TE_CONVERT);
  put13(tree[AFF_THREADS + tid], c);
  for (uint32_t size = AFF_THREADS / 2; size >= 1; size >>= 1) {
    __syncthreads();
    if (tid < size) put13(tree[size + tid], Fp::mul_lz(get13(tree[2 * (size + tid)]), get13(tree[2 * (size + tid) + 1])));
  }
  __syncthreads();
  uint32_t* out = trees + (size_t)blk * (2 * AFF_THREADS * 13);
  const uint32_t* flat = &tree[0][0];
  for (uint32_t k = tid; k < 2 * AFF_THREADS * 13; k += AFF_THREADS) out[k] = flat[k];
  if (tid == 0) {  // the root in the host's field format (radix 2^384), like the partial records
    const Fp::El root = Fp::mul(get13(tree[1]), Fp::from_const(K::TO64));
    uint32_t w[12];
    Fp::to_words<12>(root, w);
#pragma unroll
    for (int j = 0; j < 12; j++) block_prod[(size_t)blk * 12 + j] = w[j];
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
                                                                const uint32_t* __restrict__ block_inv, uint32_t* __restrict__ bases, uint32_t prio) {
  if (prio) __builtin_amdgcn_s_setprio(3);  // sequencer.hip: the conversion is the front end's critical path
  __shared__ uint32_t tree[2 * AFF_THREADS][13];
  const uint32_t tid = threadIdx.x, blk = blockIdx.x;
  const uint32_t* in = trees + (size_t)blk * (2 * AFF_THREADS * 13);
  uint32_t* flat = &tree[0][0];
  for (uint32_t k = tid; k < 2 * AFF_THREADS * 13; k += AFF_THREADS) flat[k] = in[k];
  __syncthreads();
  if (tid == 0) {
    uint32_t w[12];
#pragma unroll
    for (int j = 0; j < 12; j++) w[j] = block_inv[(size_t)blk * 12 + j];
    put13(tree[1], Fp::from_words<12>(w));
  }
  // downwards: a node's slot turns from the product of its leaves into the inverse of that product
  for (uint32_t size = 1; size < AFF_THREADS; size <<= 1) {
    __syncthreads();
    if (tid < size) {
      const uint32_t k = size + tid;
      const Fp::El inv_k = get13(tree[k]), a = get13(tree[2 * k]), b = get13(tree[2 * k + 1]);
      put13(tree[2 * k], Fp::mul_lz(inv_k, b));
      put13(tree[2 * k + 1], Fp::mul_lz(inv_k, a));
    }
  }
  __syncthreads();
  Fp::El inv = get13(tree[AFF_THREADS + tid]);  // 1 / (the product of this thread's Z's)
  const uint64_t base = (uint64_t)blk * AFF_BLOCK_POINTS + tid;
#pragma unroll 1
  for (int j = (int)AFF_K - 1; j >= 0; j--) {
    const uint64_t i = base + (uint64_t)j * AFF_THREADS;
    if (i >= n) continue;
    uint32_t w[AFF_STASH_WORDS];
    {
      const uint4* src = reinterpret_cast<const uint4*>(stash) + ((size_t)blk * AFF_K + j) * (AFF_STASH_WORDS / 4) * AFF_THREADS + tid;
#pragma unroll
      for (int k = 0; k < (int)AFF_STASH_WORDS / 4; k++) {
        const uint4 v = src[(size_t)k * AFF_THREADS];
        w[4 * k + 0] = v.x;
        w[4 * k + 1] = v.y;
        w[4 * k + 2] = v.z;
        w[4 * k + 3] = v.w;
      }
    }
    const Fp::El zi = Fp::mul_lz(inv, get13(w + 39));  // 1 / Z_j = (1 / C_j) C_(j-1)
    inv = Fp::mul_lz(inv, get13(w + 26));              // 1 / C_(j-1)
    const Fp::El x = Fp::mul(get13(w), zi), y = Fp::mul(get13(w + 13), zi);  // canonical: y -+ x are stored canonical
    // 2d x y stays a lazy product (< p + 2^354): te377.hpp madd_affine takes it as a factor or subtracts it limb-wise from 2p
    const Fp::El ymx = Fp::sub(y, x), ypx = Fp::add(y, x), kt = Fp::mul_lz(Fp::mul_lz(x, y), Fp::from_const(G1Consts::TE_2D));
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


}  // namespace
}  // namespace msm377
