// Host tail: see host_tail.hpp.  Compiled by hipcc like the rest of the library (context.hpp names HIP types), no device code.
#include "host_tail.hpp"

#include <memory>

#include "context.hpp"

namespace msm377 {
namespace eng {

namespace {

int worker_timeout(msm377_ctx* ctx) {
  ctx->err = "host tail: a helper thread did not finish its job in time (MSM377_TAIL_WAIT_MS)";
  return MSM377_EHIP;
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
// Everything a worker touches lives in a shared State: if a worker misses the deadline the caller leaves, and a late
// worker still writes into live memory.
template <class Pt, class Extra>
struct TailState {
  static constexpr int MAXC = TailPool::WORKERS + 1;
  int bounds[MAXC + 1];
  Pt part[MAXC];
  Extra extra;  // per-piece scratch the piece function needs (the Edwards form's exceptional-case flags)
  struct alignas(128) Mark {
    int64_t t0, t1;
    int cpu;
  } mark[MAXC];
};

template <class Pt, class Extra, class PieceFn, class AddFn>
int tail_horner_mt(msm377_ctx* ctx, PieceFn piece, AddFn add, int positions, Pt* out, std::shared_ptr<TailState<Pt, Extra>> st) {
  constexpr int MAXC = TailPool::WORKERS + 1;
  TailPool& pool = ctx->tail_pool;
  const int want = std::max(1, std::min(ctx->tail_threads, MAXC));
  if (want > 1) pool.start();
  int worker_of[MAXC];  // helper threads that are free to take a piece (one that still holds a job it never started is not)
  const int helpers = want > 1 ? pool.idle_workers(worker_of, want - 1) : 0;
  const int used = tail_split(positions, helpers + 1, st->bounds);
  // MSM377_TAIL_TRACE=1: per-piece start / end (us after the call) and CPU, on stderr
  const bool trace = ctx->tail_trace;
  const int64_t t_call = trace ? TailPool::now_ns() : 0;
  auto chain = [st, trace, piece](int k) {
    if (trace) st->mark[k].t0 = TailPool::now_ns(), st->mark[k].cpu = sched_getcpu();
    st->part[k] = piece(*st, st->bounds[k], st->bounds[k + 1], k);
    if (trace) st->mark[k].t1 = TailPool::now_ns();
  };
  // pieces are shares (tail_pool.hpp): a piece whose helper has not touched it when the caller gets to it runs here
  auto shares = std::make_shared<TailPool::Shares>();
  for (int k = 0; k + 1 < used; k++) pool.post_share(worker_of[k], shares, k, [chain, k] { chain(k); });
  chain(used - 1);  // the top piece: the fewest positions, the most doublings
  Pt acc = st->part[used - 1];
  for (int k = used - 2; k >= 0; k--) {
    if (!pool.finish_share(shares, k, [chain, k] { chain(k); })) return worker_timeout(ctx);
    acc = add(*st, acc, st->part[k], used - 1);
  }
  if (trace) {
    fprintf(stderr, "tail trace:");
    for (int k = 0; k < used; k++)
      fprintf(stderr, "  [%d cpu %d: %.1f..%.1f]", k, st->mark[k].cpu, (st->mark[k].t0 - t_call) / 1e3, (st->mark[k].t1 - t_call) / 1e3);
    fprintf(stderr, "  done %.1f us\n", (TailPool::now_ns() - t_call) / 1e3);
  }
  *out = acc;
  return TAIL_OK;
}

struct TeFlags {
  TeChecked chk[TailPool::WORKERS + 1];  // one per piece
};
struct NoExtra {};

inline bool single_threaded(const msm377_ctx* ctx) { return ctx->tail_threads <= 1 || ctx->tail_pool.poisoned.load(); }

}  // namespace

int te_tail(msm377_ctx* ctx, const uint32_t* partials, uint8_t out_xy[96], int num_windows, int cbits, int planes, int short_from) {
  if (single_threaded(ctx) || num_windows < 8) return teh_combine(partials, num_windows, out_xy, cbits, planes, short_from) ? TAIL_EXCEPTIONAL : TAIL_OK;
  using St = TailState<TeH::Ext, TeFlags>;
  auto st = std::make_shared<St>();
  TeH::Ext r;
  const int rc = tail_horner_mt<TeH::Ext, TeFlags>(
      ctx, [partials, cbits, planes, short_from](St& s, int lo, int hi, int k) { return teh_tail_piece(partials, lo, hi, s.extra.chk[k], cbits, planes, short_from); },
      [](St& s, const TeH::Ext& a, const TeH::Ext& b, int k) { return s.extra.chk[k].add(a, b); }, tail_positions(num_windows, cbits, short_from), &r, st);
  if (rc) return rc;
  for (const TeChecked& c : st->extra.chk)
    if (c.bad) return TAIL_EXCEPTIONAL;
  teh_to_wire(r, out_xy);
  return TAIL_OK;
}

int xyzz_tail(msm377_ctx* ctx, const uint32_t* partials, uint8_t out_xy[96], int short_from) {
  if (single_threaded(ctx)) {
    g1h_combine(partials, MSM377_NUM_WINDOWS, out_xy, short_from);
    return TAIL_OK;
  }
  using St = TailState<G1H::XYZZ, NoExtra>;
  G1H::XYZZ r;
  const int rc = tail_horner_mt<G1H::XYZZ, NoExtra>(
      ctx,
      [partials, short_from](St&, int lo, int hi, int) {
        G1H::XYZZ acc = g1h_horner_bits(partials, lo, hi, 0, short_from);
        for (int i = 0; i < lo; i++) acc = G1H::dbl(acc);
        return acc;
      },
      [](St&, const G1H::XYZZ& a, const G1H::XYZZ& b, int) { return G1H::add(a, b); }, tail_positions(MSM377_NUM_WINDOWS, 16, short_from), &r, std::make_shared<St>());
  if (rc) return rc;
  g1h_to_wire(r, out_xy);
  return TAIL_OK;
}

int ed_tail(msm377_ctx* ctx, const uint32_t* partials, uint8_t out_xy[64], int short_from) {
  if (single_threaded(ctx)) {
    edh_combine(partials, out_xy, short_from);
    return TAIL_OK;
  }
  using St = TailState<EdH::Ext, NoExtra>;
  EdH::Ext r;
  const int rc = tail_horner_mt<EdH::Ext, NoExtra>(
      ctx,
      [partials, short_from](St&, int lo, int hi, int) {
        EdH::Ext acc = edh_horner_bits(partials, lo, hi, short_from);
        for (int i = 0; i + 1 < lo; i++) acc = EdH::dbl_nt(acc);
        if (lo > 0) acc = EdH::dbl(acc);
        return acc;
      },
      [](St&, const EdH::Ext& a, const EdH::Ext& b, int) { return EdH::add(a, b); }, tail_positions(MSM377_NUM_WINDOWS, 16, short_from), &r, std::make_shared<St>());
  if (rc) return rc;
  edh_to_wire(r, out_xy);
  return TAIL_OK;
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

int invert_block_products_mt(msm377_ctx* ctx, uint32_t b0, uint32_t b1) {
  const uint32_t nblk = b1 > b0 ? b1 - b0 : 0;
  if (nblk >= 32 && !single_threaded(ctx)) {
    TailPool& pool = ctx->tail_pool;
    pool.start();
    int worker_of[TailPool::WORKERS + 1];
    const int parts = 1 + pool.idle_workers(worker_of, std::min(ctx->tail_threads, TailPool::WORKERS + 1) - 1);
    const uint32_t per = (nblk + parts - 1) / parts;
    // shares (tail_pool.hpp): the caller takes over a range whose helper has not started on it.  A helper that comes back
    // late finds its share claimed and touches nothing of this call (ctx->aff_scratch, h_aff_inv belong to the next one by then).
    auto shares = std::make_shared<TailPool::Shares>();
    auto range = [ctx, per, b0, b1](int k) { invert_block_products(ctx, std::min(b1, b0 + (uint32_t)(k + 1) * per), std::min(b1, b0 + (uint32_t)(k + 2) * per)); };
    for (int k = 0; k + 1 < parts; k++) pool.post_share(worker_of[k], shares, k, [range, k] { range(k); });
    invert_block_products(ctx, b0, std::min(b1, b0 + per));
    for (int k = 0; k + 1 < parts; k++)
      if (!pool.finish_share(shares, k, [range, k] { range(k); })) return worker_timeout(ctx);
  } else {
    invert_block_products(ctx, b0, b1);
  }
  return MSM377_OK;
}

}  // namespace eng
}  // namespace msm377
