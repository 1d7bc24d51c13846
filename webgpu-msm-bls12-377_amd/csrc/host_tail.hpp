// Host tail of an MSM (replaces the reference's CPU tail, src/submission/submission.ts:290-321): Horner over the window
// records the GPU leaves in pinned memory, one inversion, wire format -- on the context's helper threads (tail_pool.hpp).
// Also the host's share of the batched affine conversion (the inverses of the block products).  Host-only code.
#pragma once
#include <stdint.h>

struct msm377_ctx;

namespace msm377 {
namespace eng {

// Return values of the tails: TAIL_OK, TAIL_EXCEPTIONAL (an addition or doubling hit an exceptional case of the twisted
// Edwards law, fp64_host.hpp TeChecked; out_xy untouched -- the caller reruns on the Weierstrass path, exactly as for the
// GPU-side flag) or MSM377_EHIP (a helper thread did not answer within TailPool::wait_limit_ns; ctx->err says so).
constexpr int TAIL_OK = 0, TAIL_EXCEPTIONAL = 1;

int te_tail(msm377_ctx* ctx, const uint32_t* partials, uint8_t out_xy[96], int num_windows = 16, int cbits = 16, int planes = 15, int short_from = 0);
int xyzz_tail(msm377_ctx* ctx, const uint32_t* partials, uint8_t out_xy[96], int short_from = 0);
int ed_tail(msm377_ctx* ctx, const uint32_t* partials, uint8_t out_xy[64], int short_from = 0);  // Edwards-BLS12: a complete law, nothing to check

// Inverses of the block products [b0, b1) the conversion's way up left in ctx->h_aff_prod, into ctx->h_aff_inv
// (Montgomery's trick with one Fermat inversion per thread; results re-based to the device's Montgomery radix).
int invert_block_products_mt(msm377_ctx* ctx, uint32_t b0, uint32_t b1);
// The same for blocks [b0, b1) on the calling thread.
void invert_block_products(msm377_ctx* ctx, uint32_t b0, uint32_t b1);

}  // namespace eng
}  // namespace msm377
