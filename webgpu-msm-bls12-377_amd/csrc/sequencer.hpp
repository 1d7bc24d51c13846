// Stage sequencer: everything that enqueues GPU work for an entry point of include/msm377.h.  capi.hip validates nothing
// and forwards here; the functions below carry the entry point's name without its msm377_ prefix and its exact
// arguments and return codes.  Kernels: kernels/*.hpp (compiled into sequencer.hip only); host tail: host_tail.hpp.
#pragma once
#include <stdint.h>

struct msm377_ctx;

namespace msm377 {
namespace eng {

int g1_msm_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint8_t out_xy[96]);
int g1_msm(msm377_ctx* ctx, const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out_xy[96]);
int ed_msm_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint8_t out_xy[64]);
int ed_msm(msm377_ctx* ctx, const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out_xy[64]);
int ed_generate_bases_device(msm377_ctx* ctx, uint64_t seed, uint64_t n, void* d_points_out);
int g1_set_bases_device(msm377_ctx* ctx, const void* d_points, uint64_t n);
int g1_set_bases(msm377_ctx* ctx, const uint8_t* points, uint64_t n);
int g1_set_bases_precomputed_device(msm377_ctx* ctx, const void* d_points, uint64_t n);
int g1_set_bases_precomputed(msm377_ctx* ctx, const uint8_t* points, uint64_t n);
int g1_msm_fixed_base_device(msm377_ctx* ctx, const void* d_scalars, uint64_t n, uint8_t out_xy[96]);
int g1_msm_fixed_base_batch_device(msm377_ctx* ctx, const void* d_scalars, uint64_t n, uint32_t batch, uint8_t* out_xy);
int g1_msm_fixed_base(msm377_ctx* ctx, const uint8_t* scalars, uint64_t n, uint8_t out_xy[96]);
int window_partials(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint32_t win_begin, uint32_t win_count, uint8_t* host_out, void* dev_out);
int g1_glv_window_partials_device(msm377_ctx* ctx, const void* d_points, const void* d_scalars, uint64_t n, uint32_t win_begin, uint32_t win_count, uint8_t* partials_out);
int g1_generate_bases_device(msm377_ctx* ctx, uint64_t seed, uint64_t n, void* d_points_out);

int reserve_host_staging(msm377_ctx* ctx);

// Shared with capi.hip (argument checks of the host-only entry points, the stage read-back).
bool hip_ok(msm377_ctx* ctx, int hip_error, const char* what);

}  // namespace eng
}  // namespace msm377
