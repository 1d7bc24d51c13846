// N-API shim: the thin native layer between the reference's TypeScript host code and the
// C ABI (include/msm377.h).  It replaces the WebGPU dispatch path behind compute_msm
// (/root/reference/src/submission/submission.ts:113-288: get_device, the five stage drivers,
// read_from_gpu) with ONE native call; compute_msm.ts / compute_msm.js in this directory keep
// the entry point's name, arguments and result.
//
// Exports (CommonJS addon, N-API >= 6 for BigInt-free Buffers only -- BigInt conversion is done
// in JS):
//   computeMsm(points: Buffer, scalars: Buffer): Promise<Buffer>   96-byte x||y, runs off the JS thread
//   computeMsmSync(points: Buffer, scalars: Buffer): Buffer
//   computeEdMsmSync(points: Buffer 64n, scalars: Buffer 32n): Buffer   the Edwards-BLS12 twin (msm377_ed_msm), 64-byte x||y
//   setBasesSync(points: Buffer 96n): void                             fixed-base batches: keep a converted base set in HBM ...
//   fixedBaseMsmSync(scalars: Buffer 32n): Buffer                      ... and run MSMs of n <= its size against it
//   version(): string
// Errors reject / throw a JS Error carrying msm377_strerror + msm377_last_error, matching the
// reference's behaviour of throwing Error (cuzk/gpu.ts:7-10).
#define NAPI_VERSION 6
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>

#include "../../include/msm377.h"

namespace {

std::mutex g_mu;  // one call in flight per context (msm377.h)
msm377_ctx* g_ctx = nullptr;
uint64_t g_cap = 0;

int ensure_ctx(uint64_t n, std::string* err) {
  if (g_ctx && g_cap >= n) return MSM377_OK;
  if (g_ctx) {
    msm377_ctx_destroy(g_ctx);
    g_ctx = nullptr;
    g_cap = 0;
  }
  uint64_t cap = 1ull << 16;  // the harness's smallest case (README.md:90)
  while (cap < n) cap <<= 1;
  int device = 0;
  if (const char* d = getenv("MSM377_DEVICE")) device = atoi(d);
  int rc = msm377_ctx_create(device, cap, &g_ctx);
  if (rc) {
    *err = std::string("msm377_ctx_create: ") + msm377_strerror(rc) + " (no usable HIP device? there is no CPU fallback)";
    return rc;
  }
  g_cap = cap;
  return MSM377_OK;
}

int run(const uint8_t* points, const uint8_t* scalars, uint64_t n, uint8_t out[96], std::string* err) {
  std::lock_guard<std::mutex> lock(g_mu);
  int rc = ensure_ctx(n ? n : 1, err);
  if (rc) return rc;
  rc = msm377_g1_msm(g_ctx, points, scalars, n, out);
  if (rc) *err = std::string("msm377_g1_msm: ") + msm377_strerror(rc) + ": " + msm377_last_error(g_ctx);
  return rc;
}

bool get_buffers(napi_env env, napi_callback_info info, uint8_t** p, size_t* pl, uint8_t** s, size_t* sl, size_t point_bytes = 96) {
  size_t argc = 2;
  napi_value argv[2];
  if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < 2) {
    napi_throw_type_error(env, nullptr, "expected (points: Buffer, scalars: Buffer)");
    return false;
  }
  bool is_buf = false;
  for (int i = 0; i < 2; i++) {
    if (napi_is_buffer(env, argv[i], &is_buf) != napi_ok || !is_buf) {
      napi_throw_type_error(env, nullptr, "points and scalars must be Buffers");
      return false;
    }
  }
  napi_get_buffer_info(env, argv[0], reinterpret_cast<void**>(p), pl);
  napi_get_buffer_info(env, argv[1], reinterpret_cast<void**>(s), sl);
  if (*sl % 32 != 0 || *pl != (*sl / 32) * point_bytes) {
    napi_throw_range_error(env, nullptr, point_bytes == 96 ? "points must hold 96 bytes and scalars 32 bytes per input" : "points must hold 64 bytes and scalars 32 bytes per input");
    return false;
  }
  return true;
}

napi_value ComputeMsmSync(napi_env env, napi_callback_info info) {
  uint8_t *p, *s;
  size_t pl, sl;
  if (!get_buffers(env, info, &p, &pl, &s, &sl)) return nullptr;
  uint8_t out[96];
  std::string err;
  if (run(p, s, sl / 32, out, &err)) {
    napi_throw_error(env, nullptr, err.c_str());
    return nullptr;
  }
  napi_value buf;
  void* data;
  napi_create_buffer_copy(env, 96, out, &data, &buf);
  return buf;
}

// ---- the other entry points of the C ABI a TypeScript host may want (synchronous forms) ----
napi_value ComputeEdMsmSync(napi_env env, napi_callback_info info) {
  uint8_t *p, *s;
  size_t pl, sl;
  if (!get_buffers(env, info, &p, &pl, &s, &sl, 64)) return nullptr;
  uint8_t out[64];
  std::string err;
  int rc;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    rc = ensure_ctx(sl / 32 ? sl / 32 : 1, &err);
    if (!rc) {
      rc = msm377_ed_msm(g_ctx, p, s, sl / 32, out);
      if (rc) err = std::string("msm377_ed_msm: ") + msm377_strerror(rc) + ": " + msm377_last_error(g_ctx);
    }
  }
  if (rc) {
    napi_throw_error(env, nullptr, err.c_str());
    return nullptr;
  }
  napi_value buf;
  void* data;
  napi_create_buffer_copy(env, 64, out, &data, &buf);
  return buf;
}

bool one_buffer(napi_env env, napi_callback_info info, uint8_t** p, size_t* len, size_t unit, const char* what) {
  size_t argc = 1;
  napi_value argv[1];
  bool is_buf = false;
  if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < 1 || napi_is_buffer(env, argv[0], &is_buf) != napi_ok || !is_buf) {
    napi_throw_type_error(env, nullptr, what);
    return false;
  }
  napi_get_buffer_info(env, argv[0], reinterpret_cast<void**>(p), len);
  if (*len % unit) {
    napi_throw_range_error(env, nullptr, what);
    return false;
  }
  return true;
}

napi_value SetBasesSync(napi_env env, napi_callback_info info) {
  uint8_t* p;
  size_t len;
  if (!one_buffer(env, info, &p, &len, 96, "expected (points: Buffer of 96 bytes per point)")) return nullptr;
  std::string err;
  std::lock_guard<std::mutex> lock(g_mu);
  int rc = ensure_ctx(len / 96 ? len / 96 : 1, &err);
  if (!rc) {
    rc = msm377_g1_set_bases(g_ctx, p, len / 96);
    if (rc) err = std::string("msm377_g1_set_bases: ") + msm377_strerror(rc) + ": " + msm377_last_error(g_ctx);
  }
  if (rc) napi_throw_error(env, nullptr, err.c_str());
  return nullptr;
}

napi_value FixedBaseMsmSync(napi_env env, napi_callback_info info) {
  uint8_t* s;
  size_t len;
  if (!one_buffer(env, info, &s, &len, 32, "expected (scalars: Buffer of 32 bytes per scalar)")) return nullptr;
  uint8_t out[96];
  std::string err;
  int rc = MSM377_ESTATE;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_ctx || g_cap < len / 32) {
      err = "fixedBaseMsmSync: call setBasesSync with at least as many points first";
    } else {
      rc = msm377_g1_msm_fixed_base(g_ctx, s, len / 32, out);
      if (rc) err = std::string("msm377_g1_msm_fixed_base: ") + msm377_strerror(rc) + ": " + msm377_last_error(g_ctx);
    }
  }
  if (rc) {
    napi_throw_error(env, nullptr, err.c_str());
    return nullptr;
  }
  napi_value buf;
  void* data;
  napi_create_buffer_copy(env, 96, out, &data, &buf);
  return buf;
}

struct Job {
  napi_async_work work = nullptr;
  napi_deferred deferred = nullptr;
  napi_ref points_ref = nullptr, scalars_ref = nullptr;  // keep the caller's Buffers alive
  uint8_t *points = nullptr, *scalars = nullptr;
  uint64_t n = 0;
  uint8_t out[96];
  int rc = 0;
  std::string err;
};

void Execute(napi_env, void* data) {
  Job* j = static_cast<Job*>(data);
  j->rc = run(j->points, j->scalars, j->n, j->out, &j->err);
}

void Complete(napi_env env, napi_status, void* data) {
  Job* j = static_cast<Job*>(data);
  if (j->rc == 0) {
    napi_value buf;
    void* p;
    napi_create_buffer_copy(env, 96, j->out, &p, &buf);
    napi_resolve_deferred(env, j->deferred, buf);
  } else {
    napi_value msg, e;
    napi_create_string_utf8(env, j->err.c_str(), NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, nullptr, msg, &e);
    napi_reject_deferred(env, j->deferred, e);
  }
  napi_delete_reference(env, j->points_ref);
  napi_delete_reference(env, j->scalars_ref);
  napi_delete_async_work(env, j->work);
  delete j;
}

napi_value ComputeMsm(napi_env env, napi_callback_info info) {
  uint8_t *p, *s;
  size_t pl, sl;
  if (!get_buffers(env, info, &p, &pl, &s, &sl)) return nullptr;
  size_t argc = 2;
  napi_value argv[2];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  Job* j = new Job();
  j->points = p;
  j->scalars = s;
  j->n = sl / 32;
  napi_value promise, name;
  napi_create_promise(env, &j->deferred, &promise);
  napi_create_reference(env, argv[0], 1, &j->points_ref);
  napi_create_reference(env, argv[1], 1, &j->scalars_ref);
  napi_create_string_utf8(env, "msm377.computeMsm", NAPI_AUTO_LENGTH, &name);
  napi_create_async_work(env, nullptr, name, Execute, Complete, j, &j->work);
  napi_queue_async_work(env, j->work);
  return promise;
}

napi_value Version(napi_env env, napi_callback_info) {
  napi_value v;
  napi_create_string_utf8(env, msm377_version(), NAPI_AUTO_LENGTH, &v);
  return v;
}

napi_value Init(napi_env env, napi_value exports) {
  napi_property_descriptor props[] = {
      {"computeMsm", nullptr, ComputeMsm, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"computeMsmSync", nullptr, ComputeMsmSync, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"computeEdMsmSync", nullptr, ComputeEdMsmSync, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"setBasesSync", nullptr, SetBasesSync, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"fixedBaseMsmSync", nullptr, FixedBaseMsmSync, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"version", nullptr, Version, nullptr, nullptr, nullptr, napi_default, nullptr},
  };
  napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props);
  return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
