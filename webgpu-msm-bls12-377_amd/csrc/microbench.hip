// Instruction- and primitive-level microbenchmarks for gfx950, used to choose the limb
// format and to price the bucket-accumulation kernel (results quoted in DESIGN.md).
// Not part of the product library.  Build: make -C webgpu-msm-bls12-377_amd/csrc microbench
//
//   ./microbench            prints one line per probe: name, Gop/s chip-wide, cycles per
//                           wave-instruction per SIMD (from s_memtime on one wave)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "g1_xyzz.hpp"

using namespace msm377;

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

constexpr int ILP = 8;

__global__ void __launch_bounds__(256) k_mad64(uint64_t* out, uint32_t b, int iters, uint64_t* cyc) {
  uint64_t acc[ILP];
  for (int k = 0; k < ILP; k++) acc[k] = threadIdx.x * 2654435761u + k;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < ILP; k++) acc[k] = (uint64_t)(uint32_t)acc[k] * b + acc[k];
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t s = 0;
  for (int k = 0; k < ILP; k++) s ^= acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ void __launch_bounds__(256) k_mullohi(uint64_t* out, uint32_t b, int iters, uint64_t* cyc) {
  uint32_t lo[ILP], hi[ILP];
  for (int k = 0; k < ILP; k++) { lo[k] = threadIdx.x * 2654435761u + k; hi[k] = k; }
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < ILP; k++) {
      uint32_t l = lo[k] * b;
      uint32_t h = __umulhi(lo[k], b);
      lo[k] = l + hi[k];
      hi[k] = h;
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t s = 0;
  for (int k = 0; k < ILP; k++) s ^= ((uint64_t)hi[k] << 32) | lo[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ void __launch_bounds__(256) k_mad24(uint64_t* out, uint32_t b, int iters, uint64_t* cyc) {
  uint32_t acc[ILP];
  for (int k = 0; k < ILP; k++) acc[k] = threadIdx.x * 2654435761u + k;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < ILP; k++) acc[k] = __umul24(acc[k], b) + acc[k];
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t s = 0;
  for (int k = 0; k < ILP; k++) s ^= acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ void __launch_bounds__(256) k_add32(uint64_t* out, uint32_t b, int iters, uint64_t* cyc) {
  uint32_t acc[ILP];
  for (int k = 0; k < ILP; k++) acc[k] = threadIdx.x * 2654435761u + k;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < ILP; k++) acc[k] = (acc[k] ^ b) + (acc[k] >> 3);
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t s = 0;
  for (int k = 0; k < ILP; k++) s ^= acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ void __launch_bounds__(256) k_dfma(uint64_t* out, double b, int iters, uint64_t* cyc) {
  double acc[ILP];
  for (int k = 0; k < ILP; k++) acc[k] = 1.0 + threadIdx.x * 1e-9 + k;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < ILP; k++) acc[k] = __builtin_fma(acc[k], b, acc[k]);
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int k = 0; k < ILP; k++) s += acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)__double_as_longlong(s);
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// Field / curve primitives: a dependent chain per thread, operands in registers.
__global__ void __launch_bounds__(256) k_femul(uint32_t* out, const uint32_t* in, int iters, uint64_t* cyc) {
  Fp::El a, b;
  const uint32_t* src = in + (threadIdx.x & 1) * 26;  // per-lane address: keeps operands in VGPRs
  for (int j = 0; j < 13; j++) { a.l[j] = src[j]; b.l[j] = src[13 + j]; }
  a.l[0] = (a.l[0] + threadIdx.x) & LMASK;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) a = Fp::mul(a, b);
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = 0; j < 13; j++) out[g * 13 + j] = a.l[j];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__global__ void __launch_bounds__(256) k_fesqr(uint32_t* out, const uint32_t* in, int iters, uint64_t* cyc) {
  Fp::El a;
  const uint32_t* src = in + (threadIdx.x & 1) * 26;
  for (int j = 0; j < 13; j++) a.l[j] = src[j];
  a.l[0] = (a.l[0] + threadIdx.x) & LMASK;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) a = Fp::sqr(a);
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = 0; j < 13; j++) out[g * 13 + j] = a.l[j];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__global__ void __launch_bounds__(256) k_feaddsub(uint32_t* out, const uint32_t* in, int iters, uint64_t* cyc) {
  Fp::El a, b;
  const uint32_t* src = in + (threadIdx.x & 1) * 26;
  for (int j = 0; j < 13; j++) { a.l[j] = src[j]; b.l[j] = src[13 + j]; }
  a.l[0] = (a.l[0] + threadIdx.x) & LMASK;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) { a = Fp::add(a, b); b = Fp::sub(b, a); }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = 0; j < 13; j++) out[g * 13 + j] = a.l[j] ^ b.l[j];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
// acc = acc + G repeatedly (G = generator): computes [iters+1]G, checked on the host.
template <int W>
__global__ void __launch_bounds__(256, W) k_madd(uint32_t* out, const uint32_t* in, int iters, uint64_t* cyc) {
  G1Affine g;
  const uint32_t* src = in + (threadIdx.x & 1) * 26;
  for (int j = 0; j < 13; j++) { g.x.l[j] = src[j]; g.y.l[j] = src[13 + j]; }
  G1XYZZ acc = g1_from_affine(g);
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) acc = g1_madd(acc, g);
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  size_t gi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = 0; j < 13; j++) {
    out[gi * 52 + j] = acc.x.l[j];
    out[gi * 52 + 13 + j] = acc.y.l[j];
    out[gi * 52 + 26 + j] = acc.zz.l[j];
    out[gi * 52 + 39 + j] = acc.zzz.l[j];
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

struct Probe {
  const char* name;
  double ops_per_thread;  // counted operations per thread
  double instr_per_thread;  // wave-instructions of the probed kind per thread (0 = n/a)
};

int main(int argc, char** argv) {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  printf("device: %s, CUs %d, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  const int cus = prop.multiProcessorCount;
  const int block = 256;
  uint64_t* d_out; uint64_t* d_cyc; uint32_t* d_out32; uint32_t* d_in;
  const int max_blocks = cus * 8;
  CK(hipMalloc(&d_out, sizeof(uint64_t) * max_blocks * block));
  CK(hipMalloc(&d_out32, sizeof(uint32_t) * 52 * max_blocks * block));
  CK(hipMalloc(&d_cyc, 8));
  CK(hipMalloc(&d_in, 52 * 4));
  uint32_t h_in[52];
  for (int j = 0; j < 13; j++) { h_in[j] = h_in[26 + j] = G1Consts::GEN_X[j]; h_in[13 + j] = h_in[39 + j] = G1Consts::GEN_Y[j]; }
  CK(hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  auto run = [&](const char* name, double ops_per_thread, int waves_per_simd, auto launch) {
    int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 wave per SIMD per block
    launch(blocks);  // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch(blocks);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    uint64_t cyc; CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
    double total = ops_per_thread * blocks * block;
    double gops = total / (ms * 1e-3) / 1e9;
    // memtime ticks at 100 MHz on gfx9-class parts; report both raw ticks and ns
    printf("%-10s waves/SIMD=%d  %8.3f ms  %10.2f Gop/s  memtime_ticks=%llu  ns/op/wave=%.2f\n", name,
           waves_per_simd, ms, gops, (unsigned long long)cyc,
           (ms * 1e6) / ops_per_thread / waves_per_simd);
  };

  const int it = 20000;
  for (int w : {1, 2, 4, 8}) {
    run("mad64", (double)it * ILP, w, [&](int b) { hipLaunchKernelGGL(k_mad64, dim3(b), dim3(block), 0, 0, d_out, 0x9e3779b9u, it, d_cyc); });
    run("mullohi", (double)it * ILP * 2, w, [&](int b) { hipLaunchKernelGGL(k_mullohi, dim3(b), dim3(block), 0, 0, d_out, 0x9e3779b9u, it, d_cyc); });
    run("mad24", (double)it * ILP, w, [&](int b) { hipLaunchKernelGGL(k_mad24, dim3(b), dim3(block), 0, 0, d_out, 0x9e3779u, it, d_cyc); });
    run("add32x3", (double)it * ILP * 3, w, [&](int b) { hipLaunchKernelGGL(k_add32, dim3(b), dim3(block), 0, 0, d_out, 0x9e3779b9u, it, d_cyc); });
    run("dfma", (double)it * ILP, w, [&](int b) { hipLaunchKernelGGL(k_dfma, dim3(b), dim3(block), 0, 0, d_out, 1.0000001, it, d_cyc); });
  }
  const int fit = 2000;
  for (int w : {1, 2, 3, 4}) {
    run("fe_mul", (double)fit, w, [&](int b) { hipLaunchKernelGGL(k_femul, dim3(b), dim3(block), 0, 0, d_out32, d_in, fit, d_cyc); });
    run("fe_sqr", (double)fit, w, [&](int b) { hipLaunchKernelGGL(k_fesqr, dim3(b), dim3(block), 0, 0, d_out32, d_in, fit, d_cyc); });
    run("fe_addsub", (double)fit * 2, w, [&](int b) { hipLaunchKernelGGL(k_feaddsub, dim3(b), dim3(block), 0, 0, d_out32, d_in, fit, d_cyc); });
  }
  const int mit = 500;
  for (int w : {1, 2}) {
    run("g1_madd<2>", (double)mit, w, [&](int b) { hipLaunchKernelGGL(k_madd<2>, dim3(b), dim3(block), 0, 0, d_out32, d_in, mit, d_cyc); });
  }
  for (int w : {1, 2, 3}) {
    run("g1_madd<3>", (double)mit, w, [&](int b) { hipLaunchKernelGGL(k_madd<3>, dim3(b), dim3(block), 0, 0, d_out32, d_in, mit, d_cyc); });
  }
  for (int w : {1, 2, 3, 4}) {
    run("g1_madd<4>", (double)mit, w, [&](int b) { hipLaunchKernelGGL(k_madd<4>, dim3(b), dim3(block), 0, 0, d_out32, d_in, mit, d_cyc); });
  }

  // Correctness: device chains against the same header compiled for the host.
  int bad = 0;
  {
    hipLaunchKernelGGL(k_femul, dim3(1), dim3(64), 0, 0, d_out32, d_in, 100, d_cyc);
    std::vector<uint32_t> h(64 * 13);
    CK(hipMemcpy(h.data(), d_out32, h.size() * 4, hipMemcpyDeviceToHost));
    for (int t = 0; t < 64; t++) {
      Fp::El a, b;
      for (int j = 0; j < 13; j++) { a.l[j] = h_in[j]; b.l[j] = h_in[13 + j]; }
      a.l[0] = (a.l[0] + t) & LMASK;
      for (int i = 0; i < 100; i++) a = Fp::mul(a, b);
      for (int j = 0; j < 13; j++) if (a.l[j] != h[t * 13 + j]) bad++;
    }
    hipLaunchKernelGGL(k_fesqr, dim3(1), dim3(64), 0, 0, d_out32, d_in, 100, d_cyc);
    CK(hipMemcpy(h.data(), d_out32, h.size() * 4, hipMemcpyDeviceToHost));
    for (int t = 0; t < 64; t++) {
      Fp::El a;
      for (int j = 0; j < 13; j++) a.l[j] = h_in[j];
      a.l[0] = (a.l[0] + t) & LMASK;
      for (int i = 0; i < 100; i++) a = Fp::sqr(a);
      for (int j = 0; j < 13; j++) if (a.l[j] != h[t * 13 + j]) bad++;
    }
    hipLaunchKernelGGL(k_feaddsub, dim3(1), dim3(64), 0, 0, d_out32, d_in, 100, d_cyc);
    CK(hipMemcpy(h.data(), d_out32, h.size() * 4, hipMemcpyDeviceToHost));
    for (int t = 0; t < 64; t++) {
      Fp::El a, b;
      for (int j = 0; j < 13; j++) { a.l[j] = h_in[j]; b.l[j] = h_in[13 + j]; }
      a.l[0] = (a.l[0] + t) & LMASK;
      for (int i = 0; i < 100; i++) { a = Fp::add(a, b); b = Fp::sub(b, a); }
      for (int j = 0; j < 13; j++) if ((a.l[j] ^ b.l[j]) != h[t * 13 + j]) bad++;
    }
    hipLaunchKernelGGL(k_madd<2>, dim3(1), dim3(64), 0, 0, d_out32, d_in, 37, d_cyc);
    std::vector<uint32_t> hp(64 * 52);
    CK(hipMemcpy(hp.data(), d_out32, hp.size() * 4, hipMemcpyDeviceToHost));
    G1Affine g;
    g.x = Fp::from_const(G1Consts::GEN_X);
    g.y = Fp::from_const(G1Consts::GEN_Y);
    G1XYZZ acc = g1_from_affine(g);
    for (int i = 0; i < 37; i++) acc = g1_madd(acc, g);
    for (int j = 0; j < 13; j++) {
      if (acc.x.l[j] != hp[j] || acc.y.l[j] != hp[13 + j] || acc.zz.l[j] != hp[26 + j] || acc.zzz.l[j] != hp[39 + j]) bad++;
    }
  }
  printf("device-vs-host mismatches: %d\n", bad);
  return bad ? 2 : 0;
}
