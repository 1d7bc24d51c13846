// Probe (not product): ways to move a 128 MB pageable host buffer to the device.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t N = 160u << 20;
  std::vector<unsigned char> src(N, 1);
  void* d; CK(hipMalloc(&d, N));
  void* pin; CK(hipHostMalloc(&pin, N));
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int rep = 0; rep < 3; rep++) {
    double t0 = now();
    CK(hipMemcpy(d, src.data(), N, hipMemcpyHostToDevice));
    double t1 = now();
    CK(hipHostRegister(src.data(), N, hipHostRegisterDefault));
    double t2 = now();
    CK(hipMemcpyAsync(d, src.data(), N, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
    double t3 = now();
    CK(hipHostUnregister(src.data()));
    double t4 = now();
    for (int threads : {1, 4, 8, 16}) {
      double a = now();
      const size_t chunk = 8u << 20;
      size_t nchunks = (N + chunk - 1) / chunk;
      // pipelined: copy chunk c into pinned by `threads` threads, then async DMA
      for (size_t c = 0; c < nchunks; c++) {
        size_t off = c * chunk, len = std::min(chunk, N - off);
        std::vector<std::thread> th;
        size_t per = (len + threads - 1) / threads;
        for (int t = 0; t < threads; t++) {
          size_t o = t * per; if (o >= len) break;
          size_t l = std::min(per, len - o);
          th.emplace_back([&, o, l] { memcpy((char*)pin + off + o, src.data() + off + o, l); });
        }
        for (auto& x : th) x.join();
        CK(hipMemcpyAsync((char*)d + off, (char*)pin + off, len, hipMemcpyHostToDevice, st));
      }
      CK(hipStreamSynchronize(st));
      printf("  staged threads=%d: %.2f ms\n", threads, now() - a);
    }
    printf("rep %d: pageable %.2f ms | register %.2f + copy %.2f + unregister %.2f ms\n", rep, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
  }
  return 0;
}
