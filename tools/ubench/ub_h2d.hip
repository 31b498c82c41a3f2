// H2D bandwidth from pinned memory: one stream / three streams, 32 MB pieces (the size of an 8K picture's work lists)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
int main() {
  const size_t N = 32u << 20; const int R = 60;
  char *h[3], *d[3]; hipStream_t s[3];
  for (int i = 0; i < 3; i++) { hipHostMalloc((void**)&h[i], N, hipHostMallocDefault); hipMalloc((void**)&d[i], N); hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); memset(h[i], i, N); }
  for (int ns = 1; ns <= 3; ns += 2) {
    for (int i = 0; i < 6; i++) hipMemcpyAsync(d[i % ns], h[i % ns], N, hipMemcpyHostToDevice, s[i % ns]);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < R; i++) hipMemcpyAsync(d[i % ns], h[i % ns], N, hipMemcpyHostToDevice, s[i % ns]);
    hipDeviceSynchronize();
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("%d stream(s): %.3f ms per 32 MB copy, %.1f GB/s\n", ns, ms / R, R * (double)N / ms / 1e6);
  }
  return 0;
}
