// Does hipMemsetAsync / hipMemcpyAsync handle sizes >= 4 GiB on this stack?  (development probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void count_ne(const uint64_t* p, uint64_t n, uint64_t v, unsigned long long* out) {
  unsigned long long c = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) c += p[i] != v;
  if (c) atomicAdd(out, c);
}
int main() {
  const uint64_t n = (12ULL << 30) / 8;
  uint64_t *a, *b; unsigned long long *d, h;
  hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&d, 8);
  hipMemset(a, 0x11, n * 8); hipMemset(b, 0x11, n * 8);      // whatever these do, then:
  count_ne<<<4096, 256>>>(a, n, 0x1111111111111111ULL, d);   // warm
  for (int async = 0; async < 2; async++) {
    hipMemset(d, 0, 8);
    if (async) hipMemsetAsync(a, 0xFF, n * 8, 0); else hipMemset(a, 0xFF, n * 8);
    count_ne<<<4096, 256>>>(a, n, ~0ULL, d);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("memset%s 12 GiB: %llu words wrong of %llu\n", async ? "Async" : "", h, (unsigned long long)n);
    hipMemset(a, 0x11, 1 << 20); hipMemset(a + n / 2, 0x22, 1 << 20); hipMemset(a + n - (1 << 17), 0x33, 1 << 20);
  }
  hipMemset(d, 0, 8);
  hipMemsetAsync(a, 0xAB, n * 8, 0);
  hipMemcpyAsync(b, a, n * 8, hipMemcpyDeviceToDevice, 0);
  count_ne<<<4096, 256>>>(b, n, 0xABABABABABABABABULL, d);
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("memcpyAsync D2D 12 GiB: %llu words wrong\n", h);
  return 0;
}
