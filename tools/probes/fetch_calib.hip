// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of this library (MI355X_MICROARCH.md, HBM: "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern"): over a 32 GB array,
//   stream16  every lane reads 16 consecutive bytes (the calibrated case: FETCH_SIZE reports half of the bytes)
//   stream8   every lane reads 8 consecutive bytes (a wavefront: 512 contiguous bytes)
//   gather8   every lane reads 8 bytes at a random 8-byte-aligned address (each load its own 64-byte sector)
//   gather64  every lane reads a whole random 64-byte line as 4 x 16 bytes
// build: hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip ; run under rocprofv3 --pmc FETCH_SIZE --output-format csv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__global__ void stream16(const ulonglong2* a, uint64_t n, unsigned long long* out) {
  unsigned long long s = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { ulonglong2 v = a[i]; s += v.x + v.y; }
  if (s == 12345) *out = s;
}
__global__ void stream8(const unsigned long long* a, uint64_t n, unsigned long long* out) {
  unsigned long long s = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 12345) *out = s;
}
__global__ void gather8(const unsigned long long* a, uint64_t n_words, uint64_t n_loads, unsigned long long* out) {
  unsigned long long s = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_loads; i += (uint64_t)gridDim.x * blockDim.x) s += a[mix(i) % n_words];
  if (s == 12345) *out = s;
}
__global__ void gather64(const ulonglong2* a, uint64_t n_lines, uint64_t n_loads, unsigned long long* out) {
  unsigned long long s = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_loads; i += (uint64_t)gridDim.x * blockDim.x) {
    const ulonglong2* p = a + (mix(i) % n_lines) * 4;
    ulonglong2 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
    s += v0.x + v1.y + v2.x + v3.y;
  }
  if (s == 12345) *out = s;
}
int main() {
  const uint64_t bytes = 32ULL << 30;
  void* a; unsigned long long* out;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 1, bytes);
  const uint64_t n_loads = 1ULL << 30;
  hipLaunchKernelGGL(stream16, dim3(8192), dim3(256), 0, 0, (const ulonglong2*)a, bytes / 16, out);
  hipLaunchKernelGGL(stream8, dim3(8192), dim3(256), 0, 0, (const unsigned long long*)a, bytes / 8, out);
  hipLaunchKernelGGL(gather8, dim3(8192), dim3(256), 0, 0, (const unsigned long long*)a, bytes / 8, n_loads, out);
  hipLaunchKernelGGL(gather64, dim3(8192), dim3(256), 0, 0, (const ulonglong2*)a, bytes / 64, n_loads, out);
  hipDeviceSynchronize();
  printf("stream16: %llu bytes read; stream8: %llu bytes; gather8: %llu loads of 8 bytes (%llu bytes of 64-byte sectors); gather64: %llu lines (%llu bytes)\n",
         (unsigned long long)bytes, (unsigned long long)bytes, (unsigned long long)n_loads, (unsigned long long)n_loads * 64, (unsigned long long)n_loads, (unsigned long long)n_loads * 64);
  return 0;
}
