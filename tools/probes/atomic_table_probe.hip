// Probe for a next-round design question: how fast is counting straight into ONE global hash table
// with device-scope atomics on MI355X, at the key distribution of BASELINE configs[1] (750 M windows; 88 % of them on
// ~8 k heavy k1-mers, the rest spread over ~4.9 M light ones), compared with the 25 ms of the partition-and-aggregate
// pipeline in csrc/count.hip?  Variants: (a) one CAS-insert + one atomicAdd per window, (b) the same after a per-block LDS
// pre-aggregation of 64 Ki windows.  Build: hipcc -O3 --offload-arch=gfx950 -o atomic_table_probe atomic_table_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__host__ __device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
// window i -> key id: heavy with probability 0.88 (uniform over n_hot), else light (uniform over n_cold)
__device__ inline uint64_t key_of(uint64_t i, uint32_t n_hot, uint32_t n_cold) {
  uint64_t h = mix(i * 0x9E3779B97F4A7C15ULL + 1);
  uint64_t id = (h % 100) < 88 ? (mix(h) % n_hot) : n_hot + (mix(h ^ 0x5555) % n_cold);
  return mix(id + 12345) | 1;          // the "k1-mer" (never 0)
}
__device__ inline void table_add(unsigned long long* keys, uint32_t* counts, uint64_t mask, uint64_t key, uint32_t c) {
  uint64_t s = mix(key) & mask;
  while (true) {
    unsigned long long cur = __hip_atomic_load(&keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0) { unsigned long long old = atomicCAS(&keys[s], 0ULL, (unsigned long long)key); cur = old == 0 ? key : old; }
    if (cur == key) { atomicAdd(&counts[s], c); return; }
    s = (s + 1) & mask;
  }
}
__global__ void direct_kernel(uint64_t n, uint32_t n_hot, uint32_t n_cold, unsigned long long* keys, uint32_t* counts, uint64_t mask) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    table_add(keys, counts, mask, key_of(i, n_hot, n_cold), 1);
}
#define LCAP 32768          // LDS slots: keys 256 KB is too much -> 16 Ki slots of (key 8 B + count 4 B) = 192 KB; use 8 Ki and spill
#define LSLOTS 8192
__global__ __launch_bounds__(1024) void preagg_kernel(uint64_t n, uint32_t n_hot, uint32_t n_cold, unsigned long long* keys, uint32_t* counts, uint64_t mask, uint32_t tile) {
  __shared__ unsigned long long lk[LSLOTS];
  __shared__ uint32_t lc[LSLOTS];
  for (uint64_t t0 = (uint64_t)blockIdx.x * tile; t0 < n; t0 += (uint64_t)gridDim.x * tile) {
    for (int j = threadIdx.x; j < LSLOTS; j += blockDim.x) { lk[j] = 0; lc[j] = 0; }
    __syncthreads();
    const uint64_t t1 = min(n, t0 + tile);
    for (uint64_t i = t0 + threadIdx.x; i < t1; i += blockDim.x) {
      const uint64_t key = key_of(i, n_hot, n_cold);
      uint32_t s = (uint32_t)(mix(key) >> 40) & (LSLOTS - 1);
      bool done = false;
      for (int probe = 0; probe < 8 && !done; probe++) {          // a few probes in LDS, then straight to the global table
        unsigned long long cur = lk[s];
        if (cur == 0) { unsigned long long old = atomicCAS(&lk[s], 0ULL, (unsigned long long)key); cur = old == 0 ? key : old; }
        if (cur == key) { atomicAdd(&lc[s], 1u); done = true; }
        s = (s + 1) & (LSLOTS - 1);
      }
      if (!done) table_add(keys, counts, mask, key, 1);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < LSLOTS; j += blockDim.x) if (lk[j]) table_add(keys, counts, mask, lk[j], lc[j]);
    __syncthreads();
  }
}
int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 750000000ULL;
  const uint32_t n_hot = 8192, n_cold = 4800000;
  const uint64_t slots = 1ULL << 24, mask = slots - 1;
  unsigned long long* keys; uint32_t* counts;
  CK(hipMalloc(&keys, slots * 8)); CK(hipMalloc(&counts, slots * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int variant = 0; variant < 3; variant++) {
    for (int rep = 0; rep < 2; rep++) {
      CK(hipMemset(keys, 0, slots * 8)); CK(hipMemset(counts, 0, slots * 4));
      CK(hipEventRecord(a));
      if (variant == 0) hipLaunchKernelGGL(direct_kernel, dim3(4096), dim3(256), 0, 0, n, n_hot, n_cold, keys, counts, mask);
      else hipLaunchKernelGGL(preagg_kernel, dim3(2048), dim3(1024), 0, 0, n, n_hot, n_cold, keys, counts, mask, variant == 1 ? 65536u : 262144u);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      std::vector<uint32_t> hc(slots);
      CK(hipMemcpy(hc.data(), counts, slots * 4, hipMemcpyDeviceToHost));
      uint64_t tot = 0, distinct = 0; for (uint32_t c : hc) { tot += c; distinct += c != 0; }
      printf("%s: %.2f ms for %llu windows (%.1f G windows/s); table holds %llu windows in %llu keys\n",
             variant == 0 ? "direct (CAS + atomicAdd per window)" : variant == 1 ? "LDS pre-aggregation, 64 Ki-window tiles" : "LDS pre-aggregation, 256 Ki-window tiles",
             ms, (unsigned long long)n, n / ms / 1e6, (unsigned long long)tot, (unsigned long long)distinct);
    }
  }
  return 0;
}
