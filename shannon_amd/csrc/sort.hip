// Stable LSD radix sort of (u64 key, u32 value) pairs, 8-bit digits.  Used for the seed order of
// the contig extension (weight desc, k1-mer asc) and other host-order-defining sorts; sizes are
// the distinct-k-mer set, not the read set, so this is not a roofline kernel.
#include "common.h"

#define SBLK 256
#define STILE 4096

__global__ __launch_bounds__(SBLK) void sort_hist_kernel(const uint64_t* __restrict__ keys, uint64_t n, int shift,
                                                         uint32_t nblocks, uint32_t* __restrict__ gh) {
  __shared__ uint32_t lh[256];
  lh[threadIdx.x] = 0;
  __syncthreads();
  uint64_t t0 = (uint64_t)blockIdx.x * STILE, t1 = min(t0 + STILE, n);
  for (uint64_t i = t0 + threadIdx.x; i < t1; i += SBLK) atomicAdd(&lh[(keys[i] >> shift) & 255], 1u);
  __syncthreads();
  gh[(uint64_t)threadIdx.x * nblocks + blockIdx.x] = lh[threadIdx.x];
}

__global__ __launch_bounds__(SBLK) void sort_scatter_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                            uint64_t n, int shift, uint32_t nblocks,
                                                            const uint64_t* __restrict__ goff, uint64_t* __restrict__ ok,
                                                            uint32_t* __restrict__ ov) {
  __shared__ uint32_t dig[STILE / 4];    // 4 digits per word
  __shared__ uint32_t pos[STILE];
  uint64_t t0 = (uint64_t)blockIdx.x * STILE, t1 = min(t0 + STILE, n);
  uint32_t cnt = (uint32_t)(t1 - t0);
  uint8_t* d8 = (uint8_t*)dig;
  for (uint32_t i = threadIdx.x; i < STILE; i += SBLK) d8[i] = i < cnt ? (uint8_t)((keys[t0 + i] >> shift) & 255) : 0;
  __syncthreads();
  // thread d assigns, in tile order, the output slots of the items whose digit is d (stable)
  {
    const uint32_t d = threadIdx.x;
    uint64_t base = goff[(uint64_t)d * nblocks + blockIdx.x];
    uint32_t run = 0;
    uint32_t words = (cnt + 3) / 4;
    for (uint32_t w = 0; w < words; w++) {
      uint32_t x = dig[w];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint32_t i = w * 4 + j;
        if (((x >> (8 * j)) & 255) == d && i < cnt) { pos[i] = (uint32_t)(base - t0 * 0 + run); run++; }
      }
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < cnt; i += SBLK) {
    uint32_t p = pos[i];
    ok[p] = keys[t0 + i];
    ov[p] = vals[t0 + i];
  }
}

int shn_sort_pairs(shn_ctx* ctx, uint64_t* keys, uint32_t* vals, uint64_t* keys_tmp, uint32_t* vals_tmp, uint64_t n,
                   int bit_lo, int bit_hi) {
  if (n == 0) return SHN_OK;
  if (n >= 0xFFFFFFFFULL) return shn_fail(SHN_ERR_ARG, "shn_sort_pairs: n too large");
  hipStream_t s = ctx->stream;
  uint32_t nblocks = (uint32_t)cdiv(n, STILE);
  void* p;
  int rc = g_shn_ws[8].get((size_t)256 * nblocks * 4 + ((size_t)256 * nblocks + 2) * 8, &p);
  if (rc) return rc;
  uint64_t* goff = (uint64_t*)p;
  uint32_t* gh = (uint32_t*)(goff + (size_t)256 * nblocks + 2);
  uint64_t *ki = keys, *ko = keys_tmp;
  uint32_t *vi = vals, *vo = vals_tmp;
  int passes = 0;
  for (int shift = bit_lo; shift < bit_hi; shift += 8) {
    hipLaunchKernelGGL(sort_hist_kernel, dim3(nblocks), dim3(SBLK), 0, s, ki, n, shift, nblocks, gh);
    if ((rc = shn_device_scan_u32(ctx, gh, (uint64_t)256 * nblocks, goff, nullptr))) return rc;
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(nblocks), dim3(SBLK), 0, s, ki, vi, n, shift, nblocks, goff, ko, vo);
    std::swap(ki, ko);
    std::swap(vi, vo);
    passes++;
  }
  if (passes & 1) {   // result is in the tmp buffers: copy back
    HIP_TRY(hipMemcpyAsync(keys, keys_tmp, n * 8, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(vals, vals_tmp, n * 4, hipMemcpyDeviceToDevice, s));
  }
  HIP_TRY(hipGetLastError());
  return SHN_OK;
}
