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

// Stable scatter of one 8-bit digit.  A tile of 4096 items is split among the block's 4 wavefronts, 1024 consecutive items each,
// taken 64 at a time: the lanes of a wavefront that hold the same digit find each other with 8 ballots (one per digit bit), a
// lane's rank among them is a popcount, and a per-wavefront LDS counter carries the digit's count from one 64-item row to the next.
// After a barrier the counts of the wavefronts before it complete a position INSIDE THE TILE: the items go to the LDS in sorted
// order first, and the tile is written out from there -- consecutive threads write consecutive items of a digit's run, whole 128-byte
// pieces -- instead of 4096 separate 8-byte stores that left the L2 as partial lines (PMC: 2.5 times the bytes of a pass written).
// (The first version let thread d walk all 4096 digits of the tile to place the items of digit d: 1 M LDS reads per tile, 480 GB/s
// per pass.)
// VALS = false: the words alone (a value packed under the key's bits travels inside the word: shn_sort_keys)
template <bool VALS>
__global__ __launch_bounds__(SBLK) void sort_scatter_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                            uint64_t n, int shift, uint32_t nblocks,
                                                            const uint64_t* __restrict__ goff, uint64_t* __restrict__ ok,
                                                            uint32_t* __restrict__ ov) {
  __shared__ uint32_t cnt[SBLK / 64][256];
  __shared__ uint32_t base[SBLK / 64][256];
  __shared__ uint32_t dstart[256];
  __shared__ unsigned long long sk[STILE];
  __shared__ uint32_t sv[VALS ? STILE : 1];
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (uint32_t i = lane; i < 256; i += 64) cnt[wv][i] = 0;
  const uint64_t tile0 = (uint64_t)blockIdx.x * STILE;
  const uint64_t t0 = tile0 + (uint64_t)wv * (STILE / (SBLK / 64));
  const unsigned long long lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;          // lanes before this one
  constexpr int ROWS = STILE / SBLK;                                            // 16 rows of 64 items per wavefront
  uint64_t k[ROWS];
  uint32_t v[ROWS], rank[ROWS];
#pragma unroll
  for (int row = 0; row < ROWS; row++) {
    const uint64_t i = t0 + (uint64_t)row * 64 + lane;
    const bool have = i < n;
    k[row] = have ? keys[i] : 0;
    v[row] = (VALS && have) ? vals[i] : 0;
    const uint32_t d = (uint32_t)((k[row] >> shift) & 255);
    unsigned long long same = __ballot(have);
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
      const unsigned long long m = __ballot(((d >> bit) & 1) != 0);
      same &= ((d >> bit) & 1) ? m : ~m;
    }
    const uint32_t before = cnt[wv][d];                                       // (every lane of a digit group reads the same counter ...
    rank[row] = before + (uint32_t)__popcll(same & lt);
    if (have && (same & lt) == 0) cnt[wv][d] = before + (uint32_t)__popcll(same);   // ... and its first lane moves it on)
  }
  __syncthreads();
  // counts of the wavefronts before this one, per digit, and the tile's count of every digit (256 digits: every thread folds one)
  {
    const uint32_t d = threadIdx.x;
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < SBLK / 64; w++) { base[w][d] = run; run += cnt[w][d]; }
    dstart[d] = run;
  }
  __syncthreads();
  // exclusive scan of the 256 digit counts: the first wavefront, four digits a lane
  if (wv == 0) {
    uint32_t c0 = dstart[4 * lane], c1 = dstart[4 * lane + 1], c2 = dstart[4 * lane + 2], c3 = dstart[4 * lane + 3];
    const uint32_t mine = c0 + c1 + c2 + c3;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64); if ((int)lane >= off) incl += o; }
    uint32_t ex = incl - mine;
    dstart[4 * lane] = ex; ex += c0; dstart[4 * lane + 1] = ex; ex += c1; dstart[4 * lane + 2] = ex; ex += c2; dstart[4 * lane + 3] = ex;
  }
  __syncthreads();
#pragma unroll
  for (int row = 0; row < ROWS; row++) {
    const uint64_t i = t0 + (uint64_t)row * 64 + lane;
    if (i >= n) continue;
    const uint32_t d = (uint32_t)((k[row] >> shift) & 255);
    const uint32_t lp = dstart[d] + base[wv][d] + rank[row];
    sk[lp] = k[row];
    if (VALS) sv[lp] = v[row];
  }
  __syncthreads();
  const uint32_t n_tile = (uint32_t)(tile0 + STILE <= n ? STILE : (n > tile0 ? n - tile0 : 0));
  for (uint32_t i = threadIdx.x; i < n_tile; i += SBLK) {
    const uint64_t key = sk[i];
    const uint32_t d = (uint32_t)((key >> shift) & 255);
    const uint64_t p = goff[(uint64_t)d * nblocks + blockIdx.x] + (i - dstart[d]);
    ok[p] = key;
    if (VALS) ov[p] = sv[i];
  }
}

int shn_sort_pairs(shn_ctx* ctx, uint64_t* keys, uint32_t* vals, uint64_t* keys_tmp, uint32_t* vals_tmp, uint64_t n,
                   int bit_lo, int bit_hi) {
  if (n == 0) return SHN_OK;
  if (n >= 0xFFFFFFFFULL) return shn_fail(SHN_ERR_ARG, "shn_sort_pairs: n too large");
  hipStream_t s = ctx->stream; shn_use_stream(s);
  uint32_t nblocks = (uint32_t)cdiv(n, STILE);
  void* p;
  int rc = shn_ws(ctx)[8].get((size_t)256 * nblocks * 4 + ((size_t)256 * nblocks + 2) * 8, &p);
  if (rc) return rc;
  uint64_t* goff = (uint64_t*)p;
  uint32_t* gh = (uint32_t*)(goff + (size_t)256 * nblocks + 2);
  uint64_t *ki = keys, *ko = keys_tmp;
  uint32_t *vi = vals, *vo = vals_tmp;
  int passes = 0;
  for (int shift = bit_lo; shift < bit_hi; shift += 8) {
    hipLaunchKernelGGL(sort_hist_kernel, dim3(nblocks), dim3(SBLK), 0, s, ki, n, shift, nblocks, gh);
    if ((rc = shn_device_scan_u32(ctx, gh, (uint64_t)256 * nblocks, goff, nullptr))) return rc;
    hipLaunchKernelGGL(sort_scatter_kernel<true>, dim3(nblocks), dim3(SBLK), 0, s, ki, vi, n, shift, nblocks, goff, ko, vo);
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

// The same sort of 64-bit words by their bits [bit_lo, bit_hi) alone -- for pairs whose value fits under the key (r-mer << 32 |
// position: a pass moves 8 bytes per item each way instead of 12).  *sorted = whichever of the two buffers holds the result.
int shn_sort_keys(shn_ctx* ctx, uint64_t* keys, uint64_t* keys_tmp, uint64_t n, int bit_lo, int bit_hi, uint64_t** sorted) {
  *sorted = keys;
  if (n == 0) return SHN_OK;
  if (n >= 0xFFFFFFFFULL) return shn_fail(SHN_ERR_ARG, "shn_sort_keys: n too large");
  hipStream_t s = ctx->stream; shn_use_stream(s);
  uint32_t nblocks = (uint32_t)cdiv(n, STILE);
  void* p;
  int rc = shn_ws(ctx)[8].get((size_t)256 * nblocks * 4 + ((size_t)256 * nblocks + 2) * 8, &p);
  if (rc) return rc;
  uint64_t* goff = (uint64_t*)p;
  uint32_t* gh = (uint32_t*)(goff + (size_t)256 * nblocks + 2);
  uint64_t *ki = keys, *ko = keys_tmp;
  for (int shift = bit_lo; shift < bit_hi; shift += 8) {
    hipLaunchKernelGGL(sort_hist_kernel, dim3(nblocks), dim3(SBLK), 0, s, ki, n, shift, nblocks, gh);
    if ((rc = shn_device_scan_u32(ctx, gh, (uint64_t)256 * nblocks, goff, nullptr))) return rc;
    hipLaunchKernelGGL(sort_scatter_kernel<false>, dim3(nblocks), dim3(SBLK), 0, s, ki, (const uint32_t*)nullptr, n, shift, nblocks, goff, ko, (uint32_t*)nullptr);
    std::swap(ki, ko);
  }
  *sorted = ki;
  HIP_TRY(hipGetLastError());
  return SHN_OK;
}
