// Final merge on gfx950 (row a31 / SURVEY 8f row 1: process_concatenated_fasta.py:6-32, faster_reps.py:98-131): the two passes over
// the bases of every transcript -- "has this sequence, or its reverse complement, been seen before" and "where do the first / last
// 24-mers of the records occur" -- as kernels over the uploaded text.  The order-dependent decisions (names, first come first
// served, the containment rule) stay in csrc/post_host.hip, which calls this through post_dev.h when it is given a context.
// Byte work: HBM / L2 bound, no MFMA.
#include "post_dev.h"
#include "flatmap.h"
#include <algorithm>
#include <numeric>

struct PostDev {
  shn_ctx* ctx;
  uint8_t* d_text;
  uint64_t n_bytes;
};

int post_dev_create(shn_ctx* ctx, const std::vector<std::pair<const uint8_t*, uint64_t>>& pieces, PostDev** out) {
  if (!ctx || !out) return shn_fail(SHN_ERR_ARG, "post_dev_create: NULL argument");
  SHN_ENTER(ctx);
  uint64_t total = 0;
  for (auto& pc : pieces) total += pc.second;
  PostDev* d = new PostDev{ctx, nullptr, total};
  if (shn_dev_malloc(&d->d_text, total + 64) != hipSuccess) { delete d; return shn_fail(SHN_ERR_NOMEM, "post_dev_create: out of device memory"); }
  uint64_t at = 0;
  for (auto& pc : pieces) {
    if (!pc.second) continue;
    hipError_t e = hipMemcpyAsync(d->d_text + at, pc.first, pc.second, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { hipStreamSynchronize(ctx->stream); shn_dev_free(d->d_text); delete d; return shn_fail(SHN_ERR_HIP, std::string("post_dev_create: ") + hipGetErrorString(e)); }
    at += pc.second;
  }
  *out = d;
  return SHN_OK;
}

int post_dev_create_cap(shn_ctx* ctx, uint64_t cap, PostDev** out) {
  if (!ctx || !out) return shn_fail(SHN_ERR_ARG, "post_dev_create_cap: NULL argument");
  SHN_ENTER(ctx);
  PostDev* d = new PostDev{ctx, nullptr, cap};
  if (shn_dev_malloc(&d->d_text, cap + 64) != hipSuccess) { delete d; return shn_fail(SHN_ERR_NOMEM, "post_dev_create_cap: out of device memory"); }
  *out = d;
  return SHN_OK;
}
uint64_t post_dev_capacity(const PostDev* d) { return d ? d->n_bytes : 0; }
int post_dev_upload(PostDev* d, shn_ctx* on, uint64_t at, const uint8_t* src, uint64_t n) {
  if (!d || !on || (n && !src) || at + n > d->n_bytes) return shn_fail(SHN_ERR_ARG, "post_dev_upload: bad argument");
  if (!n) return SHN_OK;
  HIP_TRY(hipSetDevice(on->device));
  HIP_TRY(hipMemcpyAsync(d->d_text + at, src, n, hipMemcpyHostToDevice, on->stream));
  return SHN_OK;
}

void post_dev_destroy(PostDev* d) {
  if (!d) return;
  hipSetDevice(d->ctx->device);
  hipStreamSynchronize(d->ctx->stream);
  shn_dev_free(d->d_text);
  delete d;
}

// ---- fingerprints: a sum over the positions of a 64-bit mix of (position, byte) -- order-sensitive through the position, a sum so
// that the lanes of a wavefront take the bytes 64 at a time; two seeds = 128 bits.  The reverse complement's fingerprint comes out of
// the same pass: its byte at position L - 1 - i is the complement of byte i.
#define FP_S1 0x9E3779B97F4A7C15ULL
#define FP_S2 0xC2B2AE3D27D4EB4FULL
__host__ __device__ __forceinline__ uint64_t fp_term(uint64_t pos, uint32_t byte, uint64_t seed) { return shn_mix64(((pos << 8) | byte) ^ seed); }
__device__ __forceinline__ uint32_t comp_byte(uint32_t c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c; }

__global__ __launch_bounds__(64) void post_fp_kernel(const uint8_t* __restrict__ text, const uint64_t* __restrict__ off, const uint32_t* __restrict__ len,
                                                     uint64_t n, uint64_t* __restrict__ out, uint8_t* __restrict__ other) {
  for (uint64_t i = blockIdx.x; i < n; i += gridDim.x) {
    const uint8_t* s = text + off[i];
    const uint32_t L = len[i];
    uint64_t f1 = 0, f2 = 0, r1 = 0, r2 = 0;
    bool odd = false;                                                    // a byte outside ACGT (its complement is itself)
    for (uint32_t p = threadIdx.x; p < L; p += 64) {
      const uint32_t c = s[p];
      f1 += fp_term(p, c, FP_S1); f2 += fp_term(p, c, FP_S2);
      const uint32_t cc = comp_byte(c);
      odd |= cc == c;
      r1 += fp_term(L - 1 - p, cc, FP_S1); r2 += fp_term(L - 1 - p, cc, FP_S2);
    }
    for (int o = 32; o > 0; o >>= 1) { f1 += __shfl_xor(f1, o, 64); f2 += __shfl_xor(f2, o, 64); r1 += __shfl_xor(r1, o, 64); r2 += __shfl_xor(r2, o, 64); }
    if (threadIdx.x == 0) {
      const uint64_t l1 = shn_mix64((uint64_t)L ^ FP_S1), l2 = shn_mix64((uint64_t)L ^ FP_S2);
      out[4 * i] = f1 ^ l1; out[4 * i + 1] = f2 ^ l2; out[4 * i + 2] = r1 ^ l1; out[4 * i + 3] = r2 ^ l2;
    }
    const unsigned long long any_odd = __ballot(odd);
    if (other && threadIdx.x == 0) other[i] = any_odd ? 1 : 0;
  }
}

int post_dev_fingerprints(PostDev* d, const uint64_t* off, const uint32_t* len, uint64_t n, uint64_t* out, shn_ctx* on, uint8_t* other_out) {
  if (!d || (n && (!off || !len || !out))) return shn_fail(SHN_ERR_ARG, "post_dev_fingerprints: NULL argument");
  if (!n) return SHN_OK;
  HIP_TRY(hipSetDevice(d->ctx->device));
  hipStream_t s = on ? on->stream : d->ctx->stream;
  ShnDevBufs bufs(s);
  uint64_t *d_off, *d_out; uint32_t* d_len; uint8_t* d_other;
  if (bufs.get(&d_off, n * 8) != hipSuccess || bufs.get(&d_len, n * 4) != hipSuccess || bufs.get(&d_out, n * 32) != hipSuccess || bufs.get(&d_other, n + 1) != hipSuccess)
    return shn_fail(SHN_ERR_NOMEM, "post_dev_fingerprints: out of device memory");
  HIP_TRY(hipMemcpyAsync(d_off, off, n * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_len, len, n * 4, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(post_fp_kernel, dim3((uint32_t)std::min<uint64_t>(n, 1u << 20)), dim3(64), 0, s, d->d_text, d_off, d_len, n, d_out, d_other);
  HIP_TRY(hipMemcpyAsync(out, d_out, n * 32, hipMemcpyDeviceToHost, s));
  if (other_out) HIP_TRY(hipMemcpyAsync(other_out, d_other, n, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return SHN_OK;
}

// ---- occurrences of the query r-mers: one thread per chunk of SCAN_CHUNK window positions of a range, a rolling key, a probe of
// the query table (open addressing in global memory: a few MB, it stays in the L2); hits are appended through a counter
#define SCAN_CHUNK 128
#define Q_EMPTY 0xFFFFFFFFFFFFFFFFULL
__device__ __forceinline__ int acgt_code(uint32_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }

__global__ __launch_bounds__(256) void post_scan_kernel(const uint8_t* __restrict__ text, const uint64_t* __restrict__ off, const uint32_t* __restrict__ len,
                                                        const uint64_t* __restrict__ chunk_start, uint64_t n, uint64_t n_chunks, int r,
                                                        const uint64_t* __restrict__ qtab, uint64_t qmask, unsigned long long* __restrict__ counter,
                                                        uint64_t cap, uint32_t* __restrict__ h_range, uint32_t* __restrict__ h_pos,
                                                        uint64_t* __restrict__ h_key, uint32_t* __restrict__ bad) {
  const uint64_t kmask = (1ULL << (2 * r)) - 1;                        // (r < 32)
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_chunks; c += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t lo = 0, hi = n;                                           // the range of chunk c: last i with chunk_start[i] <= c
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (chunk_start[mid] <= c) lo = mid; else hi = mid; }
    const uint64_t i = lo;
    const uint8_t* s = text + off[i];
    const uint32_t L = len[i];
    const uint32_t p0 = (uint32_t)(c - chunk_start[i]) * SCAN_CHUNK;   // first window position of this chunk
    const uint32_t n_win = L >= (uint32_t)r ? L - r + 1 : 0;
    // every byte of the range is looked at by exactly one chunk (for the check of the alphabet): the chunk's window starts plus,
    // for the last chunk, the tail
    const uint32_t p1 = min(p0 + SCAN_CHUNK, n_win);
    uint64_t key = 0;
    int any_bad = 0;
    if (p0 < n_win) {
      for (int j = 0; j < r - 1; j++) { const int cd = acgt_code(s[p0 + j]); any_bad |= cd; key = (key << 2) | (uint64_t)(cd & 3); }
      for (uint32_t p = p0; p < p1; p++) {
        const int cd = acgt_code(s[p + r - 1]);
        any_bad |= cd;
        key = ((key << 2) | (uint64_t)(cd & 3)) & kmask;
        uint64_t sl = shn_mix64(key) & qmask;
        for (;;) {
          const uint64_t q = qtab[sl];
          if (q == Q_EMPTY) break;
          if (q == key) {
            const unsigned long long h = atomicAdd(counter, 1ULL);
            if (h < cap) { h_range[h] = (uint32_t)i; h_pos[h] = p; h_key[h] = key; }
            break;
          }
          sl = (sl + 1) & qmask;
        }
      }
    } else if (p0 == 0) {                                              // a range shorter than r: only the alphabet check
      for (uint32_t p = 0; p < L; p++) any_bad |= acgt_code(s[p]);
    }
    if (any_bad < 0) atomicOr(bad, 1u);
  }
}

int post_dev_scan(PostDev* d, const uint64_t* off, const uint32_t* len, uint64_t n, int r, const uint64_t* queries, uint64_t nq, uint64_t cap,
                  std::vector<uint32_t>& hit_range, std::vector<uint32_t>& hit_pos, std::vector<uint64_t>& hit_key, int* bad) {
  hit_range.clear(); hit_pos.clear(); hit_key.clear();
  if (bad) *bad = 0;
  if (!d || (n && (!off || !len)) || (nq && !queries) || r < 1 || r >= 32) return shn_fail(SHN_ERR_ARG, "post_dev_scan: bad argument");
  if (!n) return SHN_OK;
  hipStream_t s = d->ctx->stream;
  std::vector<uint64_t> cs(n + 1);
  uint64_t nc = 0;
  for (uint64_t i = 0; i < n; i++) {
    cs[i] = nc;
    const uint64_t nw = len[i] >= (uint32_t)r ? len[i] - r + 1 : 0;
    nc += std::max<uint64_t>(1, (nw + SCAN_CHUNK - 1) / SCAN_CHUNK);   // (a short range still gets a chunk: its bytes are checked)
  }
  cs[n] = nc;
  uint64_t qcap = 1024;
  while (qcap < nq * 2 + 16) qcap <<= 1;
  std::vector<uint64_t> qt(qcap, Q_EMPTY);
  for (uint64_t q = 0; q < nq; q++) {
    uint64_t sl = shn_mix64(queries[q]) & (qcap - 1);
    while (qt[sl] != Q_EMPTY && qt[sl] != queries[q]) sl = (sl + 1) & (qcap - 1);
    qt[sl] = queries[q];
  }
  ShnDevBufs bufs(s);
  uint64_t *d_off, *d_cs, *d_q, *d_hk; uint32_t *d_len, *d_hr, *d_hp, *d_bad; unsigned long long* d_cnt;
  if (bufs.get(&d_off, n * 8) != hipSuccess || bufs.get(&d_len, n * 4) != hipSuccess || bufs.get(&d_cs, (n + 1) * 8) != hipSuccess ||
      bufs.get(&d_q, qcap * 8) != hipSuccess || bufs.get(&d_hk, (cap + 1) * 8) != hipSuccess || bufs.get(&d_hr, (cap + 1) * 4) != hipSuccess ||
      bufs.get(&d_hp, (cap + 1) * 4) != hipSuccess || bufs.get(&d_bad, 64) != hipSuccess || bufs.get(&d_cnt, 64) != hipSuccess)
    return shn_fail(SHN_ERR_NOMEM, "post_dev_scan: out of device memory");
  HIP_TRY(hipMemcpyAsync(d_off, off, n * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_len, len, n * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_cs, cs.data(), (n + 1) * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_q, qt.data(), qcap * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(d_bad, 0, 4, s));
  HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, s));
  hipLaunchKernelGGL(post_scan_kernel, dim3((uint32_t)std::min<uint64_t>(cdiv(nc, 256), 1u << 20)), dim3(256), 0, s, d->d_text, d_off, d_len, d_cs, n, nc, r, d_q,
                     qcap - 1, d_cnt, cap, d_hr, d_hp, d_hk, d_bad);
  unsigned long long nh = 0;
  uint32_t b = 0;
  HIP_TRY(hipMemcpyAsync(&nh, d_cnt, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&b, d_bad, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  if (bad) *bad = b ? 1 : 0;
  if (nh > cap) return shn_fail(SHN_ERR_OVERFLOW, "post_dev_scan: more occurrences than the hit buffer holds");
  std::vector<uint32_t> hr(nh), hp(nh);
  std::vector<uint64_t> hk(nh);
  if (nh) {
    HIP_TRY(hipMemcpyAsync(hr.data(), d_hr, nh * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hp.data(), d_hp, nh * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hk.data(), d_hk, nh * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  // by (range, position): counting passes over 11 bits of range << bits(position) | position (the atomics of the scan hand the hits
  // over in no order; a comparison sort of the 1.1 M hits of bench.py --config 2p was 0.09 s of the merge's 0.5)
  uint32_t max_p = 0, max_r = 0;
  for (uint64_t j = 0; j < nh; j++) { max_p = std::max(max_p, hp[j]); max_r = std::max(max_r, hr[j]); }
  int bits_p = 1, bits_r = 1;
  while (bits_p < 32 && (max_p >> bits_p)) bits_p++;
  while (bits_r < 32 && (max_r >> bits_r)) bits_r++;
  std::vector<uint64_t> key(nh);
  for (uint64_t j = 0; j < nh; j++) key[j] = ((uint64_t)hr[j] << bits_p) | hp[j];
  std::vector<uint32_t> order(nh), order2(nh);
  std::iota(order.begin(), order.end(), 0u);
  std::vector<uint32_t> cnt(2049);
  for (int sh = 0; sh < bits_p + bits_r; sh += 11) {
    std::fill(cnt.begin(), cnt.end(), 0u);
    for (uint64_t j = 0; j < nh; j++) cnt[((key[j] >> sh) & 2047) + 1]++;
    for (int b2 = 0; b2 < 2048; b2++) cnt[b2 + 1] += cnt[b2];
    for (uint64_t j = 0; j < nh; j++) { const uint32_t e = order[j]; order2[cnt[(key[e] >> sh) & 2047]++] = e; }
    order.swap(order2);
  }
  hit_range.resize(nh); hit_pos.resize(nh); hit_key.resize(nh);
  for (uint64_t j = 0; j < nh; j++) { hit_range[j] = hr[order[j]]; hit_pos[j] = hp[order[j]]; hit_key[j] = hk[order[j]]; }
  return SHN_OK;
}
