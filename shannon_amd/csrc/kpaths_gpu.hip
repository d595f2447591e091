// known_paths, the part every read goes through (mbgraph.py:1355-1388, known_paths -> search_sequence at :114-160): a read whose
// first K-mer occurs in node n at offset o and whose text equals the node's from there lies inside that node if it ends before
// the node does -- its only path is [n], it adds no known edge and no known path; otherwise it runs on into the node's
// successors and has to be searched.  Nearly every read is of the first kind (96 % at BASELINE configs[2]).  The host stage
// used to index every K-mer of every node (a 600 k-item sort per partition: 7 of the graph stage's 31 thread-seconds per
// step), look up every read's first and last K-mer on the device and compare every read with its node on host threads
// (4 thread-seconds).  Here the whole test runs where the reads already are:
//   kp_insert     one thread per base position of the concatenated node texts: its K-mer into an open-addressing table, a count
//                 per K-mer;  scan;  kp_fill: the positions of every K-mer side by side
//   kp_classify   per read: first / last K-mer looked up; the occurrences of the first K-mer by ascending position (= the order of
//                 the host's index: node order, then offset), text compared base by base against the node
//                 -> 0 (nothing to do), 1 + node (inside that node), 2 (search on the host), 3 + node + offset (search on the host
//                 from the K-mer's only occurrence)
// (a first version sorted the (K-mer, position) items: 35 launches per partition against 5)
// The host keeps the sequential part: reads of kind 2 and 3, in read order; those of kind 2 (rare: a K-mer occurs once in a de Bruijn
// graph until bridging copies nodes) against an index of just their first K-mers.
#include "common.h"

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>

namespace {

__device__ __forceinline__ int kp_code(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
__device__ __forceinline__ uint32_t kp_node_of(const uint64_t* __restrict__ off, uint32_t n_nodes, uint64_t p) {   // largest i with off[i] <= p
  uint32_t lo = 0, hi = n_nodes;
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= p) lo = mid; else hi = mid; }
  return lo;
}

constexpr uint64_t KP_EMPTY = ~0ULL;                             // (a K-mer of K <= 31 bases is < 2^62)
__device__ __forceinline__ uint64_t kp_mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

// one thread per base position of the concatenated node texts: the K-mer starting there (if it lies inside its node) enters an
// open-addressing table; slot_of[p] = its slot (the K-mer's group), cnt[slot] = occurrences
__global__ void kp_insert(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ off, uint32_t n_nodes, uint64_t total, int K,
                          unsigned long long* hkeys, uint64_t mask, uint32_t* __restrict__ cnt, uint32_t* __restrict__ slot_of,
                          unsigned long long* __restrict__ counters) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const uint32_t nd = kp_node_of(off, n_nodes, p);
  uint32_t slot = 0xFFFFFFFFu;
  if (p + (uint64_t)K <= off[nd + 1]) {
    uint64_t k = 0;
    bool ok = true;
    for (int j = 0; j < K; j++) { const int c = kp_code(bases[p + j]); if (c < 0) { ok = false; break; } k = (k << 2) | (uint64_t)c; }
    if (!ok) atomicAdd(&counters[1], 1ULL);
    else {
      uint64_t s = kp_mix(k) & mask;
      while (true) {
        unsigned long long cur = hkeys[s];
        if (cur == KP_EMPTY) { const unsigned long long old = atomicCAS(&hkeys[s], KP_EMPTY, (unsigned long long)k); cur = old == KP_EMPTY ? k : old; }
        if (cur == k) break;
        s = (s + 1) & mask;
      }
      slot = (uint32_t)s;
      atomicAdd(&cnt[s], 1u);
    }
  }
  slot_of[p] = slot;
}
// the occurrences of every K-mer side by side (in any order: the reader takes them by ascending position)
__global__ void kp_fill(const uint32_t* __restrict__ slot_of, uint64_t total, const uint64_t* __restrict__ goff, uint32_t* __restrict__ fill,
                        uint32_t* __restrict__ occ) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const uint32_t s = slot_of[p];
  if (s == 0xFFFFFFFFu) return;
  occ[goff[s] + atomicAdd(&fill[s], 1u)] = (uint32_t)p;
}

struct RView { const uint64_t* words; const uint64_t* woff; const uint32_t* len; uint64_t n; uint32_t fixed_len, wpr; };

__device__ __forceinline__ int64_t kp_find(const unsigned long long* __restrict__ hkeys, uint64_t mask, uint64_t key) {
  uint64_t s = kp_mix(key) & mask;
  while (true) {
    const unsigned long long cur = hkeys[s];
    if (cur == key) return (int64_t)s;
    if (cur == KP_EMPTY) return -1;
    s = (s + 1) & mask;
  }
}

__global__ void kp_classify(RView v, int K, const unsigned long long* __restrict__ hkeys, uint64_t mask, const uint64_t* __restrict__ goff,
                            const uint32_t* __restrict__ occ, const uint8_t* __restrict__ bases, const uint64_t* __restrict__ off, uint32_t n_nodes,
                            uint8_t* __restrict__ state, int32_t* __restrict__ node_out, uint32_t* __restrict__ off_out) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= v.n) return;
  const uint32_t L = v.len ? v.len[r] : v.fixed_len;
  uint8_t st = 0;
  int32_t fn = -1;
  uint32_t fo = 0;
  if (L >= (uint32_t)K) {
    const uint64_t* w = v.words + (v.woff ? v.woff[r] : r * v.wpr);
    const uint64_t kf = shn_extract(w, 0, K), kl = shn_extract(w, L - K, K);
    const int64_t sf = kp_find(hkeys, mask, kf);
    if (sf >= 0 && kp_find(hkeys, mask, kl) >= 0) {
      bool any = false, need = false;
      const uint64_t g0 = goff[sf], g1 = goff[sf + 1];
      int64_t last = -1;                                             // occurrences in index order = by ascending position
      for (uint64_t it = g0; it < g1 && !need; it++) {
        uint64_t p = ~0ULL;
        for (uint64_t j = g0; j < g1; j++) { const uint64_t q = occ[j]; if ((int64_t)q > last && q < p) p = q; }
        last = (int64_t)p;
        const uint32_t nd = kp_node_of(off, n_nodes, p);
        const uint64_t left = off[nd + 1] - p;                       // bases of the node from the occurrence on
        const uint32_t n = (uint32_t)min((uint64_t)L, left);
        bool same = true;
        for (uint32_t i = K; i < n && same; i++) {                   // (the first K bases are the K-mer itself)
          const uint32_t rb = (uint32_t)((w[i >> 5] >> (62 - 2 * (i & 31))) & 3ULL);
          same = kp_code(bases[p + i]) == (int)rb;
        }
        if (!same) continue;
        if ((uint64_t)L <= left) { fn = (int32_t)nd; any = true; }
        else {
          need = true;
          // the K-mer's only occurrence: the host starts its search right there, without an index of its own
          if (off_out && g1 - g0 == 1) { fn = (int32_t)nd; fo = (uint32_t)(p - off[nd]); }
        }
      }
      st = need ? (off_out && g1 - g0 == 1 ? 3 : 2) : any ? 1 : 0;
    }
  }
  state[r] = st;
  node_out[r] = fn;
  if (off_out) off_out[r] = fo;
}

}  // namespace

// reads: the distinct reads of the partition (ACGT only); node_bases / node_off: the texts of the partition's nodes one after
// the other, in the order the host's seed index would list them; state_out[r] / node_out[r] as described above (node = index
// into that order).  SHN_ERR_ARG if a node holds a base outside ACGT or K > 31.
extern "C" int shn_known_paths_scan(shn_ctx* ctx, const shn_reads* reads, int K, const uint8_t* node_bases, const uint64_t* node_off,
                                    uint64_t n_nodes, uint8_t* state_out, int32_t* node_out, uint32_t* offset_out) {
  if (!ctx || !reads || !node_off || (n_nodes && !node_bases) || (reads->n_reads && (!state_out || !node_out)))
    return shn_fail(SHN_ERR_ARG, "shn_known_paths_scan: NULL argument");
  if (K < 1 || K > 31) return shn_fail(SHN_ERR_ARG, "shn_known_paths_scan: K must be in [1,31]");
  if (reads->n_invalid) return shn_fail(SHN_ERR_ARG, "shn_known_paths_scan: reads contain non-ACGT bases");
  const uint64_t nr = reads->n_reads, total = n_nodes ? node_off[n_nodes] : 0;
  if (!nr) return SHN_OK;
  if (!total || n_nodes >= 0x7FFFFFFFULL || total >= 0xFFFFFFFFULL) {
    if (total) return shn_fail(SHN_ERR_ARG, "shn_known_paths_scan: too many nodes / bases");
    std::fill(state_out, state_out + nr, (uint8_t)0);
    for (uint64_t i = 0; i < nr; i++) node_out[i] = -1;
    if (offset_out) std::fill(offset_out, offset_out + nr, 0u);
    return SHN_OK;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  TimerRegion treg(ctx, T_SEEDS);
  ShnDevBufs bufs(s);
  uint64_t T = 1024;
  while (T < 2 * total) T <<= 1;
  uint8_t *d_bases = nullptr, *d_state = nullptr;
  uint64_t *d_off = nullptr, *d_goff = nullptr;
  unsigned long long *d_hkeys = nullptr, *d_cnt2 = nullptr;
  uint32_t *d_cnt = nullptr, *d_fill = nullptr, *d_slot = nullptr, *d_occ = nullptr;
  int32_t* d_node = nullptr;
  uint32_t* d_ofs = nullptr;
#define TRYK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return shn_fail(SHN_ERR_HIP, std::string("shn_known_paths_scan: ") + hipGetErrorString(e_)); } while (0)
  TRYK(bufs.get(&d_bases, total + 8));
  TRYK(bufs.get(&d_off, (n_nodes + 1) * 8));
  TRYK(bufs.get(&d_hkeys, T * 8));
  TRYK(bufs.get(&d_cnt, (T + 1) * 4)); TRYK(bufs.get(&d_fill, (T + 1) * 4));
  TRYK(bufs.get(&d_goff, (T + 2) * 8));
  TRYK(bufs.get(&d_slot, (total + 1) * 4)); TRYK(bufs.get(&d_occ, (total + 1) * 4));
  TRYK(bufs.get(&d_state, nr + 1)); TRYK(bufs.get(&d_node, (nr + 1) * 4));
  TRYK(bufs.get(&d_cnt2, 16));
  if (offset_out) TRYK(bufs.get(&d_ofs, (nr + 1) * 4));
  TRYK(hipMemcpyAsync(d_bases, node_bases, total, hipMemcpyHostToDevice, s));
  TRYK(hipMemcpyAsync(d_off, node_off, (n_nodes + 1) * 8, hipMemcpyHostToDevice, s));
  TRYK(hipMemsetAsync(d_cnt2, 0, 16, s));
  TRYK(hipMemsetAsync(d_hkeys, 0xFF, T * 8, s));
  TRYK(hipMemsetAsync(d_cnt, 0, (T + 1) * 4, s));
  TRYK(hipMemsetAsync(d_fill, 0, (T + 1) * 4, s));
  const uint32_t gp = (uint32_t)cdiv(total, 256);
  hipLaunchKernelGGL(kp_insert, dim3(gp), dim3(256), 0, s, d_bases, d_off, (uint32_t)n_nodes, total, K, d_hkeys, T - 1, d_cnt, d_slot, d_cnt2);
  { int rc = shn_device_scan_u32(ctx, d_cnt, T, d_goff, nullptr); if (rc) return rc; }
  hipLaunchKernelGGL(kp_fill, dim3(gp), dim3(256), 0, s, d_slot, total, d_goff, d_fill, d_occ);
  RView v{reads->d_words, reads->d_woff, reads->d_len, nr, reads->fixed_len, reads->wpr};
  hipLaunchKernelGGL(kp_classify, dim3((uint32_t)cdiv(nr, 256)), dim3(256), 0, s, v, K, d_hkeys, T - 1, d_goff, d_occ, d_bases, d_off, (uint32_t)n_nodes,
                     d_state, d_node, offset_out ? d_ofs : nullptr);
  TRYK(hipGetLastError());
  unsigned long long cnt[2] = {0, 0};
  TRYK(hipMemcpyAsync(cnt, d_cnt2, 16, hipMemcpyDeviceToHost, s));
  TRYK(hipMemcpyAsync(state_out, d_state, nr, hipMemcpyDeviceToHost, s));
  TRYK(hipMemcpyAsync(node_out, d_node, nr * 4, hipMemcpyDeviceToHost, s));
  if (offset_out) TRYK(hipMemcpyAsync(offset_out, d_ofs, nr * 4, hipMemcpyDeviceToHost, s));
  TRYK(hipStreamSynchronize(s));
  if (cnt[1]) return shn_fail(SHN_ERR_ARG, "shn_known_paths_scan: a node holds a base outside ACGT");
#undef TRYK
  return SHN_OK;
}
