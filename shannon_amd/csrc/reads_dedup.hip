// The distinct reads of a partition, found on the device.
//
// multibridging.py:185-236 (load_reads / load_mated_reads) reads one read (pair) after the other into Read.reads, a dictionary
// string -> Read: a string seen before only has its copy count raised (mbgraph.py:56-70), and for pairs every occurrence sets the
// two reads' mate / mate_pair fields again, so each distinct read ends with the role and mate of its LAST occurrence.  The native
// graph stage keeps the reads as ids in order of first occurrence.  At BASELINE configs[2] the 111 partitions hold 240 M read
// slots (the largest 8 M, 3.6 M of them distinct); decoding, hashing and de-duplicating them on host threads was the longest
// part of the stage.  Here the slots are rows of the packed read sets already resident on the device (slot j = read j / nm of
// the partition's routed list, mate j % nm; shn_route_reads' doubled numbering names the row and the strand), so the same
// answer is an open-addressing insert keyed by the packed words, two atomics per slot and one prefix sum:
//   dd_insert   every slot looks its 2-bit words up (CAS into an empty slot, else compare words with the slot's
//               representative; equal -> atomicMin of the slot index, so the representative ends as the FIRST occurrence)
//   dd_tally    count and last occurrence per first occurrence; flag = "is a first occurrence"
//   (scan)      id = number of first occurrences before the slot = the sequential numbering
//   dd_emit     per distinct read: its first slot, count, and (pairs) role + mate id of its last occurrence
// Output order and values equal the host code they replace (mbgraph_host.hip, bulk numbering) and the one-at-a-time interner
// (tests/test_host_graph.py, tests/test_e2e_gpu.py).
#include "common.h"
#include "graph_dev.h"

#include <algorithm>
#include <chrono>
#include <string>

namespace {

struct SlotSrc {
  const uint64_t* wa; const uint64_t* wb; const uint32_t* didx; uint64_t n_in; uint32_t wpr, L; int paired;
};

// words of slot j (see shn_mbgraph_run_resident for the numbering): SE d < N -> R[d], d >= N -> RC(R[d-N]);
// PE d < N -> (R1[d], RC(R1[d])), d >= N -> (RC(R2[d-N]), R2[d-N])
__device__ __forceinline__ void slot_origin(const SlotSrc& S, uint64_t j, const uint64_t*& src, bool& rc) {
  const uint64_t i = S.paired ? (j >> 1) : j;
  const int mate = S.paired ? (int)(j & 1) : 0;
  const uint64_t d = S.didx[i];
  const bool second = d >= S.n_in;
  const uint64_t row = second ? d - S.n_in : d;
  bool setb;
  if (!S.paired) { setb = false; rc = second; }
  else if (mate == 0) { setb = second; rc = second; }
  else { setb = second; rc = !second; }
  src = (setb ? S.wb : S.wa) + row * S.wpr;
}
__device__ __forceinline__ uint64_t slot_word(const SlotSrc& S, const uint64_t* src, bool rc, uint32_t w) {
  if (!rc) return src[w];
  // bases [32 w, 32 w + nb) of the reverse complement = the complement of bases [L - 32 w - nb, L - 32 w) read backwards
  if (32 * w >= S.L) return 0;
  const uint32_t nb = min(32u, S.L - 32 * w);
  return shn_revcomp(shn_extract(src, S.L - 32 * w - nb, (int)nb), (int)nb) << (64 - 2 * nb);
}
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr int MAXW = 16;                                  // reads of up to 512 bases

// MW: compile-time bound of the words per read (the words of a slot stay in registers: a run-time bound put them in scratch)
template <int MW>
__global__ void dd_insert(SlotSrc S, uint64_t nh, uint32_t* __restrict__ tab, uint64_t mask, uint32_t* __restrict__ slot_of) {
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nh; j += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t* src; bool rc;
    slot_origin(S, j, src, rc);
    uint64_t w[MW];
    uint64_t h = 0x9E3779B97F4A7C15ULL;
#pragma unroll
    for (uint32_t t = 0; t < (uint32_t)MW; t++) {
      w[t] = 0;
      if (t < S.wpr) { w[t] = slot_word(S, src, rc, t); h = mix64(h ^ w[t]) + 0x9E3779B97F4A7C15ULL * (t + 1); }
    }
    uint64_t s = h & mask;
    while (true) {
      uint32_t cur = tab[s];
      if (cur == EMPTY) {
        cur = atomicCAS(&tab[s], EMPTY, (uint32_t)j);
        if (cur == EMPTY) { slot_of[j] = (uint32_t)s; break; }
      }
      const uint64_t* osrc; bool orc;
      slot_origin(S, cur, osrc, orc);
      bool same = true;
#pragma unroll
      for (uint32_t t = 0; t < (uint32_t)MW; t++) if (t < S.wpr && same) same = slot_word(S, osrc, orc, t) == w[t];
      if (same) { atomicMin(&tab[s], (uint32_t)j); slot_of[j] = (uint32_t)s; break; }
      s = (s + 1) & mask;
    }
  }
}

__global__ void dd_tally(uint64_t nh, const uint32_t* __restrict__ tab, const uint32_t* __restrict__ slot_of, uint32_t* __restrict__ first,
                         uint32_t* __restrict__ cnt, uint32_t* __restrict__ last, uint32_t* __restrict__ flag) {
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nh; j += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t f = tab[slot_of[j]];
    first[j] = f;
    atomicAdd(&cnt[f], 1u);
    atomicMax(&last[f], (uint32_t)j);
    flag[j] = f == (uint32_t)j;
  }
}

__global__ void dd_emit(uint64_t nh, int paired, const uint32_t* __restrict__ first, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ last,
                        const uint64_t* __restrict__ pos, uint32_t* __restrict__ o_slot, uint32_t* __restrict__ o_cnt, int32_t* __restrict__ o_mate,
                        uint8_t* __restrict__ o_role, SlotSrc S, uint32_t* __restrict__ o_row, uint8_t* __restrict__ o_flag) {
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nh; j += (uint64_t)gridDim.x * blockDim.x) {
    if (first[j] != (uint32_t)j) continue;
    const uint64_t id = pos[j];
    if (o_slot) o_slot[id] = (uint32_t)j;
    o_cnt[id] = cnt[j];
    if (paired) {
      const uint32_t l = last[j];
      o_role[id] = (l & 1) ? 2 : 1;
      o_mate[id] = (int32_t)pos[first[l ^ 1u]];
    } else { o_role[id] = 0; o_mate[id] = -1; }
    if (o_row) {
      // the slot's place in the resident input (mbgraph_run_impl's origin_of): row, bit 0 = set b, bit 1 = reverse complement
      const uint64_t i = S.paired ? (j >> 1) : j;
      const uint64_t d = S.didx[i];
      const bool second = d >= S.n_in;
      o_row[id] = (uint32_t)(second ? d - S.n_in : d);
      uint8_t fl;
      if (!S.paired) fl = second ? 2 : 0;
      else if ((j & 1) == 0) fl = second ? 3 : 0;
      else fl = second ? 1 : 2;
      o_flag[id] = fl;
    }
  }
}

}  // namespace

// a / b: the resident fixed-length read sets (b NULL for single-end); didx[n]: doubled read indices of the partition's reads
// (host); outputs (host, room for n * nm entries each): slot_out[id] = first slot of distinct read id, count_out, mate_out
// (-1 single-end), role_out (0 / 1 / 2); *n_distinct = number of distinct reads.
extern "C" int shn_reads_dedup(shn_ctx* ctx, const shn_reads* a, const shn_reads* b, const uint32_t* didx, uint64_t n, int paired,
                               uint64_t* n_distinct, uint32_t* slot_out, uint32_t* count_out, int32_t* mate_out, uint8_t* role_out) {
  if (!ctx || !a || !n_distinct || (n && (!didx || !slot_out || !count_out || !mate_out || !role_out)) || (paired && !b))
    return shn_fail(SHN_ERR_ARG, "shn_reads_dedup: NULL argument");
  if (!a->fixed_len || a->wpr > (uint32_t)MAXW || (b && (b->fixed_len != a->fixed_len || b->wpr != a->wpr)))
    return shn_fail(SHN_ERR_ARG, "shn_reads_dedup: fixed-length read sets of one length (<= 512 bases) only");
  const uint64_t nm = paired ? 2 : 1, nh = n * nm;
  *n_distinct = 0;
  if (!nh) return SHN_OK;
  if (nh >= (1ULL << 31)) return shn_fail(SHN_ERR_ARG, "shn_reads_dedup: more than 2^31 read slots");
  const uint64_t n_in = a->n_reads;
  for (uint64_t i = 0; i < n; i++) {
    const uint64_t d = didx[i];
    const uint64_t lim = (d >= n_in) ? (paired ? b->n_reads : a->n_reads) + n_in : n_in;
    if (d >= lim) return shn_fail(SHN_ERR_ARG, "shn_reads_dedup: read index out of range");
  }
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  const bool dbg = getenv("SHN_DEBUG") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t0 = now(), t_alloc = 0, t_up = 0, t_k = 0, t_scan = 0, t_alloc2 = 0;
  uint64_t T = 1024;
  while (T < 2 * nh) T <<= 1;
  ShnDevBufs bufs(s);
  uint32_t *d_idx = nullptr, *d_tab = nullptr, *d_slot = nullptr, *d_first = nullptr, *d_cnt = nullptr, *d_last = nullptr, *d_flag = nullptr;
  uint64_t* d_pos = nullptr;
  uint32_t *d_oslot = nullptr, *d_ocnt = nullptr; int32_t* d_omate = nullptr; uint8_t* d_orole = nullptr;
#define TRYD(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return shn_fail(SHN_ERR_HIP, std::string("shn_reads_dedup: ") + hipGetErrorString(e_)); } while (0)
  TRYD(bufs.get(&d_idx, n * 4));
  TRYD(bufs.get(&d_tab, T * 4));
  TRYD(bufs.get(&d_slot, nh * 4));
  TRYD(bufs.get(&d_first, nh * 4));
  TRYD(bufs.get(&d_cnt, nh * 4));
  TRYD(bufs.get(&d_last, nh * 4));
  TRYD(bufs.get(&d_flag, nh * 4));
  TRYD(bufs.get(&d_pos, (nh + 1) * 8));
  t_alloc = now() - t0;
  TRYD(hipMemcpyAsync(d_idx, didx, n * 4, hipMemcpyHostToDevice, s));
  t_up = now() - t0;
  TRYD(hipMemsetAsync(d_tab, 0xFF, T * 4, s));
  TRYD(hipMemsetAsync(d_cnt, 0, nh * 4, s));
  TRYD(hipMemsetAsync(d_last, 0, nh * 4, s));
  SlotSrc S{a->d_words, b ? b->d_words : a->d_words, d_idx, n_in, a->wpr, a->fixed_len, paired ? 1 : 0};
  const uint32_t grid = (uint32_t)std::min<uint64_t>(cdiv(nh, 256), 1u << 20);
  { TimerRegion tdi(ctx, T_DD_INSERT); tdi.bytes(nh * ((uint64_t)a->wpr * 8 * 2 + 4 + 4 + 4));      // per read slot: its index, its words and the words of the slot it meets, the table word, its slot written
    if (a->wpr <= 4) hipLaunchKernelGGL(dd_insert<4>, dim3(grid), dim3(256), 0, s, S, nh, d_tab, T - 1, d_slot);
    else hipLaunchKernelGGL(dd_insert<MAXW>, dim3(grid), dim3(256), 0, s, S, nh, d_tab, T - 1, d_slot); }
  hipLaunchKernelGGL(dd_tally, dim3(grid), dim3(256), 0, s, nh, d_tab, d_slot, d_first, d_cnt, d_last, d_flag);
  TRYD(hipGetLastError());
  uint64_t nd = 0;
  if (dbg) { hipStreamSynchronize(s); t_k = now() - t0; }
  int rc = shn_device_scan_u32(ctx, d_flag, nh, d_pos, &nd);
  if (rc) return rc;
  t_scan = now() - t0;
  TRYD(bufs.get(&d_oslot, nd * 4));
  TRYD(bufs.get(&d_ocnt, nd * 4));
  TRYD(bufs.get(&d_omate, nd * 4));
  TRYD(bufs.get(&d_orole, nd));
  t_alloc2 = now() - t0;
  hipLaunchKernelGGL(dd_emit, dim3(grid), dim3(256), 0, s, nh, paired ? 1 : 0, d_first, d_cnt, d_last, d_pos, d_oslot, d_ocnt, d_omate, d_orole, S, (uint32_t*)nullptr, (uint8_t*)nullptr);
  TRYD(hipGetLastError());
  TRYD(hipMemcpyAsync(slot_out, d_oslot, nd * 4, hipMemcpyDeviceToHost, s));
  TRYD(hipMemcpyAsync(count_out, d_ocnt, nd * 4, hipMemcpyDeviceToHost, s));
  TRYD(hipMemcpyAsync(mate_out, d_omate, nd * 4, hipMemcpyDeviceToHost, s));
  TRYD(hipMemcpyAsync(role_out, d_orole, nd, hipMemcpyDeviceToHost, s));
  TRYD(hipStreamSynchronize(s));
#undef TRYD
  if (dbg) fprintf(stderr, "[dedup] slots=%llu distinct=%llu: alloc %.3f upload %.3f kernels %.3f scan %.3f alloc2 %.3f done %.3f s\n", (unsigned long long)nh,
                   (unsigned long long)nd, t_alloc, t_up, t_k, t_scan, t_alloc2, now() - t0);
  *n_distinct = nd;
  return SHN_OK;
}

// ---- the same search with its result left on the device (graph_dev.h)
void shn_dedup_destroy(shn_dedup* d) {
  if (!d) return;
  if (d->ctx) hipSetDevice(d->ctx->device);
  shn_dev_free(d->d_cnt); shn_dev_free(d->d_mate); shn_dev_free(d->d_role); shn_dev_free(d->d_row); shn_dev_free(d->d_flag);
  delete d;
}
int shn_reads_dedup_dev(shn_ctx* ctx, const shn_reads* a, const shn_reads* b, const uint32_t* didx, const uint32_t* d_didx, uint64_t n, int paired,
                        shn_dedup** out) {
  if (!ctx || !a || !out || (n && !didx && !d_didx) || (paired && !b)) return shn_fail(SHN_ERR_ARG, "shn_reads_dedup_dev: NULL argument");
  if (!a->fixed_len || a->wpr > (uint32_t)MAXW || (b && (b->fixed_len != a->fixed_len || b->wpr != a->wpr)))
    return shn_fail(SHN_ERR_ARG, "shn_reads_dedup_dev: fixed-length read sets of one length (<= 512 bases) only");
  const uint64_t nm = paired ? 2 : 1, nh = n * nm;
  if (nh >= (1ULL << 31)) return shn_fail(SHN_ERR_ARG, "shn_reads_dedup_dev: more than 2^31 read slots");
  const uint64_t n_in = a->n_reads;
  if (!d_didx)                                                   // (indices from the host are checked; a slice of shn_routes is the library's own)
    for (uint64_t i = 0; i < n; i++) {
      const uint64_t d = didx[i];
      const uint64_t lim = (d >= n_in) ? (paired ? b->n_reads : a->n_reads) + n_in : n_in;
      if (d >= lim) return shn_fail(SHN_ERR_ARG, "shn_reads_dedup_dev: read index out of range");
    }
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  shn_dedup* D = new shn_dedup();
  D->ctx = ctx; D->n_slots = nh; D->paired = paired ? 1 : 0;
  if (!nh) { *out = D; return SHN_OK; }
  uint64_t T = 1024;
  while (T < 2 * nh) T <<= 1;
  ShnDevBufs bufs(s);
  uint32_t *d_idx = nullptr, *d_tab = nullptr, *d_slot = nullptr, *d_first = nullptr, *d_cnt = nullptr, *d_last = nullptr, *d_flag = nullptr;
  uint64_t* d_pos = nullptr;
#define TRYD(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { shn_dedup_destroy(D); return shn_fail(SHN_ERR_HIP, std::string("shn_reads_dedup_dev: ") + hipGetErrorString(e_)); } } while (0)
  if (!d_didx) { TRYD(bufs.get(&d_idx, n * 4)); TRYD(hipMemcpyAsync(d_idx, didx, n * 4, hipMemcpyHostToDevice, s)); }
  TRYD(bufs.get(&d_tab, T * 4));
  TRYD(bufs.get(&d_slot, nh * 4));
  TRYD(bufs.get(&d_first, nh * 4));
  TRYD(bufs.get(&d_cnt, nh * 4));
  TRYD(bufs.get(&d_last, nh * 4));
  TRYD(bufs.get(&d_flag, nh * 4));
  TRYD(bufs.get(&d_pos, (nh + 1) * 8));
  TRYD(hipMemsetAsync(d_tab, 0xFF, T * 4, s));
  TRYD(hipMemsetAsync(d_cnt, 0, nh * 4, s));
  TRYD(hipMemsetAsync(d_last, 0, nh * 4, s));
  SlotSrc S{a->d_words, b ? b->d_words : a->d_words, d_didx ? d_didx : d_idx, n_in, a->wpr, a->fixed_len, paired ? 1 : 0};
  const uint32_t grid = (uint32_t)std::min<uint64_t>(cdiv(nh, 256), 1u << 20);
  { TimerRegion tdi(ctx, T_DD_INSERT); tdi.bytes(nh * ((uint64_t)a->wpr * 8 * 2 + 4 + 4 + 4));      // per read slot: its index, its words and the words of the slot it meets, the table word, its slot written
    if (a->wpr <= 4) hipLaunchKernelGGL(dd_insert<4>, dim3(grid), dim3(256), 0, s, S, nh, d_tab, T - 1, d_slot);
    else hipLaunchKernelGGL(dd_insert<MAXW>, dim3(grid), dim3(256), 0, s, S, nh, d_tab, T - 1, d_slot); }
  hipLaunchKernelGGL(dd_tally, dim3(grid), dim3(256), 0, s, nh, d_tab, d_slot, d_first, d_cnt, d_last, d_flag);
  TRYD(hipGetLastError());
  uint64_t nd = 0;
  { int rc = shn_device_scan_u32(ctx, d_flag, nh, d_pos, &nd); if (rc) { shn_dedup_destroy(D); return rc; } }
  D->n_distinct = nd;
  TRYD(shn_dev_malloc(&D->d_cnt, (nd + 1) * 4)); TRYD(shn_dev_malloc(&D->d_mate, (nd + 1) * 4)); TRYD(shn_dev_malloc(&D->d_role, nd + 1));
  TRYD(shn_dev_malloc(&D->d_row, (nd + 1) * 4)); TRYD(shn_dev_malloc(&D->d_flag, nd + 1));
  hipLaunchKernelGGL(dd_emit, dim3(grid), dim3(256), 0, s, nh, paired ? 1 : 0, d_first, d_cnt, d_last, d_pos, (uint32_t*)nullptr, D->d_cnt, D->d_mate, D->d_role, S,
                     D->d_row, D->d_flag);
  TRYD(hipGetLastError());
  TRYD(hipStreamSynchronize(s));
#undef TRYD
  *out = D;
  return SHN_OK;
}
int shn_dedup_origin(const shn_dedup* d, uint32_t* row_out, uint8_t* flag_out) {
  if (!d || (d->n_distinct && (!row_out || !flag_out))) return shn_fail(SHN_ERR_ARG, "shn_dedup_origin: NULL argument");
  if (!d->n_distinct) return SHN_OK;
  HIP_TRY(hipSetDevice(d->ctx->device));
  HIP_TRY(hipMemcpyAsync(row_out, d->d_row, d->n_distinct * 4, hipMemcpyDeviceToHost, d->ctx->stream));
  HIP_TRY(hipMemcpyAsync(flag_out, d->d_flag, d->n_distinct, hipMemcpyDeviceToHost, d->ctx->stream));
  HIP_TRY(hipStreamSynchronize(d->ctx->stream));
  return SHN_OK;
}
int shn_dedup_attrs(const shn_dedup* d, uint32_t* cnt_out, int32_t* mate_out, uint8_t* role_out) {
  if (!d || (d->n_distinct && (!cnt_out || !mate_out || !role_out))) return shn_fail(SHN_ERR_ARG, "shn_dedup_attrs: NULL argument");
  if (!d->n_distinct) return SHN_OK;
  HIP_TRY(hipSetDevice(d->ctx->device));
  HIP_TRY(hipMemcpyAsync(cnt_out, d->d_cnt, d->n_distinct * 4, hipMemcpyDeviceToHost, d->ctx->stream));
  HIP_TRY(hipMemcpyAsync(mate_out, d->d_mate, d->n_distinct * 4, hipMemcpyDeviceToHost, d->ctx->stream));
  HIP_TRY(hipMemcpyAsync(role_out, d->d_role, d->n_distinct, hipMemcpyDeviceToHost, d->ctx->stream));
  HIP_TRY(hipStreamSynchronize(d->ctx->stream));
  return SHN_OK;
}
