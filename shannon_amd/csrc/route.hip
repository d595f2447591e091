// Read -> partition routing on gfx950 (row a10): replaces the per-read Python loops of
// kmers_for_component.py:322-403 (get_rmers :186-192, get_comps :194-205).
//
// The reference streams the strand-doubled read files; here the doubled index d enumerates
//   SE:  d <  N : R[d]                       d >= N : RC(R[d-N])                (shannon.py:396-403)
//   PE:  d <  N : (R1[d], RC(R1[d]))         d >= N : (RC(R2[d-N]), R2[d-N])    (shannon.py:413-424,
//        reads_1 = R1 ++ RC(R2), reads_2 = RC(R1) ++ R2 -- reproduced as written)
// and reverse complements are computed on chip from the packed forward reads.  A read (pair) with
// any non-ACGT base is dropped (:336, :376).  Probes: k1-windows at 0,k1,2k1,... while i < len-k1,
// plus the last window; the read goes to the union of the partition sets of the probes that hit.
#include "common.h"
#include <cstring>
#include <algorithm>

#define RBLK 256
#define MAXP 32   // distinct partitions one read (pair) can hit: 2 mates x probes x set size, deduplicated

struct RView {
  const uint64_t* words; const uint64_t* woff; const uint32_t* len; const uint8_t* bad;
  uint64_t n; uint32_t fixed_len, wpr;
};

__device__ __forceinline__ uint64_t probe_key(const RView& v, uint64_t r, uint32_t len, uint32_t pos, int k, bool rc) {
  // k-window at offset pos of the read (rc=false) or of its reverse complement (rc=true)
  uint64_t wb = v.woff ? v.woff[r] : r * v.wpr;
  if (!rc) return shn_extract(v.words + wb, pos, k);
  return shn_revcomp(shn_extract(v.words + wb, len - k - pos, k), k);
}

// ---- the probe table as a one-line dictionary (the scheme of the adjacency build, extend.hip): a 128-byte line holds ten keys (80
// bytes), their ten values (40 bytes) and the number of keys that hashed there; four keys per line on average, a full line sends
// its keys on to the next (PD_HOPS of them, then the table itself).  HBM serves 128 bytes per request whatever is asked for, and a
// probe through the table was three requests (bucket offsets, keys, value): 186 GB per launch for 19 GB of algorithmic bytes.
// The line of a key grows with the key's hash, i.e. with its bucket: the build goes through the table front to back and its
// atomics stay in the L2.
#define PD_SLOTS 10
#define PD_PER_LINE 4
#define PD_HOPS 4
struct ProbeDict { const unsigned long long* lines; uint64_t n_lines; };
__global__ void pd_build_kernel(const uint64_t* __restrict__ tkeys, const uint32_t* __restrict__ tvals, uint64_t n, unsigned long long* __restrict__ lines,
                                uint64_t n_lines) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = tkeys[i];
  unsigned long long* line = lines + __umul64hi(shn_mix64(key), n_lines) * 16;
  for (int hop = 0; hop < PD_HOPS; hop++, line += 16) {
    const uint32_t slot = atomicAdd((uint32_t*)line + 30, 1u);
    if (slot < PD_SLOTS) { line[slot] = key; ((uint32_t*)line)[20 + slot] = tvals[i]; break; }
  }
}
// value of `key` (table values are >= 1) or 0
__device__ __forceinline__ uint32_t pd_find(const ProbeDict& D, uint64_t key, int k, const uint64_t* __restrict__ tkeys, const uint32_t* __restrict__ tvals,
                                            const uint64_t* __restrict__ boff, int bits) {
  uint64_t line = __umul64hi(shn_mix64(key), D.n_lines);
  for (int hop = 0; hop < PD_HOPS; hop++, line++) {
    const ulonglong2* L = (const ulonglong2*)(D.lines + line * 16);
    const ulonglong2 q0 = L[0], q1 = L[1], q2 = L[2], q3 = L[3], q4 = L[4], v0 = L[5], v1 = L[6], v2 = L[7];
    const uint32_t cnt = (uint32_t)v2.y;                               // word 30
    const uint32_t nk = cnt < PD_SLOTS ? cnt : PD_SLOTS;
    int slot = -1;
    if (nk > 0 && q0.x == key) slot = 0; else if (nk > 1 && q0.y == key) slot = 1;
    else if (nk > 2 && q1.x == key) slot = 2; else if (nk > 3 && q1.y == key) slot = 3;
    else if (nk > 4 && q2.x == key) slot = 4; else if (nk > 5 && q2.y == key) slot = 5;
    else if (nk > 6 && q3.x == key) slot = 6; else if (nk > 7 && q3.y == key) slot = 7;
    else if (nk > 8 && q4.x == key) slot = 8; else if (nk > 9 && q4.y == key) slot = 9;
    if (slot >= 0) {
      const unsigned long long w = slot < 2 ? v0.x : slot < 4 ? v0.y : slot < 6 ? v1.x : slot < 8 ? v1.y : v2.x;   // words 20 + slot: two per 64-bit word
      return (slot & 1) ? (uint32_t)(w >> 32) : (uint32_t)w;
    }
    if (cnt <= PD_SLOTS) return 0u;
  }
  const int64_t j = shn_table_find_k(tkeys, boff, bits, key, 2 * k);  // (PD_HOPS full lines in a row)
  return j >= 0 ? tvals[j] : 0u;
}

// collects the partitions hit by one read into loc[] (dedup), returns new count
__device__ __forceinline__ int collect(const RView& v, uint64_t r, bool rc, int k, const uint64_t* __restrict__ tkeys,
                                       const uint32_t* __restrict__ tvals, const uint64_t* __restrict__ boff, int bits, const ProbeDict& D,
                                       const uint32_t* __restrict__ set_off, const uint32_t* __restrict__ set_mem,
                                       uint32_t* loc, int nloc, uint32_t* overflow) {
  uint32_t len = v.len ? v.len[r] : v.fixed_len;
  if (len < (uint32_t)k) {
    // get_rmers on a short read returns [read[-R:]] = the whole read, which is not a k1-mer key: no hit
    return nloc;
  }
  uint32_t i = 0;
  bool last_done = false;
  while (true) {
    uint32_t pos;
    if (i < len - k) { pos = i; i += k; }
    else if (!last_done) { pos = len - k; last_done = true; }
    else break;
    uint64_t key = probe_key(v, r, len, pos, k, rc);
    const uint32_t val = D.lines ? pd_find(D, key, k, tkeys, tvals, boff, bits) : 0u;
    int64_t j = D.lines ? -1 : shn_table_find_k(tkeys, boff, bits, key, 2 * k);
    if (val || j >= 0) {
      uint32_t sid = (val ? val : tvals[j]) - 1;
      for (uint32_t m = set_off[sid]; m < set_off[sid + 1]; m++) {
        uint32_t p = set_mem[m];
        bool seen = false;
        for (int q = 0; q < nloc; q++) seen |= (loc[q] == p);
        if (!seen) { if (nloc < MAXP) loc[nloc++] = p; else atomicExch(overflow, 1u); }
      }
    }
  }
  return nloc;
}

template <bool FILL>
__global__ __launch_bounds__(RBLK) void route_kernel(RView a, RView b, int paired, int ss, int k, const uint64_t* __restrict__ tkeys,
                                                     const uint32_t* __restrict__ tvals, const uint64_t* __restrict__ boff, int bits, ProbeDict D,
                                                     const uint32_t* __restrict__ set_off, const uint32_t* __restrict__ set_mem,
                                                     uint32_t* __restrict__ counts, const uint64_t* __restrict__ offs,
                                                     uint64_t* __restrict__ out, uint32_t* __restrict__ overflow, uint2* __restrict__ first2) {
  uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t N = a.n;
  if (d >= 2 * N) return;
  // the counting pass leaves the first two partitions of every pair behind: the filling pass probes again only for the pairs
  // that hit more (nearly every pair hits one partition or none; the probes are the cost of this kernel)
  if (FILL && first2) {
    const uint32_t n = counts[d];
    if (n <= 2) {
      if (n) { const uint2 f = first2[d]; const uint64_t o = offs[d]; out[o] = ((uint64_t)f.x << 32) | (uint64_t)(uint32_t)d; if (n > 1) out[o + 1] = ((uint64_t)f.y << 32) | (uint64_t)(uint32_t)d; }
      return;
    }
  }
  uint32_t loc[MAXP];
  int nloc = 0;
  bool second = d >= N;
  uint64_t i = second ? d - N : d;
  bool dropped;
  if (ss) {
    // -s / --ss (shannon.py:407-411): the read files are reads (SE) or reads_1 and RC(reads_2) (PE), not doubled: index d < N only
    dropped = second || (a.bad && a.bad[i]) || (paired && b.bad && b.bad[i]);
    if (!dropped) {
      nloc = collect(a, i, false, k, tkeys, tvals, boff, bits, D, set_off, set_mem, loc, nloc, overflow);
      if (paired) nloc = collect(b, i, true, k, tkeys, tvals, boff, bits, D, set_off, set_mem, loc, nloc, overflow);
    }
  } else if (!paired) {
    dropped = a.bad && a.bad[i];
    if (!dropped) nloc = collect(a, i, second, k, tkeys, tvals, boff, bits, D, set_off, set_mem, loc, nloc, overflow);
  } else {
    const RView& src = second ? b : a;        // d<N: (R1, RC(R1)); d>=N: (RC(R2), R2)
    dropped = src.bad && src.bad[i];
    if (!dropped) {
      nloc = collect(src, i, second, k, tkeys, tvals, boff, bits, D, set_off, set_mem, loc, nloc, overflow);   // mate 1
      nloc = collect(src, i, !second, k, tkeys, tvals, boff, bits, D, set_off, set_mem, loc, nloc, overflow);  // mate 2
    }
  }
  if (!FILL) { counts[d] = (uint32_t)nloc; if (first2) first2[d] = make_uint2(nloc > 0 ? loc[0] : 0u, nloc > 1 ? loc[1] : 0u); return; }
  uint64_t o = offs[d];
  for (int q = 0; q < nloc; q++) out[o + q] = ((uint64_t)loc[q] << 32) | (uint64_t)(uint32_t)d;
}

__global__ void split_u64_kernel(const uint64_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ hi, uint32_t* __restrict__ lo) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  hi[i] = (uint32_t)(in[i] >> 32);
  lo[i] = (uint32_t)in[i];
}

static RView rview(const shn_reads* r) {
  RView v;
  v.words = r->d_words; v.woff = r->d_woff; v.len = r->d_len; v.bad = r->n_invalid ? r->d_bad : nullptr;
  v.n = r->n_reads; v.fixed_len = r->fixed_len; v.wpr = r->wpr;
  return v;
}

struct shn_routes {
  shn_ctx* ctx;
  int device;
  uint64_t n;          // number of (partition, doubled read index) pairs
  uint32_t* d_pid;     // sorted by (pid, read index)
  uint32_t* d_ridx;
};

// (internal, graph_dev.h) entries [lo, lo + n) of the routed read indices where they lie on the device
int shn_routes_device_slice(const shn_routes* r, uint64_t lo, uint64_t n, const uint32_t** out) {
  if (!r || !out) return shn_fail(SHN_ERR_ARG, "shn_routes_device_slice: NULL argument");
  if (lo + n > r->n) return shn_fail(SHN_ERR_ARG, "shn_routes_device_slice: range outside the routes");
  *out = r->d_ridx + lo;
  return SHN_OK;
}

extern "C" void shn_routes_destroy(shn_routes* r) {
  if (!r) return;
  hipSetDevice(r->device);
  if (r->d_pid) shn_dev_free(r->d_pid);
  if (r->d_ridx) shn_dev_free(r->d_ridx);
  delete r;
}
extern "C" uint64_t shn_routes_size(const shn_routes* r) { return r ? r->n : 0; }

extern "C" int shn_routes_download(shn_ctx* ctx, const shn_routes* r, uint32_t* pid, uint32_t* ridx) {
  if (!ctx || !r) return shn_fail(SHN_ERR_ARG, "shn_routes_download: NULL argument");
  SHN_ENTER(ctx);
  if (pid) HIP_TRY(hipMemcpyAsync(pid, r->d_pid, r->n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (ridx) HIP_TRY(hipMemcpyAsync(ridx, r->d_ridx, r->n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SHN_OK;
}

// Routes are sorted by (partition, doubled read index): where every partition starts, and how many of its routes lie
// below `split` (= the forward half of the strand-doubled order) -- one thread per partition, two binary searches.
__global__ void routes_bounds_kernel(const uint32_t* __restrict__ pid, const uint32_t* __restrict__ ridx, uint64_t n, uint32_t n_parts,
                                     uint32_t split, uint64_t* __restrict__ start, uint64_t* __restrict__ below) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > n_parts) return;
  auto lower_pid = [&](uint32_t v) { uint64_t lo = 0, hi = n; while (lo < hi) { uint64_t m = (lo + hi) >> 1; if (pid[m] < v) lo = m + 1; else hi = m; } return lo; };
  const uint64_t a = lower_pid(p);
  start[p] = a;
  if (p == n_parts) return;
  const uint64_t b = lower_pid(p + 1);
  uint64_t lo = a, hi = b;
  while (lo < hi) { uint64_t m = (lo + hi) >> 1; if (ridx[m] < split) lo = m + 1; else hi = m; }
  below[p] = lo - a;
}

extern "C" int shn_routes_bounds(shn_ctx* ctx, const shn_routes* r, uint32_t n_parts, uint32_t split, uint64_t* start, uint64_t* below) {
  if (!ctx || !r || !start || !below) return shn_fail(SHN_ERR_ARG, "shn_routes_bounds: NULL argument");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  uint64_t* d = nullptr;
  HIP_TRY(shn_dev_malloc(&d, (size_t)(2 * n_parts + 2) * 8));
  hipLaunchKernelGGL(routes_bounds_kernel, dim3((n_parts + 1 + 63) / 64), dim3(64), 0, s, r->d_pid, r->d_ridx, r->n, n_parts, split, d, d + n_parts + 1);
  HIP_TRY(hipMemcpyAsync(start, d, (size_t)(n_parts + 1) * 8, hipMemcpyDeviceToHost, s));
  if (n_parts) HIP_TRY(hipMemcpyAsync(below, d + n_parts + 1, (size_t)n_parts * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  shn_dev_free(d);
  HIP_TRY(hipGetLastError());
  return SHN_OK;
}

extern "C" int shn_routes_download_range(shn_ctx* ctx, const shn_routes* r, uint64_t lo, uint64_t n, uint32_t* ridx) {
  if (!ctx || !r || (n && !ridx)) return shn_fail(SHN_ERR_ARG, "shn_routes_download_range: NULL argument");
  if (lo + n > r->n) return shn_fail(SHN_ERR_ARG, "shn_routes_download_range: range outside the routes");
  SHN_ENTER(ctx);
  if (n) HIP_TRY(hipMemcpyAsync(ridx, r->d_ridx + lo, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SHN_OK;
}

extern "C" int shn_route_reads(shn_ctx* ctx, const shn_reads* r1, const shn_reads* r2, int k1, const shn_table* probe,
                               const uint32_t* set_off, const uint32_t* set_members, uint32_t n_sets, shn_routes** out) {
  return shn_route_reads_mode(ctx, r1, r2, k1, probe, set_off, set_members, n_sets, 0, out);
}
extern "C" int shn_route_reads_mode(shn_ctx* ctx, const shn_reads* r1, const shn_reads* r2, int k1, const shn_table* probe,
                                    const uint32_t* set_off, const uint32_t* set_members, uint32_t n_sets, int strand_specific, shn_routes** out) {
  if (!ctx || !r1 || !probe || !set_off || !out) return shn_fail(SHN_ERR_ARG, "shn_route_reads: NULL argument");
  if (r2 && r2->n_reads != r1->n_reads) return shn_fail(SHN_ERR_ARG, "shn_route_reads: mate files differ in length");
  if (probe->canonical) return shn_fail(SHN_ERR_ARG, "shn_route_reads: probe table must hold plain (non-canonical) k1-mers");
  if (2 * r1->n_reads >= 0xFFFFFFFFULL) return shn_fail(SHN_ERR_ARG, "shn_route_reads: too many reads for 32-bit doubled indices");
  SHN_ENTER(ctx);
  shn_stage_begin(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  TimerRegion treg(ctx, T_ROUTE);
  uint64_t N2 = 2 * r1->n_reads;
  uint32_t n_mem = set_off[n_sets];
  void *pso, *pcnt, *poff, *pflag;
  int rc;
  if ((rc = shn_ws(ctx)[14].get((size_t)(n_sets + 1 + n_mem + 1) * 4, &pso)) || (rc = shn_ws(ctx)[15].get((N2 + 1) * 4, &pcnt)) ||
      (rc = shn_ws(ctx)[16].get((N2 + 2) * 8, &poff)) || (rc = shn_ws(ctx)[17].get(64, &pflag))) return rc;
  uint32_t* d_so = (uint32_t*)pso;
  uint32_t* d_sm = d_so + n_sets + 1;
  HIP_TRY(hipMemcpyAsync(d_so, set_off, (size_t)(n_sets + 1) * 4, hipMemcpyHostToDevice, s));
  if (n_mem) HIP_TRY(hipMemcpyAsync(d_sm, set_members, (size_t)n_mem * 4, hipMemcpyHostToDevice, s));
  uint32_t* d_ovf = (uint32_t*)pflag;
  HIP_TRY(hipMemsetAsync(d_ovf, 0, 4, s));
  RView a = rview(r1), b = r2 ? rview(r2) : rview(r1);
  shn_routes* R = new shn_routes();
  memset(R, 0, sizeof(*R));
  R->ctx = ctx;
  R->device = ctx->device;
  if (N2 == 0) { *out = R; return SHN_OK; }
  uint32_t grid = (uint32_t)cdiv(N2, RBLK);
  // the one-line dictionary over the probe table (SHN_ROUTE_DICT=0: probes through the table)
  ProbeDict D{nullptr, 0};
  if (probe->n >= 4096 && !(getenv("SHN_ROUTE_DICT") && getenv("SHN_ROUTE_DICT")[0] == '0')) {
    const uint64_t n_lines = probe->n / PD_PER_LINE + 1;
    void* pl;
    if (shn_ws(ctx)[31].get((n_lines + PD_HOPS + 1) * 128, &pl) == 0) {
      HIP_TRY(hipMemsetAsync(pl, 0, (n_lines + PD_HOPS + 1) * 128, s));
      hipLaunchKernelGGL(pd_build_kernel, dim3((uint32_t)cdiv(probe->n, 256)), dim3(256), 0, s, (const uint64_t*)probe->d_keys, (const uint32_t*)probe->d_counts,
                         probe->n, (unsigned long long*)pl, n_lines);
      D.lines = (const unsigned long long*)pl; D.n_lines = n_lines;
    } else (void)hipGetLastError();
  }
  void* pf2 = nullptr;
  uint2* d_first2 = shn_ws(ctx)[30].get((N2 + 1) * 8, &pf2) == 0 ? (uint2*)pf2 : nullptr;          // (without it the second pass probes again)
  hipLaunchKernelGGL(route_kernel<false>, dim3(grid), dim3(RBLK), 0, s, a, b, r2 ? 1 : 0, strand_specific ? 1 : 0, k1, probe->d_keys, probe->d_counts,
                     probe->d_bucket_off, probe->bits, D, d_so, d_sm, (uint32_t*)pcnt, nullptr, nullptr, d_ovf, d_first2);
  uint64_t total = 0;
  if ((rc = shn_device_scan_u32(ctx, (uint32_t*)pcnt, N2, (uint64_t*)poff, &total))) { delete R; return rc; }
  uint32_t ovf = 0;
  HIP_TRY(hipMemcpyAsync(&ovf, d_ovf, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (ovf) { delete R; return shn_fail(SHN_ERR_OVERFLOW, "shn_route_reads: a read hit more than 32 partitions"); }
  if (total >= 0xFFFFFFFFULL) { delete R; return shn_fail(SHN_ERR_OVERFLOW, "shn_route_reads: more than 2^32 routed pairs"); }
  R->n = total;
  void *pk, *pk2, *pv, *pv2;
  if ((rc = shn_ws(ctx)[9].get((total + 2) * 8, &pk)) || (rc = shn_ws(ctx)[11].get((total + 2) * 8, &pk2)) ||
      (rc = shn_ws(ctx)[10].get((total + 2) * 4, &pv)) || (rc = shn_ws(ctx)[12].get((total + 2) * 4, &pv2))) { delete R; return rc; }
  if (total) {
    hipLaunchKernelGGL(route_kernel<true>, dim3(grid), dim3(RBLK), 0, s, a, b, r2 ? 1 : 0, strand_specific ? 1 : 0, k1, probe->d_keys, probe->d_counts,
                       probe->d_bucket_off, probe->bits, D, d_so, d_sm, (uint32_t*)pcnt, (const uint64_t*)poff, (uint64_t*)pk, d_ovf, d_first2);
    // pairs were written in doubled-read order; a stable sort on the partition id keeps that order
    HIP_TRY(hipMemsetAsync(pv, 0, total * 4, s));
    int pbits = 1;
    while (pbits < 32 && (1ULL << pbits) < (uint64_t)n_mem + 2) pbits++;
    if ((rc = shn_sort_pairs(ctx, (uint64_t*)pk, (uint32_t*)pv, (uint64_t*)pk2, (uint32_t*)pv2, total, 32, 32 + ((pbits + 7) / 8) * 8))) { delete R; return rc; }
    HIP_TRY(shn_dev_malloc(&R->d_pid, total * 4));
    HIP_TRY(shn_dev_malloc(&R->d_ridx, total * 4));
    hipLaunchKernelGGL(split_u64_kernel, dim3((uint32_t)cdiv(total, 256)), dim3(256), 0, s, (const uint64_t*)pk, total, R->d_pid, R->d_ridx);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  *out = R;
  return SHN_OK;
}

// Build a lookup table from host (key, value) pairs with unique keys (value stored in `counts`).
extern "C" int shn_table_create(shn_ctx* ctx, const uint64_t* keys, const uint32_t* values, uint64_t n, int k, int canonical,
                                shn_table** out) {
  if (!ctx || !out || (n && (!keys || !values))) return shn_fail(SHN_ERR_ARG, "shn_table_create: NULL argument");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  if (n <= (1u << 22) && !getenv("SHN_TABLE_DEVICE_BUILD")) {
    // Small dictionaries (the graph stage's K-mer seed tables: a few 10^5 keys, one or two per partition, made by many host threads
    // at once) are laid out on the host -- bucket = top bits of fmix64(key), ascending inside a bucket, duplicates summed -- and
    // uploaded: three copies on the caller's stream instead of the dozen launches and syncs of the device pipeline, and no use of
    // the process-wide workspaces, so that calls of different threads (each on its own context) overlap.
    int bits = 0;
    while (bits < 23 && (n >> bits) > 96) bits++;
    const uint64_t nbk = 1ULL << bits;
    std::vector<uint32_t> bucket(n);
    std::vector<uint64_t> boff(nbk + 1, 0);
    for (uint64_t i = 0; i < n; i++) { bucket[i] = bits ? (uint32_t)(shn_mix64(keys[i]) >> (64 - bits)) : 0u; boff[bucket[i] + 1]++; }
    for (uint64_t b = 0; b < nbk; b++) boff[b + 1] += boff[b];
    std::vector<uint64_t> cur(boff.begin(), boff.end() - 1), sk(n);
    std::vector<uint32_t> sv(n);
    for (uint64_t i = 0; i < n; i++) { const uint64_t at = cur[bucket[i]]++; sk[at] = keys[i]; sv[at] = values[i]; }
    std::vector<uint32_t> idx;
    uint64_t total = 0, w = 0;
    std::vector<uint64_t> nboff(nbk + 1, 0);
    for (uint64_t b = 0; b < nbk; b++) {
      const uint64_t lo = boff[b], hi = boff[b + 1];
      idx.resize(hi - lo);
      for (uint64_t i = 0; i < hi - lo; i++) idx[i] = (uint32_t)i;
      std::sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return sk[lo + x] < sk[lo + y]; });
      nboff[b] = w;
      // (in place: w never passes the read position of the bucket being written)
      std::vector<uint64_t> tk(hi - lo); std::vector<uint32_t> tv(hi - lo);
      for (uint64_t i = 0; i < hi - lo; i++) { tk[i] = sk[lo + idx[i]]; tv[i] = sv[lo + idx[i]]; }
      for (uint64_t i = 0; i < hi - lo; i++) {
        total += tv[i];
        if (w > nboff[b] && sk[w - 1] == tk[i]) sv[w - 1] += tv[i];
        else { sk[w] = tk[i]; sv[w] = tv[i]; w++; }
      }
    }
    nboff[nbk] = w;
    shn_table* t = new shn_table();
    memset(t, 0, sizeof(*t));
    t->ctx = ctx; t->device = ctx->device; t->k = k; t->canonical = canonical; t->bits = bits; t->n_buckets = nbk; t->total = total; t->n = w;
    hipError_t e = shn_dev_malloc(&t->d_keys, (w + 1) * 8);
    if (e == hipSuccess) e = shn_dev_malloc(&t->d_counts, (w + 1) * 4);
    if (e == hipSuccess) e = shn_dev_malloc(&t->d_bucket_off, (nbk + 1) * 8);
    if (e == hipSuccess && w) e = hipMemcpyAsync(t->d_keys, sk.data(), w * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && w) e = hipMemcpyAsync(t->d_counts, sv.data(), w * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(t->d_bucket_off, nboff.data(), (nbk + 1) * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { shn_table_destroy(t); return shn_fail(SHN_ERR_HIP, std::string("shn_table_create: ") + hipGetErrorString(e)); }
    *out = t;
    return SHN_OK;
  }
  uint64_t* dk = nullptr; uint32_t* dv = nullptr;
  HIP_TRY(shn_dev_malloc(&dk, (n + 1) * 8));
  HIP_TRY(shn_dev_malloc(&dv, (n + 1) * 4));
  if (n) {
    HIP_TRY(hipMemcpyAsync(dk, keys, n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dv, values, n * 4, hipMemcpyHostToDevice, s));
  }
  int rc = shn_table_from_pairs(ctx, dk, dv, n, k, canonical, out);
  hipStreamSynchronize(s);
  shn_dev_free(dk); shn_dev_free(dv);
  return rc;
}
