// The contig stage of extension_correction.run_correction (rows a5-a6) for inputs with hundreds of thousands of
// candidate contigs: duplicate_check (extension_correction.py:247-270) and contig_connections (:372-397), same results
// as the sequential stage of contig_host.hip (shn_contig_graph), with the bulk of the work on the GPU.
//
// duplicate_check of a candidate only ever looks at accepted contigs that share an r-mer with it.  So: the r-mers of ALL
// candidates are sorted on the GPU and candidates sharing an r-mer are united (lock-free union-find) -- a candidate's
// decision depends on the candidates of its own cluster only.  A cluster of one is accepted outright; the other
// clusters are decided independently of each other, each sequentially in seed order with the reference's rule (a
// small index per cluster), on host threads.  At 20,000 genes nearly every cluster is one contig or a handful.
// contig_connections: the K-mers of the accepted contigs are sorted on the GPU; only K-mers occurring in two different
// contigs matter, those runs are compacted and sent to the host, which replays the reference's loop over them
// (neighbour lists in dict insertion order, weights counted per position pair).
#include "contig_graph.h"
#include <atomic>
#include <mutex>

namespace {

#define CG_BLK 256

// contig id of every base: cid[g] for off[c] <= g < off[c+1]
__global__ void cg_cid_kernel(const uint64_t* __restrict__ off, uint64_t n_cand, uint32_t* __restrict__ cid) {
  // one wavefront per contig: contigs are 75 .. a few thousand bases
  const uint64_t c = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (c >= n_cand) return;
  const uint64_t lo = off[c], hi = off[c + 1];
  for (uint64_t g = lo + (threadIdx.x & 63); g < hi; g += 64) cid[g] = (uint32_t)c;
}

// flag[g] = 1 iff a k-window starts at base g inside its contig, the contig is selected (use == NULL: all) and the
// window holds ACGT only
__global__ void cg_flag_kernel(const uint8_t* __restrict__ bases, const uint64_t* __restrict__ off, const uint32_t* __restrict__ cid,
                               const int32_t* __restrict__ use, uint64_t total, int k, uint32_t* __restrict__ flag) {
  for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t c = cid[g];
    flag[g] = (g + (uint64_t)k <= off[c + 1] && (!use || use[c] != 0)) ? 1u : 0u;
  }
}

__device__ __forceinline__ uint32_t cg_code(uint8_t b) { return b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : 4u; }

__global__ void cg_keys_kernel(const uint8_t* __restrict__ bases, const uint32_t* __restrict__ flag, const uint64_t* __restrict__ pos,
                               uint64_t total, int k, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                               unsigned long long* __restrict__ n_bad) {
  for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
    if (!flag[g]) continue;
    uint64_t key = 0;
    bool bad = false;
    for (int j = 0; j < k; j++) { const uint32_t c = cg_code(bases[g + j]); bad |= c > 3; key = (key << 2) | (uint64_t)(c & 3); }
    if (bad) atomicAdd(n_bad, 1ULL);
    keys[pos[g]] = key;
    vals[pos[g]] = (uint32_t)g;
  }
}

__device__ __forceinline__ uint32_t cg_find(uint32_t* lab, uint32_t x) {
  uint32_t cur = x;
  while (true) {
    uint32_t p = __hip_atomic_load(&lab[cur], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == cur) return cur;
    uint32_t gp = __hip_atomic_load(&lab[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gp != p) __hip_atomic_store(&lab[cur], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    cur = p;
  }
}
__global__ void cg_iota_kernel(uint32_t* __restrict__ lab, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) lab[i] = (uint32_t)i;
}
// neighbours in the sorted order with equal keys and different contigs: unite the contigs (roots link to smaller ids)
__global__ void cg_union_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint64_t n,
                                const uint32_t* __restrict__ cid, uint32_t* lab) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    if (keys[i] != keys[i - 1]) continue;
    const uint32_t u = cid[vals[i]], v = cid[vals[i - 1]];
    if (u == v) continue;
    while (true) {
      const uint32_t ru = cg_find(lab, u), rv = cg_find(lab, v);
      if (ru == rv) break;
      const uint32_t hi = ru > rv ? ru : rv, lo = ru > rv ? rv : ru;
      if (atomicCAS(&lab[hi], hi, lo) == hi) break;
    }
  }
}
__global__ void cg_flatten_kernel(uint32_t* lab, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) lab[i] = cg_find(lab, (uint32_t)i);
}

// runs of equal keys that span two different contigs: flag every entry of such a run (the sort is stable and the
// windows were generated contig by contig, so a run's first and last entries differ in contig iff the run does)
__global__ void cg_shared_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint64_t n,
                                 const uint32_t* __restrict__ cid, uint32_t* __restrict__ flag) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    if (i && keys[i] == keys[i - 1]) continue;                 // not a run head
    uint64_t e = i + 1;
    while (e < n && keys[e] == keys[i]) e++;
    const uint32_t f = (e - i > 1 && cid[vals[i]] != cid[vals[e - 1]]) ? 1u : 0u;
    for (uint64_t j = i; j < e; j++) flag[j] = f;
  }
}
__global__ void cg_compact_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ flag,
                                  const uint64_t* __restrict__ pos, uint64_t n, uint64_t* __restrict__ okeys, uint32_t* __restrict__ ovals) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    if (flag[i]) { okeys[pos[i]] = keys[i]; ovals[pos[i]] = vals[i]; }
}

struct DevBufs {
  std::vector<void*> p;
  template <class T> hipError_t get(T** out, size_t bytes) { hipError_t e = shn_dev_malloc(out, bytes); if (e == hipSuccess) p.push_back((void*)*out); return e; }
  ~DevBufs() { for (void* q : p) shn_dev_free(q); }
};

static inline uint32_t grid_for(uint64_t n) { return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(cdiv(n, CG_BLK), 1), 1u << 20); }

// sorted (key, base index) pairs of all k-windows of the selected contigs; device arrays in *keys / *vals (owned by bufs)
static int sorted_windows(shn_ctx* ctx, DevBufs& bufs, const uint8_t* d_bases, const uint64_t* d_off, const uint32_t* d_cid,
                          const int32_t* d_use, uint64_t total, int k, uint64_t** keys, uint32_t** vals, uint64_t* n_out) {
  hipStream_t s = ctx->stream;
  uint32_t* d_flag; uint64_t* d_pos; unsigned long long* d_bad;
  HIP_TRY(bufs.get(&d_flag, (total + 1) * 4));
  HIP_TRY(bufs.get(&d_pos, (total + 2) * 8));
  HIP_TRY(bufs.get(&d_bad, 8));
  HIP_TRY(hipMemsetAsync(d_bad, 0, 8, s));
  hipLaunchKernelGGL(cg_flag_kernel, dim3(grid_for(total)), dim3(CG_BLK), 0, s, d_bases, d_off, d_cid, d_use, total, k, d_flag);
  uint64_t nv = 0;
  int rc = shn_device_scan_u32(ctx, d_flag, total, d_pos, &nv);
  if (rc) return rc;
  *n_out = nv;
  *keys = nullptr; *vals = nullptr;
  if (!nv) return SHN_OK;
  uint64_t *k1, *k2; uint32_t *v1, *v2;
  HIP_TRY(bufs.get(&k1, nv * 8)); HIP_TRY(bufs.get(&k2, nv * 8)); HIP_TRY(bufs.get(&v1, nv * 4)); HIP_TRY(bufs.get(&v2, nv * 4));
  hipLaunchKernelGGL(cg_keys_kernel, dim3(grid_for(total)), dim3(CG_BLK), 0, s, d_bases, d_flag, d_pos, total, k, k1, v1, d_bad);
  unsigned long long bad = 0;
  HIP_TRY(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (bad) return shn_fail(SHN_ERR_ARG, "shn_contig_stage: contig with a base outside ACGT");
  if ((rc = shn_sort_pairs(ctx, k1, v1, k2, v2, nv, 0, 2 * k))) return rc;
  *keys = k1; *vals = v1;
  return SHN_OK;
}

}  // namespace

// Same contract as shn_cgraph_add on a fresh graph: candidates in seed order in one call; accepted_out[i] = 1-based accepted
// index or 0, best_counts_out[i] (may be NULL) = hit count of the candidate's `best` contig; *out receives the contig graph
// (shn_cgraph_sizes / shn_cgraph_export / shn_cgraph_destroy).
extern "C" int shn_contig_stage(shn_ctx* ctx, const uint8_t* bases, const uint64_t* off, uint64_t n_cand, int k1, int r, double f,
                                int32_t* accepted_out, int32_t* best_counts_out, shn_cgraph** out) {
  if (!ctx || !out || k1 < 2 || k1 > 33 || r < 1 || r > 32 || (n_cand && (!bases || !off || !accepted_out)))
    return shn_fail(SHN_ERR_ARG, "shn_contig_stage: bad argument");
  *out = nullptr;
  const bool dbg = getenv("SHN_DEBUG") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t0 = now();
  auto lap = [&](const char* what) { if (dbg) { double t = now(); fprintf(stderr, "[contig_stage] %-34s %8.3f s\n", what, t - t0); t0 = t; } };
  shn_cgraph* G = new shn_cgraph(k1, r, f);
  struct Guard { shn_cgraph* g; ~Guard() { delete g; } } guard{G};
  std::vector<int32_t> best_tmp;
  if (!best_counts_out) { best_tmp.assign(n_cand + 1, 0); best_counts_out = best_tmp.data(); }
  for (uint64_t i = 0; i < n_cand; i++) { accepted_out[i] = 0; best_counts_out[i] = 0; }
  const uint64_t total = n_cand ? off[n_cand] - off[0] : 0;
  if (!n_cand || !total) { *out = G; guard.g = nullptr; return SHN_OK; }
  if (off[0] != 0) return shn_fail(SHN_ERR_ARG, "shn_contig_stage: offsets must start at 0");
  if (total >= 0xFFFFFFF0ULL || n_cand >= 0x7FFFFFF0ULL) return shn_fail(SHN_ERR_OVERFLOW, "shn_contig_stage: more than 2^32 contig bases");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  TimerRegion treg(ctx, T_CONTIG);
  DevBufs bufs;
  uint8_t* d_bases; uint64_t* d_off; uint32_t *d_cid, *d_lab;
  HIP_TRY(bufs.get(&d_bases, total + 64));
  HIP_TRY(bufs.get(&d_off, (n_cand + 1) * 8));
  HIP_TRY(bufs.get(&d_cid, (total + 1) * 4));
  HIP_TRY(bufs.get(&d_lab, (n_cand + 1) * 4));
  HIP_TRY(hipMemcpyAsync(d_bases, bases, total, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_off, off, (n_cand + 1) * 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(cg_cid_kernel, dim3((uint32_t)cdiv(n_cand * 64, CG_BLK)), dim3(CG_BLK), 0, s, d_off, n_cand, d_cid);
  hipLaunchKernelGGL(cg_iota_kernel, dim3((uint32_t)cdiv(n_cand, CG_BLK)), dim3(CG_BLK), 0, s, d_lab, n_cand);

  // ---- clusters of candidates that share an r-mer
  std::vector<uint32_t> lab(n_cand);
  {
    DevBufs tmp;
    uint64_t* keys; uint32_t* vals; uint64_t nv = 0;
    int rc = sorted_windows(ctx, tmp, d_bases, d_off, d_cid, nullptr, total, r, &keys, &vals, &nv);
    if (rc) return rc;
    if (nv > 1) hipLaunchKernelGGL(cg_union_kernel, dim3(grid_for(nv)), dim3(CG_BLK), 0, s, keys, vals, nv, d_cid, d_lab);
    hipLaunchKernelGGL(cg_flatten_kernel, dim3((uint32_t)cdiv(n_cand, CG_BLK)), dim3(CG_BLK), 0, s, d_lab, n_cand);
    HIP_TRY(hipMemcpyAsync(lab.data(), d_lab, n_cand * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  lap("r-mer sort + clusters (GPU)");

  // ---- duplicate_check per cluster (members ascending = seed order), clusters in parallel on host threads
  std::vector<uint32_t> csize(n_cand, 0), cstart(n_cand + 1, 0), members(n_cand);
  for (uint64_t c = 0; c < n_cand; c++) csize[lab[c]]++;
  for (uint64_t c = 0; c < n_cand; c++) cstart[c + 1] = cstart[c] + csize[c];
  { std::vector<uint32_t> cur(cstart.begin(), cstart.end() - 1);
    for (uint64_t c = 0; c < n_cand; c++) members[cur[lab[c]]++] = (uint32_t)c; }
  std::vector<uint8_t> acc(n_cand, 0);
  std::vector<uint32_t> multi;                       // roots of the clusters with more than one candidate
  uint64_t multi_bases = 0, biggest = 0;
  for (uint64_t c = 0; c < n_cand; c++) {
    if (csize[c] == 1) acc[members[cstart[c]]] = 1;
    else if (csize[c] > 1) { multi.push_back((uint32_t)c); biggest = std::max<uint64_t>(biggest, csize[c]); }
  }
  // heaviest clusters first (by number of members; a work queue hands them out)
  std::sort(multi.begin(), multi.end(), [&](uint32_t a, uint32_t b) { return csize[a] != csize[b] ? csize[a] > csize[b] : a < b; });
  {
    std::atomic<size_t> next{0};
    auto worker = [&]() {
      FlatMultiMap rmer(1 << 10);
      std::vector<uint64_t> rk;
      std::vector<int32_t> hits, dupcnt, touched, cov;
      while (true) {
        const size_t q = next.fetch_add(1);
        if (q >= multi.size()) break;
        const uint32_t root = multi[q];
        rmer = FlatMultiMap((size_t)1 << 10);
        int32_t idx = 0;                             // accepted so far in this cluster (local 1-based index, seed order)
        for (uint32_t m = cstart[root]; m < cstart[root + 1]; m++) {
          const uint32_t c = members[m];
          const uint8_t* sq = bases + off[c];
          const uint32_t L = (uint32_t)(off[c + 1] - off[c]);
          window_keys(sq, L, r, rk);
          hits.assign(rk.size(), -1);
          if (dupcnt.size() < (size_t)idx + 1) dupcnt.resize((size_t)idx + 1, 0);
          int32_t max_till_now = 0, best = -1;
          for (size_t i = 0; i < rk.size(); i++) {
            int32_t v = rmer.find(rk[i]);
            hits[i] = v;
            for (; v != -1; v = rmer.nxt(v)) {
              const int32_t d = rmer.va(v);
              if (dupcnt[d] == 0) touched.push_back(d);
              const int32_t cnt = ++dupcnt[d];
              if (cnt >= max_till_now) { max_till_now = cnt; best = d; }      // `>=`: the latest wins (:258-259)
            }
          }
          for (int32_t d : touched) dupcnt[d] = 0;
          touched.clear();
          best_counts_out[c] = max_till_now;
          bool suspect = false;
          if (best >= 0) {
            cov.assign(L + 1, 0);
            for (size_t i = 0; i < rk.size(); i++) {
              bool has = false;
              for (int32_t v = hits[i]; v != -1 && !has; v = rmer.nxt(v)) has = rmer.va(v) == best;
              if (has) { cov[i] += 1; cov[i + r] -= 1; }
            }
            int64_t run = 0, covered = 0;
            for (uint32_t i = 0; i < L; i++) { run += cov[i]; if (run > 0) covered++; }
            suspect = (double)covered > f * (double)L;
          }
          if (!suspect) {
            acc[c] = 1;
            idx++;
            for (size_t i = 0; i < rk.size(); i++) rmer.add(rk[i], idx);
          }
        }
      }
    };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nt = (unsigned)std::min<size_t>(std::min(64u, hw), std::max<size_t>(1, multi.size()));
    if (nt <= 1) worker();
    else { std::vector<std::thread> th; for (unsigned t = 0; t < nt; t++) th.emplace_back(worker); for (auto& x : th) x.join(); }
  }
  std::vector<int32_t> use(n_cand);
  int32_t n_acc = 0;
  for (uint64_t c = 0; c < n_cand; c++) { use[c] = acc[c] ? ++n_acc : 0; accepted_out[c] = use[c]; }
  if (dbg) fprintf(stderr, "[contig_stage] %llu candidates (%llu bases): %zu clusters of >1 (largest %llu), accepted %d\n",
                   (unsigned long long)n_cand, (unsigned long long)total, multi.size(), (unsigned long long)biggest, n_acc);
  (void)multi_bases;
  lap("duplicate_check per cluster (host)");

  // ---- contig_connections: K-mers occurring in two different accepted contigs
  G->idx = n_acc;
  G->conns.assign((size_t)n_acc + 1, Conn());
  G->n_cand_total = n_cand;
  const int C = k1 - 1;
  if (n_acc > 1) {
    int32_t* d_use;
    HIP_TRY(bufs.get(&d_use, n_cand * 4));
    HIP_TRY(hipMemcpyAsync(d_use, use.data(), n_cand * 4, hipMemcpyHostToDevice, s));
    DevBufs tmp;
    uint64_t* keys; uint32_t* vals; uint64_t nv = 0;
    int rc = sorted_windows(ctx, tmp, d_bases, d_off, d_cid, d_use, total, C, &keys, &vals, &nv);
    if (rc) return rc;
    uint64_t ns = 0;
    std::vector<uint64_t> hk; std::vector<uint32_t> hg;
    if (nv > 1) {
      uint32_t* d_flag; uint64_t* d_pos;
      HIP_TRY(tmp.get(&d_flag, (nv + 1) * 4));
      HIP_TRY(tmp.get(&d_pos, (nv + 2) * 8));
      hipLaunchKernelGGL(cg_shared_kernel, dim3(grid_for(nv)), dim3(CG_BLK), 0, s, keys, vals, nv, d_cid, d_flag);
      if ((rc = shn_device_scan_u32(ctx, d_flag, nv, d_pos, &ns))) return rc;
      if (ns) {
        uint64_t* ok; uint32_t* ov;
        HIP_TRY(tmp.get(&ok, ns * 8)); HIP_TRY(tmp.get(&ov, ns * 4));
        hipLaunchKernelGGL(cg_compact_kernel, dim3(grid_for(nv)), dim3(CG_BLK), 0, s, keys, vals, d_flag, d_pos, nv, ok, ov);
        hk.resize(ns); hg.resize(ns);
        HIP_TRY(hipMemcpyAsync(hk.data(), ok, ns * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(hg.data(), ov, ns * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
      }
    }
    lap("K-mer sort + shared runs (GPU)");
    // host replay of the reference's loop over the shared runs only.  Entry e: K-mer run `run[e]`, accepted contig `ea[e]`,
    // position `ep[e]`; inside a run the entries are in (contig, position) order = the order of cmer_to_contig's list.
    if (ns) {
      std::vector<uint32_t> run(ns), ea(ns), ep(ns), run_start;
      for (uint64_t e = 0; e < ns; e++) {
        if (e == 0 || hk[e] != hk[e - 1]) run_start.push_back((uint32_t)e);
        run[e] = (uint32_t)run_start.size() - 1;
        const uint64_t g = hg[e];
        const uint64_t c = (uint64_t)(std::upper_bound(off, off + n_cand + 1, g) - off) - 1;
        ea[e] = (uint32_t)use[c];
        ep[e] = (uint32_t)(g - off[c]);
      }
      run_start.push_back((uint32_t)ns);
      // the shared windows of every accepted contig in position order (counting sort by contig, then by position)
      std::vector<uint32_t> order(ns);
      for (uint64_t e = 0; e < ns; e++) order[e] = (uint32_t)e;
      std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return ea[x] != ea[y] ? ea[x] < ea[y] : ep[x] < ep[y]; });
      std::vector<int32_t> connw((size_t)n_acc + 1, 0), newnb;
      size_t q = 0;
      while (q < ns) {
        const uint32_t a = ea[order[q]];
        newnb.clear();
        for (; q < ns && ea[order[q]] == a; q++) {
          const uint32_t rn = run[order[q]];
          for (uint32_t e = run_start[rn]; e < run_start[rn + 1]; e++) {
            const int32_t c2 = (int32_t)ea[e];
            if (c2 >= (int32_t)a) break;                 // later contigs are not in the index yet; own occurrences are skipped
            if (connw[c2]++ == 0) newnb.push_back(c2);
            Conn& b = G->conns[c2];
            if (!b.nb.empty() && b.nb.back() == (int32_t)a) b.w.back()++;
            else { b.nb.push_back((int32_t)a); b.w.push_back(1); }
          }
        }
        // contig `a` was accepted before any later contig touched its list: its own neighbours come first
        Conn& A = G->conns[a];
        std::vector<int32_t> nb(newnb), w(newnb.size());
        for (size_t j = 0; j < newnb.size(); j++) { w[j] = connw[newnb[j]]; connw[newnb[j]] = 0; }
        nb.insert(nb.end(), A.nb.begin(), A.nb.end());
        w.insert(w.end(), A.w.begin(), A.w.end());
        A.nb.swap(nb); A.w.swap(w);
      }
    }
    lap("contig_connections replay (host)");
  }
  HIP_TRY(hipGetLastError());
  *out = G;
  guard.g = nullptr;
  return SHN_OK;
}
