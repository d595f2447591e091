// k1mers2component on the GPU (row a9, kmers_for_component.py:244-305): every k1-window of every partition contig -> the set
// of partitions it occurs in, as the probe table the routing kernel reads (value = set id + 1).  The host version sorted
// ~60 M (k1-mer, partition) codes with numpy at 20,000 genes (4.4 s); here the k1-windows of all partition contigs are
// sorted once on the device.  A k1-mer lies on exactly one contig, so its set has one partition (remaining bins) or two
// (gpmetis run + r2 run of a big component): singleton sets are set p = {p}; the k1-mers of several partitions are few, they
// go to the host where their sets are interned.
#include "common.h"
#include <algorithm>
#include <cstring>
#include <map>

#define PG_BLK 256
#define PG_FOR(i, n) for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (uint64_t)gridDim.x * blockDim.x)

struct shn_probe {
  shn_table* table = nullptr;
  std::vector<uint32_t> set_off, set_mem;
};

namespace {
static inline uint32_t pg_grid(uint64_t n) { return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(cdiv(n, PG_BLK), 1), 1u << 20); }

// run heads of the sorted windows: one entry per distinct k1-mer; single[i] = 1 if all its windows lie in one partition
__global__ void pg_head_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ cid,
                               const uint32_t* __restrict__ part_of, uint64_t n, uint32_t* __restrict__ head, uint32_t* __restrict__ multi) {
  PG_FOR(i, n) {
    if (i && keys[i] == keys[i - 1]) { head[i] = 0; multi[i] = 0; continue; }
    head[i] = 1;
    uint64_t e = i + 1;
    while (e < n && keys[e] == keys[i]) e++;
    // (stable sort + contigs grouped by ascending partition: a run's partitions ascend)
    multi[i] = part_of[cid[vals[i]]] != part_of[cid[vals[e - 1]]] ? 1u : 0u;
  }
}
__global__ void pg_unique_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ cid,
                                 const uint32_t* __restrict__ part_of, const uint32_t* __restrict__ head, const uint64_t* __restrict__ hpos, uint64_t n,
                                 uint64_t* __restrict__ ukeys, uint32_t* __restrict__ uvals) {
  PG_FOR(i, n) if (head[i]) { ukeys[hpos[i]] = keys[i]; uvals[hpos[i]] = part_of[cid[vals[i]]] + 1; }    // set p = {p}; multi ones are patched
}
// number of different partitions among the windows of the run that starts at i (they come in ascending partition order: the sort
// is stable and the contigs were laid out partition by partition); the first two in p1 / p2
__device__ __forceinline__ uint32_t pg_run_parts(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ cid,
                                                 const uint32_t* __restrict__ part_of, uint64_t n, uint64_t i, uint32_t& p1, uint32_t& p2) {
  uint32_t nd = 0, last = 0xFFFFFFFFu;
  for (uint64_t e = i; e < n && keys[e] == keys[i]; e++) {
    const uint32_t p = part_of[cid[vals[e]]];
    if (p == last) continue;
    if (nd == 0) p1 = p; else if (nd == 1) p2 = p;
    nd++; last = p;
  }
  return nd;
}
// the k1-mers that occur in exactly TWO partitions -- every k1-mer of a component that went through gpmetis lies in its partition
// of the first run and in its partition of the second (kmers_for_component.py:244-305): at `bench.py --config 2p` that is all 54 M
// of them, and interning their sets on the host (a sort of 108 M windows and a map look-up per k1-mer) took 7 s of a step.  Their
// sets are pairs: (pair, unique index) here, sorted by pair on the device, one set id per run of equal pairs.
__global__ void pg_pair_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ cid,
                               const uint32_t* __restrict__ part_of, const uint32_t* __restrict__ multi, const uint64_t* __restrict__ hpos,
                               uint64_t n, unsigned long long* __restrict__ n_out, uint64_t* __restrict__ o_pair, uint32_t* __restrict__ o_uidx) {
  PG_FOR(i, n) {
    if (!multi[i]) continue;
    uint32_t p1 = 0, p2 = 0;
    if (pg_run_parts(keys, vals, cid, part_of, n, i, p1, p2) != 2) continue;
    const unsigned long long at = atomicAdd(n_out, 1ULL);
    o_pair[at] = ((uint64_t)p1 << 32) | p2;
    o_uidx[at] = (uint32_t)hpos[i];
  }
}
__global__ void pg_pair_head_kernel(const uint64_t* __restrict__ pairs, uint64_t n, uint32_t* __restrict__ head) {
  PG_FOR(i, n) head[i] = (i == 0 || pairs[i] != pairs[i - 1]) ? 1u : 0u;
}
// value of every such k1-mer = id of its pair's set + 1; the distinct pairs, in order, for the host
__global__ void pg_pair_patch_kernel(const uint64_t* __restrict__ pairs, const uint32_t* __restrict__ uidx, const uint32_t* __restrict__ head,
                                     const uint64_t* __restrict__ pos, uint64_t n, uint32_t first_sid, uint32_t* __restrict__ uvals,
                                     uint64_t* __restrict__ distinct) {
  PG_FOR(i, n) {
    const uint64_t run = pos[i] - (head[i] ? 0 : 1);
    uvals[uidx[i]] = first_sid + (uint32_t)run + 1;
    if (head[i]) distinct[run] = pairs[i];
  }
}
// the windows of the k1-mers that occur in three or more partitions (none in the reference's flow: a contig is in one partition per
// gpmetis run): (unique index, partition) per window, for the host
__global__ void pg_multi_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ cid,
                                const uint32_t* __restrict__ part_of, const uint32_t* __restrict__ multi, const uint64_t* __restrict__ hpos,
                                uint64_t n, unsigned long long* __restrict__ n_out, uint64_t cap, uint32_t* __restrict__ o_uidx, uint32_t* __restrict__ o_part) {
  PG_FOR(i, n) {
    if (!multi[i]) continue;
    uint32_t p1 = 0, p2 = 0;
    if (pg_run_parts(keys, vals, cid, part_of, n, i, p1, p2) < 3) continue;
    for (uint64_t e = i; e < n && keys[e] == keys[i]; e++) {
      const unsigned long long at = atomicAdd(n_out, 1ULL);
      if (at < cap) { o_uidx[at] = (uint32_t)hpos[i]; o_part[at] = part_of[cid[vals[e]]]; }
    }
  }
}
__global__ void pg_patch_kernel(const uint32_t* __restrict__ uidx, const uint32_t* __restrict__ setid, uint64_t n, uint32_t* __restrict__ uvals) {
  PG_FOR(i, n) uvals[uidx[i]] = setid[i] + 1;
}
}  // namespace

extern "C" void shn_probe_destroy(shn_probe* p) { if (p) { shn_table_destroy(p->table); delete p; } }
extern "C" const shn_table* shn_probe_table(const shn_probe* p) { return p ? p->table : nullptr; }
extern "C" uint32_t shn_probe_n_sets(const shn_probe* p) { return p ? (uint32_t)(p->set_off.size() - 1) : 0; }
extern "C" uint64_t shn_probe_n_members(const shn_probe* p) { return p ? p->set_mem.size() : 0; }
extern "C" int shn_probe_sets(const shn_probe* p, uint32_t* set_off, uint32_t* set_mem) {
  if (!p || !set_off || !set_mem) return shn_fail(SHN_ERR_ARG, "shn_probe_sets: NULL argument");
  std::copy(p->set_off.begin(), p->set_off.end(), set_off);
  std::copy(p->set_mem.begin(), p->set_mem.end(), set_mem);
  return SHN_OK;
}

// bases / off: the contigs of all partitions one after the other (ASCII), part_of[c] = partition of contig c, ascending.
extern "C" int shn_probe_build(shn_ctx* ctx, const uint8_t* bases, const uint64_t* off, uint64_t n_contigs, const uint32_t* part_of, uint32_t n_parts,
                               int k1, shn_probe** out) {
  if (!ctx || !out || (n_contigs && (!bases || !off || !part_of)) || k1 < 2 || k1 > 32) return shn_fail(SHN_ERR_ARG, "shn_probe_build: bad argument");
  *out = nullptr;
  const uint64_t total = n_contigs ? off[n_contigs] : 0;
  if (total >= 0xFFFFFFF0ULL) return shn_fail(SHN_ERR_OVERFLOW, "shn_probe_build: more than 2^32 contig bases");
  for (uint64_t c = 0; c < n_contigs; c++)
    if (part_of[c] >= n_parts || (c && part_of[c] < part_of[c - 1])) return shn_fail(SHN_ERR_ARG, "shn_probe_build: part_of must be ascending and < n_parts");
  SHN_ENTER(ctx);
  shn_stage_begin(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  shn_probe* P = new shn_probe();
  struct Guard { shn_probe* p; ~Guard() { shn_probe_destroy(p); } } guard{P};
  // set p = {p} for every partition; sets of several partitions follow
  P->set_off.resize(n_parts + 1);
  P->set_mem.resize(n_parts);
  for (uint32_t p = 0; p <= n_parts; p++) P->set_off[p] = p;
  for (uint32_t p = 0; p < n_parts; p++) P->set_mem[p] = p;
  ShnDevBufs B(ctx->stream);
  uint64_t* ukeys = nullptr; uint32_t* uvals = nullptr;
  uint64_t nu = 0;
  if (total) {
    uint8_t* d_bases; uint64_t* d_off; uint32_t *d_cid, *d_part;
    HIP_TRY(B.get(&d_bases, total + 64)); HIP_TRY(B.get(&d_off, (n_contigs + 1) * 8)); HIP_TRY(B.get(&d_cid, (total + 1) * 4)); HIP_TRY(B.get(&d_part, n_contigs * 4));
    HIP_TRY(hipMemcpyAsync(d_bases, bases, total, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_off, off, (n_contigs + 1) * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_part, part_of, n_contigs * 4, hipMemcpyHostToDevice, s));
    shn_contig_ids(s, d_off, n_contigs, d_cid);
    uint64_t* keys; uint32_t* vals; uint64_t nv = 0;
    int rc = shn_sorted_windows(ctx, B, d_bases, d_off, d_cid, nullptr, total, k1, &keys, &vals, &nv);
    if (rc) return rc;
    if (nv) {
      uint32_t *d_head, *d_multi; uint64_t *d_hpos;
      HIP_TRY(B.get(&d_head, (nv + 1) * 4)); HIP_TRY(B.get(&d_multi, (nv + 1) * 4)); HIP_TRY(B.get(&d_hpos, (nv + 2) * 8));
      hipLaunchKernelGGL(pg_head_kernel, dim3(pg_grid(nv)), dim3(PG_BLK), 0, s, keys, vals, d_cid, d_part, nv, d_head, d_multi);
      if ((rc = shn_device_scan_u32(ctx, d_head, nv, d_hpos, &nu))) return rc;
      HIP_TRY(B.get(&ukeys, (nu + 1) * 8)); HIP_TRY(B.get(&uvals, (nu + 1) * 4));
      hipLaunchKernelGGL(pg_unique_kernel, dim3(pg_grid(nv)), dim3(PG_BLK), 0, s, keys, vals, d_cid, d_part, d_head, d_hpos, nv, ukeys, uvals);
      unsigned long long* d_nm; uint32_t *d_mu = nullptr, *d_mp = nullptr;
      HIP_TRY(B.get(&d_nm, 8));
      // k1-mers of exactly two partitions: their sets are interned on the device (pg_pair_kernel)
      {
        uint64_t *d_pair, *d_pair2, *d_ppos, *d_dist; uint32_t *d_pu, *d_pu2, *d_phead;
        HIP_TRY(B.get(&d_pair, (nu + 1) * 8)); HIP_TRY(B.get(&d_pair2, (nu + 1) * 8)); HIP_TRY(B.get(&d_pu, (nu + 1) * 4)); HIP_TRY(B.get(&d_pu2, (nu + 1) * 4));
        HIP_TRY(hipMemsetAsync(d_nm, 0, 8, s));
        hipLaunchKernelGGL(pg_pair_kernel, dim3(pg_grid(nv)), dim3(PG_BLK), 0, s, keys, vals, d_cid, d_part, d_multi, d_hpos, nv, d_nm, d_pair, d_pu);
        unsigned long long n_pair = 0;
        HIP_TRY(hipMemcpyAsync(&n_pair, d_nm, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (n_pair) {
          int pbits = 1;
          while ((1ULL << pbits) < (uint64_t)n_parts) pbits++;
          if ((rc = shn_sort_pairs(ctx, d_pair, d_pu, d_pair2, d_pu2, n_pair, 0, 32 + pbits))) return rc;
          HIP_TRY(B.get(&d_phead, (n_pair + 1) * 4)); HIP_TRY(B.get(&d_ppos, (n_pair + 2) * 8));
          hipLaunchKernelGGL(pg_pair_head_kernel, dim3(pg_grid(n_pair)), dim3(PG_BLK), 0, s, d_pair, (uint64_t)n_pair, d_phead);
          uint64_t n_dist = 0;
          if ((rc = shn_device_scan_u32(ctx, d_phead, n_pair, d_ppos, &n_dist))) return rc;
          HIP_TRY(B.get(&d_dist, (n_dist + 1) * 8));
          const uint32_t first_sid = (uint32_t)(P->set_off.size() - 1);
          hipLaunchKernelGGL(pg_pair_patch_kernel, dim3(pg_grid(n_pair)), dim3(PG_BLK), 0, s, d_pair, d_pu, d_phead, d_ppos, (uint64_t)n_pair, first_sid, uvals, d_dist);
          std::vector<uint64_t> dist(n_dist);
          HIP_TRY(hipMemcpyAsync(dist.data(), d_dist, n_dist * 8, hipMemcpyDeviceToHost, s));
          HIP_TRY(hipStreamSynchronize(s));
          for (uint64_t pr : dist) {
            P->set_mem.push_back((uint32_t)(pr >> 32)); P->set_mem.push_back((uint32_t)pr);
            P->set_off.push_back((uint32_t)P->set_mem.size());
          }
        }
      }
      // k1-mers of three or more partitions: their windows to the host, which interns the sets
      uint64_t cap = 1 << 16;
      unsigned long long nm = 0;
      while (true) {
        HIP_TRY(B.get(&d_mu, cap * 4)); HIP_TRY(B.get(&d_mp, cap * 4));
        HIP_TRY(hipMemsetAsync(d_nm, 0, 8, s));
        hipLaunchKernelGGL(pg_multi_kernel, dim3(pg_grid(nv)), dim3(PG_BLK), 0, s, keys, vals, d_cid, d_part, d_multi, d_hpos, nv, d_nm, cap, d_mu, d_mp);
        HIP_TRY(hipMemcpyAsync(&nm, d_nm, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (nm <= cap) break;
        cap = nm;
      }
      if (nm) {
        std::vector<uint32_t> mu(nm), mp(nm);
        HIP_TRY(hipMemcpy(mu.data(), d_mu, nm * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(mp.data(), d_mp, nm * 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> order(nm);
        for (uint64_t i = 0; i < nm; i++) order[i] = (uint32_t)i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return mu[a] != mu[b] ? mu[a] < mu[b] : mp[a] < mp[b]; });
        std::map<std::vector<uint32_t>, uint32_t> ids;
        std::vector<uint32_t> pu, ps, cur;
        for (uint64_t i = 0; i < nm;) {
          const uint32_t u = mu[order[i]];
          cur.clear();
          for (; i < nm && mu[order[i]] == u; i++) if (cur.empty() || cur.back() != mp[order[i]]) cur.push_back(mp[order[i]]);
          auto it = ids.find(cur);
          uint32_t sid;
          if (it != ids.end()) sid = it->second;
          else {
            sid = (uint32_t)(P->set_off.size() - 1);
            ids[cur] = sid;
            P->set_mem.insert(P->set_mem.end(), cur.begin(), cur.end());
            P->set_off.push_back((uint32_t)P->set_mem.size());
          }
          pu.push_back(u); ps.push_back(sid);
        }
        uint32_t *d_pu, *d_ps;
        HIP_TRY(B.get(&d_pu, pu.size() * 4)); HIP_TRY(B.get(&d_ps, ps.size() * 4));
        HIP_TRY(hipMemcpyAsync(d_pu, pu.data(), pu.size() * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_ps, ps.data(), ps.size() * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(pg_patch_kernel, dim3(pg_grid(pu.size())), dim3(PG_BLK), 0, s, d_pu, d_ps, (uint64_t)pu.size(), uvals);
        HIP_TRY(hipStreamSynchronize(s));
      }
    }
  }
  uint64_t dummy_k = 0; uint32_t dummy_v = 0;
  (void)dummy_k; (void)dummy_v;
  if (!ukeys) { HIP_TRY(B.get(&ukeys, 8)); HIP_TRY(B.get(&uvals, 4)); }
  int rc = shn_table_from_pairs(ctx, ukeys, uvals, nu, k1, 0, &P->table);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  *out = P;
  guard.p = nullptr;
  return SHN_OK;
}
