// Context, error handling, event timers and the read uploader / 2-bit packer.
#include "common.h"
#include <deque>
#include <algorithm>
#include <map>
#include "graph_dev.h"
#include <sched.h>
#include <mutex>
#include <vector>
#include <cstring>
#include <cstdio>
#include <thread>
#include <atomic>

static thread_local std::string g_err;
void shn_set_error(const std::string& msg) { g_err = msg; }
int shn_fail(int code, const std::string& msg) { g_err = msg; return code; }

extern "C" const char* shn_last_error(void) { return g_err.c_str(); }
extern "C" const char* shn_version(void) { return "shannon_hip 0.1.0 (gfx950)"; }

static const char* kTimerNames[T_N] = {
  "pack", "count.hist1", "count.scatter1", "count.hist2", "count.scatter2", "count.buckets", "count.compact",
  "count.total", "table.lookup", "extend", "route", "graph", "lp", "extend.prepare", "extend.sort", "extend.walk", "graph.seeds",
  "extend.walk_thread", "extend.walk_wave", "extend.mark", "extend.emit", "table.build", "count.direct", "contig.stage", "graph.unitigs", "extend.adjacency",
  "count.sk_hist", "count.sk_emit", "count.sk_hist2", "count.sk_scatter2", "count.sk_buckets", "count.sk_big",
  // round 5: one timer per template instance where the launches of a step differ by an order of magnitude -- the first (bulk) round of
  // a rank block apart from the re-run rounds (ext_walk_kernel<true> / <false>), the begin pass, the second table size of the buckets
  "extend.walk_fresh", "extend.begin", "count.sk_buckets2",
  "contig.sort", "contig.hits", "contig.cover", "contig.compact", "graph.kp_search", "graph.kp_classify", "graph.seed_scan", "graph.dd_insert", "lp.trials",
  "extend.audit"};
extern "C" const char* shn_timer_name(int slot) {
  if (slot < 0 || slot >= T_N || !kTimerNames[slot]) return "";
  return kTimerNames[slot];
}

TimerRegion::TimerRegion(shn_ctx* ctx, int s) : TimerRegion(ctx, s, ctx->stream) { shn_use_stream(ctx->stream); }
TimerRegion::TimerRegion(shn_ctx* ctx, int s, hipStream_t stream) : c(ctx), slot(s), a(nullptr), b(nullptr), st(stream) {
  if (!c->timing) return;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventRecord(a, st);
}
TimerRegion::~TimerRegion() {
  if (!a) return;
  hipEventRecord(b, st);
  std::lock_guard<std::mutex> lk(c->tmu);
  auto& q = c->pending[slot];
  q.push_back({a, b});
  // a caller that never reads its timers (the CLI, a service looping over steps, the forks of kept graph threads) must not pile up
  // events without bound: beyond 64 pending regions of a slot, the finished ones at the front are folded into the sums here
  if (q.size() > 64) {
    size_t done = 0;
    while (done + 1 < q.size() && hipEventQuery(q[done].second) == hipSuccess) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, q[done].first, q[done].second) == hipSuccess) { c->ms[slot] += ms; c->regions[slot]++; }
      hipEventDestroy(q[done].first); hipEventDestroy(q[done].second);
      done++;
    }
    (void)hipGetLastError();                                   // (hipErrorNotReady of the query that ended the loop)
    if (done) q.erase(q.begin(), q.begin() + (ptrdiff_t)done);
  }
}

// The caching allocator (shared by every context and host thread of the process, one process per GPU).  A block remembers the
// streams that may still have work queued on it: the stream it was handed out on and the current stream of the thread that freed
// it.  The next caller on one of those streams gets it at once (stream order makes that safe whatever is still queued); a caller
// on ANOTHER stream gets it behind an event on each of them, waited for by its own stream (hipStreamWaitEvent: nobody blocks on
// the host).  For the stream of a forked context (a graph thread's: a handful of short passes queued at any time) that event is
// recorded at the moment of the reuse -- everything queued there so far, a superset of what was queued at the free -- so that a
// free on such a stream costs no HIP call; for an owner's stream it is recorded AT THE FREE (round 6): with two batches in flight
// (bench.py's overlap) a graph thread that took a block last used by the other batch's extension otherwise waited for that
// stream's whole queue -- hundreds of milliseconds of walker launches issued long after the free -- and the graph stage beside a
// front half took 1.05 s instead of 0.2.  An event that has completed by the time of the reuse costs no wait at all.  (Until
// round 4 a freed block went to the next caller at once and correctness rested on every caller having synchronised its stream
// before every free, on its error paths too.)  A stream that goes away (a forked context with its host thread) is synchronised
// first and struck from the blocks (shn_stream_retired).
// The "current stream" of a host thread is the stream of the context it last entered the library with (shn_use_stream: set by
// SHN_ENTER, TimerRegion, shn_thread_ctx, every `s = ctx->stream`); a caller that knows better passes the stream.
// Debug switches (environment): SHN_DEV_POISON=<byte 0..255>: every block handed out (and every workspace slot that has just grown;
// SHN_DEV_POISON_WS=1: on every request) is filled with that byte first -- a kernel that reads what it never wrote then reads the same
// garbage on every run instead of what the last user left behind; SHN_DEV_NOCACHE=1: no reuse at all; SHN_DEV_LEGACY=1: the
// allocator of rounds 1-4 (no ordering).  A block freed twice is reported on stderr and counted (shn_debug_counter(0)) -- the second
// free would hand a block that is in use to the next caller.
namespace {
struct DevBlock { void* p; size_t cap; bool used; hipStream_t stream; hipStream_t pend[2]; int n_pend; hipEvent_t ev[2]; };
std::vector<DevBlock> g_blocks;
std::mutex g_blocks_mu;
std::vector<hipEvent_t> g_ev_pool;
std::deque<hipEvent_t> g_ev_free;                // events of frees, given back at the reuse; one is recorded again only after 256 others
std::vector<hipStream_t> g_fork_streams;         // (g_blocks_mu) streams of forked contexts: short queues, ordered at the reuse
std::atomic<uint64_t> g_dbg[8];
thread_local hipStream_t t_stream = nullptr;
int poison_byte() { static const int v = getenv("SHN_DEV_POISON") ? (atoi(getenv("SHN_DEV_POISON")) & 255) : -1; return v; }
bool legacy_reuse() { static const bool v = getenv("SHN_DEV_LEGACY") && getenv("SHN_DEV_LEGACY")[0] == '1'; return v; }   // (A/B: the allocator of rounds 1-4 -- a freed block goes to the next caller at once)
bool no_cache() { static const bool v = getenv("SHN_DEV_NOCACHE") && getenv("SHN_DEV_NOCACHE")[0] == '1'; return v; }
bool free_events() { static const bool v = !(getenv("SHN_DEV_FREE_EVENTS") && getenv("SHN_DEV_FREE_EVENTS")[0] == '0'); return v; }   // (A/B: 0 = every hand-over ordered at the reuse, as in round 5)
bool is_fork_stream(hipStream_t s) { for (hipStream_t f : g_fork_streams) if (f == s) return true; return false; }
hipEvent_t take_event() {
  hipEvent_t e = nullptr;
  if (g_ev_free.size() > 256) { e = g_ev_free.front(); g_ev_free.pop_front(); return e; }
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return e;
}
void give_event(hipEvent_t e) { if (e) g_ev_free.push_back(e); }
// (g_blocks_mu held) make `asker` wait for what the block's pending streams had queued at the free (an owner's stream) or have queued
// so far (a fork's); false: no event to be had (the caller skips the block)
bool order_behind(DevBlock& b, hipStream_t asker) {
  if (legacy_reuse()) { for (int i = 0; i < b.n_pend; i++) { give_event(b.ev[i]); b.ev[i] = nullptr; } b.n_pend = 0; return true; }
  for (int i = 0; i < b.n_pend; i++) {
    if (b.pend[i] == asker) continue;
    if (b.ev[i]) {
      const hipError_t q = hipEventQuery(b.ev[i]);
      if (q == hipSuccess) { g_dbg[5].fetch_add(1); continue; }                 // hand-overs that needed no wait
      (void)hipGetLastError();
      if (hipStreamWaitEvent(asker, b.ev[i], 0) != hipSuccess) { (void)hipGetLastError(); g_dbg[3].fetch_add(1); return false; }
      g_dbg[4].fetch_add(1);                   // cross-stream hand-overs
      continue;
    }
    // (a ring of 256 events: one is recorded again only after 255 other hand-overs -- long after the wait queued on its last record has run)
    static size_t ring_at = 0;
    hipEvent_t e = nullptr;
    if (g_ev_pool.size() < 256) {
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
      g_ev_pool.push_back(e);
    } else e = g_ev_pool[ring_at++ & 255];
    const bool ok = hipEventRecord(e, b.pend[i]) == hipSuccess && hipStreamWaitEvent(asker, e, 0) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); g_dbg[3].fetch_add(1); return false; }
    g_dbg[4].fetch_add(1);                     // cross-stream hand-overs
  }
  for (int i = 0; i < b.n_pend; i++) { give_event(b.ev[i]); b.ev[i] = nullptr; }
  b.n_pend = 0;
  return true;
}
}
void shn_fork_stream_added(hipStream_t s) { std::lock_guard<std::mutex> lk(g_blocks_mu); g_fork_streams.push_back(s); }
namespace {
struct CopyCensus {
  const bool on = getenv("SHN_COPY_CENSUS") != nullptr;
  std::mutex mu;
  std::map<std::pair<std::string, int>, std::pair<uint64_t, uint64_t>> sites;      // (file, line) -> calls, bytes
  void add(const char* file, int line, size_t n) {
    std::lock_guard<std::mutex> lk(mu);
    const char* base = strrchr(file, '/');
    auto& v = sites[{std::string(base ? base + 1 : file), line}];
    v.first++; v.second += n;
  }
  ~CopyCensus() {
    if (!on) return;
    std::vector<std::pair<uint64_t, std::string>> rows;
    uint64_t calls = 0;
    for (auto& kv : sites) {
      char b[160];
      snprintf(b, sizeof b, "%-22s:%5d %10llu calls %14llu bytes", kv.first.first.c_str(), kv.first.second, (unsigned long long)kv.second.first, (unsigned long long)kv.second.second);
      rows.push_back({kv.second.first, b}); calls += kv.second.first;
    }
    std::sort(rows.begin(), rows.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
    fprintf(stderr, "[copy census] %llu copies / fills at %zu call sites\n", (unsigned long long)calls, rows.size());
    for (size_t i = 0; i < rows.size() && i < 60; i++) fprintf(stderr, "[copy census] %s\n", rows[i].second.c_str());
  }
};
CopyCensus g_copy_census;
}
hipError_t shn_counted_memcpy(const char* file, int line, void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s) {
  if (g_copy_census.on) g_copy_census.add(file, line, n);
  return (hipMemcpyAsync)(dst, src, n, kind, s);
}
hipError_t shn_counted_memset(const char* file, int line, void* dst, int value, size_t n, hipStream_t s) {
  if (g_copy_census.on) g_copy_census.add(file, line, n);
  return (hipMemsetAsync)(dst, value, n, s);
}
void shn_use_stream(hipStream_t s) { t_stream = s; }
hipStream_t shn_current_stream() { return t_stream; }
extern "C" uint64_t shn_debug_counter(int i) { return (i >= 0 && i < 8) ? g_dbg[i].load() : 0; }
void shn_debug_count(int i) { if (i >= 0 && i < 8) g_dbg[i].fetch_add(1); }
void shn_poison(void* p, size_t bytes, hipStream_t s) {
  const int v = poison_byte();
  if (v >= 0 && p && bytes) (void)hipMemsetAsync(p, v, bytes, s);
}
// a stream is about to be destroyed (the caller has synchronised it): nothing is pending on it any more
void shn_stream_retired(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_blocks_mu);
  for (auto& b : g_blocks) {
    int k = 0;
    for (int i = 0; i < b.n_pend; i++) {
      if (b.pend[i] != s) { b.pend[k] = b.pend[i]; b.ev[k] = b.ev[i]; k++; }
      else give_event(b.ev[i]);
    }
    for (int i = k; i < 2; i++) b.ev[i] = nullptr;
    b.n_pend = k;
    if (b.stream == s) b.stream = nullptr;         // (a block that outlives the stream it was handed out on: nothing of that stream is left to wait for)
  }
  for (size_t i = 0; i < g_fork_streams.size(); i++) if (g_fork_streams[i] == s) { g_fork_streams.erase(g_fork_streams.begin() + (ptrdiff_t)i); break; }
}
hipError_t shn_dev_malloc_on(void** p, size_t bytes, hipStream_t stream) {
  if (bytes == 0) bytes = 1;
  if (!no_cache()) {
    std::lock_guard<std::mutex> lk(g_blocks_mu);
    // best fit among the free blocks; among equal sizes one whose pending streams are the asker's own (no event needed)
    int best = -1; bool best_own = false;
    for (size_t i = 0; i < g_blocks.size(); i++) {
      DevBlock& b = g_blocks[i];
      if (b.used || b.cap < bytes || b.cap > 2 * bytes + (1u << 20)) continue;
      bool own = true;
      for (int j = 0; j < b.n_pend; j++) own = own && b.pend[j] == stream;
      if (best < 0 || b.cap < g_blocks[best].cap || (b.cap == g_blocks[best].cap && own && !best_own)) { best = (int)i; best_own = own; }
    }
    if (best >= 0 && order_behind(g_blocks[best], stream)) {
      DevBlock& b = g_blocks[best];
      b.used = true; b.stream = stream; *p = b.p;
      shn_poison(b.p, b.cap, stream);
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {                       // make room: drop what the cache holds, then the workspaces of earlier stages
    (void)hipGetLastError();
    shn_dev_trim();
    e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      shn_ws_release_idle();
      e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess) return e;
  }
  shn_poison(*p, bytes, stream);
  std::lock_guard<std::mutex> lk(g_blocks_mu);
  g_blocks.push_back(DevBlock{*p, bytes, true, stream, {nullptr, nullptr}, 0, {nullptr, nullptr}});
  return hipSuccess;
}
hipError_t shn_dev_malloc_raw(void** p, size_t bytes) { return shn_dev_malloc_on(p, bytes, t_stream); }
hipError_t shn_hip_malloc(void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess) return e;
  (void)hipGetLastError();
  shn_dev_trim();
  e = hipMalloc(p, bytes);
  if (e == hipSuccess) return e;
  (void)hipGetLastError();
  shn_ws_release_idle();
  return hipMalloc(p, bytes);
}
void shn_dev_free_on(void* p, hipStream_t stream) {
  if (!p) return;
  std::unique_lock<std::mutex> lk(g_blocks_mu);
  for (size_t i = 0; i < g_blocks.size(); i++) {
    DevBlock& b = g_blocks[i];
    if (b.p != p) continue;
    if (!b.used) {
      g_dbg[0].fetch_add(1);
      fprintf(stderr, "[shannon_hip] shn_dev_free: block %p (%zu bytes) freed twice\n", p, b.cap);
      return;
    }
    // what may still be queued on the block: on the stream it was handed out on and on the freeing thread's stream
    // (plus whatever was pending when its own stream took it back and has not been waited for since)
    hipStream_t on[4]; int n_on = 0;
    auto add = [&](hipStream_t s) { for (int j = 0; j < n_on; j++) if (on[j] == s) return; on[n_on++] = s; };
    add(b.stream); add(stream);
    for (int j = 0; j < b.n_pend; j++) add(b.pend[j]);
    if (n_on > 2 || no_cache()) {               // more streams than a block remembers (or no cache): wait here, once
      DevBlock gone = b;
      lk.unlock();
      for (int j = 0; j < n_on; j++) (void)hipStreamSynchronize(on[j]);
      lk.lock();
      for (size_t k = 0; k < g_blocks.size(); k++) if (g_blocks[k].p == p) {
        if (no_cache()) { g_blocks.erase(g_blocks.begin() + (ptrdiff_t)k); lk.unlock(); (void)hipFree(gone.p); return; }
        g_blocks[k].n_pend = 0; g_blocks[k].ev[0] = g_blocks[k].ev[1] = nullptr; g_blocks[k].used = false; break;
      }
      return;
    }
    b.n_pend = n_on;
    for (int j = 0; j < n_on; j++) {
      b.pend[j] = on[j];
      b.ev[j] = nullptr;
      if (free_events() && !legacy_reuse() && !is_fork_stream(on[j])) {          // an owner's stream: what is queued on it NOW is all the block has to wait for
        hipEvent_t e = take_event();
        if (e && hipEventRecord(e, on[j]) == hipSuccess) { b.ev[j] = e; g_dbg[6].fetch_add(1); }
        else { (void)hipGetLastError(); give_event(e); }
      }
    }
    b.used = false;
    return;
  }
  lk.unlock();
  g_dbg[1].fetch_add(1);
  fprintf(stderr, "[shannon_hip] shn_dev_free: %p is not a block of the caching allocator\n", p);
  hipFree(p);                                  // not ours (should not happen)
}
void shn_dev_free(void* p) { shn_dev_free_on(p, t_stream); }
void shn_dev_trim() {
  std::lock_guard<std::mutex> lk(g_blocks_mu);
  std::vector<DevBlock> keep;
  for (auto& b : g_blocks) { if (b.used) keep.push_back(b); else { for (int i = 0; i < b.n_pend; i++) give_event(b.ev[i]); hipFree(b.p); } }      // (hipFree waits for the device)
  g_blocks.swap(keep);
}

// ---- test hooks of the allocator (tests/test_allocator_gpu.py): a block of the caching allocator on a context's stream, and a
// fill kernel that can be made slow (every thread spins `spin` times before it writes) -- a block freed while such a kernel is
// queued on one stream and taken at once by another is exactly the hazard the stream ordering above removes
__global__ void shn_debug_fill_kernel(uint32_t* __restrict__ p, uint64_t n, uint32_t value, uint32_t spin) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t x = (uint32_t)i;
  for (uint32_t s = 0; s < spin; s++) x = x * 1664525u + 1013904223u;
  p[i] = value + (x == 0xDEADBEEFu && spin == 0xFFFFFFFFu ? 1u : 0u);       // (x keeps the loop alive)
}
extern "C" int shn_debug_alloc(shn_ctx* ctx, uint64_t bytes, void** out) {
  if (!ctx || !out) return shn_fail(SHN_ERR_ARG, "shn_debug_alloc: NULL argument");
  SHN_ENTER(ctx);
  HIP_TRY(shn_dev_malloc_on(out, bytes, ctx->stream));
  return SHN_OK;
}
extern "C" void shn_debug_free(shn_ctx* ctx, void* p) { if (ctx) shn_use_stream(ctx->stream); shn_dev_free(p); }
extern "C" int shn_debug_fill(shn_ctx* ctx, void* p, uint64_t n_words, uint32_t value, uint32_t spin) {
  if (!ctx || !p) return shn_fail(SHN_ERR_ARG, "shn_debug_fill: NULL argument");
  SHN_ENTER(ctx);
  if (n_words) hipLaunchKernelGGL(shn_debug_fill_kernel, dim3((uint32_t)cdiv(n_words, 256)), dim3(256), 0, ctx->stream, (uint32_t*)p, n_words, value, spin);
  HIP_TRY(hipGetLastError());
  return SHN_OK;
}
extern "C" int shn_debug_read(shn_ctx* ctx, const void* p, uint64_t n_words, uint32_t* host_out) {
  if (!ctx || !p || !host_out) return shn_fail(SHN_ERR_ARG, "shn_debug_read: NULL argument");
  SHN_ENTER(ctx);
  HIP_TRY(hipMemcpyAsync(host_out, p, n_words * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SHN_OK;
}

TimingOff::TimingOff(shn_ctx* ctx) : c(ctx), old(ctx->timing) { c->timing = false; }
TimingOff::~TimingOff() { c->timing = old; }

extern "C" int shn_ctx_create(int device, void* stream, shn_ctx** out) {
  if (!out) return shn_fail(SHN_ERR_ARG, "shn_ctx_create: out is NULL");
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return shn_fail(SHN_ERR_ARG, "shn_ctx_create: no such device");
  HIP_TRY(hipSetDevice(device));
  shn_ctx* c = new shn_ctx();
  c->device = device;
  c->stream = (hipStream_t)stream;
  c->timing = true;
  c->count_direct_log2 = 0;
  c->sk_pool_ratio = 0;
  c->owns_stream = false;
  for (int i = 0; i < T_N; i++) { c->ms[i] = 0; c->regions[i] = 0; c->abytes[i] = 0; }
  const char* rule = getenv("SHN_LP_RULE");
  c->lp_rule = (rule && !strcmp(rule, "vertex")) ? SHN_LP_RULE_VERTEX : SHN_LP_RULE_CENTER;
  for (auto& v : c->lp_stats) v = 0;
  c->wsset = shn_default_wsset();
  *out = c;
  return SHN_OK;
}
// A workspace set of its own for the context's top-level stages (count, extension, contig stage, probe table, routing, unitigs):
// until this call they use the process's set, which serves ONE pipeline at a time.  For a second pipeline in the process whose
// stages run beside the first one's (bench.py: the back half of two batches in flight, cut behind the extension).
extern "C" int shn_ctx_own_workspaces(shn_ctx* c) {
  if (!c) return shn_fail(SHN_ERR_ARG, "shn_ctx_own_workspaces: ctx is NULL");
  if (c->owns_wsset) return SHN_OK;
  if (c->parent) return shn_fail(SHN_ERR_ARG, "shn_ctx_own_workspaces: a forked context uses its parent's set");
  c->wsset = shn_wsset_create();
  c->owns_wsset = true;
  return SHN_OK;
}

// the forks alive (shn_ctx_fork): a context's timers are read together with those of its forks -- the seed scans of the graph
// threads and the LP batches of the sparse flow run on forks
static std::mutex g_forks_mu;
static std::vector<shn_ctx*> g_forks;
static int drain_own(shn_ctx* c) {
  HIP_TRY(hipStreamSynchronize(c->stream));
  std::lock_guard<std::mutex> lk(c->tmu);
  for (int i = 0; i < T_N; i++) {
    for (auto& p : c->pending[i]) {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
      c->ms[i] += ms;
      c->regions[i]++;
      hipEventDestroy(p.first);
      hipEventDestroy(p.second);
    }
    c->pending[i].clear();
  }
  return SHN_OK;
}
static void fold_into_parent(shn_ctx* f) {                      // (g_forks_mu held) what the fork has measured so far goes to its parent
  shn_ctx* p = f->parent;
  if (!p) return;
  std::lock_guard<std::mutex> lf(f->tmu);
  std::lock_guard<std::mutex> lp(p->tmu);
  for (int i = 0; i < T_N; i++) {
    p->ms[i] += f->ms[i]; p->regions[i] += f->regions[i]; f->ms[i] = 0; f->regions[i] = 0;
    p->abytes[i] += __atomic_exchange_n(&f->abytes[i], 0, __ATOMIC_RELAXED);
  }
}
void shn_lp_census_add(shn_ctx* c, const uint64_t* v8) {
  std::lock_guard<std::mutex> lk(g_forks_mu);
  shn_ctx* to = c->parent ? c->parent : c;
  for (int i = 0; i < 7; i++) if (v8[i]) __atomic_fetch_add(&to->lp_stats[i], v8[i], __ATOMIC_RELAXED);
}
extern "C" void shn_ctx_destroy(shn_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  {
    std::lock_guard<std::mutex> lk(g_forks_mu);
    if (c->parent) { (void)drain_own(c); fold_into_parent(c); }
    for (size_t i = 0; i < g_forks.size();) { if (g_forks[i] == c) g_forks.erase(g_forks.begin() + (ptrdiff_t)i); else { if (g_forks[i]->parent == c) g_forks[i]->parent = nullptr; i++; } }
  }
  for (int i = 0; i < T_N; i++)
    for (auto& p : c->pending[i]) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
  if (c->owns_stream && c->stream) { shn_stream_retired(c->stream); hipStreamDestroy(c->stream); }
  for (auto& w : c->cws) if (w.p) { hipFree(w.p); w.p = nullptr; w.cap = 0; }
  for (auto& h : c->hpin) h.release();
  if (c->owns_wsset) {
    // (its forks may outlive it -- kept graph threads: they go back to the process's set)
    { std::lock_guard<std::mutex> lk(g_forks_mu); for (shn_ctx* f : g_forks) if (f->wsset == c->wsset) f->wsset = shn_default_wsset(); }
    shn_wsset_destroy(c->wsset); c->wsset = nullptr; c->owns_wsset = false;
  }
  if (!c->owns_stream) shn_dev_trim();         // (a forked context goes with its host thread, in the middle of a run)
  delete c;
}

// A second context on the same device with a stream of its own, for a host thread that works beside the owner of `parent`
// (the graph threads: their seed scans overlap on the GPU instead of taking turns).  Its timers and LP census are read with the parent's.
extern "C" int shn_ctx_fork(const shn_ctx* parent, shn_ctx** out) {
  if (!parent || !out) return shn_fail(SHN_ERR_ARG, "shn_ctx_fork: NULL argument");
  HIP_TRY(hipSetDevice(parent->device));
  hipStream_t st = nullptr;
  // A fork's kernels are the short, latency-bound passes of one partition (seed scans, distinct reads, path searches, LP batches)
  // whose host thread waits for each: on a stream of the highest priority their workgroups are placed ahead of those of a long
  // kernel queued on the owner's stream -- which matters when another batch's counting / extension runs beside the graph stage
  // (bench.py's two batches in flight) and costs nothing otherwise.  SHN_FORK_PRIORITY=0: default priority, as until round 6.
  static const bool fork_high = !(getenv("SHN_FORK_PRIORITY") && getenv("SHN_FORK_PRIORITY")[0] == '0');
  int pr_least = 0, pr_greatest = 0;
  if (fork_high && hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest) == hipSuccess && pr_greatest != pr_least)
    HIP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, pr_greatest));
  else
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  shn_ctx* c = new shn_ctx();
  c->device = parent->device; c->stream = st; c->timing = parent->timing; c->count_direct_log2 = 0; c->sk_pool_ratio = 0; c->owns_stream = true;
  c->parent = const_cast<shn_ctx*>(parent);
  c->wsset = parent->wsset;
  shn_fork_stream_added(st);
  { std::lock_guard<std::mutex> lk(g_forks_mu); g_forks.push_back(c); }
  for (int i = 0; i < T_N; i++) { c->ms[i] = 0; c->regions[i] = 0; c->abytes[i] = 0; }
  c->lp_rule = parent->lp_rule;
  for (auto& v : c->lp_stats) v = 0;
  *out = c;
  return SHN_OK;
}

// A context of its own (same device, own stream) for the calling host thread: the partitions of the graph stage run on host
// threads at once, and so do their sparse-flow calls; with a stream each, their GPU sections overlap instead of taking turns.
// The fork lives as long as the thread (or until the thread asks for a fork of another parent).  NULL parent -> NULL.
// SHN_GRAPH_FORK=0 (with SHN_GRAPH_THREADS=1): everything on the caller's context and stream -- rocprofv3's kernel trace aborts
// (stream_stack.cpp) when threads it has not seen create streams; tools/profile_r0x.sh profile that way.
struct ShnThreadCtx {
  shn_ctx* c = nullptr; const shn_ctx* parent = nullptr;
  ~ShnThreadCtx() { if (c) shn_ctx_destroy(c); }
};
static thread_local ShnThreadCtx t_thread_ctx;
static shn_ctx* thread_ctx_impl(shn_ctx* p) {
  if (!p) return nullptr;
  static const bool no_fork = getenv("SHN_GRAPH_FORK") && getenv("SHN_GRAPH_FORK")[0] == '0';
  if (no_fork) return p;
  ShnThreadCtx& t = t_thread_ctx;
  if (t.c && t.parent == p && t.c->parent == p) { t.c->lp_rule = p->lp_rule; t.c->timing = p->timing; return t.c; }      // (rule / timing may have been set on the parent since the fork; a parent that went away orphaned the fork: a new one)
  if (t.c) { shn_ctx_destroy(t.c); t.c = nullptr; }
  if (shn_ctx_fork(p, &t.c)) { t.c = nullptr; return p; }
  t.parent = p;
  return t.c;
}
shn_ctx* shn_thread_ctx(shn_ctx* p) {
  shn_ctx* c = thread_ctx_impl(p);
  if (c) shn_use_stream(c->stream);            // (the calling thread works on this context's stream from here on)
  return c;
}

extern "C" int shn_ctx_sync(shn_ctx* c) {
  if (!c) return shn_fail(SHN_ERR_ARG, "ctx is NULL");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return SHN_OK;
}

static int drain_timers(shn_ctx* c) {
  int rc = drain_own(c);
  if (rc || c->parent) return rc;
  std::lock_guard<std::mutex> lk(g_forks_mu);
  for (shn_ctx* f : g_forks) if (f->parent == c) { if ((rc = drain_own(f))) return rc; fold_into_parent(f); }
  return SHN_OK;
}

extern "C" int shn_timer_reset(shn_ctx* c) {
  if (!c) return shn_fail(SHN_ERR_ARG, "ctx is NULL");
  int rc = drain_timers(c);
  if (rc) return rc;
  for (int i = 0; i < T_N; i++) { c->ms[i] = 0; c->regions[i] = 0; c->abytes[i] = 0; }
  return SHN_OK;
}

// the algorithmic bytes the launch sites of a slot have declared since the last reset (0 for the slots that declare none)
extern "C" int shn_timer_bytes(shn_ctx* c, int slot, uint64_t* bytes) {
  if (!c || slot < 0 || slot >= T_N || !bytes) return shn_fail(SHN_ERR_ARG, "shn_timer_bytes: bad argument");
  int rc = drain_timers(c);
  if (rc) return rc;
  *bytes = c->abytes[slot];
  return SHN_OK;
}
extern "C" int shn_timer_ms(shn_ctx* c, int slot, double* ms, uint64_t* n_regions) {
  if (!c || slot < 0 || slot >= T_N) return shn_fail(SHN_ERR_ARG, "shn_timer_ms: bad argument");
  int rc = drain_timers(c);
  if (rc) return rc;
  if (ms) *ms = c->ms[slot];
  if (n_regions) *n_regions = c->regions[slot];
  return SHN_OK;
}

// ------------------------------------------------------------------------------------ packer
__device__ __forceinline__ uint32_t enc_base(uint8_t b, int enc) {
  if (enc == SHN_ENC_CODES) return b < 4 ? b : 4u;
  switch (b) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}

// One thread per 64-base group (two 2-bit words + one mask word) of one read.
__global__ void pack_kernel(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ boff,
                            const uint64_t* __restrict__ woff, uint64_t n_reads, uint32_t fixed_len, uint32_t wpr,
                            int enc, uint64_t* __restrict__ words, uint64_t* __restrict__ mask,
                            uint32_t* __restrict__ lens, uint64_t n_groups_fixed) {
  uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t r, g, wbase, bstart;
  uint32_t len;
  if (boff == nullptr) {
    uint32_t gpr = wpr / 2;
    if (gid >= n_groups_fixed) return;
    r = gid / gpr; g = gid % gpr;
    wbase = r * wpr; bstart = r * (uint64_t)fixed_len; len = fixed_len;
  } else {
    // ragged: gid enumerates (read, group) through woff (binary search over reads)
    uint64_t tw = woff[n_reads];
    if (gid * 2 >= tw) return;
    uint64_t lo = 0, hi = n_reads;            // largest r with woff[r] <= gid*2
    while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (woff[mid] <= gid * 2) lo = mid; else hi = mid; }
    r = lo; wbase = woff[r]; g = (gid * 2 - wbase) / 2;
    bstart = boff[r]; len = (uint32_t)(boff[r + 1] - boff[r]);
    if (g == 0) lens[r] = len;
  }
  uint64_t w0 = 0, w1 = 0, m = 0;
  uint32_t base0 = (uint32_t)g * 64;
  for (uint32_t j = 0; j < 64; j++) {
    uint32_t p = base0 + j;
    uint32_t c = 0;
    if (p < len) {
      c = enc_base(bytes[bstart + p], enc);
      if (c == 4) { m |= 1ULL << (63 - j); c = 0; }
    }
    if (j < 32) w0 |= (uint64_t)c << (62 - 2 * j);
    else w1 |= (uint64_t)c << (62 - 2 * (j - 32));
  }
  words[wbase + 2 * g] = w0;
  words[wbase + 2 * g + 1] = w1;
  mask[wbase / 2 + g] = m;
}

__global__ void bad_reads_kernel(const uint64_t* __restrict__ mask, const uint64_t* __restrict__ woff,
                                 uint64_t n_reads, uint32_t wpr, uint8_t* __restrict__ bad,
                                 unsigned long long* __restrict__ n_bad) {
  uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  uint64_t m0 = woff ? woff[r] / 2 : r * (wpr / 2);
  uint64_t m1 = woff ? woff[r + 1] / 2 : (r + 1) * (wpr / 2);
  uint64_t acc = 0;
  for (uint64_t i = m0; i < m1; i++) acc |= mask[i];
  bad[r] = acc ? 1 : 0;
  if (acc) atomicAdd(n_bad, 1ULL);
}

extern "C" int shn_reads_create(shn_ctx* ctx, const uint8_t* bytes, const uint64_t* offsets, uint64_t n_reads,
                                uint32_t fixed_len, int enc, shn_reads** out) {
  if (!ctx || !out || (!bytes && n_reads)) return shn_fail(SHN_ERR_ARG, "shn_reads_create: NULL argument");
  if (enc != SHN_ENC_ASCII && enc != SHN_ENC_CODES) return shn_fail(SHN_ERR_ARG, "shn_reads_create: bad encoding");
  if (!offsets && fixed_len == 0 && n_reads) return shn_fail(SHN_ERR_ARG, "shn_reads_create: fixed_len is 0");
  SHN_ENTER(ctx);
  shn_reads* r = new shn_reads();
  memset(r, 0, sizeof(*r));
  r->ctx = ctx;
  r->device = ctx->device;
  r->n_reads = n_reads;
  std::vector<uint64_t> woff;
  uint64_t total_bytes;
  if (offsets) {
    woff.resize(n_reads + 1);
    uint64_t w = 0;
    uint32_t mx = 0;
    for (uint64_t i = 0; i < n_reads; i++) {
      if (offsets[i + 1] < offsets[i]) { delete r; return shn_fail(SHN_ERR_ARG, "shn_reads_create: offsets not monotone"); }
      uint64_t len = offsets[i + 1] - offsets[i];
      if (len > 0x7fffffffULL) { delete r; return shn_fail(SHN_ERR_ARG, "shn_reads_create: read too long"); }
      woff[i] = w;
      w += 2 * cdiv(len ? len : 1, 64);
      if (len > mx) mx = (uint32_t)len;
    }
    woff[n_reads] = w;
    r->n_words = w;
    r->max_len = mx;
    r->fixed_len = 0;
    total_bytes = offsets[n_reads] - offsets[0];
    r->total_bases = total_bytes;
  } else {
    r->fixed_len = fixed_len;
    r->max_len = fixed_len;
    r->wpr = (uint32_t)(2 * cdiv(fixed_len, 64));
    r->n_words = n_reads * r->wpr;
    total_bytes = n_reads * (uint64_t)fixed_len;
    r->total_bases = total_bytes;
  }
  uint8_t* d_bytes = nullptr;
  uint64_t* d_boff = nullptr;
  hipStream_t s = ctx->stream; shn_use_stream(s);
  auto cleanup = [&]() { if (d_bytes) hipFree(d_bytes); if (d_boff) hipFree(d_boff); };
#define TRY2(e) do { hipError_t _e = (e); if (_e != hipSuccess) { cleanup(); shn_reads_destroy(r); \
      return shn_fail(SHN_ERR_HIP, std::string(#e) + ": " + hipGetErrorString(_e)); } } while (0)
  TRY2(shn_hip_malloc(&r->d_words, (r->n_words + 2) * 8));
  TRY2(shn_hip_malloc(&r->d_mask, (r->n_words / 2 + 2) * 8));
  TRY2(hipMemsetAsync(r->d_words, 0, (r->n_words + 2) * 8, s));
  TRY2(hipMemsetAsync(r->d_mask, 0, (r->n_words / 2 + 2) * 8, s));
  if (n_reads) {
    TRY2(shn_hip_malloc(&d_bytes, total_bytes ? total_bytes : 1));
    TRY2(hipMemcpyAsync(d_bytes, bytes + (offsets ? offsets[0] : 0), total_bytes, hipMemcpyHostToDevice, s));
    if (offsets) {
      std::vector<uint64_t> rel(n_reads + 1);
      for (uint64_t i = 0; i <= n_reads; i++) rel[i] = offsets[i] - offsets[0];
      TRY2(shn_hip_malloc(&d_boff, (n_reads + 1) * 8));
      TRY2(hipMemcpyAsync(d_boff, rel.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, s));
      TRY2(shn_hip_malloc(&r->d_woff, (n_reads + 1) * 8));
      TRY2(hipMemcpyAsync(r->d_woff, woff.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, s));
      TRY2(shn_hip_malloc(&r->d_len, n_reads * 4));
      TRY2(hipStreamSynchronize(s));   // rel / woff are stack-owned vectors
    }
    {
      TimerRegion t(ctx, T_PACK);
      uint64_t n_groups = r->n_words / 2;
      uint32_t blocks = (uint32_t)cdiv(n_groups, 256);
      hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, s, d_bytes, d_boff, r->d_woff, n_reads, fixed_len,
                         r->wpr, enc, r->d_words, r->d_mask, r->d_len, n_groups);
      unsigned long long* d_nbad = nullptr;
      TRY2(shn_hip_malloc(&d_nbad, 8));
      TRY2(hipMemsetAsync(d_nbad, 0, 8, s));
      TRY2(shn_hip_malloc(&r->d_bad, n_reads));
      hipLaunchKernelGGL(bad_reads_kernel, dim3((uint32_t)cdiv(n_reads, 256)), dim3(256), 0, s, r->d_mask, r->d_woff,
                         n_reads, r->wpr, r->d_bad, d_nbad);
      unsigned long long nb = 0;
      TRY2(hipMemcpyAsync(&nb, d_nbad, 8, hipMemcpyDeviceToHost, s));
      TRY2(hipStreamSynchronize(s));
      hipFree(d_nbad);
      r->n_invalid = nb;
    }
    TRY2(hipGetLastError());
  }
  cleanup();
#undef TRY2
  *out = r;
  return SHN_OK;
}

// pieces of shn_reads_create for callers that fill a fixed-length read set group by group (ingest.hip)
int shn_pack_fixed_codes(shn_ctx* ctx, const uint8_t* d_codes, uint64_t n, uint32_t L, uint32_t wpr, uint64_t* d_words, uint64_t* d_mask,
                         hipStream_t s) {
  if (!n) return SHN_OK;
  TimerRegion t(ctx, T_PACK, s);
  const uint64_t n_groups = n * (wpr / 2);
  hipLaunchKernelGGL(pack_kernel, dim3((uint32_t)cdiv(n_groups, 256)), dim3(256), 0, s, d_codes, (const uint64_t*)nullptr, (const uint64_t*)nullptr, n, L,
                     wpr, SHN_ENC_CODES, d_words, d_mask, (uint32_t*)nullptr, n_groups);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SHN_OK : shn_fail(SHN_ERR_HIP, std::string("pack_kernel: ") + hipGetErrorString(e));
}
int shn_reads_finish_fixed(shn_ctx* ctx, shn_reads* r) {
  hipStream_t s = ctx->stream; shn_use_stream(s);
  unsigned long long* d_nbad = nullptr;
  HIP_TRY(shn_hip_malloc(&d_nbad, 8));
  hipError_t e = hipMemsetAsync(d_nbad, 0, 8, s);
  if (e == hipSuccess) e = shn_hip_malloc(&r->d_bad, r->n_reads ? r->n_reads : 1);
  unsigned long long nb = 0;
  if (e == hipSuccess && r->n_reads) {
    hipLaunchKernelGGL(bad_reads_kernel, dim3((uint32_t)cdiv(r->n_reads, 256)), dim3(256), 0, s, r->d_mask, (const uint64_t*)nullptr, r->n_reads, r->wpr,
                       r->d_bad, d_nbad);
    e = hipMemcpyAsync(&nb, d_nbad, 8, hipMemcpyDeviceToHost, s);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(d_nbad);
  if (e != hipSuccess) return shn_fail(SHN_ERR_HIP, std::string("shn_reads_finish_fixed: ") + hipGetErrorString(e));
  r->n_invalid = nb;
  return SHN_OK;
}

// Rows of resident fixed-length read sets as a new read set, without a trip through the host: read i = row rows[i] of set a
// (flags[i] bit 0 clear) or b (set), reverse-complemented if bit 1 is set.  The graph stage builds its distinct-read set this
// way (the reads of a partition are rows of the input the routing kernel selected; uploading their text again was the
// serialized part of the stage).  The selected rows must hold ACGT only (routed reads do, kmers_for_component.py:336,376).
__global__ void reads_gather_kernel(const uint64_t* __restrict__ wa, const uint64_t* __restrict__ wb, uint32_t wpr, uint32_t L,
                                    const uint32_t* __restrict__ rows, const uint8_t* __restrict__ flags, uint64_t n, uint64_t* __restrict__ out) {
  for (uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < n * wpr; gid += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t i = gid / wpr;
    const uint32_t w = (uint32_t)(gid - i * wpr);
    const uint8_t f = flags[i];
    const uint64_t* src = ((f & 1) ? wb : wa) + (uint64_t)rows[i] * wpr;
    uint64_t v = 0;
    if (!(f & 2)) v = src[w];
    else if (32 * w < L) {
      // bases [32 w, 32 w + nb) of the reverse complement = the complement of bases [L - 32 w - nb, L - 32 w) read backwards
      const uint32_t nb = min(32u, L - 32 * w);
      v = shn_revcomp(shn_extract(src, L - 32 * w - nb, (int)nb), (int)nb) << (64 - 2 * nb);
    }
    out[gid] = v;
  }
}
// rows / flags on the host (uploaded) or already on the device (d_rows / d_flags: graph_dev.h, the result of the duplicate search)
static int reads_gather_impl(shn_ctx* ctx, const shn_reads* a, const shn_reads* b, const uint32_t* rows, const uint8_t* flags, const uint32_t* dev_rows,
                             const uint8_t* dev_flags, uint64_t n, shn_reads** out) {
  if (!a->fixed_len || (b && (b->fixed_len != a->fixed_len || b->wpr != a->wpr))) return shn_fail(SHN_ERR_ARG, "shn_reads_gather: fixed-length read sets of one length only");
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  shn_reads* r = new shn_reads();
  memset(r, 0, sizeof(*r));
  r->ctx = ctx; r->device = ctx->device; r->n_reads = n; r->fixed_len = a->fixed_len; r->max_len = a->fixed_len; r->wpr = a->wpr;
  r->n_words = n * a->wpr; r->total_bases = n * (uint64_t)a->fixed_len; r->cached = true;
  uint32_t* d_rows = nullptr; uint8_t* d_flags = nullptr;
  auto fail = [&](hipError_t e) { shn_dev_free(d_rows); shn_dev_free(d_flags); shn_reads_destroy(r); return shn_fail(SHN_ERR_HIP, std::string("shn_reads_gather: ") + hipGetErrorString(e)); };
  hipError_t e;
  if ((e = shn_dev_malloc(&r->d_words, (r->n_words + 2) * 8)) != hipSuccess) return fail(e);
  if ((e = shn_dev_malloc(&r->d_mask, (r->n_words / 2 + 2) * 8)) != hipSuccess) return fail(e);
  if (!dev_rows) {
    if ((e = shn_dev_malloc(&d_rows, (n + 1) * 4)) != hipSuccess) return fail(e);
    if ((e = shn_dev_malloc(&d_flags, n + 1)) != hipSuccess) return fail(e);
  }
  if ((e = hipMemsetAsync(r->d_mask, 0, (r->n_words / 2 + 2) * 8, s)) != hipSuccess) return fail(e);
  if ((e = hipMemsetAsync(r->d_words + r->n_words, 0, 16, s)) != hipSuccess) return fail(e);
  if (n) {
    if (!dev_rows) {
      if ((e = hipMemcpyAsync(d_rows, rows, n * 4, hipMemcpyHostToDevice, s)) != hipSuccess) return fail(e);
      if ((e = hipMemcpyAsync(d_flags, flags, n, hipMemcpyHostToDevice, s)) != hipSuccess) return fail(e);
    }
    hipLaunchKernelGGL(reads_gather_kernel, dim3((uint32_t)std::min<uint64_t>(cdiv(r->n_words, 256), 1u << 20)), dim3(256), 0, s, a->d_words,
                       b ? b->d_words : a->d_words, a->wpr, a->fixed_len, dev_rows ? dev_rows : d_rows, dev_flags ? dev_flags : d_flags, n, r->d_words);
  }
  if ((e = hipStreamSynchronize(s)) != hipSuccess) return fail(e);
  shn_dev_free(d_rows); shn_dev_free(d_flags);
  HIP_TRY(hipGetLastError());
  *out = r;
  return SHN_OK;
}
extern "C" int shn_reads_gather(shn_ctx* ctx, const shn_reads* a, const shn_reads* b, const uint32_t* rows, const uint8_t* flags, uint64_t n,
                                shn_reads** out) {
  if (!ctx || !a || !out || (n && (!rows || !flags))) return shn_fail(SHN_ERR_ARG, "shn_reads_gather: NULL argument");
  return reads_gather_impl(ctx, a, b, rows, flags, nullptr, nullptr, n, out);
}
int shn_reads_gather_dev(shn_ctx* ctx, const shn_reads* a, const shn_reads* b, const uint32_t* d_rows, const uint8_t* d_flags, uint64_t n, shn_reads** out) {
  if (!ctx || !a || !out || (n && (!d_rows || !d_flags))) return shn_fail(SHN_ERR_ARG, "shn_reads_gather_dev: NULL argument");
  return reads_gather_impl(ctx, a, b, nullptr, nullptr, d_rows, d_flags, n, out);
}

extern "C" void shn_reads_destroy(shn_reads* r) {
  if (!r) return;
  hipSetDevice(r->device);
  if (r->cached) { shn_dev_free(r->d_words); shn_dev_free(r->d_mask); delete r; return; }
  if (r->d_words) hipFree(r->d_words);
  if (r->d_mask) hipFree(r->d_mask);
  if (r->d_woff) hipFree(r->d_woff);
  if (r->d_len) hipFree(r->d_len);
  if (r->d_bad) hipFree(r->d_bad);
  delete r;
}

extern "C" uint64_t shn_reads_count(const shn_reads* r) { return r ? r->n_reads : 0; }
extern "C" uint64_t shn_reads_total_bases(const shn_reads* r) { return r ? r->total_bases : 0; }
extern "C" uint32_t shn_reads_max_len(const shn_reads* r) { return r ? r->max_len : 0; }
extern "C" uint64_t shn_reads_n_invalid(const shn_reads* r) { return r ? r->n_invalid : 0; }


// ---- host threads this process may keep busy: the smallest of the hardware threads, the affinity mask and the cgroup CPU
// quota (cpu.max: a container that sees 256 hardware threads may be allowed 16 CPUs' worth of time per period -- running more
// threads than that does not add throughput, it gets every thread of the process throttled until the next period, the
// serial chains of the graph stage included), divided by the ranks of the node.  SHN_HOST_CPUS overrides.  Every thread count of
// the host stages derives from it.
// The host stages (graph threads, sparse flow, merge, the numpy buffers between them) allocate and free hundreds of blocks of
// 0.1-30 MB per step.  glibc serves those by mmap and gives them back on free: every step pays the page faults of fresh zero
// pages again (~90 ms of a 3.2 s step at BASELINE configs[2]).  OPT-IN (an embedding application's allocator is not this library's
// to change): with SHN_MALLOC_TUNE=1 in the environment, loading the library raises the mmap threshold to its maximum and keeps
// freed memory in the heap (MALLOC_*_ environment settings win as usual); bench.py and shannon.py ask for it, shn_malloc_tune_now()
// is the call for a host program that wants it.
#include <malloc.h>
#include <climits>
static void shn_malloc_tune_apply();
__attribute__((constructor)) static void shn_malloc_tune() {
  const char* v = getenv("SHN_MALLOC_TUNE");
  if (!v || v[0] != '1') return;
  shn_malloc_tune_apply();
}
extern "C" void shn_malloc_tune_now(void) { shn_malloc_tune_apply(); }
static void shn_malloc_tune_apply() {
  if (!getenv("MALLOC_MMAP_THRESHOLD_")) mallopt(M_MMAP_THRESHOLD, 32 << 20);
  if (!getenv("MALLOC_TRIM_THRESHOLD_")) mallopt(M_TRIM_THRESHOLD, -1);          // (size_t) -1: the heap is never trimmed
  if (!getenv("MALLOC_TOP_PAD_")) mallopt(M_TOP_PAD, 256 << 20);
}

extern "C" int shn_host_cpus(void) {
  static int cached = 0;
  if (cached) return cached;
  int n = (int)std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min(n, c); }
  auto quota = [&](const char* path, bool v2) {
    FILE* f = fopen(path, "r");
    if (!f) return;
    char a[64] = {0}, b[64] = {0};
    if (v2) {
      if (fscanf(f, "%63s %63s", a, b) == 2 && strcmp(a, "max") != 0) {
        const double q = atof(a), per = atof(b);
        if (q > 0 && per > 0) n = std::min(n, std::max(1, (int)(q / per + 0.999)));
      }
    } else if (fscanf(f, "%63s", a) == 1) {
      const double q = atof(a);
      FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
      double per = 100000;
      if (g) { if (fscanf(g, "%63s", b) == 1 && atof(b) > 0) per = atof(b); fclose(g); }
      if (q > 0) n = std::min(n, std::max(1, (int)(q / per + 0.999)));
    }
    fclose(f);
  };
  quota("/sys/fs/cgroup/cpu.max", true);
  quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", false);
  // one process per GPU: the ranks of a node share the allowance (LOCAL_WORLD_SIZE is torchrun's count of them; SHN_LOCAL_RANKS
  // for other launchers) -- eight ranks each sizing its pools for the whole quota is the oversubscription described above
  for (const char* name : {"SHN_LOCAL_RANKS", "LOCAL_WORLD_SIZE"})
    if (const char* e = getenv(name)) { const int v = atoi(e); if (v > 1) { n = std::max(1, n / v); break; } }
  if (const char* e = getenv("SHN_HOST_CPUS")) { const int v = atoi(e); if (v > 0) n = v; }
  cached = std::max(1, n);
  return cached;
}

// ---- host utility: dst[i] = src row idx[i] (rows of row_bytes bytes), split over host threads.  The read rows that
// travel to a partition's owner and their re-ordering there are 100+ MB gathers; numpy does them on one thread.
extern "C" int shn_gather_rows(const uint8_t* src, uint64_t n_src_rows, uint64_t row_bytes, const int64_t* idx, uint64_t n, uint8_t* dst,
                               int threads) {
  if ((n && (!src || !idx || !dst)) || !row_bytes) return shn_fail(SHN_ERR_ARG, "shn_gather_rows: NULL argument");
  for (uint64_t i = 0; i < n; i++)
    if (idx[i] < 0 || (uint64_t)idx[i] >= n_src_rows) return shn_fail(SHN_ERR_ARG, "shn_gather_rows: row index out of range");
  const uint64_t per_thread_min = (1ULL << 20) / row_bytes + 1;          // at least ~1 MB per thread
  int T = (int)std::min<uint64_t>((uint64_t)std::max(1, threads), n / per_thread_min + 1);
  auto work = [&](uint64_t a, uint64_t b) {
    for (uint64_t i = a; i < b; i++) memcpy(dst + i * row_bytes, src + (uint64_t)idx[i] * row_bytes, row_bytes);
  };
  if (T <= 1) { work(0, n); return SHN_OK; }
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++) th.emplace_back(work, n * t / T, n * (t + 1) / T);
  for (auto& x : th) x.join();
  return SHN_OK;
}

// ---- host utility: segment i of dst = segment order[i] of src (segments given by offset arrays), split over host threads.  The
// candidate contigs of all ranks are merged into the global seed order this way (300 MB at BASELINE configs[2]).
extern "C" int shn_gather_segments(const uint8_t* src, const uint64_t* src_off, uint64_t n_src, const int64_t* order, uint64_t n, uint8_t* dst,
                                   const uint64_t* dst_off, int threads) {
  if (n && (!src || !src_off || !order || !dst || !dst_off)) return shn_fail(SHN_ERR_ARG, "shn_gather_segments: NULL argument");
  for (uint64_t i = 0; i < n; i++) {
    if (order[i] < 0 || (uint64_t)order[i] >= n_src) return shn_fail(SHN_ERR_ARG, "shn_gather_segments: segment index out of range");
    if (dst_off[i + 1] - dst_off[i] != src_off[order[i] + 1] - src_off[order[i]]) return shn_fail(SHN_ERR_ARG, "shn_gather_segments: segment lengths differ");
  }
  const uint64_t total = n ? dst_off[n] - dst_off[0] : 0;
  int T = (int)std::min<uint64_t>((uint64_t)std::max(1, threads), total / (1ULL << 20) + 1);
  auto work = [&](uint64_t a, uint64_t b) {
    for (uint64_t i = a; i < b; i++) memcpy(dst + dst_off[i], src + src_off[order[i]], dst_off[i + 1] - dst_off[i]);
  };
  if (T <= 1) { work(0, n); return SHN_OK; }
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++) th.emplace_back(work, n * t / T, n * (t + 1) / T);
  for (auto& x : th) x.join();
  return SHN_OK;
}

// ---- host utility: a two-line FASTA from byte segments: record i = ">" prefix i "\n" + segment order[i] of src + "\n" (the
// reconstructed_single_contigs.fasta of extension_correction.py:506-513 straight from the candidate buffer of the contig stage).
// dst == NULL: only *dst_len (the size) is set.
extern "C" int shn_fasta_records(const uint8_t* src, const uint64_t* src_off, uint64_t n_src, const int64_t* order, uint64_t n, const char* prefix,
                                 uint8_t* dst, uint64_t dst_cap, uint64_t* dst_len) {
  if (!dst_len || !prefix || (n && (!src || !src_off || !order))) return shn_fail(SHN_ERR_ARG, "shn_fasta_records: NULL argument");
  const size_t lp = strlen(prefix);
  uint64_t need = 0;
  for (uint64_t i = 0; i < n; i++) {
    if (order[i] < 0 || (uint64_t)order[i] >= n_src) return shn_fail(SHN_ERR_ARG, "shn_fasta_records: segment index out of range");
    uint64_t digits = 1; for (uint64_t v = i; v >= 10; v /= 10) digits++;
    need += 1 + lp + digits + 1 + (src_off[order[i] + 1] - src_off[order[i]]) + 1;
  }
  *dst_len = need;
  if (!dst) return SHN_OK;
  if (dst_cap < need) return shn_fail(SHN_ERR_ARG, "shn_fasta_records: destination too small");
  uint8_t* w = dst;
  char num[24];
  for (uint64_t i = 0; i < n; i++) {
    *w++ = '>';
    memcpy(w, prefix, lp); w += lp;
    const int nd = snprintf(num, sizeof num, "%llu", (unsigned long long)i);
    memcpy(w, num, (size_t)nd); w += nd;
    *w++ = '\n';
    const uint64_t len = src_off[order[i] + 1] - src_off[order[i]];
    memcpy(w, src + src_off[order[i]], len); w += len;
    *w++ = '\n';
  }
  return SHN_OK;
}

// ---- host utility: every k-window of every string (ASCII ACGT, strings given by offsets into one text), in order: as
// packed 2-bit keys (keys_out, k <= 32) and/or as fixed-width byte rows (rows_out, k bytes per window).  The partition
// stage needs both for every partition contig (k1mers2component, kmers_for_component.py:244-305; the k1-mer files,
// :452-477).  Returns SHN_ERR_ARG on a base outside ACGT (the caller then takes its general path).
extern "C" int shn_string_windows(const uint8_t* text, const uint64_t* off, uint64_t n_strings, int k, uint64_t* keys_out, uint8_t* rows_out) {
  if ((n_strings && (!text || !off)) || k < 1 || (keys_out && k > 32)) return shn_fail(SHN_ERR_ARG, "shn_string_windows: bad argument");
  const uint64_t mask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
  uint64_t w = 0;
  for (uint64_t i = 0; i < n_strings; i++) {
    const uint8_t* s = text + off[i];
    const uint64_t L = off[i + 1] - off[i];
    uint64_t key = 0;
    for (uint64_t p = 0; p < L; p++) {
      int c;
      switch (s[p]) { case 'A': c = 0; break; case 'C': c = 1; break; case 'G': c = 2; break; case 'T': c = 3; break;
                      default: return shn_fail(SHN_ERR_ARG, "shn_string_windows: base outside ACGT"); }
      key = ((key << 2) | (uint64_t)c) & mask;
      if (p + 1 >= (uint64_t)k) {
        if (keys_out) keys_out[w] = key;
        if (rows_out) memcpy(rows_out + w * (uint64_t)k, s + p + 1 - k, (size_t)k);
        w++;
      }
    }
  }
  return SHN_OK;
}
