// Batched sparse-flow node decomposition LPs on gfx950 (row a28): replaces the <=100
// cvxopt.solvers.lp calls per decomposed node of path_decompose (path_decompose_sparse.py:100-117).
//
// Each trial is a tiny transportation problem  min c.x, row sums a, column sums b, x >= 0  with
// cost numerators c = |Irwin-Hall(12) - 6| * 2^32 on unsupported cells, 0 on supported cells
// (counter-based splitmix64 stream -- integer arithmetic, identical on CPU and GPU).  cvxopt solves
// it with an interior-point method, whose limit on a degenerate optimal face (the normal case: supported
// cells cost nothing) is the face's analytic centre.  That limit is computed here in two kernels:
//   lp_trials_kernel   an exact vertex by successive shortest paths with Jacobi Bellman-Ford rounds and
//                      lowest-index tie-breaks (oracle/lp.py:transport_vertex) -- it fixes the flows on the
//                      unsupported cells, which are the same all over the optimal face;
//   lp_center_kernel   the supported cells moved to the analytic centre of the face: classes of the residual
//                      digraph (which cells can be positive), then an infeasible-start Newton iteration per class
//                      with more cells than a tree (oracle/lp.py:face_center / center_component, repeated
//                      operation by operation: IEEE doubles, -ffp-contract=off, so the two agree bit for bit).
// The trials of one node run in the lanes of a wavefront (lane = trial): same m, n, a, b, support mask,
// different costs.  All per-trial state lives in a [element][trial] workspace so lane accesses coalesce.
// SHN_LP_RULE=vertex (or shn_lp_set_rule) keeps the vertex as the answer (the rule of rounds 1-2).
// No MFMA: no dense contraction anywhere.
#include "common.h"
#include <cstring>
#include <atomic>

#define LBLK 64
#define LP_INF (1LL << 62)
#define GOLD 0x9E3779B97F4A7C15ULL

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += GOLD;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

__host__ __device__ __forceinline__ int64_t lp_cell_cost(uint64_t seed, uint64_t pid, uint64_t trial, uint64_t cell) {
  uint64_t h = splitmix64(seed ^ ((pid + 1) * 0xD1B54A32D192ED03ULL));
  h = splitmix64(h ^ ((trial + 1) * 0xAEF17502108EF2D9ULL));
  h = splitmix64(h ^ ((cell + 1) * 0x8CB92BA72F3D8DD7ULL));
  uint64_t s = 0, st = h;
  for (int i = 0; i < 6; i++) { st = splitmix64(st); s += (st & 0xFFFFFFFFULL) + (st >> 32); }
  int64_t v = (int64_t)s - (int64_t)(6ULL << 32);
  return v < 0 ? -v : v;
}

struct LpProblem {
  uint32_t m, n, trials, pad;
  uint64_t pid;
  uint64_t in_off;     // doubles: a_s[m], b_s[n] ; then mask bytes at mask_off
  uint64_t mask_off;   // bytes: p[j*m+i] (1 = unsupported)
  uint64_t ws_off;     // 8-byte words of workspace for this problem
  uint64_t out_off;    // doubles: [cell k = j*m+i][trial]
  uint64_t ws2_off;    // 8-byte words of the centre kernel's workspace for this problem
  uint64_t stat_off;   // first trial of this problem in the per-trial statistics
};

// workspace layout per problem (T = trials), all [elem][trial]:
//   C[mn] int64 | X[mn] double | ra[m] | rb[n] | ds[m] int64 | dt[n] int64 | ps[m] int64 | pt[n] int64
// LDS = true: the trial's state (2 mn + 3 (m + n) words per lane) lives in LDS, [element][lane] -- the successive-shortest-path loops
// are chains of dependent reads and writes of that state, and through HBM a launch of tiny problems took 0.65 ms (71 ms per step of
// BASELINE configs[2] over its 109 dependent batches).  The host puts a problem on this kernel when 64 lanes of its state fit
// LP_LDS_WORDS words; the vertex goes back to the HBM workspace at the end (lp_center_kernel starts from it).  Same operations in
// the same order: the same bits.
#define LP_LDS_WORDS 20000         // 156 KB of the 160 KB of LDS a gfx950 CU has: problems with 2 mn + 3 (m + n) <= 312 words per trial (m = n = 11); 64 KB (m = n = 6) left nine launches per step of BASELINE configs[2] on the HBM form at 2.8 ms each
template <bool LDS>
__global__ __launch_bounds__(LBLK) void lp_trials_kernel(const LpProblem* __restrict__ probs, const uint32_t* __restrict__ block_prob,
                                                         const uint32_t* __restrict__ block_first, const double* __restrict__ in,
                                                         const uint8_t* __restrict__ masks, uint64_t seed,
                                                         uint64_t* __restrict__ ws, double* __restrict__ out) {
  extern __shared__ uint64_t lp_lds[];
  const LpProblem P = probs[block_prob[blockIdx.x]];
  const uint32_t t = block_first[blockIdx.x] + threadIdx.x;
  if (t >= P.trials) return;
  const uint32_t m = P.m, n = P.n, T = P.trials, mn = m * n;
  const double* a_s = in + P.in_off;
  const double* b_s = a_s + m;
  const uint8_t* pm = masks + P.mask_off;
  const uint64_t TS = LDS ? (uint64_t)LBLK : (uint64_t)T;              // stride between the elements of an array
  const uint64_t tl = LDS ? (uint64_t)threadIdx.x : (uint64_t)t;       // this trial's column
  int64_t* C = LDS ? (int64_t*)lp_lds : (int64_t*)(ws + P.ws_off);
  double* X = (double*)(C + (uint64_t)mn * TS);
  double* ra = X + (uint64_t)mn * TS;
  double* rb = ra + (uint64_t)m * TS;
  int64_t* ds = (int64_t*)(rb + (uint64_t)n * TS);
  int64_t* dt = ds + (uint64_t)m * TS;
  int64_t* ps = dt + (uint64_t)n * TS;
  int64_t* pt = ps + (uint64_t)m * TS;
#define AT(arr, e) arr[(uint64_t)(e) * TS + tl]
#define CX(i, j) ((uint64_t)((j) * m + (i)))
  for (uint32_t k = 0; k < mn; k++) {
    AT(C, k) = pm[k] ? lp_cell_cost(seed, P.pid, t, k) : 0;
    AT(X, k) = 0.0;
  }
  for (uint32_t i = 0; i < m; i++) AT(ra, i) = a_s[i];
  for (uint32_t j = 0; j < n; j++) AT(rb, j) = b_s[j];
  const uint32_t max_it = 4 * (m + n) + mn + 16;
  for (uint32_t it = 0; it < max_it; it++) {
    bool anyS = false, anyT = false;
    for (uint32_t i = 0; i < m; i++) { bool s = AT(ra, i) > 0; anyS |= s; AT(ds, i) = s ? 0 : LP_INF; AT(ps, i) = -1; }
    for (uint32_t j = 0; j < n; j++) { anyT |= AT(rb, j) > 0; AT(dt, j) = LP_INF; AT(pt, j) = -1; }
    if (!anyS || !anyT) break;
    for (uint32_t r = 0; r < m + n; r++) {
      bool changed = false;
      for (uint32_t j = 0; j < n; j++) {
        int64_t best = LP_INF; int64_t bi = -1;
        for (uint32_t i = 0; i < m; i++) {
          int64_t d = AT(ds, i);
          if (d < LP_INF) { int64_t v = d + AT(C, CX(i, j)); if (v < best) { best = v; bi = i; } }
        }
        if (bi >= 0 && best < AT(dt, j)) { AT(dt, j) = best; AT(pt, j) = bi; changed = true; }
      }
      for (uint32_t i = 0; i < m; i++) {
        int64_t best = LP_INF; int64_t bj = -1;
        for (uint32_t j = 0; j < n; j++) {
          if (AT(X, CX(i, j)) > 0) {
            int64_t d = AT(dt, j);
            if (d < LP_INF) { int64_t v = d - AT(C, CX(i, j)); if (v < best) { best = v; bj = j; } }
          }
        }
        if (bj >= 0 && best < AT(ds, i)) { AT(ds, i) = best; AT(ps, i) = bj; changed = true; }
      }
      if (!changed) break;
    }
    int64_t tt = -1, bd = LP_INF;
    for (uint32_t j = 0; j < n; j++) if (AT(rb, j) > 0 && AT(dt, j) < bd) { bd = AT(dt, j); tt = j; }
    if (tt < 0) break;
    // pass 1: delta along the predecessor chain
    double delta = AT(rb, tt);
    int64_t j = tt, s = -1;
    for (uint32_t g = 0; g <= m + n + 2; g++) {
      int64_t i = AT(pt, j);
      int64_t pj = AT(ps, i);
      if (pj < 0) { s = i; break; }
      double xv = AT(X, CX(i, pj));
      if (xv < delta) delta = xv;
      j = pj;
    }
    if (s < 0) break;                       // cannot happen with exact integer distances
    if (AT(ra, s) < delta) delta = AT(ra, s);
    // pass 2: apply
    j = tt;
    for (uint32_t g = 0; g <= m + n + 2; g++) {
      int64_t i = AT(pt, j);
      AT(X, CX(i, j)) += delta;
      int64_t pj = AT(ps, i);
      if (pj < 0) break;
      AT(X, CX(i, pj)) -= delta;
      j = pj;
    }
    AT(ra, s) -= delta;
    AT(rb, tt) -= delta;
  }
  double* o = out + P.out_off;
  for (uint32_t k = 0; k < mn; k++) o[(uint64_t)k * T + t] = AT(X, k);
  if (LDS) {                                                            // the vertex, where lp_center_kernel looks for it
    double* Xg = (double*)(ws + P.ws_off) + (uint64_t)mn * T;
    for (uint32_t k = 0; k < mn; k++) Xg[(uint64_t)k * T + t] = AT(X, k);
  }
#undef AT
#undef CX
}

// ---- The same vertex, one WAVEFRONT per trial (round 6).  With lane = trial a launch lasts as long as its largest problem's
// successive-shortest-path chain -- O((m + n)^2 m n) dependent LDS operations of ONE lane: 1.6 ms per launch at bench.py --config 2p,
// where nearly every one of the 890 dependent rounds of a step holds a node with more than 11 in- or out-edges (beyond 11 x 11 the
// state did not fit the LDS form either and went through HBM) -- while the chip runs a few hundred wavefronts.  Here the 64 lanes of a
// wavefront share one trial: the Bellman-Ford half-rounds are parallel over the columns (then the rows) -- within a half-round no
// entry depends on another of the same half, exactly as in the sequential form, where the j loop reads ds[] only and the i loop dt[]
// only -- every lane keeps the sequential loop over the other index (ascending, strict <: the lowest index wins a tie), the sink is
// the lowest column of least distance (two wave reductions), and lane 0 walks the predecessor chain.  The arithmetic of a trial is
// operation for operation that of lp_trials_kernel (integer distances; the doubles only meet min, += and -= along one simple
// path), so the answers are the same bits -- tests/test_lp_gpu.py holds both against oracle/lp.py and against each other.
__device__ __forceinline__ int64_t wave_min_i64(int64_t v) {
  for (int o = 32; o > 0; o >>= 1) { const int64_t w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
  return v;
}
__global__ __launch_bounds__(LBLK) void lp_trials_coop_kernel(const LpProblem* __restrict__ probs, const uint32_t* __restrict__ block_prob,
                                                              const uint32_t* __restrict__ block_first, const double* __restrict__ in,
                                                              const uint8_t* __restrict__ masks, uint64_t seed,
                                                              uint64_t* __restrict__ ws, double* __restrict__ out) {
  extern __shared__ uint64_t lp_lds[];
  const LpProblem P = probs[block_prob[blockIdx.x]];
  const uint32_t t = block_first[blockIdx.x] + blockIdx.y;            // grid.y = the 64 trials of the chunk
  if (t >= P.trials) return;
  const uint32_t lane = threadIdx.x;
  const uint32_t m = P.m, n = P.n, T = P.trials, mn = m * n;
  const double* a_s = in + P.in_off;
  const double* b_s = a_s + m;
  const uint8_t* pm = masks + P.mask_off;
  int64_t* C = (int64_t*)lp_lds;
  double* X = (double*)(C + mn);
  double* ra = X + mn;
  double* rb = ra + m;
  int64_t* ds = (int64_t*)(rb + n);
  int64_t* dt = ds + m;
  int64_t* ps = dt + n;
  int64_t* pt = ps + m;
#define CX(i, j) ((j) * m + (i))
  for (uint32_t k = lane; k < mn; k += LBLK) { C[k] = pm[k] ? lp_cell_cost(seed, P.pid, t, k) : 0; X[k] = 0.0; }
  for (uint32_t i = lane; i < m; i += LBLK) ra[i] = a_s[i];
  for (uint32_t j = lane; j < n; j += LBLK) rb[j] = b_s[j];
  __syncthreads();
  const uint32_t max_it = 4 * (m + n) + mn + 16;
  for (uint32_t it = 0; it < max_it; it++) {
    bool sS = false, sT = false;
    for (uint32_t i = lane; i < m; i += LBLK) { const bool s = ra[i] > 0; sS |= s; ds[i] = s ? 0 : LP_INF; ps[i] = -1; }
    for (uint32_t j = lane; j < n; j += LBLK) { sT |= rb[j] > 0; dt[j] = LP_INF; pt[j] = -1; }
    const bool anyS = __ballot(sS) != 0, anyT = __ballot(sT) != 0;
    __syncthreads();
    if (!anyS || !anyT) break;
    for (uint32_t r = 0; r < m + n; r++) {
      bool changed = false;
      for (uint32_t j = lane; j < n; j += LBLK) {
        int64_t best = LP_INF; int64_t bi = -1;
        for (uint32_t i = 0; i < m; i++) {
          const int64_t d = ds[i];
          if (d < LP_INF) { const int64_t v = d + C[CX(i, j)]; if (v < best) { best = v; bi = i; } }
        }
        if (bi >= 0 && best < dt[j]) { dt[j] = best; pt[j] = bi; changed = true; }
      }
      __syncthreads();
      for (uint32_t i = lane; i < m; i += LBLK) {
        int64_t best = LP_INF; int64_t bj = -1;
        for (uint32_t j = 0; j < n; j++) {
          if (X[CX(i, j)] > 0) {
            const int64_t d = dt[j];
            if (d < LP_INF) { const int64_t v = d - C[CX(i, j)]; if (v < best) { best = v; bj = j; } }
          }
        }
        if (bj >= 0 && best < ds[i]) { ds[i] = best; ps[i] = bj; changed = true; }
      }
      __syncthreads();
      if (__ballot(changed) == 0) break;
    }
    // the sink: least distance among the columns that still want flow, the lowest column on a tie
    int64_t lbd = LP_INF; int64_t ltt = -1;
    for (uint32_t j = lane; j < n; j += LBLK) if (rb[j] > 0 && dt[j] < lbd) { lbd = dt[j]; ltt = j; }
    const int64_t bd = wave_min_i64(lbd);
    const int64_t tt = wave_min_i64((ltt >= 0 && lbd == bd) ? ltt : LP_INF);
    if (bd >= LP_INF || tt >= LP_INF) break;
    int ok = 1;
    if (lane == 0) {
      double delta = rb[tt];
      int64_t j = tt, s = -1;
      for (uint32_t g = 0; g <= m + n + 2; g++) {
        const int64_t i = pt[j];
        const int64_t pj = ps[i];
        if (pj < 0) { s = i; break; }
        const double xv = X[CX(i, pj)];
        if (xv < delta) delta = xv;
        j = pj;
      }
      if (s < 0) ok = 0;                      // cannot happen with exact integer distances
      else {
        if (ra[s] < delta) delta = ra[s];
        j = tt;
        for (uint32_t g = 0; g <= m + n + 2; g++) {
          const int64_t i = pt[j];
          X[CX(i, j)] += delta;
          const int64_t pj = ps[i];
          if (pj < 0) break;
          X[CX(i, pj)] -= delta;
          j = pj;
        }
        ra[s] -= delta;
        rb[tt] -= delta;
      }
    }
    ok = __shfl(ok, 0, 64);
    __syncthreads();
    if (!ok) break;
  }
  double* o = out + P.out_off;
  double* Xg = (double*)(ws + P.ws_off) + (uint64_t)mn * T;              // the vertex, where lp_center_kernel looks for it
  for (uint32_t k = lane; k < mn; k += LBLK) { const double v = X[k]; o[(uint64_t)k * T + t] = v; Xg[(uint64_t)k * T + t] = v; }
#undef CX
}

#define NEWTON_MAX 100
#define NEWTON_TOL2 1e-20
#define FLOW_EPS 1e-6

// nodes (rows + columns) up to which the supported cells are centred; larger nodes keep the vertex (counted).  Up to 64 nodes the
// reachability matrix of the residual digraph is one word per node; above, (m + n) / 64 words per node in the centre workspace.
#define LP_CENTER_MAX_NODES 512
// per-trial words of the centre kernel's own workspace: y, y2, dy, yn [mn each] | S [n*n] | 14 vectors of <= max(m, n) + 1 |
// the reachability rows of a node with more than 64 rows + columns
__host__ __device__ __forceinline__ uint64_t lp_ws2_words(uint64_t m, uint64_t n) {
  const uint64_t N = m + n;
  return 4 * m * n + n * n + 14 * (N + 1) + ((N > 64 && N <= LP_CENTER_MAX_NODES) ? N * ((N + 63) / 64) : 0);
}

// oracle/lp.py:face_center + center_component for the trial of this lane.  X is the vertex left by lp_trials_kernel.
// stat[4 * trial + 0..3]: classes centred, Newton steps, classes not converged, 1 if m + n > LP_CENTER_MAX_NODES (vertex kept)
__global__ __launch_bounds__(LBLK) void lp_center_kernel(const LpProblem* __restrict__ probs, const uint32_t* __restrict__ block_prob,
                                                         const uint32_t* __restrict__ block_first, const uint8_t* __restrict__ masks,
                                                         uint64_t* __restrict__ ws, uint64_t* __restrict__ ws2, double* __restrict__ out,
                                                         uint32_t* __restrict__ stat) {
  const LpProblem P = probs[block_prob[blockIdx.x]];
  const uint32_t t = block_first[blockIdx.x] + threadIdx.x;
  if (t >= P.trials) return;
  const uint32_t m = P.m, n = P.n, T = P.trials, mn = m * n, N = m + n;
  const uint8_t* pm = masks + P.mask_off;
  int64_t* C = (int64_t*)(ws + P.ws_off);                   // reused: the cells of the class being centred
  double* X = (double*)(C + (uint64_t)mn * T);
  double* ra_ = X + (uint64_t)mn * T;
  double* rb_ = ra_ + (uint64_t)m * T;
  uint64_t* reach = (uint64_t*)(rb_ + (uint64_t)n * T);     // ds | dt : N words
  int64_t* label = (int64_t*)(reach + (uint64_t)N * T);     // ps | pt : N words
  double* W = (double*)(ws2 + P.ws2_off);
  double* y = W;
  double* y2 = y + (uint64_t)mn * T;
  double* dy = y2 + (uint64_t)mn * T;
  double* yn = dy + (uint64_t)mn * T;
  double* S = yn + (uint64_t)mn * T;
  double* vec = S + (uint64_t)n * n * T;
  const uint64_t VL = (uint64_t)(N + 1) * T;                // one vector slot (any of them holds m or n + 1 entries)
  double *an = vec, *bn = vec + VL, *nur = vec + 2 * VL, *nuc = vec + 3 * VL, *nrn = vec + 4 * VL, *ncn = vec + 5 * VL, *Dr = vec + 6 * VL,
         *Dc = vec + 7 * VL, *gr = vec + 8 * VL, *gc = vec + 9 * VL, *wr = vec + 10 * VL, *wc = vec + 11 * VL, *h = vec + 12 * VL;
  int64_t* rstart = (int64_t*)(vec + 13 * VL);              // first cell of every local row (E is row-major), nr + 1 entries
#define AT(arr, e) arr[(uint64_t)(e) * T + t]
#define CX(i, j) ((uint64_t)((j) * m + (i)))
  uint32_t st_comp = 0, st_steps = 0, st_bad = 0, st_large = 0;
  double* o = out + P.out_off;
  if (N > LP_CENTER_MAX_NODES) {
    st_large = 1;
  } else {
    // ---- classes of the residual digraph (flows at or below 1e-6 of the largest flow open no arc: rounding residue of the balancing)
    double xmax = 0.0;
    for (uint32_t i = 0; i < m; i++)
      for (uint32_t j = 0; j < n; j++) { const double v = AT(X, CX(i, j)); if (v > xmax) xmax = v; }
    const double eps = FLOW_EPS * xmax;
    if (N <= 64) {
      for (uint32_t i = 0; i < m; i++) {
        uint64_t r = 1ULL << i;
        for (uint32_t j = 0; j < n; j++) if (!pm[CX(i, j)]) r |= 1ULL << (m + j);
        AT(reach, i) = r;
      }
      for (uint32_t j = 0; j < n; j++) {
        uint64_t r = 1ULL << (m + j);
        for (uint32_t i = 0; i < m; i++) if (!pm[CX(i, j)] && AT(X, CX(i, j)) > eps) r |= 1ULL << i;
        AT(reach, m + j) = r;
      }
      for (bool changed = true; changed;) {
        changed = false;
        for (uint32_t u = 0; u < N; u++) {
          const uint64_t r = AT(reach, u);
          uint64_t acc = r;
          for (uint32_t v = 0; v < N; v++) if ((r >> v) & 1) acc |= AT(reach, v);
          if (acc != r) { AT(reach, u) = acc; changed = true; }
        }
      }
      for (uint32_t u = 0; u < N; u++) {
        const uint64_t r = AT(reach, u);
        int64_t l = u;
        for (uint32_t v = 0; v < N; v++) if (((r >> v) & 1) && ((AT(reach, v) >> u) & 1)) { l = v; break; }
        AT(label, u) = l;
      }
    } else {
      // the same closure with NW words per node (Warshall, one pass: after step k a row holds what its node reaches through
      // nodes <= k): the transitive closure is unique, so the classes are those of the sweep above
      const uint32_t NW = (N + 63) >> 6;
      uint64_t* RB = (uint64_t*)(vec + 14 * VL);
#define RW(u, w) AT(RB, (uint64_t)(u) * NW + (w))
#define RBIT(u, v) ((RW(u, (v) >> 6) >> ((v) & 63)) & 1ULL)
      for (uint32_t u = 0; u < N; u++) for (uint32_t w = 0; w < NW; w++) RW(u, w) = 0;
      for (uint32_t i = 0; i < m; i++) {
        RW(i, i >> 6) |= 1ULL << (i & 63);
        for (uint32_t j = 0; j < n; j++) if (!pm[CX(i, j)]) RW(i, (m + j) >> 6) |= 1ULL << ((m + j) & 63);
      }
      for (uint32_t j = 0; j < n; j++) {
        RW(m + j, (m + j) >> 6) |= 1ULL << ((m + j) & 63);
        for (uint32_t i = 0; i < m; i++) if (!pm[CX(i, j)] && AT(X, CX(i, j)) > eps) RW(m + j, i >> 6) |= 1ULL << (i & 63);
      }
      for (uint32_t k = 0; k < N; k++)
        for (uint32_t u = 0; u < N; u++)
          if (u != k && RBIT(u, k))
            for (uint32_t w = 0; w < NW; w++) RW(u, w) |= RW(k, w);
      for (uint32_t u = 0; u < N; u++) {
        int64_t l = u;
        for (uint32_t v = 0; v < N; v++) if (RBIT(u, v) && RBIT(v, u)) { l = v; break; }
        AT(label, u) = l;
      }
#undef RBIT
#undef RW
    }
    // ---- every class with more supported cells than a tree
    for (uint32_t L = 0; L < m; L++) {
      if (AT(label, L) != (int64_t)L) continue;
      uint32_t nr = 0, nc = 0, ne = 0;
      for (uint32_t j = 0; j < n; j++) if (AT(label, m + j) == (int64_t)L) nc++;
      if (!nc) continue;
      // cells in row-major order: packed (i, j, local row, local column), 16 bits each
      for (uint32_t i = 0; i < m; i++) {
        if (AT(label, i) != (int64_t)L) continue;
        AT(rstart, nr) = ne;
        uint32_t cl = 0;
        for (uint32_t j = 0; j < n; j++) {
          if (AT(label, m + j) != (int64_t)L) continue;
          if (!pm[CX(i, j)]) { AT(C, ne) = (int64_t)((uint64_t)i | ((uint64_t)j << 16) | ((uint64_t)nr << 32) | ((uint64_t)cl << 48)); ne++; }
          cl++;
        }
        nr++;
      }
      AT(rstart, nr) = ne;
      if (ne <= nr + nc - 1) continue;
      st_comp++;
      const uint32_t q = nc - 1;
#define EI(e) ((uint32_t)((uint64_t)AT(C, e) & 0xFFFF))
#define EJ(e) ((uint32_t)(((uint64_t)AT(C, e) >> 16) & 0xFFFF))
#define ER(e) ((uint32_t)(((uint64_t)AT(C, e) >> 32) & 0xFFFF))
#define EC(e) ((uint32_t)(((uint64_t)AT(C, e) >> 48) & 0xFFFF))
      for (uint32_t k = 0; k < nr; k++) AT(an, k) = 0.0;
      for (uint32_t k = 0; k < nc; k++) AT(bn, k) = 0.0;
      for (uint32_t e = 0; e < ne; e++) {
        const double v = AT(X, CX(EI(e), EJ(e)));
        AT(an, ER(e)) += v;
        AT(bn, EC(e)) += v;
      }
      double Tt = 0.0;
      for (uint32_t k = 0; k < nr; k++) Tt += AT(an, k);
      const double sc = Tt / (double)ne;
      for (uint32_t k = 0; k < nr; k++) AT(an, k) = AT(an, k) / sc;
      for (uint32_t k = 0; k < nc; k++) AT(bn, k) = AT(bn, k) / sc;
      const double Tn = Tt / sc;
      for (uint32_t e = 0; e < ne; e++) AT(y, e) = AT(an, ER(e)) * AT(bn, EC(e)) / Tn;
      for (uint32_t k = 0; k < nr; k++) AT(nur, k) = 0.0;
      for (uint32_t k = 0; k < nc; k++) AT(nuc, k) = 0.0;
      // |r|^2 of (yy, nr_, nc_): dual part cell by cell, then the rows, then the kept columns (sums over e ascending)
#define RESIDUAL2(yy, nr_, nc_, res)                                                             \
      {                                                                                            \
        double acc_ = 0.0;                                                                         \
        for (uint32_t e = 0; e < ne; e++) {                                                        \
          const double d_ = AT(nr_, ER(e)) + AT(nc_, EC(e)) - 1.0 / AT(yy, e);                     \
          acc_ += d_ * d_;                                                                         \
        }                                                                                          \
        for (uint32_t k = 0; k < nr; k++) AT(gr, k) = 0.0;                                         \
        for (uint32_t k = 0; k < nc; k++) AT(gc, k) = 0.0;                                         \
        for (uint32_t e = 0; e < ne; e++) { AT(gr, ER(e)) += AT(yy, e); AT(gc, EC(e)) += AT(yy, e); } \
        for (uint32_t k = 0; k < nr; k++) { const double d_ = AT(gr, k) - AT(an, k); acc_ += d_ * d_; } \
        for (uint32_t k = 0; k < q; k++) { const double d_ = AT(gc, k) - AT(bn, k); acc_ += d_ * d_; }  \
        res = acc_;                                                                                \
      }
      double r2;
      RESIDUAL2(y, nur, nuc, r2);
      bool done = false;
      uint32_t its = 0;
      for (its = 1; its <= NEWTON_MAX; its++) {
        for (uint32_t k = 0; k < nr; k++) { AT(Dr, k) = 0.0; AT(gr, k) = 0.0; }
        for (uint32_t k = 0; k < nc; k++) { AT(Dc, k) = 0.0; AT(gc, k) = 0.0; }
        for (uint32_t e = 0; e < ne; e++) {
          const double v = AT(y, e), v2 = v * v;
          AT(y2, e) = v2;
          AT(Dr, ER(e)) += v2; AT(gr, ER(e)) += v;
          AT(Dc, EC(e)) += v2; AT(gc, EC(e)) += v;
        }
        for (uint32_t k = 0; k < nr; k++) AT(gr, k) = 2.0 * AT(gr, k) - AT(an, k);
        for (uint32_t k = 0; k < nc; k++) AT(gc, k) = 2.0 * AT(gc, k) - AT(bn, k);
        // Schur complement on the kept columns; its diagonal in the cancellation-free form (see oracle/lp.py)
#define SS(a_, b_) AT(S, (a_) * q + (b_))
        for (uint32_t k = 0; k < q; k++) {
          for (uint32_t k2 = 0; k2 < q; k2++) SS(k, k2) = 0.0;
          AT(h, k) = AT(gc, k);
        }
        for (uint32_t i = 0; i < nr; i++) {
          const uint32_t e0 = (uint32_t)AT(rstart, i), e1 = (uint32_t)AT(rstart, i + 1);
          const double inv = 1.0 / AT(Dr, i);
          const double gri = AT(gr, i);
          for (uint32_t ea = e0; ea < e1; ea++) {
            const uint32_t k1 = EC(ea);
            if (k1 >= q) continue;
            const double f = AT(y2, ea) * inv;
            AT(h, k1) -= f * gri;
            double oth = 0.0;
            for (uint32_t eb = e0; eb < e1; eb++) {
              if (eb == ea) continue;
              const double v2 = AT(y2, eb);
              oth += v2;
              const uint32_t k2 = EC(eb);
              if (k2 < q) SS(k1, k2) -= f * v2;
            }
            SS(k1, k1) += f * oth;
          }
        }
        bool fail = false;
        for (uint32_t k = 0; k < q; k++) {
          const double piv = SS(k, k);
          if (!(piv > 0.0)) { fail = true; break; }
          for (uint32_t r_ = k + 1; r_ < q; r_++) {
            const double f = SS(r_, k) / piv;
            if (f != 0.0) {
              for (uint32_t c_ = k + 1; c_ < q; c_++) SS(r_, c_) -= f * SS(k, c_);
              AT(h, r_) -= f * AT(h, k);
            }
          }
        }
        if (fail) break;
        for (uint32_t k = 0; k < nc; k++) AT(wc, k) = 0.0;
        for (int32_t k = (int32_t)q - 1; k >= 0; k--) {
          double acc = AT(h, k);
          for (uint32_t c_ = k + 1; c_ < q; c_++) acc -= SS(k, c_) * AT(wc, c_);
          AT(wc, k) = acc / SS(k, k);
        }
        for (uint32_t i = 0; i < nr; i++) {
          double acc = AT(gr, i);
          for (uint32_t e = (uint32_t)AT(rstart, i); e < (uint32_t)AT(rstart, i + 1); e++)
            if (EC(e) < q) acc -= AT(y2, e) * AT(wc, EC(e));
          AT(wr, i) = acc / AT(Dr, i);
        }
        for (uint32_t e = 0; e < ne; e++) AT(dy, e) = AT(y, e) - AT(y2, e) * (AT(wr, ER(e)) + AT(wc, EC(e)));
        double tt = 1.0;
        const double tmin = 1.0 / 1099511627776.0;           // 2^-40
        bool ok = false;
        while (tt >= tmin) {
          ok = true;
          for (uint32_t e = 0; e < ne; e++) if (!(AT(y, e) + tt * AT(dy, e) > 0.0)) { ok = false; break; }
          if (ok) break;
          tt *= 0.5;
        }
        if (!ok) break;
        double r2n;
        for (;;) {
          for (uint32_t e = 0; e < ne; e++) AT(yn, e) = AT(y, e) + tt * AT(dy, e);
          for (uint32_t k = 0; k < nr; k++) AT(nrn, k) = AT(nur, k) + tt * (AT(wr, k) - AT(nur, k));
          for (uint32_t k = 0; k < nc; k++) AT(ncn, k) = AT(nuc, k) + tt * (AT(wc, k) - AT(nuc, k));
          RESIDUAL2(yn, nrn, ncn, r2n);
          const double f = 1.0 - 0.01 * tt;
          if (r2n <= f * f * r2) break;
          tt *= 0.5;
          if (tt < tmin) { fail = true; break; }
        }
        if (fail) break;
        for (uint32_t e = 0; e < ne; e++) AT(y, e) = AT(yn, e);
        for (uint32_t k = 0; k < nr; k++) AT(nur, k) = AT(nrn, k);
        for (uint32_t k = 0; k < nc; k++) AT(nuc, k) = AT(ncn, k);
        r2 = r2n;
        if (r2 <= NEWTON_TOL2) { done = true; break; }
      }
      if (its > NEWTON_MAX) its = NEWTON_MAX;
      st_steps += its;
      if (!done) st_bad++;                    // (the class keeps its vertex flows)
      else for (uint32_t e = 0; e < ne; e++) AT(X, CX(EI(e), EJ(e))) = AT(y, e) * sc;
#undef SS
#undef RESIDUAL2
#undef EI
#undef EJ
#undef ER
#undef EC
    }
  }
  for (uint32_t k = 0; k < mn; k++) o[(uint64_t)k * T + t] = AT(X, k);
  uint32_t* sp = stat + 4 * (P.stat_off + t);
  sp[0] = st_comp; sp[1] = st_steps; sp[2] = st_bad; sp[3] = st_large;
#undef AT
#undef CX
}

// n_problems problems; for problem p: m[p], n[p], trials[p], pid[p]; a_s/b_s concatenated in `ab`
// (m+n doubles per problem), unsupported-cell masks concatenated in `mask` (m*n bytes, index j*m+i).
// flows_out: for problem p, trials[p]*m*n doubles laid out [cell][trial], problems concatenated.
extern "C" int shn_lp_solve_batch(shn_ctx* ctx, uint32_t n_problems, const uint32_t* m, const uint32_t* n, const uint32_t* trials,
                                  const uint64_t* pid, const double* ab, const uint8_t* mask, uint64_t seed, double* flows_out) {
  if (!ctx || (n_problems && (!m || !n || !trials || !pid || !ab || !mask || !flows_out)))
    return shn_fail(SHN_ERR_ARG, "shn_lp_solve_batch: NULL argument");
  if (!n_problems) return SHN_OK;
  SHN_ENTER(ctx);
  hipStream_t s = ctx->stream; shn_use_stream(s);
  const bool center = ctx->lp_rule != SHN_LP_RULE_VERTEX;
  std::vector<LpProblem> probs(n_problems);
  std::vector<uint32_t> bprob, bfirst, lprob, lfirst, cprob, cfirst, wprob, wfirst;       // blocks of the vertex kernel (state in HBM / in LDS) / of the centre kernel / chunks of the wavefront-per-trial kernel
  uint64_t lds_words = 0, coop_words = 0;
  const char* coop_env = getenv("SHN_LP_COOP");
  const bool use_coop = !(coop_env && coop_env[0] == '0');
  // the in-LDS trial kernel asks for up to LP_LDS_WORDS * 8 = 156 KB of dynamic LDS: asked for once per process; a device or driver
  // that does not grant it gets the HBM form of the same kernel for every problem
  static const bool lds_granted = []() {
    const hipError_t e = hipFuncSetAttribute((const void*)lp_trials_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LP_LDS_WORDS * 8));
    if (e != hipSuccess) { (void)hipGetLastError(); fprintf(stderr, "[shannon_amd] LP: %d bytes of LDS per block not granted (%s): trial state stays in HBM\n", (int)(LP_LDS_WORDS * 8), hipGetErrorString(e)); }
    return e == hipSuccess;
  }();
  uint64_t n_large_trials = 0;
  uint64_t in_off = 0, mask_off = 0, ws_off = 0, out_off = 0, ws2_off = 0, stat_off = 0;
  for (uint32_t p = 0; p < n_problems; p++) {
    if (m[p] == 0 || n[p] == 0 || trials[p] == 0) return shn_fail(SHN_ERR_ARG, "shn_lp_solve_batch: empty problem");
    if (m[p] >= 65536 || n[p] >= 65536) return shn_fail(SHN_ERR_ARG, "shn_lp_solve_batch: more than 65535 rows or columns");
    LpProblem& P = probs[p];
    P.m = m[p]; P.n = n[p]; P.trials = trials[p]; P.pad = 0; P.pid = pid[p];
    P.in_off = in_off; P.mask_off = mask_off; P.ws_off = ws_off; P.out_off = out_off; P.ws2_off = ws2_off; P.stat_off = stat_off;
    uint64_t mn = (uint64_t)m[p] * n[p];
    in_off += m[p] + n[p];
    mask_off += mn;
    ws_off += (2 * mn + 3ULL * (m[p] + n[p])) * trials[p];
    stat_off += trials[p];
    out_off += mn * trials[p];
    // a wavefront per trial (lp_trials_coop_kernel) wherever one trial's state fits 64 KB of LDS (m = n = 62); SHN_LP_COOP=0: lane = trial
    const uint64_t words = 2 * mn + 3ULL * (m[p] + n[p]);
    const bool coop = use_coop && words * 8 <= 65536;
    const bool in_lds = !coop && lds_granted && words * LBLK <= LP_LDS_WORDS;
    for (uint32_t f = 0; f < trials[p]; f += LBLK) {
      if (coop) { wprob.push_back(p); wfirst.push_back(f); coop_words = std::max<uint64_t>(coop_words, words); continue; }
      (in_lds ? lprob : bprob).push_back(p); (in_lds ? lfirst : bfirst).push_back(f);
      lds_words = in_lds ? std::max<uint64_t>(lds_words, words * LBLK) : lds_words;
    }
    // the centre kernel has work only where the supported cells of the problem hold a cycle (rows and columns as vertices, a
    // supported cell as an edge): on a forest every class is a tree and its face a point.  At BASELINE configs[2] 9,000
    // problems per step, a handful with a cycle.
    if (center && m[p] + n[p] <= LP_CENTER_MAX_NODES) {
      int par[LP_CENTER_MAX_NODES];
      for (uint32_t v = 0; v < m[p] + n[p]; v++) par[v] = (int)v;
      auto find = [&](int v) { while (par[v] != v) { par[v] = par[par[v]]; v = par[v]; } return v; };
      bool cyc = false;
      const uint8_t* pm = mask + P.mask_off;
      for (uint32_t j = 0; j < n[p] && !cyc; j++)
        for (uint32_t i = 0; i < m[p]; i++) {
          if (pm[(uint64_t)j * m[p] + i]) continue;
          const int a = find((int)i), b = find((int)(m[p] + j));
          if (a == b) { cyc = true; break; }
          par[a] = b;
        }
      if (cyc) {
        ws2_off += lp_ws2_words(m[p], n[p]) * trials[p];            // (P.ws2_off was taken before: this problem's share follows)
        for (uint32_t f = 0; f < trials[p]; f += LBLK) { cprob.push_back(p); cfirst.push_back(f); }
      }
    } else if (center) n_large_trials += trials[p];
  }
  TimerRegion treg(ctx, T_LP);
  // per-context workspaces: two batches may be in flight on two contexts (the deferred back half of a step beside the next
  // step's front half), and shn_ws_release_idle never touches a context's own slots
  // ONE image of everything that goes up (problems, block lists, a / b, masks) in pinned memory and one transfer of it, one
  // transfer back (flows + the centre kernel's statistics): a batch used to be seven small pageable copies up and two down, and
  // the 110 dependent batches of a step are mostly the latency of their copies (0.54 ms per batch on the stream, 0.1 of it kernels)
  void *pup, *pws, *pdown, *pws2;
  int rc;
  auto al = [](uint64_t x) { return (x + 15) & ~15ULL; };
  const uint64_t o_probs = 0, o_blocks = al(o_probs + probs.size() * sizeof(LpProblem)),
                 o_in = al(o_blocks + (bprob.size() + lprob.size() + cprob.size() + wprob.size()) * 8), o_mask = al(o_in + in_off * 8), up_bytes = al(o_mask + mask_off) + 16;
  const uint64_t o_out = 0, o_stat = al(out_off * 8), down_bytes = o_stat + stat_off * 16 + 16;
  void *h_up, *h_down;
  if ((rc = ctx->cws[4].get(up_bytes, &pup)) || (rc = ctx->cws[8].get(ws_off * 8 + 16, &pws)) || (rc = ctx->cws[9].get(down_bytes, &pdown)) ||
      (rc = ctx->cws[10].get(ws2_off * 8 + 16, &pws2)) || (rc = ctx->hpin[0].get(up_bytes, &h_up)) || (rc = ctx->hpin[1].get(down_bytes, &h_down))) return rc;
  uint8_t* hu = (uint8_t*)h_up;
  memcpy(hu + o_probs, probs.data(), probs.size() * sizeof(LpProblem));
  uint32_t* hb = (uint32_t*)(hu + o_blocks);
  auto put = [&](const std::vector<uint32_t>& v) { if (!v.empty()) memcpy(hb, v.data(), v.size() * 4); hb += v.size(); };
  put(bprob); put(bfirst); put(cprob); put(cfirst); put(lprob); put(lfirst); put(wprob); put(wfirst);
  memcpy(hu + o_in, ab, in_off * 8);
  memcpy(hu + o_mask, mask, mask_off);
  uint8_t* du = (uint8_t*)pup;
  void* pp = du + o_probs;
  uint32_t* d_bprob = (uint32_t*)(du + o_blocks);
  uint32_t* d_bfirst = d_bprob + bprob.size();
  uint32_t* d_cprob = d_bfirst + bprob.size();
  uint32_t* d_cfirst = d_cprob + cprob.size();
  uint32_t* d_lprob = d_cfirst + cprob.size();
  uint32_t* d_lfirst = d_lprob + lprob.size();
  uint32_t* d_wprob = d_lfirst + lprob.size();
  uint32_t* d_wfirst = d_wprob + wprob.size();
  void* pin = du + o_in;
  void* pm = du + o_mask;
  void* pout = (uint8_t*)pdown + o_out;
  void* pst = (uint8_t*)pdown + o_stat;
  HIP_TRY(hipMemcpyAsync(pup, h_up, up_bytes - 16, hipMemcpyHostToDevice, s));
  if (!wprob.empty()) {
    TimerRegion ttr(ctx, T_LP_TRIALS);
    { uint64_t b = 0;       // per trial: a, b and the mask read (8 (m + n) + m n), the vertex written twice (the answer and the centre kernel's copy)
      for (uint32_t p = 0; p < n_problems; p++) { const uint64_t mn_ = (uint64_t)m[p] * n[p]; if ((2 * mn_ + 3ULL * (m[p] + n[p])) * 8 <= 65536) b += (uint64_t)trials[p] * (8ULL * (m[p] + n[p]) + mn_ + 16 * mn_); }
      ttr.bytes(b); }
    hipLaunchKernelGGL(lp_trials_coop_kernel, dim3((uint32_t)wprob.size(), LBLK), dim3(LBLK), coop_words * 8, s, (const LpProblem*)pp, d_wprob, d_wfirst,
                       (const double*)pin, (const uint8_t*)pm, seed, (uint64_t*)pws, (double*)pout);
  }
  if (!lprob.empty()) {
    hipLaunchKernelGGL(lp_trials_kernel<true>, dim3((uint32_t)lprob.size()), dim3(LBLK), lds_words * 8, s, (const LpProblem*)pp, d_lprob, d_lfirst,
                       (const double*)pin, (const uint8_t*)pm, seed, (uint64_t*)pws, (double*)pout);
  }
  if (!bprob.empty())
    hipLaunchKernelGGL(lp_trials_kernel<false>, dim3((uint32_t)bprob.size()), dim3(LBLK), 0, s, (const LpProblem*)pp, d_bprob, d_bfirst,
                       (const double*)pin, (const uint8_t*)pm, seed, (uint64_t*)pws, (double*)pout);
  const bool with_stat = center && !cprob.empty();
  if (with_stat) {
    HIP_TRY(hipMemsetAsync(pst, 0, stat_off * 16, s));                     // (trials the centre kernel does not visit: nothing centred)
    hipLaunchKernelGGL(lp_center_kernel, dim3((uint32_t)cprob.size()), dim3(LBLK), 0, s, (const LpProblem*)pp, d_cprob, d_cfirst,
                       (const uint8_t*)pm, (uint64_t*)pws, (uint64_t*)pws2, (double*)pout, (uint32_t*)pst);
  }
  HIP_TRY(hipMemcpyAsync(h_down, pdown, with_stat ? o_stat + stat_off * 16 : out_off * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  memcpy(flows_out, (const uint8_t*)h_down + o_out, out_off * 8);
  std::vector<uint32_t> stat;
  if (with_stat) { stat.resize(4 * stat_off); memcpy(stat.data(), (const uint8_t*)h_down + o_stat, stat.size() * 4); }
  // census (bench.py: lp_calls / lp_degenerate): a problem is degenerate when the optimal face of one of its trials was not a point
  // (a fork's census is its parent's: the batches of the sparse flow run on the graph threads' forks, several at a time)
  // (summed here, added to the owner's census in one locked step at the end: shn_lp_census_add looks the parent up under the lock
  // shn_ctx_destroy takes when it orphans the forks of a context that goes away)
  uint64_t census[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto tally = [&](int i, uint64_t v) { census[i] += v; };
  tally(0, n_problems);
  tally(2, stat_off);
  tally(6, n_large_trials);
  if (n_large_trials) {                                                  // (said once per process: these problems keep the vertex answer)
    static std::atomic<bool> told{false};
    if (!told.exchange(true))
      fprintf(stderr, "[shannon_amd] LP: a decomposition with more than 512 rows + columns keeps the vertex of its optimal face (the analytic centre is "
                      "computed for problems up to that size); counted in shn_lp_stats[6]\n");
  }
  if (!stat.empty()) {
    uint64_t t0 = 0;
    for (uint32_t p = 0; p < n_problems; p++) {
      bool deg = false;
      for (uint32_t t = 0; t < trials[p]; t++) {
        const uint32_t* sp = &stat[4 * (t0 + t)];
        if (sp[0]) { deg = true; tally(3, 1); }
        tally(4, sp[1]);
        tally(5, sp[2]);
      }
      if (deg) tally(1, 1);
      t0 += trials[p];
    }
  }
  shn_lp_census_add(ctx, census);
  return SHN_OK;
}

extern "C" int shn_lp_set_rule(shn_ctx* ctx, int rule) {
  if (!ctx || (rule != SHN_LP_RULE_VERTEX && rule != SHN_LP_RULE_CENTER)) return shn_fail(SHN_ERR_ARG, "shn_lp_set_rule: bad argument");
  ctx->lp_rule = rule;
  return SHN_OK;
}

extern "C" int shn_lp_stats(shn_ctx* ctx, uint64_t* out8, int reset) {
  if (!ctx || !out8) return shn_fail(SHN_ERR_ARG, "shn_lp_stats: NULL argument");
  for (int i = 0; i < 8; i++) out8[i] = ctx->lp_stats[i];
  out8[7] = (uint64_t)ctx->lp_rule;
  if (reset) for (int i = 0; i < 7; i++) ctx->lp_stats[i] = 0;
  return SHN_OK;
}
