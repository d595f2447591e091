// Batched sparse-flow node decomposition LPs on gfx950 (row a28): replaces the <=100
// cvxopt.solvers.lp calls per decomposed node of path_decompose (path_decompose_sparse.py:100-117).
//
// Each trial is a tiny transportation problem  min c.x, row sums a, column sums b, x >= 0  with
// cost numerators c = |Irwin-Hall(12) - 6| * 2^32 on unsupported cells, 0 on supported cells
// (counter-based splitmix64 stream -- integer arithmetic, identical on CPU and GPU).  It is solved
// by successive shortest paths with Jacobi Bellman-Ford rounds and lowest-index tie-breaks; the
// exact sequence is specified in oracle/lp.py:transport_vertex and restated here.  The trials of
// one node run in the lanes of a wavefront (lane = trial): same m, n, a, b, support mask, different
// costs.  All per-trial state lives in a [element][trial] workspace so lane accesses coalesce.
// No MFMA: no dense contraction anywhere.
#include "common.h"

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
};

// workspace layout per problem (T = trials), all [elem][trial]:
//   C[mn] int64 | X[mn] double | ra[m] | rb[n] | ds[m] int64 | dt[n] int64 | ps[m] int64 | pt[n] int64
__global__ __launch_bounds__(LBLK) void lp_trials_kernel(const LpProblem* __restrict__ probs, const uint32_t* __restrict__ block_prob,
                                                         const uint32_t* __restrict__ block_first, const double* __restrict__ in,
                                                         const uint8_t* __restrict__ masks, uint64_t seed,
                                                         uint64_t* __restrict__ ws, double* __restrict__ out) {
  const LpProblem P = probs[block_prob[blockIdx.x]];
  const uint32_t t = block_first[blockIdx.x] + threadIdx.x;
  if (t >= P.trials) return;
  const uint32_t m = P.m, n = P.n, T = P.trials, mn = m * n;
  const double* a_s = in + P.in_off;
  const double* b_s = a_s + m;
  const uint8_t* pm = masks + P.mask_off;
  int64_t* C = (int64_t*)(ws + P.ws_off);
  double* X = (double*)(C + (uint64_t)mn * T);
  double* ra = X + (uint64_t)mn * T;
  double* rb = ra + (uint64_t)m * T;
  int64_t* ds = (int64_t*)(rb + (uint64_t)n * T);
  int64_t* dt = ds + (uint64_t)m * T;
  int64_t* ps = dt + (uint64_t)n * T;
  int64_t* pt = ps + (uint64_t)m * T;
#define AT(arr, e) arr[(uint64_t)(e) * T + t]
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
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  std::vector<LpProblem> probs(n_problems);
  std::vector<uint32_t> bprob, bfirst;
  uint64_t in_off = 0, mask_off = 0, ws_off = 0, out_off = 0;
  for (uint32_t p = 0; p < n_problems; p++) {
    if (m[p] == 0 || n[p] == 0 || trials[p] == 0) return shn_fail(SHN_ERR_ARG, "shn_lp_solve_batch: empty problem");
    LpProblem& P = probs[p];
    P.m = m[p]; P.n = n[p]; P.trials = trials[p]; P.pad = 0; P.pid = pid[p];
    P.in_off = in_off; P.mask_off = mask_off; P.ws_off = ws_off; P.out_off = out_off;
    uint64_t mn = (uint64_t)m[p] * n[p];
    in_off += m[p] + n[p];
    mask_off += mn;
    ws_off += (2 * mn + 3ULL * (m[p] + n[p])) * trials[p];
    out_off += mn * trials[p];
    for (uint32_t f = 0; f < trials[p]; f += LBLK) { bprob.push_back(p); bfirst.push_back(f); }
  }
  TimerRegion treg(ctx, T_LP);
  void *pp, *pb, *pin, *pm, *pws, *pout;
  int rc;
  if ((rc = g_shn_ws[18].get(probs.size() * sizeof(LpProblem), &pp)) || (rc = g_shn_ws[19].get(bprob.size() * 8 + 16, &pb)) ||
      (rc = g_shn_ws[20].get(in_off * 8 + 16, &pin)) || (rc = g_shn_ws[21].get(mask_off + 16, &pm)) ||
      (rc = g_shn_ws[22].get(ws_off * 8 + 16, &pws)) || (rc = g_shn_ws[23].get(out_off * 8 + 16, &pout))) return rc;
  uint32_t* d_bprob = (uint32_t*)pb;
  uint32_t* d_bfirst = d_bprob + bprob.size();
  HIP_TRY(hipMemcpyAsync(pp, probs.data(), probs.size() * sizeof(LpProblem), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_bprob, bprob.data(), bprob.size() * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_bfirst, bfirst.data(), bfirst.size() * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(pin, ab, in_off * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(pm, mask, mask_off, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(lp_trials_kernel, dim3((uint32_t)bprob.size()), dim3(LBLK), 0, s, (const LpProblem*)pp, d_bprob, d_bfirst,
                     (const double*)pin, (const uint8_t*)pm, seed, (uint64_t*)pws, (double*)pout);
  HIP_TRY(hipMemcpyAsync(flows_out, pout, out_off * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return SHN_OK;
}
