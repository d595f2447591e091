/* CPU restatement of load_kmers + the greedy contig extension (rows a3-a4) in plain C -- TEST INFRASTRUCTURE and part of the
 * `cpu_baseline` ("port") of bench.py; never linked into the product.
 *
 * Follows extension_correction.py:142-149 (lowComplexity), :202-221 (load_kmers), :223-245 (extend / extend_right / extend_left),
 * :159-166 (argmax: strict >, candidates in the order A, G, C, T), :334-354 (seeds by descending weight -- stable ascending sort +
 * pop(), so with the input written k1-mer-descending equal weights come in ascending k1-mer order -- until the weight
 * drops below min_weight; a seed already traversed is skipped; every k1-mer is traversed at most once, globally).
 *
 * Input: the canonical count table of the strand-doubled reads (what oracle_count_canonical returns: canonical keys ascending,
 * counts).  The dictionary the reference loads holds BOTH orientations of every k1-mer (count(x) == count(rc x) in the doubled
 * input; a k1-mer that is its own reverse complement occurs once with twice the count); it is rebuilt here as a sorted array of
 * oriented keys.  Keys: 2 bits per base, first base in the high bits, A C G T = 0 1 2 3 (integer order = string order).
 *
 * Output, for every walk that is not void, in seed order: the seed (oriented key), the number of right and left steps, the sum of
 * the weights (seed included), and the appended / prepended bases (codes 0..3; right steps in walking order, then left steps in
 * walking order) -- the contig is reverse(left bases) + seed + right bases.  Returns the number of such walks.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint64_t rc_key(uint64_t key, int k) {
  uint64_t out = 0;
  for (int i = 0; i < k; i++) { out = (out << 2) | (3 - (key & 3)); key >>= 2; }
  return out;
}

static int low_complexity(uint64_t key, int k) {
  int cnt[4] = {0, 0, 0, 0};
  for (int i = 0; i < k; i++) { cnt[key & 3]++; key >>= 2; }
  int mx = cnt[0];
  for (int b = 1; b < 4; b++) if (cnt[b] > mx) mx = cnt[b];
  return mx >= k - 2;
}

typedef struct { uint64_t key; uint32_t w; } Ent;


/* seeds: weight descending, key ascending */
typedef struct { uint32_t w; uint32_t idx; } Seed;
static int cmp_seed(const void* a, const void* b) {
  const Seed* x = (const Seed*)a; const Seed* y = (const Seed*)b;
  if (x->w != y->w) return x->w > y->w ? -1 : 1;
  return x->idx < y->idx ? -1 : x->idx > y->idx ? 1 : 0;       /* the entries are sorted by key: index order = key order */
}

/* the dictionary: entries sorted by key + a table of where every value of the key's top PBITS bits begins (a look-up is a
 * bisection of a handful of entries instead of 25 cache-missing steps over tens of millions; the reference's dict is O(1) too) */
#define PBITS 24
static const uint32_t* g_pref;     /* [2^PBITS + 1] */
static int g_shift;                /* key >> g_shift = its top PBITS bits (0 if the keys are shorter) */
static int64_t find(const Ent* e, uint64_t n, uint64_t key) {
  (void)n;
  const uint64_t p = key >> g_shift;
  uint64_t lo = g_pref[p], hi = g_pref[p + 1];
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if (e[mid].key == key) return (int64_t)mid;
    if (e[mid].key < key) lo = mid + 1; else hi = mid;
  }
  return -1;
}

/* stable LSD radix sort of the entries by key (16-bit digits over the 2 k1 key bits): qsort of 42 M entries was most of the
 * baseline's extension stage at a million reads */
static void sort_entries(Ent* e, uint64_t n, int key_bits) {
  Ent* tmp = (Ent*)malloc((n + 1) * sizeof(Ent));
  uint64_t* cnt = (uint64_t*)malloc(65537 * sizeof(uint64_t));
  Ent *a = e, *b = tmp;
  for (int shift = 0; shift < key_bits; shift += 16) {
    memset(cnt, 0, 65537 * sizeof(uint64_t));
    for (uint64_t i = 0; i < n; i++) cnt[((a[i].key >> shift) & 0xFFFF) + 1]++;
    for (int d = 0; d < 65536; d++) cnt[d + 1] += cnt[d];
    for (uint64_t i = 0; i < n; i++) b[cnt[(a[i].key >> shift) & 0xFFFF]++] = a[i];
    Ent* t = a; a = b; b = t;
  }
  if (a != e) memcpy(e, a, n * sizeof(Ent));
  free(tmp); free(cnt);
}

/* one direction of extension_correction.py:223-245 from the k1-mer `cur`; returns the steps taken */
static uint32_t extend(const Ent* e, uint64_t n, uint8_t* trav, uint64_t cur, int right, int k1, uint64_t mask, uint8_t* bases, uint64_t* totw) {
  static const int order[4] = {0, 2, 1, 3};                      /* BASES = A, G, C, T */
  uint32_t steps = 0;
  for (;;) {
    int64_t best = -1; uint32_t bw = 0; int bb = 0;
    for (int q = 0; q < 4; q++) {
      const uint64_t b = (uint64_t)order[q];
      const uint64_t cand = right ? (((cur << 2) | b) & mask) : ((cur >> 2) | (b << (2 * (k1 - 1))));
      const int64_t j = find(e, n, cand);
      if (j < 0 || trav[j]) continue;
      if (best < 0 || e[j].w > bw) { best = j; bw = e[j].w; bb = (int)b; }
    }
    if (best < 0) return steps;
    bases[steps++] = (uint8_t)bb;
    *totw += bw;
    trav[best] = 1;
    cur = e[best].key;
  }
}

uint64_t oracle_extend(const uint64_t* ckeys, const uint32_t* ccounts, uint64_t n_canon, int k1, uint32_t min_weight,
                       uint64_t* out_seed, uint32_t* out_nr, uint32_t* out_nl, uint64_t* out_totw, uint8_t* out_bases,
                       uint64_t* n_entries_out) {
  const uint64_t mask = k1 == 32 ? ~0ULL : ((1ULL << (2 * k1)) - 1);
  Ent* e = (Ent*)malloc((2 * n_canon + 1) * sizeof(Ent));
  uint64_t n = 0;
  for (uint64_t i = 0; i < n_canon; i++) {
    const uint64_t key = ckeys[i];
    if (low_complexity(key, k1)) continue;                       /* (its reverse complement is low-complexity as well) */
    const uint64_t r = rc_key(key, k1);
    uint64_t c = ccounts[i];
    if (r == key) { c *= 2; if (c > 0xFFFFFFFFULL) c = 0xFFFFFFFFULL; e[n].key = key; e[n].w = (uint32_t)c; n++; }
    else { e[n].key = key; e[n].w = (uint32_t)c; n++; e[n].key = r; e[n].w = (uint32_t)c; n++; }
  }
  sort_entries(e, n, 2 * k1);
  *n_entries_out = n;
  {
    g_shift = 2 * k1 > PBITS ? 2 * k1 - PBITS : 0;
    uint32_t* pref = (uint32_t*)calloc((1u << PBITS) + 2, sizeof(uint32_t));
    for (uint64_t i = 0; i < n; i++) pref[(e[i].key >> g_shift) + 1]++;
    for (uint64_t p = 0; p < (1u << PBITS); p++) pref[p + 1] += pref[p];
    g_pref = pref;
  }
  uint64_t ns = 0;
  for (uint64_t i = 0; i < n; i++) if (e[i].w >= min_weight) ns++;
  Seed* sd = (Seed*)malloc((ns + 1) * sizeof(Seed));
  ns = 0;
  for (uint64_t i = 0; i < n; i++) if (e[i].w >= min_weight) { sd[ns].w = e[i].w; sd[ns].idx = (uint32_t)i; ns++; }
  qsort(sd, ns, sizeof(Seed), cmp_seed);
  uint8_t* trav = (uint8_t*)calloc(n + 1, 1);
  uint64_t n_walks = 0, nb = 0;
  for (uint64_t s = 0; s < ns; s++) {
    const uint32_t i = sd[s].idx;
    if (trav[i]) continue;
    trav[i] = 1;
    uint64_t tot = e[i].w;
    const uint32_t nr = extend(e, n, trav, e[i].key, 1, k1, mask, out_bases + nb, &tot);
    const uint32_t nl = extend(e, n, trav, e[i].key, 0, k1, mask, out_bases + nb + nr, &tot);
    out_seed[n_walks] = e[i].key; out_nr[n_walks] = nr; out_nl[n_walks] = nl; out_totw[n_walks] = tot;
    nb += (uint64_t)nr + nl;
    n_walks++;
  }
  free(e); free(sd); free(trav); free((void*)g_pref); g_pref = NULL;
  return n_walks;
}
