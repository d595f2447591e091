/* CPU restatement of the (K+1)-mer counting stage in plain C -- TEST INFRASTRUCTURE and the
 * `cpu_baseline` ("port") of bench.py; never linked into the product.
 *
 * Follows the contract of the reference's external counter call (shannon.py:439-441:
 * `jellyfish count -m K+1 ...; jellyfish dump -c -t -L 1` on the strand-doubled reads): exact
 * count of every ACGT-only k1-window.  Strand doubling (shannon.py:396-424) is folded in by
 * counting canonical keys (count(x) == count(rc x) in the doubled input).  Single thread:
 * emit keys, LSD radix sort (8-bit digits), run-length count.
 *
 *   codes  : n_reads x L bytes, 0..3 = A,C,G,T, anything else = non-ACGT
 *   returns number of distinct canonical keys; fills keys/counts (caller allocates n_reads*(L-k1+1))
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint64_t revcomp(uint64_t key, int k) {
  uint64_t out = 0;
  for (int i = 0; i < k; i++) { out = (out << 2) | (3 - (key & 3)); key >>= 2; }
  return out;
}

uint64_t oracle_count_canonical(const uint8_t* codes, uint64_t n_reads, uint32_t L, int k1, int canonical,
                                uint64_t* keys_out, uint32_t* counts_out, uint64_t* n_windows) {
  if ((int)L < k1) { *n_windows = 0; return 0; }
  uint64_t cap = n_reads * (uint64_t)(L - k1 + 1);
  uint64_t* a = (uint64_t*)malloc((cap + 1) * 8);
  uint64_t* b = (uint64_t*)malloc((cap + 1) * 8);
  uint64_t n = 0;
  uint64_t mask = k1 == 32 ? ~0ULL : ((1ULL << (2 * k1)) - 1);
  for (uint64_t r = 0; r < n_reads; r++) {
    const uint8_t* s = codes + r * L;
    uint64_t fw = 0, rc = 0;
    int valid = 0;
    for (uint32_t i = 0; i < L; i++) {
      uint8_t c = s[i];
      if (c > 3) { valid = 0; fw = 0; rc = 0; continue; }
      fw = ((fw << 2) | c) & mask;
      rc = (rc >> 2) | ((uint64_t)(3 - c) << (2 * (k1 - 1)));
      if (++valid >= k1) a[n++] = canonical ? (fw < rc ? fw : rc) : fw;
    }
  }
  *n_windows = n;
  int passes = (2 * k1 + 7) / 8;
  for (int p = 0; p < passes; p++) {
    uint64_t hist[257];
    memset(hist, 0, sizeof(hist));
    int sh = 8 * p;
    for (uint64_t i = 0; i < n; i++) hist[((a[i] >> sh) & 255) + 1]++;
    for (int i = 0; i < 256; i++) hist[i + 1] += hist[i];
    for (uint64_t i = 0; i < n; i++) b[hist[(a[i] >> sh) & 255]++] = a[i];
    uint64_t* t = a; a = b; b = t;
  }
  uint64_t d = 0;
  for (uint64_t i = 0; i < n;) {
    uint64_t j = i;
    while (j < n && a[j] == a[i]) j++;
    keys_out[d] = a[i];
    counts_out[d] = (uint32_t)(j - i);
    d++;
    i = j;
  }
  free(a); free(b);
  (void)revcomp;
  return d;
}
