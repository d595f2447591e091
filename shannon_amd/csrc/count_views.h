// The view of a packed read set that the counting kernels take (count.hip, count_sk.hip).
#pragma once
#include "common.h"
#include <vector>

struct ReadsView {
  const uint64_t* words; const uint64_t* mask; const uint64_t* woff; const uint32_t* len;
  uint64_t n_reads; uint32_t fixed_len, wpr, wmax, rt; int has_n;
};

// super-k-mer counting path (count_sk.hip): *handled = 0 when the input is outside what the path takes (the caller goes on with
// the partition pipeline)
int shn_count_superkmers(shn_ctx* ctx, const std::vector<ReadsView>& views, uint64_t upper, int k1, int both_strands, shn_table** out, int* handled);
