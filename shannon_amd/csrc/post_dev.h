// Device aids of the final merge (csrc/post_gpu.hip), used by the native merge stage (csrc/post_host.hip) when it is given a
// context: the text of all_reconstructed.fasta is uploaded once; the device then (1) fingerprints every sequence line and its
// reverse complement (process_concatenated_fasta.py:26-31: "seen before, on either strand" becomes a look-up of 128-bit
// fingerprints, verified on the text) and (2) scans the surviving records for occurrences of the query r-mers of faster_reps
// (faster_reps.py:104-116: the index of every 24-mer of every record -- of which only the records' first and last r-mers, plain and
// reverse-complemented, are ever looked up: they form the query set).
#pragma once
#include "common.h"
#include <utility>
#include <vector>

struct PostDev;
// uploads the pieces back to back; byte g of the concatenation is what the offsets below refer to
int post_dev_create(shn_ctx* ctx, const std::vector<std::pair<const uint8_t*, uint64_t>>& pieces, PostDev** out);
void post_dev_destroy(PostDev* d);
// room for `cap` bytes, filled piece by piece (post_dev_upload: bytes [at, at + n) of the device text, on the stream of `on`)
int post_dev_create_cap(shn_ctx* ctx, uint64_t cap, PostDev** out);
int post_dev_upload(PostDev* d, shn_ctx* on, uint64_t at, const uint8_t* src, uint64_t n);
uint64_t post_dev_capacity(const PostDev* d);
// for n ranges (offset, length) of the text: out[4 i .. 4 i + 3] = fingerprint of the bytes (two words), of their reverse
// complement (two words; complement of A C G T, other bytes as they are)
// on: the context (stream) to run on, default the text's own; other_out[i] = 1: range i holds a byte outside ACGT
int post_dev_fingerprints(PostDev* d, const uint64_t* off, const uint32_t* len, uint64_t n, uint64_t* out, shn_ctx* on = nullptr, uint8_t* other_out = nullptr);
// occurrences of the r-mers `queries` (packed 2 bits per base, A C G T = 0 1 2 3; r < 32) in n ranges: (range, position, key),
// sorted by (range, position).  *bad: a range holds a byte outside ACGT.  SHN_ERR_OVERFLOW when there are more hits than `cap`.
int post_dev_scan(PostDev* d, const uint64_t* off, const uint32_t* len, uint64_t n, int r, const uint64_t* queries, uint64_t nq, uint64_t cap,
                  std::vector<uint32_t>& hit_range, std::vector<uint32_t>& hit_pos, std::vector<uint64_t>& hit_key, int* bad);
