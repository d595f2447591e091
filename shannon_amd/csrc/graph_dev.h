// Device-resident per-read state of a partition's graph stage (internal to the library: reads_dedup.hip, kpaths_gpu.hip,
// core.hip and mbgraph_host.hip share it).  A partition of BASELINE configs[2] has 10^5 - 4x10^6 distinct reads and the host
// stage used to carry six arrays per read through the bus and through loops of its own (copies, mate, role from the duplicate
// search; state, node, offset from the known-paths scan; first / last node for the mate pairs): 22 bytes per read each way and
// three passes over every read on a host thread -- 2.5 of the stage's 4 thread-seconds per step.  Here the arrays stay where
// they are made; the host gets the read's row and strand (5 bytes, for the text of the few reads it ever looks at) and the short
// lists it really works on (bridging hits, reads left to search, path records, mate-pair candidates).
#pragma once
#include "common.h"
#include <vector>

struct shn_dedup {                      // distinct reads of a partition (reads_dedup.hip), ids in order of first occurrence
  shn_ctx* ctx = nullptr;
  uint64_t n_slots = 0, n_distinct = 0;
  int paired = 0;
  uint32_t* d_cnt = nullptr;            // copies
  int32_t* d_mate = nullptr;            // id of the mate of the read's LAST occurrence (-1: single-end)
  uint8_t* d_role = nullptr;            // 0 none, 1 first mate, 2 second mate (of the last occurrence)
  uint32_t* d_row = nullptr;            // row of the read in its resident set
  uint8_t* d_flag = nullptr;            // bit 0: set b, bit 1: reverse complement
};
// didx (host) or d_didx (device, e.g. a slice of shn_routes): the doubled read indices of the partition's first n reads
int shn_reads_dedup_dev(shn_ctx* ctx, const shn_reads* a, const shn_reads* b, const uint32_t* didx, const uint32_t* d_didx, uint64_t n, int paired,
                        shn_dedup** out);
int shn_dedup_origin(const shn_dedup* d, uint32_t* row_out, uint8_t* flag_out);                 // -> host, n_distinct entries each
int shn_dedup_attrs(const shn_dedup* d, uint32_t* cnt_out, int32_t* mate_out, uint8_t* role_out);   // -> host (the fall-back of the host forms)
void shn_dedup_destroy(shn_dedup* d);
// the distinct reads as a read set of their own, gathered on the device from rows / flags that are already there
int shn_reads_gather_dev(shn_ctx* ctx, const shn_reads* a, const shn_reads* b, const uint32_t* d_rows, const uint8_t* d_flags, uint64_t n, shn_reads** out);

struct shn_routes;
int shn_routes_device_slice(const shn_routes* r, uint64_t lo, uint64_t n, const uint32_t** out);   // route.hip

struct KpSlow { uint32_t read, state, node, offset, cnt; };     // a read left to the host's search (state 2 / 3, see kpaths_gpu.hip)
struct shn_kp {                         // known_paths on the device (kpaths_gpu.hip): what find_mate_pairs needs afterwards
  shn_ctx* ctx = nullptr;
  uint64_t n_reads = 0, n_nodes = 0;
  int32_t *d_first = nullptr, *d_last = nullptr;      // per read: index (into the node order given) of the first / last node of its last path, -1 none
  uint32_t *d_eoff = nullptr, *d_edst = nullptr;      // out-edges of the nodes (CSR)
};
// Classifies and searches every read on the device (shn_known_paths_search's kernels, the reads whose first K-mer occurs more than
// once included).  left: the reads the host still has to search; records: [read, length, nodes ...] of the reads searched on the
// device; rec_cnt: (read, copies) of those reads.  kp stays alive for shn_kp_mate_pairs.
int shn_known_paths_dev(shn_ctx* ctx, const shn_reads* reads, int K, const uint8_t* node_bases, const uint64_t* node_off, uint64_t n_nodes,
                        const uint32_t* edge_off, const uint32_t* edge_dst, const uint32_t* edge_ov, const shn_dedup* dd, std::vector<KpSlow>* left,
                        std::vector<int32_t>* records, std::vector<uint32_t>* rec_cnt, shn_kp** kp);
// patches: (read, first, last) triples of the reads the host searched (indices into the node order, -1 none); pairs_out: (a, b) =
// (last node of a first mate, first node of its mate) with a != b and no edge a -> b, every such pair of reads (repeats included)
int shn_kp_mate_pairs(shn_kp* kp, const shn_dedup* dd, const int32_t* patches, uint64_t n_patches, std::vector<uint32_t>* pairs_out);
void shn_kp_destroy(shn_kp* kp);
