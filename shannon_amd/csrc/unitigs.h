// Result of shn_unitigs_build (csrc/graph_gpu.hip), consumed by the graph stage (csrc/mbgraph_host.hip).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

struct shn_unitigs {
  int K;
  uint32_t n_parts;
  // per partition
  std::vector<uint64_t> n_kmers;          // distinct K-mers (Node count after loading, multibridging.py:383)
  std::vector<uint8_t> cyclic;            // 1: holds a cycle of condensable edges -- build this partition on the host
  std::vector<uint64_t> node_off;         // [n_parts+1] into the node arrays (final nodes in creation order)
  std::vector<uint64_t> edge_off;         // [n_parts+1] into the edge arrays
  // final nodes
  std::vector<uint64_t> base_off;         // [n_nodes+1] into bases
  std::string bases;
  std::vector<uint32_t> n_len;            // K-mers merged into the node
  std::vector<uint32_t> n_tail_out;       // out-degree of the chain's last K-mer
  // edges between final nodes (partition-local node indices), in edge-id order; out_rank / in_rank: position in the
  // source's out-list / the destination's in-list
  std::vector<uint32_t> e_src, e_dst, e_out_rank, e_in_rank;
};

