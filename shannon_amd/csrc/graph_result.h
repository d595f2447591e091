// Output of the multibridged-graph stage of one partition (csrc/mbgraph_host.hip), input of the native sparse-flow stage
// (csrc/sflow_host.hip): the content of single_nodes.txt and nodes / edges / paths{c}.txt, flattened.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

struct shn_graph {
  // flattened output_components (multibridging.py:271-325)
  std::vector<uint64_t> s_off; std::string s_bases; std::vector<double> s_cc, s_norm;
  std::vector<uint64_t> comp_node_off, comp_edge_off, comp_path_off;
  std::vector<uint64_t> n_off; std::string n_bases; std::vector<double> n_cc, n_norm; std::vector<uint8_t> n_cc_int;
  std::vector<int32_t> e_in, e_out, e_w; std::vector<double> e_cc, e_norm;
  std::vector<uint64_t> p_off; std::vector<int32_t> p_ids;
  std::vector<int32_t> info;   // nodes_after[4], final_nodes, n_known, n_mate, n_bridged_rounds, bridged...
};

